// jg_symbolic.hpp -- host-side symbolic analysis for the batched block-sparse LU engine.
//
// Replaces (as a whole) the symbolic half of the reference's third-party factorizations
// (`lu`/`klu`/`ldlt` first call, /root/reference/src/backend/utility.jl:470-476, 486-492, 534-540):
// fill-reducing ordering, fill pattern, and -- new here -- static dependency levels and replay tables that
// the device numeric kernels walk every iteration (the `lu!`/`klu!` path, utility.jl:478-484).
//
// The matrix is an n x n BLOCK matrix with a structurally symmetric pattern (Ybus pattern for the
// Newton-Raphson Jacobian, pattern of H'H for the Gauss-Newton gain); every block is 2x2
// ((theta_i, V_i) per bus).  No numerical pivoting between blocks: the bus-block Jacobian and the
// SPD gain matrix are factorised with a static pivot order (min degree); 2x2 diagonal blocks are
// inverted exactly.  Integer work only -- nothing here touches floating point.
#pragma once
#include <cstdint>
#include <vector>

namespace jg {

// ---- environment switches: ONE reader (jg_symbolic.cpp: knob) -------------------------------------------------------------------------------------------
// A user needs six: JG_TRACE (per-iteration history of jg_nr_run / jg_gn_run on stderr), JG_PLAN_CACHE=0 (no sharing of symbolic analyses between
// handles), JG_POLL=0 (the host always blocks in the stream synchronise), JG_PLAN_THREADS (host threads of the table builder), JG_PLAN_TIMING, JG_HOST_TIMING
// (timings on stderr).  The rest are TEST HOOKS that force a code path the library also takes by itself, so that the suites can hold the variants against each
// other (tests/test_top_variants_gpu.py, tests/test_plan_cpu.py): JG_TOP_PW, JG_TOP_FUSE, JG_TOP_SYM, JG_JORDAN, JG_CHAIN_SMALL, JG_NO_PREFACTOR, JG_LANES_INPLACE,
// JG_TOP_LEVEL, JG_ROW_TASKS, JG_ORDER_CHECK, JG_TOP_PROFILE, JG_SINGLE.  Nothing else is read: the switches of retired experiments left with their kernels
// (tools/experiments/*.patch).
int knob(const char* name, int unset);      // integer value of the environment variable JG_<name>; `unset` when it is not set (or not in the lists above)
inline bool knob_set(const char* name) { return knob(name, -2147483647) != -2147483647; }

// ---- device replay tables ("wave records") -----------------------------------------------------------------
// The numeric kernels never chase pointers: every wave's work is a fixed-size 64-byte record at an address that
// follows from (segment, chunk, wave) arithmetic alone, so it is fetched with ONE scalar load and can be
// prefetched while the previous chunk (or the level barrier) is still in flight.
//   segment = items of one dependency level that share (wpi, rpw); chunk = 16 waves = 16 / wpi items;
//   record index = seg.rec_base + ((chunk * 16 + wave) * rpw + j), j < rpw.
struct Segment {
    int rec_base;      // first record
    int nchunks;       // chunks of 16 waves
    int wpi;           // waves sharing one item (update list dealt round-robin, LDS reduction)
    int rpw;           // records per wave (a wave's share of the list: FACT_T / BWD_T terms per record)
    int level;         // dependency level (1-based)
    int last;          // 1: last segment of its level (a barrier / kernel boundary follows)
    int items;         // items in the segment
    int pad;
};
constexpr int FACT_T = 4;
// The factorisation level kernel runs FACT_WAVES-wave workgroups, 16 / FACT_WAVES per 16-wave chunk of the tables (an
// item's waves, wpi <= FACT_WAVES, stay inside one workgroup).  Two independent 8-wave workgroups per CU interleave their
// phases (record fetch, operand loads, reduction) where one 16-wave workgroup walks through them in lockstep: measured
// 1.79 -> 1.59 ms (512 scenarios), 0.68 -> 0.64 ms (64); 4-wave workgroups: 1.59 / 0.80 ms (long lists get too few waves).
constexpr int FACT_WAVES = 8;   // FactRec: {kind, id, src, nterms, (a, d, b) x 4}; kind -1 = idle wave
constexpr int BWD_T = 6;    // BwdRec:  {k, bus, diag, nterms, (u_ent, u_col) x 6}; k -1 = idle wave
// ---- factorisation TASKS (plans with policy bit 50, round 4) --------------------------------------------------------------
// A level item pulls three operand blocks per update term -- Lh(i,k), D(k), U(k,j) -- through the vector memory pipe of its CU, and that
// pipe (one 1 KiB wave-load per 16 clocks), not HBM, is what the level launches of a large batch wait for (DESIGN_LOG.md 3.6).  But the items that
// become final with pivot p share operands: every term of D(p), U(p, .) and y_p with pivot k reads the SAME Lh(p,k) and D(k), every term of
// Lh(., p) the same U(k,p) and D(k).  A TASK is one 8-wave workgroup that owns a run of such items (whole pivot rows / columns, in the pivot order
// of the level): it first STAGES the shared operands of its items in LDS, already multiplied by the pivot block --
//     row items    (side 0):  slot = Lh(p,k) D(k)^-1      (right solve on the stored 2x2 LU), a term is  c -= slot * U(k,j)   resp.  slot * y_k
//     column items (side 1):  slot = D(k)^-1 U(k,p)       (left solve),                        a term is  c -= Lh(i,k) * slot
// -- two block loads per staged operand, once per task; after that a term costs ONE block load from memory, one block read from LDS
// (256 B/clk against 64 for the vector memory pipe) and 8 multiply-adds: no dsolve per term.  What is stored does not change (A = Lh D^-1 U,
// unscaled): the top tasks, the backward sweep, the forward-only tables and the selected inverse read the same factor as before.
// Tables: a segment holds the tasks of one level with the same shape (R rounds of one record per wave, the first spw of them carry staging
// entries: Segment::rpw = R, Segment::wpi = spw); record index = seg.rec_base + ((task * 8 + wave) * rpw + j).  ONE record format:
//   w0 = kind (0 entry stored raw, 2 diagonal block, 3 rhs row, 7 no item) | TK_FIRST | TK_LAST | TK_BAR | TK_SIDE | TK_DIRECT | sub << 8 | wpi << 12,
//   w1 id, w2 src, w3 = terms | staging entries << 8,
//   w4 .. w9   TASK_T x (memory operand | slot << 24)      or, TK_DIRECT, 2 x (a, d, b) as in a FactRec (terms whose shared operand found no
//              slot: rows with more than TASK_SLOTS lower entries)
//   w10 .. w15 TASK_STAGE x (operand entry | transpose << 30 | right-solve << 29, pivot entry, slot): what THIS wave stages before the task's barrier
//   An item of more than TASK_T terms is dealt over wpi = 2, 4, 8 neighbouring waves (shares end in the same round; TK_BAR marks that round for
//   EVERY wave of the task: partial sums meet in LDS, fixed order); a share of more than one record continues in the wave's next round
//   (TK_FIRST / TK_LAST bracket it).
constexpr int TASK_WAVES = 8, TASK_SLOTS = 30, TASK_T = 6, TASK_STAGE = 2, TASK_DIRECT_T = 2;
constexpr int TK_KIND = 7, TK_FIRST = 8, TK_LAST = 16, TK_BAR = 32, TK_SIDE = 64, TK_DIRECT = 128;
// Backward CHAINS: consecutive pivots k..k+b-1 of one supernode (each the parent of the previous, nested structure)
// are solved by ONE workgroup in one launch -- the dense in-chain triangle is a sequence of workgroup barriers, not
// of kernel launches.  A chain segment has wpi = 0 and ONE record per task: {b, nE, off, wpr}; at bwd_chain[off]:
//   rows[b] x {pivot, original block index, diagonal entry} | ecol[nE] external columns (pivots) |
//   uext[b][nE] entries U(row, ecol) | uin[b][b] entries U(row p, row c) for c > p (else -1)
// SMALL chains (at most CHAIN_SMALL_ROWS rows: every chain below the top; segment wpi = -1, same task data, wpr = 8 / pow2ceil(b)): eight
// waves, wave p * wpr owns row p -- its sum, its diagonal block and its in-chain blocks stay in registers, the external columns are read
// straight from the solution vector, LDS only carries the partial sums and x_c.  A third of the latency of the general task, and a level
// whose chains are all small runs the 8-wave kernel (two workgroups per CU) instead of the 16-wave one with 125 KB of LDS.
constexpr int CHAIN_MAX_ROWS = 32, CHAIN_MAX_EXT = 72, CHAIN_SMALL_ROWS = 8;
struct Rec { int w[16]; };

// ---- multifrontal TOP of the elimination tree ---------------------------------------------------------------------
// Above a dependency level the tree is a handful of long chains with few items per level: per-level launches there cost a
// kernel boundary + a cold record fetch + a round of operand loads each (7.7 us at 64 scenarios, 9-20 us at 512) for a
// handful of blocks.  Those pivots are factorised by TOP TASKS instead: a task = a connected piece of the tree with one root
// (a path k0 .. k0+m-1 with parent(k) = k + 1 first; siblings and their descendants while the front has room), worked on by
// ONE workgroup PER SCENARIO -- lanes run across the dense front
// (m pivots in ascending order + e external rows / columns, e = |struct(root)|, + the rhs as one more column), which lives in the
// REGISTERS of a 16 x 16 thread grid (thread (i mod 16, c mod 16) owns block (i, c): `cls` x `cls` blocks per thread); a
// pivot step publishes the next pivot row / column through LDS and costs one workgroup barrier (jg_engine.hip: k_fact_top).
// Tasks talk multifrontally: a task leaves its e x (e + 1) update matrix | vector on a scenario-major stack, its parent adds
// it into its own front (extend-add), so the 3-block-reads-per-term pull of the level kernel disappears for every term whose
// pivot is in a task (half to three quarters of all terms on transmission grids).
// Contributions of BOTTOM pivots (all others) to task-owned entries still arrive through level items: those items carry
// only the bottom terms of the entry and store the partial sum raw (also for diagonal blocks).
//   header (one 64-byte record per task; level-major; a launch = one level, compiled for its widest front):
//     w0 m, w1 e, w2 root pivot, w3 offset of the task's data in top_data, w4 stack offset of its update block (doubles, -1: root),
//     w5 children, w6 offset of the pivot list, w7 offset of the child records, w8 offset of the diagonal entries (all relative
//     to w3), w9 class (2, 3, 4: the front has at most 16 class rows and columns), w10 task level, w11 f + 1,
//     w12 log2 of the scenario interleave of the task's update block on the stack (= w13 of its parent), w13 log2 G of the task's own
//     geometry (below), w14 first block of the task's JORDAN rows in the factor storage (-1: none; below)
//   data: entry map [f][f + 1] (see build_top), diagonal entries [m], child records {stack offset, e_c, inv[f + 1]}, pivots [m]
// GROUPED tasks (plans with a "mid" policy, large batches).  One workgroup per scenario is the right shape for the dense top (fronts of
// 30-60 block rows), but a (task, scenario) workgroup costs ~10 us + ~1 us per pivot whatever its front, and its loads touch 16 bytes
// of every 1 KiB line of the batch-minor storage.  Below the top the fronts are small (5-30 block rows) and there are many of them, so
// a workgroup takes G = 4 or 16 CONSECUTIVE scenarios there: the 256 threads form G grids of T x T threads (T = 8 / 4: fronts of up
// to 31 / 15 block rows at 4 x 4 blocks per thread), a load instruction covers G x 16 contiguous bytes, and a pivot step of the
// workgroup advances G scenarios (k_fact_grp).  Tasks of all geometries exchange update blocks through the same stack; the block of
// a task is interleaved over the G scenarios of its PARENT's workgroup ([scenario / G][stack][scenario % G] in 16-byte units), so the
// extend-add of a grouped task is coalesced as well.  A launch of grouped tasks holds every G > 1 task of one task level: workgroup x
// of a 64-scenario group finds its (task, scenario block) in top_wgmap.
// JORDAN rows (plans with policy bit 49, no grouped tasks).  The backward sweep over the top used to be the one sequential
// piece left in the solve: a chain of m pivots is m dependent steps (one workgroup barrier each, jg_engine.hip: bwd_chain_task), and the
// top of a transmission grid is ~120 pivots deep.  A task has its pivot rows in registers anyway, so it eliminates each pivot column
// ABOVE the diagonal as well (Gauss-Jordan inside the task: rows i < q get  row_i -= U(i,q) D(q)^-1 row_q  in the very bulk update that
// serves the rows below, which touches every block of the thread grid whether it needs to or not).  What leaves the task for pivot row
// i is then  J(i, .) = the row over the EXTERNAL columns of the front only  and  y'_i, with
//     x_i = D(i)^-1 (y'_i - sum_{c in ext(task)} J(i,c) x_c):
// every row of a task depends on pivots of ANCESTOR tasks only, the rows of a task are one backward level of plain wave records, and the
// depth of the sweep over the top falls from its pivots to its task levels.  J is dense (m x e blocks per task) where U(i, ext) is not,
// so it lives behind the factor entries: block w14 + i * e + c with w14 >= n_entries.  The in-task triangle of U keeps the multipliers U(i,q) as
// they stood when column q was eliminated.  L, D and the update matrices are untouched: the forward elimination of another right-hand
// side (Engine::forward) still produces y, not y' -- users of that path (iterative refinement, fast Newton-Raphson) switch the engine
// back to the plain sweep (Engine::jordan = false: plain rows from the tasks, the chain tables below).
constexpr int TOP_FRONT_MAX = 63;       // m + e of a task: 64 columns with the rhs = class 4 on the 16 x 16 thread grid
struct TopLaunch { int task_begin, ntasks, cls, level, grouped, wg_begin, nwg, pad; };   // grouped: wgmap[wg_begin .. wg_begin + nwg) = task << 8 | scenario block

struct BlockSymbolic {
    int n = 0;
    std::vector<int> perm;              // perm[k]  = original block index eliminated k-th
    std::vector<int> iperm;             // iperm[i] = pivot position of original index i
    // Factor entries (pattern of L+D+U in pivot numbering), ids are row-major over (row, sorted col)
    int n_entries = 0;
    std::vector<int> row_ptr;           // [n+1] entries of pivot-row r
    std::vector<int> e_col;             // column (pivot numbering)
    std::vector<int> e_row;
    std::vector<int> e_src;             // index into the caller's block CSR (original order), -1 = fill
    std::vector<int> e_diag;            // for lower entries (row>col): entry id of D(col,col); else -1
    std::vector<int> diag;              // [n] entry id of D(k,k)
    // Factorisation A = Lh * inv(D) * U with UNSCALED lower blocks Lh (so a pivot's diagonal inverse,
    // its U row and its Lh column all become final in the same dependency level):
    //   entry e accumulates  - X[t_a] * X[t_d] * X[t_b]   (Lh(i,k) * Dinv(k) * U(k,j))  for t in [t_ptr[e], t_ptr[e+1])
    std::vector<int> t_ptr, t_a, t_d, t_b;
    std::vector<int> e_level;           // dependency level of each entry (1-based)
    // Row lists (pivot numbering): strictly-lower and strictly-upper entries of pivot row k
    std::vector<int> l_ptr, l_ent, l_col;   // forward elimination fused into the factorisation:
                                            //   y_k = f_k - sum_c Lh(k,c) Dinv(c) y_c      (item id n_entries + k)
    std::vector<int> u_ptr, u_ent, u_col;   // x_k = Dinv_k (y_k - sum_c U(k,c) x_c)
    std::vector<int> y_level, bwd_level;
    long long n_terms = 0;
    long long n_sched_terms = 0;        // update terms the factorisation launches actually execute (entries + rhs rows; half of
                                        // the entry terms in symmetric mode)
    // replay tables (see above).  policy bit 0 ("in place"): the caller assembles its blocks straight into the factor
    // storage (entry src_entry[p] for block p), so off-diagonal entries without update terms need no work at all and
    // are not scheduled; FactRec.src then names the entry itself.
    int inplace = 0;
    int symmetric = 0;                  // policy bit 1: the matrix is symmetric, Lh(i,k) = U(k,i)' is read through the upper entry
    // policy bit 2 ("the producer finishes level 0", in-place plans only): a pivot whose diagonal block receives no update term
    // -- no earlier pivot touches it: the leaves of the elimination tree, half of the pivots of a transmission grid -- needs
    // nothing from the factorisation but the 2x2 LU of its block as assembled and y_k = rhs_k.  A producer that has the block
    // in registers anyway (the Jacobian assembly) stores it FACTORISED and writes y_k itself (pre_pivot names those pivots);
    // those items then sit at dependency level 0, every other level moves down by one, and the first (widest) level launch
    // disappears.  For producers that deliver plain blocks the same items form the PRE tables, run ahead of level 1
    // (Engine::factor(..., level0_done = false)).
    int prefactor = 0;
    int top_split = 0;                  // policy bit 3: one top launch per (task level, class) instead of one per level
    std::vector<char> pre_pivot;        // [n] pivot k: D(k) and y_k are level-0 items
    std::vector<Segment> pre_seg; std::vector<Rec> pre_rec; int n_pre_levels = 0;
    std::vector<int> src_entry;         // [nnz of the caller's pattern] -> entry id
    std::vector<Segment> fact_seg, bwd_seg, fwd_seg;    // fwd: the forward elimination alone (rhs rows of the fact tables)
    std::vector<Rec> fact_rec, bwd_rec, fwd_rec;
    int fact_tasks = 0;                 // policy bit 50: fact_seg / fact_rec hold factorisation TASKS (above), not wave records of single items
    int task_rounds = 0;                // ... and a task is filled up to this many rounds of 8 shares
    long long n_staged = 0;             // operands staged by the tasks (statistics)
    long long n_direct_terms = 0;       // terms of task items that kept the three-operand form
    int jordan = 0;                     // policy bit 49 and the plan qualifies: the top tasks can leave Jordan rows (see TopLaunch)
    int n_jordan = 0;                   // blocks of Jordan rows behind the n_entries factor entries
    std::vector<Segment> bwdj_seg;      // the backward sweep over Jordan rows (top pivots: wave records over ext(task); chains only below the top)
    std::vector<Rec> bwdj_rec;
    int n_bwdj_levels = 0;
    std::vector<int> bwd_chain;         // chain task data (see CHAIN_MAX_ROWS)
    std::vector<int> chain_level;       // [n] backward level of the chain (or single row) a pivot belongs to
    int n_fact_levels = 0, n_bwd_levels = 0, n_fwd_levels = 0;
    std::vector<Segment> sel_seg;       // selected inverse (built on demand: build_selected_inverse)
    std::vector<Rec> sel_rec;
    int n_sel_levels = 0;
    // multifrontal top (see TopLaunch): empty when the plan has no top tasks
    int top_level = 0;                  // pivots whose diagonal becomes final at this level or later belong to top tasks (0: none)
    int want_single = 0;                // policy bit 60: also build the factorisation tables of a single instance (below)
    bool single_fact_ok = false;
    std::vector<Rec> f_rec;             // records of f1 and f2
    std::vector<int> f1_first, f1_wg, f2_first;
    int n_f1_wg = 0;
    int top_mmin = 8;                   // policy bits 54-59: a task takes at least this many pivots where its front has room (0 = default 8)
    std::vector<int> top_task_of;       // [n] task index of a pivot, -1 = bottom pivot
    std::vector<Rec> top_task;          // task headers, launch order
    std::vector<int> top_data;
    std::vector<TopLaunch> top_launch;
    std::vector<int> top_wgmap;         // grouped launches: workgroup -> (task, scenario block of its 64-scenario group)
    long long top_stack = 0;            // doubles per scenario on the update stacks
    long long top_stack_cls[3] = {0, 0, 0};   // ... of the blocks interleaved over 1 / 4 / 16 scenarios (a task's w4 is an offset inside its class)
    long long top_terms = 0;            // update terms executed inside tasks
};

// ---- shared-factor solve (compensation, jg_comp.hip): tables of x = A^-1 r for MANY right-hand sides on ONE numeric factor ----------
// When every scenario of a batch starts from one state (an N-1 screen from the base-case solution: the reference's user loop
// updateBranch! -> powerFlow!, /root/reference/src/powerSystem/branch.jl:453-459 with lu! + ldiv! of acPowerFlow.jl:890-897), the first Newton
// matrices differ from ONE matrix by a rank <= 4 term per scenario: the step is then a solve with the SHARED factor of that matrix
// (values wave-uniform: scalar loads; only the right-hand sides are batch-minor) plus a 4 x 4 correction per scenario.
// The factor being constant, the sequential top of the elimination tree is replaced by the explicit dense inverse of its Schur
// complement (one GEMM-shaped launch instead of ~60 dependency levels), and only the wide bottom levels stay level launches:
//   forward  (bottom rows, by level): y_k = r_k - sum_c [Lh(k,c) D(c)^-1] y_c;  top rows collect their bottom terms only -> yt_t (row n + t)
//   top:                              x_T = Sinv * yt_T
//   backward (bottom rows, by level): x_k = D(k)^-1 y_k - sum_c [D(k)^-1 U(k,c)] x_c
// Record (both sweeps): w0 target row of W (pivot k, or n + t for the partial row of top pivot T[t]), w1 rhs row / bus (original index of the
// pivot), w2 diagonal entry (backward; -1 forward), w3 terms, then COMP_T x (factor entry, W row).  top_cap < 0: no top (every pivot a level
// item: the bootstrap solver that forms Sinv).
constexpr int COMP_T = 6;
struct CompTables {
    int n_top = 0, split = 0;           // pivots of the dense top; forward level above which a pivot belongs to it
    std::vector<int> top;               // [n_top] pivots, ascending
    std::vector<int> tpos;              // [n] index in top, -1 = bottom pivot
    std::vector<Segment> fwd_seg, bwd_seg;
    std::vector<Rec> fwd_rec, bwd_rec;
    int n_fwd_levels = 0, n_bwd_levels = 0;
};
// top_cap: at most this many pivots in the dense top (the split is the lowest forward level that satisfies it); < 0: none.
void build_comp_tables(const BlockSymbolic& S, int top_cap, CompTables& out);

// ---- backward sweep of a SINGLE instance (round 6, jg_engine.hip: k_bwd1_top / k_bwd1_bottom) -------------------------------------------
// One scenario leaves 63 of a wave's 64 lanes idle in the level kernels, and its sweep is 15 dependent launches of ~4.7 us whatever they hold
// (a Jordan plan of the 10k-bus grid: 8 task levels + 5 levels below the top).  Here the lanes are ROWS:
//   top rows (pivots of the top tasks, Jordan rows over ext(task)): ONE workgroup walks their levels (root task first) with a workgroup barrier
//     between levels; four lanes share a row, the solution of the top lives in LDS (slot = position of the row in t_row);
//   bottom rows: a workgroup takes a run of whole bottom SUBTREES (a bottom pivot whose parent is a top pivot, with its descendants: contiguous
//     in the postorder), one thread per row -- every operand that does not depend on the sweep (blocks, right-hand side, the top's solution,
//     state-update operands) is requested before the first barrier, then the rows of a level finish between two barriers, at most
//     TOP_LEVEL_MIN of them.
// Rows (ints): t_row [.][4] = pivot, bus (original index), diagonal entry, terms; t_ptr [.][2] = first block of the row in the COMPACT Jordan rows (what k_fact_top
//              leaves for a single instance: block j of the plan at 4 (j - n_entries) doubles, the e blocks of a row side by side), first slot of its column list;
//              t_term = slots of the columns, one list per task (its rows share ext(task)).
//              b_row [.][6] = pivot, bus, diagonal entry, terms, first term, level inside its subtree (0: top columns only);
//              b_term [.][2] = entry, column: >= 0 a top pivot (row of W), < 0: -(1 + thread of the column's row in this workgroup).
// The factorisation below the top of a SINGLE instance (policy bit 60, k_fact1 in jg_engine.hip): the level launches of a handful of scenarios give a wave to
// every item and hold one live lane each; here a THREAD takes an item.  The items of the bottom pivots (D(p), U(p, .), Lh(., p), y_p of every pivot outside the top
// tasks) depend on items of their own subtree only, so a workgroup takes a run of whole bottom subtrees and walks its levels with workgroup barriers: ONE launch
// (f1); the partial sums of the task-owned entries / rhs rows (bottom terms only, jg_symbolic.cpp: build_tables) depend on bottom items alone: a second, flat
// launch (f2).  Five level launches become two.
//   record (16 ints): kind (0 / 1 raw entry, 2 diagonal block, 3 rhs row), id, src, word 3, then up to four terms (a, d, b) as in a FactRec.  The FIRST record of item j is
//   record j (word 3 = terms of the item | first continuation record << 10); an item of more than four terms continues in records behind all first ones (words 4 .. 15
//   only).  f1_wg [n_f1_wg][SINGLE_FACT_LEVELS + 1]: the workgroup's items of level l + 1 are f1_wg[w][l] .. f1_wg[w][l + 1]; the partial items follow the bottom items
//   (f1_first / f2_first: the item numbers, kept for the CPU replay).
constexpr int SINGLE_FACT_LEVELS = 8;      // most levels below the top such a plan may have
constexpr int SINGLE_FACT_ITEMS = 192;     // items a bottom workgroup is filled up to (256 threads = 64 quads, four lanes per item: most levels in one round)
constexpr int SINGLE_BOTTOM_ROWS = 256;   // threads of a bottom workgroup = most rows of ONE subtree
constexpr int SINGLE_BOTTOM_FILL = 128;   // rows a bottom workgroup is filled up to with whole subtrees (more, smaller workgroups: more requests in flight)
struct SingleTables {
    bool ok = false;                    // false: the plan does not qualify (no Jordan rows, a subtree above SINGLE_BOTTOM_ROWS rows, ...)
    int n_top = 0, n_top_levels = 0;
    std::vector<int> t_row, t_ptr, t_term, t_level;     // t_level [n_top_levels + 1]
    // the same rows with the TERMS as lanes (k_bwd1_top2): the compact Jordan blocks of a level are contiguous and in (row, column) order, so term g of the plan is
    // one lane's work -- block g, the slot of its column (t_cslot[g]) -- and a row sums its products from LDS (t_toff[row]: its first term inside its level)
    bool flat_ok = false;                               // false: the blocks of some level are not contiguous in row order (k_bwd1_top sweeps instead)
    std::vector<int> t_jb, t_cslot, t_toff;             // t_jb [n_top_levels][2]: first compact block of the level, its terms; t_cslot [n_jordan]; t_toff [n_top]
    int max_level_terms = 0, max_level_rows = 0;
    int n_bottom = 0, n_wg = 0, b_levels = 0;
    std::vector<int> b_wg;                              // [n_wg + 1][2]: first row, levels of the workgroup
    std::vector<int> b_row, b_term;
};
void build_single_tables(const BlockSymbolic& S, SingleTables& out);

// pattern: CSR (rowptr[n+1], col[nnz]) 0-based, must contain the diagonal and be structurally
// symmetric. policy bit 0: in-place factor storage (see BlockSymbolic::inplace); bit 1: symmetric VALUES (LDL' by
// reading U(k,i)' for Lh(i,k): half the update terms; only the blocks on and above the diagonal must be assembled).
// policy bit 3: top launches per (level, class) (large batches).
// policy bits 8-15: dependency level from which pivots go to top tasks (0 = default: where the level schedule gets narrow,
// 255 = no top tasks); bits 16-23: soft cap of a task's front (0 = default); bits 24-30: what "narrow" means, in units of 8
// items per level (0 = default, 127 = no limit); bits 4-7: at most this many pivots per level in the top (0 = any; 13 / 14 / 15 = 18 / 24 / 36).
// policy bits 32-39 ("mid", 0 = off): pivots with at least this many neighbours at elimination -- and their ancestors -- go to tasks
// wherever they sit in the tree, and tasks get a geometry by the size of their front (GROUPED tasks, above); bits 40-47: a task's
// geometry is chosen so that it can take at least this many pivots (0 = default 6); bit 48: a task only absorbs pivots that need
// its geometry (smaller fronts form tasks of their own below it).
// policy bit 49: Jordan rows for the pivots of the top tasks + a second set of backward tables over them (see TOP_FRONT_MAX above).
// policy bit 50: the factorisation tables are TASKS (see TASK_WAVES above); bits 51-53: rounds a task is filled up to (0 = default 3).
// policy bit 60: the plan serves ONE scenario: the factorisation tables of a single instance are built as well (SINGLE_FACT_LEVELS above).
// policy bits 54-59: least number of pivots a top task takes where its front has room (0 = default 8; a handful of scenarios: 12 -- a task costs ~10 us of
// gather / extend-add / store whatever it eliminates, and a lone workgroup steps through a 50-row front as fast as through a 30-row one).
// Returns 0, or 1 on a malformed pattern.
int analyze(int n, const int* rowptr, const int* col, long long policy, BlockSymbolic& out);
// Replay tables of the selected inverse of a SYMMETRIC matrix on the factor pattern (see jg_symbolic.cpp); idempotent.
void build_selected_inverse(BlockSymbolic& S);
// entry id of block (r, c) in pivot numbering, -1 if outside the factor pattern
int entry_of(const BlockSymbolic& S, int r, int c);

}  // namespace jg
