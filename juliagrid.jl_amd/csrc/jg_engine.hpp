// jg_engine.hpp -- batched block-sparse LU / triangular-solve engine on the device (gfx950).
//
// Numeric half of the factorization seam of the reference
// (/root/reference/src/backend/utility.jl:478-484 `lu!`/`klu!`, :576-586 `ldiv!`), for B
// independent scenarios at once.  Layout is batch-minor with the components of an item interleaved per scenario:
//     block entry e, component c = 2h + c', scenario b:  X[((e * 2 + h) * ld + b) * 2 + c']   (block row h = 16 bytes per scenario)
//     2-vector row k, component c, scenario b:           W[(k * ld + b) * 2 + c]             (16 bytes per scenario)
// so the 64 lanes of a wavefront work on the same structural item of 64 scenarios and move it with 16-byte-per-lane
// instructions (global_load/store_dwordx4), each covering 1 KiB of CONTIGUOUS memory (a [e][ld][4] interleave would
// make every instruction touch 2 KiB at half density: measured 0.16 -> 0.25 ms on the assembly's write stream).  Why 16 bytes: a CU retires one vector
// memory instruction per ~16 clocks whatever its width (measured, tools/microbench/ldrate.hip: 6.8 ns per wave-instruction
// for 8 B and for 16 B per lane), and the narrow dependency levels of the LU are bound by exactly that issue rate,
// so a 2x2 block costs 2 instructions instead of 4.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "jg_symbolic.hpp"

namespace jg {

void set_last_error(const std::string& msg);   // thread-local text behind jg_last_error()
}
struct jg_comm;
namespace jg {
int comm_allgather(jg_comm* c, const double* send, double* recv, size_t count, hipStream_t st);   // jg_comm.cpp: ncclAllGather on st, not synchronised
int comm_rank(const jg_comm* c);
int comm_device(const jg_comm* c);
std::mutex& capture_mutex();                   // one graph capture at a time per process (host threads must not interleave captures)

// one per-level launch: grid.y walks the level's segments (at most one per wpi class 1, 2, 4, 8, 16)
struct DevLaunch {
    int seg_begin, seg_end;     // range in the segment table
    int grid;                   // most chunks of any of its segments (workgroups along x before the group stride)
    int nseg;
    int chain;                  // 1: the level holds backward chain tasks (larger LDS staging)
    int wpi_max;                // most waves per item of any of its segments
};

// Optional state update fused into the backward solve (NR: x <- x - dx, masked by bus flags).
struct StateUpdate {
    double* va = nullptr;          // [n][ld]
    double* vm = nullptr;          // [n][ld]
    const signed char* flags = nullptr;  // [n] bit0: angle is a state, bit1: magnitude is a state
    const int* active = nullptr;   // [ld] 1 = apply update for this scenario
    double sign = 0.0;             // -1 (Newton-Raphson) / +1 (Gauss-Newton)
};

// Which 64-scenario groups a launch works on.  flags (nullable): [ld/64], a group whose flag is 0 is skipped.
// list/count (nullable): the active groups are list[0 .. *count) -- the launch then spreads exactly those over the
// chip.  Both live in device memory so captured graphs follow the per-iteration verdicts.
struct GroupSel {
    const int* flags = nullptr;
    const int* list = nullptr;
    const int* count = nullptr;
};

// Workgroup -> (scenario group, work chunk) with the group as the FAST index of a 1-D grid.  Workgroup b lands on
// XCD b % 8 (MI355X_MICROARCH.md, dispatch), so with 8 | groups every XCD -- and its private 4 MiB L2 -- serves only
// the groups congruent to it: operands re-read by many items of one group stay in ONE L2 instead of eight.
// groups < 8 are rounded to a power of two so a group still maps to a fixed subset of XCDs.  More than 8 groups: a
// multiple of 8 keeps the pinning (group g on XCD g % 8); any other count is NOT padded (padding 9 groups to 16 slots
// gave one XCD two groups and the others one: 1.79 -> 2.83 ms) -- workgroup (x, slot) then lands on XCD
// (x * groups + slot) % 8, which rotates a group over the XCDs and balances them.
// Every other count takes the BALANCED mapping below (groups_balanced), so what this function still decides is 1, 2, 4 and the multiples of 8.
__host__ __device__ inline int group_stride(int groups) {
    return groups >= 8 ? groups : (groups <= 1 ? 1 : (groups <= 2 ? 2 : (groups <= 4 ? 4 : 8)));
}
// BALANCED mapping (round 5, JG_BALANCED_MAP below): every group count that cannot be pinned -- not 1, 2, 4 and no multiple of 8 -- takes the (group, chunk) pairs of a launch in
// group-major order, cuts that list into EIGHT equal runs and gives run k to the workgroups that land on XCD k (id % 8 == k): every XCD gets the same number of workgroups whatever
// the count, and a group's operands meet in the one to three L2s its run(s) touch instead of all eight (the rotation) -- 640 lanes = 10 groups is what a rank of the 8-GPU run
// merges at the driver's K = 20, 5 - 7 groups what a 512-lane batch compacts to.
#ifndef JG_BALANCED_MAP
#define JG_BALANCED_MAP 1
#endif
__host__ __device__ inline bool groups_balanced(int groups) { return JG_BALANCED_MAP && groups > 2 && groups != 4 && (groups & 7) != 0; }
// workgroups of a launch of nx chunks per group (host side; the count of a handle's lanes bounds every count its launches can see after compaction)
__host__ inline unsigned grid_blocks(int groups, long long nx) {
    return groups_balanced(groups) ? (unsigned)(((long long)groups * nx + 7) / 8 * 8) : (unsigned)(nx * group_stride(groups));
}

#ifdef __HIPCC__
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 2x2 block / 2-vector of one scenario: two (one) 16-byte accesses
struct Blk { double v00, v01, v10, v11; };
// Addresses are formed as (wave-uniform base of the item) + (32-bit byte offset of the lane): the item index comes from a record in scalar
// registers, so the base is scalar arithmetic and every access of a wave shares ONE offset register (global_load ... v_off, s[base] instead of a
// 64-bit address pair computed per access on the vector ALU).  A batch row is at most 2^32 bytes: ld < 2^28 scenarios.
__device__ __forceinline__ Blk load_blk(const double* base, size_t item, size_t b, size_t ld) {
    const char* p = (const char*)base + item * ld * 32;
    const unsigned off = (unsigned)b * 16u;
    const double2 r0 = *(const double2*)(p + off), r1 = *(const double2*)(p + ld * 16 + off);
    return Blk{r0.x, r0.y, r1.x, r1.y};
}
__device__ __forceinline__ void store_blk(double* base, size_t item, size_t b, size_t ld, double v00, double v01, double v10, double v11) {
    char* p = (char*)base + item * ld * 32;
    const unsigned off = (unsigned)b * 16u;
    *(double2*)(p + off) = double2{v00, v01}; *(double2*)(p + ld * 16 + off) = double2{v10, v11};
}
__device__ __forceinline__ void store_blk_nt(double* base, size_t item, size_t b, size_t ld, double v00, double v01, double v10, double v11) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    char* p = (char*)base + item * ld * 32;       // write-once stream: nontemporal, two 16-byte stores of 1 KiB per wave each
    const unsigned off = (unsigned)b * 16u;
    __builtin_nontemporal_store(d2{v00, v01}, (d2*)(p + off));
    __builtin_nontemporal_store(d2{v10, v11}, (d2*)(p + ld * 16 + off));
}
__device__ __forceinline__ double2 load_vec(const double* base, size_t item, size_t b, size_t ld) {
    return *(const double2*)((const char*)base + item * ld * 16 + (unsigned)b * 16u);
}
__device__ __forceinline__ void store_vec(double* base, size_t item, size_t b, size_t ld, double v0, double v1) {
    *(double2*)((char*)base + item * ld * 16 + (unsigned)b * 16u) = double2{v0, v1};
}

// Hand-placed requests (jg_engine.hip: fact_record explains why): `global_load_dwordx4 / x2  dst, lane offset, scalar base`.  The base must be
// wave-uniform (it lands in a scalar register pair); the caller waits with an explicit s_waitcnt tied to the destinations.
typedef double d2v __attribute__((ext_vector_type(2)));
struct BlkV { d2v r0, r1; };
#ifndef JG_LOAD_POLICY
#define JG_LOAD_POLICY ""               // probe builds: -DJG_LOAD_POLICY='" nt"' (streaming), '" sc1"', ...
#endif
__device__ __forceinline__ void gload16(d2v& dst, const void* base, unsigned off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" JG_LOAD_POLICY : "=v"(dst) : "v"(off), "s"(base) : "memory");   // "memory": the compiler must not move its stores across (the waits count them)
}
__device__ __forceinline__ void gload8(double& dst, const void* base, unsigned off) {
    asm volatile("global_load_dwordx2 %0, %1, %2" JG_LOAD_POLICY : "=v"(dst) : "v"(off), "s"(base) : "memory");
}

// A diagonal block whose eliminated form is smaller than PIVOT_EPS x (largest entry ITS ROW of the block started from) marks the
// scenario (status bit 2 -> per-scenario status 3): a block that cancelled to rounding level is the signature of a structurally
// singular matrix (an islanding outage) under a static pivot order.  Row-wise because a gain block mixes |V|- and theta-scaled rows.
constexpr double PIVOT_EPS = 0x1p-36;       // 1.5e-11: far below any legitimate Schur complement of a power grid, far above rounding
__device__ __forceinline__ double2 row_max(const Blk& c) { return double2{fmax(fabs(c.v00), fabs(c.v01)), fmax(fabs(c.v10), fabs(c.v11))}; }
// 1 / x without the IEEE division sequence (v_rcp_f64 + two Newton steps: 5 dependent operations and 3 registers instead of ~14
// and a dozen; the result is within an ulp or two, which only perturbs the stored pivot factors at rounding level -- the
// factorisation stays the exact product of what is stored).  0 -> inf -> NaN and NaN -> NaN, both caught by the pivot check.
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// 2x2 LU with in-block partial pivoting in the stored form {1/u11, u12, l (+ 4 if the rows were swapped), 1/u22}: what a
// diagonal item of the level kernels stores, and what a producer stores for the pivots of a prefactor plan (same arithmetic:
// the two paths are bitwise interchangeable).
__device__ __forceinline__ Blk diag_lu(const Blk& c, double2 ref, bool& bad) {
#pragma clang fp contract(off)                  // two kernels inline this: what is fused is written as fma(), nothing else may be (l + 4.0 was)
    const bool sw = fabs(c.v10) > fabs(c.v00);
    const double u11 = sw ? c.v10 : c.v00, u12 = sw ? c.v11 : c.v01;
    const double o21 = sw ? c.v00 : c.v10, o22 = sw ? c.v01 : c.v11;
    const double iu11 = rcp_nr(u11);
    const double l = o21 * iu11;
    const double u22 = fma(-l, u12, o22);      // explicit: the assembly kernel and the level kernel must contract the same way
    const double iu22 = rcp_nr(u22);
    const double f1 = PIVOT_EPS * (sw ? ref.y : ref.x), f2 = PIVOT_EPS * (sw ? ref.x : ref.y);
    bad = !(fabs(u11) > f1) || !(fabs(u22) > f2) || !(fabs(iu11) < 1.0e300) || !(fabs(iu22) < 1.0e300);
    return Blk{iu11, u12, sw ? l + 4.0 : l, iu22};
}

// blockIdx.x -> (scenario group g, chunk x) of an nx-chunk launch; false: nothing to do for this workgroup.
// The group list and its length sit on the critical path of EVERY level launch (a workgroup cannot fetch its record before it
// knows its group), so they are read through the scalar cache, the count and the first eight list entries in parallel (the
// list is padded to eight entries; both are written by an earlier kernel and constant within a launch): one round trip
// instead of two dependent vector loads + readfirstlane -- 0.88 -> 0.78 ms per iteration graph of a single instance.
typedef const int __attribute__((address_space(4)))* SelIntPtr;
typedef int SelInt8 __attribute__((ext_vector_type(8)));
typedef const SelInt8 __attribute__((address_space(4)))* SelInt8Ptr;
__device__ __forceinline__ bool map_block(const GroupSel& sel, int ld, int nx, int& g, int& x) {
    const int id = blockIdx.x;
    int gact = ld / 64;
    SelInt8 first{};
    if (sel.list) { first = *(SelInt8Ptr)sel.list; gact = *(SelIntPtr)sel.count; }
    int slot;
    if (groups_balanced(gact)) {
        const int total = gact * nx, chunk = (total + 7) >> 3;
        const int j = id >> 3, flat = (id & 7) * chunk + j;
        if (j >= chunk || flat >= total) return false;
        slot = flat / nx;
        x = flat - slot * nx;
    } else {
        const int gs = group_stride(gact);
        slot = id % gs;
        x = id / gs;
        if (slot >= gact || x >= nx) return false;
    }
    if (sel.list) {
        if (gact <= 8) {
            g = first[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) g = slot == k ? first[k] : g;
        } else g = ((SelIntPtr)sel.list)[slot];
    } else g = slot;
    return !(sel.flags && !((SelIntPtr)sel.flags)[g]);
}
#endif

// The symbolic analysis of one pattern under one policy and the DEVICE copy of its replay tables: immutable once built, shared by every
// engine that factorises that pattern on that device (the handles of a ContingencyPipeline, a second analysis of the same grid, the
// rebuild of an analysis whose pattern did not change) through a process-wide cache -- the reference pays its symbolic analysis once per
// `newtonRaphson()` call and optimised exactly that (docs/src/background/releasenotes.md:7-9).
struct SharedPlan {
    BlockSymbolic S;
    int device = 0;
    long long policy = 0;
    std::vector<int> key_rowptr, key_col;          // the pattern the plan was built for (cache hits are verified, not trusted to a hash)
    unsigned long long key_hash = 0;
    Rec* fact_rec = nullptr; Rec* bwd_rec = nullptr; Rec* pre_rec = nullptr; Rec* fwd_rec = nullptr; Rec* sel_rec = nullptr; Rec* top_task = nullptr;
    Segment* fact_seg = nullptr; Segment* bwd_seg = nullptr; Segment* pre_seg = nullptr; Segment* fwd_seg = nullptr; Segment* sel_seg = nullptr;
    int* pre_row = nullptr; int* bwd_chain = nullptr; int* top_data = nullptr; int* top_wgmap = nullptr;
    Rec* bwdj_rec = nullptr; Segment* bwdj_seg = nullptr;    // Jordan plans: the backward sweep over Jordan rows (jg_symbolic.hpp)
    std::mutex sel_mutex;                           // the selected-inverse tables are built on first use
    bool sel_ready = false;
    // single-instance sweep (jg_symbolic.hpp: SingleTables), built by the first engine of ONE scenario that takes this plan
    std::mutex single_mutex;
    bool single_tried = false;
    SingleTables single;                            // host copy (counts; the index vectors are released after the upload)
    int* s1_t_row = nullptr; int* s1_t_ptr = nullptr; int* s1_t_term = nullptr; int* s1_t_level = nullptr;
    int* s1_b_wg = nullptr; int* s1_b_row = nullptr; int* s1_b_term = nullptr;
    int* s1_t_jb = nullptr; int* s1_t_cslot = nullptr; int* s1_t_toff = nullptr;    // the top rows with the terms as lanes (SingleTables::flat_ok)
    Rec* f1_rec = nullptr; int* f1_wg = nullptr;   // plans with policy bit 60: the single-instance factorisation below the top (records; the workgroups' item ranges by level)
    ~SharedPlan();
};
// (n, pattern, policy, current device) -> plan; analysis + upload on a miss.  st: stream for the uploads.  nullptr + error on failure.
std::shared_ptr<SharedPlan> acquire_plan(int n, const int* rowptr, const int* col, long long policy, hipStream_t st, std::string& error, int& rc);
void clear_plan_cache();                            // drops the cache's own references (live engines keep their plans)

struct Engine {
    std::shared_ptr<SharedPlan> plan;
    int ld = 0;                    // padded batch (multiple of 64)
    int lanes = 0;                 // real scenarios (<= ld; set by the owner, default ld): lanes beyond alias the last real one,
                                   // so a small batch moves 8 bytes per load instruction instead of 512
    Rec* fact_rec = nullptr; Rec* bwd_rec = nullptr;           // wave records (jg_symbolic.hpp), replay order
    Rec* pre_rec = nullptr; Segment* pre_seg = nullptr; int* pre_row = nullptr;   // level 0 of a prefactor plan (jg_symbolic.hpp); pre_row: device, [n]
    Segment* fact_seg = nullptr; Segment* bwd_seg = nullptr;
    Rec* sel_rec = nullptr; Segment* sel_seg = nullptr; double* Zs = nullptr;   // selected inverse (allocated on first use)
    Rec* fwd_rec = nullptr; Segment* fwd_seg = nullptr;          // forward elimination alone (factor once, solve many)
    int* bwd_chain = nullptr;                                   // backward chain task data (jg_symbolic.hpp)
    double* X = nullptr;           // factor values [n_entries][4][ld]: U, unscaled Lh, factored diagonal blocks
    double* W = nullptr;           // [n][2][ld] pivot order: y after factor(), x after backsolve()
    int* status = nullptr;         // [ld] bit 2 set on zero / non-finite pivot
    std::vector<DevLaunch> fact, bwd, bwdj, fwd, selv, pre;
    // Jordan plans (policy bit 49): factor() leaves Jordan rows for the pivots of the top tasks and backsolve() sweeps over them (one
    // backward level per task level instead of a sequential chain per task).  false: plain rows and the chain tables -- what
    // forward() + backsolve() need (the forward elimination of another right-hand side produces y, not the y' Jordan rows go with).
    // Read by factor() and backsolve() at launch time: change it only between a backsolve and the next factorisation.
    bool jordan = false;
    int probe_part = 0;            // TIMING PROBE (JG_PROBE_FACT_PART at create: 1 = factor() launches the bottom levels only, 2 = the top only; wrong numbers) -- tools/r05_overlap_probe.py
    double* jc = nullptr;          // single_bwd: the Jordan rows of the top tasks, compact ([n_jordan][4] doubles: k_fact_top writes, k_bwd1_top reads)
    bool single_bwd = false;       // ONE scenario on a Jordan plan: backsolve() runs the row-per-lane sweep (k_bwd1_top / k_bwd1_bottom, jg_symbolic.hpp: SingleTables)
    bool shared = false;           // hint (jg_nr_set_shared): other batches are in flight on this GPU -- the top launches take the 4-wave variant (same bits)
    Rec* top_task = nullptr; int* top_data = nullptr;          // multifrontal top (jg_symbolic.hpp): task headers, task data
    int* top_wgmap = nullptr;                                  // workgroup map of the grouped launches
    double* top_stack = nullptr;                               // update matrices of the tasks, scenario-major [ld][S.top_stack]
    long long* top_prof = nullptr;                             // JG_TOP_PROFILE: per-task phase stamps (printed by destroy)
    int device = 0;
    std::string error;

    int create(int n, const int* rowptr, const int* col, int ld_, long long policy, hipStream_t st);   // st: the owner's stream (setup copies)
    void destroy();
    // A: block values in the caller's CSR order [nnz][4][ld]; rhs: [n][2][ld] original block order.
    // Computes A = Lh inv(D) U and y = (Lh inv(D))^-1 rhs in the same launches.
    // sel: which 64-scenario groups take part (workgroups of the others exit at once).
    // In-place engines (policy bit 0) ignore A: the caller has assembled into X (entry S.src_entry[p] for its block p).
    // level0_done (plans with policy bit 2 only): the producer has stored the diagonal blocks of S.pre_pivot FACTORISED (diag_lu below)
    // and written their rhs rows into W (pre_row: pivot + 1 per ORIGINAL block index, 0 elsewhere); otherwise the PRE tables run first.
    int factor(hipStream_t st, const double* A, const double* rhs, const GroupSel& sel, bool level0_done = false);
    // y = (Lh inv(D))^-1 rhs with the factor of the last factor() call (solve many right-hand sides with one factorisation).
    int forward(hipStream_t st, const double* rhs, const GroupSel& sel);
    // Zs = A^-1 on the upper factor pattern + diagonal (same entry numbering as X) for a SYMMETRIC matrix, from the factor
    // of the last factor() call: Takahashi recursion, two launches per backward level (off-diagonals, then diagonals).
    int selected_inverse(hipStream_t st, const GroupSel& sel);
    // Fill the (in-place) factor storage with ONE shared matrix: blocks [nnz of the caller's pattern][4] (row-major 2x2),
    // replicated over every scenario; fill-in entries need nothing.
    int set_shared_matrix(hipStream_t st, const double* blocks_host);
    // x = U^-1 D y, scattered to original order into out [n][2][ld]; optional fused state update.
    int backsolve(hipStream_t st, double* out, const StateUpdate& upd, const GroupSel& sel);
    size_t factor_bytes() const { return ((size_t)plan->S.n_entries + (size_t)plan->S.n_jordan) * 4 * ld * sizeof(double); }   // Jordan rows sit behind the entries
};

#define JG_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess) {                                                        \
            this->error = std::string(#expr) + ": " + hipGetErrorString(err__);           \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

// Blocking copies / fills on the HANDLE's stream, never on the legacy stream: handles are driven from several host
// threads (ContingencyPipeline), and a legacy-stream operation issued while another thread captures its graphs fails
// ("would make the legacy stream depend on a capturing blocking stream").
inline hipError_t sync_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
inline hipError_t sync_fill(void* dst, int value, size_t bytes, hipStream_t st) {
    const hipError_t e = hipMemsetAsync(dst, value, bytes, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

template <class T>
int upload(T** dst, const std::vector<T>& src, std::string& err, hipStream_t st) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e == hipSuccess && !src.empty()) e = sync_copy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { err = std::string("upload: ") + hipGetErrorString(e); return 2; }
    return 0;
}

}  // namespace jg
