// jg_engine.hip -- device kernels of the batched block-sparse LU engine (gfx950, wave64).
//
// Factorisation A = Lh * inv(D) * U (Lh unscaled, D kept as 2x2 LU factors) with the forward elimination of the right-hand
// side fused in, then one backward sweep.  Lanes = 64 scenarios of one group; a wave works on one structural item, `wpi`
// waves share an item's update list (dealt round-robin, partial sums meet in LDS in a fixed order => bitwise
// run-to-run determinism).
//
// Replay tables (jg_symbolic.hpp): each wave's work is a 64-byte RECORD whose address follows from
// (segment, chunk, wave) arithmetic, fetched with ONE scalar load -- no descriptor -> index -> value pointer chase
// (that chain cost ~10 us per item, measured).  The bottom of the elimination tree is replayed with one launch per
// dependency level (k_fact_level / k_bwd_level: whole chip per level); the top of the tree -- long chains, few items per
// level -- by multifrontal tasks with one workgroup per scenario (k_fact_top).
#include "jg_engine.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <mutex>

namespace jg {

namespace {

struct FactArgs {
    const Rec* rec; const Segment* seg;
    const double* A; const double* rhs; double* X; double* W; int* status; GroupSel sel;
    int ld, seg_begin;         // per-level launches: blockIdx.y selects the level's segment seg_begin + y
    int lanes;                 // real scenarios: lane offsets are clamped to lanes - 1
    int s0_base, s0_nchunks, s0_wpi, s0_rpw;   // the level's FIRST segment travels in the kernel arguments (one dependent
                                               // scalar load less on the critical path of every level; most narrow levels have one)
};

struct BwdArgs {
    const Rec* rec; const Segment* seg; const int* chain;
    const double* X; double* W; double* out; GroupSel sel;
    StateUpdate upd;
    int ld, seg_begin;
    int lanes;
    int s0_base, s0_nchunks, s0_wpi, s0_rpw;
};


// Diagonal blocks are kept FACTORED, not inverted: a 2x2 LU with partial pivoting inside the block,
//   v00 = 1/u11, v01 = u12, v10 = l (+4 when the two rows were swapped; |l| <= 1), v11 = 1/u22.
// Applying D^-1 through this form is backward stable even when the block itself is badly conditioned
// (gain matrices with widely spread weights reach block condition numbers ~1e9, where an explicit
// inverse loses the solution).  y = D^-1 r:
__device__ __forceinline__ void dsolve(const Blk& d, double r1, double r2, double& y1, double& y2) {
    const bool sw = d.v10 > 2.0;
    const double l = sw ? d.v10 - 4.0 : d.v10;
    const double a = sw ? r2 : r1, b = sw ? r1 : r2;
    y2 = (b - l * a) * d.v11;
    y1 = (a - d.v01 * y2) * d.v00;
}

// c -= Lh(a) * D(d)^-1 * U(b)
__device__ __forceinline__ void term3(Blk& c, const Blk& l, const Blk& d, const Blk& u) {
    double z00, z10, z01, z11;
    dsolve(d, u.v00, u.v10, z00, z10);
    dsolve(d, u.v01, u.v11, z01, z11);
    c.v00 -= l.v00 * z00 + l.v01 * z10;
    c.v01 -= l.v00 * z01 + l.v01 * z11;
    c.v10 -= l.v10 * z00 + l.v11 * z10;
    c.v11 -= l.v10 * z01 + l.v11 * z11;
}

// 64-byte record through the scalar cache: ONE s_load_dwordx16 into 16 SGPRs.  The tables are immutable for the
// life of the engine, so they are read through the constant address space -- that is what lets the compiler keep the
// load scalar inside loops that also store (the factor values), and hoist the prefetch of the next record.
typedef int RecS __attribute__((ext_vector_type(16)));
typedef const RecS __attribute__((address_space(4)))* RecPtr;
__device__ __forceinline__ RecS load_rec(const Rec* base, size_t index) { return ((RecPtr)base)[index]; }
__device__ __forceinline__ int rec_word(const RecS& r, int k) { return r[k]; }

// Hand-placed requests: `global_load_dwordx4 dst, lane offset, scalar base` (the item index of every operand comes from a record in scalar
// registers).  hipcc fences loads that sit behind a branch on a run-time count -- `if (t < nt) load` -- with waits of their own: it reuses the
// destination registers of one load as address registers of the next and drains (part of) the queue before each address computation, so the
// operands of a record arrived one term after the other (rocprof, round 4; cdna_hip_programming.md lists the same trap).  Requests written this way
// go out back to back; ONE s_waitcnt, tied to the destination registers by its operands, stands before the first use.  The compiler does not count
// these loads: its own waits can only come earlier than needed (loads return in order), never later.
// (d2v, BlkV, gload16, gload8: jg_engine.hpp)

// one record of a factorisation item: up to FACT_T update terms, every operand requested before the first use (nothing in the request loop
// uses a loaded value: the transposition of a symmetric plan's operand is a select at the point of use; a rhs row's 2-vector is requested like a
// block whose second half re-reads the first: one kind of request, no branch)
__device__ __forceinline__ void fact_record(const FactArgs& a, const RecS& r, int kind, size_t b, size_t ld, Blk& c) {
    static_assert(FACT_T == 4, "operand list of the wait");
    const int nt = rec_word(r, 3);
    BlkV l[FACT_T], d[FACT_T], u[FACT_T];
#pragma unroll
    for (int t = 0; t < FACT_T; ++t) asm volatile("" : "=v"(l[t].r0), "=v"(l[t].r1), "=v"(d[t].r0), "=v"(d[t].r1), "=v"(u[t].r0), "=v"(u[t].r1));   // (no instruction)
    const char* const usrc = kind == 3 ? (const char*)a.W : (const char*)a.X;
    const size_t urow = kind == 3 ? ld * 16 : ld * 32, uhalf = kind == 3 ? 0 : ld * 16;
    const unsigned off = (unsigned)b * 16u;
#pragma unroll
    for (int t = 0; t < FACT_T; ++t) {
        if (t < nt) {
            const char* pl = (const char*)a.X + (size_t)(rec_word(r, 4 + 3 * t) & 0x3fffffff) * ld * 32;
            const char* pd = (const char*)a.X + (size_t)rec_word(r, 5 + 3 * t) * ld * 32;
            const char* pu = usrc + (size_t)rec_word(r, 6 + 3 * t) * urow;
            gload16(l[t].r0, pl, off); gload16(l[t].r1, pl + ld * 16, off);
            gload16(d[t].r0, pd, off); gload16(d[t].r1, pd + ld * 16, off);
            gload16(u[t].r0, pu, off); gload16(u[t].r1, pu + uhalf, off);
        }
    }
    {                                                            // (unconditional: a wait behind a second test of the count is a path no static check can follow)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(l[0].r0), "+v"(l[0].r1), "+v"(d[0].r0), "+v"(d[0].r1), "+v"(u[0].r0), "+v"(u[0].r1),
                          "+v"(l[1].r0), "+v"(l[1].r1), "+v"(d[1].r0), "+v"(d[1].r1), "+v"(u[1].r0), "+v"(u[1].r1));
        asm volatile("" : "+v"(l[2].r0), "+v"(l[2].r1), "+v"(d[2].r0), "+v"(d[2].r1), "+v"(u[2].r0), "+v"(u[2].r1),
                          "+v"(l[3].r0), "+v"(l[3].r1), "+v"(d[3].r0), "+v"(d[3].r1), "+v"(u[3].r0), "+v"(u[3].r1));
    }
#pragma unroll
    for (int t = 0; t < FACT_T; ++t) {
        if (t < nt) {
            const bool tr = (rec_word(r, 4 + 3 * t) >> 30) != 0;  // bit 30: read the block transposed (symmetric matrices: Lh(i,k) = U(k,i)')
            const Blk lt{l[t].r0.x, tr ? l[t].r1.x : l[t].r0.y, tr ? l[t].r0.y : l[t].r1.x, l[t].r1.y};
            const Blk dt{d[t].r0.x, d[t].r0.y, d[t].r1.x, d[t].r1.y};
            if (kind == 3) {          // y -= Lh(a) * D(d)^-1 * y_c   (y_c = the first half of u)
                double z0, z1;
                dsolve(dt, u[t].r0.x, u[t].r0.y, z0, z1);
                c.v00 -= lt.v00 * z0 + lt.v01 * z1;
                c.v01 -= lt.v10 * z0 + lt.v11 * z1;
            } else {
                term3(c, lt, dt, Blk{u[t].r0.x, u[t].r0.y, u[t].r1.x, u[t].r1.y});
            }
        }
    }
}

// Pivot guard of the static schedule.  The reference's solvers pivot (UMFPACK / KLU threshold pivoting,
// /root/reference/src/backend/utility.jl:470-484) and raise SingularException; here the pivot ORDER is fixed, so a pivot block
// that cancels to rounding level must be caught, not divided by: a diagonal block whose 2x2 LU has a pivot below
// PIVOT_EPS x (largest entry ITS ROW of the block started from) marks the scenario (status bit 2 -> per-scenario status 3
// in k_check / k_gn_check).  Row-wise, because the two rows of a block may live on very different scales (gain matrices:
// a slack angle row of 1 beside a PMU-weighted magnitude row of 1e10) and only cancellation is a defect.  A Jacobian of an islanded sub-grid (bridge outage: no slack in the island) ends exactly there:
// its last block is a difference of equal numbers, ~1e-16 of what went in, not an exact zero.
// (PIVOT_EPS, row_max and the 2x2 LU itself -- diag_lu -- live in jg_engine.hpp: a producer of a prefactor plan uses them too.)

__device__ __forceinline__ void fact_finish(const FactArgs& a, int kind, int id, size_t b, size_t ld, const Blk& c, double2 ref) {
    if (kind == 3) {
        store_vec(a.W, (size_t)id, b, ld, c.v00, c.v01);
        return;
    }
    if (kind == 2) {                            // diagonal block: 2x2 LU with in-block partial pivoting
        bool bad;
        const Blk f = diag_lu(c, ref, bad);
        if (bad) atomicOr(a.status + b, 4);
        store_blk(a.X, (size_t)id, b, ld, f.v00, f.v01, f.v10, f.v11);
    } else {
        store_blk(a.X, (size_t)id, b, ld, c.v00, c.v01, c.v10, c.v11);
    }
}

// One chunk (16 waves) of a factorisation segment for the 64 scenarios at lane offset b.  `first` is the wave's first
// record (already loaded); its remaining rpw - 1 records follow it.  Two workgroup barriers when wpi > 1.
__device__ __forceinline__ void fact_chunk(const FactArgs& a, double* red, const RecS& first, size_t rec_index, int rpw, int wpi,
                                           int wave, int lane, size_t b, size_t ld) {
    const int sub = wave & (wpi - 1);
    const int kind = rec_word(first, 0), id = rec_word(first, 1), src = rec_word(first, 2);
    Blk c{0.0, 0.0, 0.0, 0.0};
    double2 ref{0.0, 0.0};                                      // what the rows of a diagonal block started from (pivot guard)
    if (kind >= 0) {
        if (sub == 0) {
            if (kind == 3) { const double2 f = load_vec(a.rhs, (size_t)src, b, ld); c.v00 = f.x; c.v01 = f.y; }
            else if (src >= 0) c = load_blk(a.A, (size_t)src, b, ld);
            ref = row_max(c);
        }
        // long lists: the wave's next record is requested before the current one is consumed (the tables are static)
        RecS cur = first;
        for (int j = 1; j < rpw; ++j) {
            const RecS nxt = load_rec(a.rec, rec_index + j);
            fact_record(a, cur, kind, b, ld, c);
            cur = nxt;
        }
        fact_record(a, cur, kind, b, ld, c);
        if (wpi > 1 && sub != 0) {                              // partial sums as two 16-byte halves: [wave][half][lane]
            double2* q = (double2*)red + (size_t)wave * 128 + lane;
            q[0] = double2{c.v00, c.v01}; q[64] = double2{c.v10, c.v11};
        }
    }
    if (wpi > 1) __syncthreads();
    if (kind >= 0 && sub == 0) {
        for (int w = 1; w < wpi; ++w) {
            const double2* q = (const double2*)red + (size_t)(wave + w) * 128 + lane;
            const double2 h0 = q[0], h1 = q[64];
            c.v00 += h0.x; c.v01 += h0.y; c.v10 += h1.x; c.v11 += h1.y;
        }
        fact_finish(a, kind, id, b, ld, c, ref);
    }
    if (wpi > 1) __syncthreads();
}

__device__ __forceinline__ void bwd_record(const BwdArgs& a, const RecS& r, size_t b, size_t ld, double& y0, double& y1) {
    const int nt = rec_word(r, 3);
    Blk m[BWD_T]; double w0[BWD_T], w1[BWD_T];
#pragma unroll
    for (int t = 0; t < BWD_T; ++t) {
        if (t < nt) {
            m[t] = load_blk(a.X, (size_t)rec_word(r, 4 + 2 * t), b, ld);
            const double2 w = load_vec(a.W, (size_t)rec_word(r, 5 + 2 * t), b, ld);
            w0[t] = w.x; w1[t] = w.y;
        }
    }
#pragma unroll
    for (int t = 0; t < BWD_T; ++t) {
        if (t < nt) {
            y0 -= m[t].v00 * w0[t] + m[t].v01 * w1[t];
            y1 -= m[t].v10 * w0[t] + m[t].v11 * w1[t];
        }
    }
}

// x_k = D_k^-1 y, stored in pivot order (W), scattered to original order (out), optional fused state update
// What the fused state update needs besides x: requested when the row's record arrives, not after its solve (the bus flag was a
// dependent vector load + readfirstlane and the old state a dependent read-modify-write at the very end of every backward level)
struct UpdPre { int fl; bool act; double va, vm; };
__device__ __forceinline__ UpdPre upd_prefetch(const BwdArgs& a, int bus, size_t b, size_t ld) {
    UpdPre p{0, false, 0.0, 0.0};
    if (a.upd.va) {
        p.act = a.upd.active ? (a.upd.active[b] != 0) : true;
        p.fl = uniform((int)a.upd.flags[bus]);
        p.va = a.upd.va[(size_t)bus * ld + b];
        p.vm = a.upd.vm[(size_t)bus * ld + b];
    }
    return p;
}
__device__ __forceinline__ double2 bwd_finish(const BwdArgs& a, const Blk& d, double y0, double y1, int k, int bus, size_t b, size_t ld, const UpdPre& p) {
    double x0, x1;
    dsolve(d, y0, y1, x0, x1);
    store_vec(a.W, (size_t)k, b, ld, x0, x1);
    if (!a.upd.va || p.act) store_vec(a.out, (size_t)bus, b, ld, x0, x1);      // a finished scenario keeps its last increment
    if (a.upd.va) {
        if (p.act && (p.fl & 1)) a.upd.va[(size_t)bus * ld + b] = p.va + a.upd.sign * x0;
        if (p.act && (p.fl & 2)) a.upd.vm[(size_t)bus * ld + b] = p.vm + a.upd.sign * x1;
    }
    return double2{x0, x1};
}

// One backward CHAIN (jg_symbolic.hpp): rows k_0 < ... < k_{nb-1} of one supernode, external columns E shared by all.
//   phase A (parallel): acc_p = y_p - sum_{e in E} U(k_p, e) x_e, x_E staged once in LDS, wpr waves per row;
//   phase B (sequential over the chain, one workgroup barrier per pivot): x_c = D_c^-1 acc_c, then acc_p -= U(k_p, k_c) x_c
//   for p < c.  Row p is owned by wave p % 16 in phase B, so the only hand-off per pivot is x_c (double-buffered in LDS);
//   the blocks of a step are requested CHAIN_PF steps ahead (they do not depend on x).
// LDS: acc[CHAIN_MAX_ROWS] | xe[CHAIN_MAX_EXT] | part[16] | xc[2]  (double2 per lane each).
constexpr int CHAIN_PF = 3;
typedef const int __attribute__((address_space(4)))* CIntPtr;     // immutable task data: scalar loads
constexpr int CHAIN_LDS_D2 = (CHAIN_MAX_ROWS + CHAIN_MAX_EXT + 16 + 2) * 64;

__device__ __forceinline__ void bwd_chain_task(const BwdArgs& a, double* lds, const RecS& rec, int wave, int lane, size_t b, size_t ld) {
    const int nb = rec[0], nE = rec[1], wpr = rec[3];
    CIntPtr rows = (CIntPtr)a.chain + rec[2];
    CIntPtr ecol = rows + 3 * nb;
    CIntPtr uext = ecol + nE;
    CIntPtr uin = uext + nb * nE;
    double2* acc = (double2*)lds;
    double2* xe = acc + CHAIN_MAX_ROWS * 64;
    double2* part = xe + CHAIN_MAX_EXT * 64;
    double2* xc = part + 16 * 64;
    for (int q = wave; q < nE; q += 16) xe[q * 64 + lane] = load_vec(a.W, (size_t)ecol[q], b, ld);
    // the state-update operands of the (at most two) pivots this wave finishes in phase B: requested now, off the sequential path
    UpdPre up0{0, false, 0.0, 0.0}, up1{0, false, 0.0, 0.0};
    if (wave < nb) up0 = upd_prefetch(a, rows[3 * wave + 1], b, ld);
    if (wave + 16 < nb) up1 = upd_prefetch(a, rows[3 * (wave + 16) + 1], b, ld);
    __syncthreads();
    // ---- phase A
    const int rpr = 16 / wpr, sub = wave & (wpr - 1);
    const int len = (nE + wpr - 1) / wpr;
    for (int p0 = 0; p0 < nb; p0 += rpr) {
        const int p = p0 + wave / wpr;
        double y0 = 0.0, y1 = 0.0;
        if (p < nb) {
            if (sub == 0) { const double2 y = load_vec(a.W, (size_t)rows[3 * p], b, ld); y0 = y.x; y1 = y.y; }   // requested with the first blocks, not after the last
            const int q1 = min(sub * len + len, nE);
            for (int q = sub * len; q < q1; q += 4) {
                Blk m[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) if (q + t < q1) m[t] = load_blk(a.X, (size_t)uext[p * nE + q + t], b, ld);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (q + t < q1) {
                        const double2 x = xe[(q + t) * 64 + lane];
                        y0 -= m[t].v00 * x.x + m[t].v01 * x.y;
                        y1 -= m[t].v10 * x.x + m[t].v11 * x.y;
                    }
            }
        }
        if (wpr > 1) {
            part[wave * 64 + lane] = double2{y0, y1};
            __syncthreads();
            if (p < nb && sub == 0) for (int w = 1; w < wpr; ++w) { const double2 t = part[(wave + w) * 64 + lane]; y0 += t.x; y1 += t.y; }
        }
        if (p < nb && sub == 0) acc[p * 64 + lane] = double2{y0, y1};
        if (wpr > 1) __syncthreads();
    }
    __syncthreads();
    // ---- phase B: wave w owns rows p = w, w + 16 (CHAIN_MAX_ROWS = 32)
    Blk mb[CHAIN_PF][2], db[CHAIN_PF];
    auto request = [&](int c, Blk (&m)[2], Blk& d) {
        if (c < 0) return;
        if (wave == (c & 15)) d = load_blk(a.X, (size_t)rows[3 * c + 2], b, ld);
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int p = wave + 16 * i; if (p < c) m[i] = load_blk(a.X, (size_t)uin[p * nb + c], b, ld); }
    };
#pragma unroll
    for (int s = 0; s < CHAIN_PF; ++s) request(nb - 1 - s, mb[s], db[s]);
    for (int c0 = nb - 1; c0 >= 0; c0 -= CHAIN_PF) {
#pragma unroll
        for (int s = 0; s < CHAIN_PF; ++s) {
            const int c = c0 - s;
            if (c >= 0) {                                       // uniform across the workgroup
                double2* slot = xc + (c & 1) * 64;
                if (wave == (c & 15)) {
                    const double2 y = acc[c * 64 + lane];
                    slot[lane] = bwd_finish(a, db[s], y.x, y.y, rows[3 * c], rows[3 * c + 1], b, ld, c < 16 ? up0 : up1);
                }
                __syncthreads();
                const double2 x = slot[lane];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int p = wave + 16 * i;
                    if (p < c) {
                        double2 t = acc[p * 64 + lane];
                        t.x -= mb[s][i].v00 * x.x + mb[s][i].v01 * x.y;
                        t.y -= mb[s][i].v10 * x.x + mb[s][i].v11 * x.y;
                        acc[p * 64 + lane] = t;
                    }
                }
                request(c - CHAIN_PF, mb[s], db[s]);
            }
        }
    }
    __syncthreads();
}

// A SMALL chain (jg_symbolic.hpp: at most CHAIN_SMALL_ROWS rows, every chain below the top): the same two phases on eight waves with
// everything a row needs in the registers of ITS wave.  Wave p * wpr owns row p: its right-hand side, diagonal block, state-update
// operands and the in-chain blocks U(k_p, k_c) (requested CHAIN_PF steps ahead) -- all asked for before the first use, together with the
// external blocks, so the task is ONE round trip to memory, a reduction of the partial sums and nb - 1 barriers of an 8-wave workgroup
// (the general task stages x_E and the sums in LDS: four dependent round trips and 16-wave barriers, ~13 us against ~6 for a 7-row chain).
// Waves beyond 8 (the 16-wave kernel) only keep the barriers company.  LDS: part[8] | xc[2] (double2 per lane each).
constexpr int CHAIN_SMALL_LDS_D2 = (8 + 2) * 64;
__device__ __forceinline__ void bwd_chain_small(const BwdArgs& a, double* lds, const RecS& rec, int wave, int lane, size_t b, size_t ld) {
    const int nb = rec[0], nE = rec[1], wpr = rec[3];
    CIntPtr rows = (CIntPtr)a.chain + rec[2];
    CIntPtr ecol = rows + 3 * nb;
    CIntPtr uext = ecol + nE;
    CIntPtr uin = uext + nb * nE;
    double2* part = (double2*)lds;
    double2* xc = part + 8 * 64;
    const int p = wave / wpr, sub = wave & (wpr - 1);
    const bool live = wave < 8 && p < nb, owner = live && sub == 0;
    UpdPre up{0, false, 0.0, 0.0};
    double y0 = 0.0, y1 = 0.0;
    Blk d{0.0, 0.0, 0.0, 0.0};
    int k = 0, bus = 0;
    Blk mb[CHAIN_PF];
    auto request = [&](int c, Blk& m) { if (owner && c > p) m = load_blk(a.X, (size_t)uin[p * nb + c], b, ld); };
    if (owner) {
        k = rows[3 * p]; bus = rows[3 * p + 1];
        up = upd_prefetch(a, bus, b, ld);
        const double2 y = load_vec(a.W, (size_t)k, b, ld); y0 = y.x; y1 = y.y;
        d = load_blk(a.X, (size_t)rows[3 * p + 2], b, ld);
    }
#pragma unroll
    for (int s = 0; s < CHAIN_PF; ++s) request(nb - 1 - s, mb[s]);
    // ---- phase A: the external columns of row p, dealt over its wpr waves
    if (live) {
        const int len = (nE + wpr - 1) / wpr;
        const int q1 = min(sub * len + len, nE);
        for (int q = sub * len; q < q1; q += 4) {
            Blk m[4]; double2 x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (q + t < q1) { m[t] = load_blk(a.X, (size_t)uext[p * nE + q + t], b, ld); x[t] = load_vec(a.W, (size_t)ecol[q + t], b, ld); }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (q + t < q1) {
                    y0 -= m[t].v00 * x[t].x + m[t].v01 * x[t].y;
                    y1 -= m[t].v10 * x[t].x + m[t].v11 * x[t].y;
                }
        }
    }
    if (wpr > 1) {                                               // uniform across the workgroup
        if (live && sub != 0) part[wave * 64 + lane] = double2{y0, y1};
        __syncthreads();
        if (owner) for (int w = 1; w < wpr; ++w) { const double2 t = part[(wave + w) * 64 + lane]; y0 += t.x; y1 += t.y; }
    }
    // ---- phase B: from the last pivot up, one barrier per pivot (none after the first row's)
    for (int c0 = nb - 1; c0 >= 0; c0 -= CHAIN_PF) {
#pragma unroll
        for (int s = 0; s < CHAIN_PF; ++s) {
            const int c = c0 - s;
            if (c >= 0) {                                        // uniform across the workgroup
                double2* slot = xc + (c & 1) * 64;
                if (owner && p == c) slot[lane] = bwd_finish(a, d, y0, y1, k, bus, b, ld, up);
                if (c > 0) {
                    __syncthreads();
                    if (owner && p < c) {
                        const double2 x = slot[lane];
                        y0 -= mb[s].v00 * x.x + mb[s].v01 * x.y;
                        y1 -= mb[s].v10 * x.x + mb[s].v11 * x.y;
                    }
                    request(c - CHAIN_PF, mb[s]);
                }
            }
        }
    }
}

// One chunk of a backward segment: x_k = Dinv_k (y_k - sum_c U(k,c) x_c), scattered to original order; optional fused
// state update (Newton-Raphson: V/theta -= increment on active scenarios).
__device__ __forceinline__ void bwd_chunk(const BwdArgs& a, double* red, const RecS& first, size_t rec_index, int rpw, int wpi,
                                          int wave, int lane, size_t b, size_t ld) {
    const int sub = wave & (wpi - 1);
    const int k = rec_word(first, 0), bus = rec_word(first, 1), dg = rec_word(first, 2);
    double y0 = 0.0, y1 = 0.0;
    Blk d{0.0, 0.0, 0.0, 0.0};
    UpdPre up{0, false, 0.0, 0.0};
    if (k >= 0) {
        if (sub == 0) {
            up = upd_prefetch(a, bus, b, ld);
            const double2 y = load_vec(a.W, (size_t)k, b, ld);
            y0 = y.x; y1 = y.y;
            d = load_blk(a.X, (size_t)dg, b, ld);
        }
        RecS cur = first;
        for (int j = 1; j < rpw; ++j) {
            const RecS nxt = load_rec(a.rec, rec_index + j);
            bwd_record(a, cur, b, ld, y0, y1);
            cur = nxt;
        }
        bwd_record(a, cur, b, ld, y0, y1);
        if (wpi > 1 && sub != 0) ((double2*)red)[(size_t)wave * 64 + lane] = double2{y0, y1};
    }
    if (wpi > 1) __syncthreads();
    if (k >= 0 && sub == 0) {
        for (int w = 1; w < wpi; ++w) { const double2 p = ((const double2*)red)[(size_t)(wave + w) * 64 + lane]; y0 += p.x; y1 += p.y; }
        bwd_finish(a, d, y0, y1, k, bus, b, ld, up);
    }
    if (wpi > 1) __syncthreads();
}

// ---- executor 1: one launch per dependency level ------------------------------------------------------------
// grid.y = segments of the level (one per wpi class), grid.x = (most chunks of any of them) x group stride with the
// scenario group fastest (jg::map_block); the segment header comes through the scalar cache.
typedef int SegS __attribute__((ext_vector_type(8)));
typedef const SegS __attribute__((address_space(4)))* SegPtr;

template <bool BWD, int BW = 16, class Args>
__device__ __forceinline__ void level_body(const Args& a, double* red) {
    int base = a.s0_base, nchunks = a.s0_nchunks, wpi = a.s0_wpi, rpw = a.s0_rpw;
    if (blockIdx.y != 0) {
        const SegS sg = ((SegPtr)a.seg)[a.seg_begin + blockIdx.y];
        base = sg[0]; nchunks = sg[1]; wpi = sg[2]; rpw = sg[3];
    }
    int grp, bx;
    if (!map_block(a.sel, a.ld, BWD ? (wpi <= 0 ? nchunks : nchunks * (16 / BW)) : nchunks * (16 / FACT_WAVES), grp, bx)) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min(grp * 64 + lane, a.lanes - 1);
    if constexpr (BWD) {
        if (wpi <= 0) {                                          // chain segment: one task (one record) per workgroup
            if (wpi < 0) bwd_chain_small(a, red, load_rec(a.rec, (size_t)base + bx), wave, lane, b, ld);
            else if constexpr (BW == 16) bwd_chain_task(a, red, load_rec(a.rec, (size_t)base + bx), wave, lane, b, ld);
            return;
        }
    }
    const size_t ri = (size_t)base + ((size_t)bx * (BWD ? BW : FACT_WAVES) + wave) * rpw;
    const RecS r = load_rec(a.rec, ri);
    if constexpr (BWD) bwd_chunk(a, red, r, ri, rpw, wpi, wave, lane, b, ld);
    else fact_chunk(a, red, r, ri, rpw, wpi, wave, lane, b, ld);
}

__global__ __launch_bounds__(64 * FACT_WAVES, 4) void k_fact_level(FactArgs a) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [16][4][64]
    level_body<false>(a, red);
}

__global__ __launch_bounds__(1024) void k_bwd_level(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [16][2][64]
    level_body<true>(a, red);
}

// levels without chain tasks and without 16-wave items: 8-wave workgroups, two per CU (as for the factorisation)
__global__ __launch_bounds__(512, 4) void k_bwd_level8(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [8][2][64]
    level_body<true, 8>(a, red);
}

// ---- backward sweep of a SINGLE instance (jg_symbolic.hpp: SingleTables): the lanes are ROWS, not scenarios ------------------------------------------
// The level kernels above give a wave to one row of 64 scenarios; with ONE scenario 63 lanes idle and the sweep is 15 dependent launches of
// ~4.7 us (ACTIVSg10k: 72 us of a 370 us iteration).  Two launches instead: the top's rows in one workgroup (levels = workgroup barriers, its
// solution in LDS), the rows below the top as whole subtrees per workgroup with every sweep-independent operand requested before the first barrier.
// Same arithmetic per row (x_k = D_k^-1 (y_k - sum U(k,c) x_c), state update fused); the sum of a top row is formed by four lanes (fixed order).
struct Bwd1Args {
    const int* t_row; const int* t_ptr; const int* t_term; const int* t_level;
    const int* b_wg; const int* b_row; const int* b_term;
    const double* X; const double* jc; double* W; double* out; GroupSel sel; StateUpdate upd;    // jc: the compact Jordan rows k_fact_top left (TopArgs::jc)
    int ld, n_top_levels, n_wg, n_top;
    const int* t_jb; const int* t_cslot; const int* t_toff; int max_terms;                // terms as lanes (k_bwd1_top2; SingleTables::flat_ok)
};
struct Upd1 { int fl; bool act; double va, vm; };
// act: is the scenario active (read ONCE per launch: a load of it per row is a round trip ahead of everything else the row asks for)
__device__ __forceinline__ bool upd1_active(const StateUpdate& u, size_t b) { return u.va ? (u.active ? (u.active[b] != 0) : true) : false; }
__device__ __forceinline__ Upd1 upd1_prefetch(const StateUpdate& u, bool act, int bus, size_t b, size_t ld) {
    Upd1 p{0, act, 0.0, 0.0};
    if (u.va) {
        p.fl = (int)u.flags[bus];
        p.va = u.va[(size_t)bus * ld + b];
        p.vm = u.vm[(size_t)bus * ld + b];
    }
    return p;
}
__device__ __forceinline__ double2 bwd1_finish(const Bwd1Args& a, const Blk& d, double y0, double y1, int k, int bus, size_t b, size_t ld, const Upd1& p) {
    double x0, x1;
    dsolve(d, y0, y1, x0, x1);
    store_vec(a.W, (size_t)k, b, ld, x0, x1);
    if (!a.upd.va || p.act) store_vec(a.out, (size_t)bus, b, ld, x0, x1);       // a finished scenario keeps its last increment
    if (a.upd.va) {
        if (p.act && (p.fl & 1)) a.upd.va[(size_t)bus * ld + b] = p.va + a.upd.sign * x0;
        if (p.act && (p.fl & 2)) a.upd.vm[(size_t)bus * ld + b] = p.vm + a.upd.sign * x1;
    }
    return double2{x0, x1};
}

// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits (s_waitcnt vmcnt(0)) for every global store in flight -- the solution and state a
// level has just written, which nobody of this launch reads back -- i.e. one store round trip per level.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(1024) void k_bwd1_top(Bwd1Args a) {
    extern __shared__ __attribute__((aligned(16))) double red[];
    // LDS: solution of the top rows (slot = position in t_row) | the rows' descriptors | their pointers -- the descriptors come in with ONE coalesced sweep at the
    // start, so a row is one round trip to memory (slots, blocks, right-hand side, diagonal block, state: in flight together) and nothing is carried in flight
    // from one row to the next (a prefetched descriptor made the compiler wait with vmcnt(0) at the loop header: behind the stores of the row before).
    double2* xs = (double2*)red;
    int4* rowd = (int4*)(xs + a.n_top);
    int2* rowp = (int2*)(rowd + a.n_top);
    int grp, bx;
    if (!map_block(a.sel, a.ld, 1, grp, bx)) return;
    const size_t b = (size_t)grp * 64, ld = (size_t)a.ld;
    const int tid = threadIdx.x, quad = tid >> 2, q = tid & 3;
    __shared__ int lvl[64];                                       // first row of every level
    if (tid <= a.n_top_levels && tid < 64) lvl[tid] = a.t_level[tid];
    for (int r = tid; r < a.n_top; r += 1024) { rowd[r] = *(const int4*)(a.t_row + 4 * (size_t)r); rowp[r] = *(const int2*)(a.t_ptr + 2 * (size_t)r); }
    const bool act = upd1_active(a.upd, b);
    __syncthreads();
    constexpr int TQ = 8;                                         // terms per lane in flight
    for (int L = 0; L < a.n_top_levels; ++L) {
        const int r1 = lvl[L + 1];
        for (int row = lvl[L] + quad; row < r1; row += 256) {
            const int4 d = rowd[row];                             // pivot, bus, diagonal entry, terms
            const int2 tp = rowp[row];                            // first block of the row in the compact Jordan rows, first slot of its column list
            // (every lane of the quad asks for the row's right-hand side and diagonal block: same addresses, no branch -- and no wait -- ahead of the term loads)
            const double2 yk = load_vec(a.W, (size_t)d.x, b, ld);
            const Blk dg = load_blk(a.X, (size_t)d.z, b, ld);
            double y0 = 0.0, y1 = 0.0;
#ifndef JG_PROBE_BWD1
#define JG_PROBE_BWD1 0                 // TIMING PROBE (tools/experiments/r06_bwd1_probe.sh; wrong numbers; measured on the form that stored inside the level loop): 1 no term loads (34 -> 19 us), 2 no stores (-> 28), 3 neither solve nor stores
#endif
            for (int i0 = 0; i0 < (JG_PROBE_BWD1 == 1 ? 0 : d.w); i0 += 4 * TQ) {            // unconditional loads at clamped positions; lane q takes terms q, q + 4, ...
                int sl[TQ]; double2 m0[TQ], m1[TQ];
#pragma unroll
                for (int u = 0; u < TQ; ++u) {
                    const int i = i0 + 4 * u + q, ic = i < d.w ? i : 0;
                    sl[u] = a.t_term[tp.y + ic];
                    const double2* p = (const double2*)(a.jc + (size_t)(tp.x + ic) * 4);
                    m0[u] = p[0]; m1[u] = p[1];
                }
#pragma unroll
                for (int u = 0; u < TQ; ++u) {
                    const bool ok = i0 + 4 * u + q < d.w;
                    const double2 x = xs[sl[u]];
                    y0 = ok ? y0 - (m0[u].x * x.x + m0[u].y * x.y) : y0;
                    y1 = ok ? y1 - (m1[u].x * x.x + m1[u].y * x.y) : y1;
                }
            }
            y0 += __shfl_xor(y0, 1); y1 += __shfl_xor(y1, 1);
            y0 += __shfl_xor(y0, 2); y1 += __shfl_xor(y1, 2);
            if (JG_PROBE_BWD1 == 3) { if (q == 0) xs[row] = double2{y0, y1}; }
            else if (q == 0) { double x0, x1; dsolve(dg, yk.x + y0, yk.y + y1, x0, x1); xs[row] = double2{x0, x1}; }   // the solution stays in LDS through the levels ...
        }
        lds_barrier();
    }
    // ... and leaves in ONE pass at the end (pivot order, bus order, fused state update): stores inside the level loop made the compiler wait at the head of the
    // row loop for the acknowledgements of the row before (s_waitcnt vmcnt counts stores too), and the state operands rode in every row's round trip
    if (JG_PROBE_BWD1 == 2) return;
    for (int row = tid; row < a.n_top; row += 1024) {
        const int4 d = rowd[row];
        const Upd1 up = upd1_prefetch(a.upd, act, d.y, b, ld);
        const double2 x = xs[row];
        store_vec(a.W, (size_t)d.x, b, ld, x.x, x.y);
        if (!a.upd.va || up.act) store_vec(a.out, (size_t)d.y, b, ld, x.x, x.y);       // a finished scenario keeps its last increment
        if (a.upd.va) {
            if (up.act && (up.fl & 1)) a.upd.va[(size_t)d.y * ld + b] = up.va + a.upd.sign * x.x;
            if (up.act && (up.fl & 2)) a.upd.vm[(size_t)d.y * ld + b] = up.vm + a.upd.sign * x.y;
        }
    }
}

// The same sweep with the TERMS as lanes (SingleTables::flat_ok).  k_bwd1_top asks for a row's blocks from four lanes, eight requests each: on the two wide levels of a
// 10k-bus grid that is 24 memory instructions per wave and round on 16 waves -- the CU retires one per ~16 clocks -- for slots that are half clamped duplicates, and a
// round trip per level on top (32 us, of which 15 are those requests: probe builds).  Here lane g - g0 of a level takes block g (the compact blocks of a level follow
// each other in row order: 64 lanes = 2 KiB of contiguous memory), multiplies it with its column's solution and leaves the product in LDS; a barrier later ONE lane per
// row adds the row's products in column order, solves and publishes x.  The blocks of level l + 1 are requested before the rows of level l are summed (they do not
// depend on the sweep).  LDS: xs [n_top] | row descriptors [n_top] | first term of a row [n_top] | products [most terms of a level].
constexpr int BWD1_ROUNDS = 6;                                    // a level holds at most 6 x 1024 terms (else: k_bwd1_top)
__global__ __launch_bounds__(1024) void k_bwd1_top2(Bwd1Args a) {
    extern __shared__ __attribute__((aligned(16))) double red[];
    double2* xs = (double2*)red;
    int4* rowd = (int4*)(xs + a.n_top);
    double2* prod = (double2*)(rowd + a.n_top);
    int* toff = (int*)(prod + a.max_terms);
    __shared__ int lvl[64], jb0[64], jbn[64];
    int grp, bx;
    if (!map_block(a.sel, a.ld, 1, grp, bx)) return;
    const size_t b = (size_t)grp * 64, ld = (size_t)a.ld;
    const int tid = threadIdx.x, wv = uniform(tid >> 6);
    if (tid <= a.n_top_levels && tid < 64) lvl[tid] = a.t_level[tid];
    if (tid < a.n_top_levels && tid < 64) { jb0[tid] = a.t_jb[2 * tid]; jbn[tid] = a.t_jb[2 * tid + 1]; }
    for (int r = tid; r < a.n_top; r += 1024) { rowd[r] = *(const int4*)(a.t_row + 4 * (size_t)r); toff[r] = a.t_toff[r]; }
    const bool act = upd1_active(a.upd, b);
    __syncthreads();
    double2 m0[BWD1_ROUNDS], m1[BWD1_ROUNDS]; int cs[BWD1_ROUNDS];
#pragma unroll
    for (int u = 0; u < BWD1_ROUNDS; ++u) { m0[u] = m1[u] = double2{0.0, 0.0}; cs[u] = 0; }
    auto request = [&](int L) {                                   // blocks and column slots of level L: lane tid of round u takes term u * 1024 + tid
        const int g0 = jb0[L], nt = jbn[L];
#pragma unroll
        for (int u = 0; u < BWD1_ROUNDS; ++u) {
            if (u * 1024 + wv * 64 < nt) {                        // (wave-uniform: a wave beyond the level's terms issues nothing)
                const int g = g0 + min(u * 1024 + tid, nt - 1);
                const double2* p = (const double2*)(a.jc + (size_t)g * 4);
                m0[u] = p[0]; m1[u] = p[1];
                cs[u] = a.t_cslot[g];
            }
        }
    };
    request(0);
    for (int L = 0; L < a.n_top_levels; ++L) {
        const int r0 = lvl[L], nrow = lvl[L + 1] - r0, nt = jbn[L];
        const int row = r0 + tid;
        // the row of this thread: right-hand side and diagonal block leave now and are back when the products are in LDS (asked for one level ahead, together with
        // the blocks, they were no faster: 1.070 against 1.060 ms per solve, tools/experiments/r06_bwd1_ab.sh)
        int4 d{0, 0, 0, 0};
        double2 yk{0.0, 0.0}; Blk dg{0.0, 0.0, 0.0, 0.0};
        if (tid < nrow) { d = rowd[row]; yk = load_vec(a.W, (size_t)d.x, b, ld); dg = load_blk(a.X, (size_t)d.z, b, ld); }
#pragma unroll
        for (int u = 0; u < BWD1_ROUNDS; ++u) {
            const int t = u * 1024 + tid;
            if (u * 1024 + wv * 64 < nt && t < nt) {
                const double2 x = xs[cs[u]];
                prod[t] = double2{m0[u].x * x.x + m0[u].y * x.y, m1[u].x * x.x + m1[u].y * x.y};
            }
        }
        if (L + 1 < a.n_top_levels) request(L + 1);
        lds_barrier();
        if (tid < nrow) {
            double s0 = 0.0, s1 = 0.0;
            const double2* pp = prod + toff[row];
            for (int t = 0; t < d.w; ++t) { const double2 v = pp[t]; s0 += v.x; s1 += v.y; }
            double x0, x1;
            dsolve(dg, yk.x - s0, yk.y - s1, x0, x1);
            xs[row] = double2{x0, x1};
        }
        lds_barrier();
    }
    for (int row = tid; row < a.n_top; row += 1024) {             // the solution leaves in one pass (k_bwd1_top)
        const int4 d = rowd[row];
        const Upd1 up = upd1_prefetch(a.upd, act, d.y, b, ld);
        const double2 x = xs[row];
        store_vec(a.W, (size_t)d.x, b, ld, x.x, x.y);
        if (!a.upd.va || up.act) store_vec(a.out, (size_t)d.y, b, ld, x.x, x.y);
        if (a.upd.va) {
            if (up.act && (up.fl & 1)) a.upd.va[(size_t)d.y * ld + b] = up.va + a.upd.sign * x.x;
            if (up.act && (up.fl & 2)) a.upd.vm[(size_t)d.y * ld + b] = up.vm + a.upd.sign * x.y;
        }
    }
}

__global__ __launch_bounds__(SINGLE_BOTTOM_ROWS) void k_bwd1_bottom(Bwd1Args a) {
    __shared__ __attribute__((aligned(16))) double2 xs[SINGLE_BOTTOM_ROWS];
    int grp, w;
    if (!map_block(a.sel, a.ld, a.n_wg, grp, w)) return;
    const size_t b = (size_t)grp * 64, ld = (size_t)a.ld;
    const int r0 = a.b_wg[2 * w], levels = a.b_wg[2 * w + 1], r1 = a.b_wg[2 * w + 2];
    const int t = threadIdx.x;
    const bool live = r0 + t < r1;
    const int* rw = a.b_row + 6 * (size_t)(live ? r0 + t : r0);
    const int k = rw[0], bus = rw[1], dgi = rw[2], nt = live ? rw[3] : 0, tp = rw[4], lev = live ? rw[5] : -1;
    // everything that does not depend on the sweep, in flight together: the row's blocks, its right-hand side, the top's solution, the state
    constexpr int TB = 8;
    int2 tm[TB]; Blk m[TB]; double2 xt[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) tm[u] = *(const int2*)(a.b_term + 2 * (size_t)(tp + (u < nt ? u : 0)));
    const Upd1 up = upd1_prefetch(a.upd, upd1_active(a.upd, b), bus, b, ld);
    double2 y = load_vec(a.W, (size_t)k, b, ld);
    const Blk dg = load_blk(a.X, (size_t)dgi, b, ld);
#pragma unroll
    for (int u = 0; u < TB; ++u) { m[u] = load_blk(a.X, (size_t)tm[u].x, b, ld); xt[u] = load_vec(a.W, (size_t)(tm[u].y >= 0 ? tm[u].y : 0), b, ld); }
    for (int L = 0; L < levels; ++L) {
        if (lev == L) {
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const bool ok = u < nt;
                const double2 x = tm[u].y >= 0 ? xt[u] : xs[ok ? -(tm[u].y + 1) : 0];
                y.x = ok ? y.x - (m[u].v00 * x.x + m[u].v01 * x.y) : y.x;
                y.y = ok ? y.y - (m[u].v10 * x.x + m[u].v11 * x.y) : y.y;
            }
            for (int u = TB; u < nt; ++u) {                       // rows of more than TB columns (rare below the top): one at a time
                const int2 tr = *(const int2*)(a.b_term + 2 * (size_t)(tp + u));
                const Blk mm = load_blk(a.X, (size_t)tr.x, b, ld);
                const double2 x = tr.y >= 0 ? load_vec(a.W, (size_t)tr.y, b, ld) : xs[-(tr.y + 1)];
                y.x -= mm.v00 * x.x + mm.v01 * x.y;
                y.y -= mm.v10 * x.x + mm.v11 * x.y;
            }
            xs[t] = bwd1_finish(a, dg, y.x, y.y, k, bus, b, ld, up);
        }
        if (L + 1 < levels) lds_barrier();
    }
}

// ---- selected inverse (Takahashi recursion on the factor pattern, symmetric matrices; tables: jg_symbolic.cpp) ----------
struct SelArgs {
    const Rec* rec; const Segment* seg;
    const double* X; double* Z; GroupSel sel;
    int ld, seg_begin, lanes;
    int s0_base, s0_nchunks, s0_wpi, s0_rpw;
};

// T += U(i,k) * Z(k,j)   (Z entry read transposed when bit 30 of its id is set); hand-placed requests, one wait (see gload16)
__device__ __forceinline__ void sel_record(const SelArgs& a, const RecS& r, size_t b, size_t ld, Blk& t) {
    static_assert(BWD_T == 6, "operand list of the wait");
    const int nt = rec_word(r, 3);
    BlkV u[BWD_T], z[BWD_T];
#pragma unroll
    for (int q = 0; q < BWD_T; ++q) asm volatile("" : "=v"(u[q].r0), "=v"(u[q].r1), "=v"(z[q].r0), "=v"(z[q].r1));   // (no instruction)
    const unsigned off = (unsigned)b * 16u;
#pragma unroll
    for (int q = 0; q < BWD_T; ++q) {
        if (q < nt) {
            const char* pu = (const char*)a.X + (size_t)rec_word(r, 4 + 2 * q) * ld * 32;
            const char* pz = (const char*)a.Z + (size_t)(rec_word(r, 5 + 2 * q) & 0x3fffffff) * ld * 32;
            gload16(u[q].r0, pu, off); gload16(u[q].r1, pu + ld * 16, off);
            gload16(z[q].r0, pz, off); gload16(z[q].r1, pz + ld * 16, off);
        }
    }
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(u[0].r0), "+v"(u[0].r1), "+v"(z[0].r0), "+v"(z[0].r1), "+v"(u[1].r0), "+v"(u[1].r1), "+v"(z[1].r0), "+v"(z[1].r1),
                          "+v"(u[2].r0), "+v"(u[2].r1), "+v"(z[2].r0), "+v"(z[2].r1));
        asm volatile("" : "+v"(u[3].r0), "+v"(u[3].r1), "+v"(z[3].r0), "+v"(z[3].r1), "+v"(u[4].r0), "+v"(u[4].r1), "+v"(z[4].r0), "+v"(z[4].r1),
                          "+v"(u[5].r0), "+v"(u[5].r1), "+v"(z[5].r0), "+v"(z[5].r1));
    }
#pragma unroll
    for (int q = 0; q < BWD_T; ++q) {
        if (q < nt) {
            const bool tr = (rec_word(r, 5 + 2 * q) >> 30) & 1;
            const double z00 = z[q].r0.x, z11 = z[q].r1.y, z01 = tr ? z[q].r1.x : z[q].r0.y, z10 = tr ? z[q].r0.y : z[q].r1.x;
            t.v00 += u[q].r0.x * z00 + u[q].r0.y * z10;
            t.v01 += u[q].r0.x * z01 + u[q].r0.y * z11;
            t.v10 += u[q].r1.x * z00 + u[q].r1.y * z10;
            t.v11 += u[q].r1.x * z01 + u[q].r1.y * z11;
        }
    }
}

__global__ __launch_bounds__(64 * FACT_WAVES, 4) void k_sel_level(SelArgs a) {   // workgroups like k_fact_level
    extern __shared__ __attribute__((aligned(16))) double red[];   // [FACT_WAVES][2][64] double2
    int base = a.s0_base, nchunks = a.s0_nchunks, wpi = a.s0_wpi, rpw = a.s0_rpw;
    if (blockIdx.y != 0) {
        const SegS sg = ((SegPtr)a.seg)[a.seg_begin + blockIdx.y];
        base = sg[0]; nchunks = sg[1]; wpi = sg[2]; rpw = sg[3];
    }
    int grp, bx;
    if (!map_block(a.sel, a.ld, nchunks * (16 / FACT_WAVES), grp, bx)) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min(grp * 64 + lane, a.lanes - 1);
    const size_t ri = (size_t)base + ((size_t)bx * FACT_WAVES + wave) * rpw;
    const RecS first = load_rec(a.rec, ri);
    const int sub = wave & (wpi - 1);
    const int target = rec_word(first, 0);
    Blk t{0.0, 0.0, 0.0, 0.0}, d{0.0, 0.0, 0.0, 0.0};
    if (target >= 0) {
        if (sub == 0) d = load_blk(a.X, (size_t)rec_word(first, 1), b, ld);
        RecS cur = first;
        for (int j = 1; j < rpw; ++j) {
            const RecS nxt = load_rec(a.rec, ri + j);
            sel_record(a, cur, b, ld, t);
            cur = nxt;
        }
        sel_record(a, cur, b, ld, t);
        if (wpi > 1 && sub != 0) {
            double2* q = (double2*)red + (size_t)wave * 128 + lane;
            q[0] = double2{t.v00, t.v01}; q[64] = double2{t.v10, t.v11};
        }
    }
    if (wpi > 1) __syncthreads();
    if (target >= 0 && sub == 0) {
        for (int w = 1; w < wpi; ++w) {
            const double2* q = (const double2*)red + (size_t)(wave + w) * 128 + lane;
            const double2 h0 = q[0], h1 = q[64];
            t.v00 += h0.x; t.v01 += h0.y; t.v10 += h1.x; t.v11 += h1.y;
        }
        const double e = rec_word(first, 2) ? 1.0 : 0.0;          // diagonal: D^-1 (I - T); off-diagonal: -D^-1 T
        double z00, z10, z01, z11;
        dsolve(d, e - t.v00, -t.v10, z00, z10);
        dsolve(d, -t.v01, e - t.v11, z01, z11);
        store_blk(a.Z, (size_t)target, b, ld, z00, z01, z10, z11);
    }
}

// ---- multifrontal top (jg_symbolic.hpp): one workgroup per (task, scenario), dense front in registers ------------------
// The pivots above BlockSymbolic::top_level are not level items: a TASK owns a chain of consecutive pivots k0 .. k0+m-1
// (parent(k) = k + 1) and its front of f = m + e block rows / columns (e = |struct(last pivot)|) plus the rhs as column f.
// Lanes run across the front of ONE scenario (the level kernels run 64 scenarios across the lanes).
//   Where the front lives.  A first version kept it in LDS and updated it row by row: a pivot step then moves struct^2 blocks
//   through the LDS pipe (reads at 256 B/clk, writes at ~80 B/clk, half-empty lanes) -- measured 1 700 - 2 900 clocks per step
//   whatever the wave roles were.  Now thread (i mod 16, c mod 16) of a 16 x 16 grid OWNS block (i, c) in registers
//   (CLS x CLS blocks per thread) and LDS only carries what a step broadcasts: the pivot row, the pivot column, the pivot.
//   bulk threads (waves 0-3), step q:  z_c = D(q)^-1 U(q, c) for their columns, F(i, c) -= L(i, q) z_c for their blocks with
//                        i, c > q (dense: blocks outside the pattern are zero and stay zero); the owners of row q + 1 and of
//                        column q + 1 publish them for the next step.
//   pivot wave (wave 4): keeps its own copy of the chain's diagonal blocks (lane k: S(k,k)), applies the same update to them
//                        from the published row / column, factorises D(q + 1) (2x2 LU with partial pivoting, every lane the same
//                        copy: no divergence) and publishes it -- the only chain that links consecutive steps runs beside the
//                        bulk update instead of behind it.
// One workgroup barrier per step.  Load and store go straight between the batch-minor factor storage and the registers (entry
// map of the task); children's update blocks are pulled from the scenario-major stack by the thread that owns the target.
// Terms of an entry are applied in ascending pivot order => bitwise run-to-run determinism.
struct TopArgs {
    const Rec* task; const int* data;
    double* X; double* W; double* stack; int* status; GroupSel sel;
    long long stack_stride;        // doubles per scenario of the scenario-major stack (class 0)
    long long sbase[3], sstride[3];   // the stacks by interleave class (1 / 4 / 16 scenarios): first double, doubles per scenario
    long long* prof;               // JG_TOP_PROFILE: [task][8] wall-clock stamps of scenario 0 (start, loaded, children, steps, stored), else null
    int ld, lanes, task_begin, ntasks, lpg;   // lpg: scenarios per 64-lane group that get a workgroup (64, or the real count of a single small group)
    const int* wgmap; int wg_begin, nwg;      // grouped launches (k_fact_grp): workgroup x of a group -> task << 8 | scenario block
    double* jc; int jc_first;                 // ONE scenario (Engine::single_bwd): Jordan rows leave COMPACT -- block j at jc + 4 (j - jc_first), 32 contiguous bytes -- for k_bwd1_top
};

// Where scenario b keeps 16-byte unit q of a task's update block: the block is interleaved over the 2^lg scenarios that share a workgroup
// of the PARENT task (lg = 0: scenario-major, what a workgroup per scenario reads; jg_symbolic.hpp).  off: the block's offset in doubles.
__device__ __forceinline__ double2* stack_unit(const TopArgs& a, size_t b, int off, int lg) {
    const long long base = lg == 0 ? a.sbase[0] : (lg == 2 ? a.sbase[1] : a.sbase[2]);
    const long long stride = lg == 0 ? a.sstride[0] : (lg == 2 ? a.sstride[1] : a.sstride[2]);
    return (double2*)(a.stack + base) + ((((b >> lg) * (size_t)(stride >> 1) + (size_t)(off >> 1)) << lg) + (b & ((1u << lg) - 1)));
}

// 1 / x without the IEEE division sequence (v_rcp_f64 + two Newton steps: 5 dependent operations instead of ~14; the result
// is within an ulp or two, which only perturbs the stored pivot factors at rounding level -- the factorisation stays the
// exact product of what is stored).  0 -> inf -> NaN and NaN -> NaN, both caught by the pivot check.
__device__ __forceinline__ double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// 2x2 LU with in-block partial pivoting in the stored form of fact_finish.  The pivot sits on the dependent chain of every
// chain step, so the two reciprocals are independent here: u22 = det / u11 with det = o22 u11 - o21 u12 (same backward
// error as o22 - (o21 / u11) u12: one rounded product in front of the subtraction either way).
__device__ __forceinline__ Blk factor_diag(const Blk& c, int& bad, double2 ref) {   // bad: sticky flag, reported once at the end of the task; ref: pivot guard
    const bool sw = fabs(c.v10) > fabs(c.v00);
    const double u11 = sw ? c.v10 : c.v00, u12 = sw ? c.v11 : c.v01;
    const double o21 = sw ? c.v00 : c.v10, o22 = sw ? c.v01 : c.v11;
    const double det = fma(o22, u11, -(o21 * u12));
    const double iu11 = rcp_fast(u11), idet = rcp_fast(det);
    const double l = o21 * iu11;
    const double iu22 = u11 * idet;
    const double f1 = PIVOT_EPS * (sw ? ref.y : ref.x), f2 = PIVOT_EPS * (sw ? ref.x : ref.y);
    bad |= (!(fabs(u11) > f1) || !(fabs(det) > f2 * fabs(u11)) || !(fabs(iu11) < 1.0e300) || !(fabs(iu22) < 1.0e300)) ? 1 : 0;
    return Blk{iu11, u12, sw ? l + 4.0 : l, iu22};
}

__device__ __forceinline__ double uniform_d(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ Blk lds_get(const double* base, int k) {
    const double2* p = (const double2*)(base + (size_t)k * 4);
    const double2 r0 = p[0], r1 = p[1];
    return Blk{r0.x, r0.y, r1.x, r1.y};
}
__device__ __forceinline__ void lds_set(double* base, int k, const Blk& v) {
    double2* p = (double2*)(base + (size_t)k * 4);
    p[0] = double2{v.v00, v.v01}; p[1] = double2{v.v10, v.v11};
}
__device__ __forceinline__ Blk zero_blk() { return Blk{0.0, 0.0, 0.0, 0.0}; }
// c -= l * z
__device__ __forceinline__ void blk_sub(Blk& c, const Blk& l, const Blk& z) {
    c.v00 = fma(-l.v01, z.v10, fma(-l.v00, z.v00, c.v00));
    c.v01 = fma(-l.v01, z.v11, fma(-l.v00, z.v01, c.v01));
    c.v10 = fma(-l.v11, z.v10, fma(-l.v10, z.v00, c.v10));
    c.v11 = fma(-l.v11, z.v11, fma(-l.v10, z.v01, c.v11));
}

// ---- executor 1b: factorisation TASKS (jg_symbolic.hpp, plans with policy bit 50) ------------------------------------------------
// One 8-wave workgroup per task.  A wave first stages its share of the task's shared operands -- Lh(p,k) D(k)^-1 for row items, D(k)^-1 U(k,p)
// for column items -- into LDS slots, with the operand loads of its first item record already in flight; after ONE workgroup barrier every
// update term is one block from memory, one block from LDS and eight multiply-adds.  Split items (wpi > 1) meet in a TK_BAR round.
// LDS: slots [TASK_SLOTS][2][64] double2 | partial sums [8][2][64] double2 = 76 KiB: two workgroups per CU, as k_fact_level.
constexpr int TASK_LDS_D2 = (TASK_SLOTS + TASK_WAVES) * 128;

// The memory operands of a record are requested with hand-placed instructions: `global_load_dwordx4 dst, lane offset, scalar base` back to back
// and ONE `s_waitcnt vmcnt(0)` before the first use.  Left to the compiler, every operand's pair of loads was fenced by waits of its own (it reuses
// the destination registers of one load as address registers of the next and drains the queue before each address computation: rocprof showed a task's
// six operands arriving in six consecutive round trips; the wave records of k_fact_level suffer from the same thing in pairs).  The compiler does not count
// these loads: its own waits can only be earlier than needed (loads return in order), never later.
// Every request of the wave has arrived: ONE wait, tied to the destination registers by the operands of an instruction-less statement (volatile
// statements keep their order; nothing that uses the values is scheduled ahead of the wait).  Counted waits -- vmcnt(n) that leave the stores of the
// previous round or the operands behind the staging loads in flight -- were built and measured against this: 1.150 / 1.153 against 1.146 / 1.153 ms
// per factorisation, no difference, and a count that depends on the path taken cannot be checked statically (tools/check_asm_loads.py).
__device__ __forceinline__ void task_wait_all(BlkV (&m)[TASK_T], BlkV& own) {
    static_assert(TASK_T == 6, "operand list of the wait");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(m[0].r0), "+v"(m[0].r1), "+v"(m[1].r0), "+v"(m[1].r1), "+v"(m[2].r0), "+v"(m[2].r1),
                      "+v"(m[3].r0), "+v"(m[3].r1), "+v"(m[4].r0), "+v"(m[4].r1), "+v"(m[5].r0), "+v"(m[5].r1), "+v"(own.r0), "+v"(own.r1));
}
__device__ __forceinline__ void task_wait_stage(BlkV (&A)[TASK_STAGE], BlkV (&D)[TASK_STAGE]) {
    static_assert(TASK_STAGE == 2, "operand list of the wait");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(A[0].r0), "+v"(A[0].r1), "+v"(A[1].r0), "+v"(A[1].r1), "+v"(D[0].r0), "+v"(D[0].r1), "+v"(D[1].r0), "+v"(D[1].r1));
}

// The memory operands of a record (and the item's own block with its first record) are requested with hand-placed instructions -- `global_load_dwordx4
// dst, lane offset, scalar base`, back to back -- and waited for with ONE `s_waitcnt` before the first use.  Left to the compiler, every operand's pair
// of loads was fenced by waits of its own (it reuses the destination registers of one load as address registers of the next and drains the queue before
// each address computation: a task's six operands arrived in six consecutive round trips; the wave records of k_fact_level suffer from the same thing in
// pairs).
__device__ __forceinline__ void task_issue(const FactArgs& a, const RecS& q, size_t b, size_t ld, BlkV& own, BlkV (&m)[TASK_T]) {
    const int h = rec_word(q, 0), kind = h & TK_KIND;
    if (kind == 7) return;
    const unsigned off = (unsigned)b * 16u;
    if ((h & TK_FIRST) && ((h >> 8) & 7) == 0 && rec_word(q, 2) >= 0) {   // the owner starts from the item's block / rhs row (a rhs row: the second half re-reads the first)
        const int src = rec_word(q, 2);
        const char* p = kind == 3 ? (const char*)a.rhs + (size_t)src * ld * 16 : (const char*)a.A + (size_t)src * ld * 32;
        gload16(own.r0, p, off);
        gload16(own.r1, p + (kind == 3 ? 0 : ld * 16), off);
    }
#pragma unroll
    for (int t = 0; t < TASK_T; ++t) asm volatile("" : "=v"(m[t].r0), "=v"(m[t].r1));   // no instruction: what the registers held before this record is dead
                                                                                          // (else the compiler carries -- and copies -- it for the operands the record does not have)
    if (h & TK_DIRECT) return;
    const int nt = rec_word(q, 3) & 0xff;
    // one code path for blocks and for the 2-vectors of a rhs row: no branch per load
    const char* const src = kind == 3 ? (const char*)a.W : (const char*)a.X;
    const size_t row = kind == 3 ? ld * 16 : ld * 32, half = kind == 3 ? 0 : ld * 16;
#pragma unroll
    for (int t = 0; t < TASK_T; ++t)
        if (t < nt) {
            const char* p = src + (size_t)(rec_word(q, 4 + t) & 0xffffff) * row;
            gload16(m[t].r0, p, off);
            gload16(m[t].r1, p + half, off);
        }
}

// the arithmetic of a record (after its requests have arrived); returns true if the wave has an item to finish (task_finish)
__device__ __forceinline__ bool task_consume(const FactArgs& a, const RecS& q, size_t b, size_t ld, int wave, int lane, const double2* slots, double2* scratch,
                                             Blk& c, BlkV& own, BlkV (&m)[TASK_T], double2& ref) {
    const int h = rec_word(q, 0), kind = h & TK_KIND, sub = (h >> 8) & 7, wpi = (h >> 12) & 15;
    const bool last = (h & TK_LAST) != 0;
    task_wait_all(m, own);                                       // every request of the record has arrived (unconditional: idle records wait for nothing)
    if (kind != 7) {
        const int nt = rec_word(q, 3) & 0xff;
        if (h & TK_FIRST) {
            if (sub == 0 && rec_word(q, 2) >= 0) c = Blk{own.r0.x, own.r0.y, own.r1.x, own.r1.y};
            else c = Blk{0.0, 0.0, 0.0, 0.0};
        }
        if (h & TK_DIRECT) {                                     // terms whose shared operand found no slot (rare): three operands each, left to the compiler
            Blk l[TASK_DIRECT_T], d[TASK_DIRECT_T], u[TASK_DIRECT_T];
#pragma unroll
            for (int t = 0; t < TASK_DIRECT_T; ++t)
                if (t < nt) {
                    const int ia = rec_word(q, 4 + 3 * t);
                    l[t] = load_blk(a.X, (size_t)(ia & 0x3fffffff), b, ld);
                    if (ia >> 30) { const double x = l[t].v01; l[t].v01 = l[t].v10; l[t].v10 = x; }
                    d[t] = load_blk(a.X, (size_t)rec_word(q, 5 + 3 * t), b, ld);
                    if (kind == 3) { const double2 w = load_vec(a.W, (size_t)rec_word(q, 6 + 3 * t), b, ld); u[t] = Blk{w.x, 0.0, w.y, 0.0}; }
                    else u[t] = load_blk(a.X, (size_t)rec_word(q, 6 + 3 * t), b, ld);
                }
#pragma unroll
            for (int t = 0; t < TASK_DIRECT_T; ++t)
                if (t < nt) {
                    if (kind == 3) {
                        double z0, z1;
                        dsolve(d[t], u[t].v00, u[t].v10, z0, z1);
                        c = Blk{c.v00 - (l[t].v00 * z0 + l[t].v01 * z1), c.v01 - (l[t].v10 * z0 + l[t].v11 * z1), c.v10, c.v11};
                    } else term3(c, l[t], d[t], u[t]);
                }
        } else if (kind == 3) {                                  // y -= [Lh D^-1] y_c
#pragma unroll
            for (int t = 0; t < TASK_T; ++t)
                if (t < nt) {
                    const double2* p = slots + (size_t)(rec_word(q, 4 + t) >> 24) * 128 + lane;
                    const double2 s0 = p[0], s1 = p[64];
                    c = Blk{fma(-s0.y, m[t].r0.y, fma(-s0.x, m[t].r0.x, c.v00)), fma(-s1.y, m[t].r0.y, fma(-s1.x, m[t].r0.x, c.v01)), c.v10, c.v11};
                }
        } else if (h & TK_SIDE) {                                // c -= Lh(i,k) [D^-1 U]
#pragma unroll
            for (int t = 0; t < TASK_T; ++t)
                if (t < nt) {
                    const double2* p = slots + (size_t)(rec_word(q, 4 + t) >> 24) * 128 + lane;
                    const double2 s0 = p[0], s1 = p[64];
                    c.v00 = fma(-m[t].r0.y, s1.x, fma(-m[t].r0.x, s0.x, c.v00));
                    c.v01 = fma(-m[t].r0.y, s1.y, fma(-m[t].r0.x, s0.y, c.v01));
                    c.v10 = fma(-m[t].r1.y, s1.x, fma(-m[t].r1.x, s0.x, c.v10));
                    c.v11 = fma(-m[t].r1.y, s1.y, fma(-m[t].r1.x, s0.y, c.v11));
                }
        } else {                                                 // c -= [Lh D^-1] U(k,j)
#pragma unroll
            for (int t = 0; t < TASK_T; ++t)
                if (t < nt) {
                    const double2* p = slots + (size_t)(rec_word(q, 4 + t) >> 24) * 128 + lane;
                    const double2 s0 = p[0], s1 = p[64];
                    c.v00 = fma(-s0.y, m[t].r1.x, fma(-s0.x, m[t].r0.x, c.v00));
                    c.v01 = fma(-s0.y, m[t].r1.y, fma(-s0.x, m[t].r0.y, c.v01));
                    c.v10 = fma(-s1.y, m[t].r1.x, fma(-s1.x, m[t].r0.x, c.v10));
                    c.v11 = fma(-s1.y, m[t].r1.y, fma(-s1.x, m[t].r0.y, c.v11));
                }
        }
    }
    if (h & TK_BAR) {                                            // uniform across the workgroup (the tables mark the round for every wave)
        const bool split = kind != 7 && last && wpi > 1;
        if (split && sub != 0) { double2* p = scratch + (size_t)wave * 128 + lane; p[0] = double2{c.v00, c.v01}; p[64] = double2{c.v10, c.v11}; }
        __syncthreads();
        if (split && sub == 0)
            for (int w = 1; w < wpi; ++w) {
                const double2* p = scratch + (size_t)(wave + w) * 128 + lane;
                const double2 h0 = p[0], h1 = p[64];
                c.v00 += h0.x; c.v01 += h0.y; c.v10 += h1.x; c.v11 += h1.y;
            }
        __syncthreads();
    }
    if (kind != 7 && last && sub == 0) {
        // pivot guard: the row maxima of the block as it was loaded -- still in `own` (taken now: the next round's requests may overwrite it)
        const bool had = rec_word(q, 2) >= 0;
        ref = double2{had ? fmax(fabs(own.r0.x), fabs(own.r0.y)) : 0.0, had ? fmax(fabs(own.r1.x), fabs(own.r1.y)) : 0.0};
        return true;
    }
    return false;
}

// the slots this wave stages for its task: the staging entries of one record (at most TASK_STAGE).  Requests first ...
__device__ __forceinline__ void task_stage_issue(const FactArgs& a, const RecS& s, size_t b, size_t ld, BlkV (&A)[TASK_STAGE], BlkV (&D)[TASK_STAGE]) {
    const int ns = rec_word(s, 3) >> 8;
    const unsigned off = (unsigned)b * 16u;
#pragma unroll
    for (int u = 0; u < TASK_STAGE; ++u)
        if (u < ns) {
            const char* pa = (const char*)a.X + (size_t)(rec_word(s, 10 + 3 * u) & 0xffffff) * ld * 32;
            const char* pd = (const char*)a.X + (size_t)rec_word(s, 11 + 3 * u) * ld * 32;
            gload16(A[u].r0, pa, off); gload16(A[u].r1, pa + ld * 16, off);
            gload16(D[u].r0, pd, off); gload16(D[u].r1, pd + ld * 16, off);
        }
}
// ... then (after task_wait_stage) the multiplication by the pivot block and the write into the slot
__device__ __forceinline__ void task_stage_write(const RecS& s, int lane, double2* slots, const BlkV (&A)[TASK_STAGE], const BlkV (&D)[TASK_STAGE]) {
    const int ns = rec_word(s, 3) >> 8;
#pragma unroll
    for (int u = 0; u < TASK_STAGE; ++u)
        if (u < ns) {
            const int aw = rec_word(s, 10 + 3 * u);
            const bool tr = (aw >> 30) & 1;
            const Blk x{A[u].r0.x, tr ? A[u].r1.x : A[u].r0.y, tr ? A[u].r0.y : A[u].r1.x, A[u].r1.y};
            const Blk d{D[u].r0.x, D[u].r0.y, D[u].r1.x, D[u].r1.y};
            const bool sw = d.v10 > 2.0;
            const double l = sw ? d.v10 - 4.0 : d.v10;
            Blk o;
            if ((aw >> 29) & 1) {                                // x D^-1, row by row: z U_d = x, y L_d = z, columns swapped back
                const double z1 = x.v00 * d.v00, w1 = x.v10 * d.v00;
                const double z2 = (x.v01 - z1 * d.v01) * d.v11, w2 = (x.v11 - w1 * d.v01) * d.v11;
                const double y1 = z1 - z2 * l, v1 = w1 - w2 * l;
                o = sw ? Blk{z2, y1, w2, v1} : Blk{y1, z2, v1, w2};
            } else {                                             // D^-1 x, column by column
                dsolve(d, x.v00, x.v10, o.v00, o.v10);
                dsolve(d, x.v01, x.v11, o.v01, o.v11);
            }
            double2* p = slots + (size_t)rec_word(s, 12 + 3 * u) * 128 + lane;
            p[0] = double2{o.v00, o.v01}; p[64] = double2{o.v10, o.v11};
        }
}

__global__ __launch_bounds__(64 * TASK_WAVES, 4) void k_fact_task(FactArgs a) {
    extern __shared__ __attribute__((aligned(16))) double red[];
    double2* const slots = (double2*)red;
    double2* const scratch = slots + TASK_SLOTS * 128;
    int base = a.s0_base, ntasks = a.s0_nchunks, spw = a.s0_wpi, rpw = a.s0_rpw;
    if (blockIdx.y != 0) {
        const SegS sg = ((SegPtr)a.seg)[a.seg_begin + blockIdx.y];
        base = sg[0]; ntasks = sg[1]; spw = sg[2]; rpw = sg[3];
    }
    int grp, bx;
    if (!map_block(a.sel, a.ld, ntasks, grp, bx)) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min(grp * 64 + lane, a.lanes - 1);
    const size_t ri = (size_t)base + ((size_t)bx * TASK_WAVES + wave) * rpw;
    RecS r = load_rec(a.rec, ri);
    Blk c{0.0, 0.0, 0.0, 0.0};
    BlkV m[TASK_T], own{d2v{0.0, 0.0}, d2v{0.0, 0.0}};
    RecS nxt = r;
    if (rpw > 1) nxt = load_rec(a.rec, ri + 1);                  // the tables are static: the next record is requested before this one is consumed
    {   // ---- first round: the slots are staged with the round's own operands already in flight
        BlkV A[TASK_STAGE], D[TASK_STAGE];
#pragma unroll
        for (int u = 0; u < TASK_STAGE; ++u) asm volatile("" : "=v"(A[u].r0), "=v"(A[u].r1), "=v"(D[u].r0), "=v"(D[u].r1));
        task_stage_issue(a, r, b, ld, A, D);
        task_issue(a, r, b, ld, own, m);                         // the round's own operands travel with the staging loads: ONE round trip
        task_wait_stage(A, D);
        task_stage_write(r, lane, slots, A, D);
        for (int i = 1; i < spw; ++i) {                          // tasks of more than 16 slots (rare): the next batch once the registers are free
            const RecS s = i == 1 ? nxt : load_rec(a.rec, ri + i);
            task_stage_issue(a, s, b, ld, A, D);
            task_wait_stage(A, D);
            task_stage_write(s, lane, slots, A, D);
        }
        __syncthreads();
    }
    for (int j = 0; j < rpw; ++j) {
        double2 ref{0.0, 0.0};
        const bool fin = task_consume(a, r, b, ld, wave, lane, slots, scratch, c, own, m, ref);
        const int kind = rec_word(r, 0) & TK_KIND, id = rec_word(r, 1);
        if (j + 1 < rpw) {                                       // the next round's requests leave before this round's result is factorised and stored
            const RecS cur = nxt;
            if (j + 2 < rpw) nxt = load_rec(a.rec, ri + j + 2);
            task_issue(a, cur, b, ld, own, m);
            r = cur;
        }
        if (fin) fact_finish(a, kind, id, b, ld, c, ref);
    }
}

// ---- the factorisation below the top of a SINGLE instance (jg_symbolic.hpp: SINGLE_FACT_LEVELS): a thread per item ---------------------------------------
struct Fact1Args {
    const Rec* rec; const int* wg; int item0;                     // item0: first of the partial items (k_fact1_partial)
    const double* rhs; double* X; double* W; int* status; GroupSel sel;
    int ld, n_wg, n_items;
};
// one item, worked on by FOUR lanes (a quad): lane q takes the item's records q, q + 4, ... (four terms each, in flight together: unconditional loads at clamped
// operands) -- the long lists of the task-owned entries (up to dozens of bottom terms) would otherwise be a chain of round trips on one thread; the four partial
// sums meet through two shuffles (fixed order), lane 0 adds the block / rhs row as assembled and stores like fact_finish
__device__ __forceinline__ void fact1_item(const Fact1Args& a, int ri, int q, size_t b, size_t ld) {
    const int4* rp = (const int4*)(a.rec + ri);
    const int4 h = rp[0];                                         // kind, id, src, terms | first continuation record << 10
    const int kind = h.x, id = h.y, src = h.z, nt = h.w & 1023;
    const int4* rc = (const int4*)(a.rec + (h.w >> 10)) - 4;      // record r >= 1 of the item: rc + 4 r
    // (every lane of the quad asks for the item's own block: same address, no branch ahead of the term loads)
    const double2 f0 = *(const double2*)(kind == 3 ? (const char*)a.rhs + ((size_t)src * ld + b) * 16 : (const char*)a.X + (size_t)(src >= 0 ? src : 0) * ld * 32 + b * 16);
    const double2 f1 = *(const double2*)((const char*)a.X + (size_t)(kind != 3 && src >= 0 ? src : 0) * ld * 32 + ld * 16 + b * 16);
    Blk c{0.0, 0.0, 0.0, 0.0};
    for (int q0 = 4 * q; q0 < nt; q0 += 16) {
        const int4* tr = q0 < 4 ? rp : rc + 4 * (q0 >> 2);
        const int4 t1 = tr[1], t2 = tr[2], t3 = tr[3];
        const int ta[4] = {t1.x, t1.w, t2.z, t3.y}, td[4] = {t1.y, t2.x, t2.w, t3.z}, tb[4] = {t1.z, t2.y, t3.x, t3.w};
        const int cnt = min(4, nt - q0);
        double2 l0[4], l1[4], d0[4], d1[4], u0[4], u1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = u < cnt ? u : 0;
            const char* pl = (const char*)a.X + (size_t)ta[x] * ld * 32 + b * 16;
            const char* pd = (const char*)a.X + (size_t)td[x] * ld * 32 + b * 16;
            const char* pu = kind == 3 ? (const char*)a.W + ((size_t)tb[x] * ld + b) * 16 : (const char*)a.X + (size_t)tb[x] * ld * 32 + b * 16;
            l0[u] = *(const double2*)pl; l1[u] = *(const double2*)(pl + ld * 16);
            d0[u] = *(const double2*)pd; d1[u] = *(const double2*)(pd + ld * 16);
            u0[u] = *(const double2*)pu; u1[u] = *(const double2*)(kind == 3 ? pu : pu + ld * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < cnt) {
                const Blk lt{l0[u].x, l0[u].y, l1[u].x, l1[u].y}, dt{d0[u].x, d0[u].y, d1[u].x, d1[u].y};
                if (kind == 3) {                                  // y -= Lh(a) D(d)^-1 y_c
                    double z0, z1;
                    dsolve(dt, u0[u].x, u0[u].y, z0, z1);
                    c.v00 -= lt.v00 * z0 + lt.v01 * z1;
                    c.v01 -= lt.v10 * z0 + lt.v11 * z1;
                } else term3(c, lt, dt, Blk{u0[u].x, u0[u].y, u1[u].x, u1[u].y});
            }
        }
    }
#pragma unroll
    for (int d = 1; d <= 2; d <<= 1) { c.v00 += __shfl_xor(c.v00, d); c.v01 += __shfl_xor(c.v01, d); c.v10 += __shfl_xor(c.v10, d); c.v11 += __shfl_xor(c.v11, d); }
    if (q != 0) return;
    Blk s{0.0, 0.0, 0.0, 0.0};                                    // the item as assembled
    if (kind == 3) { s.v00 = f0.x; s.v01 = f0.y; }
    else if (src >= 0) s = Blk{f0.x, f0.y, f1.x, f1.y};
    const double2 ref = row_max(s);                               // what the rows of a diagonal block started from (pivot guard)
    c.v00 += s.v00; c.v01 += s.v01; c.v10 += s.v10; c.v11 += s.v11;
    if (kind == 3) { store_vec(a.W, (size_t)id, b, ld, c.v00, c.v01); return; }
    if (kind == 2) {                                              // diagonal block: 2x2 LU with in-block partial pivoting, pivot guard
        bool bad;
        const Blk f = diag_lu(c, ref, bad);
        if (bad) atomicOr(a.status + b, 4);
        store_blk(a.X, (size_t)id, b, ld, f.v00, f.v01, f.v10, f.v11);
    } else store_blk(a.X, (size_t)id, b, ld, c.v00, c.v01, c.v10, c.v11);
}

// f1: a workgroup = a run of whole bottom subtrees; its items level by level, a workgroup barrier between levels (what a level reads was written by this workgroup:
// the CU's own L1 is coherent for its waves once the stores have been acknowledged -- __syncthreads waits for them)
__global__ __launch_bounds__(256) void k_fact1_bottom(Fact1Args a) {
    int grp, w;
    if (!map_block(a.sel, a.ld, a.n_wg, grp, w)) return;
    const size_t b = (size_t)grp * 64, ld = (size_t)a.ld;
    const int* hdr = a.wg + (size_t)w * (SINGLE_FACT_LEVELS + 1);
    const int last = hdr[SINGLE_FACT_LEVELS];
    const int quad = (int)threadIdx.x >> 2, q = (int)threadIdx.x & 3;
    for (int l = 0; l < SINGLE_FACT_LEVELS; ++l) {
        const int i0 = hdr[l], i1 = hdr[l + 1];
        for (int j = i0 + quad; j < i1; j += 64) fact1_item(a, j, q, b, ld);
        if (i1 >= last) break;                                    // (uniform) nothing above this level in this workgroup
        __syncthreads();
    }
}
// f2: the partial sums of the task-owned entries and rhs rows (bottom terms only): every operand is final after f1
__global__ __launch_bounds__(256) void k_fact1_partial(Fact1Args a) {
    int grp, w;
    if (!map_block(a.sel, a.ld, (a.n_items + 63) / 64, grp, w)) return;
    const int j = w * 64 + ((int)threadIdx.x >> 2);
    if (j < a.n_items) fact1_item(a, a.item0 + j, (int)threadIdx.x & 3, (size_t)grp * 64, (size_t)a.ld);
}

// TIMING PROBES of a pivot step (compile with -DJG_PROBE_TOP=1: the next pivot's row / column are not published, =3: the bulk threads take the published row as D^-1 U(q, .) -- no pivot read, no solve --, =2: one block update per
// thread instead of CLS x CLS; wrong numbers -- tools/experiments/level_bound_probe.sh, DESIGN_LOG.md 3.3): what a step is made of.
#ifndef JG_PROBE_TOP
#define JG_PROBE_TOP 0
#endif
#ifndef JG_PROBE_STEP
#define JG_PROBE_STEP 0                 // TIMING PROBE (-DJG_PROBE_STEP=1 with JG_TOP_PROFILE=1): shader-clock stamps inside the pivot step of thread 0 -- tools/r05_step_probe.sh
#endif
constexpr int TOP_THREADS = 320;        // 16 x 16 bulk threads + the pivot wave

// PW = false: no pivot wave (256 threads).  The thread that owns the diagonal block of the NEXT pivot factorises it right after its
// own update of that block and publishes it; the arithmetic is the pivot wave's, block for block, so both variants give the same bits.
// A 5-wave workgroup puts two waves on one SIMD, which caps a CU at 2 (CLS = 3) or 1 (CLS = 4) workgroups; four waves sit one
// per SIMD: 4 resp. 2 workgroups.  A lone workgroup (small batches) is faster WITH the pivot wave (its chain runs beside the bulk
// update), so the launch picks the variant by its size (Engine::factor).
// FUSE (no pivot wave): TWO pivots per barrier.  Besides row / column / factorised block of pivot q the owners publish the row and
// column of pivot q + 1 and its diagonal block as they stand BEFORE pivot q is applied; every thread then forms, for its own rows and
// columns, L(i, q+1) = L0(i, q+1) - L(i, q) z_q(q+1), U(q+1, c) = U0(q+1, c) - L(q+1, q) z_q(c) and the factorised D(q+1) itself -- the
// very operations, in the very order, that the owners of those blocks perform in the one-pivot step (same bits) -- and applies both
// pivots to its blocks.  A third more arithmetic per pivot, half the barriers and half the publish / read round trips: the step is
// bound by those (DESIGN_LOG.md 3.3), not by the arithmetic.
// JORDAN (jg_symbolic.hpp: Jordan rows): the column of the next pivot is published with the blocks of the FINISHED pivot rows as they
// stand instead of zeros, so the bulk update -- which touches every block of the grid anyway -- also eliminates column q from the rows
// above it: rows i < q get row_i -= U(i,q) D(q)^-1 row_q over the columns c > q (the row of pivot q is published with zeros up to
// column q, so nothing left of it moves: the diagonal blocks, Lh and the in-task multipliers U(i,q) stay what they are).  After the
// last step a pivot row holds J(i, ext) and y'_i; the rows leave for their own region behind the factor entries.  Same step, same
// barrier count, no extra arithmetic issued.
template <int CLS, bool PW, bool FUSE = false, bool JORDAN = false>
__global__ __launch_bounds__(PW ? TOP_THREADS : 256) __attribute__((amdgpu_waves_per_eu(CLS == 4 || (FUSE && CLS == 3) ? 2 : (PW && CLS == 3 ? 3 : 4)))) void k_fact_top(TopArgs a) {
    constexpr int GL = 4, G = 1 << GL;                           // the front on a 16 x 16 thread grid (a 32 x 32 grid and one- / two-wave kernels were measured slower: tools/experiments/r05_*.patch)
    static_assert(!(PW && FUSE), "the fused step has no pivot wave");
    static_assert(!(JORDAN && FUSE), "the fused step keeps plain rows");
    __shared__ __attribute__((aligned(16))) double Dbuf[2][4];         // factorised pivot of the current / next step
    __shared__ __attribute__((aligned(16))) double Ubuf[2][64 * 4];    // pivot row  U(q, c)
    __shared__ __attribute__((aligned(16))) double Lbuf[2][64 * 4];    // pivot column Lh(i, q)
    __shared__ __attribute__((aligned(16))) double Dini[64 * 4];       // the chain's diagonal blocks as loaded (for the pivot wave)
    __shared__ __attribute__((aligned(16))) double Dref[64 * 2];       // pivot guard: row maxima of those blocks BEFORE the children's update matrices came in
    __shared__ __attribute__((aligned(16))) double U1buf[FUSE ? 2 : 1][FUSE ? 64 * 4 : 4];   // FUSE: row / column / diagonal block of the pair's
    __shared__ __attribute__((aligned(16))) double L1buf[FUSE ? 2 : 1][FUSE ? 64 * 4 : 4];   // SECOND pivot, not yet touched by the first
    __shared__ __attribute__((aligned(16))) double Sbuf[2][4];
    int grp, x;
    if (!map_block(a.sel, a.ld, a.ntasks * a.lpg, grp, x)) return;
    const int ti = x / a.lpg;
    const int bb = grp * 64 + (x - ti * a.lpg);
    if (bb >= a.lanes) return;                                   // padding lanes of the last group: no scenario, no work
    const int tid = threadIdx.x;
    const bool pivot_wave = PW && tid >= G * G;
    const int lane = tid & 63;
    const int gi = (tid >> GL) & (G - 1), gj = tid & (G - 1);     // bulk thread: row / column class on the G x G grid (16 x 16; 32 x 32 for lone workgroups)
    const bool prof = a.prof && tid == 0;                        // every scenario: the host prints scenario 0 and the spread over the batch
    long long* pt = a.prof + ((size_t)(a.task_begin + ti) * a.ld + bb) * 8;
    if (prof) {
        pt[0] = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        pt[5] = (long long)((xcc & 0xf) << 16 | ((hw >> 13) & 0x7) << 8 | ((hw >> 8) & 0xf));      // XCC | SE | CU
    }
    const RecS h = load_rec(a.task, (size_t)a.task_begin + ti);
    const int m = h[0], e = h[1], nchild = h[5], fprime = h[11];
    const int f = fprime - 1;
    const int* td = a.data + h[3];
    const size_t b = (size_t)bb, ld = (size_t)a.ld;
    double* stk = a.stack + b * (size_t)a.stack_stride;
    int bad = 0;
    Blk T[CLS][CLS];
    // The pivot wave owns no part of the front: its state lives in the registers the bulk threads use for T (a variable of its
    // own would stay allocated in every wave through the step loop -- 12 of the 128 registers that let two workgroups share a CU)
    Blk& mydiag = T[0][0];                                       // pivot wave, lane k < m: S(k,k), later the factorised D(k)
    double& myref_x = T[0][1].v00; double& myref_y = T[0][1].v01;   // pivot wave, lane k: row maxima S(k,k) entered the task with
    if (pivot_wave) { mydiag = Blk{0.0, 0.0, 0.0, 0.0}; myref_x = 0.0; myref_y = 0.0; }

    if (!pivot_wave) {
        // ---- load: entry map, then every gather of the thread in flight together (the map is read again for the store: nine
        // registers that would otherwise live through the step loop)
        int code[CLS][CLS];
#pragma unroll
        for (int r = 0; r < CLS; ++r)
#pragma unroll
            for (int c = 0; c < CLS; ++c) {
                const int i = r * G + gi, j = c * G + gj;
                code[r][c] = (i < f && j < fprime) ? td[i * fprime + j] : -1;
            }
        // (round 6, the variant of launches with FEW workgroups -- PW -- where nothing hides a latency: UNCONDITIONAL loads at clamped addresses, selected
        // afterwards, all of a chunk of rows in flight together.  Behind `if (cd ...)` the compiler puts every block's two loads into a branch of their own with
        // an s_waitcnt vmcnt(0) behind them: CLS^2 dependent round trips, 2.5 - 3.2 us of the ~20 us a task of a single instance takes.  The batched variant keeps
        // the conditional form: 17 registers less at class 2, which is a workgroup more per CU.)
        if constexpr (PW) {
            constexpr int RC = CLS == 4 ? 2 : CLS;               // rows per chunk (class 4: two chunks, the loads of one need 64 registers)
#pragma unroll
            for (int r0 = 0; r0 < CLS; r0 += RC) {
                double2 g0[RC][CLS], g1[RC][CLS];
#pragma unroll
                for (int r = r0; r < r0 + RC; ++r)
#pragma unroll
                    for (int c = 0; c < CLS; ++c) {
                        const int cd = code[r][c];
                        const bool isrhs = cd <= -2, isblk = cd >= 0 && !((cd >> 28) & 1);
                        const char* px = (const char*)a.X + (size_t)(isblk ? (cd & 0x0fffffff) : 0) * ld * 32 + (unsigned)b * 16u;
                        const char* pw = (const char*)a.W + (size_t)(isrhs ? -(cd + 2) : 0) * ld * 16 + (unsigned)b * 16u;
                        g0[r - r0][c] = *(const double2*)(isrhs ? pw : px);
                        g1[r - r0][c] = *(const double2*)(px + ld * 16);
                    }
#pragma unroll
                for (int r = r0; r < r0 + RC; ++r)
#pragma unroll
                    for (int c = 0; c < CLS; ++c) {
                        const int cd = code[r][c];
                        const bool isrhs = cd <= -2, isblk = cd >= 0 && !((cd >> 28) & 1), tr = ((cd >> 28) & 2) != 0;   // tr: symmetric plans, Lh(i,c) = U(c,i)'
                        const double2 a0 = g0[r - r0][c], a1 = g1[r - r0][c];
                        Blk v;
                        v.v00 = (isrhs || isblk) ? a0.x : 0.0;
                        v.v01 = isblk ? (tr ? a1.x : a0.y) : 0.0;
                        v.v10 = isrhs ? a0.y : (isblk ? (tr ? a0.y : a1.x) : 0.0);
                        v.v11 = isblk ? a1.y : 0.0;
                        T[r][c] = v;
                    }
            }
        } else {
#pragma unroll
            for (int r = 0; r < CLS; ++r)
#pragma unroll
                for (int c = 0; c < CLS; ++c) {
                    const int cd = code[r][c];
                    Blk v{0.0, 0.0, 0.0, 0.0};
                    if (cd <= -2) { const double2 y = load_vec(a.W, (size_t)(-(cd + 2)), b, ld); v.v00 = y.x; v.v10 = y.y; }
                    else if (cd >= 0 && !((cd >> 28) & 1)) {
                        v = load_blk(a.X, (size_t)(cd & 0x0fffffff), b, ld);
                        if ((cd >> 28) & 2) { const double s = v.v01; v.v01 = v.v10; v.v10 = s; }        // symmetric plans: Lh(i,c) = U(c,i)'
                    }
                    T[r][c] = v;
                }
        }
        // what the guard compares a pivot with: the block as it entered the task -- a pivot that the children's update matrices (or the
        // task's own steps) cancel to rounding level is the signature of an island whose root sits in the top (taken after
        // the extend-add the reference scale was the cancelled value itself)
#pragma unroll
        for (int r = 0; r < CLS; ++r)
            if (gi == gj && r * G + gi < m) *(double2*)(Dref + (size_t)(r * G + gi) * 2) = row_max(T[r][r]);
        if (prof) pt[1] = wall_clock64();
        // ---- extend-add: every thread pulls what the children left for its blocks (child order fixed => deterministic)
        if constexpr (PW && CLS == 2) {
            // class 2 (four blocks per thread): the children come in PAIRS -- both records travel together, then both sets of blocks: a task of four children pays three
            // round trips where child after child paid five (the 9241-bus grid: the extend-add was 3 - 6 of a task's ~16 us); added in child order under selects: same bits
            constexpr int GC = 2;
            const int* cd = td + h[7];
            const int stride = 2 + fprime;
            int n_coff[GC], n_ce[GC], n_ri[GC][CLS], n_cj[GC][CLS];
            auto fetch = [&](int ch0) {
#pragma unroll
                for (int g = 0; g < GC; ++g) {
                    const int* p = cd + (size_t)(ch0 + g < nchild ? ch0 + g : ch0) * stride;     // a missing second child re-reads the first (masked below)
                    n_coff[g] = p[0]; n_ce[g] = p[1];
#pragma unroll
                    for (int r = 0; r < CLS; ++r) { const int i = r * G + gi; const int v = p[2 + (i < f ? i : 0)]; n_ri[g][r] = i < f ? v : -1; }
#pragma unroll
                    for (int c = 0; c < CLS; ++c) { const int j = c * G + gj; const int v = p[2 + (j < fprime ? j : 0)]; n_cj[g][c] = j < fprime ? v : -1; }
                }
            };
            if (nchild > 0) fetch(0);
            for (int ch0 = 0; ch0 < nchild; ch0 += GC) {
                int coff[GC], ce[GC], ri[GC][CLS], cj[GC][CLS];
#pragma unroll
                for (int g = 0; g < GC; ++g) {
                    coff[g] = n_coff[g]; ce[g] = n_ce[g];
#pragma unroll
                    for (int r = 0; r < CLS; ++r) ri[g][r] = n_ri[g][r];
#pragma unroll
                    for (int c = 0; c < CLS; ++c) cj[g][c] = n_cj[g][c];
                }
                if (ch0 + GC < nchild) fetch(ch0 + GC);
                double2 s0[GC][CLS][CLS], s1[GC][CLS][CLS];
#pragma unroll
                for (int g = 0; g < GC; ++g)
#pragma unroll
                    for (int r = 0; r < CLS; ++r)
#pragma unroll
                        for (int c = 0; c < CLS; ++c) {
                            const bool ok = ri[g][r] >= 0 && cj[g][c] >= 0;
                            const double2* p = (const double2*)(stk + coff[g] + (ok ? ((size_t)ri[g][r] * (ce[g] + 1) + cj[g][c]) * 4 : 0));
                            s0[g][r][c] = p[0]; s1[g][r][c] = p[1];
                        }
#pragma unroll
                for (int g = 0; g < GC; ++g)
#pragma unroll
                    for (int r = 0; r < CLS; ++r)
#pragma unroll
                        for (int c = 0; c < CLS; ++c) {
                            const bool ok = ch0 + g < nchild && ri[g][r] >= 0 && cj[g][c] >= 0;
                            const double2 x0 = s0[g][r][c], x1 = s1[g][r][c];
                            T[r][c].v00 = ok ? T[r][c].v00 + x0.x : T[r][c].v00; T[r][c].v01 = ok ? T[r][c].v01 + x0.y : T[r][c].v01;
                            T[r][c].v10 = ok ? T[r][c].v10 + x1.x : T[r][c].v10; T[r][c].v11 = ok ? T[r][c].v11 + x1.y : T[r][c].v11;
                        }
            }
        } else if constexpr (PW) {
            // (round 6, launches with few workgroups: a child's blocks are requested TOGETHER -- unconditional loads at clamped offsets, added under a select: same
            // bits -- and the record of the next child travels with them.  The conditional loads below compile to one round trip per block and two more per
            // child record: 3 - 6 us of a single instance's task.)
            const int* cd = td + h[7];
            int n_coff = 0, n_ce = 0, n_ri[CLS], n_cj[CLS];
            auto fetch = [&](const int* p) {
                n_coff = p[0]; n_ce = p[1];
#pragma unroll
                for (int r = 0; r < CLS; ++r) { const int i = r * G + gi; const int v = p[2 + (i < f ? i : 0)]; n_ri[r] = i < f ? v : -1; }
#pragma unroll
                for (int c = 0; c < CLS; ++c) { const int j = c * G + gj; const int v = p[2 + (j < fprime ? j : 0)]; n_cj[c] = j < fprime ? v : -1; }
            };
            if (nchild > 0) fetch(cd);
            for (int ch = 0; ch < nchild; ++ch) {
                const int coff = n_coff, ce = n_ce;
                int ri[CLS], cj[CLS];
#pragma unroll
                for (int r = 0; r < CLS; ++r) ri[r] = n_ri[r];
#pragma unroll
                for (int c = 0; c < CLS; ++c) cj[c] = n_cj[c];
                cd += 2 + fprime;
                if (ch + 1 < nchild) fetch(cd);
                const double* C = stk + coff;
                constexpr int RC = CLS == 4 ? 2 : CLS;
#pragma unroll
                for (int r0 = 0; r0 < CLS; r0 += RC) {
                    double2 s0[RC][CLS], s1[RC][CLS];
#pragma unroll
                    for (int r = r0; r < r0 + RC; ++r)
#pragma unroll
                        for (int c = 0; c < CLS; ++c) {
                            const bool ok = ri[r] >= 0 && cj[c] >= 0;
                            const double2* p = (const double2*)(C + (ok ? ((size_t)ri[r] * (ce + 1) + cj[c]) * 4 : 0));
                            s0[r - r0][c] = p[0]; s1[r - r0][c] = p[1];
                        }
#pragma unroll
                    for (int r = r0; r < r0 + RC; ++r)
#pragma unroll
                        for (int c = 0; c < CLS; ++c) {
                            const bool ok = ri[r] >= 0 && cj[c] >= 0;
                            const double2 x0 = s0[r - r0][c], x1 = s1[r - r0][c];
                            T[r][c].v00 = ok ? T[r][c].v00 + x0.x : T[r][c].v00; T[r][c].v01 = ok ? T[r][c].v01 + x0.y : T[r][c].v01;
                            T[r][c].v10 = ok ? T[r][c].v10 + x1.x : T[r][c].v10; T[r][c].v11 = ok ? T[r][c].v11 + x1.y : T[r][c].v11;
                        }
                }
            }
        } else {
            const int* cd = td + h[7];
            for (int ch = 0; ch < nchild; ++ch) {
                const int coff = cd[0], ce = cd[1];
                const int* inv = cd + 2;
                const double* C = stk + coff;
                int ri[CLS], cj[CLS];
#pragma unroll
                for (int r = 0; r < CLS; ++r) { const int i = r * G + gi; ri[r] = i < f ? inv[i] : -1; }
#pragma unroll
                for (int c = 0; c < CLS; ++c) { const int j = c * G + gj; cj[c] = j < fprime ? inv[j] : -1; }
#pragma unroll
                for (int r = 0; r < CLS; ++r)
#pragma unroll
                    for (int c = 0; c < CLS; ++c)
                        if (ri[r] >= 0 && cj[c] >= 0) {
                            const double2* p = (const double2*)(C + ((size_t)ri[r] * (ce + 1) + cj[c]) * 4);
                            const double2 s0 = p[0], s1 = p[1];
                            T[r][c].v00 += s0.x; T[r][c].v01 += s0.y; T[r][c].v10 += s1.x; T[r][c].v11 += s1.y;
                        }
                cd += 2 + fprime;
            }
        }
        if (prof) pt[2] = wall_clock64();
        // ---- publish step 0: row 0, column 0, the chain's diagonal blocks; thread 0 factorises D(0).
        // What is published is what the step needs and ZERO elsewhere (row: columns <= q, column: rows <= q), so the bulk update
        // below runs without a single predicate: a finished block sees L = 0 or z = 0 and keeps its value.
        const Blk zero{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < CLS; ++r)
#pragma unroll
            for (int c = 0; c < CLS; ++c) {
                const int i = r * G + gi, j = c * G + gj;
                if (i == 0) lds_set(Ubuf[0], j, j > 0 ? T[r][c] : zero);
                if (j == 0) lds_set(Lbuf[0], i, i > 0 ? T[r][c] : zero);
                if (i == j && i < m) lds_set(Dini, i, T[r][c]);
                if (FUSE) {
                    if (i == 1) lds_set(U1buf[0], j, j > 1 ? T[r][c] : zero);
                    if (j == 1) lds_set(L1buf[0], i, i > 1 ? T[r][c] : zero);
                    if (i == 1 && j == 1) lds_set(Sbuf[0], 0, T[r][c]);
                }
            }
        if (tid == 0) {
            const Blk d0 = factor_diag(T[0][0], bad, *(const double2*)Dref);
            lds_set(Dbuf[0], 0, d0);
            if (!PW) T[0][0] = d0;                               // without a pivot wave the owner keeps the factorised block for the store
        }
    }
    __syncthreads();
    if (pivot_wave) {
        if (lane < m) { mydiag = lds_get(Dini, lane); const double2 rm = *(const double2*)(Dref + (size_t)lane * 2); myref_x = rm.x; myref_y = rm.y; }
        if (lane == 0) mydiag = lds_get(Dbuf[0], 0);
    }
    int q_done = 0;                                              // pivots finished by fused steps
    if constexpr (FUSE) {
        for (int qv = 0; qv + 1 < m; qv += 2) {
            const int q = uniform(qv), q1 = q + 1;
            const int cur = (q >> 1) & 1, nxt = cur ^ 1;
            Blk D = lds_get(Dbuf[cur], 0);
            D = Blk{uniform_d(D.v00), uniform_d(D.v01), uniform_d(D.v10), uniform_d(D.v11)};
            const int sw = uniform(D.v10 > 2.0 ? 1 : 0);
            const double dl = D.v10 - 4.0 * sw;
            auto zcol = [&](const double* ub, int k) {           // z = D(q)^-1 U(q, k)
                const double2* p = (const double2*)(ub + (size_t)k * 4);
                const double2 a0 = p[sw], a1 = p[sw ^ 1];
                Blk z;
                z.v10 = (a1.x - dl * a0.x) * D.v11; z.v00 = (a0.x - D.v01 * z.v10) * D.v00;
                z.v11 = (a1.y - dl * a0.y) * D.v11; z.v01 = (a0.y - D.v01 * z.v11) * D.v00;
                return z;
            };
            // what the owners of row / column / diagonal q + 1 would do in step q: here every thread does it for itself
            const Blk zq1 = zcol(Ubuf[cur], q1);                 // z_q(q + 1)
            const Blk Lq1q = lds_get(Lbuf[cur], q1);             // Lh(q + 1, q)
            Blk S1 = lds_get(Sbuf[cur], 0);
            blk_sub(S1, Lq1q, zq1);
            int bad1 = 0;
            const Blk D1 = factor_diag(S1, bad1, *(const double2*)(Dref + (size_t)q1 * 2));
            const int rq1 = q1 >> GL, tq1 = q1 & (G - 1);
            if (gi == tq1 && gj == tq1) bad |= bad1;             // reported once, by the owner of the block
            const bool sw1 = D1.v10 > 2.0;
            const double dl1 = sw1 ? D1.v10 - 4.0 : D1.v10;
            Blk Lq[CLS], L1[CLS];
#pragma unroll
            for (int r = 0; r < CLS; ++r) {
                Lq[r] = lds_get(Lbuf[cur], r * G + gi);
                L1[r] = lds_get(L1buf[cur], r * G + gi);
                blk_sub(L1[r], Lq[r], zq1);
                if (r * G + gi <= q1) L1[r] = zero_blk();       // rows up to q + 1 are finished for pivot q + 1
            }
#pragma unroll
            for (int c = 0; c < CLS; ++c) {
                const Blk z = zcol(Ubuf[cur], c * G + gj);
                Blk U1 = lds_get(U1buf[cur], c * G + gj);
                blk_sub(U1, Lq1q, z);
                if (c * G + gj <= q1) U1 = zero_blk();
                const double a0x = sw1 ? U1.v10 : U1.v00, a0y = sw1 ? U1.v11 : U1.v01, a1x = sw1 ? U1.v00 : U1.v10, a1y = sw1 ? U1.v01 : U1.v11;
                Blk z1;                                          // z_{q+1}(c) = D(q + 1)^-1 U(q + 1, c)
                z1.v10 = (a1x - dl1 * a0x) * D1.v11; z1.v00 = (a0x - D1.v01 * z1.v10) * D1.v00;
                z1.v11 = (a1y - dl1 * a0y) * D1.v11; z1.v01 = (a0y - D1.v01 * z1.v11) * D1.v00;
#pragma unroll
                for (int r = 0; r < CLS; ++r) { blk_sub(T[r][c], Lq[r], z); blk_sub(T[r][c], L1[r], z1); }
            }
            // the owner of S(q+1, q+1) keeps the factorised block for the store
            if (gi == tq1 && gj == tq1) {
#pragma unroll
                for (int r = 0; r < CLS; ++r) if (r == rq1) T[r][r] = D1;
            }
            q_done = q + 2;
            if (q + 2 < m) {                                     // the next pair leaves its owners: pivot q + 2 final, pivot q + 3 as it stands
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int p = q + 2 + w;
                    if (p < m) {
                        const int rp = p >> GL, tp = p & (G - 1);
                        double* ub = w ? U1buf[nxt] : Ubuf[nxt];
                        double* lb = w ? L1buf[nxt] : Lbuf[nxt];
                        if (gi == tp && gj == tp) {
#pragma unroll
                            for (int r = 0; r < CLS; ++r)
                                if (r == rp) {
                                    if (w == 0) {
                                        const Blk dn = factor_diag(T[r][r], bad, *(const double2*)(Dref + (size_t)p * 2));
                                        lds_set(Dbuf[nxt], 0, dn);
                                        T[r][r] = dn;
                                    } else lds_set(Sbuf[nxt], 0, T[r][r]);
                                }
                        }
                        if (gi == tp) {
#pragma unroll
                            for (int r = 0; r < CLS; ++r)
                                if (r == rp) {
#pragma unroll
                                    for (int c = 0; c < CLS; ++c) {
                                        if (c < rp) lds_set(ub, c * G + gj, zero_blk());
                                        else if (c > rp) lds_set(ub, c * G + gj, T[r][c]);
                                        else lds_set(ub, c * G + gj, gj > tp ? T[r][c] : zero_blk());
                                    }
                                }
                        }
                        if (gj == tp) {
#pragma unroll
                            for (int c = 0; c < CLS; ++c)
                                if (c == rp) {
#pragma unroll
                                    for (int r = 0; r < CLS; ++r) {
                                        if (r < rp) lds_set(lb, r * G + gi, zero_blk());
                                        else if (r > rp) lds_set(lb, r * G + gi, T[r][c]);
                                        else lds_set(lb, r * G + gi, gi > tp ? T[r][c] : zero_blk());
                                    }
                                }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---- pivot steps.  Straight-line bulk code: the pivot is the same block for every lane, so the row swap of its 2x2 LU is
    // folded into the ADDRESS of the two halves of U(q, c) (scalar), finished blocks see zeros (no predicates, no skipping).
    // (after fused steps: the one pivot an odd chain has left; its row / column / block sit in the buffers of the pair it would have led)
#if JG_PROBE_STEP
    long long ps[5] = {0, 0, 0, 0, 0};                           // read | update | publish | barrier (clocks of thread 0's wave, summed over the steps)
#define JG_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" ::: "memory"); const long long now_ = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); if (k >= 0) ps[k] += now_ - last_; last_ = now_; } while (0)
    long long last_ = 0;
#else
#define JG_STAMP(k) do { } while (0)
#endif
    for (int qv = q_done; qv < m; ++qv) {
        JG_STAMP(-1);
        const int q = uniform(qv);                               // the step number is wave-uniform: which row / column class publishes, the
        const int cur = FUSE ? (q >> 1) & 1 : q & 1, nxt = cur ^ 1;   // LDS buffer in use and the lane of the next pivot are scalar decisions
        Blk D = (JG_PROBE_TOP == 3 && !pivot_wave) ? Blk{1.0, 0.0, 0.0, 1.0} : lds_get(Dbuf[cur], 0);     // (probe 3: what a published D^-1 U(q, .) would save)
        D = Blk{uniform_d(D.v00), uniform_d(D.v01), uniform_d(D.v10), uniform_d(D.v11)};   // the same block in every lane: scalar registers
        const int sw = uniform(D.v10 > 2.0 ? 1 : 0);
        const double dl = D.v10 - 4.0 * sw;
        auto zcol = [&](const double* ub, int k) {               // z = D^-1 U(q, k): rows of U read in pivot order
            const double2* p = (const double2*)(ub + (size_t)k * 4);
            const double2 a0 = p[sw], a1 = p[sw ^ 1];
            Blk z;
            z.v10 = (a1.x - dl * a0.x) * D.v11; z.v00 = (a0.x - D.v01 * z.v10) * D.v00;
            z.v11 = (a1.y - dl * a0.y) * D.v11; z.v01 = (a0.y - D.v01 * z.v11) * D.v00;
            return z;
        };
        if (!pivot_wave) {
            Blk Lq[CLS];
#pragma unroll
            for (int r = 0; r < CLS; ++r) Lq[r] = lds_get(Lbuf[cur], r * G + gi);
#if JG_PROBE_STEP
            Blk Uq[CLS];                                         // (probe: the row is read before the stamp, the solve z = D^-1 U(q, .) counts as update)
#pragma unroll
            for (int c = 0; c < CLS; ++c) Uq[c] = lds_get(Ubuf[cur], c * G + gj);
            JG_STAMP(0);
#endif
#pragma unroll
            for (int c = 0; c < CLS; ++c) {
#if JG_PROBE_STEP
                Blk z;                                           // zcol on the registers read above
                {
                    const double a0x = sw ? Uq[c].v10 : Uq[c].v00, a0y = sw ? Uq[c].v11 : Uq[c].v01, a1x = sw ? Uq[c].v00 : Uq[c].v10, a1y = sw ? Uq[c].v01 : Uq[c].v11;
                    z.v10 = (a1x - dl * a0x) * D.v11; z.v00 = (a0x - D.v01 * z.v10) * D.v00;
                    z.v11 = (a1y - dl * a0y) * D.v11; z.v01 = (a0y - D.v01 * z.v11) * D.v00;
                }
#else
                const Blk z = JG_PROBE_TOP == 3 ? lds_get(Ubuf[cur], c * G + gj) : zcol(Ubuf[cur], c * G + gj);
#endif
#pragma unroll
                for (int r = 0; r < CLS; ++r) if (JG_PROBE_TOP != 2 || (r == 0 && c == 0)) blk_sub(T[r][c], Lq[r], z);
            }
#if JG_PROBE_STEP
            { double sink = 0.0;
#pragma unroll
              for (int r = 0; r < CLS; ++r)
#pragma unroll
                  for (int c = 0; c < CLS; ++c) sink += T[r][c].v11;
              asm volatile("" :: "v"(sink)); }                   // the updates have left the pipe
            JG_STAMP(1);
#endif
            if (q + 1 < m && JG_PROBE_TOP != 1) {                     // the next pivot row / column leave their owners
                // classes before the pivot's are finished (zeros), classes after it go out as they are (both uniform); only the
                // pivot's own class needs a per-lane select
                const int rq = (q + 1) >> GL, tq = (q + 1) & (G - 1);
                // (the owner's chain of ~55 dependent instructions BEFORE the row and the column leave: the other order -- their LDS writes travelling while the chain runs --
                // measured 0.7 % slower at 512 scenarios, profiles/r05_reorder_ab.txt)
                if (!PW && gi == tq && gj == tq) {               // the owner of S(q+1, q+1): final now, factorised here
                    const double2 ref = *(const double2*)(Dref + (size_t)(q + 1) * 2);
#pragma unroll
                    for (int r = 0; r < CLS; ++r)
                        if (r == rq) {
                            const Blk dn = factor_diag(T[r][r], bad, ref);
                            lds_set(Dbuf[nxt], 0, dn);
                            T[r][r] = dn;                        // later steps see L = 0 for this row: it stays what it is
                        }
                }
                if (gi == tq) {
#pragma unroll
                    for (int r = 0; r < CLS; ++r)
                        if (r == rq) {
#pragma unroll
                            for (int c = 0; c < CLS; ++c) {
                                if (c < rq) lds_set(Ubuf[nxt], c * G + gj, zero_blk());
                                else if (c > rq) lds_set(Ubuf[nxt], c * G + gj, T[r][c]);
                                else lds_set(Ubuf[nxt], c * G + gj, gj > tq ? T[r][c] : zero_blk());
                            }
                        }
                }
                if (gj == tq) {
#pragma unroll
                    for (int c = 0; c < CLS; ++c)
                        if (c == rq) {
#pragma unroll
                            for (int r = 0; r < CLS; ++r) {
                                if (r < rq) lds_set(Lbuf[nxt], r * G + gi, JORDAN ? T[r][c] : zero_blk());     // JORDAN: the rows above lose column q + 1 too
                                else if (r > rq) lds_set(Lbuf[nxt], r * G + gi, T[r][c]);
                                else lds_set(Lbuf[nxt], r * G + gi, (JORDAN ? gi != tq : gi > tq) ? T[r][c] : zero_blk());
                            }
                        }
                }
            }
        } else if (q + 1 < m) {
            // lane k: S(k,k) -= Lh(k,q) D(q)^-1 U(q,k) (lanes <= q see zeros); the block of lane q + 1 is then final
            const Blk z = zcol(Ubuf[cur], lane), Lk = lds_get(Lbuf[cur], lane);
            blk_sub(mydiag, Lk, z);
            auto bc = [&](double v) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(v), q + 1), hi = __builtin_amdgcn_readlane(__double2hiint(v), q + 1);
                return __hiloint2double(hi, lo);
            };
            const Blk dn = factor_diag(Blk{bc(mydiag.v00), bc(mydiag.v01), bc(mydiag.v10), bc(mydiag.v11)}, bad, double2{bc(myref_x), bc(myref_y)});
            if (lane == q + 1) { mydiag = dn; lds_set(Dbuf[nxt], 0, dn); }
        }
        JG_STAMP(2);
        __syncthreads();
        JG_STAMP(3);
    }
#if JG_PROBE_STEP
    if (prof) { pt[6] = ps[0] << 32 | (ps[1] & 0xffffffffll); pt[7] = ps[2] << 32 | (ps[3] & 0xffffffffll); }
#endif
    if (prof) pt[3] = wall_clock64();
    // ---- store
    if (!pivot_wave) {
        const int lgo = h[12];                                   // scenario interleave of the update block: that of the parent's workgroup
        double2* out = e > 0 ? stack_unit(a, b, h[4], lgo) : nullptr;
        const int jb = JORDAN ? h[14] : -1;                      // Jordan rows: block jb + i e + (j - m) for pivot row i, external column j
        // (round 6, launches with few workgroups: the entry map of the thread's slots is requested in one go, ahead of the stores; read slot by slot every code
        // waits -- s_waitcnt vmcnt(0) -- for the stores of the slot before it as well: 3 - 4 us per task of a single instance)
        int code[CLS][CLS];
#pragma unroll
        for (int r = 0; r < CLS; ++r)
#pragma unroll
            for (int c = 0; c < CLS; ++c) {
                const int i = r * G + gi, j = c * G + gj;
                const bool in = i < f && j < fprime;
                if constexpr (PW) { const int v = td[in ? i * fprime + j : 0]; code[r][c] = in ? v : -1; }
                else code[r][c] = in ? td[i * fprime + j] : -1;
            }
#pragma unroll
        for (int r = 0; r < CLS; ++r)
#pragma unroll
            for (int c = 0; c < CLS; ++c) {
                const int i = r * G + gi, j = c * G + gj;
                const int cd = code[r][c];
                const Blk& v = T[r][c];
                if (JORDAN && i < m && j >= m && j < f) {
                    if (a.jc) {                                   // a single instance: the row's blocks side by side (the batch-minor storage puts the two halves of a block 1 KiB apart)
                        double2* p = (double2*)(a.jc + (size_t)(jb - a.jc_first + i * e + (j - m)) * 4);
                        p[0] = double2{v.v00, v.v01}; p[1] = double2{v.v10, v.v11};
                    } else store_blk(a.X, (size_t)(jb + i * e + (j - m)), b, ld, v.v00, v.v01, v.v10, v.v11);
                }
                else if (cd <= -2) store_vec(a.W, (size_t)(-(cd + 2)), b, ld, v.v00, v.v10);
                else if (cd >= 0 && (!((cd >> 28) & 4) || (!PW && i == j && i < m))) store_blk(a.X, (size_t)(cd & 0x0fffffff), b, ld, v.v00, v.v01, v.v10, v.v11);
                else if (i >= m && i < f && j >= m && j < fprime) {                              // update matrix | vector: scenario-major stack
                    double2* p = out + ((((size_t)(i - m) * (e + 1) + (j - m)) * 2) << lgo);
                    p[0] = double2{v.v00, v.v01}; p[(size_t)1 << lgo] = double2{v.v10, v.v11};
                }
            }
        if ((PW ? tid == 0 : true) && bad) atomicOr(a.status + b, 4);
    } else {
        if (lane < m) store_blk(a.X, (size_t)td[h[8] + lane], b, ld, mydiag.v00, mydiag.v01, mydiag.v10, mydiag.v11);
        if (bad && lane == 0) atomicOr(a.status + b, 4);
    }
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pt[4] = wall_clock64(); }
}

// ---- the top of a SYMMETRIC plan (the Gauss-Newton gain), round 5 ---------------------------------------------------------------------------
// k_fact_top mirrors a symmetric front from its upper entries and eliminates all of it: twice the gathers, twice the extend-add, twice the block updates and
// twice the registers of what LDL' needs (class 4 -- fronts of 49 - 63 rows, which only the gain has -- at 198 VGPRs: two workgroups per CU).  Here a thread
// keeps only its blocks ON AND ABOVE the diagonal (i <= j; the right-hand side is column f): CLS (CLS + 1) / 2 class blocks instead of CLS^2.  The pivot
// column is the pivot row transposed -- Lh(i, q) = U(q, i)' --, so the owners of row q publish BOTH operands of a step; with Jordan rows the blocks above the
// pivot, U(i, q) for i < q, come from the owners of column q as before.  Children leave (and parents read) the upper triangle of an update matrix at the
// offsets of the full one.  No pivot wave, Jordan rows only: what a Gauss-Newton handle runs by default; the plain-row paths (jg_gn_set_method(h, 1), the
// selected inverse) keep k_fact_top.  Another summation order than the mirrored elimination (whose lower half is not bitwise the transpose of its upper half): results
// agree to rounding, tests/test_se_*.py hold both against the oracle; JG_TOP_SYM=0 runs the mirrored kernel.
__device__ __forceinline__ Blk blk_t(const Blk& v) { return Blk{v.v00, v.v10, v.v01, v.v11}; }
template <int CLS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CLS == 4 ? 3 : 4))) void k_fact_top_sym(TopArgs a) {
    __shared__ __attribute__((aligned(16))) double Dbuf[2][4];
    __shared__ __attribute__((aligned(16))) double Ubuf[2][64 * 4];    // pivot row U(q, c), zero for c <= q
    __shared__ __attribute__((aligned(16))) double Lbuf[2][64 * 4];    // i > q: U(q, i)';  i < q: U(i, q) (Jordan);  i == q and i >= f: zero
    __shared__ __attribute__((aligned(16))) double Dref[64 * 2];
    int grp, x;
    if (!map_block(a.sel, a.ld, a.ntasks * a.lpg, grp, x)) return;
    const int ti = x / a.lpg;
    const int bb = grp * 64 + (x - ti * a.lpg);
    if (bb >= a.lanes) return;
    const int tid = threadIdx.x;
    const int gi = (tid >> 4) & 15, gj = tid & 15;
    const bool prof = a.prof && tid == 0;
    long long* pt = a.prof + ((size_t)(a.task_begin + ti) * a.ld + bb) * 8;
    if (prof) {
        pt[0] = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        pt[5] = (long long)((xcc & 0xf) << 16 | ((hw >> 13) & 0x7) << 8 | ((hw >> 8) & 0xf));
    }
    const RecS h = load_rec(a.task, (size_t)a.task_begin + ti);
    const int m = h[0], e = h[1], nchild = h[5], fprime = h[11];
    const int f = fprime - 1;
    const int* td = a.data + h[3];
    const size_t b = (size_t)bb, ld = (size_t)a.ld;
    double* stk = a.stack + b * (size_t)a.stack_stride;
    int bad = 0;
    Blk T[CLS][CLS];                                             // T[r][c] with r <= c only (the others are never named: no registers)
    {   // ---- load: the upper entries of the front, once each
        int code[CLS][CLS];
#pragma unroll
        for (int r = 0; r < CLS; ++r)
#pragma unroll
            for (int c = r; c < CLS; ++c) {
                const int i = r * 16 + gi, j = c * 16 + gj;
                code[r][c] = (i <= j && i < f && j < fprime) ? td[i * fprime + j] : -1;
            }
#pragma unroll
        for (int r = 0; r < CLS; ++r)
#pragma unroll
            for (int c = r; c < CLS; ++c) {
                const int cd = code[r][c];
                Blk v{0.0, 0.0, 0.0, 0.0};
                if (cd <= -2) { const double2 y = load_vec(a.W, (size_t)(-(cd + 2)), b, ld); v.v00 = y.x; v.v10 = y.y; }
                else if (cd >= 0 && !((cd >> 28) & 1)) v = load_blk(a.X, (size_t)(cd & 0x0fffffff), b, ld);
                T[r][c] = v;
                const int i = r * 16 + gi, j = c * 16 + gj;
                if (i == j && i < m) *(double2*)(Dref + (size_t)i * 2) = row_max(v);
            }
    }
    if (prof) pt[1] = wall_clock64();
    {   // ---- extend-add: the upper triangle (+ update vector) of every child's update matrix; front orders are ascending in the pivot number on both
        // sides, so an upper block of the parent is an upper block of the child
        const int* cd = td + h[7];
        for (int ch = 0; ch < nchild; ++ch) {
            const int coff = cd[0], cen = cd[1];
            const int* inv = cd + 2;
            const double* C = stk + coff;
            int ri[CLS], cj[CLS];
#pragma unroll
            for (int r = 0; r < CLS; ++r) { const int i = r * 16 + gi; ri[r] = i < f ? inv[i] : -1; }
#pragma unroll
            for (int c = 0; c < CLS; ++c) { const int j = c * 16 + gj; cj[c] = j < fprime ? inv[j] : -1; }
#pragma unroll
            for (int r = 0; r < CLS; ++r)
#pragma unroll
                for (int c = r; c < CLS; ++c)
                    if (r * 16 + gi <= c * 16 + gj && ri[r] >= 0 && cj[c] >= 0) {
                        const double2* p = (const double2*)(C + ((size_t)ri[r] * (cen + 1) + cj[c]) * 4);
                        const double2 q0 = p[0], q1 = p[1];
                        T[r][c].v00 += q0.x; T[r][c].v01 += q0.y; T[r][c].v10 += q1.x; T[r][c].v11 += q1.y;
                    }
            cd += 2 + fprime;
        }
    }
    if (prof) pt[2] = wall_clock64();
    // ---- publish step 0: the owners of row 0 write the row and, transposed, the column
    if (gi == 0) {
#pragma unroll
        for (int c = 0; c < CLS; ++c) {
            const int j = c * 16 + gj;
            lds_set(Ubuf[0], j, j > 0 ? T[0][c] : zero_blk());
            lds_set(Lbuf[0], j, (j > 0 && j < f) ? blk_t(T[0][c]) : zero_blk());
        }
    }
    if (tid == 0) {
        const Blk d0 = factor_diag(T[0][0], bad, *(const double2*)Dref);
        lds_set(Dbuf[0], 0, d0);
        T[0][0] = d0;
    }
    __syncthreads();
    // ---- pivot steps
    for (int qv = 0; qv < m; ++qv) {
        const int q = uniform(qv);
        const int cur = q & 1, nxt = cur ^ 1;
        Blk D = lds_get(Dbuf[cur], 0);
        D = Blk{uniform_d(D.v00), uniform_d(D.v01), uniform_d(D.v10), uniform_d(D.v11)};
        const int sw = uniform(D.v10 > 2.0 ? 1 : 0);
        const double dl = D.v10 - 4.0 * sw;
        Blk Lq[CLS];
#pragma unroll
        for (int r = 0; r < CLS; ++r) Lq[r] = lds_get(Lbuf[cur], r * 16 + gi);
#pragma unroll
        for (int c = 0; c < CLS; ++c) {
            const double2* p = (const double2*)(Ubuf[cur] + (size_t)(c * 16 + gj) * 4);
            const double2 a0 = p[sw], a1 = p[sw ^ 1];
            Blk z;
            z.v10 = (a1.x - dl * a0.x) * D.v11; z.v00 = (a0.x - D.v01 * z.v10) * D.v00;
            z.v11 = (a1.y - dl * a0.y) * D.v11; z.v01 = (a0.y - D.v01 * z.v11) * D.v00;
#pragma unroll
            for (int r = 0; r <= c; ++r) blk_sub(T[r][c], Lq[r], z);     // (a diagonal class also moves the unused blocks below the diagonal: never read, never stored)
        }
        if (q + 1 < m) {
            const int rq = (q + 1) >> 4, tq = (q + 1) & 15;
            if (gi == tq && gj == tq) {                          // the owner of S(q+1, q+1): final now
                const double2 ref = *(const double2*)(Dref + (size_t)(q + 1) * 2);
#pragma unroll
                for (int r = 0; r < CLS; ++r)
                    if (r == rq) {
                        const Blk dn = factor_diag(T[r][r], bad, ref);
                        lds_set(Dbuf[nxt], 0, dn);
                        T[r][r] = dn;
                    }
            }
            if (gi == tq) {                                      // the owners of row q + 1: the row, and the column below the pivot as its transpose
#pragma unroll
                for (int r = 0; r < CLS; ++r)
                    if (r == rq) {
#pragma unroll
                        for (int c = 0; c < CLS; ++c) {
                            const int j = c * 16 + gj;
                            if (c < r) lds_set(Ubuf[nxt], j, zero_blk());                    // (columns left of the pivot's class: finished; their Lbuf entries belong to the Jordan writers)
                            else {
                                const bool right = j > q + 1;
                                lds_set(Ubuf[nxt], j, right ? T[r][c] : zero_blk());
                                if (j >= q + 1) lds_set(Lbuf[nxt], j, (right && j < f) ? blk_t(T[r][c]) : zero_blk());
                            }
                        }
                    }
            }
            if (gj == tq) {                                      // Jordan: the blocks of column q + 1 ABOVE the pivot, U(i, q + 1), i <= q
#pragma unroll
                for (int c = 0; c < CLS; ++c)
                    if (c == rq) {
#pragma unroll
                        for (int r = 0; r <= c; ++r) {
                            const int i = r * 16 + gi;
                            if (i < q + 1) lds_set(Lbuf[nxt], i, T[r][c]);
                        }
                    }
            }
        }
        __syncthreads();
    }
    if (prof) pt[3] = wall_clock64();
    // ---- store (upper blocks only)
    const int lgo = h[12];
    double2* out = e > 0 ? stack_unit(a, b, h[4], lgo) : nullptr;
    const int jb = h[14];
#pragma unroll
    for (int r = 0; r < CLS; ++r)
#pragma unroll
        for (int c = r; c < CLS; ++c) {
            const int i = r * 16 + gi, j = c * 16 + gj;
            if (i > j) continue;
            const int cd = (i < f && j < fprime) ? td[i * fprime + j] : -1;
            const Blk& v = T[r][c];
            if (i < m && j >= m && j < f) store_blk(a.X, (size_t)(jb + i * e + (j - m)), b, ld, v.v00, v.v01, v.v10, v.v11);
            else if (cd <= -2) store_vec(a.W, (size_t)(-(cd + 2)), b, ld, v.v00, v.v10);
            else if (cd >= 0 && (!((cd >> 28) & 4) || (i == j && i < m))) store_blk(a.X, (size_t)(cd & 0x0fffffff), b, ld, v.v00, v.v01, v.v10, v.v11);
            else if (i >= m && i < f && j >= m && j < fprime) {
                double2* p = out + ((((size_t)(i - m) * (e + 1) + (j - m)) * 2) << lgo);
                p[0] = double2{v.v00, v.v01}; p[(size_t)1 << lgo] = double2{v.v10, v.v11};
            }
        }
    if (bad) atomicOr(a.status + b, 4);
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pt[4] = wall_clock64(); }
}

// per-level launch table: segment ranges and chunk totals
void level_launches(const std::vector<Segment>& segs, std::vector<DevLaunch>& out) {
    out.clear();
    size_t s = 0;
    while (s < segs.size()) {
        DevLaunch d{};
        d.seg_begin = (int)s;
        d.grid = 0; d.nseg = 0;
        d.chain = 0;
        while (true) {
            d.nseg++; d.grid = std::max(d.grid, segs[s].nchunks);
            d.wpi_max = std::max(d.wpi_max, segs[s].wpi);
            if (segs[s].wpi == 0) d.chain = 1;
            if (segs[s++].last) break;
        }
        d.seg_end = (int)s;
        out.push_back(d);
    }
}

}  // namespace

std::mutex& capture_mutex() {
    static std::mutex m;
    return m;
}

SharedPlan::~SharedPlan() {
    hipFree(fact_rec); hipFree(bwd_rec); hipFree(pre_rec); hipFree(fwd_rec); hipFree(sel_rec); hipFree(top_task);
    hipFree(fact_seg); hipFree(bwd_seg); hipFree(pre_seg); hipFree(fwd_seg); hipFree(sel_seg);
    hipFree(pre_row); hipFree(bwd_chain); hipFree(top_data); hipFree(top_wgmap);
    hipFree(bwdj_rec); hipFree(bwdj_seg);
    hipFree(f1_rec); hipFree(f1_wg);
    hipFree(s1_t_jb); hipFree(s1_t_cslot); hipFree(s1_t_toff);
    hipFree(s1_t_row); hipFree(s1_t_ptr); hipFree(s1_t_term); hipFree(s1_t_level); hipFree(s1_b_wg); hipFree(s1_b_row); hipFree(s1_b_term);
}

namespace {
//  The cache is a deliberately LEAKED heap object: idle plans would otherwise be destroyed -- ~17 hipFree each -- during static
// destruction at process exit or dlclose, when the HIP runtime (Python ML frameworks ship their own copy; unload order is not ours) may already be gone.
std::mutex& plan_mutex() { static std::mutex* m = new std::mutex(); return *m; }
std::vector<std::shared_ptr<SharedPlan>>& plans() { static auto* v = new std::vector<std::shared_ptr<SharedPlan>>(); return *v; }   // most recently used last
#define g_plan_mutex plan_mutex()
#define g_plans plans()
constexpr size_t PLAN_CACHE_KEEP = 6;                        // plans without a live engine that the cache keeps around on its own

unsigned long long pattern_hash(int n, const int* rowptr, const int* col, long long policy, int device) {
    unsigned long long h = 1469598103934665603ULL;
    auto mix = [&](unsigned long long v) { h ^= v; h *= 1099511628211ULL; };
    mix((unsigned long long)n); mix((unsigned long long)policy); mix((unsigned long long)device);
    for (int i = 0; i <= n; ++i) mix((unsigned)rowptr[i]);
    for (int p = 0; p < rowptr[n]; ++p) mix((unsigned)col[p]);
    return h;
}
}  // namespace

void clear_plan_cache() {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    g_plans.clear();
}

std::shared_ptr<SharedPlan> acquire_plan(int n, const int* rowptr, const int* col, long long policy, hipStream_t st, std::string& error, int& rc) {
    rc = 0;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { error = "no HIP device"; rc = 2; return nullptr; }
    if (n <= 0 || !rowptr || !col || rowptr[0] != 0) { error = "block pattern must be structurally symmetric with a full diagonal"; rc = 1; return nullptr; }
    static const bool nocache = knob("PLAN_CACHE", 1) == 0;
    const unsigned long long h = pattern_hash(n, rowptr, col, policy, device);
    //  The lock covers the look-up and the publication, NOT the analysis: creates of different patterns (other grids, other devices, other
    // threads) run side by side; a second handle of the SAME key waits for the first one's analysis, then hits.
    struct Pending { unsigned long long hash; long long policy; int device, n; };
    static std::vector<Pending>* pending = new std::vector<Pending>();
    static std::condition_variable* pending_cv = new std::condition_variable();
    auto is_pending = [&] { for (const Pending& q : *pending) if (q.hash == h && q.policy == policy && q.device == device && q.n == n) return true; return false; };
    std::unique_lock<std::mutex> lock(g_plan_mutex);
    for (;;) {
        if (!nocache)
            for (size_t i = 0; i < g_plans.size(); ++i) {
                const std::shared_ptr<SharedPlan>& p = g_plans[i];
                if (p->key_hash == h && p->device == device && p->policy == policy && p->S.n == n && (int)p->key_col.size() == rowptr[n] &&
                    std::equal(rowptr, rowptr + n + 1, p->key_rowptr.begin()) && std::equal(col, col + rowptr[n], p->key_col.begin())) {
                    std::shared_ptr<SharedPlan> hit = p;
                    g_plans.erase(g_plans.begin() + i);
                    g_plans.push_back(hit);
                    return hit;
                }
            }
        if (nocache || !is_pending()) break;
        pending_cv->wait(lock);
    }
    if (!nocache) pending->push_back(Pending{h, policy, device, n});
    lock.unlock();
    struct Done {                                                // whatever way the analysis ends: the key is no longer pending
        std::unique_lock<std::mutex>& lock; std::vector<Pending>* pending; std::condition_variable* cv; unsigned long long h; long long policy; int device, n; bool armed;
        ~Done() {
            if (!armed) return;
            if (!lock.owns_lock()) lock.lock();
            for (size_t i = 0; i < pending->size(); ++i)
                if ((*pending)[i].hash == h && (*pending)[i].policy == policy && (*pending)[i].device == device && (*pending)[i].n == n) { pending->erase(pending->begin() + i); break; }
            cv->notify_all();
        }
    } done{lock, pending, pending_cv, h, policy, device, n, !nocache};
    std::shared_ptr<SharedPlan> p = std::make_shared<SharedPlan>();
    if (analyze(n, rowptr, col, policy, p->S)) { error = "block pattern must be structurally symmetric with a full diagonal"; rc = 1; return nullptr; }
    p->device = device; p->policy = policy; p->key_hash = h;
    p->key_rowptr.assign(rowptr, rowptr + n + 1); p->key_col.assign(col, col + rowptr[n]);
    const BlockSymbolic& S = p->S;
    const double tu0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::vector<int> prow(n, 0);                                 // by ORIGINAL block index: pivot + 1 where the producer finishes D and y
    for (int k = 0; k < n; ++k) if (S.prefactor && S.pre_pivot[k]) prow[S.perm[k]] = k + 1;
    if (upload(&p->pre_rec, S.pre_rec, error, st) || upload(&p->pre_seg, S.pre_seg, error, st) || upload(&p->pre_row, prow, error, st) ||
        upload(&p->fact_rec, S.fact_rec, error, st) || upload(&p->bwd_rec, S.bwd_rec, error, st) || upload(&p->fact_seg, S.fact_seg, error, st) ||
        upload(&p->bwd_seg, S.bwd_seg, error, st) || upload(&p->bwd_chain, S.bwd_chain, error, st) ||
        (S.jordan && (upload(&p->bwdj_rec, S.bwdj_rec, error, st) || upload(&p->bwdj_seg, S.bwdj_seg, error, st))) ||
        upload(&p->fwd_rec, S.fwd_rec, error, st) || upload(&p->fwd_seg, S.fwd_seg, error, st) ||
        (S.single_fact_ok && (upload(&p->f1_rec, S.f_rec, error, st) || upload(&p->f1_wg, S.f1_wg, error, st))) ||
        (!S.top_launch.empty() && (upload(&p->top_task, S.top_task, error, st) || upload(&p->top_data, S.top_data, error, st) || upload(&p->top_wgmap, S.top_wgmap, error, st)))) {
        rc = 2;
        return nullptr;
    }
    if (knob_set("PLAN_TIMING")) fprintf(stderr, "[jg engine]   table upload %6.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - tu0);
    if (!nocache) {
        lock.lock();
        g_plans.push_back(p);
        size_t idle = 0;                                         // plans nobody else uses leave first, oldest first
        for (const auto& q : g_plans) if (q.use_count() == 1) ++idle;
        for (size_t i = 0; i < g_plans.size() && idle > PLAN_CACHE_KEEP;)
            if (g_plans[i].use_count() == 1) { g_plans.erase(g_plans.begin() + i); --idle; } else ++i;
    }
    return p;
}

int Engine::create(int n, const int* rowptr, const int* col, int ld_, long long policy, hipStream_t st) {
    if (ld_ <= 0 || ld_ % 64) { error = "batch leading dimension must be a positive multiple of 64"; return 1; }
    // Where the multifrontal top starts depends on the batch: a top task occupies a workgroup per scenario (~10 us + ~1 us per
    // pivot), a level launch a wave per 64 scenarios.  Up to 128 scenarios every level that holds fewer than ~400 items is
    // cheaper as tasks (latency regime); from 256 on only the levels with at most 4 pivots -- parallel chains -- are (measured
    // at 512 scenarios: case1354pegase 0.190 -> 0.154 ms, 9241-bus grid 0.853 -> 0.777, ACTIVSg10k 1.60 -> 1.46 with 40
    // launches instead of 84; with the small-batch threshold 0.221 / 0.849 / 1.62).
    // small batches: fronts of 26 block rows on the large grids (measured at 1, 64 and 128 scenarios with caps 24 / 26 / 28 / 30: ACTIVSg10k
    // 0.358 / 0.339 / 0.337 / 0.344 ms at 64, the 9241-bus grid 0.213 / 0.202 / 0.207 / 0.217), 24 on small ones (case1354pegase 0.088 / 0.094 / ...)
    // large batches: the top starts where a level holds at most 384 items and 8 pivots on the large grids (ACTIVSg10k at 512 scenarios: 1.279 -> 1.241 ms
    // against 280 items / 4 pivots, which the small grids keep: case1354pegase 0.148 against 0.160 ms; the 9241-bus grid does not care)
    const bool defaults = !((policy >> 16) & 0x7fff);
    if (defaults && ld_ >= 256) policy |= 8;
    // round 4: large batches factorise the bottom of the tree in TASKS (jg_symbolic.hpp: shared operands of a pivot row / column staged in LDS, one
    // block load per update term instead of three).  JG_ROW_TASKS=0: the wave records of round 3; =2: tasks for every batch size.
    {
        static const int tasks_env = knob("ROW_TASKS", 1);
        if (defaults && !((policy >> 32) & 0xff) && (tasks_env == 2 || (tasks_env == 1 && ld_ >= 256))) policy |= 1LL << 50;
    }
    // round 3 (the top launches got cheaper: Jordan rows, 4-wave variant where it pays): on the large grids a large batch starts the top where a level
    // holds at most 12 pivots, whatever its item count (narrow = 127: no limit) -- ACTIVSg10k: level 22 instead of 27, 1.236 -> 1.213 ms at 512
    // scenarios, three interleaved runs each; the 9241-bus grid does not care; symmetric plans -- the Gauss-Newton gain -- keep the rule of round 2: 2.96 against 2.99 ms for factorisation + sweep
    constexpr bool top_r02 = false;                               // (the rule of round 2 -- 384 items, 8 pivots -- stays for symmetric plans: below)
    // a handful of scenarios (the owner sets `lanes` before create: a single power flow, up to 32 scenarios) of an unsymmetric matrix: the top starts as
    // low as the level schedule allows (no item limit) -- a task costs one workgroup per REAL scenario, and since the Jordan rows the pivots of the top
    // are the cheap ones of the backward sweep as well.  ACTIVSg10k, one scenario: factorisation 0.283 -> 0.244 ms, backward sweep 0.104 -> 0.060,
    // 2.04 -> 1.67 ms per solve (9241-bus grid 1.76 -> 1.58); 32 scenarios 0.419 -> 0.363; from 64 scenarios on and for the gain matrices (whose
    // factorisation loses what their sweep wins) the threshold stays 384 items.
    const bool tiny = !top_r02 && ld_ == 64 && lanes > 0 && lanes <= 32 && !(policy & 2);
    // one lane group with more than 32 scenarios, large grids: the top starts where a level holds at most 36 pivots, whatever its item count (ACTIVSg10k at 64
    // scenarios: level 12 instead of 17, factorisation + backward sweep 0.466 -> 0.417 ms; the 9241-bus grid 0.283 -> 0.29; two lane groups and more: no difference)
    const bool lane64 = !top_r02 && ld_ == 64 && !tiny && n >= 4000 && !(policy & 2);
    if (defaults) policy |= ld_ >= 256 ? (n >= 4000 ? (top_r02 || (policy & 2) ? (47 << 16 | (384 / 8) << 24 | 8 << 4) : (47 << 16 | 127 << 24 | 12 << 4)) : (47 << 16 | (280 / 8) << 24 | 4 << 4))
                                              : ((n >= 4000 ? 26 : 24) << 16 | (tiny || lane64 ? 127 : 384 / 8) << 24 | (lane64 ? 15 << 4 : 0));
    // round 6, a handful of scenarios: a top task takes at least 12 pivots instead of 8 where its front has room (jg_symbolic.hpp, policy bits 54-59).  The
    // critical path of a single instance of the 10k-bus grid is a chain of 8-pivot tasks with 25-34 external rows -- 20 us each, 13 of them gather / extend-add /
    // store -- and a lone workgroup steps through a class-4 front as fast as through a class-3 one: 1.603 -> 1.507 ms per solve with 12 (10: 1.549, 16: 1.500,
    // 20: 1.556, 24: 1.66; the 9241-bus grid 1.488 / 1.495 / 1.490 / 1.54 for 8 / 10 / 12 / 16; case1354pegase does not care; a larger soft cap loses everywhere:
    // tools/experiments/r06_mmin_sweep.sh, profiles/r06_mmin_sweep.txt)
    if (defaults && tiny && !((policy >> 54) & 0x3f)) policy |= 12LL << 54;
    if (defaults && tiny && lanes == 1 && knob("SINGLE", 1) != 0) policy |= 1LL << 60;      // ONE scenario: the plan also carries the thread-per-item tables of the bottom (k_fact1)
    const bool timing = knob_set("PLAN_TIMING");
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double te0 = tnow();
    {
        int rc = 0;
        plan = acquire_plan(n, rowptr, col, policy, st, error, rc);
        if (!plan) return rc;
    }
    if (timing) fprintf(stderr, "[jg engine] plan (analysis or cache hit, table upload) %6.1f ms\n", tnow() - te0);
    ld = ld_;
    level_launches(plan->S.fact_seg, fact);
    level_launches(plan->S.pre_seg, pre);
    level_launches(plan->S.bwd_seg, bwd);
    level_launches(plan->S.bwdj_seg, bwdj);
    level_launches(plan->S.fwd_seg, fwd);
    jordan = plan->S.jordan && knob("JORDAN", 1) != 0;
    // ONE scenario on a Jordan plan: the tables of the row-per-lane sweep (built and uploaded by the first such engine of the plan; JG_SINGLE=0: the level launches)
    if (ld == 64 && lanes == 1 && jordan && knob("SINGLE", 1) != 0) {
        std::lock_guard<std::mutex> lock(plan->single_mutex);
        if (!plan->single_tried) {
            plan->single_tried = true;
            SingleTables& T = plan->single;
            build_single_tables(plan->S, T);
            if (T.ok && ((size_t)T.n_top * 40 > 144 * 1024 || T.n_top_levels > 62)) T.ok = false;      // the top's solution (and its level table) must fit the LDS of one workgroup
            if (T.ok) {
                if (upload(&plan->s1_t_row, T.t_row, error, st) || upload(&plan->s1_t_ptr, T.t_ptr, error, st) || upload(&plan->s1_t_term, T.t_term, error, st) ||
                    upload(&plan->s1_t_level, T.t_level, error, st) || upload(&plan->s1_b_wg, T.b_wg, error, st) || upload(&plan->s1_b_row, T.b_row, error, st) ||
                    upload(&plan->s1_b_term, T.b_term, error, st)) { T.ok = false; return 2; }      // (a later engine of the plan must not take tables that are not there)
                // terms as lanes (k_bwd1_top2): a level's rows on one thread each, its terms in BWD1_ROUNDS rounds of the workgroup, everything in its LDS
                if (T.flat_ok && (T.max_level_rows > 1024 || T.max_level_terms > BWD1_ROUNDS * 1024 || (size_t)T.n_top * 36 + (size_t)T.max_level_terms * 16 > 144 * 1024)) T.flat_ok = false;
                if (T.flat_ok && (upload(&plan->s1_t_jb, T.t_jb, error, st) || upload(&plan->s1_t_cslot, T.t_cslot, error, st) || upload(&plan->s1_t_toff, T.t_toff, error, st))) { T.ok = false; return 2; }
                for (std::vector<int>* v : {&T.t_row, &T.t_ptr, &T.t_term, &T.t_level, &T.b_wg, &T.b_row, &T.b_term, &T.t_jb, &T.t_cslot, &T.t_toff}) std::vector<int>().swap(*v);
            }
        }
        single_bwd = plan->single.ok;
        if (single_bwd) {
            JG_HIP(hipFuncSetAttribute((const void*)k_bwd1_top, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            JG_HIP(hipFuncSetAttribute((const void*)k_bwd1_top2, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            const size_t jb = (size_t)std::max(plan->S.n_jordan, 1) * 4 * sizeof(double);
            JG_HIP(hipMalloc((void**)&jc, jb));
            JG_HIP(sync_fill(jc, 0, jb, st));
        }
    }
    // the device tables belong to the plan; the engine keeps plain aliases for its launches
    fact_rec = plan->fact_rec; bwd_rec = plan->bwd_rec; pre_rec = plan->pre_rec; fwd_rec = plan->fwd_rec;
    fact_seg = plan->fact_seg; bwd_seg = plan->bwd_seg; pre_seg = plan->pre_seg; fwd_seg = plan->fwd_seg;
    pre_row = plan->pre_row; bwd_chain = plan->bwd_chain; top_task = plan->top_task; top_data = plan->top_data; top_wgmap = plan->top_wgmap;
    JG_HIP(hipFuncSetAttribute((const void*)k_bwd_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CHAIN_LDS_D2 * sizeof(double2))));
    JG_HIP(hipFuncSetAttribute((const void*)k_fact_task, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(TASK_LDS_D2 * sizeof(double2))));
    if (!plan->S.top_launch.empty()) {
        const size_t sb = (size_t)(std::max<long long>(plan->S.top_stack_cls[0], 2) + plan->S.top_stack_cls[1] + plan->S.top_stack_cls[2]) * ld * sizeof(double);
        JG_HIP(hipMalloc((void**)&top_stack, sb));
        JG_HIP(sync_fill(top_stack, 0, sb, st));
        if (knob_set("TOP_PROFILE")) {
            JG_HIP(hipMalloc((void**)&top_prof, plan->S.top_task.size() * ld * 8 * sizeof(long long)));
            JG_HIP(sync_fill(top_prof, 0, plan->S.top_task.size() * ld * 8 * sizeof(long long), st));
        }
    }
    JG_HIP(hipMalloc((void**)&X, factor_bytes()));
    JG_HIP(sync_fill(X, 0, factor_bytes(), st));
    JG_HIP(hipMalloc((void**)&W, (size_t)n * 2 * ld * sizeof(double)));
    JG_HIP(sync_fill(W, 0, (size_t)n * 2 * ld * sizeof(double), st));
    JG_HIP(hipMalloc((void**)&status, (size_t)ld * sizeof(int)));
    JG_HIP(sync_fill(status, 0, (size_t)ld * sizeof(int), st));
    JG_HIP(hipGetDevice(&device));
    if (timing) fprintf(stderr, "[jg engine] + factor storage, stacks               %6.1f ms\n", tnow() - te0);
    return 0;
}

void Engine::destroy() {
    if (top_prof) {                                              // phase times of every task (scenario 0, last factorisation)
        std::vector<long long> t(plan->S.top_task.size() * ld * 8);
        if (hipMemcpy(t.data(), top_prof, t.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[jg top profile] task level class m e | load children steps store total (us) | us per step | batch: workgroups, start spread, "
                            "total min / median / max, first start -> last end, CUs used, most workgroups on one CU\n");
            for (size_t i = 0; i < plan->S.top_task.size(); ++i) {
                const long long* p = &t[i * ld * 8];
                if (!p[0]) continue;
                const Rec& h = plan->S.top_task[i];
                fprintf(stderr, "[jg top profile] %3zu %2d %d %2d %2d | %6.2f %6.2f %6.2f %6.2f %7.2f | %5.3f", i, h.w[10], h.w[9], h.w[0], h.w[1],
                        (p[1] - p[0]) * 0.01, (p[2] - p[1]) * 0.01, (p[3] - p[2]) * 0.01, (p[4] - p[3]) * 0.01, (p[4] - p[0]) * 0.01, (p[3] - p[2]) * 0.01 / h.w[0]);
                std::vector<long long> st0, tot; long long e1 = 0; std::vector<int> cu;
                for (int b = 0; b < ld; ++b) { const long long* q = p + (size_t)b * 8; if (q[0] && q[4]) { st0.push_back(q[0]); tot.push_back(q[4] - q[0]); e1 = std::max(e1, q[4]); cu.push_back((int)q[5]); } }
                if (!st0.empty()) {
                    const long long s0 = *std::min_element(st0.begin(), st0.end()), s1 = *std::max_element(st0.begin(), st0.end());
                    std::sort(tot.begin(), tot.end()); std::sort(cu.begin(), cu.end());
                    int ncu = 0, most = 0, run = 0;
                    for (size_t x = 0; x < cu.size(); ++x) { if (x == 0 || cu[x] != cu[x - 1]) { ++ncu; run = 0; } most = std::max(most, ++run); }
                    fprintf(stderr, " | %zu %6.2f  %6.2f / %6.2f / %6.2f  %7.2f  %d %d", st0.size(), (s1 - s0) * 0.01, tot.front() * 0.01, tot[tot.size() / 2] * 0.01, tot.back() * 0.01,
                            (e1 - s0) * 0.01, ncu, most);
                }
                if (JG_PROBE_STEP && p[6]) fprintf(stderr, " | step clocks: read %lld update %lld publish %lld barrier %lld", (p[6] >> 32) / h.w[0], (p[6] & 0xffffffffll) / h.w[0], (p[7] >> 32) / h.w[0], (p[7] & 0xffffffffll) / h.w[0]);
                fprintf(stderr, "\n");
            }
        }
        hipFree(top_prof); top_prof = nullptr;
    }
    fwd_rec = nullptr; fwd_seg = nullptr; sel_rec = nullptr; sel_seg = nullptr;                  // tables: the plan's (freed with its last user)
    top_task = nullptr; top_data = nullptr; top_wgmap = nullptr; bwd_chain = nullptr;
    fact_rec = bwd_rec = nullptr; fact_seg = bwd_seg = nullptr; pre_rec = nullptr; pre_seg = nullptr; pre_row = nullptr;
    hipFree(Zs); Zs = nullptr;
    hipFree(top_stack); top_stack = nullptr;
    hipFree(jc); jc = nullptr; single_bwd = false;
    hipFree(X); hipFree(W); hipFree(status);
    status = nullptr;
    X = W = nullptr;
    plan.reset();
}

int Engine::factor(hipStream_t st, const double* A, const double* rhs, const GroupSel& sel, bool level0_done) {
    if (!plan->S.inplace && !A) { error = "factor: no source matrix"; return 1; }
    FactArgs a{fact_rec, fact_seg, plan->S.inplace ? X : A, rhs, X, W, status, sel, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    if (!level0_done && !pre.empty()) {                          // prefactor plan, plain blocks from the producer: its level 0 first
        FactArgs p = a;
        p.rec = pre_rec; p.seg = pre_seg;
        for (const DevLaunch& L : pre) {
            p.seg_begin = L.seg_begin;
            { const Segment& g = plan->S.pre_seg[L.seg_begin]; p.s0_base = g.rec_base; p.s0_nchunks = g.nchunks; p.s0_wpi = g.wpi; p.s0_rpw = g.rpw; }
            hipLaunchKernelGGL(k_fact_level, dim3(grid_blocks(ld / 64, (long long)L.grid * (16 / FACT_WAVES)), L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 256 * sizeof(double), st, p);
        }
    }
    const bool single_fact = plan->S.single_fact_ok && lanes == 1 && plan->f1_rec && probe_part == 0;
    if (single_fact) {                                           // ONE scenario: a thread per item -- the bottom subtrees in one launch, the partial sums of the task-owned items in a second
        Fact1Args s{plan->f1_rec, plan->f1_wg, (int)plan->S.f1_first.size(), rhs, X, W, status, sel, ld, plan->S.n_f1_wg, 0};
        if (s.n_wg > 0) hipLaunchKernelGGL(k_fact1_bottom, dim3(grid_blocks(ld / 64, s.n_wg)), dim3(256), 0, st, s);
        s.n_items = (int)plan->S.f2_first.size();
        if (s.n_items > 0) hipLaunchKernelGGL(k_fact1_partial, dim3(grid_blocks(ld / 64, (s.n_items + 63) / 64)), dim3(256), 0, st, s);
    }
    for (const DevLaunch& L : fact) {
        if (probe_part == 2 || single_fact) break;
        a.seg_begin = L.seg_begin;
        { const Segment& g = plan->S.fact_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        if (plan->S.fact_tasks)                                  // TASKS (jg_symbolic.hpp): a workgroup per task, not per 8 item waves
            hipLaunchKernelGGL(k_fact_task, dim3(grid_blocks(ld / 64, L.grid), L.nseg), dim3(64, TASK_WAVES), TASK_LDS_D2 * sizeof(double2), st, a);
        else
            hipLaunchKernelGGL(k_fact_level, dim3(grid_blocks(ld / 64, (long long)L.grid * (16 / FACT_WAVES)), L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 256 * sizeof(double), st, a);
    }
    // the top of the elimination tree: multifrontal tasks, one workgroup per (task, scenario), launch = (task level, class)
    if (!plan->S.top_launch.empty() && probe_part != 1) {
        const long long s0 = std::max<long long>(plan->S.top_stack_cls[0], 2);
        TopArgs t{top_task, top_data, X, W, top_stack, status, sel, s0, {0, s0 * ld, (s0 + plan->S.top_stack_cls[1]) * ld}, {s0, plan->S.top_stack_cls[1], plan->S.top_stack_cls[2]},
                  top_prof, ld, a.lanes, 0, 0, 64, top_wgmap, 0, 0, single_bwd && jordan ? jc : nullptr, plan->S.n_entries};
        if (ld == 64 && t.lanes < 64) t.lpg = t.lanes;
        for (const TopLaunch& L : plan->S.top_launch) {
            t.task_begin = L.task_begin; t.ntasks = L.ntasks;
            const bool jordan = this->jordan && plan->S.jordan;
            if (L.grouped) {                                     // plans with a "mid" policy (jg_symbolic.hpp): their kernel was retired with its measurements
                error = "this plan holds grouped top tasks (policy bits 32-39): k_fact_grp left the library in round 6 (tools/experiments/r06_retired_kernels.patch)";
                return 1;
            }
            // round 5: symmetric plans with Jordan rows (the Gauss-Newton gain by default) eliminate the upper triangle only (k_fact_top_sym); JG_TOP_SYM=0: the mirrored front
            static const bool sym_env = knob("TOP_SYM", 1) != 0;
            if (plan->S.symmetric && jordan && sym_env) {
                const dim3 grids(grid_blocks(ld / 64, (long long)L.ntasks * t.lpg));
                if (L.cls == 2) hipLaunchKernelGGL((k_fact_top_sym<2>), grids, dim3(256), 0, st, t);
                else if (L.cls == 3) hipLaunchKernelGGL((k_fact_top_sym<3>), grids, dim3(256), 0, st, t);
                else hipLaunchKernelGGL((k_fact_top_sym<4>), grids, dim3(256), 0, st, t);
                continue;
            }
            const dim3 grid(grid_blocks(ld / 64, (long long)L.ntasks * t.lpg));
            // More workgroups than the CUs can hold WITH a pivot wave (1 per CU at CLS = 4, 2 at CLS = 3, 3 at CLS = 2): the 4-wave
            // variant, of which a CU holds twice as many; else the pivot-wave variant, whose step is 15 % shorter (measured at 512
            // scenarios: 0.86 against 1.02 us per step with two workgroups on a CU).
            static const int pw_env = knob("TOP_PW", -1);
            const long long wgs = (long long)L.ntasks * std::min<long long>(t.lanes, (long long)t.lpg * (ld / 64));
            // (shared: the handle is one of several batches in flight on this GPU -- a CU with two 4-wave top workgroups still has room for a level
            // workgroup of another batch, with 5-wave ones it has not: +2-3 % on the 512 x 3 pipeline, -0.6 % on a lone factorisation; same bits)
            const bool pw = pw_env >= 0 ? pw_env != 0 : (!shared && wgs <= 256 * (L.cls == 4 ? 1 : (L.cls == 3 ? 2 : 3)));
            static const int fuse_env = knob("TOP_FUSE", 0);
            if (jordan) {                                        // Jordan rows (jg_symbolic.hpp); the plain sweep's tables would read garbage after this
                if (pw) {
                    if (L.cls == 2) hipLaunchKernelGGL((k_fact_top<2, true, false, true>), grid, dim3(TOP_THREADS), 0, st, t);
                    else if (L.cls == 3) hipLaunchKernelGGL((k_fact_top<3, true, false, true>), grid, dim3(TOP_THREADS), 0, st, t);
                    else hipLaunchKernelGGL((k_fact_top<4, true, false, true>), grid, dim3(TOP_THREADS), 0, st, t);
                } else {
                    if (L.cls == 2) hipLaunchKernelGGL((k_fact_top<2, false, false, true>), grid, dim3(256), 0, st, t);
                    else if (L.cls == 3) hipLaunchKernelGGL((k_fact_top<3, false, false, true>), grid, dim3(256), 0, st, t);
                    else hipLaunchKernelGGL((k_fact_top<4, false, false, true>), grid, dim3(256), 0, st, t);
                }
            } else if (fuse_env && L.cls <= 3) {                        // two pivots per barrier (k_fact_top<CLS, false, true>; at 4 x 4 blocks per thread
                                                                 // the second pivot's row / column do not fit the register file beside the front)
                if (L.cls == 2) hipLaunchKernelGGL((k_fact_top<2, false, true>), grid, dim3(256), 0, st, t);
                else hipLaunchKernelGGL((k_fact_top<3, false, true>), grid, dim3(256), 0, st, t);
            } else if (pw) {
                if (L.cls == 2) hipLaunchKernelGGL((k_fact_top<2, true>), grid, dim3(TOP_THREADS), 0, st, t);
                else if (L.cls == 3) hipLaunchKernelGGL((k_fact_top<3, true>), grid, dim3(TOP_THREADS), 0, st, t);
                else hipLaunchKernelGGL((k_fact_top<4, true>), grid, dim3(TOP_THREADS), 0, st, t);
            } else {
                if (L.cls == 2) hipLaunchKernelGGL((k_fact_top<2, false>), grid, dim3(256), 0, st, t);
                else if (L.cls == 3) hipLaunchKernelGGL((k_fact_top<3, false>), grid, dim3(256), 0, st, t);
                else hipLaunchKernelGGL((k_fact_top<4, false>), grid, dim3(256), 0, st, t);
            }
        }
    }
    JG_HIP(hipGetLastError());
    return 0;
}

int Engine::selected_inverse(hipStream_t st, const GroupSel& sel) {
    if (!Zs) {                                          // tables (the plan's, built once) and storage (this engine's) on first use
        {
            std::lock_guard<std::mutex> lock(plan->sel_mutex);
            if (!plan->sel_ready) {
                build_selected_inverse(plan->S);
                if (upload(&plan->sel_rec, plan->S.sel_rec, error, st) || upload(&plan->sel_seg, plan->S.sel_seg, error, st)) {
                    hipFree(plan->sel_rec); hipFree(plan->sel_seg); plan->sel_rec = nullptr; plan->sel_seg = nullptr;   // a retry starts clean
                    return 2;
                }
                plan->sel_ready = true;
            }
        }
        level_launches(plan->S.sel_seg, selv);
        sel_rec = plan->sel_rec; sel_seg = plan->sel_seg;
        JG_HIP(hipMalloc((void**)&Zs, factor_bytes()));
        JG_HIP(sync_fill(Zs, 0, factor_bytes(), st));
    }
    SelArgs a{sel_rec, sel_seg, X, Zs, sel, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    for (const DevLaunch& L : selv) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = plan->S.sel_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL(k_sel_level, dim3(grid_blocks(ld / 64, (long long)L.grid * (16 / FACT_WAVES)), L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 128 * sizeof(double2), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

int Engine::forward(hipStream_t st, const double* rhs, const GroupSel& sel) {
    FactArgs a{fwd_rec, fwd_seg, X, rhs, X, W, status, sel, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    for (const DevLaunch& L : fwd) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = plan->S.fwd_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL(k_fact_level, dim3(grid_blocks(ld / 64, (long long)L.grid * (16 / FACT_WAVES)), L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 256 * sizeof(double), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

namespace {
// X[entry of block p] <- blocks[p] for every scenario
__global__ void k_fill_shared(const double* blocks, const int* src_entry, double* X, int nnz, int ld) {
    const int p = blockIdx.x * blockDim.y + threadIdx.y;
    if (p >= nnz) return;
    const size_t b = (size_t)blockIdx.y * 64 + threadIdx.x;
    const double* v = blocks + (size_t)p * 4;
    store_blk(X, (size_t)src_entry[p], b, (size_t)ld, v[0], v[1], v[2], v[3]);
}
}  // namespace

int Engine::set_shared_matrix(hipStream_t st, const double* blocks_host) {
    if (!plan->S.inplace) { error = "set_shared_matrix needs an in-place engine"; return 1; }
    const int nnz = (int)plan->S.src_entry.size();
    double* dblk = nullptr; int* dmap = nullptr;
    std::vector<double> hb(blocks_host, blocks_host + (size_t)nnz * 4);
    if (upload(&dblk, hb, error, st) || upload(&dmap, plan->S.src_entry, error, st)) { hipFree(dblk); hipFree(dmap); return 2; }
    hipLaunchKernelGGL(k_fill_shared, dim3((nnz + 3) / 4, ld / 64), dim3(64, 4), 0, st, dblk, dmap, X, nnz, ld);
    hipError_t e = hipStreamSynchronize(st);
    hipFree(dblk); hipFree(dmap);
    JG_HIP(e);
    return 0;
}

int Engine::backsolve(hipStream_t st, double* out, const StateUpdate& upd, const GroupSel& sel) {
    if (single_bwd && jordan) {                                  // ONE scenario: rows per lane, two launches (k_bwd1_top, k_bwd1_bottom)
        const SingleTables& T = plan->single;
        Bwd1Args s{plan->s1_t_row, plan->s1_t_ptr, plan->s1_t_term, plan->s1_t_level, plan->s1_b_wg, plan->s1_b_row, plan->s1_b_term, X, jc, W, out, sel, upd, ld, T.n_top_levels, T.n_wg, T.n_top,
                   plan->s1_t_jb, plan->s1_t_cslot, plan->s1_t_toff, T.max_level_terms};
        const int flat_env = knob("SINGLE", 1);                   // JG_SINGLE=2: the quad-per-row sweep (k_bwd1_top) where the terms-as-lanes one would run
        if (T.flat_ok && flat_env != 2) hipLaunchKernelGGL(k_bwd1_top2, dim3(grid_blocks(ld / 64, 1)), dim3(1024), (size_t)T.n_top * 36 + (size_t)T.max_level_terms * 16, st, s);
        else hipLaunchKernelGGL(k_bwd1_top, dim3(grid_blocks(ld / 64, 1)), dim3(1024), (size_t)T.n_top * 40, st, s);
        if (T.n_wg > 0) hipLaunchKernelGGL(k_bwd1_bottom, dim3(grid_blocks(ld / 64, T.n_wg)), dim3(SINGLE_BOTTOM_ROWS), 0, st, s);
        JG_HIP(hipGetLastError());
        return 0;
    }
    // jordan: the last factor() left Jordan rows (the flag must not change between a factorisation and its solves)
    BwdArgs a{jordan ? plan->bwdj_rec : bwd_rec, jordan ? plan->bwdj_seg : bwd_seg, bwd_chain, X, W, out, sel, upd, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    const std::vector<Segment>& segs = jordan ? plan->S.bwdj_seg : plan->S.bwd_seg;
    for (const DevLaunch& L : (jordan ? bwdj : bwd)) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = segs[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        if (!L.chain && L.wpi_max <= 8)                         // 0.387 -> 0.381 ms at 512 scenarios
            hipLaunchKernelGGL(k_bwd_level8, dim3(grid_blocks(ld / 64, (long long)L.grid * 2), L.nseg), dim3(64, 8), CHAIN_SMALL_LDS_D2 * sizeof(double2), st, a);
        else
        hipLaunchKernelGGL(k_bwd_level, dim3(grid_blocks(ld / 64, L.grid), L.nseg), dim3(64, 16),
                           L.chain ? (size_t)CHAIN_LDS_D2 * sizeof(double2) : 16 * 128 * sizeof(double), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

}  // namespace jg
