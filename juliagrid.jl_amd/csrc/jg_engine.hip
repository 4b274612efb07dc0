// jg_engine.hip -- device kernels of the batched block-sparse LU engine (gfx950, wave64).
//
// Factorisation A = Lh * inv(D) * U (Lh unscaled, D kept as 2x2 LU factors) with the forward elimination of the right-hand
// side fused in, then one backward sweep.  Lanes = 64 scenarios of one group; a wave works on one structural item, `wpi`
// waves share an item's update list (dealt round-robin, partial sums meet in LDS in a fixed order => bitwise
// run-to-run determinism).
//
// Replay tables (jg_symbolic.hpp): each wave's work is a 64-byte RECORD whose address follows from
// (segment, chunk, wave) arithmetic, fetched with ONE scalar load -- no descriptor -> index -> value pointer chase
// (that chain cost ~10 us per item, measured).  Two executors replay the same tables:
//   * per-level launches  (k_fact_level / k_bwd_level): one kernel per dependency level, whole chip per level;
//   * persistent walker   (k_fact_walk / k_bwd_walk):   ONE launch, XCD teams, team-local level barriers (~0.7 us),
//     records prefetched one chunk ahead -- across barriers too, the tables are static.
#include "jg_engine.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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

// one record of a factorisation item: up to FACT_T update terms, every operand load issued before the first use
__device__ __forceinline__ void fact_record(const FactArgs& a, const RecS& r, int kind, size_t b, size_t ld, Blk& c) {
    const int nt = rec_word(r, 3);
    Blk l[FACT_T], d[FACT_T], u[FACT_T];
#pragma unroll
    for (int t = 0; t < FACT_T; ++t) {
        if (t < nt) {
            const int ia = rec_word(r, 4 + 3 * t);              // bit 30: read the block transposed (symmetric matrices: Lh(i,k) = U(k,i)')
            l[t] = load_blk(a.X, (size_t)(ia & 0x3fffffff), b, ld);
            if (ia >> 30) { const double x = l[t].v01; l[t].v01 = l[t].v10; l[t].v10 = x; }
            d[t] = load_blk(a.X, (size_t)rec_word(r, 5 + 3 * t), b, ld);
            if (kind == 3) {
                const double2 w = load_vec(a.W, (size_t)rec_word(r, 6 + 3 * t), b, ld);
                u[t].v00 = w.x; u[t].v10 = w.y; u[t].v01 = 0.0; u[t].v11 = 0.0;
            } else {
                u[t] = load_blk(a.X, (size_t)rec_word(r, 6 + 3 * t), b, ld);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < FACT_T; ++t) {
        if (t < nt) {
            if (kind == 3) {          // y -= Lh(a) * D(d)^-1 * y_c
                double z0, z1;
                dsolve(d[t], u[t].v00, u[t].v10, z0, z1);
                c.v00 -= l[t].v00 * z0 + l[t].v01 * z1;
                c.v01 -= l[t].v10 * z0 + l[t].v11 * z1;
            } else {
                term3(c, l[t], d[t], u[t]);
            }
        }
    }
}

__device__ __forceinline__ void fact_finish(const FactArgs& a, int kind, int id, size_t b, size_t ld, const Blk& c) {
    if (kind == 3) {
        store_vec(a.W, (size_t)id, b, ld, c.v00, c.v01);
        return;
    }
    if (kind == 2) {                            // diagonal block: 2x2 LU with in-block partial pivoting
        const bool sw = fabs(c.v10) > fabs(c.v00);
        const double u11 = sw ? c.v10 : c.v00, u12 = sw ? c.v11 : c.v01;
        const double o21 = sw ? c.v00 : c.v10, o22 = sw ? c.v01 : c.v11;
        const double iu11 = 1.0 / u11;
        const double l = o21 * iu11;
        const double u22 = o22 - l * u12;
        const double iu22 = 1.0 / u22;
        if (!(fabs(u11) > 0.0) || !(fabs(u22) > 0.0) || !(fabs(iu11) < 1.0e300) || !(fabs(iu22) < 1.0e300)) atomicOr(a.status + b, 4);
        store_blk(a.X, (size_t)id, b, ld, iu11, u12, sw ? l + 4.0 : l, iu22);
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
    if (kind >= 0) {
        if (sub == 0) {
            if (kind == 3) { const double2 f = load_vec(a.rhs, (size_t)src, b, ld); c.v00 = f.x; c.v01 = f.y; }
            else if (src >= 0) c = load_blk(a.A, (size_t)src, b, ld);
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
        fact_finish(a, kind, id, b, ld, c);
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
    store_vec(a.out, (size_t)bus, b, ld, x0, x1);
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
    if (!map_block(a.sel, a.ld, BWD ? nchunks * (16 / BW) : nchunks * (16 / FACT_WAVES), grp, bx)) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min(grp * 64 + lane, a.lanes - 1);
    if constexpr (BWD) {
        if (wpi == 0) {                                          // chain segment: one task (one record) per workgroup
            bwd_chain_task(a, red, load_rec(a.rec, (size_t)base + bx), wave, lane, b, ld);
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

// ---- selected inverse (Takahashi recursion on the factor pattern, symmetric matrices; tables: jg_symbolic.cpp) ----------
struct SelArgs {
    const Rec* rec; const Segment* seg;
    const double* X; double* Z; GroupSel sel;
    int ld, seg_begin, lanes;
    int s0_base, s0_nchunks, s0_wpi, s0_rpw;
};

// T += U(i,k) * Z(k,j)   (Z entry read transposed when bit 30 of its id is set)
__device__ __forceinline__ void sel_record(const SelArgs& a, const RecS& r, size_t b, size_t ld, Blk& t) {
    const int nt = rec_word(r, 3);
    Blk u[BWD_T], z[BWD_T];
#pragma unroll
    for (int q = 0; q < BWD_T; ++q) {
        if (q < nt) {
            u[q] = load_blk(a.X, (size_t)rec_word(r, 4 + 2 * q), b, ld);
            z[q] = load_blk(a.Z, (size_t)(rec_word(r, 5 + 2 * q) & 0x3fffffff), b, ld);
        }
    }
#pragma unroll
    for (int q = 0; q < BWD_T; ++q) {
        if (q < nt) {
            const bool tr = (rec_word(r, 5 + 2 * q) >> 30) & 1;
            const double z01 = tr ? z[q].v10 : z[q].v01, z10 = tr ? z[q].v01 : z[q].v10;
            t.v00 += u[q].v00 * z[q].v00 + u[q].v01 * z10;
            t.v01 += u[q].v00 * z01 + u[q].v01 * z[q].v11;
            t.v10 += u[q].v10 * z[q].v00 + u[q].v11 * z10;
            t.v11 += u[q].v10 * z01 + u[q].v11 * z[q].v11;
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

// ---- executor 2: persistent level walker ----------------------------------------------------------------------
// One launch replays ALL dependency levels.  The chip's workgroups (one 16-wave workgroup per CU) form TEAMS by
// the XCD they physically run on (HW_REG_XCC_ID, read at run time -- nothing is assumed about blockIdx -> XCD
// placement).  A 64-scenario group belongs to exactly one team for the whole walk, so every value that crosses a
// level travels CU -> that XCD's L2 -> CU and the level barrier is a team-local arrival counter (0.6-0.8 us measured)
// instead of a kernel boundary.
// Visibility argument (no fences needed; holds for ANY placement because the teams ARE the physical XCDs):
//   * every factor entry / rhs row is written exactly ONCE per launch (by one wave, whole 128-byte lines) and read by
//     other CUs only after the barrier that follows its level, so no CU's L1 can hold a line older than the launch
//     (L1 is invalidated at kernel start and fills on demand only); rows rewritten in place (y -> x in the backward
//     sweep) are read before the rewrite by their owner wave alone;
//   * producer and consumers of a group share one L2 (the coherence point of an XCD); `s_waitcnt vmcnt(0)` before the
//     arrival makes the write-through stores L2-visible;
//   * counters are agent-scope atomics (L1-bypassing), polled by one lane per workgroup.
// Every spin is bounded (wall clock): a stalled walk sets the error words and all workgroups drain out.
enum { SYNC_REG = 0, SYNC_ERR = 1, SYNC_TEAM = 16 /* +xcc: team size */, SYNC_BAR = 64 /* +32*xcc: arrivals */,
       SYNC_WORDS = 64 + 32 * 16 /* zeroed before every walk; word [SYNC_WORDS] is a sticky error flag */ };

struct WalkArgs {
    int n_seg;
    int* sync;
    long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
    long long* prof;           // optional [n_levels][3] timestamps of team 0 / rank 0 (JG_WALK_PROFILE), else nullptr
};

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane: wait until *p >= target; false on timeout / foreign error
__device__ __forceinline__ bool spin_ge(const int* p, int target, int* err, long long ticks) {
    const long long t0 = wall_clock64();
    for (unsigned it = 0;; ++it) {
        if (ld_agent(p) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((it & 63) == 63) {
            if (ld_agent(err) != 0) return false;
            if (wall_clock64() - t0 > ticks) { atomicExch(err, 2); atomicExch(err + (SYNC_WORDS - SYNC_ERR), 2); return false; }   // + the sticky word
        }
    }
}

struct Team { int xcc, size, rank, index, nteams; };

// registration: every workgroup announces its XCD, waits for the whole grid, then reads the team census
__device__ __forceinline__ bool team_join(const WalkArgs& w, int* sh, Team& t) {
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;      // HW_REG_XCC_ID[3:0]
        const int rank = atomicAdd(w.sync + SYNC_TEAM + xcc, 1);
        atomicAdd(w.sync + SYNC_REG, 1);
        int ok = spin_ge(w.sync + SYNC_REG, (int)gridDim.x, w.sync + SYNC_ERR, w.timeout_ticks) ? 1 : 0;
        int nteams = 0, index = 0, size = 0;
        for (int x = 0; x < 16; ++x) {
            const int sz = ld_agent(w.sync + SYNC_TEAM + x);
            if (x == xcc) { index = nteams; size = sz; }
            nteams += sz > 0;
        }
        sh[0] = ok; sh[1] = xcc; sh[2] = size; sh[3] = rank; sh[4] = index; sh[5] = nteams;
    }
    __syncthreads();
    t.xcc = sh[1]; t.size = sh[2]; t.rank = sh[3]; t.index = sh[4]; t.nteams = sh[5];
    const bool ok = sh[0] != 0;
    __syncthreads();
    return ok;
}

// team-local level barrier: arrivals are counted monotonically, barrier number q completes at size * (q + 1)
__device__ __forceinline__ bool team_barrier(const WalkArgs& w, int* sh, const Team& t, int q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have reached the XCD's L2
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        int* bar = w.sync + SYNC_BAR + 32 * t.xcc;
        atomicAdd(bar, 1);
        sh[0] = spin_ge(bar, t.size * (q + 1), w.sync + SYNC_ERR, w.timeout_ticks) ? 1 : 0;
    }
    __syncthreads();
    const bool ok = sh[0] != 0;
    __syncthreads();
    return ok;
}

// position of a workgroup in its team's work list: (segment s, unit u); units of a segment = chunks x team groups
struct Cursor { int s, u; };

template <bool BWD, class Args>
__device__ __forceinline__ void walk_body(Args& a, const WalkArgs& w, double* red, int* sh, const Segment* segs /* LDS copy */) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    Team t;
    if (!team_join(w, sh, t)) return;
    // groups of this team: slots index, index + nteams, ... < gact
    const int gact = a.sel.list ? uniform(*a.sel.count) : a.ld / 64;
    const int ng = gact > t.index ? (gact - t.index + t.nteams - 1) / t.nteams : 0;
    if (ng == 0) return;                                       // a team without groups shares no data and no barrier
    const bool prof = w.prof && t.index == 0 && t.rank == 0 && wave == 0 && lane == 0;
    auto units = [&](int s) { return uniform(segs[s].nchunks) * ng; };
    auto normalise = [&](Cursor& c) {                          // first segment at or after c.s that holds a unit for this rank
        while (c.s < w.n_seg && c.u >= units(c.s)) { ++c.s; c.u = t.rank; }
    };
    auto rec_index = [&](const Cursor& c) {
        const int chunk = c.u / ng;
        if (uniform(segs[c.s].wpi) == 0) return (size_t)uniform(segs[c.s].rec_base) + (size_t)chunk;     // chain task: one record
        return (size_t)uniform(segs[c.s].rec_base) + ((size_t)chunk * 16 + wave) * uniform(segs[c.s].rpw);
    };
    // cursor of the unit this workgroup executes next, and the (prefetched) first record of this wave in it
    Cursor cur{0, t.rank};
    normalise(cur);
    RecS nxt{};
    if (cur.s < w.n_seg) nxt = load_rec(a.rec, rec_index(cur));
    int barriers = 0;
    for (int s = 0; s < w.n_seg; ++s) {
        if (prof && (s == 0 || segs[s - 1].last)) w.prof[3 * (segs[s].level - 1)] = wall_clock64();
        while (cur.s == s) {
            const RecS r = nxt;
            const size_t ri = rec_index(cur);
            const int gs = cur.u % ng;
            const int wpi = uniform(segs[s].wpi), rpw = uniform(segs[s].rpw);
            Cursor nx{cur.s, cur.u + t.size};
            normalise(nx);
            if (nx.s < w.n_seg) nxt = load_rec(a.rec, rec_index(nx));      // prefetch: static tables, safe across barriers
            const int slot = t.index + gs * t.nteams;
            const int g = a.sel.list ? uniform(a.sel.list[slot]) : slot;
            if (!(a.sel.flags && !a.sel.flags[g])) {
                const size_t b = (size_t)min(g * 64 + lane, a.lanes - 1);
                if constexpr (BWD) {
                    if (wpi == 0) bwd_chain_task(a, red, r, wave, lane, b, ld);
                    else bwd_chunk(a, red, r, ri, rpw, wpi, wave, lane, b, ld);
                } else fact_chunk(a, red, r, ri, rpw, wpi, wave, lane, b, ld);
            }
            cur = nx;
        }
        if (segs[s].last) {
            if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); w.prof[3 * (segs[s].level - 1) + 1] = wall_clock64(); }
            if (s + 1 < w.n_seg && !team_barrier(w, sh, t, barriers++)) return;
            if (prof) w.prof[3 * (segs[s].level - 1) + 2] = wall_clock64();
        }
    }
}

constexpr int WALK_MAX_SEG = 1024;      // segment table staged in LDS (32 bytes each)

__device__ __forceinline__ void stage_segments(const Segment* g, Segment* l, int n) {
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int4* src = (const int4*)g;
    int4* dst = (int4*)l;
    for (int i = tid; i < 2 * n; i += 1024) dst[i] = src[i];
    __syncthreads();
}

constexpr int WALK_RED = CHAIN_LDS_D2 * 2;      // doubles: covers the factor partial sums (16*256) and the chain staging

__global__ __launch_bounds__(1024) void k_fact_walk(FactArgs a, WalkArgs w) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // WALK_RED doubles | 16 ints | segments
    int* sh = (int*)(red + WALK_RED);
    Segment* segs = (Segment*)(sh + 16);
    stage_segments(a.seg, segs, w.n_seg);
    walk_body<false>(a, w, red, sh, segs);
}

__global__ __launch_bounds__(1024) void k_bwd_walk(BwdArgs a, WalkArgs w) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // WALK_RED doubles | 16 ints | segments
    int* sh = (int*)(red + WALK_RED);
    Segment* segs = (Segment*)(sh + 16);
    stage_segments(a.seg, segs, w.n_seg);
    walk_body<true>(a, w, red, sh, segs);
}

size_t walk_lds(int n_seg) { return WALK_RED * sizeof(double) + 64 + (size_t)n_seg * sizeof(Segment); }

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

std::mutex g_walk_mu;
hipEvent_t g_walk_ev[64] = {};

}  // namespace

std::mutex& capture_mutex() {
    static std::mutex m;
    return m;
}

int Engine::create(int n, const int* rowptr, const int* col, int ld_, int policy, hipStream_t st) {
    if (ld_ <= 0 || ld_ % 64) { error = "batch leading dimension must be a positive multiple of 64"; return 1; }
    if (analyze(n, rowptr, col, policy, S)) { error = "block pattern must be structurally symmetric with a full diagonal"; return 1; }
    ld = ld_;
    level_launches(S.fact_seg, fact);
    level_launches(S.bwd_seg, bwd);
    level_launches(S.fwd_seg, fwd);
    if (upload(&fact_rec, S.fact_rec, error, st) || upload(&bwd_rec, S.bwd_rec, error, st) || upload(&fact_seg, S.fact_seg, error, st) ||
        upload(&bwd_seg, S.bwd_seg, error, st) || upload(&bwd_chain, S.bwd_chain, error, st) ||
        upload(&fwd_rec, S.fwd_rec, error, st) || upload(&fwd_seg, S.fwd_seg, error, st))
        return 2;
    JG_HIP(hipFuncSetAttribute((const void*)k_bwd_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CHAIN_LDS_D2 * sizeof(double2))));
    JG_HIP(hipMalloc((void**)&X, factor_bytes()));
    JG_HIP(sync_fill(X, 0, factor_bytes(), st));
    JG_HIP(hipMalloc((void**)&W, (size_t)n * 2 * ld * sizeof(double)));
    JG_HIP(sync_fill(W, 0, (size_t)n * 2 * ld * sizeof(double), st));
    JG_HIP(hipMalloc((void**)&status, (size_t)ld * sizeof(int)));
    JG_HIP(sync_fill(status, 0, (size_t)ld * sizeof(int), st));
    JG_HIP(hipMalloc((void**)&sync, (SYNC_WORDS + 1) * sizeof(int)));
    JG_HIP(sync_fill(sync, 0, (SYNC_WORDS + 1) * sizeof(int), st));
    JG_HIP(hipGetDevice(&device));
    int cus = 0;
    JG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    walk_grid = cus;                                   // one 16-wave workgroup per CU: all co-resident by construction
    if (getenv("JG_WALK_PROFILE")) {
        JG_HIP(hipMalloc((void**)&prof, (size_t)S.n_fact_levels * 3 * sizeof(long long)));
        JG_HIP(sync_fill(prof, 0, (size_t)S.n_fact_levels * 3 * sizeof(long long), st));
    }
    // The walker is OPT-IN (JG_WALKER=1).  Measured on MI355X (ACTIVSg10k): its barrier costs 0.6-0.8 us against ~2 us
    // for a kernel boundary, but binding a scenario group to ONE XCD caps the group at 32 CUs x 64 B/clk of L1 fill,
    // and a level's cost is set by exactly that (one item's update list = up to 540 KB through one CU): 3.8 ms per
    // factorisation against 1.5 ms (64 scenarios) / 4.3 against 2.7 ms (512) for the per-level launches, which spread
    // every level over all 256 CUs.
    const char* env = getenv("JG_WALKER");
    walker = false;
    const bool fits = (int)S.fact_seg.size() <= WALK_MAX_SEG && (int)S.bwd_seg.size() <= WALK_MAX_SEG;
    if (env && env[0] == '1' && walk_grid > 0 && fits) {
        JG_HIP(hipFuncSetAttribute((const void*)k_fact_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds(WALK_MAX_SEG)));
        JG_HIP(hipFuncSetAttribute((const void*)k_bwd_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds(WALK_MAX_SEG)));
        // census: can every workgroup of a walk become resident and do the XCD teams form?  (a walk with no segments)
        FactArgs a{};
        a.seg = fact_seg; a.ld = ld; a.lanes = ld;
        WalkArgs w{0, sync, 5000000LL /* 50 ms */, nullptr};
        hipLaunchKernelGGL(k_fact_walk, dim3(walk_grid), dim3(64, 16), walk_lds(0), st, a, w);
        JG_HIP(hipStreamSynchronize(st));
        int h[SYNC_TEAM + 16];
        JG_HIP(sync_copy(h, sync, sizeof(h), hipMemcpyDeviceToHost, st));
        int members = 0;
        for (int x = 0; x < 16; ++x) members += h[SYNC_TEAM + x];
        walker = h[SYNC_ERR] == 0 && h[SYNC_REG] == walk_grid && members == walk_grid;
        JG_HIP(sync_fill(sync, 0, (SYNC_WORDS + 1) * sizeof(int), st));
    }
    return 0;
}

void Engine::destroy() {
    if (prof) {
        const int nl = S.n_fact_levels;
        std::vector<long long> t((size_t)nl * 3);
        if (hipMemcpy(t.data(), prof, t.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess && nl > 0 && t[0]) {
            fprintf(stderr, "[jg walk profile] level | work_us barrier_us\n");
            for (int l = 0; l < nl; ++l)
                fprintf(stderr, "[jg walk profile] %3d | %7.2f %7.2f\n", l, (t[3 * l + 1] - t[3 * l]) * 0.01,
                        l + 1 < nl ? (t[3 * l + 2] - t[3 * l + 1]) * 0.01 : 0.0);
            fprintf(stderr, "[jg walk profile] total %.2f us\n", (t[3 * (size_t)nl - 2] - t[0]) * 0.01);
        }
        hipFree(prof); prof = nullptr;
    }
    hipFree(fact_rec); hipFree(bwd_rec); hipFree(fact_seg); hipFree(bwd_seg); hipFree(bwd_chain); hipFree(sync);
    hipFree(fwd_rec); hipFree(fwd_seg); fwd_rec = nullptr; fwd_seg = nullptr;
    hipFree(sel_rec); hipFree(sel_seg); hipFree(Zs); sel_rec = nullptr; sel_seg = nullptr; Zs = nullptr;
    bwd_chain = nullptr;
    hipFree(X); hipFree(W); hipFree(status);
    fact_rec = bwd_rec = nullptr; fact_seg = bwd_seg = nullptr; sync = nullptr; status = nullptr;
    X = W = nullptr;
}

// Walks of different handles form ONE chain per process: wait for the previous walk's event, enqueue, record the
// event again -- atomically with respect to other host threads (the mutex is held from begin to end; use WalkTurn).
void Engine::serialize_begin(hipStream_t st) {
    if (!walker || device < 0 || device >= 64) return;
    g_walk_mu.lock();
    if (g_walk_ev[device]) hipStreamWaitEvent(st, g_walk_ev[device], 0);
}

void Engine::serialize_end(hipStream_t st) {
    if (!walker || device < 0 || device >= 64) return;
    if (g_walk_ev[device] || hipEventCreateWithFlags(&g_walk_ev[device], hipEventDisableTiming) == hipSuccess) hipEventRecord(g_walk_ev[device], st);
    else g_walk_ev[device] = nullptr;
    g_walk_mu.unlock();
}

int Engine::walk_status(hipStream_t st) {
    if (!walker) return 0;
    int e = 0;
    JG_HIP(hipMemcpyAsync(&e, sync + SYNC_WORDS, sizeof(int), hipMemcpyDeviceToHost, st));
    JG_HIP(hipStreamSynchronize(st));
    if (e) { error = "persistent level walk stalled (a workgroup never became resident: is another process using this GPU?); set JG_WALKER=0"; return 2; }
    return 0;
}

int Engine::factor(hipStream_t st, const double* A, const double* rhs, const GroupSel& sel, int mode) {
    if (!S.inplace && !A) { error = "factor: no source matrix"; return 1; }
    FactArgs a{fact_rec, fact_seg, S.inplace ? X : A, rhs, X, W, status, sel, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    if (walker && mode != 1) {
        WalkArgs w{(int)S.fact_seg.size(), sync, 100000000LL /* 1 s */, prof};
        JG_HIP(hipMemsetAsync(sync, 0, SYNC_WORDS * sizeof(int), st));
        hipLaunchKernelGGL(k_fact_walk, dim3(walk_grid), dim3(64, 16), walk_lds(w.n_seg), st, a, w);
        JG_HIP(hipGetLastError());
        return 0;
    }
    const int gs = group_stride(ld / 64);
    for (const DevLaunch& L : fact) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = S.fact_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL(k_fact_level, dim3((unsigned)L.grid * (16 / FACT_WAVES) * gs, L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 256 * sizeof(double), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

int Engine::selected_inverse(hipStream_t st, const GroupSel& sel) {
    if (!Zs) {                                          // tables and storage on first use
        build_selected_inverse(S);
        level_launches(S.sel_seg, selv);
        if (upload(&sel_rec, S.sel_rec, error, st) || upload(&sel_seg, S.sel_seg, error, st)) return 2;
        JG_HIP(hipMalloc((void**)&Zs, factor_bytes()));
        JG_HIP(sync_fill(Zs, 0, factor_bytes(), st));
    }
    SelArgs a{sel_rec, sel_seg, X, Zs, sel, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    const int gs = group_stride(ld / 64);
    for (const DevLaunch& L : selv) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = S.sel_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL(k_sel_level, dim3((unsigned)L.grid * (16 / FACT_WAVES) * gs, L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 128 * sizeof(double2), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

int Engine::forward(hipStream_t st, const double* rhs, const GroupSel& sel) {
    FactArgs a{fwd_rec, fwd_seg, X, rhs, X, W, status, sel, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    const int gs = group_stride(ld / 64);
    for (const DevLaunch& L : fwd) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = S.fwd_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL(k_fact_level, dim3((unsigned)L.grid * (16 / FACT_WAVES) * gs, L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 256 * sizeof(double), st, a);
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
    if (!S.inplace) { error = "set_shared_matrix needs an in-place engine"; return 1; }
    const int nnz = (int)S.src_entry.size();
    double* dblk = nullptr; int* dmap = nullptr;
    std::vector<double> hb(blocks_host, blocks_host + (size_t)nnz * 4);
    if (upload(&dblk, hb, error, st) || upload(&dmap, S.src_entry, error, st)) { hipFree(dblk); hipFree(dmap); return 2; }
    hipLaunchKernelGGL(k_fill_shared, dim3((nnz + 3) / 4, ld / 64), dim3(64, 4), 0, st, dblk, dmap, X, nnz, ld);
    hipError_t e = hipStreamSynchronize(st);
    hipFree(dblk); hipFree(dmap);
    JG_HIP(e);
    return 0;
}

int Engine::backsolve(hipStream_t st, double* out, const StateUpdate& upd, const GroupSel& sel, int mode) {
    BwdArgs a{bwd_rec, bwd_seg, bwd_chain, X, W, out, sel, upd, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    if (walker && mode != 1) {
        WalkArgs w{(int)S.bwd_seg.size(), sync, 100000000LL /* 1 s */, nullptr};
        JG_HIP(hipMemsetAsync(sync, 0, SYNC_WORDS * sizeof(int), st));
        hipLaunchKernelGGL(k_bwd_walk, dim3(walk_grid), dim3(64, 16), walk_lds(w.n_seg), st, a, w);
        JG_HIP(hipGetLastError());
        return 0;
    }
    const int gs = group_stride(ld / 64);
    for (const DevLaunch& L : bwd) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = S.bwd_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        if (!L.chain && L.wpi_max <= 8)                         // 0.387 -> 0.381 ms at 512 scenarios
            hipLaunchKernelGGL(k_bwd_level8, dim3((unsigned)L.grid * 2 * gs, L.nseg), dim3(64, 8), 8 * 128 * sizeof(double), st, a);
        else
        hipLaunchKernelGGL(k_bwd_level, dim3((unsigned)L.grid * gs, L.nseg), dim3(64, 16),
                           L.chain ? (size_t)CHAIN_LDS_D2 * sizeof(double2) : 16 * 128 * sizeof(double), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

}  // namespace jg
