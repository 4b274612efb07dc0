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

// ---- multifrontal top (jg_symbolic.hpp): one workgroup per (task, scenario), dense front in LDS -------------------------
// The pivots above BlockSymbolic::top_level are not level items: a TASK owns a chain of consecutive pivots k0 .. k0+m-1
// (parent(k) = k + 1) and its front of f = m + e block rows / columns (e = |struct(last pivot)|).  Lanes run across the front
// of ONE scenario (the level kernels run 64 scenarios across the lanes): a pivot step is LDS reads + 2x2 block FMAs + one
// workgroup barrier, not a kernel boundary with cold caches.
//   load:   owned entries (rows / columns of the chain) from the batch-minor factor storage -- they hold the assembled value
//           plus the terms of bottom pivots (level items) --, the rhs rows, the children's update matrices from the stack;
//   steps:  for pivot q: F(i,j) -= F(i,q) D(q)^-1 F(q,j) over struct(q)^2 (+ the rhs column), the lane that finishes
//           D(q+1) factorises it in place (2x2 LU with partial pivoting, as fact_finish);
//   store:  U, unscaled Lh, factored D, y back to the batch-minor storage (the backward sweep, the forward-only sweep and the
//           selected inverse read them there), the e x e update matrix + update vector to the scenario-major stack.
// Same arithmetic per term as term3(), terms of an entry in ascending pivot order => bitwise run-to-run determinism.
struct TopArgs {
    const Rec* task; const int* data;
    double* X; double* W; double* stack; int* status; GroupSel sel;
    long long stack_stride;        // doubles per scenario
    int ld, lanes, task_begin, ntasks, lpg;   // lpg: scenarios per 64-lane group that get a workgroup (64, or the real count of a single small group)
};

__device__ __forceinline__ Blk lds_blk(const double* F, int r, int c, int fp) {
    const double2* p = (const double2*)(F + ((size_t)r * fp + c) * 4);
    const double2 r0 = p[0], r1 = p[1];
    return Blk{r0.x, r0.y, r1.x, r1.y};
}
__device__ __forceinline__ void lds_put(double* F, int r, int c, int fp, const Blk& v) {
    double2* p = (double2*)(F + ((size_t)r * fp + c) * 4);
    p[0] = double2{v.v00, v.v01}; p[1] = double2{v.v10, v.v11};
}
// 2x2 LU with in-block partial pivoting, stored form of fact_finish
__device__ __forceinline__ Blk factor_diag(const Blk& c, int* status) {
    const bool sw = fabs(c.v10) > fabs(c.v00);
    const double u11 = sw ? c.v10 : c.v00, u12 = sw ? c.v11 : c.v01;
    const double o21 = sw ? c.v00 : c.v10, o22 = sw ? c.v01 : c.v11;
    const double iu11 = 1.0 / u11;
    const double l = o21 * iu11;
    const double u22 = o22 - l * u12;
    const double iu22 = 1.0 / u22;
    if (!(fabs(u11) > 0.0) || !(fabs(u22) > 0.0) || !(fabs(iu11) < 1.0e300) || !(fabs(iu22) < 1.0e300)) atomicOr(status, 4);
    return Blk{iu11, u12, sw ? l + 4.0 : l, iu22};
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_fact_top(TopArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NT = 64 * NW;
    int grp, x;
    if (!map_block(a.sel, a.ld, a.ntasks * a.lpg, grp, x)) return;
    const int ti = x / a.lpg;
    const int bb = grp * 64 + (x - ti * a.lpg);
    if (bb >= a.lanes) return;                                   // padding lanes of the last group: no scenario, no work
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uniform(tid >> 6);
    const RecS h = load_rec(a.task, (size_t)a.task_begin + ti);
    const int m = h[0], e = h[1], k0 = h[2], nload = h[5], nchild = h[6], fp = h[7], nlist = h[8];
    const int* td = a.data + h[3];
    const int f = m + e;
    double* F = lds;                                             // [f][fp] blocks of 4 doubles
    double2* Y = (double2*)(F + (size_t)f * fp * 4);             // [f] rhs column
    int* L = (int*)(Y + f);                                      // step table + struct lists
    const size_t b = (size_t)bb, ld = (size_t)a.ld;
    double* stk = a.stack + b * (size_t)a.stack_stride;

    // ---- load
    for (int i = tid; i < nlist; i += NT) L[i] = td[i];
    for (int r = m + wave; r < f; r += NW)
        if (lane < e) lds_put(F, r, m + lane, fp, Blk{0.0, 0.0, 0.0, 0.0});
    if (tid < e) Y[m + tid] = double2{0.0, 0.0};
    const int2* ll = (const int2*)(td + h[9]);
    for (int i0 = tid; i0 < nload; i0 += 4 * NT) {              // four gathers in flight per lane
        int2 d[4]; Blk v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * NT; d[u] = ll[i < nload ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = Blk{0.0, 0.0, 0.0, 0.0};
            if (i0 + u * NT < nload && !((unsigned)d[u].x >> 28 & 1)) v[u] = load_blk(a.X, (size_t)(d[u].x & 0x0fffffff), b, ld);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * NT < nload) {
                const int r = d[u].y >> 8, c = d[u].y & 255;
                lds_put(F, r, c, fp, v[u]);
                if ((unsigned)d[u].x >> 28 & 2) lds_put(F, c, r, fp, Blk{v[u].v00, v[u].v10, v[u].v01, v[u].v11});   // symmetric plans: Lh(c,r) = U(r,c)'
            }
        }
    }
    for (int q = tid; q < m; q += NT) Y[q] = load_vec(a.W, (size_t)(k0 + q), b, ld);
    __syncthreads();
    // ---- extend-add of the children's update matrices (two children may hit the same block: one after the other)
    const int* cd = td + h[10];
    for (int ch = 0; ch < nchild; ++ch) {
        const int coff = uniform(cd[0]), ce = uniform(cd[1]);
        const int* cmap = cd + 2;
        const double* C = stk + coff;
        const int cl = lane < ce ? cmap[lane] : 0;
        for (int ca = wave; ca < ce; ca += 2 * NW) {
            const int ca2 = ca + NW;
            const int r0 = uniform(cmap[ca]), r1 = uniform(cmap[ca2 < ce ? ca2 : ca]);
            double2 s0{0.0, 0.0}, s1{0.0, 0.0}, s2{0.0, 0.0}, s3{0.0, 0.0};
            if (lane < ce) {
                const double2* p = (const double2*)(C + ((size_t)ca * ce + lane) * 4);
                s0 = p[0]; s1 = p[1];
                if (ca2 < ce) { const double2* p2 = (const double2*)(C + ((size_t)ca2 * ce + lane) * 4); s2 = p2[0]; s3 = p2[1]; }
                Blk t = lds_blk(F, r0, cl, fp);
                lds_put(F, r0, cl, fp, Blk{t.v00 + s0.x, t.v01 + s0.y, t.v10 + s1.x, t.v11 + s1.y});
                if (ca2 < ce) {
                    t = lds_blk(F, r1, cl, fp);
                    lds_put(F, r1, cl, fp, Blk{t.v00 + s2.x, t.v01 + s2.y, t.v10 + s3.x, t.v11 + s3.y});
                }
            }
        }
        if (tid < ce) {
            const double2 v = ((const double2*)(C + (size_t)ce * ce * 4))[tid];
            double2 y = Y[cmap[tid]];
            y.x += v.x; y.y += v.y;
            Y[cmap[tid]] = y;
        }
        __syncthreads();
        cd += 2 + ce;
    }
    if (tid == 0) lds_put(F, 0, 0, fp, factor_diag(lds_blk(F, 0, 0, fp), a.status + b));
    __syncthreads();
    // ---- pivot steps
    const unsigned char* lists = (const unsigned char*)(L + 3 * m);
    for (int q = 0; q < m; ++q) {
        const int s = uniform(L[3 * q]), lo = uniform(L[3 * q + 1]), lg = uniform(L[3 * q + 2]);
        const unsigned char* lst = lists + lo;
        const int bcol = lane & ((1 << lg) - 1), ar = lane >> lg, rpp = 64 >> lg;
        const Blk D = lds_blk(F, q, q, fp);
        double z00 = 0.0, z10 = 0.0, z01 = 0.0, z11 = 0.0;
        int ib = 0;
        if (bcol < s) {                                          // column operand D^-1 U(q, ib), once per lane and step
            ib = lst[bcol];
            const Blk U = lds_blk(F, q, ib, fp);
            dsolve(D, U.v00, U.v10, z00, z10);
            dsolve(D, U.v01, U.v11, z01, z11);
        } else if (bcol == s) {                                  // the rhs column: D^-1 y_q
            const double2 y = Y[q];
            dsolve(D, y.x, y.y, z00, z10);
        }
        for (int a0 = wave * rpp + ar; a0 < s; a0 += NW * rpp) {
            const int ia = lst[a0];
            const Blk Lb = lds_blk(F, ia, q, fp);
            if (bcol < s) {
                Blk t = lds_blk(F, ia, ib, fp);
                t.v00 -= Lb.v00 * z00 + Lb.v01 * z10;
                t.v01 -= Lb.v00 * z01 + Lb.v01 * z11;
                t.v10 -= Lb.v10 * z00 + Lb.v11 * z10;
                t.v11 -= Lb.v10 * z01 + Lb.v11 * z11;
                if (a0 == 0 && bcol == 0 && q + 1 < m) t = factor_diag(t, a.status + b);     // D(q+1) is final: the next pivot
                lds_put(F, ia, ib, fp, t);
            } else if (bcol == s) {
                double2 y = Y[ia];
                y.x -= Lb.v00 * z00 + Lb.v01 * z10;
                y.y -= Lb.v10 * z00 + Lb.v11 * z10;
                Y[ia] = y;
            }
        }
        __syncthreads();
    }
    // ---- store
    for (int i = tid; i < nload; i += NT) {
        const int2 d = ll[i];
        const Blk v = lds_blk(F, d.y >> 8, d.y & 255, fp);
        store_blk(a.X, (size_t)(d.x & 0x0fffffff), b, ld, v.v00, v.v01, v.v10, v.v11);
    }
    for (int q = tid; q < m; q += NT) { const double2 y = Y[q]; store_vec(a.W, (size_t)(k0 + q), b, ld, y.x, y.y); }
    if (e > 0) {
        double* out = stk + h[4];
        for (int ca = wave; ca < e; ca += NW)
            if (lane < e) {
                const Blk v = lds_blk(F, m + ca, m + lane, fp);
                double2* p = (double2*)(out + ((size_t)ca * e + lane) * 4);
                p[0] = double2{v.v00, v.v01}; p[1] = double2{v.v10, v.v11};
            }
        if (tid < e) ((double2*)(out + (size_t)e * e * 4))[tid] = Y[m + tid];
    }
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
    if (!S.top_launch.empty()) {
        if (upload(&top_task, S.top_task, error, st) || upload(&top_data, S.top_data, error, st)) return 2;
        int lds1 = 0, lds4 = 0;
        for (const TopLaunch& L : S.top_launch) (L.waves == 1 ? lds1 : lds4) = std::max(L.waves == 1 ? lds1 : lds4, L.lds_bytes);
        if (lds1) JG_HIP(hipFuncSetAttribute((const void*)k_fact_top<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
        if (lds4) JG_HIP(hipFuncSetAttribute((const void*)k_fact_top<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
        const size_t sb = (size_t)std::max<long long>(S.top_stack, 2) * ld * sizeof(double);
        JG_HIP(hipMalloc((void**)&top_stack, sb));
        JG_HIP(sync_fill(top_stack, 0, sb, st));
    }
    JG_HIP(hipMalloc((void**)&X, factor_bytes()));
    JG_HIP(sync_fill(X, 0, factor_bytes(), st));
    JG_HIP(hipMalloc((void**)&W, (size_t)n * 2 * ld * sizeof(double)));
    JG_HIP(sync_fill(W, 0, (size_t)n * 2 * ld * sizeof(double), st));
    JG_HIP(hipMalloc((void**)&status, (size_t)ld * sizeof(int)));
    JG_HIP(sync_fill(status, 0, (size_t)ld * sizeof(int), st));
    JG_HIP(hipGetDevice(&device));
    return 0;
}

void Engine::destroy() {
    hipFree(fact_rec); hipFree(bwd_rec); hipFree(fact_seg); hipFree(bwd_seg); hipFree(bwd_chain);
    hipFree(fwd_rec); hipFree(fwd_seg); fwd_rec = nullptr; fwd_seg = nullptr;
    hipFree(sel_rec); hipFree(sel_seg); hipFree(Zs); sel_rec = nullptr; sel_seg = nullptr; Zs = nullptr;
    hipFree(top_task); hipFree(top_data); hipFree(top_stack); top_task = nullptr; top_data = nullptr; top_stack = nullptr;
    bwd_chain = nullptr;
    hipFree(X); hipFree(W); hipFree(status);
    fact_rec = bwd_rec = nullptr; fact_seg = bwd_seg = nullptr; status = nullptr;
    X = W = nullptr;
}

int Engine::factor(hipStream_t st, const double* A, const double* rhs, const GroupSel& sel) {
    if (!S.inplace && !A) { error = "factor: no source matrix"; return 1; }
    FactArgs a{fact_rec, fact_seg, S.inplace ? X : A, rhs, X, W, status, sel, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
    const int gs = group_stride(ld / 64);
    for (const DevLaunch& L : fact) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = S.fact_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL(k_fact_level, dim3((unsigned)L.grid * (16 / FACT_WAVES) * gs, L.nseg), dim3(64, FACT_WAVES), FACT_WAVES * 256 * sizeof(double), st, a);
    }
    // the top of the elimination tree: multifrontal tasks, one workgroup per (task, scenario), launch = (task level, class)
    if (!S.top_launch.empty()) {
        TopArgs t{top_task, top_data, X, W, top_stack, status, sel, std::max<long long>(S.top_stack, 2), ld, a.lanes, 0, 0, 64};
        if (ld == 64 && t.lanes < 64) t.lpg = t.lanes;
        for (const TopLaunch& L : S.top_launch) {
            t.task_begin = L.task_begin; t.ntasks = L.ntasks;
            const dim3 grid((unsigned)L.ntasks * t.lpg * gs);
            if (L.waves == 1) hipLaunchKernelGGL(k_fact_top<1>, grid, dim3(64), (size_t)L.lds_bytes, st, t);
            else hipLaunchKernelGGL(k_fact_top<4>, grid, dim3(256), (size_t)L.lds_bytes, st, t);
        }
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

int Engine::backsolve(hipStream_t st, double* out, const StateUpdate& upd, const GroupSel& sel) {
    BwdArgs a{bwd_rec, bwd_seg, bwd_chain, X, W, out, sel, upd, ld, 0, lanes > 0 ? lanes : ld, 0, 0, 1, 1};
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
