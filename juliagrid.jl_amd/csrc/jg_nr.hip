// jg_nr.hip -- Newton-Raphson AC power flow on MI355X: C ABI (include/jgrid.h) + fused
// mismatch/Jacobian assembly kernel + per-scenario convergence control.
//
// Reference behaviour restated (paths relative to /root/reference):
//   newtonJacobian      src/powerFlow/acPowerFlow.jl:89-175   (index maps, CSC pattern; bit-exact)
//   mismatch!           src/powerFlow/acPowerFlow.jl:645-685  + src/backend/equations.jl:63-103,126-128
//   solve! (fill)       src/powerFlow/acPowerFlow.jl:820-888  + src/backend/equations.jl:105-144
//   solve! (linear)     src/powerFlow/acPowerFlow.jl:890-906  -> jg_engine (device LU / solves / update)
//   powerFlow!          src/powerFlow/acPowerFlow.jl:1389-1433
//
// Design (not a translation): the reference walks Ybus three times per iteration with one sincos
// each (mismatch, off-diagonal partials, diagonal re-sum).  Here ONE pass over bus row i with one
// sincos per stored (i,j) produces f_P, f_Q and the whole 2x2-block Jacobian row; PV / slack rows
// and columns are padded with identity so the Jacobian is an n x n matrix of 2x2 blocks with the
// Ybus pattern -- the layout the block LU engine consumes directly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/jgrid.h"
#include "jg_engine.hpp"
#include "jg_comp.hpp"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define NR_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t err__ = (expr);                                                     \
        if (err__ != hipSuccess) return fail(2, std::string(#expr) + ": " + hipGetErrorString(err__)); \
    } while (0)

constexpr int ASM_ROWS = 16;   // bus rows per workgroup
constexpr int ASM_WAVES = 4;
constexpr int CH = 4;          // Ybus entries whose gathers are in flight together

struct AsmArgs {
    const int* rowptr; const int* colm; const double2* GB; const int* rowtype;   // colm = col | (mask << 24); rowtype: int per bus (scalar-loadable)
    const int* dst;            // Ybus CSR position -> entry of the LU factor storage: the Jacobian is assembled IN PLACE
    const double* vm; const double* va; const double* p; const double* q;
    const int* ppos; const double* pdg; const double* pdb;
    double* A; double* F; double* part; double* pq_out; jg::GroupSel sel;   // pq_out (nullable): [n][ld][2] calculated injections P_i, Q_i
    double* R; int fd_mode;    // fast decoupled passes (acPowerFlow.jl:687-730, 952-962): 1 = mismatches / V, R = (f_P, 0); 2 = R = (0, f_Q)
    int n; int ld; int mp; int nchunk; int lanes;
    const int* only_if;        // nullable: the launch does nothing unless *only_if != 0 (re-assembly after a lane compaction)
    double* W; int* lu_status; // (lu_status: unused by the kernel -- an atomic there costs 14 VGPRs and a wave per SIMD; a singular level-0 block
                               // turns the row's mismatch into NaN instead, which k_check reports as status 3)
                               // level 0 of the engine's prefactor plan (jg_symbolic.hpp): where rowtype carries (pivot + 1) << 2 the diagonal block
                               // leaves FACTORISED and the mismatch row is also written as the rhs row W[pivot]
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void store_vec_nt(double* base, size_t item, size_t b, size_t ld, double v0, double v1) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(d2{v0, v1}, (d2*)(base + (item * ld + b) * 2));
}

// Fused mismatch + Jacobian assembly. blockDim (64, ASM_WAVES); 1-D grid, scenario group fastest (jg::map_block:
// a group's V/theta gathers stay in one XCD's L2).
// JAC = false: mismatch only (no Jacobian stores) -- the cheap pre-pass that decides which scenarios are still active.
// WAVES: waves of a workgroup (which always takes ASM_ROWS bus rows).  4: a wave walks four rows one after the other -- the throughput form; 16 (round 6, a handful of
// scenarios): one row per wave -- a row is three dependent round trips (row header, Ybus entries, V / theta gathers), and with nothing else on the chip to hide them
// four rows in sequence made the pass of a single instance 23 us long.
template <int MP, bool JAC, int WAVES = ASM_WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_assemble(AsmArgs a) {
    __shared__ double red[2][WAVES][64];
    int grp, bx;
    if (a.only_if && !uniform(*a.only_if)) return;
    if (!jg::map_block(a.sel, a.ld, a.nchunk, grp, bx)) return;   // every scenario of a skipped 64-lane group is finished
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min(grp * 64 + lane, a.lanes - 1);        // lanes beyond the batch alias its last scenario
    int ppos[MP > 0 ? MP : 1];
#pragma unroll
    for (int m = 0; m < MP; ++m) ppos[m] = a.ppos[(size_t)m * ld + b];
    double maxp = 0.0, maxq = 0.0;
    const int r0 = bx * ASM_ROWS;
    const int r1 = min(r0 + ASM_ROWS, a.n);
    for (int i = r0 + wave; i < r1; i += WAVES) {
        const int p0 = uniform(a.rowptr[i]), p1 = uniform(a.rowptr[i + 1]);
        const int tfull = (int)((unsigned)uniform(a.rowtype[i]));   // bus type | (pivot + 1) << 2 where this pass also finishes the plan's level 0
        const int ti = tfull & 3, pre = tfull >> 2;
        const double vi = a.vm[(size_t)i * ld + b];
        const double thi = a.va[(size_t)i * ld + b];
        const double pinj = a.p[(size_t)i * ld + b], qinj = a.q[(size_t)i * ld + b];   // issued early, used after the row
        double s1 = 0.0, s2 = 0.0, gii = 0.0, bii = 0.0;
        int pd = 0;                                        // factor entry of the diagonal block
        // rows are short (3.4 entries on average, <= 19): the cost is the latency of the V/theta gathers,
        // so the loads of four entries are issued together before any of them is consumed
        for (int pc = p0; pc < p1; pc += CH) {
            const int cnt = min(CH, p1 - pc);
            int cm[CH], de[CH]; double2 gb[CH]; double vv[CH], tt[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {                       // scalar loads of the whole chunk first ...
                const int p = pc + (k < cnt ? k : 0);
                cm[k] = uniform(a.colm[p]);
                de[k] = JAC ? uniform(a.dst[p]) : 0;
                gb[k] = a.GB[p];
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {                       // ... then all gathers in flight together
                const size_t j = (size_t)(cm[k] & 0xffffff);
                vv[k] = a.vm[j * ld + b];
                tt[k] = a.va[j * ld + b];
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (k >= cnt) break;
                const int p = pc + k;
                const int j = cm[k] & 0xffffff;
                const int mk = cm[k] >> 24;                    // bit0 dP/dth, bit1 dP/dV, bit2 dQ/dth, bit3 dQ/dV exist
                double g = gb[k].x, bb = gb[k].y;
#pragma unroll
                for (int m = 0; m < MP; ++m)
                    if (ppos[m] == p) { g += a.pdg[(size_t)m * ld + b]; bb += a.pdb[(size_t)m * ld + b]; }
                const double vj = vv[k];
                double s, c;
                sincos(thi - tt[k], &s, &c);
                const double ac = g * c + bb * s;      // G cos + B sin
                const double ad = g * s - bb * c;      // G sin - B cos
                s1 += vj * ac;
                s2 += vj * ad;
                if (j == i) { pd = de[k]; gii = g; bii = bb; continue; }
                if (!JAC) continue;
                // rows: P exists unless slack, Q exists for PQ; cols: theta unless slack, V for PQ (mask from the host)
                // write-once stream (read back by the factorisation only): nontemporal, so the V/theta gathers keep the L2
                jg::store_blk_nt(a.A, (size_t)de[k], b, ld,
                                 (mk & 1) ? vi * vj * ad : 0.0,            // dP_i/dtheta_j   equations.jl:109-111
                                 (mk & 2) ? vi * ac : 0.0,                 // dP_i/dV_j       equations.jl:117-119
                                 (mk & 4) ? -(vi * vj) * ac : 0.0,         // dQ_i/dtheta_j   equations.jl:134-136
                                 (mk & 8) ? vi * ad : 0.0);                // dQ_i/dV_j       equations.jl:142-144
            }
        }
        if (!JAC && a.pq_out) jg::store_vec(a.pq_out, (size_t)i, b, ld, vi * s1, vi * s2);     // PiQi (acAnalysis.jl:891-896)
        double fp = vi * s1 - pinj;                        // acPowerFlow.jl:676
        double fq = vi * s2 - qinj;                        // acPowerFlow.jl:679
        if (!JAC && a.fd_mode) {                           // fast Newton-Raphson: calc - spec / V  (acPowerFlow.jl:720-726, 952-961)
            const double vinv = 1.0 / vi;
            fp = s1 - pinj * vinv;
            fq = s2 - qinj * vinv;
        }
        double d00 = -vi * s2 - bii * (vi * vi);           // equations.jl:105-107, acPowerFlow.jl:872
        double d01 = s1 + gii * vi;                        // equations.jl:113-115
        double d10 = vi * s1 - gii * (vi * vi);            // equations.jl:130-132
        double d11 = s2 - bii * vi;                        // equations.jl:138-140
        if (ti == 3) { d00 = 1.0; d01 = 0.0; d10 = 0.0; d11 = 1.0; fp = 0.0; fq = 0.0; }
        else if (ti == 2) { d01 = 0.0; d10 = 0.0; d11 = 1.0; fq = 0.0; }
        if (JAC) {
            if (pre) {                                     // nobody updates this block: its 2x2 LU is the factorisation's whole work on it
                const jg::Blk raw{d00, d01, d10, d11};
                bool bad;
                const jg::Blk f = jg::diag_lu(raw, jg::row_max(raw), bad);
                if (bad) fp = __builtin_nan("");           // a singular block marks its scenario through the mismatch (k_check: NaN -> status 3)
                d00 = f.v00; d01 = f.v01; d10 = f.v10; d11 = f.v11;
                store_vec_nt(a.W, (size_t)pre - 1, b, ld, fp, fq);   // nontemporal like every other store of this pass: a plain store here
                                                                       // cost 0.09 ms per pass at 512 scenarios (the V / theta gathers lost their L2 lines)
            }
            jg::store_blk_nt(a.A, (size_t)pd, b, ld, d00, d01, d10, d11);
        }
        if (!JAC && a.fd_mode) {
            if (a.fd_mode == 1) { jg::store_vec(a.F, (size_t)i, b, ld, fp, fq); jg::store_vec(a.R, (size_t)i, b, ld, fp, 0.0); }
            else jg::store_vec(a.R, (size_t)i, b, ld, 0.0, fq);
        } else {
            store_vec_nt(a.F, (size_t)i, b, ld, fp, fq);
        }
        // NaN-propagating max: a NaN mismatch must not look converged
        const double afp = fabs(fp), afq = fabs(fq);
        maxp = (afp > maxp || afp != afp) ? afp : maxp;
        maxq = (afq > maxq || afq != afq) ? afq : maxq;
    }
    red[0][wave][lane] = maxp;
    red[1][wave][lane] = maxq;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            const double x = red[0][w][lane], y = red[1][w][lane];
            maxp = (x > maxp || x != x) ? x : maxp;
            maxq = (y > maxq || y != y) ? y : maxq;
        }
        a.part[((size_t)bx * 2) * ld + b] = maxp;
        a.part[((size_t)bx * 2 + 1) * ld + b] = maxq;
    }
}

// The assembly of ONE scenario (round 6): a QUAD of lanes per bus row.  k_assemble gives a wave to a row of 64 scenarios and walks four rows one after the other; with one
// scenario that is four times three dependent round trips per wave (row header, Ybus entries, V / theta gathers) around the work of one lane.  Here lane q of a row's quad
// takes its Ybus entries q, q + 4, ... two at a time (a row of 19 entries is three trips instead of five, and a trip is two sincos per lane instead of four), the row sums
// meet through two shuffles (fixed order), lane 0 finishes the row as k_assemble does (the leaf pivots of a prefactor plan leave factorised, the rhs row with them); the
// chunk maxima of sixteen rows meet through shuffles as well.  Same formulas per entry; the row sums add in another order (V, theta within 1e-10: tests/test_top_variants_gpu.py).
template <int MP, bool JAC>
__global__ __launch_bounds__(256) void k_assemble1(AsmArgs a) {
    int grp, bx;
    if (!jg::map_block(a.sel, a.ld, (a.n + 63) / 64, grp, bx)) return;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)grp * 64;
    const int q = (int)threadIdx.x & 3;
    const int i0 = bx * 64 + ((int)threadIdx.x >> 2);
    const bool live = i0 < a.n;
    const int i = live ? i0 : a.n - 1;
    int ppos[MP > 0 ? MP : 1];
#pragma unroll
    for (int m = 0; m < MP; ++m) ppos[m] = a.ppos[(size_t)m * ld + b];
    const int p0 = a.rowptr[i], p1 = live ? a.rowptr[i + 1] : p0;
    const int tfull = (int)((unsigned)a.rowtype[i]);
    const int ti = tfull & 3, pre = tfull >> 2;
    const double vi = a.vm[(size_t)i * ld + b];
    const double thi = a.va[(size_t)i * ld + b];
    const double pinj = a.p[(size_t)i * ld + b], qinj = a.q[(size_t)i * ld + b];
    double s1 = 0.0, s2 = 0.0, gii = 0.0, bii = 0.0;
    int pd = 0;
    constexpr int C2 = 2;                                         // entries per lane and trip
    for (int pc = p0 + q; pc < p1; pc += 4 * C2) {               // this lane's entries: pc, pc + 4
        int cm[C2], de[C2]; double2 gb[C2]; double vv[C2], tt[C2];
#pragma unroll
        for (int k = 0; k < C2; ++k) {
            const int p = pc + 4 * k < p1 ? pc + 4 * k : pc;
            cm[k] = a.colm[p];
            de[k] = JAC ? a.dst[p] : 0;
            gb[k] = a.GB[p];
        }
#pragma unroll
        for (int k = 0; k < C2; ++k) {
            const size_t j = (size_t)(cm[k] & 0xffffff);
            vv[k] = a.vm[j * ld + b];
            tt[k] = a.va[j * ld + b];
        }
#pragma unroll
        for (int k = 0; k < C2; ++k) {
            const int p = pc + 4 * k;
            if (p < p1) {
                const int j = cm[k] & 0xffffff;
                const int mk = cm[k] >> 24;
                double g = gb[k].x, bb = gb[k].y;
#pragma unroll
                for (int m = 0; m < MP; ++m)
                    if (ppos[m] == p) { g += a.pdg[(size_t)m * ld + b]; bb += a.pdb[(size_t)m * ld + b]; }
                const double vj = vv[k];
                double s, c;
                sincos(thi - tt[k], &s, &c);
                const double ac = g * c + bb * s;
                const double ad = g * s - bb * c;
                s1 += vj * ac;
                s2 += vj * ad;
                if (j == i) { pd = de[k]; gii = g; bii = bb; }
                else if (JAC) {
                    jg::store_blk(a.A, (size_t)de[k], b, ld,
                                  (mk & 1) ? vi * vj * ad : 0.0, (mk & 2) ? vi * ac : 0.0, (mk & 4) ? -(vi * vj) * ac : 0.0, (mk & 8) ? vi * ad : 0.0);
                }
            }
        }
    }
    // the quad's sums; the diagonal entry sat on exactly one lane (the others hold zeros)
#pragma unroll
    for (int d = 1; d <= 2; d <<= 1) {
        s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d);
        gii += __shfl_xor(gii, d); bii += __shfl_xor(bii, d); pd += __shfl_xor(pd, d);
    }
    double fp = vi * s1 - pinj;
    double fq = vi * s2 - qinj;
    double d00 = -vi * s2 - bii * (vi * vi);
    double d01 = s1 + gii * vi;
    double d10 = vi * s1 - gii * (vi * vi);
    double d11 = s2 - bii * vi;
    if (ti == 3) { d00 = 1.0; d01 = 0.0; d10 = 0.0; d11 = 1.0; fp = 0.0; fq = 0.0; }
    else if (ti == 2) { d01 = 0.0; d10 = 0.0; d11 = 1.0; fq = 0.0; }
    if (JAC && pre) {
        const jg::Blk raw{d00, d01, d10, d11};
        bool bad;
        const jg::Blk f = jg::diag_lu(raw, jg::row_max(raw), bad);
        if (bad) fp = __builtin_nan("");
        d00 = f.v00; d01 = f.v01; d10 = f.v10; d11 = f.v11;
    }
    if (live && q == 0) {
        if (JAC) {
            if (pre) jg::store_vec(a.W, (size_t)pre - 1, b, ld, fp, fq);
            jg::store_blk(a.A, (size_t)pd, b, ld, d00, d01, d10, d11);
        }
        jg::store_vec(a.F, (size_t)i, b, ld, fp, fq);
    }
    // chunk maxima: ASM_ROWS consecutive rows = 4 ASM_ROWS consecutive lanes = one wave (NaN-propagating: a NaN mismatch must not look converged)
    static_assert(ASM_ROWS == 16, "a chunk of rows is one wave of quads");
    double maxp = live ? fabs(fp) : 0.0, maxq = live ? fabs(fq) : 0.0;
    for (int d = 4; d < 64; d <<= 1) {
        const double x = __shfl_xor(maxp, d), y = __shfl_xor(maxq, d);
        maxp = (x > maxp || x != x) ? x : maxp;
        maxq = (y > maxq || y != y) ? y : maxq;
    }
    if (live && ((int)threadIdx.x & 63) == 0) {
        const size_t ck = (size_t)(i0 / ASM_ROWS);
        a.part[(ck * 2) * ld + b] = maxp;
        a.part[(ck * 2 + 1) * ld + b] = maxq;
    }
}

// ---- iterative refinement of the Newton step (opt-in, jg_nr_set_refine) ------------------------------------------------
// The reference's default solver refines: UMFPACK's solve runs up to two steps of iterative refinement behind `ldiv!`
// (/root/reference/src/backend/utility.jl:576-586 -> umfpack_solve, UMFPACK_IRSTEP = 2), which is what covers a weak pivot
// of ITS threshold pivoting.  The device keeps a static pivot order, so the same guard is offered here: after the first solve
//   rho = f - J d         (this kernel: J is NOT stored -- the factor overwrote it -- so the product is formed from Ybus
//                          and the unchanged state with the same sincos terms the assembly uses, one pass over the rows)
//   d  += J^-1 rho        (forward-only sweep + backward sweep on the factor that is already there), then x -= d.
struct RefineArgs {
    const int* rowptr; const int* colm; const double2* GB; const int* rowtype;
    const double* vm; const double* va;
    const int* ppos; const double* pdg; const double* pdb;
    const double* F; const double* inc; double* R; jg::GroupSel sel;
    int n; int ld; int mp; int nchunk; int lanes;
};

template <int MP>
__global__ __launch_bounds__(64 * ASM_WAVES) void k_refine_residual(RefineArgs a) {
    int grp, bx;
    if (!jg::map_block(a.sel, a.ld, a.nchunk, grp, bx)) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min(grp * 64 + lane, a.lanes - 1);
    int ppos[MP > 0 ? MP : 1];
#pragma unroll
    for (int m = 0; m < MP; ++m) ppos[m] = a.ppos[(size_t)m * ld + b];
    const int r0 = bx * ASM_ROWS, r1 = min(r0 + ASM_ROWS, a.n);
    for (int i = r0 + wave; i < r1; i += ASM_WAVES) {
        const int p0 = uniform(a.rowptr[i]), p1 = uniform(a.rowptr[i + 1]);
        const int ti = (int)((unsigned)uniform(a.rowtype[i]));
        const double vi = a.vm[(size_t)i * ld + b], thi = a.va[(size_t)i * ld + b];
        const double2 f = jg::load_vec(a.F, (size_t)i, b, ld);
        const double2 di = jg::load_vec(a.inc, (size_t)i, b, ld);
        double s1 = 0.0, s2 = 0.0, gii = 0.0, bii = 0.0, rp = f.x, rq = f.y;
        for (int p = p0; p < p1; ++p) {
            const int cm = uniform(a.colm[p]);
            const int j = cm & 0xffffff, mk = cm >> 24;
            double2 gb = a.GB[p];
#pragma unroll
            for (int m = 0; m < MP; ++m)
                if (ppos[m] == p) { gb.x += a.pdg[(size_t)m * ld + b]; gb.y += a.pdb[(size_t)m * ld + b]; }
            const double vj = a.vm[(size_t)j * ld + b];
            double s, c;
            sincos(thi - a.va[(size_t)j * ld + b], &s, &c);
            const double ac = gb.x * c + gb.y * s, ad = gb.x * s - gb.y * c;
            s1 += vj * ac;
            s2 += vj * ad;
            if (j == i) { gii = gb.x; bii = gb.y; continue; }
            const double2 dj = jg::load_vec(a.inc, (size_t)j, b, ld);
            rp -= ((mk & 1) ? vi * vj * ad : 0.0) * dj.x + ((mk & 2) ? vi * ac : 0.0) * dj.y;
            rq -= ((mk & 4) ? -(vi * vj) * ac : 0.0) * dj.x + ((mk & 8) ? vi * ad : 0.0) * dj.y;
        }
        double d00 = -vi * s2 - bii * (vi * vi), d01 = s1 + gii * vi, d10 = vi * s1 - gii * (vi * vi), d11 = s2 - bii * vi;
        if (ti == 3) { d00 = 1.0; d01 = 0.0; d10 = 0.0; d11 = 1.0; }
        else if (ti == 2) { d01 = 0.0; d10 = 0.0; d11 = 1.0; }
        rp -= d00 * di.x + d01 * di.y;
        rq -= d10 * di.x + d11 * di.y;
        jg::store_vec(a.R, (size_t)i, b, ld, rp, rq);
    }
}

// d = d1 + d2 (kept as method.increment), then x -= d on the state components of the scenarios still iterating
__global__ void k_refine_apply(double* inc, const double* inc1, const double* inc2, double* va, double* vm, const signed char* flags, const int* active,
                               const jg::GroupSel sel, int n, int ld, int lanes) {
    int grp, bx;
    if (!jg::map_block(sel, ld, (n + 15) / 16, grp, bx)) return;
    const int i = bx * 16 + threadIdx.y;
    if (i >= n) return;
    const int lb = grp * 64 + threadIdx.x;
    if (lb >= lanes) return;
    const size_t b = (size_t)lb, l = (size_t)ld;
    if (active && !active[b]) return;
    const double2 d1 = jg::load_vec(inc1, (size_t)i, b, l), d2 = jg::load_vec(inc2, (size_t)i, b, l);
    const double dx = d1.x + d2.x, dy = d1.y + d2.y;
    jg::store_vec(inc, (size_t)i, b, l, dx, dy);
    const int fl = flags[i];
    if (fl & 1) va[(size_t)i * l + b] -= dx;
    if (fl & 2) vm[(size_t)i * l + b] -= dy;
}

// ---- post-processing: power!/current! branch quantities (acAnalysis.jl:66-81, 688-701) -------------------------
struct BranchArgs {
    const int* from; const int* to; const signed char* status; const double* param;   // param [nb][16], see jgrid.h
    const double* vm; const double* va; const int* outage;                              // outage [ld]: 1-based branch out in that scenario
    double* from_pq; double* to_pq; double* series_pq; double* charging_pq; double* from_i; double* to_i; double* series_i;
    int nb; int ld; int lanes;
};

// one wave per branch x 64 scenarios, 16 branches per workgroup; the branch table travels through the scalar cache
__global__ __launch_bounds__(1024) void k_branch_quantities(BranchArgs a) {
    typedef const double __attribute__((address_space(4)))* CDbl;
    typedef const int __attribute__((address_space(4)))* CInt;
    const int lane = threadIdx.x;
    const int k = blockIdx.x * 16 + uniform(threadIdx.y);
    if (k >= a.nb) return;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min((int)blockIdx.y * 64 + lane, a.lanes - 1);
    CDbl p = (CDbl)a.param + (size_t)k * 16;
    const int i = ((CInt)a.from)[k], j = ((CInt)a.to)[k];
    const bool on = a.status[k] == 1 && a.outage[b] != k + 1;
    const double Vi = a.vm[(size_t)i * ld + b], Vj = a.vm[(size_t)j * ld + b];
    double si, ci, sj, cj;
    sincos(a.va[(size_t)i * ld + b], &si, &ci);
    sincos(a.va[(size_t)j * ld + b], &sj, &cj);
    const double vir = Vi * ci, vii = Vi * si, vjr = Vj * cj, vji = Vj * sj;
    // Iij = Vi yff + Vj yft, Iji = Vi ytf + Vj ytt (:921-927);  Vij = tij Vi - Vj (:846-851), Is = y Vij (:929-931)
    const double ifr = vir * p[0] - vii * p[1] + vjr * p[2] - vji * p[3], ifi = vir * p[1] + vii * p[0] + vjr * p[3] + vji * p[2];
    const double itr = vir * p[4] - vii * p[5] + vjr * p[6] - vji * p[7], iti = vir * p[5] + vii * p[4] + vjr * p[7] + vji * p[6];
    const double wr = p[10] * vir - p[11] * vii - vjr, wi = p[10] * vii + p[11] * vir - vji;
    const double isr = p[8] * wr - p[9] * wi, isi = p[8] * wi + p[9] * wr;
    const double z = on ? 1.0 : 0.0;                                         // out of service: zeros, like initialize! (:66, :688)
    if (a.from_pq) jg::store_vec(a.from_pq, (size_t)k, b, ld, z * (vir * ifr + vii * ifi), z * (vii * ifr - vir * ifi));      // Vi conj(Iij)
    if (a.to_pq) jg::store_vec(a.to_pq, (size_t)k, b, ld, z * (vjr * itr + vji * iti), z * (vji * itr - vjr * iti));          // Vj conj(Iji)
    if (a.series_pq) jg::store_vec(a.series_pq, (size_t)k, b, ld, z * (wr * isr + wi * isi), z * (wi * isr - wr * isi));      // Vij conj(y Vij)
    if (a.charging_pq) {                                                     // 0.5 conj(g + jb) ((Vi/tau)^2 + Vj^2)  (:910-919)
        const double m = 0.5 * ((p[14] * Vi) * (p[14] * Vi) + Vj * Vj);
        jg::store_vec(a.charging_pq, (size_t)k, b, ld, z * p[12] * m, -z * p[13] * m);
    }
    if (a.from_i) jg::store_vec(a.from_i, (size_t)k, b, ld, z * hypot(ifr, ifi), on ? atan2(ifi, ifr) : 0.0);
    if (a.to_i) jg::store_vec(a.to_i, (size_t)k, b, ld, z * hypot(itr, iti), on ? atan2(iti, itr) : 0.0);
    if (a.series_i) jg::store_vec(a.series_i, (size_t)k, b, ld, z * hypot(isr, isi), on ? atan2(isi, isr) : 0.0);
}

// ---- contingency screen summary (SURVEY 8f "next"; jgrid.h: jg_nr_screen): worst branch loading, largest flow, lowest / highest voltage per scenario
// Partial maxima per (chunk of 256 branches resp. 1024 buses, scenario), then one reduction per scenario.  Ties go to the lowest index (fixed order).
struct ScreenArgs {
    const int* from; const int* to; const signed char* status; const double* param; const double* rating;   // rating [nb] or null
    const double* vm; const double* va; const int* outage;
    double* part;                  // [chunks][4][ld]: branch chunks {loading, its branch, flow, its branch}, then bus chunks {vmin, its bus, vmax, its bus}
    const int* iters; const int* status_sc;
    double* rec;                   // [batch][10]
    int nb, n, ld, lanes, bchunks, vchunks;
};
constexpr int SCREEN_BR = 256, SCREEN_BUS = 1024;

__global__ __launch_bounds__(1024) void k_screen_partial(ScreenArgs a) {
    typedef const double __attribute__((address_space(4)))* CDbl;
    typedef const int __attribute__((address_space(4)))* CInt;
    __shared__ double red[16][4][64];
    const int lane = threadIdx.x, wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min((int)blockIdx.y * 64 + lane, a.lanes - 1);
    const int c = blockIdx.x;
    double m0, i0, m1, i1;
    if (c < a.bchunks) {                                         // branches: apparent power at both ends (PijQij, PjiQji: acAnalysis.jl:898-904)
        m0 = 0.0; i0 = 0.0; m1 = 0.0; i1 = 0.0;
        const int k0 = c * SCREEN_BR + wave * (SCREEN_BR / 16);
        for (int k = k0; k < min(k0 + SCREEN_BR / 16, a.nb); ++k) {
            if (a.status[k] != 1) continue;
            CDbl p = (CDbl)a.param + (size_t)k * 16;
            const int i = ((CInt)a.from)[k], j = ((CInt)a.to)[k];
            const bool on = a.outage[b] != k + 1;
            const double Vi = a.vm[(size_t)i * ld + b], Vj = a.vm[(size_t)j * ld + b];
            double si, ci, sj, cj;
            sincos(a.va[(size_t)i * ld + b], &si, &ci);
            sincos(a.va[(size_t)j * ld + b], &sj, &cj);
            const double vir = Vi * ci, vii = Vi * si, vjr = Vj * cj, vji = Vj * sj;
            const double ifr = vir * p[0] - vii * p[1] + vjr * p[2] - vji * p[3], ifi = vir * p[1] + vii * p[0] + vjr * p[3] + vji * p[2];
            const double itr = vir * p[4] - vii * p[5] + vjr * p[6] - vji * p[7], iti = vir * p[5] + vii * p[4] + vjr * p[7] + vji * p[6];
            const double sf = hypot(vir * ifr + vii * ifi, vii * ifr - vir * ifi), st = hypot(vjr * itr + vji * iti, vji * itr - vjr * iti);
            const double s = on ? fmax(sf, st) : 0.0;
            if (s > m1) { m1 = s; i1 = (double)(k + 1); }
            const double r = a.rating ? ((CDbl)a.rating)[k] : 0.0;
            if (r > 0.0 && s / r > m0) { m0 = s / r; i0 = (double)(k + 1); }
        }
    } else {                                                     // buses: lowest and highest voltage magnitude
        m0 = 1.0e300; i0 = 0.0; m1 = -1.0e300; i1 = 0.0;
        const int v0 = (c - a.bchunks) * SCREEN_BUS + wave * (SCREEN_BUS / 16);
        for (int i = v0; i < min(v0 + SCREEN_BUS / 16, a.n); ++i) {
            const double v = a.vm[(size_t)i * ld + b];
            if (v < m0) { m0 = v; i0 = (double)(i + 1); }
            if (v > m1) { m1 = v; i1 = (double)(i + 1); }
        }
    }
    red[wave][0][lane] = m0; red[wave][1][lane] = i0; red[wave][2][lane] = m1; red[wave][3][lane] = i1;
    __syncthreads();
    if (wave != 0) return;
    const bool lo = c >= a.bchunks;                              // the first pair of a bus chunk is a MINIMUM
    for (int w = 1; w < 16; ++w) {                               // ascending index order: a strict comparison keeps the lowest index on ties
        const double x0 = red[w][0][lane], x1 = red[w][2][lane];
        if (lo ? x0 < m0 : x0 > m0) { m0 = x0; i0 = red[w][1][lane]; }
        if (x1 > m1) { m1 = x1; i1 = red[w][3][lane]; }
    }
    double* q = a.part + (size_t)c * 4 * ld + b;
    q[0] = m0; q[ld] = i0; q[2 * ld] = m1; q[3 * ld] = i1;
}

__global__ __launch_bounds__(64) void k_screen_final(ScreenArgs a) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.lanes) return;
    const size_t ld = (size_t)a.ld;
    double load = 0.0, lidx = 0.0, flow = 0.0, fidx = 0.0, vmin = 1.0e300, vminidx = 0.0, vmax = -1.0e300, vmaxidx = 0.0;
    for (int c = 0; c < a.bchunks; ++c) {
        const double* q = a.part + (size_t)c * 4 * ld + b;
        if (q[0] > load) { load = q[0]; lidx = q[ld]; }
        if (q[2 * ld] > flow) { flow = q[2 * ld]; fidx = q[3 * ld]; }
    }
    for (int c = a.bchunks; c < a.bchunks + a.vchunks; ++c) {
        const double* q = a.part + (size_t)c * 4 * ld + b;
        if (q[0] < vmin) { vmin = q[0]; vminidx = q[ld]; }
        if (q[2 * ld] > vmax) { vmax = q[2 * ld]; vmaxidx = q[3 * ld]; }
    }
    double* r = a.rec + (size_t)b * 10;
    const int st = a.status_sc[b];
    if (st == 3) {                                               //  numeric failure: V is NaN, every comparison above was false -- the sentinels would read
        const double nan = __builtin_nan("");                     // as "nothing overloaded".  The summary of such a scenario is NaN (jgrid.h); rec[8], rec[9] say why.
        load = flow = vmin = vmax = nan; lidx = fidx = vminidx = vmaxidx = 0.0;
    }
    r[0] = load; r[1] = lidx; r[2] = flow; r[3] = fidx; r[4] = vmin; r[5] = vminidx; r[6] = vmax; r[7] = vmaxidx;
    r[8] = (double)a.iters[b]; r[9] = (double)st;
}

struct CheckArgs {
    const double* part; int nchunk; int ld; int batch;
    const double* params;      // [0] tolerance, [1] max iterations
    double* normp; double* normq; int* active; int* iters; int* status; const int* lu_status; int* counter; const int* group;
    int mode;                  // 0 = norms only, 1 = powerFlow! loop control
};

// blockDim (64, 16): one workgroup per 64-scenario group; the 16 waves split the per-chunk partial norms
__global__ __launch_bounds__(1024) void k_check(CheckArgs a) {
    __shared__ double red[2][16][64];
    const int lane = threadIdx.x, wave = threadIdx.y;
    const int b = blockIdx.x * 64 + lane;
    if (a.mode == 1 && a.group && !a.group[blockIdx.x]) return;   // finished group: assembly was skipped, keep its verdict
    double mp = 0.0, mq = 0.0;
    constexpr int CU = 8;
    for (int c0 = wave; c0 < a.nchunk; c0 += 16 * CU) {           // eight chunks per trip: sixteen loads in flight (625 chunks on a 10 000-bus
        double x[CU], y[CU];                                      // grid were 39 dependent round trips per wave, 26 us per verdict)
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int c = min(c0 + 16 * u, a.nchunk - 1);         // a repeated chunk changes no maximum
            x[u] = a.part[((size_t)c * 2) * a.ld + b]; y[u] = a.part[((size_t)c * 2 + 1) * a.ld + b];
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            mp = (x[u] > mp || x[u] != x[u]) ? x[u] : mp;
            mq = (y[u] > mq || y[u] != y[u]) ? y[u] : mq;
        }
    }
    red[0][wave][lane] = mp;
    red[1][wave][lane] = mq;
    __syncthreads();
    if (wave != 0) return;
    for (int w = 1; w < 16; ++w) {
        const double x = red[0][w][lane], y = red[1][w][lane];
        mp = (x > mp || x != x) ? x : mp;
        mq = (y > mq || y != y) ? y : mq;
    }
    a.normp[b] = mp;
    a.normq[b] = mq;
    if (a.mode == 0) return;
    const double tol = a.params[0];
    const int maxit = (int)a.params[1];
    const bool real = b < a.batch;
    const bool conv = mp < tol && mq < tol;                       // acPowerFlow.jl:1410 (strict)
    const bool bad = (a.lu_status[b] & 4) || mp != mp || mq != mq;
    const bool act = real && !conv && !bad && a.iters[b] < maxit; // acPowerFlow.jl:1414
    a.active[b] = act ? 1 : 0;
    a.status[b] = conv ? 0 : (bad ? 3 : 1);
    if (act) { a.iters[b] += 1; atomicAdd(a.counter, 1); }        // solve! follows: iteration += 1 (:908)
}

// ---- scenario compaction ------------------------------------------------------------------------
// Scenarios converge after different iteration counts (2..10 on N-1 sets).  Once enough have finished,
// the still-active ones are packed into the leading lanes (stable partition) so whole 64-lane groups
// drop out of every later launch.  flags: [0] permute this iteration, [1] active lanes, [2] groups in use,
// [3] length of glist (the ids of the groups that still hold an active lane; the launches spread exactly those).
struct CompactArgs {
    int* active; int* iters; int* status; int* lu_status; int* lid; int* ppos; int mp;
    int* dest; int* group; int* glist; int* flags; int* tmp; int ld; int restore;
    int* host_count;           // nullable: pinned host word that receives the number of active scenarios (what the host loop polls:
                               // a store from this kernel instead of a memset node and a 4-byte copy node per iteration, ~20 us each)
    int* host_res;             // nullable (handles of ONE lane group): pinned host words [128] that receive iterations | status of the 64 lanes when the last scenario
                               // has finished -- jg_nr_run then returns them without two blocking 4-byte-per-lane copies (~25 us each: a tenth of a single instance's solve
                               // went into run_setup's and run_finish's small transfers, round 6)
    const double* params;      // [2] = defer_at of the run (jg_nr_run_defer; 0: none).  Once at most that many scenarios are active the host stops the batch and
                               // hands them to a pool (jg_nr_move_lanes): packing them into the first lane group -- and sending every lane home again when the
                               // batch finishes -- moved every row of the handle twice for lanes that leave anyway (round 6: k_lanes_permute was 5.8 % of the GPU
                               // time of the pipeline).  The lanes stay where they are; dest[0 .. n_active) lists them for the hand-off, flags[4] says so.
};

// inclusive scan over the 1024 threads of the workgroup: shuffles inside a wave, the 16 wave totals through LDS (two barriers; the
// Hillis-Steele loop it replaces had twenty and made the verdict of a single instance 12 us long)
__device__ __forceinline__ int block_scan_1024(int v, int* wsum, int t, int& total) {
    const int lane = t & 63, w = t >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int q = wsum[k]; off += k < w ? q : 0; total += q; }
    __syncthreads();
    return x + off;
}

// run_setup on the device: parameters of the run, every real scenario active with zero iterations, lanes in home order, all lane groups in use
__global__ void k_run_setup(double* params, double tol, double max_iter, double defer_at, int* iters, int* lu_status, int* active, int* lid, int* cflags, int* group, int* glist,
                            int ld, int lanes, int keep_iters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < ld) {
        if (!keep_iters) { iters[b] = 0; lu_status[b] = 0; active[b] = b < lanes ? 1 : 0; }
        lid[b] = b;
    }
    if (b < ld / 64) { group[b] = 1; glist[b] = b; }
    if (b == 0) {
        params[0] = tol; params[1] = max_iter; params[2] = defer_at;
        cflags[0] = 0; cflags[1] = lanes; cflags[2] = ld / 64; cflags[3] = ld / 64; cflags[4] = 0; cflags[5] = 0; cflags[6] = 0; cflags[7] = 0;
    }
}

__global__ __launch_bounds__(1024) void k_compact(CompactArgs a) {
    __shared__ int wsum[16];
    const int t = threadIdx.x;
    const int ld = a.ld, ngroups = ld / 64;
    const int per = (ld + 1023) / 1024;
    const int b0 = t * per, b1 = min(b0 + per, ld);
    int cnt = 0;
    for (int b = b0; b < b1; ++b) cnt += a.active[b] != 0;
    int n_active;
    (void)block_scan_1024(cnt, wsum, t, n_active);
    const int groups_new = (n_active + 63) / 64;
    const int groups_cur = a.flags[2];
    const int defer_at = a.params ? (int)a.params[2] : 0;
    const bool hold = !a.restore && defer_at > 0 && n_active > 0 && n_active <= defer_at;
    bool moved = false;
    if (a.restore) {                                              // lanes that never left home need no pass over the rows
        int away = 0;
        for (int b = b0; b < b1; ++b) away += a.lid[b] != b;
        int any;
        (void)block_scan_1024(away, wsum, t, any);
        moved = any > 0;
    }
    const bool permute = a.restore ? moved : (!hold && n_active > 0 && groups_new < groups_cur);
    if (a.host_res && n_active == 0 && t < 64) { a.host_res[t] = a.iters[t]; a.host_res[64 + t] = a.status[t]; __threadfence_system(); }   // ... ahead of the word the host polls
    __syncthreads();
    if (t == 0) { a.flags[0] = permute ? 1 : 0; a.flags[1] = n_active; a.flags[4] = hold ? 1 : 0; if (permute) a.flags[2] = a.restore ? ngroups : groups_new; if (a.host_count) *a.host_count = n_active; }
    if (a.restore && !permute) {
        for (int g = t; g < ngroups; g += 1024) { a.group[g] = 1; a.glist[g] = g; }
        if (t == 0) { a.flags[2] = ngroups; a.flags[3] = ngroups; }
        return;
    }
    if (hold) {                                                   // the hand-off's list of the lanes that leave, in lane order
        int tot;
        const int incl = block_scan_1024(cnt, wsum, t, tot);
        int r = incl - cnt;
        for (int b = b0; b < b1; ++b) if (a.active[b] != 0) a.dest[r++] = b;
    }
    if (!permute) {                                               // groups = those that still hold an active lane
        for (int g = t; g < ngroups; g += 1024) {
            int any = 0;
            for (int l = 0; l < 64; ++l) any |= a.active[g * 64 + l];
            a.group[g] = any;
        }
        __syncthreads();
        if (t == 0) {
            int m = 0;
            for (int g = 0; g < ngroups; ++g) if (a.group[g]) a.glist[m++] = g;
            a.flags[3] = m;
        }
        return;
    }
    if (a.restore) {
        for (int b = b0; b < b1; ++b) a.dest[b] = a.lid[b];      // send every lane home
    } else {
        // Minimal moves: an active lane inside the leading groups_new groups stays where it is; the k-th active lane behind them
        // swaps with the k-th inactive lane inside them.  (A stable partition shifted almost every lane: 1 GB of lane rows per
        // compaction of 512 scenarios of a 10 000-bus grid, and again for the way home; 37 stragglers of 512 now move 64 lanes.)
        const int L = groups_new * 64;
        int holes = 0, outs = 0;
        for (int b = b0; b < b1; ++b) { const bool act = a.active[b] != 0; holes += (b < L && !act); outs += (b >= L && act); a.dest[b] = b; }
        int both;
        const int incl = block_scan_1024(holes | outs << 16, wsum, t, both);   // both counts in one scan: ld < 65 536
        int hr = (incl & 0xffff) - holes, orank = (incl >> 16) - outs;         // holes / outside actives in front of this thread's chunk
        int* hole_at = a.tmp;                                     // [k] position of the k-th hole, [ld / 2 + k] of the k-th outside active lane
        int* out_at = a.tmp + ld / 2;
        const int n_out = both >> 16;                             // <= min(L, ld - L) <= ld / 2; the leading groups hold at least as many holes
        for (int b = b0; b < b1; ++b) {
            const bool act = a.active[b] != 0;
            if (b < L && !act) { if (hr < n_out) hole_at[hr] = b; ++hr; }
            if (b >= L && act) out_at[orank++] = b;
        }
        __syncthreads();
        for (int k = t; k < n_out; k += 1024) { const int hpos = hole_at[k], opos = out_at[k]; a.dest[opos] = hpos; a.dest[hpos] = opos; }
    }
    __syncthreads();
    int* arrays[5] = {a.active, a.iters, a.status, a.lu_status, a.lid};
    for (int k = 0; k < 5 + a.mp; ++k) {
        int* x = k < 5 ? arrays[k] : a.ppos + (size_t)(k - 5) * ld;
        for (int b = b0; b < b1; ++b) a.tmp[a.dest[b]] = x[b];
        __syncthreads();
        for (int b = b0; b < b1; ++b) x[b] = a.tmp[b];
        __syncthreads();
    }
    for (int g = t; g < ngroups; g += 1024) { a.group[g] = a.restore ? 1 : (g < groups_new); a.glist[g] = g; }
    if (t == 0) a.flags[3] = a.restore ? ngroups : groups_new;
}

// dst[row][dest[b]] = src[row][b] for `rows` rows (no-op unless flags[0]); then the copy back
// ELEM doubles per (row, lane): 1 for V / theta / P / Q rows, 2 for the interleaved mismatch / increment rows
template <int ELEM>
__global__ void k_lane_permute(const double* src, double* dst, const int* dest, const int* flags, int rows, int ld) {
    if (!flags[0]) return;
    const int b = blockIdx.y * 256 + threadIdx.x;
    if (b >= ld) return;
    const int d = dest[b];
    if (d == b) return;
    for (int r = blockIdx.x; r < rows; r += gridDim.x)
#pragma unroll
        for (int e = 0; e < ELEM; ++e) dst[((size_t)r * ld + d) * ELEM + e] = src[((size_t)r * ld + b) * ELEM + e];
}

template <int ELEM>
__global__ void k_lane_copy(const double* src, double* dst, const int* dest, const int* flags, int rows, int ld) {
    if (!flags[0]) return;
    const int b = blockIdx.y * 256 + threadIdx.x;
    if (b >= ld || dest[b] == b) return;
    for (int r = blockIdx.x; r < rows; r += gridDim.x)
#pragma unroll
        for (int e = 0; e < ELEM; ++e) dst[((size_t)r * ld + b) * ELEM + e] = src[((size_t)r * ld + b) * ELEM + e];
}

// All per-scenario arrays of a compaction in TWO launches (scatter into the staging area, copy back) instead of two per array:
// the launches are predicated on flags[0] and mostly return at once, so what they cost is their number.
struct LaneSet { double* x[8]; int rows[8]; int elem[8]; long long off[8]; };
__global__ void k_lanes_move(LaneSet s, double* tmp, const int* dest, const int* flags, int ld, int back) {
    if (!flags[0]) return;
    const int b = blockIdx.y * 256 + threadIdx.x;
    if (b >= ld) return;
    const int a = blockIdx.z;
    const int rows = s.rows[a], E = s.elem[a];
    double* x = s.x[a];
    double* t = tmp + s.off[a];
    const int d = dest[b];
    if (d == b) return;                            // this lane stays (a permutation maps the lanes that move onto themselves)
    if (E == 2) {                                  // interleaved 2-vectors: one 16-byte access per lane
        double2* x2 = (double2*)x;
        double2* t2 = (double2*)t;
        for (int r = blockIdx.x; r < rows; r += gridDim.x) {
            if (back) x2[(size_t)r * ld + b] = t2[(size_t)r * ld + b];
            else t2[(size_t)r * ld + d] = x2[(size_t)r * ld + b];
        }
        return;
    }
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        if (back) x[(size_t)r * ld + b] = t[(size_t)r * ld + b];
        else t[(size_t)r * ld + d] = x[(size_t)r * ld + b];
    }
}

// The same move IN PLACE, one launch (round 5): dest is a permutation of the lanes (k_compact: swaps of a hole with a straggler; restore: every lane to its home), so a
// workgroup that holds ALL lanes of a row reads what moves, waits for the values, meets at a barrier and writes them to their new lanes -- no staging area, half
// the traffic and half the launches of the two-pass move (which remains for handles of more than LANES_INPLACE lanes).
constexpr int LANES_INPLACE = 4096;
__global__ __launch_bounds__(1024) void k_lanes_permute(LaneSet s, const int* dest, const int* flags, int ld) {
    if (!flags[0]) return;
    constexpr int NL = LANES_INPLACE / 1024;
    const int a = blockIdx.z;
    const int rows = s.rows[a], E = s.elem[a];
    int b[NL], d[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) { b[k] = (int)threadIdx.x + k * 1024; d[k] = b[k] < ld ? dest[b[k]] : b[k]; }
    if (E == 2) {
        double2* x = (double2*)s.x[a];
        for (int r = blockIdx.x; r < rows; r += gridDim.x) {
            double2 v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) if (d[k] != b[k]) v[k] = x[(size_t)r * ld + b[k]];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the values are HERE before any lane of the row is overwritten
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NL; ++k) if (d[k] != b[k]) x[(size_t)r * ld + d[k]] = v[k];
        }
        return;
    }
    double* x = s.x[a];
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        double v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) if (d[k] != b[k]) v[k] = x[(size_t)r * ld + b[k]];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NL; ++k) if (d[k] != b[k]) x[(size_t)r * ld + d[k]] = v[k];
    }
}

// [n][ld] batch-minor -> [batch][n] scenario-major, tiled through LDS so both sides stay coalesced
__global__ void k_to_scenario_major(const double* src, double* dst, int n, int ld, int batch) {
    __shared__ double tile[32][33];
    const int i0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int i = i0 + r, b = b0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && b < ld) ? src[(size_t)i * ld + b] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int b = b0 + r, i = i0 + threadIdx.x;
        if (b < batch && i < n) dst[(size_t)b * n + i] = tile[threadIdx.x][r];
    }
}

// [n][ld] batch-minor -> column block `off` of a scenario-major result record [batch][stride] (the packed result of a batch:
// V | theta | iterations | status per scenario, one buffer for the one gather of a sharded run)
// (64 x 64 tiles: a wave reads and writes 512 contiguous bytes; both arrays of a record in one launch, blockIdx.z)
__global__ __launch_bounds__(512) void k_pack_bus(const double* vm, const double* va, double* dst, int n, int ld, int batch, long long stride) {
    __shared__ double tile[64][65];
    const double* src = blockIdx.z ? va : vm;
    const int off = blockIdx.z ? n : 0;
    const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int i = i0 + r, b = b0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && b < ld) ? src[(size_t)i * ld + b] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int b = b0 + r, i = i0 + threadIdx.x;
        if (b < batch && i < n) dst[(size_t)b * stride + off + i] = tile[threadIdx.x][r];
    }
}
__global__ void k_pack_tail(const int* iters, const int* status, double* dst, int batch, long long stride, int off) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) { dst[(size_t)b * stride + off] = (double)iters[b]; dst[(size_t)b * stride + off + 1] = (double)status[b]; }
}

// ---- straggler hand-off (jg_nr_move_lanes): the scenarios of a paused batch that are still active -- all inside its first
// lane group -- continue in another handle of the same grid (a pool that collects the stragglers of several batches).
// k_move_plan (one wave): pool lane of every active position, the per-lane integers, the home lane of each moved scenario; the
// source lanes become inactive with status 4 (deferred).  k_move_rows: the per-lane rows (V, theta, P, Q, patch values).
struct MovePlanArgs {
    int* s_active; int* s_iters; int* s_status; const int* s_lid; const int* s_ppos; int s_ld;
    int* d_active; int* d_iters; int* d_status; int* d_lu; int* d_ppos; int d_ld;
    int* map; int* home; int* count; int mp; int lane0; int cap;
    const int* s_flags; const int* s_list; int* srcl;          // the source's compaction flags / list of the lanes that leave (k_compact: hold); [64] source lane of candidate p
};
__global__ __launch_bounds__(64) void k_move_plan(MovePlanArgs a) {
    const int p0 = threadIdx.x;
    // where the p-th candidate sits: packed into the first lane group by a compaction, or -- the batch stopped for this hand-off (k_compact: hold) -- wherever
    // it was, listed in the source's dest[]
    const bool listed = a.s_flags[4] != 0;
    const int p = listed ? (p0 < a.s_flags[1] ? a.s_list[p0] : -1) : p0;
    a.srcl[p0] = p;
    const bool act = p >= 0 && a.s_active[p] != 0;
    const unsigned long long m = __ballot(act);
    const int r = __popcll(m & ((1ull << p0) - 1ull));
    const int total = __popcll(m);
    const bool fits = a.lane0 + total <= a.cap;
    if (p0 == 0) a.count[0] = fits ? total : -1;
    a.map[p0] = -1;
    if (!act || !fits) return;
    const int d = a.lane0 + r;
    a.map[p0] = d;
    a.home[r] = a.s_lid[p];
    a.d_active[d] = 1; a.d_iters[d] = a.s_iters[p]; a.d_status[d] = 1; a.d_lu[d] = 0;
    for (int k = 0; k < a.mp; ++k) a.d_ppos[(size_t)k * a.d_ld + d] = a.s_ppos[(size_t)k * a.s_ld + p];
    a.s_active[p] = 0; a.s_status[p] = 4;
}
struct MoveRowsArgs { const double* src[6]; double* dst[6]; int rows[6]; int s_ld; int d_ld; const int* map; const int* srcl; };
__global__ __launch_bounds__(256) void k_move_rows(MoveRowsArgs a) {
    const int p0 = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int d = a.map[p0];
    if (d < 0) return;
    const int p = a.srcl[p0];
    const int arr = blockIdx.y;
    const double* s = a.src[arr];
    double* t = a.dst[arr];
    for (int r = blockIdx.x * 4 + sub; r < a.rows[arr]; r += gridDim.x * 4) t[(size_t)r * a.d_ld + d] = s[(size_t)r * a.s_ld + p];
}
// rows of a result record [.][stride] <- V | theta | iterations | status of lanes lane0 .. lane0 + count - 1 (rows[i] = record row)
__global__ __launch_bounds__(256) void k_pack_rows(const double* vm, const double* va, const int* iters, const int* status, const int* rows,
                                                   double* dst, int n, int ld, int lane0, long long stride) {
    const int i = blockIdx.y;
    const int b = lane0 + i;
    double* out = dst + (size_t)rows[i] * stride;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) { out[j] = vm[(size_t)j * ld + b]; out[n + j] = va[(size_t)j * ld + b]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[2 * n] = (double)iters[b]; out[2 * n + 1] = (double)status[b]; }
}

__global__ void k_add_iter(int* iters, int n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) iters[b] += 1;
}

}  // namespace

namespace jg {
void set_last_error(const std::string& msg) { g_error = msg; }
}

struct jg_nr {
    int n = 0, nnz = 0, batch = 0, ld = 0, mp = 0, device = 0, nchunk = 0;
    int64_t dimJ = 0, nnzJ = 0, slack = 0;
    std::vector<int64_t> colptr, rowval, pq, pvpq, pcount, jcolptr, jrowval;
    std::vector<int8_t> type;
    std::vector<int> tperm;          // Ybus CSC pointer -> block CSR index of the same (row, col)
    std::vector<int64_t> jmap;       // Jacobian CSC nz -> (factor entry holding the block)*4 + component
    // device
    int* d_rowptr = nullptr; int* d_col = nullptr; double* d_G = nullptr; double* d_B = nullptr; double2* d_GB = nullptr; int* d_rowtype = nullptr; int* d_rowtype_pre = nullptr;   // _pre: type | (pivot + 1) << 2 for the pivots of the plan's level 0
    signed char* d_type = nullptr; signed char* d_flags = nullptr;
    void* d_arena = nullptr;                   // one allocation behind the per-handle state below (jg_nr_create)
    double* d_vm = nullptr; double* d_va = nullptr; double* d_p = nullptr; double* d_q = nullptr;
    int* d_ppos = nullptr; double* d_pdg = nullptr; double* d_pdb = nullptr;
    int* d_dst = nullptr; double* d_F = nullptr; double* d_inc = nullptr; double* d_part = nullptr;
    double* d_stage = nullptr; size_t stage_bytes = 0;           // host rows on their way to a [n][ld] array (put_bus_array)
    double* d_normp = nullptr; double* d_normq = nullptr; double* d_params = nullptr;
    double* d_vm0 = nullptr; double* d_va0 = nullptr;   // snapshot of the start point
    int* d_active = nullptr; int* d_iters = nullptr; int* d_status = nullptr; int* d_counter = nullptr; int* d_group = nullptr;
    int* d_lid = nullptr; int* d_dest = nullptr; int* d_cflags = nullptr; int* d_itmp = nullptr; int* d_glist = nullptr;   // scenario compaction
    bool refine = false;                              // one step of iterative refinement per Newton step (jg_nr_set_refine)
    bool fast = false;                                // fast decoupled mode (jg_nr_fast_setup): constant B', B'' factorised once
    std::vector<double> fast_blk;                     // ... the shared block matrix diag(B', B'') as set up [nnz][4] (engine block order)
    std::vector<int> fp_entry; std::vector<double> fp_dp, fp_dq;   // per-scenario edits of B' / B'' (jg_nr_fast_patch_batch): [FAST_MP][batch] factor entry (-1: none), deltas
    int* d_fp_entry = nullptr; double* d_fp_dp = nullptr; double* d_fp_dq = nullptr;   // ... their device copies (allocated on first use)
    double* d_R = nullptr;                            // rhs of the half-iterations
    double* d_inc2[2] = {nullptr, nullptr};           // increments of the theta / V half-iterations
    const int* fast_mask = nullptr;                   // scenarios that take the update (nullptr: all)
    hipGraph_t graphFA = nullptr, graphFB = nullptr; hipGraphExec_t execFA = nullptr, execFB = nullptr;
    int nb = 0;                                       // post-processing (jg_nr_set_branches)
    int* d_bfrom = nullptr; int* d_bto = nullptr; signed char* d_bstatus = nullptr; double* d_bparam = nullptr; int* d_outage = nullptr;
    double* d_post = nullptr; size_t post_bytes = 0;  // staging for branch / bus quantities, grown on demand
    double* d_rating = nullptr; double* d_screen = nullptr; double* d_screc = nullptr;   // contingency screen: ratings [nb], partial maxima, the record [batch][10]
    jg::Engine eng;
    hipStream_t stream = nullptr;
    hipGraph_t graphA = nullptr, graphB = nullptr, graphBm = nullptr, graphJ = nullptr;
    hipGraphExec_t execA = nullptr, execB = nullptr, execBm = nullptr, execJ = nullptr;   // Bm / J: an iteration whose verdict assembles NO Jacobian / the Jacobian alone (run_loop)
    hipGraph_t graphM = nullptr; hipGraphExec_t execM = nullptr; int m_iters = 0;          // ONE scenario: a whole solve of m_iters iterations as one graph (run_whole)
    int stop_hint = 0, iter_graphs = 0;               // iteration graphs after which the last run of this handle stopped / launched by the current run
    long long last_guess_hit = 0, last_guess_miss = 0;
    bool jac_valid = false;
    bool level0_done = false;        // the Jacobian in the factor storage came from an assembly that finished the plan's level 0
    bool f_stale = false;            // d_F does not hold every scenario's final mismatch yet (see run_finish)
    bool paused = false;             // jg_nr_run_defer stopped with scenarios still active (lanes compacted, not yet sent home)
    int* d_move = nullptr;           // straggler hand-off: map[64] | home[64] | count[1] | source lane[64] (device), home/count mirrored in h_move (pinned)
    int* h_move = nullptr;
    double host_launch_us = 0.0, host_wait_us = 0.0; long long host_iters = 0;   // JG_HOST_TIMING=1: what the host spent in hipGraphLaunch / waiting per iteration of run_loop
    double wait_us = 0.0;            // how long the host waited for the last verdicts (running mean; wait_verdict: polls the pinned word while this is short)
    bool res_pinned = false;         // this run's verdicts write iterations | status behind the pinned verdict word (one lane group; run_finish)
    int* h_counter = nullptr;        // pinned (129 ints: the verdict word, then iterations [64] | status [64] of a handle of one lane group)
    int* h_counter_dev = nullptr;    // its device alias
    // ---- first iteration on a shared factor (jg_comp.hpp; jg_nr_attach_base) ----
    std::vector<int> csr_row, csr_col;                // Ybus row-CSR position -> (row, column)
    jg_nr_base* base = nullptr;                       // the base case this handle's scenarios start from (nullptr: none attached)
    double* d_cw = nullptr;                           // scratch of the shared-factor sweeps [(n + top rows)][ld][2]
    bool start_is_base = false;                       // V, theta of every lane = the base's start (jg_nr_start_from_base; any other write clears it)
    int first_mode = 1;                               // jg_nr_set_first_iteration: 1 = compensated when the conditions hold, 0 = always refactorise
    std::vector<unsigned char> patch_state;           // [batch] 0 no edit, 1 edits within two adjacent buses (compensable), 2 other
    bool inj_checked = false, inj_equal = false;      // the lanes' injections were compared with the base's / equal it
    int* d_cmp = nullptr;                             // one word: mismatching injections (inj check)
    hipGraph_t graphA2 = nullptr, graphC = nullptr;
    hipGraphExec_t execA2 = nullptr, execC = nullptr;
    long long first_comp = 0, first_full = 0;         // runs whose first iteration went the one / the other way (jg_nr_first_iteration_counts)
};

// The base case of a screen: ONE factorisation of the Jacobian at the common start state and what the per-scenario corrections read (jg_comp.hpp).
struct jg_nr_base {
    jg::CompBase cb;
    int n = 0, nnz = 0, device = 0;
    std::vector<int64_t> colptr, rowval;
    std::vector<int8_t> type;
    hipStream_t stream = nullptr;
    std::atomic<int> refs{1};                         // the creator + every attached handle
    double create_ms = 0.0;
};

namespace {

int set_device(jg_nr* h) { NR_HIP(hipSetDevice(h->device)); return 0; }

// What the first iteration on a shared factor may assume of a scenario's Ybus edits (jg_comp.hpp): they lie in the rows / columns of at most two buses,
// and two buses are joined by one of the edited entries (so J_0^-1 has their blocks on the Ybus pattern) -- what a branch outage / parameter change / a
// shunt change produce.  pos: row-CSR positions (-1 = none).
unsigned char classify_patch(const jg_nr* h, const int* pos, int k) {
    int a = -1, b = -1;
    bool any = false, joined = false;
    for (int m = 0; m < k; ++m) {
        if (pos[m] < 0) continue;
        any = true;
        const int ij[2] = {h->csr_row[pos[m]], h->csr_col[pos[m]]};
        for (int q = 0; q < 2; ++q) {
            if (a < 0 || ij[q] == a) a = ij[q];
            else if (b < 0 || ij[q] == b) b = ij[q];
            else return 2;
        }
        if (ij[0] != ij[1]) joined = true;
    }
    if (!any) return 0;
    return (b < 0 || joined) ? 1 : 2;
}

jg::GroupSel active_groups(jg_nr* h) { return jg::GroupSel{nullptr, h->d_glist, h->d_cflags + 3}; }

// level0: the assembly also finishes level 0 of the engine's plan (factorised diagonal blocks + rhs rows of the pivots nobody
// updates); the factorisation that follows must be told (h->level0_done).  Plain assemblies (getters, jg_nr_mismatch) leave it off.
void launch_assemble(jg_nr* h, const jg::GroupSel& sel = jg::GroupSel{}, bool jac = true, double* pq_out = nullptr, int fd_mode = 0, const int* only_if = nullptr,
                     bool level0 = false) {
    const bool pre = jac && level0 && h->eng.plan->S.prefactor && h->d_rowtype_pre;
    AsmArgs a{h->d_rowptr, h->d_col, h->d_GB, pre ? h->d_rowtype_pre : h->d_rowtype, h->d_dst, h->d_vm, h->d_va, h->d_p, h->d_q,
              h->d_ppos, h->d_pdg, h->d_pdb, h->eng.X, h->d_F, h->d_part, pq_out, sel, fd_mode ? h->d_R : nullptr, fd_mode, h->n, h->ld, h->mp, h->nchunk, h->batch, only_if,
              h->eng.W, h->eng.status};
    if (jac && !only_if) h->level0_done = pre;
    dim3 grid(jg::grid_blocks(h->ld / 64, h->nchunk)), block(64, ASM_WAVES);
    const bool single_env = jg::knob("SINGLE", 1) != 0;           // (read at every capture: bench.py holds the two forms against each other in one process)
    if (h->ld == 64 && h->batch == 1 && !fd_mode && !pq_out && !only_if && single_env) {      // ONE scenario: a quad of lanes per bus row (k_assemble1)
        const dim3 g1(jg::grid_blocks(1, (h->n + 63) / 64));
        if (jac) {
            switch (h->mp) {
                case 0: hipLaunchKernelGGL((k_assemble1<0, true>), g1, dim3(256), 0, h->stream, a); break;
                case 4: hipLaunchKernelGGL((k_assemble1<4, true>), g1, dim3(256), 0, h->stream, a); break;
                default: hipLaunchKernelGGL((k_assemble1<8, true>), g1, dim3(256), 0, h->stream, a); break;
            }
        } else {
            switch (h->mp) {
                case 0: hipLaunchKernelGGL((k_assemble1<0, false>), g1, dim3(256), 0, h->stream, a); break;
                case 4: hipLaunchKernelGGL((k_assemble1<4, false>), g1, dim3(256), 0, h->stream, a); break;
                default: hipLaunchKernelGGL((k_assemble1<8, false>), g1, dim3(256), 0, h->stream, a); break;
            }
        }
        return;
    }
    if (h->ld == 64 && h->batch <= 32 && !fd_mode && !pq_out) {      // a handful of scenarios: one row per wave (k_assemble: WAVES = 16); same arithmetic per row, same chunk maxima
        const bool w16 = h->nchunk <= 512;                          // grids whose workgroups are all resident at once: one row per wave; larger ones: two
        const dim3 wide(64, w16 ? 16 : 8);
#define JG_ASM_WIDE(MPV, JACV) do { if (w16) hipLaunchKernelGGL((k_assemble<MPV, JACV, 16>), grid, wide, 0, h->stream, a); else hipLaunchKernelGGL((k_assemble<MPV, JACV, 8>), grid, wide, 0, h->stream, a); } while (0)
        if (jac) {
            switch (h->mp) {
                case 0: JG_ASM_WIDE(0, true); break;
                case 4: JG_ASM_WIDE(4, true); break;
                default: JG_ASM_WIDE(8, true); break;
            }
        } else {
            switch (h->mp) {
                case 0: JG_ASM_WIDE(0, false); break;
                case 4: JG_ASM_WIDE(4, false); break;
                default: JG_ASM_WIDE(8, false); break;
            }
        }
#undef JG_ASM_WIDE
        return;
    }
    if (jac) {
        switch (h->mp) {
            case 0: hipLaunchKernelGGL((k_assemble<0, true>), grid, block, 0, h->stream, a); break;
            case 4: hipLaunchKernelGGL((k_assemble<4, true>), grid, block, 0, h->stream, a); break;
            default: hipLaunchKernelGGL((k_assemble<8, true>), grid, block, 0, h->stream, a); break;
        }
    } else {
        switch (h->mp) {
            case 0: hipLaunchKernelGGL((k_assemble<0, false>), grid, block, 0, h->stream, a); break;
            case 4: hipLaunchKernelGGL((k_assemble<4, false>), grid, block, 0, h->stream, a); break;
            default: hipLaunchKernelGGL((k_assemble<8, false>), grid, block, 0, h->stream, a); break;
        }
    }
}

// The verdict of a handle of ONE lane group in one launch (round 6): k_check's norms and loop control, then what k_compact comes to when there is nothing to pack --
// the count of active scenarios for the host, the group list of the next launches, iterations | status for jg_nr_run when the last scenario has finished.  Two
// launches (12.5 + 4 us of a single instance's 330 us iteration: five trips of dependent loads over the 625 chunk norms, a launch boundary) become one of two trips.
__global__ __launch_bounds__(1024) void k_verdict64(CheckArgs a, CompactArgs p) {
    __shared__ double red[2][16][64];
    const int lane = threadIdx.x, wave = threadIdx.y;
    const int b = lane;
    // every request of the launch leaves at once: the group flag (a finished group keeps its verdict: its assembly was skipped), what the loop control reads
    // besides the norms, and the chunk maxima -- nothing is asked for behind a branch on something else that was asked for
    const int grp_on = a.group ? a.group[0] : 1;
    const double tol = a.params[0];
    const int maxit = (int)a.params[1];
    const int lus = a.lu_status[b], its = a.iters[b], act0 = a.active[b], st0 = a.status[b];
    double mp = 0.0, mq = 0.0;
    auto nmax = [](double x, double m) { return (x > m || x != x) ? x : m; };      // NaN-propagating: a NaN mismatch must not look converged
    if (a.batch == 1) {
        // ONE scenario: the threads are CHUNKS (scenario 0 of chunk t, t + 1024, ...): one round trip where 16 waves of 64 identical lanes took two to five
        for (int c = wave * 64 + lane; c < a.nchunk; c += 1024) {
            mp = nmax(a.part[((size_t)c * 2) * a.ld], mp);
            mq = nmax(a.part[((size_t)c * 2 + 1) * a.ld], mq);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mp = nmax(__shfl_xor(mp, d), mp); mq = nmax(__shfl_xor(mq, d), mq); }
        if (lane < 16) { red[0][wave][0] = mp; red[1][wave][0] = mq; }            // (lanes 0 .. 15 write the same value)
    } else {
        constexpr int CU = 20;
        for (int c0 = wave; c0 < a.nchunk; c0 += 16 * CU) {
            double x[CU], y[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int c = min(c0 + 16 * u, a.nchunk - 1);     // a repeated chunk changes no maximum
                x[u] = a.part[((size_t)c * 2) * a.ld + b]; y[u] = a.part[((size_t)c * 2 + 1) * a.ld + b];
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) { mp = nmax(x[u], mp); mq = nmax(y[u], mq); }
        }
        red[0][wave][lane] = mp;
        red[1][wave][lane] = mq;
    }
    __syncthreads();
    if (wave != 0) return;
    const bool skip = grp_on == 0;
    int act = act0, it_out = its, st_out = st0;
    if (!skip) {
        const int src = a.batch == 1 ? 0 : lane;                  // one scenario: every lane takes its maxima (lanes beyond the batch alias the last scenario)
        for (int w = a.batch == 1 ? 0 : 1; w < 16; ++w) { mp = nmax(red[0][w][src], mp); mq = nmax(red[1][w][src], mq); }
        a.normp[b] = mp;
        a.normq[b] = mq;
        const bool real = b < a.batch;
        const bool conv = mp < tol && mq < tol;                   // acPowerFlow.jl:1410 (strict)
        const bool bad = (lus & 4) || mp != mp || mq != mq;
        const bool go = real && !conv && !bad && its < maxit;     // acPowerFlow.jl:1414
        act = go ? 1 : 0;
        a.active[b] = act;
        st_out = conv ? 0 : (bad ? 3 : 1);
        a.status[b] = st_out;
        if (go) { it_out = its + 1; a.iters[b] = it_out; atomicAdd(a.counter, 1); }     // solve! follows: iteration += 1 (:908)
    }
    const int n_active = __popcll(__ballot(act != 0));
    if (p.host_res && n_active == 0) { p.host_res[lane] = it_out; p.host_res[64 + lane] = st_out; __threadfence_system(); }   // ... ahead of the word the host polls
    if (lane == 0) {
        p.flags[0] = 0; p.flags[1] = n_active; p.flags[4] = 0;
        p.group[0] = n_active > 0 ? 1 : 0;
        if (n_active > 0) p.glist[0] = 0;
        p.flags[3] = n_active > 0 ? 1 : 0;
        if (p.host_count) *p.host_count = n_active;
    }
}

void launch_check(jg_nr* h, int mode, const int* group = nullptr) {
    CheckArgs c{h->d_part, h->nchunk, h->ld, h->batch, h->d_params, h->d_normp, h->d_normq, h->d_active,
                h->d_iters, h->d_status, h->eng.status, h->d_counter, group, mode};
    hipLaunchKernelGGL(k_check, dim3(h->ld / 64), dim3(64, 16), 0, h->stream, c);
}

// host [batch][n] (or one [n] broadcast) -> device [n][ld]
// [rows][n] (scenario-major, as the host hands it over; rows = 1: one row for every scenario) -> dst [n][ld]; lanes beyond the batch
// repeat the last scenario.  64 x 64 tiles through LDS: contiguous reads along the bus index, contiguous writes along the lanes.
__global__ __launch_bounds__(512) void k_spread_bus(const double* src, double* dst, int n, int ld, int batch, int rows) {
    __shared__ double tile[64][65];
    const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int b = min(min(b0 + r, batch - 1), rows - 1), i = i0 + threadIdx.x;
        tile[r][threadIdx.x] = i < n ? src[(size_t)b * n + i] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int i = i0 + r, b = b0 + threadIdx.x;
        if (i < n && b < ld) dst[(size_t)i * ld + b] = tile[threadIdx.x][r];
    }
}

// host [batch][n] (stride n) or [n] (stride 0: every scenario the same) -> device [n][ld].  The rows go up as they are (80 KB for one row
// of a 10 000-bus grid; the host-side transposition to [n][ld] that used to happen here moved 5 MB per array even for one scenario) and
// a kernel spreads them over the lanes.
int put_bus_array(jg_nr* h, double* dst, const double* src, int64_t stride) {
    const int rows = stride == 0 ? 1 : h->batch;
    if (stride != 0 && stride != h->n) {                         // a caller's own row pitch: the general (slow) way
        std::vector<double> t((size_t)h->n * h->ld, 0.0);
        for (int b = 0; b < h->ld; ++b) {
            const double* s = src + (size_t)(b < h->batch ? b : h->batch - 1) * (size_t)stride;   // pad with last scenario
            for (int i = 0; i < h->n; ++i) t[(size_t)i * h->ld + b] = s[i];
        }
        NR_HIP(jg::sync_copy(dst, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        return 0;
    }
    const size_t need = (size_t)rows * h->n * sizeof(double);
    if (need > h->stage_bytes) {
        hipFree(h->d_stage); h->d_stage = nullptr; h->stage_bytes = 0;
        NR_HIP(hipMalloc((void**)&h->d_stage, need));
        h->stage_bytes = need;
    }
    NR_HIP(hipMemcpyAsync(h->d_stage, src, need, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_spread_bus, dim3((h->n + 63) / 64, h->ld / 64), dim3(64, 8), 0, h->stream, (const double*)h->d_stage, dst, h->n, h->ld, h->batch, rows);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

// pack the active scenarios into the leading lanes (restore = 1: send every lane back home)
void launch_compact(jg_nr* h, int restore, bool report = false) {
    CompactArgs c{h->d_active, h->d_iters, h->d_status, h->eng.status, h->d_lid, h->d_ppos, h->mp, h->d_dest, h->d_group,
                  h->d_glist, h->d_cflags, h->d_itmp, h->ld, restore, report ? h->h_counter_dev : nullptr, report && h->ld == 64 ? h->h_counter_dev + 1 : nullptr, h->d_params};
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, c);
    if (h->ld == 64) return;                       // one lane group: nothing to pack, lanes never leave their home order
    double* tmp = h->eng.X;                        // the factor is dead here (rebuilt by the next factorisation)
    const dim3 block(256), gy((unsigned)((h->ld + 255) / 256));
    {
        LaneSet ls{};
        int na = 0, max_rows = 0;
        long long off = 0;
        auto add = [&](double* x, int rows, int elem) {
            ls.x[na] = x; ls.rows[na] = rows; ls.elem[na] = elem; ls.off[na] = off;
            off += (long long)rows * elem * h->ld; max_rows = std::max(max_rows, rows); ++na;
        };
        add(h->d_vm, h->n, 1); add(h->d_va, h->n, 1); add(h->d_p, h->n, 1); add(h->d_q, h->n, 1);
        if (h->mp > 0) { add(h->d_pdg, h->mp, 1); add(h->d_pdb, h->mp, 1); }
        add(h->d_inc, h->n, 2);                    // a finished scenario keeps ITS last increment (method.increment) wherever its lane goes
        static const bool inplace_env = jg::knob("LANES_INPLACE", 1) != 0;
        if (h->ld <= LANES_INPLACE && inplace_env) {                      // one launch, in place (k_lanes_permute); JG_LANES_INPLACE=0: the two-pass move
            hipLaunchKernelGGL(k_lanes_permute, dim3((unsigned)std::min(max_rows, 2048), 1, (unsigned)na), dim3((unsigned)std::min(1024, h->ld)), 0, h->stream, ls, h->d_dest, h->d_cflags, h->ld);
            return;
        }
        if ((size_t)off * sizeof(double) <= h->eng.factor_bytes()) {      // always, except for grids of a handful of buses
            const dim3 grid((unsigned)std::min(max_rows, 1024), gy.x, (unsigned)na);
            hipLaunchKernelGGL(k_lanes_move, grid, block, 0, h->stream, ls, tmp, h->d_dest, h->d_cflags, h->ld, 0);
            hipLaunchKernelGGL(k_lanes_move, grid, block, 0, h->stream, ls, tmp, h->d_dest, h->d_cflags, h->ld, 1);
            return;
        }
    }
    auto permute = [&](double* x, int rows) {
        const dim3 grid((unsigned)std::min(rows, 2048), gy.x);
        hipLaunchKernelGGL(k_lane_permute<1>, grid, block, 0, h->stream, x, tmp, h->d_dest, h->d_cflags, rows, h->ld);
        hipLaunchKernelGGL(k_lane_copy<1>, grid, block, 0, h->stream, tmp, x, h->d_dest, h->d_cflags, rows, h->ld);
    };
    auto permute2 = [&](double* x, int rows) {                 // rows of interleaved 2-vectors
        const dim3 grid((unsigned)std::min(rows, 2048), gy.x);
        hipLaunchKernelGGL(k_lane_permute<2>, grid, block, 0, h->stream, x, tmp, h->d_dest, h->d_cflags, rows, h->ld);
        hipLaunchKernelGGL(k_lane_copy<2>, grid, block, 0, h->stream, tmp, x, h->d_dest, h->d_cflags, rows, h->ld);
    };
    permute(h->d_vm, h->n); permute(h->d_va, h->n); permute(h->d_p, h->n); permute(h->d_q, h->n);
    if (h->mp > 0) { permute(h->d_pdg, h->mp); permute(h->d_pdb, h->mp); }
    permute2(h->d_inc, h->n);
}

// solve! numerics on the groups of `sel`: factorise the Jacobian that is in place, solve, update the state (active: nullable)
int newton_step(jg_nr* h, const jg::GroupSel& sel, const int* active) {
    if (int rc = h->eng.factor(h->stream, nullptr, h->d_F, sel, h->level0_done)) return rc;
    if (!h->refine) {
        jg::StateUpdate upd{h->d_va, h->d_vm, h->d_flags, active, -1.0};
        return h->eng.backsolve(h->stream, h->d_inc, upd, sel);
    }
    // the plain solve goes to a scratch vector: method.increment (d_inc) is written for ACTIVE scenarios only, by k_refine_apply -- a
    // finished scenario of an active lane group keeps its last increment, as without refinement
    jg::StateUpdate none{};
    if (int rc = h->eng.backsolve(h->stream, h->d_inc2[1], none, sel)) return rc;
    RefineArgs a{h->d_rowptr, h->d_col, h->d_GB, h->d_rowtype, h->d_vm, h->d_va, h->d_ppos, h->d_pdg, h->d_pdb,
                 h->d_F, h->d_inc2[1], h->d_R, sel, h->n, h->ld, h->mp, h->nchunk, h->batch};
    dim3 grid(jg::grid_blocks(h->ld / 64, h->nchunk)), block(64, ASM_WAVES);
    switch (h->mp) {
        case 0: hipLaunchKernelGGL((k_refine_residual<0>), grid, block, 0, h->stream, a); break;
        case 4: hipLaunchKernelGGL((k_refine_residual<4>), grid, block, 0, h->stream, a); break;
        default: hipLaunchKernelGGL((k_refine_residual<8>), grid, block, 0, h->stream, a); break;
    }
    if (int rc = h->eng.forward(h->stream, h->d_R, sel)) return rc;
    if (int rc = h->eng.backsolve(h->stream, h->d_inc2[0], none, sel)) return rc;
    hipLaunchKernelGGL(k_refine_apply, dim3(jg::grid_blocks(h->ld / 64, (h->n + 15) / 16)), dim3(64, 16), 0, h->stream,
                       h->d_inc, h->d_inc2[1], h->d_inc2[0], h->d_va, h->d_vm, h->d_flags, active, sel, h->n, h->ld, h->batch);
    return 0;
}

int build_graphs(jg_nr* h) {
    if (h->execA) return 0;
    std::lock_guard<std::mutex> lk(jg::capture_mutex());
    NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    // One pass decides AND prepares: mismatch + Jacobian in one assembly (the Jacobian of a finished batch is the only waste:
    // one write stream per solve, against a second walk over Ybus per iteration), verdict per scenario, compaction of the
    // still-active lanes.  A compaction moves lanes (and stages them in the factor storage), so the assembly is repeated
    // on the packed lanes -- a launch that returns at once unless the compaction flag is set.
    auto check_and_compact = [&]() {
        if (h->ld == 64) {                             // one lane group: ONE launch (k_verdict64)
            CheckArgs c{h->d_part, h->nchunk, h->ld, h->batch, h->d_params, h->d_normp, h->d_normq, h->d_active, h->d_iters, h->d_status, h->eng.status, h->d_counter, h->d_group, 1};
            CompactArgs p{h->d_active, h->d_iters, h->d_status, h->eng.status, h->d_lid, h->d_ppos, h->mp, h->d_dest, h->d_group,
                          h->d_glist, h->d_cflags, h->d_itmp, h->ld, 0, h->h_counter_dev, h->h_counter_dev + 1, h->d_params};
            hipLaunchKernelGGL(k_verdict64, dim3(1), dim3(64, 16), 0, h->stream, c, p);
            return;
        }
        launch_check(h, 1, h->d_group);
        launch_compact(h, 0, true);                    // also reports the number of active scenarios to the host word
    };
    auto verdict = [&]() {
        launch_assemble(h, active_groups(h), true, nullptr, 0, nullptr, true);
        check_and_compact();
        if (h->ld > 64) launch_assemble(h, active_groups(h), true, nullptr, 0, h->d_cflags, true);
    };
    // graph A: the verdict on the start point
    verdict();
    NR_HIP(hipStreamEndCapture(h->stream, &h->graphA));
    NR_HIP(hipGraphInstantiate(&h->execA, h->graphA, nullptr, nullptr, 0));
    NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    // graph B: one iteration on the packed lanes (factorise the Jacobian that is already in place, solve, update), then the
    // verdict on the new state
    int rc = newton_step(h, active_groups(h), h->d_active);
    verdict();
    hipError_t e = hipStreamEndCapture(h->stream, &h->graphB);
    if (rc) return fail(rc, h->eng.error);
    NR_HIP(e);
    NR_HIP(hipGraphInstantiate(&h->execB, h->graphB, nullptr, nullptr, 0));
    // graph Bm: the same iteration with a verdict from a MISMATCH-ONLY pass -- for the iteration the handle expects to be the last of its run (run_loop): the
    // Jacobian a verdict assembles is wasted on every scenario that converges (nobody factorises it) and on every scenario that leaves for a pool (jg_nr_resume
    // assembles where they arrive), 0.87 GB of a 512-scenario batch against 0.26 GB for the mismatch pass.  graph J: the Jacobian alone, when the guess was wrong.
    NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    rc = newton_step(h, active_groups(h), h->d_active);
    launch_assemble(h, active_groups(h), false);
    check_and_compact();
    if (h->ld > 64) launch_assemble(h, active_groups(h), false, nullptr, 0, h->d_cflags);
    e = hipStreamEndCapture(h->stream, &h->graphBm);
    if (rc) return fail(rc, h->eng.error);
    NR_HIP(e);
    NR_HIP(hipGraphInstantiate(&h->execBm, h->graphBm, nullptr, nullptr, 0));
    NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    launch_assemble(h, active_groups(h), true, nullptr, 0, nullptr, true);
    NR_HIP(hipStreamEndCapture(h->stream, &h->graphJ));
    NR_HIP(hipGraphInstantiate(&h->execJ, h->graphJ, nullptr, nullptr, 0));
    return 0;
}

void drop_iteration_graphs(jg_nr* h) {
    for (hipGraphExec_t* e : {&h->execA, &h->execB, &h->execBm, &h->execJ, &h->execM}) if (*e) { hipGraphExecDestroy(*e); *e = nullptr; }
    for (hipGraph_t* g : {&h->graphA, &h->graphB, &h->graphBm, &h->graphJ, &h->graphM}) if (*g) { hipGraphDestroy(*g); *g = nullptr; }
    h->m_iters = 0;
}

// ONE scenario: the whole solve as ONE graph (round 6).  Every graph costs ~10 us on the device before its first kernel starts (the gap between two iteration graphs in
// the trace; launching the next graph ahead of the verdict did not close it), and a single instance launches one per iteration + one for the verdict on its start: 5 % of
// its solve.  A handle of one scenario that has solved before knows how many iterations to expect (stop_hint); its next solve is the start verdict + that many iterations
// in one graph -- only the LAST verdict reports to the host, from a mismatch-only pass as in graph Bm; an iteration behind the convergence runs as launches that return
// at once (the group list is empty).  Fewer iterations than expected: nothing to do; more: graph J supplies the Jacobian and run_loop carries on.
int build_whole_graph(jg_nr* h, int k) {
    if (h->execM && h->m_iters == k) return 0;
    if (h->execM) { hipGraphExecDestroy(h->execM); h->execM = nullptr; }
    if (h->graphM) { hipGraphDestroy(h->graphM); h->graphM = nullptr; }
    h->m_iters = 0;
    std::lock_guard<std::mutex> lk(jg::capture_mutex());
    NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    auto verdict64 = [&](bool report) {
        CheckArgs c{h->d_part, h->nchunk, h->ld, h->batch, h->d_params, h->d_normp, h->d_normq, h->d_active, h->d_iters, h->d_status, h->eng.status, h->d_counter, h->d_group, 1};
        CompactArgs p{h->d_active, h->d_iters, h->d_status, h->eng.status, h->d_lid, h->d_ppos, h->mp, h->d_dest, h->d_group,
                      h->d_glist, h->d_cflags, h->d_itmp, h->ld, 0, report ? h->h_counter_dev : nullptr, report ? h->h_counter_dev + 1 : nullptr, h->d_params};
        hipLaunchKernelGGL(k_verdict64, dim3(1), dim3(64, 16), 0, h->stream, c, p);
    };
    launch_assemble(h, active_groups(h), true, nullptr, 0, nullptr, true);       // the verdict on the start point (graph A)
    verdict64(false);
    int rc = 0;
    for (int it = 1; it <= k && !rc; ++it) {
        rc = newton_step(h, active_groups(h), h->d_active);
        if (it < k) { launch_assemble(h, active_groups(h), true, nullptr, 0, nullptr, true); verdict64(false); }      // graph B
        else { launch_assemble(h, active_groups(h), false); verdict64(true); }                                         // graph Bm: the expected last iteration
    }
    hipError_t e = hipStreamEndCapture(h->stream, &h->graphM);
    if (rc) return fail(rc, h->eng.error);
    NR_HIP(e);
    NR_HIP(hipGraphInstantiate(&h->execM, h->graphM, nullptr, nullptr, 0));
    h->m_iters = k;
    return 0;
}

void drop_comp_graphs(jg_nr* h) {
    if (h->execA2) { hipGraphExecDestroy(h->execA2); h->execA2 = nullptr; }
    if (h->graphA2) { hipGraphDestroy(h->graphA2); h->graphA2 = nullptr; }
    if (h->execC) { hipGraphExecDestroy(h->execC); h->execC = nullptr; }
    if (h->graphC) { hipGraphDestroy(h->graphC); h->graphC = nullptr; }
}

void release_base(jg_nr_base* b) {
    if (!b) return;
    if (b->refs.fetch_sub(1) == 1) {
        hipSetDevice(b->device);
        b->cb.destroy();
        if (b->stream) hipStreamDestroy(b->stream);
        delete b;
    }
}

void detach_base(jg_nr* h) {
    if (!h->base) return;
    if (h->stream) hipStreamSynchronize(h->stream);
    drop_comp_graphs(h);
    hipFree(h->d_cw); h->d_cw = nullptr;
    release_base(h->base);
    h->base = nullptr;
    h->start_is_base = false;
}

// The two graphs of a compensated start (jg_comp.hpp).  A2: the verdict on the start point from a MISMATCH-ONLY pass (no Jacobian is assembled: nobody
// factorises it).  C: the first iteration -- per-scenario correction of the right-hand side, ONE sweep pair on the base's factor with the state update
// fused in, then the usual verdict on the new state (which assembles the Jacobian the second iteration factorises).
int build_comp_graphs(jg_nr* h) {
    if (h->execC) return 0;
    const jg::CompBase& cb = h->base->cb;
    std::lock_guard<std::mutex> lk(jg::capture_mutex());
    NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    launch_assemble(h, active_groups(h), false);
    launch_check(h, 1, h->d_group);
    launch_compact(h, 0, true);
    if (h->ld > 64) launch_assemble(h, active_groups(h), false, nullptr, 0, h->d_cflags);      // lanes moved: their mismatch rows again
    NR_HIP(hipStreamEndCapture(h->stream, &h->graphA2));
    NR_HIP(hipGraphInstantiate(&h->execA2, h->graphA2, nullptr, nullptr, 0));
    NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    jg::CompFixArgs fa{h->d_ppos, h->d_pdg, h->d_pdb, h->mp, cb.rowptr, cb.colm, cb.posrow, cb.rowtype, cb.v0, cb.th0, cb.y0, cb.Zc,
                       h->d_F, h->eng.status, h->d_active, h->ld, h->batch};
    if (h->mp > 0) jg::launch_comp_fix(fa, h->stream);
    jg::StateUpdate upd{h->d_va, h->d_vm, h->d_flags, h->d_active, -1.0};
    const int rc = cb.solve(cb.split, h->stream, h->d_F, h->d_cw, h->d_inc, h->ld, h->batch, upd, active_groups(h));
    launch_assemble(h, active_groups(h), true, nullptr, 0, nullptr, true);
    launch_check(h, 1, h->d_group);
    launch_compact(h, 0, true);
    if (h->ld > 64) launch_assemble(h, active_groups(h), true, nullptr, 0, h->d_cflags, true);
    hipError_t e = hipStreamEndCapture(h->stream, &h->graphC);
    if (rc) return fail(rc, "the shared-factor sweep could not be captured");
    NR_HIP(e);
    NR_HIP(hipGraphInstantiate(&h->execC, h->graphC, nullptr, nullptr, 0));
    return 0;
}

// mismatching injections between the lanes and the base (one word)
__global__ void k_cmp_injection(const double* p, const double* q, const double* p0, const double* q0, int n, int ld, int batch, int* count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    int bad = 0;
    for (int i = blockIdx.y; i < n; i += gridDim.y) bad += (p[(size_t)i * ld + b] != p0[i]) || (q[(size_t)i * ld + b] != q0[i]);
    if (bad) atomicAdd(count, bad);
}

// Does the next run start with the compensated iteration?  (every condition the algebra of jg_comp.hpp rests on)
int comp_ready(jg_nr* h, bool& ready) {
    ready = false;
    if (!h->base || !h->first_mode || !h->start_is_base || h->refine || h->fast) return 0;
    bool patched = false;
    for (unsigned char c : h->patch_state) { if (c == 2) return 0; patched |= c == 1; }
    if (patched) {                                               // an outage moves the mismatch of its two buses only if the injections are the base's
        if (!h->inj_checked) {
            NR_HIP(hipMemsetAsync(h->d_cmp, 0, sizeof(int), h->stream));
            hipLaunchKernelGGL(k_cmp_injection, dim3((h->batch + 255) / 256, 64), dim3(256), 0, h->stream, h->d_p, h->d_q, h->base->cb.p0, h->base->cb.q0, h->n, h->ld, h->batch, h->d_cmp);
            int bad = 0;
            NR_HIP(jg::sync_copy(&bad, h->d_cmp, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            h->inj_checked = true; h->inj_equal = bad == 0;
        }
        if (!h->inj_equal) return 0;
    }
    if (int rc = build_comp_graphs(h)) return rc;
    ready = true;
    return 0;
}

}  // namespace

extern "C" {

const char* jg_last_error(void) { return g_error.c_str(); }

void jg_plan_cache_clear(void) { jg::clear_plan_cache(); }

int jg_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return -1;
    return c;
}

int jg_nr_create(jg_nr** out, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* y_reim,
                 const double* yt_reim, const int8_t* type, int64_t slack, int64_t batch, int64_t max_patch,
                 int device) {
    if (batch > 65472) return fail(1, "jg_nr_create: at most 65 472 scenarios per handle (lane bookkeeping of the compaction: 16-bit counts)");
    if (!out || n < 1 || !colptr || !rowval || !y_reim || !yt_reim || !type || batch < 1 || max_patch < 0 || max_patch > 8)
        return fail(1, "jg_nr_create: bad argument");
    if (slack < 1 || slack > n || type[slack - 1] != 3) return fail(1, "The slack bus is missing.");
    int ndev = 0;
    NR_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(1, "jg_nr_create: no such HIP device");
    const double tc0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    jg_nr* h = new jg_nr();
    h->n = (int)n; h->batch = (int)batch; h->ld = (int)((batch + 63) / 64 * 64);
    h->mp = max_patch == 0 ? 0 : (max_patch <= 4 ? 4 : 8);
    h->device = device; h->slack = slack;
    h->nnz = (int)(colptr[n] - 1);
    h->colptr.assign(colptr, colptr + n + 1);
    h->rowval.assign(rowval, rowval + h->nnz);
    h->type.assign(type, type + n);
    const int nnz = h->nnz;
    if (n >= (1 << 24)) { delete h; return fail(1, "jg_nr_create: more than 2^24 buses"); }
    for (int c = 0; c < n; ++c) {
        if (colptr[c + 1] < colptr[c] || colptr[c] < 1 || colptr[c + 1] - 1 > nnz) { delete h; return fail(1, "jg_nr_create: malformed column pointers"); }
        for (int64_t p = colptr[c] - 1; p < colptr[c + 1] - 1; ++p)
            if (rowval[p] < 1 || rowval[p] > n) { delete h; return fail(1, "jg_nr_create: row index out of range"); }
    }
    // The symbolic analysis of the block LU (or its look-up in the plan cache), the upload of its tables and the factor storage need the pattern
    // and nothing else: they run on a thread of their own while this one builds the reference's maps, uploads the model and allocates the state
    // (round 4: what a first power flow on a grid waits for is the analysis -- 19 of 30 ms on the 10 000-bus grid -- not analysis + the rest).
    int rc = set_device(h);
    if (rc) { delete h; return rc; }
    std::vector<int> rp(n + 1), cl(nnz);
    for (int i = 0; i <= n; ++i) rp[i] = (int)(colptr[i] - 1);
    for (int p = 0; p < nnz; ++p) cl[p] = (int)(rowval[p] - 1);
    // in place: the assembly kernel writes into the factor storage; bit 2: it also finishes the plan's level 0 (jg_symbolic.hpp: prefactor)
    // bit 49: Jordan rows for the pivots of the top tasks (jg_symbolic.hpp): the backward sweep over the top of the tree is a handful of
    // plain levels instead of sequential chains (refined steps and fast Newton-Raphson switch the engine back: Engine::jordan)
    h->eng.lanes = h->batch;                                    // (known before the plan is chosen: a handful of scenarios gets a deeper top)
    int eng_rc = 0;
    std::atomic<int> stream_state{0};                           // 1: h->stream exists, -1: its creation failed (the analysis thread creates it first)
    auto eng_work = [&] {
        if (hipSetDevice(h->device) != hipSuccess) { h->eng.error = "hipSetDevice failed on the analysis thread"; eng_rc = 2; stream_state = -1; return; }
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->eng.error = "jg_nr_create: stream creation failed"; eng_rc = 2; stream_state = -1; return; }
        stream_state = 1;
        eng_rc = h->eng.create((int)n, rp.data(), cl.data(), h->ld, (jg::knob_set("NO_PREFACTOR") ? 1LL : 1LL | 4) | 1LL << 49, h->stream);
    };
    std::thread eng_thread;
    try { eng_thread = std::thread(eng_work); } catch (const std::system_error&) { eng_work(); }     // no thread to be had: the analysis first, then the rest
    struct JoinEng { std::thread& t; ~JoinEng() { if (t.joinable()) t.join(); } } join_eng{eng_thread};     // every early return waits for it before the handle goes
    h->csr_col = cl;
    h->csr_row.assign(nnz, 0);
    for (int i = 0; i < n; ++i) for (int p = rp[i]; p < rp[i + 1]; ++p) h->csr_row[p] = i;
    h->patch_state.assign((size_t)batch, 0);
    // transpose permutation of the (structurally symmetric) pattern: tperm[p of (r,c)] = pointer of (c,r)
    h->tperm.assign(nnz, -1);
    for (int c = 0; c < n; ++c)
        for (int64_t p = colptr[c] - 1; p < colptr[c + 1] - 1; ++p) {
            const int64_t r = rowval[p] - 1;
            int64_t lo = colptr[r] - 1, hi = colptr[r + 1] - 2, q = -1;
            while (lo <= hi) { int64_t m = (lo + hi) >> 1; if (rowval[m] - 1 < c) lo = m + 1; else if (rowval[m] - 1 > c) hi = m - 1; else { q = m; break; } }
            if (q < 0) { if (eng_thread.joinable()) eng_thread.join(); jg_nr_destroy(h); return fail(1, "jg_nr_create: Ybus pattern is not structurally symmetric"); }
            h->tperm[p] = (int)q;
        }
    // ---- newtonJacobian (acPowerFlow.jl:89-175), integer only -------------------------------
    h->pq.assign(n, 0); h->pvpq.assign(n, 0); h->pcount.assign(n, 0);
    int64_t pvpqNum = 0, pqNum = 0;
    for (int i = 0; i < n; ++i) {
        if (type[i] == 1) { pqNum++; h->pq[i] = pqNum + n - 1; }
        if (type[i] != 3) { pvpqNum++; h->pvpq[i] = pvpqNum; }
    }
    h->dimJ = n + pqNum - 1;
    std::vector<int64_t> colcount(h->dimJ, 0), qcount(n, 0);
    for (int i = 0; i < n; ++i) {
        if (i + 1 == slack) continue;
        for (int64_t p = colptr[i] - 1; p < colptr[i + 1] - 1; ++p) {
            const int8_t tr = type[rowval[p] - 1];
            if (tr != 3) h->pcount[i]++;
            if (tr == 1) qcount[i]++;
        }
        colcount[h->pvpq[i] - 1] = h->pcount[i] + qcount[i];
        if (type[i] == 1) colcount[h->pq[i] - 1] = h->pcount[i] + qcount[i];
    }
    h->jcolptr.assign(h->dimJ + 1, 1);
    for (int64_t c = 0; c < h->dimJ; ++c) h->jcolptr[c + 1] = h->jcolptr[c] + colcount[c];
    h->nnzJ = h->jcolptr[h->dimJ] - 1;
    h->jrowval.assign(h->nnzJ, 0);
    h->jmap.assign(h->nnzJ, 0);
    for (int i = 0; i < n; ++i) {
        if (i + 1 == slack) continue;
        const bool isPQ = type[i] == 1;
        int64_t pA = h->jcolptr[h->pvpq[i] - 1], qA = pA + h->pcount[i];
        int64_t pM = isPQ ? h->jcolptr[h->pq[i] - 1] : 0, qM = isPQ ? pM + h->pcount[i] : 0;
        for (int64_t p = colptr[i] - 1; p < colptr[i + 1] - 1; ++p) {
            const int64_t row = rowval[p] - 1;
            const int8_t tr = type[row];
            const int64_t blk = (int64_t)h->tperm[p] * 4;      // block (row, i) in row-CSR order
            if (tr != 3) {
                h->jrowval[pA - 1] = h->pvpq[row]; h->jmap[pA - 1] = blk + 0; pA++;
                if (isPQ) { h->jrowval[pM - 1] = h->pvpq[row]; h->jmap[pM - 1] = blk + 1; pM++; }
            }
            if (tr == 1) {
                h->jrowval[qA - 1] = h->pq[row]; h->jmap[qA - 1] = blk + 2; qA++;
                if (isPQ) { h->jrowval[qM - 1] = h->pq[row]; h->jmap[qM - 1] = blk + 3; qM++; }
            }
        }
    }
    // ---- device upload ----------------------------------------------------------------------
    while (stream_state.load() == 0) std::this_thread::yield();
    if (stream_state.load() < 0) { if (eng_thread.joinable()) eng_thread.join(); std::string m = h->eng.error; jg_nr_destroy(h); return fail(2, m); }
    if (jg::knob_set("PLAN_TIMING")) fprintf(stderr, "[jg nr create] reference maps done at                  %6.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - tc0);
    std::vector<double> G(nnz), B(nnz);
    for (int p = 0; p < nnz; ++p) { G[p] = yt_reim[2 * p]; B[p] = yt_reim[2 * p + 1]; }
    std::vector<int> colm(nnz);                 // column | existence mask of the 2x2 block entries (row i, col j types)
    for (int i = 0; i < n; ++i)
        for (int p = rp[i]; p < rp[i + 1]; ++p) {
            const int j = cl[p];
            const int rpm = type[i] != 3, rq = type[i] == 1, ct = type[j] != 3, cv = type[j] == 1;
            colm[p] = j | (((rpm & ct) | ((rpm & cv) << 1) | ((rq & ct) << 2) | ((rq & cv) << 3)) << 24);
        }
    std::vector<double2> GBv(nnz);
    for (int p = 0; p < nnz; ++p) GBv[p] = double2{G[p], B[p]};
    // consistency of the two value arrays the reference keeps (T1): yT[p] must equal y[tperm[p]]
    for (int p = 0; p < nnz; ++p)
        if (y_reim[2 * (size_t)h->tperm[p]] != yt_reim[2 * p] || y_reim[2 * (size_t)h->tperm[p] + 1] != yt_reim[2 * p + 1]) {
            if (eng_thread.joinable()) eng_thread.join(); jg_nr_destroy(h); return fail(4, "jg_nr_create: nodalMatrix and nodalMatrixTranspose disagree (stale model)");
        }
    std::vector<signed char> flags(n);
    for (int i = 0; i < n; ++i) flags[i] = (signed char)((type[i] != 3 ? 1 : 0) | (type[i] == 1 ? 2 : 0));
    std::string err;
    std::vector<signed char> tp(type, type + n);
    if (jg::upload(&h->d_rowptr, rp, err, h->stream) || jg::upload(&h->d_col, colm, err, h->stream) || jg::upload(&h->d_G, G, err, h->stream) ||
        jg::upload(&h->d_B, B, err, h->stream) || jg::upload(&h->d_GB, GBv, err, h->stream) || jg::upload(&h->d_rowtype, std::vector<int>(type, type + n), err, h->stream) || jg::upload(&h->d_type, tp, err, h->stream) || jg::upload(&h->d_flags, flags, err, h->stream)) {
        if (eng_thread.joinable()) eng_thread.join(); jg_nr_destroy(h); return fail(2, err);
    }
    h->nchunk = (h->n + ASM_ROWS - 1) / ASM_ROWS;
    const size_t ld = h->ld;
    // the per-handle state as ONE allocation and ONE fill (round 4: 23 x (hipMalloc + fill + stream synchronisation) were 2 of the 4 ms a handle on a
    // cached plan costs); every array starts on a 256-byte boundary
    const size_t mpn = h->mp > 0 ? h->mp : 1;
    struct Part { void** p; size_t bytes; };
    const Part parts[] = {
        {(void**)&h->d_vm, n * ld * 8}, {(void**)&h->d_va, n * ld * 8}, {(void**)&h->d_p, n * ld * 8}, {(void**)&h->d_q, n * ld * 8},
        {(void**)&h->d_ppos, mpn * ld * 4}, {(void**)&h->d_pdg, mpn * ld * 8}, {(void**)&h->d_pdb, mpn * ld * 8},
        {(void**)&h->d_F, n * 2 * ld * 8}, {(void**)&h->d_inc, n * 2 * ld * 8}, {(void**)&h->d_part, (size_t)h->nchunk * 2 * ld * 8},
        {(void**)&h->d_normp, ld * 8}, {(void**)&h->d_normq, ld * 8}, {(void**)&h->d_params, 4 * 8}, {(void**)&h->d_active, ld * 4},
        {(void**)&h->d_iters, ld * 4}, {(void**)&h->d_status, ld * 4}, {(void**)&h->d_counter, 4}, {(void**)&h->d_group, (ld / 64) * 4},
        {(void**)&h->d_lid, ld * 4}, {(void**)&h->d_dest, ld * 4}, {(void**)&h->d_cflags, 32}, {(void**)&h->d_itmp, ld * 4},
        {(void**)&h->d_glist, std::max<size_t>(ld / 64, 8) * 4}};    // map_block reads the first eight entries in one scalar load
    size_t arena_bytes = 0;
    for (const Part& q : parts) arena_bytes += (q.bytes + 255) / 256 * 256;
    bool ok = hipMalloc((void**)&h->d_arena, arena_bytes) == hipSuccess && jg::sync_fill(h->d_arena, 0, arena_bytes, h->stream) == hipSuccess;
    if (ok) {
        size_t off = 0;
        for (const Part& q : parts) { *q.p = (char*)h->d_arena + off; off += (q.bytes + 255) / 256 * 256; }
    }
    if (!ok) { if (eng_thread.joinable()) eng_thread.join(); jg_nr_destroy(h); return fail(2, "jg_nr_create: device allocation failed"); }
    if (jg::sync_fill(h->d_ppos, 0xff, mpn * ld * 4, h->stream) != hipSuccess ||     // -1 = no patch
        hipHostMalloc((void**)&h->h_counter, 129 * sizeof(int)) != hipSuccess) {       // the verdict word + iterations | status of one lane group (k_compact: host_res)
        if (eng_thread.joinable()) eng_thread.join(); jg_nr_destroy(h); return fail(2, "jg_nr_create: pinned allocation failed");
    }
    if (hipHostGetDevicePointer((void**)&h->h_counter_dev, h->h_counter, 0) != hipSuccess) { if (eng_thread.joinable()) eng_thread.join(); jg_nr_destroy(h); return fail(2, "jg_nr_create: pinned host word is not device-visible"); }
    const bool timing = jg::knob_set("PLAN_TIMING");
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    if (timing) fprintf(stderr, "[jg nr create] maps, model upload, state arena done at %6.1f ms\n", tnow() - tc0);
    if (eng_thread.joinable()) eng_thread.join();
    if (timing) fprintf(stderr, "[jg nr create] engine joined at                        %6.1f ms\n", tnow() - tc0);
    if (eng_rc) { std::string m = h->eng.error; jg_nr_destroy(h); return fail(eng_rc, m); }
    if (jg::upload(&h->d_dst, h->eng.plan->S.src_entry, err, h->stream)) { jg_nr_destroy(h); return fail(2, err); }
    if (h->eng.plan->S.prefactor) {                            // the row table of the assemblies that also finish the plan's level 0
        std::vector<int> rt(type, type + n);
        for (int k = 0; k < (int)n; ++k) if (h->eng.plan->S.pre_pivot[k]) rt[h->eng.plan->S.perm[k]] |= (k + 1) << 2;
        if (jg::upload(&h->d_rowtype_pre, rt, err, h->stream)) { jg_nr_destroy(h); return fail(2, err); }
    }
    h->eng.lanes = h->batch;
    for (int64_t k = 0; k < h->nnzJ; ++k) h->jmap[k] = (int64_t)h->eng.plan->S.src_entry[h->jmap[k] >> 2] * 4 + (h->jmap[k] & 3);
    if (timing) fprintf(stderr, "[jg nr create] done at                                 %6.1f ms\n", tnow() - tc0);
    *out = h;
    return 0;
}

void jg_nr_destroy(jg_nr* h) {
    if (h && jg::knob_set("HOST_TIMING") && h->host_iters > 0)
        fprintf(stderr, "[jg host timing] %d lanes: %lld iterations, hipGraphLaunch %.1f us, wait for the verdict %.1f us per iteration; last iteration guessed right %lld times, wrong %lld\n", h->ld, h->host_iters,
                h->host_launch_us / h->host_iters, h->host_wait_us / h->host_iters, h->last_guess_hit, h->last_guess_miss);
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    detach_base(h);
    hipFree(h->d_cmp);
    if (h->d_move) { hipFree(h->d_move); hipHostFree(h->h_move); }
    drop_iteration_graphs(h);
    h->eng.destroy();
    hipFree(h->d_stage);
    hipFree(h->d_R); hipFree(h->d_inc2[0]); hipFree(h->d_inc2[1]);
    hipFree(h->d_fp_entry); hipFree(h->d_fp_dp); hipFree(h->d_fp_dq);
    if (h->execFA) hipGraphExecDestroy(h->execFA);
    if (h->execFB) hipGraphExecDestroy(h->execFB);
    if (h->graphFA) hipGraphDestroy(h->graphFA);
    if (h->graphFB) hipGraphDestroy(h->graphFB);
    hipFree(h->d_bfrom); hipFree(h->d_bto); hipFree(h->d_bstatus); hipFree(h->d_bparam); hipFree(h->d_outage); hipFree(h->d_post);
    hipFree(h->d_rating); hipFree(h->d_screen); hipFree(h->d_screc);
    hipFree(h->d_rowptr); hipFree(h->d_col); hipFree(h->d_G); hipFree(h->d_B); hipFree(h->d_GB); hipFree(h->d_rowtype); hipFree(h->d_rowtype_pre); hipFree(h->d_type); hipFree(h->d_flags);
    hipFree(h->d_arena);                                         // V, theta, P, Q, patches, mismatch, increment, norms, lane bookkeeping: one allocation (jg_nr_create)
    hipFree(h->d_dst);
    hipFree(h->d_vm0); hipFree(h->d_va0);
    if (h->h_counter) hipHostFree(h->h_counter);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int jg_nr_dims(jg_nr* h, int64_t* dims) {
    if (!h || !dims) return fail(1, "jg_nr_dims: bad argument");
    dims[0] = h->dimJ; dims[1] = h->nnzJ; dims[2] = h->eng.plan->S.n_entries; dims[3] = h->eng.plan->S.n_sched_terms;
    dims[4] = (int64_t)(h->eng.fact.size() + h->eng.plan->S.top_launch.size());      // dependent launches of one factorisation
    dims[5] = (int64_t)(h->eng.jordan ? h->eng.bwdj.size() : h->eng.bwd.size());     // ... of the backward sweep the handle runs (Jordan rows or plain)
    return 0;
}

int jg_nr_set_injection(jg_nr* h, const double* p, const double* q, int64_t stride) {
    if (!h || !p || !q || stride < 0) return fail(1, "jg_nr_set_injection: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    if (int rc = put_bus_array(h, h->d_p, p, stride)) return rc;
    if (int rc = put_bus_array(h, h->d_q, q, stride)) return rc;
    h->jac_valid = false;
    h->inj_checked = false;                                      // compared with the base's again before the next compensated start
    return 0;
}

int jg_nr_set_voltage(jg_nr* h, const double* vm, const double* va, int64_t stride) {
    if (!h || !vm || !va || stride < 0) return fail(1, "jg_nr_set_voltage: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    if (int rc = put_bus_array(h, h->d_vm, vm, stride)) return rc;
    if (int rc = put_bus_array(h, h->d_va, va, stride)) return rc;
    h->jac_valid = false;
    h->start_is_base = false;
    return 0;
}

// device [n][ld] -> device [batch][n] (64 x 64 tiles: contiguous on both sides)
__global__ __launch_bounds__(512) void k_gather_bus(const double* src, double* dst, int n, int ld, int batch) {
    __shared__ double tile[64][65];
    const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int i = i0 + r, b = b0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && b < ld) ? src[(size_t)i * ld + b] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 64; r += blockDim.y) {
        const int b = b0 + r, i = i0 + threadIdx.x;
        if (b < batch && i < n) dst[(size_t)b * n + i] = tile[threadIdx.x][r];
    }
}

static int get_bus_array(jg_nr* h, const double* src, double* dst, int comps) {
    // device [n][ld][comps] -> host [batch][n*comps]
    if (comps == 1) {                                            // transposed on the device, then ONE copy of exactly the rows asked for
        const size_t need = (size_t)h->batch * h->n * sizeof(double);
        if (need > h->stage_bytes) {
            hipFree(h->d_stage); h->d_stage = nullptr; h->stage_bytes = 0;
            NR_HIP(hipMalloc((void**)&h->d_stage, need));
            h->stage_bytes = need;
        }
        hipLaunchKernelGGL(k_gather_bus, dim3((h->n + 63) / 64, h->ld / 64), dim3(64, 8), 0, h->stream, src, h->d_stage, h->n, h->ld, h->batch);
        NR_HIP(hipGetLastError());
        NR_HIP(jg::sync_copy(dst, h->d_stage, need, hipMemcpyDeviceToHost, h->stream));
        return 0;
    }
    const size_t rows = (size_t)h->n * comps;
    std::vector<double> t(rows * h->ld);
    NR_HIP(jg::sync_copy(t.data(), src, t.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b)
        for (size_t i = 0; i < (size_t)h->n; ++i)
            for (int c = 0; c < comps; ++c) dst[(size_t)b * rows + i * comps + c] = t[(i * h->ld + b) * comps + c];
    return 0;
}

int jg_nr_get_voltage(jg_nr* h, double* vm, double* va) {
    if (!h || !vm || !va) return fail(1, "jg_nr_get_voltage: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    if (int rc = get_bus_array(h, h->d_vm, vm, 1)) return rc;
    return get_bus_array(h, h->d_va, va, 1);
}

int jg_nr_snapshot_voltage(jg_nr* h) {
    if (!h) return fail(1, "jg_nr_snapshot_voltage: bad argument");
    if (int rc = set_device(h)) return rc;
    const size_t bytes = (size_t)h->n * h->ld * 8;
    if (!h->d_vm0) { NR_HIP(hipMalloc((void**)&h->d_vm0, bytes)); NR_HIP(hipMalloc((void**)&h->d_va0, bytes)); }
    NR_HIP(hipMemcpyAsync(h->d_vm0, h->d_vm, bytes, hipMemcpyDeviceToDevice, h->stream));
    NR_HIP(hipMemcpyAsync(h->d_va0, h->d_va, bytes, hipMemcpyDeviceToDevice, h->stream));
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_nr_restore_voltage(jg_nr* h) {
    if (!h || !h->d_vm0) return fail(1, "jg_nr_restore_voltage: no snapshot");
    if (int rc = set_device(h)) return rc;
    const size_t bytes = (size_t)h->n * h->ld * 8;
    NR_HIP(hipMemcpyAsync(h->d_vm, h->d_vm0, bytes, hipMemcpyDeviceToDevice, h->stream));
    NR_HIP(hipMemcpyAsync(h->d_va, h->d_va0, bytes, hipMemcpyDeviceToDevice, h->stream));
    h->jac_valid = false;
    h->start_is_base = false;
    return 0;
}

int jg_nr_get_voltage_device(jg_nr* h, double* vm_dev, double* va_dev) {
    if (!h || !vm_dev || !va_dev) return fail(1, "jg_nr_get_voltage_device: bad argument");
    if (int rc = set_device(h)) return rc;
    dim3 grid((h->n + 31) / 32, (h->ld + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(k_to_scenario_major, grid, block, 0, h->stream, h->d_vm, vm_dev, h->n, h->ld, h->batch);
    hipLaunchKernelGGL(k_to_scenario_major, grid, block, 0, h->stream, h->d_va, va_dev, h->n, h->ld, h->batch);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_nr_pack_results_device(jg_nr* h, double* dst_dev) {
    if (!h || !dst_dev) return fail(1, "jg_nr_pack_results_device: bad argument");
    if (int rc = set_device(h)) return rc;
    const long long stride = 2LL * h->n + 2;
    dim3 grid((h->n + 63) / 64, (h->ld + 63) / 64, 2), block(64, 8);
    hipLaunchKernelGGL(k_pack_bus, grid, block, 0, h->stream, h->d_vm, h->d_va, dst_dev, h->n, h->ld, h->batch, stride);
    hipLaunchKernelGGL(k_pack_tail, dim3((h->batch + 255) / 256), dim3(256), 0, h->stream, h->d_iters, h->d_status, dst_dev, h->batch, stride, 2 * h->n);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_nr_allgather_results(jg_nr* h, jg_comm* c, double* dst_dev) {
    if (!h || !c || !dst_dev) return fail(1, "jg_nr_allgather_results: bad argument");
    if (jg::comm_device(c) != h->device) return fail(1, "jg_nr_allgather_results: communicator and handle live on different devices");
    if (int rc = set_device(h)) return rc;
    const long long stride = 2LL * h->n + 2;
    const size_t count = (size_t)h->batch * stride;
    double* mine = dst_dev + (size_t)jg::comm_rank(c) * count;   // in-place all-gather: this rank's record sits in its own block
    dim3 grid((h->n + 63) / 64, (h->ld + 63) / 64, 2), block(64, 8);
    hipLaunchKernelGGL(k_pack_bus, grid, block, 0, h->stream, h->d_vm, h->d_va, mine, h->n, h->ld, h->batch, stride);
    hipLaunchKernelGGL(k_pack_tail, dim3((h->batch + 255) / 256), dim3(256), 0, h->stream, h->d_iters, h->d_status, mine, h->batch, stride, 2 * h->n);
    NR_HIP(hipGetLastError());
    if (int rc = jg::comm_allgather(c, mine, dst_dev, count, h->stream)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_nr_patch_ybus(jg_nr* h, int64_t scenario, int64_t k, const int64_t* ptr, const double* dy) {
    if (!h || scenario < 0 || scenario >= h->batch || k < 0 || k > h->mp || (k > 0 && (!ptr || !dy)))
        return fail(1, "jg_nr_patch_ybus: bad argument (scenario / entry count beyond max_patch)");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    int allpos[8];
    for (int m = 0; m < 8; ++m) allpos[m] = -1;
    for (int m = 0; m < h->mp; ++m) {
        int pos = -1; double g = 0.0, b = 0.0;
        if (m < k) {
            if (ptr[m] < 1 || ptr[m] > h->nnz) return fail(1, "jg_nr_patch_ybus: pointer out of range");
            pos = h->tperm[ptr[m] - 1]; g = dy[2 * m]; b = dy[2 * m + 1];
            for (int mm = 0; mm < m; ++mm) if (ptr[mm] == ptr[m]) return fail(1, "jg_nr_patch_ybus: duplicate pointer");
        }
        allpos[m] = pos;
        const size_t off = (size_t)m * h->ld + scenario;
        NR_HIP(jg::sync_copy(h->d_ppos + off, &pos, sizeof(int), hipMemcpyHostToDevice, h->stream));
        NR_HIP(jg::sync_copy(h->d_pdg + off, &g, sizeof(double), hipMemcpyHostToDevice, h->stream));
        NR_HIP(jg::sync_copy(h->d_pdb + off, &b, sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    h->patch_state[(size_t)scenario] = classify_patch(h, allpos, h->mp);
    h->jac_valid = false;
    return 0;
}

int jg_nr_patch_ybus_batch(jg_nr* h, int64_t scenario0, int64_t count, int64_t k, const int64_t* ptr, const double* dy) {
    if (!h || scenario0 < 0 || count < 1 || scenario0 + count > h->batch || k < 0 || k > h->mp || (k > 0 && (!ptr || !dy)))
        return fail(1, "jg_nr_patch_ybus_batch: bad argument (scenario range / entry count beyond max_patch)");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    // A whole batch (what a screen uploads per job): the three slot arrays travel as ONE copy each -- [mp][ld] is contiguous -- instead of 3 x mp
    // copies with a stream synchronisation apiece (a job of the pipeline pays this inside the timed region)
    const bool whole = scenario0 == 0 && count == h->batch;
    const size_t wld = whole ? (size_t)h->ld : (size_t)count;
    std::vector<int> pos((size_t)h->mp * wld, -1), allpos((size_t)count * 8, -1);
    std::vector<double> g((size_t)h->mp * wld, 0.0), b((size_t)h->mp * wld, 0.0);
    for (int m = 0; m < h->mp; ++m) {
        int* pm = pos.data() + (size_t)m * wld; double* gm = g.data() + (size_t)m * wld; double* bm = b.data() + (size_t)m * wld;
        for (int64_t s = 0; s < count; ++s) {
            if (m < k && ptr[s * k + m] != 0) {
                const int64_t p = ptr[s * k + m];
                if (p < 1 || p > h->nnz) return fail(1, "jg_nr_patch_ybus_batch: pointer out of range");
                for (int mm = 0; mm < m; ++mm) if (ptr[s * k + mm] == p) return fail(1, "jg_nr_patch_ybus_batch: duplicate pointer");
                pm[s] = h->tperm[p - 1]; gm[s] = dy[2 * (s * k + m)]; bm[s] = dy[2 * (s * k + m) + 1];
            }
            allpos[(size_t)s * 8 + m] = pm[s];
        }
    }
    if (whole) {
        NR_HIP(hipMemcpyAsync(h->d_ppos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        NR_HIP(hipMemcpyAsync(h->d_pdg, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        NR_HIP(hipMemcpyAsync(h->d_pdb, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        NR_HIP(hipStreamSynchronize(h->stream));
    } else
    for (int m = 0; m < h->mp; ++m) {
        const size_t off = (size_t)m * h->ld + scenario0;
        NR_HIP(jg::sync_copy(h->d_ppos + off, pos.data() + (size_t)m * wld, (size_t)count * sizeof(int), hipMemcpyHostToDevice, h->stream));
        NR_HIP(jg::sync_copy(h->d_pdg + off, g.data() + (size_t)m * wld, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
        NR_HIP(jg::sync_copy(h->d_pdb + off, b.data() + (size_t)m * wld, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    for (int64_t s = 0; s < count; ++s) h->patch_state[(size_t)(scenario0 + s)] = classify_patch(h, &allpos[(size_t)s * 8], h->mp);
    h->jac_valid = false;
    return 0;
}

int jg_nr_set_ybus(jg_nr* h, const double* y_reim, const double* yt_reim) {
    if (!h || !y_reim || !yt_reim) return fail(1, "jg_nr_set_ybus: bad argument");
    if (int rc = set_device(h)) return rc;
    std::vector<double> G(h->nnz), B(h->nnz);
    for (int p = 0; p < h->nnz; ++p) {
        G[p] = yt_reim[2 * p]; B[p] = yt_reim[2 * p + 1];
        if (y_reim[2 * (size_t)h->tperm[p]] != G[p] || y_reim[2 * (size_t)h->tperm[p] + 1] != B[p])
            return fail(4, "jg_nr_set_ybus: nodalMatrix and nodalMatrixTranspose disagree (stale model)");
    }
    NR_HIP(hipStreamSynchronize(h->stream));
    std::vector<double2> GBv(h->nnz);
    for (int p = 0; p < h->nnz; ++p) GBv[p] = double2{G[p], B[p]};
    NR_HIP(jg::sync_copy(h->d_GB, GBv.data(), GBv.size() * sizeof(double2), hipMemcpyHostToDevice, h->stream));
    h->jac_valid = false;
    detach_base(h);                                              // an attached base was factorised for the old nodal matrix
    return 0;
}

int jg_nr_mismatch(jg_nr* h, double* max_p, double* max_q) {
    if (!h) return fail(1, "jg_nr_mismatch: bad argument");
    if (h->fast) return fail(1, "jg_nr_mismatch: this handle runs fast Newton-Raphson (its factor storage holds B', B''); use jg_nr_fast_mismatch");
    if (int rc = set_device(h)) return rc;
    launch_assemble(h);
    launch_check(h, 0);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    h->jac_valid = true;
    h->f_stale = false;
    if (max_p) NR_HIP(jg::sync_copy(max_p, h->d_normp, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
    if (max_q) NR_HIP(jg::sync_copy(max_q, h->d_normq, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_nr_solve(jg_nr* h) {
    if (!h) return fail(1, "jg_nr_solve: bad argument");
    if (h->fast) return fail(1, "jg_nr_solve: this handle runs fast Newton-Raphson; use jg_nr_fast_solve");
    if (int rc = set_device(h)) return rc;
    if (!h->jac_valid) launch_assemble(h);
    h->start_is_base = false;
    h->f_stale = false;                                          // method.mismatch = the mismatch this step started from, for every scenario
    NR_HIP(hipMemsetAsync(h->eng.status, 0, (size_t)h->ld * 4, h->stream));
    if (int rc = newton_step(h, jg::GroupSel{}, nullptr)) return fail(rc, h->eng.error);
    hipLaunchKernelGGL(k_add_iter, dim3((h->ld + 255) / 256), dim3(256), 0, h->stream, h->d_iters, h->ld);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    h->jac_valid = false;
    std::vector<int> st(h->ld);
    NR_HIP(jg::sync_copy(st.data(), h->eng.status, (size_t)h->ld * 4, hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b) if (st[b] & 4) return fail(3, "jg_nr_solve: zero or non-finite pivot (singular Jacobian)");
    return 0;
}

int jg_nr_set_refine(jg_nr* h, int mode) {
    if (!h || mode < 0 || mode > 1) return fail(1, "jg_nr_set_refine: bad argument");
    if (h->fast) return fail(1, "jg_nr_set_refine: fast Newton-Raphson solves with constant matrices");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    if (mode && !h->d_R) {
        const size_t vec = (size_t)h->n * 2 * h->ld * 8;
        NR_HIP(hipMalloc((void**)&h->d_R, vec));
        NR_HIP(hipMalloc((void**)&h->d_inc2[0], vec));
        if (!h->d_inc2[1]) NR_HIP(hipMalloc((void**)&h->d_inc2[1], vec));
        NR_HIP(jg::sync_fill(h->d_R, 0, vec, h->stream));
        NR_HIP(jg::sync_fill(h->d_inc2[0], 0, vec, h->stream));
        NR_HIP(jg::sync_fill(h->d_inc2[1], 0, vec, h->stream));
    }
    if ((mode != 0) != h->refine) {                              // the iteration graph is captured for one mode
        drop_iteration_graphs(h);
    }
    h->refine = mode != 0;
    // a refined step runs forward() + backsolve() on the factor of the step: plain rows (Engine::jordan); back on when refinement goes off
    h->eng.jordan = !h->refine && h->eng.plan->S.jordan && jg::knob("JORDAN", 1) != 0;
    return 0;
}

int jg_nr_set_shared(jg_nr* h, int mode) {
    if (!h || mode < 0 || mode > 1) return fail(1, "jg_nr_set_shared: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    if ((mode != 0) != h->eng.shared) {                          // the iteration graph holds the launches of the other choice
        drop_iteration_graphs(h);
    }
    h->eng.shared = mode != 0;
    return 0;
}

namespace {

// Start of a batched solve: every real scenario active, lanes in home order, all groups in use.  keep_iters: the lanes carry
// their iteration counts (a pool of moved scenarios, jg_nr_resume).
// ONE launch (round 6): until then three values, two fills and five small arrays went to the device through eight stream operations and a synchronise --
// ~90 us on the host's side of every solve, 7 % of a single instance's.
int run_setup(jg_nr* h, int64_t max_iter, double tol, int lanes, bool keep_iters, int defer_at = 0) {
    if (int rc = build_graphs(h)) return rc;
    hipLaunchKernelGGL(k_run_setup, dim3((unsigned)((h->ld + 255) / 256)), dim3(256), 0, h->stream, h->d_params, tol, (double)max_iter, (double)defer_at,   // defer_at: see CompactArgs
                       h->d_iters /* acPowerFlow.jl:1401 */, h->eng.status, h->d_active, h->d_lid, h->d_cflags, h->d_group, h->d_glist, h->ld, lanes, keep_iters ? 1 : 0);
    NR_HIP(hipGetLastError());
    h->res_pinned = h->ld == 64;
    return 0;
}

// The host learns the verdict of an iteration from ONE pinned word the verdict kernel stores into (k_compact: host_count).  Waiting for it with
// hipStreamSynchronize costs a wake-up of ~20 us per iteration -- a tenth of a single instance's iteration; round 5: the host ARMS the word (-1) before
// the launch and polls it (bounded spin, then yields; hipStreamSynchronize after 2 s as the safety net).  The next graph is then launched while the tail of
// the previous one (the predicated re-assembly of a compaction) still runs -- stream order keeps them apart.  JG_POLL=0: the synchronise of round 4.
constexpr double POLL_BELOW_US = 800.0;   // a handle whose waits average more than this blocks in hipStreamSynchronize instead
static bool poll_enabled() { static const bool on = jg::knob("POLL", 1) != 0; return on; }
static void arm_verdict(jg_nr* h) { if (poll_enabled()) *(volatile int*)h->h_counter = -1; }
static hipError_t wait_verdict(jg_nr* h, bool whole = false) {    // whole: the wait is a whole solve of one scenario (run_whole): polled whatever its length, and not counted into wait_us
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    // Polling pays where an iteration is SHORT (a single instance of the 10k-bus grid: 366 us per iteration; 1.595 ms per solve spinning against 1.656 with the
    // synchronise).  Where it is long -- a 512-lane batch: 1.6 ms, three of them in flight on a thread each -- the blocking wait is the right one: measured on one
    // box, interleaved (profiles/r05_poll_ab.txt), the pipeline loses 7 - 10 % to three spinning / napping host threads (254 - 264k against 282 - 287k NR it/s at
    // the driver's K = 20).  The handle remembers how long its last waits took and picks by that.
    if (!poll_enabled() || (!whole && h->wait_us > POLL_BELOW_US)) {
        const hipError_t e = hipStreamSynchronize(h->stream);
        h->wait_us = 0.5 * h->wait_us + 0.5 * elapsed();
        return e;
    }
    volatile int* w = (volatile int*)h->h_counter;
    for (long spins = 0; *w == -1; ++spins) {
        if ((spins & 63) == 63) {
            const double us = elapsed();
            if (us > 2.0e6) return hipStreamSynchronize(h->stream);          // something is wrong (or very slow): the blocking wait reports it
            if (us > 2.0 * POLL_BELOW_US) std::this_thread::yield();          // longer than anything this branch is meant for (the first wait of a big batch): give the core away between looks
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!whole) h->wait_us = 0.5 * h->wait_us + 0.5 * elapsed();
    return hipSuccess;
}

// The iteration loop: one graph per iteration until no scenario is active (the iteration limit itself is kept on the device,
// k_check) -- or, defer_at > 0, until at most defer_at (<= 64) scenarios are: they stay in their lanes, listed for the hand-off (k_compact: hold).
int run_loop(jg_nr* h, int64_t max_iter, int defer_at) {
    const bool trace = jg::knob_set("TRACE");
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int64_t it = h->iter_graphs; it <= max_iter && *h->h_counter != 0; ++it) {
        if (defer_at > 0 && *h->h_counter <= defer_at) break;
        h->iter_graphs += 1;
        // The iteration after which this handle's LAST run stopped is expected to end this one too (the batches of a screen behave alike): its verdict comes from a
        // mismatch-only pass (graph Bm, build_graphs).  A wrong guess costs one more launch -- the Jacobian alone (graph J) -- before the next iteration.
        const bool guess_last = h->stop_hint > 0 && h->iter_graphs == h->stop_hint;
        const double tc = now_us();
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (trace) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, h->stream); }
        if (!trace) arm_verdict(h);
        NR_HIP(hipGraphLaunch(guess_last ? h->execBm : h->execB, h->stream));  // solve!, then mismatch! and the verdict on the new state
        const double tl = now_us();
        if (trace) hipEventRecord(e1, h->stream);
        NR_HIP(trace ? hipStreamSynchronize(h->stream) : wait_verdict(h));
        h->host_launch_us += tl - tc; h->host_wait_us += now_us() - tl; h->host_iters += 1;     // JG_HOST_TIMING: printed when the handle goes
        if (guess_last) {
            const bool stops = *h->h_counter == 0 || (defer_at > 0 && *h->h_counter <= defer_at) || it + 1 > max_iter;
            if (stops) h->last_guess_hit += 1;
            else { h->last_guess_miss += 1; NR_HIP(hipGraphLaunch(h->execJ, h->stream)); }     // the scenarios that go on need their Jacobian after all
        }
        if (trace) {
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1); fprintf(stderr, "[jg_nr_run] graph on the device: %.1f us\n", 1e3 * ms); hipEventDestroy(e0); hipEventDestroy(e1);
            int cf[4];
            jg::sync_copy(cf, h->d_cflags, sizeof(cf), hipMemcpyDeviceToHost, h->stream);
            fprintf(stderr, "[jg_nr_run] iteration %lld: %.1f us, %d scenarios still active, %d of %d lane groups in use%s%s\n", (long long)it + 1,
                    now_us() - tc, *h->h_counter, cf[2], h->ld / 64, cf[0] ? " (compacted)" : "", guess_last ? " (verdict without a Jacobian)" : "");
        }
    }
    h->stop_hint = h->iter_graphs;
    return 0;
}

// The verdict on the start point and -- when every scenario starts from the attached base case (comp_ready) -- the first iteration on the base's
// shared factor instead of a batched refactorisation (jg_comp.hpp).  The loop (run_loop) continues with refactorising iterations either way.
int run_start(jg_nr* h, int64_t max_iter) {
    bool comp = false;
    if (int rc = comp_ready(h, comp)) return rc;
    h->start_is_base = false;                                                  // whatever follows moves the state
    h->iter_graphs = 0;
    arm_verdict(h);
    NR_HIP(hipGraphLaunch(comp ? h->execA2 : h->execA, h->stream));
    NR_HIP(wait_verdict(h));
    if (!comp) { h->first_full += 1; return 0; }
    h->first_comp += 1;
    if (max_iter < 1 || *h->h_counter == 0) {                                  // nothing to iterate: the Jacobian getters find no assembly in place
        h->jac_valid = false;
        return 0;
    }
    const bool trace = jg::knob_set("TRACE");
    const double t0 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    arm_verdict(h);
    NR_HIP(hipGraphLaunch(h->execC, h->stream));
    NR_HIP(wait_verdict(h));
    h->iter_graphs = 1;
    if (trace) fprintf(stderr, "[jg_nr_run] iteration 1 on the shared base factor: %.1f us, %d scenarios still active\n",
                       std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0, *h->h_counter);
    return 0;
}

// ONE scenario whose last run iterated: start verdict + that many iterations as one graph (build_whole_graph).  -1: the handle does not qualify (run_start / run_loop).
int run_whole(jg_nr* h, int64_t max_iter) {
    const bool on = jg::knob("SINGLE", 1) == 1 && h->eng.single_bwd;   // (the handle was created as a single instance: JG_SINGLE at its creation)
    const int k = h->stop_hint;
    if (!on || h->ld != 64 || h->batch != 1 || k < 1 || k > 12 || (int64_t)k > max_iter || h->refine || jg::knob_set("TRACE") || jg::knob_set("HOST_TIMING")) return -1;
    bool comp = false;
    if (int rc = comp_ready(h, comp)) return rc;
    if (comp) return -1;                                         // a base case is attached: the compensated start (run_start)
    if (int rc = build_whole_graph(h, k)) return rc;
    h->start_is_base = false;
    h->first_full += 1;
    arm_verdict(h);
    NR_HIP(hipGraphLaunch(h->execM, h->stream));
    NR_HIP(wait_verdict(h, true));
    h->iter_graphs = k;
    if (*h->h_counter != 0) {                                    // more iterations than the last run took: the Jacobian of the new state, then the loop
        h->last_guess_miss += 1;
        NR_HIP(hipGraphLaunch(h->execJ, h->stream));
        return run_loop(h, max_iter, 0);
    }
    h->last_guess_hit += 1;
    h->stop_hint = std::max(1, (int)h->h_counter[1]);            // the iterations this solve took (pinned: k_verdict64) -- fewer than k: the next solve's graph shrinks
    return 0;
}

// End of a batched solve: lanes back to their home order, method.mismatch of every scenario at its final state, outputs.
int run_finish(jg_nr* h, int32_t* iters, int32_t* status) {
    if (h->ld == 64 && *h->h_counter == 0 && h->res_pinned) {                  // one lane group, every scenario finished: the last verdict (k_compact) has left iterations
        h->f_stale = false;                                                    // and status in pinned host memory, the lanes never left their home order, and what
        h->jac_valid = false;                                                  // k_compact's restore pass resets is rewritten by the next run_setup: no launch, no
        h->paused = false;                                                     // blocking copies (the stream orders whatever follows behind the last graph)
        h->res_pinned = false;
        if (iters) std::memcpy(iters, h->h_counter + 1, (size_t)h->batch * 4);
        if (status) std::memcpy(status, h->h_counter + 65, (size_t)h->batch * 4);
        return 0;
    }
    launch_compact(h, 1);                                                      // lanes back to their home order
    h->f_stale = h->ld > 64;                                                   // method.mismatch of every scenario at its final state: groups that dropped
                                                                               // out early hold other scenarios' rows after a compaction -- one mismatch-only
                                                                               // pass, run when jg_nr_get_mismatch asks for it (0.1 ms per batch of 512)
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    h->jac_valid = false;
    h->paused = false;
    if (iters) NR_HIP(jg::sync_copy(iters, h->d_iters, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    if (status) NR_HIP(jg::sync_copy(status, h->d_status, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

}  // namespace


// ---- first iteration of a common-start batch on ONE shared factor (jg_comp.hpp) -------------------------------------------------------
__global__ void k_lane0(const double* src, double* dst, int n, int ld, int elem) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < elem; ++c) dst[(size_t)i * elem + c] = src[((size_t)i * ld) * elem + c];
}
__global__ void k_broadcast(const double* src, double* dst, int n, int ld) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (b < ld) dst[(size_t)i * ld + b] = src[i];
}

int jg_nr_base_create(jg_nr_base** out, jg_nr* s, int64_t top_cap) {
    if (!out || !s) return fail(1, "jg_nr_base_create: bad argument");
    if (s->fast || s->refine) return fail(1, "jg_nr_base_create: the single-instance handle must run plain Newton-Raphson");
    if (s->batch != 1) return fail(1, "jg_nr_base_create: pass a single-instance handle (batch = 1) that holds the base case's grid, injections and start state");
    if (int rc = set_device(s)) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    NR_HIP(hipStreamSynchronize(s->stream));
    // ONE factorisation of the Jacobian at the start state, plain rows (the compact factor is read from the entries), and y0 = J_0^-1 f_0
    const bool jordan = s->eng.jordan;
    s->eng.jordan = false;
    NR_HIP(hipMemsetAsync(s->eng.status, 0, (size_t)s->ld * 4, s->stream));
    launch_assemble(s, jg::GroupSel{}, true, nullptr, 0, nullptr, true);
    int rc = s->eng.factor(s->stream, nullptr, s->d_F, jg::GroupSel{}, s->level0_done);
    if (!rc) rc = s->eng.backsolve(s->stream, s->d_inc, jg::StateUpdate{}, jg::GroupSel{});
    s->eng.jordan = jordan;
    s->jac_valid = false;
    if (rc) return fail(rc, s->eng.error);
    NR_HIP(hipStreamSynchronize(s->stream));
    int st0 = 0;
    NR_HIP(jg::sync_copy(&st0, s->eng.status, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    if (st0 & 4) return fail(3, "jg_nr_base_create: zero or non-finite pivot (the Jacobian of the base case is singular)");
    jg_nr_base* b = new jg_nr_base();
    b->n = s->n; b->nnz = s->nnz; b->device = s->device;
    b->colptr = s->colptr; b->rowval = s->rowval; b->type = s->type;
    jg::CompBase& cb = b->cb;
    cb.nnz = s->nnz;
    auto bail = [&](int code, const std::string& m) { release_base(b); return fail(code, m); };
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return bail(2, "jg_nr_base_create: stream creation failed");
    const int n = s->n, nnz = s->nnz;
    std::string err;
    {   // the Ybus pattern as the correction kernel reads it: row of a position, position of the transposed entry
        std::vector<int> rp(n + 1), tp(nnz, -1);
        for (int i = 0; i <= n; ++i) rp[i] = (int)(s->colptr[i] - 1);
        for (int p = 0; p < nnz; ++p) {
            const int i = s->csr_row[p], j = s->csr_col[p];
            const int* lo = s->csr_col.data() + rp[j]; const int* hi = s->csr_col.data() + rp[j + 1];
            const int* q = std::lower_bound(lo, hi, i);
            tp[p] = (int)(q - s->csr_col.data());
        }
        if (jg::upload(&cb.rowptr, rp, err, b->stream) || jg::upload(&cb.posrow, s->csr_row, err, b->stream) || jg::upload(&cb.tpos, tp, err, b->stream)) return bail(2, err);
        hipError_t e = hipMalloc((void**)&cb.colm, (size_t)nnz * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&cb.rowtype, (size_t)n * sizeof(int));
        if (e == hipSuccess) e = jg::sync_copy(cb.colm, s->d_col, (size_t)nnz * sizeof(int), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = jg::sync_copy(cb.rowtype, s->d_rowtype, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, b->stream);
        for (double** q : {&cb.v0, &cb.th0, &cb.p0, &cb.q0}) if (e == hipSuccess) e = hipMalloc((void**)q, (size_t)n * sizeof(double));
        for (double** q : {&cb.f0, &cb.y0}) if (e == hipSuccess) e = hipMalloc((void**)q, (size_t)n * 2 * sizeof(double));
        if (e != hipSuccess) return bail(2, std::string("jg_nr_base_create: ") + hipGetErrorString(e));
        const dim3 g((n + 255) / 256), t(256);
        hipLaunchKernelGGL(k_lane0, g, t, 0, b->stream, (const double*)s->d_vm, cb.v0, n, s->ld, 1);
        hipLaunchKernelGGL(k_lane0, g, t, 0, b->stream, (const double*)s->d_va, cb.th0, n, s->ld, 1);
        hipLaunchKernelGGL(k_lane0, g, t, 0, b->stream, (const double*)s->d_p, cb.p0, n, s->ld, 1);
        hipLaunchKernelGGL(k_lane0, g, t, 0, b->stream, (const double*)s->d_q, cb.q0, n, s->ld, 1);
        hipLaunchKernelGGL(k_lane0, g, t, 0, b->stream, (const double*)s->d_F, cb.f0, n, s->ld, 2);
        hipLaunchKernelGGL(k_lane0, g, t, 0, b->stream, (const double*)s->d_inc, cb.y0, n, s->ld, 2);
        if (hipStreamSynchronize(b->stream) != hipSuccess) return bail(2, "jg_nr_base_create: copy of the base state failed");
    }
    // default top: what keeps the dense product near the cost of the level launches it replaces (10 000-bus grid: 502 pivots = the levels above the 11th)
    const int cap = top_cap < 0 ? -1 : (top_cap == 0 ? 512 : (int)std::min<int64_t>(top_cap, 4096));
    if (int rc2 = cb.create(s->eng.plan->S, s->eng.X, s->eng.ld, cap, b->stream)) return bail(rc2, "jg_nr_base_create: " + cb.error);
    if (int rc2 = jg::comp_form_z(cb, b->stream)) return bail(rc2, "jg_nr_base_create: " + cb.error);
    b->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = b;
    return 0;
}

void jg_nr_base_destroy(jg_nr_base* b) { release_base(b); }

int jg_nr_base_info(jg_nr_base* b, int64_t* info) {
    if (!b || !info) return fail(1, "jg_nr_base_info: bad argument");
    const jg::CompBase& cb = b->cb;
    info[0] = cb.split.T.n_top; info[1] = cb.split.T.split; info[2] = (int64_t)cb.split.fwd.size(); info[3] = (int64_t)cb.split.bwd.size();
    info[4] = (int64_t)cb.full.fwd.size(); info[5] = (int64_t)cb.full.bwd.size(); info[6] = (int64_t)(b->create_ms * 1000.0); info[7] = b->refs.load() - 1;
    return 0;
}

int jg_nr_base_get(jg_nr_base* b, int which, double* out, int64_t cap) {
    if (!b || !out) return fail(1, "jg_nr_base_get: bad argument");
    const jg::CompBase& cb = b->cb;
    const double* src = nullptr; size_t count = 0;
    switch (which) {
        case 0: src = cb.Zc; count = (size_t)cb.nnz * 4; break;                                  // J_0^-1 on the Ybus pattern, row-CSR order
        case 1: src = cb.y0; count = (size_t)cb.n * 2; break;
        case 2: src = cb.f0; count = (size_t)cb.n * 2; break;
        case 3: src = cb.Sinv; count = cb.Sinv ? (size_t)((2 * cb.split.T.n_top + 15) / 16) * cb.lds * 64 : 0; break;   // MFMA fragment order (jg_comp.hip: k_ctop)
        case 4: src = cb.Mc; count = (size_t)cb.n_entries * 4; break;
        default: return fail(1, "jg_nr_base_get: which = 0 (inverse on the pattern) | 1 (J0^-1 f0) | 2 (f0) | 3 (dense top inverse) | 4 (compact factor)");
    }
    if ((int64_t)count > cap) return fail(1, "jg_nr_base_get: buffer too small");
    NR_HIP(hipSetDevice(b->device));
    if (count) NR_HIP(jg::sync_copy(out, src, count * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    return 0;
}

int jg_nr_attach_base(jg_nr* h, jg_nr_base* b) {
    if (!h) return fail(1, "jg_nr_attach_base: bad argument");
    if (int rc = set_device(h)) return rc;
    detach_base(h);
    if (!b) return 0;
    if (h->fast) return fail(1, "jg_nr_attach_base: fast Newton-Raphson factorises constant matrices once anyway");
    if (b->device != h->device || b->n != h->n || b->nnz != h->nnz || b->colptr != h->colptr || b->rowval != h->rowval || b->type != h->type)
        return fail(1, "jg_nr_attach_base: the base was built for another grid (pattern, bus types) or device");
    const size_t bytes = b->cb.scratch_rows(b->cb.split) * (size_t)h->ld * 2 * sizeof(double);
    NR_HIP(hipMalloc((void**)&h->d_cw, bytes));
    NR_HIP(jg::sync_fill(h->d_cw, 0, bytes, h->stream));
    if (!h->d_cmp) NR_HIP(hipMalloc((void**)&h->d_cmp, sizeof(int)));
    b->refs.fetch_add(1);
    h->base = b;
    h->inj_checked = false;
    return 0;
}

int jg_nr_start_from_base(jg_nr* h) {
    if (!h || !h->base) return fail(1, "jg_nr_start_from_base: no base attached (jg_nr_attach_base)");
    if (int rc = set_device(h)) return rc;
    const dim3 g((h->ld + 255) / 256, h->n), t(256);
    hipLaunchKernelGGL(k_broadcast, g, t, 0, h->stream, (const double*)h->base->cb.v0, h->d_vm, h->n, h->ld);
    hipLaunchKernelGGL(k_broadcast, g, t, 0, h->stream, (const double*)h->base->cb.th0, h->d_va, h->n, h->ld);
    NR_HIP(hipGetLastError());
    h->jac_valid = false;
    h->start_is_base = true;
    return 0;
}

int jg_nr_set_first_iteration(jg_nr* h, int mode) {
    if (!h || mode < 0 || mode > 1) return fail(1, "jg_nr_set_first_iteration: mode = 0 (always refactorise) | 1 (shared base factor when the conditions hold)");
    h->first_mode = mode;
    return 0;
}

int jg_nr_first_iteration_counts(jg_nr* h, int64_t* compensated, int64_t* refactorised) {
    if (!h) return fail(1, "jg_nr_first_iteration_counts: bad argument");
    if (compensated) *compensated = h->first_comp;
    if (refactorised) *refactorised = h->first_full;
    return 0;
}

int jg_nr_run(jg_nr* h, int64_t max_iter, double tol, int32_t* iters, int32_t* status) {
    if (!h || max_iter < 0 || !(tol > 0.0)) return fail(1, "jg_nr_run: bad argument");
    if (h->fast) return fail(1, "jg_nr_run: this handle runs fast Newton-Raphson; use jg_nr_fast_run");
    if (int rc = set_device(h)) return rc;
    if (int rc = run_setup(h, max_iter, tol, h->batch, false)) return rc;
    const int rw = run_whole(h, max_iter);                                     // ONE scenario that has solved before: the whole solve as one graph; -1: not such a handle
    if (rw > 0) return rw;
    if (rw < 0) {
        if (int rc = run_start(h, max_iter)) return rc;                        // acPowerFlow.jl:1406: mismatch!, verdict (+ a compensated first iteration)
        if (int rc = run_loop(h, max_iter, 0)) return rc;
    }
    return run_finish(h, iters, status);
}

int jg_nr_run_defer(jg_nr* h, int64_t max_iter, double tol, int64_t defer_at, int32_t* n_left) {
    if (!h || max_iter < 0 || !(tol > 0.0) || defer_at < 0 || defer_at > 64 || !n_left) return fail(1, "jg_nr_run_defer: bad argument (defer_at in 0..64)");
    if (h->fast) return fail(1, "jg_nr_run_defer: this handle runs fast Newton-Raphson");
    if (int rc = set_device(h)) return rc;
    if (int rc = run_setup(h, max_iter, tol, h->batch, false, h->ld > 64 ? (int)defer_at : 0)) return rc;
    if (int rc = run_start(h, max_iter)) return rc;
    if (int rc = run_loop(h, max_iter, h->ld > 64 ? (int)defer_at : 0)) return rc;   // one lane group: lanes are never packed, nothing to hand off
    *n_left = *h->h_counter;
    h->paused = true;
    return 0;
}

int jg_nr_finish(jg_nr* h, int32_t* iters, int32_t* status) {
    if (!h || !h->paused) return fail(1, "jg_nr_finish: the handle is not paused (jg_nr_run_defer)");
    if (int rc = set_device(h)) return rc;
    return run_finish(h, iters, status);
}

int jg_nr_move_lanes(jg_nr* dst, int64_t dst_lane0, jg_nr* src, int32_t* home, int32_t* count) {
    if (!dst || !src || dst == src || !home || !count || dst_lane0 < 0) return fail(1, "jg_nr_move_lanes: bad argument");
    if (!src->paused) return fail(1, "jg_nr_move_lanes: the source is not paused (jg_nr_run_defer)");
    if (dst->device != src->device || dst->n != src->n || dst->nnz != src->nnz || dst->mp != src->mp || dst->fast || src->fast)
        return fail(1, "jg_nr_move_lanes: the two handles must hold the same grid on the same device");
    //  the factorisation plan -- where the top starts, front caps, tasks or wave records: the summation order -- is chosen by the class of
    // batch a handle was created for (Engine::create, by the lane count PADDED to a multiple of 64: 64 with at most 32 scenarios, 64, 128 / 192, 256 and more).  A straggler that finishes under another
    // plan is no longer bitwise the scenario of a lockstep batch, silently: refuse the hand-off instead.
    if (dst->eng.plan && src->eng.plan && dst->eng.plan->policy != src->eng.plan->policy)
        return fail(1, "jg_nr_move_lanes: the pool runs another factorisation plan than the batch (create both for the same class of batch: "
                       "1-32 scenarios, 33-64, 65-192, 193 and more = padded lane count 64 with at most 32 scenarios, 64, 128 / 192, 256 and more)");
    if (int rc = set_device(dst)) return rc;
    for (jg_nr* h : {dst, src})
        if (!h->d_move) {
            NR_HIP(hipMalloc((void**)&h->d_move, 193 * sizeof(int)));
            NR_HIP(hipHostMalloc((void**)&h->h_move, 65 * sizeof(int)));
        }
    NR_HIP(hipStreamSynchronize(src->stream));
    int* map = dst->d_move; int* d_home = dst->d_move + 64; int* d_count = dst->d_move + 128;
    MovePlanArgs pa{src->d_active, src->d_iters, src->d_status, src->d_lid, src->d_ppos, src->ld,
                    dst->d_active, dst->d_iters, dst->d_status, dst->eng.status, dst->d_ppos, dst->ld,
                    map, d_home, d_count, src->mp, (int)dst_lane0, dst->batch, src->d_cflags, src->d_dest, dst->d_move + 129};
    hipLaunchKernelGGL(k_move_plan, dim3(1), dim3(64), 0, dst->stream, pa);
    MoveRowsArgs ra{};
    int na = 0;
    auto add = [&](const double* s, double* t, int rows) { ra.src[na] = s; ra.dst[na] = t; ra.rows[na] = rows; ++na; };
    add(src->d_vm, dst->d_vm, src->n); add(src->d_va, dst->d_va, src->n); add(src->d_p, dst->d_p, src->n); add(src->d_q, dst->d_q, src->n);
    if (src->mp > 0) { add(src->d_pdg, dst->d_pdg, src->mp); add(src->d_pdb, dst->d_pdb, src->mp); }
    ra.s_ld = src->ld; ra.d_ld = dst->ld; ra.map = map; ra.srcl = dst->d_move + 129;
    hipLaunchKernelGGL(k_move_rows, dim3((unsigned)std::min((src->n + 3) / 4, 1024), (unsigned)na), dim3(256), 0, dst->stream, ra);
    NR_HIP(hipGetLastError());
    NR_HIP(hipMemcpyAsync(dst->h_move, d_home, 65 * sizeof(int), hipMemcpyDeviceToHost, dst->stream));
    NR_HIP(hipStreamSynchronize(dst->stream));
    const int c = dst->h_move[64];
    if (c < 0) return fail(1, "jg_nr_move_lanes: the destination has not enough free lanes");
    for (int i = 0; i < c; ++i) home[i] = dst->h_move[i];
    *count = c;
    *src->h_counter = 0;
    src->res_pinned = false;                                    // (the source's last verdict saw active scenarios: nothing stands behind its pinned word -- jg_nr_finish copies)
    dst->jac_valid = false;
    return 0;
}

int jg_nr_resume(jg_nr* h, int64_t lanes, int64_t max_iter, double tol, int32_t* iters, int32_t* status) {
    if (!h || lanes < 0 || lanes > h->batch || max_iter < 0 || !(tol > 0.0)) return fail(1, "jg_nr_resume: bad argument");
    if (h->fast) return fail(1, "jg_nr_resume: this handle runs fast Newton-Raphson");
    if (int rc = set_device(h)) return rc;
    // lanes [0, lanes) hold moved scenarios (active, their iteration counts with them); the others are idle
    if (int rc = run_setup(h, max_iter, tol, (int)lanes, true)) return rc;
    if (lanes < h->ld) {
        NR_HIP(hipMemsetAsync(h->d_active + lanes, 0, (size_t)(h->ld - lanes) * 4, h->stream));
        NR_HIP(hipMemsetAsync(h->eng.status + lanes, 0, (size_t)(h->ld - lanes) * 4, h->stream));
        NR_HIP(hipMemsetD32Async((hipDeviceptr_t)(h->d_iters + lanes), (int)max_iter, (size_t)(h->ld - lanes), h->stream));   // idle lanes: at the limit, never picked up by a verdict
    }
    launch_compact(h, 0, true);                                                // groups in use (the lanes are already packed)
    launch_assemble(h, active_groups(h), true, nullptr, 0, nullptr, true);     // the Jacobian of the state they arrived with; NO verdict:
    NR_HIP(hipStreamSynchronize(h->stream));                                   // theirs was taken (and counted) where they came from
    h->iter_graphs = 0;
    if (int rc = run_loop(h, max_iter, 0)) return rc;
    h->paused = true;
    if (int rc = run_finish(h, nullptr, nullptr)) return rc;
    if (iters) NR_HIP(jg::sync_copy(iters, h->d_iters, (size_t)lanes * 4, hipMemcpyDeviceToHost, h->stream));
    if (status) NR_HIP(jg::sync_copy(status, h->d_status, (size_t)lanes * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_nr_pack_rows_device(jg_nr* h, double* dst_dev, int64_t lane0, int64_t count, const int32_t* rows) {
    if (!h || !dst_dev || !rows || lane0 < 0 || count < 0 || lane0 + count > h->batch || count > h->ld) return fail(1, "jg_nr_pack_rows_device: bad argument");
    if (count == 0) return 0;
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipMemcpyAsync(h->d_itmp, rows, (size_t)count * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)std::min((h->n + 255) / 256, 64), (unsigned)count), dim3(256), 0, h->stream,
                       h->d_vm, h->d_va, h->d_iters, h->d_status, h->d_itmp, dst_dev, h->n, h->ld, (int)lane0, 2LL * h->n + 2);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_nr_get_mismatch(jg_nr* h, double* mism) {
    if (!h || !mism) return fail(1, "jg_nr_get_mismatch: bad argument");
    if (int rc = set_device(h)) return rc;
    if (h->f_stale) { launch_assemble(h, jg::GroupSel{}, false); NR_HIP(hipGetLastError()); h->f_stale = false; }
    NR_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> t((size_t)h->n * 2 * h->batch);
    if (int rc = get_bus_array(h, h->d_F, t.data(), 2)) return rc;
    for (int b = 0; b < h->batch; ++b) {
        const double* s = t.data() + (size_t)b * h->n * 2;
        double* d = mism + (size_t)b * h->dimJ;
        for (int i = 0; i < h->n; ++i) {
            if (h->pvpq[i]) d[h->pvpq[i] - 1] = s[2 * i];
            if (h->pq[i]) d[h->pq[i] - 1] = s[2 * i + 1];
        }
    }
    return 0;
}

int jg_nr_get_increment(jg_nr* h, double* incr) {
    if (!h || !incr) return fail(1, "jg_nr_get_increment: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> t((size_t)h->n * 2 * h->batch);
    if (int rc = get_bus_array(h, h->d_inc, t.data(), 2)) return rc;
    for (int b = 0; b < h->batch; ++b) {
        const double* s = t.data() + (size_t)b * h->n * 2;
        double* d = incr + (size_t)b * h->dimJ;
        for (int i = 0; i < h->n; ++i) {
            if (h->pvpq[i]) d[h->pvpq[i] - 1] = s[2 * i];
            if (h->pq[i]) d[h->pq[i] - 1] = s[2 * i + 1];
        }
    }
    return 0;
}

int jg_nr_get_jacobian(jg_nr* h, double* nzval) {
    if (!h || !nzval) return fail(1, "jg_nr_get_jacobian: bad argument");
    if (h->fast) return fail(1, "jg_nr_get_jacobian: this handle runs fast Newton-Raphson: the factor storage holds the factorised B', B'', assembling the full Jacobian there would destroy them");
    if (int rc = set_device(h)) return rc;
    if (!h->jac_valid) { launch_assemble(h); h->jac_valid = true; }      // Jacobian at the current state
    NR_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> t((size_t)h->eng.plan->S.n_entries * 4 * h->ld);     // the factor storage holds the Jacobian until the next factorisation
    NR_HIP(jg::sync_copy(t.data(), h->eng.X, t.size() * 8, hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b)
        for (int64_t k = 0; k < h->nnzJ; ++k) nzval[(size_t)b * h->nnzJ + k] = t[(((size_t)(h->jmap[k] >> 2) * 2 + ((h->jmap[k] >> 1) & 1)) * h->ld + b) * 2 + (h->jmap[k] & 1)];
    return 0;
}

int jg_nr_get_maps(jg_nr* h, int64_t* pq, int64_t* pvpq, int64_t* pcount, int64_t* jcolptr, int64_t* jrowval) {
    if (!h) return fail(1, "jg_nr_get_maps: bad argument");
    if (pq) std::memcpy(pq, h->pq.data(), h->pq.size() * 8);
    if (pvpq) std::memcpy(pvpq, h->pvpq.data(), h->pvpq.size() * 8);
    if (pcount) std::memcpy(pcount, h->pcount.data(), h->pcount.size() * 8);
    if (jcolptr) std::memcpy(jcolptr, h->jcolptr.data(), h->jcolptr.size() * 8);
    if (jrowval) std::memcpy(jrowval, h->jrowval.data(), h->jrowval.size() * 8);
    return 0;
}

int jg_nr_get_iteration(jg_nr* h, int32_t* iters) {
    if (!h || !iters) return fail(1, "jg_nr_get_iteration: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    NR_HIP(jg::sync_copy(iters, h->d_iters, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

// ---- fast decoupled Newton-Raphson (fastNewtonRaphsonBX / XB, acPowerFlow.jl:215-537, 687-730, 913-983) ---------------
constexpr int FAST_MP = 4;                            // edits per scenario: a branch touches (i,i), (j,j), (i,j), (j,i) of B' and of B''
// X[entry] += diag(dp, dq) for the scenario of the lane: the matrices of a batch differ from the shared ones in a handful of entries
__global__ void k_fast_patch(double* X, const int* entry, const double* dp, const double* dq, int ld, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (b >= batch) return;
    const int e = entry[(size_t)m * batch + b];
    if (e < 0) return;
    double* x = X + ((size_t)e * 2 * ld + b) * 2;                 // block row 0: (v00, v01); block row 1 sits ld * 2 doubles further
    x[0] += dp[(size_t)m * batch + b];
    x[(size_t)ld * 2 + 1] += dq[(size_t)m * batch + b];
}

// shared matrices + every scenario's edits into the factor storage, then ONE factorisation for the whole batch (fastNewtonRaphson factorises
// once per matrix, acPowerFlow.jl:923-927; after updateBranch! the reference calls lu! again, :476-537).  A scenario whose B' or B'' is singular
// (an islanding outage) keeps status bit 2 in eng.status: jg_nr_fast_run reports it as status 3, its neighbours are untouched.
static int fast_refactor(jg_nr* h) {
    if (int rc = h->eng.set_shared_matrix(h->stream, h->fast_blk.data())) return fail(rc, h->eng.error);
    bool any = false;
    for (int e : h->fp_entry) any = any || e >= 0;
    if (any) {
        //  the edit tables live in buffers of the handle: three hipMalloc / hipFree pairs per call were device-wide synchronisations that could stall
        // other pipeline threads in the middle of a graph capture
        const size_t cnt = h->fp_entry.size();
        if (!h->d_fp_entry) {
            NR_HIP(hipMalloc((void**)&h->d_fp_entry, cnt * sizeof(int)));
            NR_HIP(hipMalloc((void**)&h->d_fp_dp, cnt * sizeof(double)));
            NR_HIP(hipMalloc((void**)&h->d_fp_dq, cnt * sizeof(double)));
        }
        NR_HIP(hipMemcpyAsync(h->d_fp_entry, h->fp_entry.data(), cnt * sizeof(int), hipMemcpyHostToDevice, h->stream));
        NR_HIP(hipMemcpyAsync(h->d_fp_dp, h->fp_dp.data(), cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        NR_HIP(hipMemcpyAsync(h->d_fp_dq, h->fp_dq.data(), cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_fast_patch, dim3((h->batch + 255) / 256, FAST_MP), dim3(256), 0, h->stream, h->eng.X, (const int*)h->d_fp_entry, (const double*)h->d_fp_dp,
                           (const double*)h->d_fp_dq, h->ld, h->batch);
        NR_HIP(hipStreamSynchronize(h->stream));
    }
    NR_HIP(hipMemsetAsync(h->eng.status, 0, (size_t)h->ld * 4, h->stream));
    h->eng.jordan = false;                                       // factor once, then forward() + backsolve() per half iteration: plain rows
    if (int rc = h->eng.factor(h->stream, nullptr, h->d_R, jg::GroupSel{})) return fail(rc, h->eng.error);   // ONCE (lu(jacobian), utility.jl:470-476)
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

static int fast_half(jg_nr* h, int pass) {
    // pass 1: rhs = (f_P / V, 0) is already in d_R (written by the mismatch pass); pass 2: Q mismatches at the new angles first
    if (pass == 2) launch_assemble(h, jg::GroupSel{}, false, nullptr, 2);
    if (int rc = h->eng.forward(h->stream, h->d_R, jg::GroupSel{})) return fail(rc, h->eng.error);
    jg::StateUpdate upd{h->d_va, h->d_vm, h->d_flags, h->fast_mask, +1.0};            // theta += / V += (:946-950, 966-970)
    if (int rc = h->eng.backsolve(h->stream, h->d_inc2[pass - 1], upd, jg::GroupSel{})) return fail(rc, h->eng.error);
    return 0;
}

int jg_nr_fast_setup(jg_nr* h, const double* bp, const double* bq) {
    if (!h || !bp || !bq) return fail(1, "jg_nr_fast_setup: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    // block (r, c) of the shared matrix = diag(B'[r,c], B''[r,c]); the caller pads slack / PV rows and columns with identity
    if (h->refine) return fail(1, "jg_nr_fast_setup: the handle is set up for refined full Newton-Raphson steps");
    h->fast_blk.assign((size_t)h->nnz * 4, 0.0);
    for (int p = 0; p < h->nnz; ++p) { h->fast_blk[(size_t)h->tperm[p] * 4] = bp[p]; h->fast_blk[(size_t)h->tperm[p] * 4 + 3] = bq[p]; }
    // (engine block q sits at row = the column that pointer q belongs to, col = rowval[q]; tperm[p of (r, c)] = pointer of (c, r))
    h->fp_entry.assign((size_t)FAST_MP * h->batch, -1); h->fp_dp.assign((size_t)FAST_MP * h->batch, 0.0); h->fp_dq.assign((size_t)FAST_MP * h->batch, 0.0);   // new matrices: no edits
    const size_t vec = (size_t)h->n * 2 * h->ld * 8;
    if (!h->d_R) NR_HIP(hipMalloc((void**)&h->d_R, vec));
    if (!h->d_inc2[0]) NR_HIP(hipMalloc((void**)&h->d_inc2[0], vec));
    if (!h->d_inc2[1]) NR_HIP(hipMalloc((void**)&h->d_inc2[1], vec));
    NR_HIP(jg::sync_fill(h->d_R, 0, vec, h->stream));
    if (int rc = fast_refactor(h)) return rc;
    std::vector<int> st(h->ld);
    NR_HIP(jg::sync_copy(st.data(), h->eng.status, (size_t)h->ld * 4, hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b) if (st[b] & 4) return fail(3, "jg_nr_fast_setup: zero or non-finite pivot (singular B matrix)");
    h->fast = true;
    h->jac_valid = false;
    return 0;
}

int jg_nr_fast_patch_batch(jg_nr* h, int64_t scenario0, int64_t count, int64_t k, const int64_t* ptr, const double* dbp, const double* dbq) {
    if (!h || !h->fast) return fail(1, "jg_nr_fast_patch_batch: call jg_nr_fast_setup first");
    if (scenario0 < 0 || count < 1 || scenario0 + count > h->batch || k < 0 || k > FAST_MP || (k > 0 && (!ptr || !dbp || !dbq)))
        return fail(1, "jg_nr_fast_patch_batch: bad argument (scenario range / more than 4 entries per scenario)");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    const jg::BlockSymbolic& S = h->eng.plan->S;
    for (int64_t sc = 0; sc < count; ++sc)                       // the whole call is checked before a single edit is recorded: a refused call changes nothing
        for (int64_t m = 0; m < k; ++m) {
            const int64_t p = ptr[sc * k + m];
            if (p == 0) continue;
            if (p < 1 || p > h->nnz) return fail(1, "jg_nr_fast_patch_batch: pointer out of range");
            for (int64_t mm = 0; mm < m; ++mm) if (ptr[sc * k + mm] == p) return fail(1, "jg_nr_fast_patch_batch: duplicate pointer");
        }
    for (int64_t sc = 0; sc < count; ++sc)
        for (int m = 0; m < FAST_MP; ++m) {
            const size_t at = (size_t)m * h->batch + scenario0 + sc;
            h->fp_entry[at] = -1; h->fp_dp[at] = 0.0; h->fp_dq[at] = 0.0;
            if (m >= k || ptr[sc * k + m] == 0) continue;
            const int64_t p = ptr[sc * k + m];
            h->fp_entry[at] = S.src_entry[h->tperm[p - 1]];
            h->fp_dp[at] = dbp[sc * k + m]; h->fp_dq[at] = dbq[sc * k + m];
        }
    return fast_refactor(h);
}

int jg_nr_fast_mismatch(jg_nr* h, double* max_p, double* max_q) {
    if (!h || !h->fast) return fail(1, "jg_nr_fast_mismatch: call jg_nr_fast_setup first");
    if (int rc = set_device(h)) return rc;
    launch_assemble(h, jg::GroupSel{}, false, nullptr, 1);
    launch_check(h, 0);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    if (max_p) NR_HIP(jg::sync_copy(max_p, h->d_normp, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
    if (max_q) NR_HIP(jg::sync_copy(max_q, h->d_normq, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_nr_fast_solve(jg_nr* h) {
    if (!h || !h->fast) return fail(1, "jg_nr_fast_solve: call jg_nr_fast_setup first");
    if (int rc = set_device(h)) return rc;
    launch_assemble(h, jg::GroupSel{}, false, nullptr, 1);                      // solve! uses the mismatches of the current state
    h->fast_mask = nullptr;
    if (int rc = fast_half(h, 1)) return rc;
    if (int rc = fast_half(h, 2)) return rc;
    hipLaunchKernelGGL(k_add_iter, dim3((h->ld + 255) / 256), dim3(256), 0, h->stream, h->d_iters, h->ld);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_nr_fast_get_increment(jg_nr* h, double* incr) {
    if (!h || !h->fast || !incr) return fail(1, "jg_nr_fast_get_increment: bad argument");
    if (int rc = set_device(h)) return rc;
    NR_HIP(hipStreamSynchronize(h->stream));
    std::vector<double> t0((size_t)h->n * 2 * h->batch), t1((size_t)h->n * 2 * h->batch);
    if (int rc = get_bus_array(h, h->d_inc2[0], t0.data(), 2)) return rc;
    if (int rc = get_bus_array(h, h->d_inc2[1], t1.data(), 2)) return rc;
    for (int b = 0; b < h->batch; ++b) {                          // [active.increment (pvpq order) | reactive.increment (pq order)]
        double* d = incr + (size_t)b * h->dimJ;
        for (int i = 0; i < h->n; ++i) {
            if (h->pvpq[i]) d[h->pvpq[i] - 1] = t0[((size_t)b * h->n + i) * 2];
            if (h->pq[i]) d[h->pq[i] - 1] = t1[((size_t)b * h->n + i) * 2 + 1];
        }
    }
    return 0;
}

int jg_nr_fast_run(jg_nr* h, int64_t max_iter, double tol, int32_t* iters, int32_t* status) {
    if (!h || !h->fast || max_iter < 0 || !(tol > 0.0)) return fail(1, "jg_nr_fast_run: bad argument (or jg_nr_fast_setup missing)");
    if (int rc = set_device(h)) return rc;
    if (!h->execFA) {
        std::lock_guard<std::mutex> lk(jg::capture_mutex());
        NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        hipMemsetAsync(h->d_counter, 0, sizeof(int), h->stream);
        launch_assemble(h, jg::GroupSel{}, false, nullptr, 1);                  // mismatch! (:687-730)
        launch_check(h, 1);                                                     // powerFlow! accounting (:1406-1420)
        hipMemcpyAsync(h->h_counter, h->d_counter, sizeof(int), hipMemcpyDeviceToHost, h->stream);
        NR_HIP(hipStreamEndCapture(h->stream, &h->graphFA));
        NR_HIP(hipGraphInstantiate(&h->execFA, h->graphFA, nullptr, nullptr, 0));
        NR_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        h->fast_mask = h->d_active;                                             // solve! (:913-983) on the scenarios still iterating
        int rc = fast_half(h, 1);
        if (!rc) rc = fast_half(h, 2);
        hipError_t e = hipStreamEndCapture(h->stream, &h->graphFB);
        if (rc) return rc;
        NR_HIP(e);
        NR_HIP(hipGraphInstantiate(&h->execFB, h->graphFB, nullptr, nullptr, 0));
    }
    const double params[2] = {tol, (double)max_iter};
    NR_HIP(hipMemcpyAsync(h->d_params, params, sizeof(params), hipMemcpyHostToDevice, h->stream));
    NR_HIP(hipMemsetAsync(h->d_iters, 0, (size_t)h->ld * 4, h->stream));
    // (eng.status is NOT cleared: nothing is factorised inside this loop -- the pivot flags of the batch's ONE factorisation, jg_nr_fast_setup /
    // jg_nr_fast_patch_batch, are what the verdict of a scenario with a singular B' or B'' must see)
    {
        std::vector<int> act(h->ld, 0);
        for (int b = 0; b < h->batch; ++b) act[b] = 1;
        NR_HIP(hipMemcpyAsync(h->d_active, act.data(), (size_t)h->ld * 4, hipMemcpyHostToDevice, h->stream));
        NR_HIP(hipStreamSynchronize(h->stream));
    }
    for (int64_t it = 0; it <= max_iter; ++it) {
        NR_HIP(hipGraphLaunch(h->execFA, h->stream));
        NR_HIP(hipStreamSynchronize(h->stream));
        if (*h->h_counter == 0) break;
        NR_HIP(hipGraphLaunch(h->execFB, h->stream));
    }
    NR_HIP(hipStreamSynchronize(h->stream));
    if (iters) NR_HIP(jg::sync_copy(iters, h->d_iters, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    if (status) NR_HIP(jg::sync_copy(status, h->d_status, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_nr_set_branches(jg_nr* h, int64_t nb, const int64_t* from, const int64_t* to, const int8_t* status, const double* param) {
    if (!h || nb < 1 || !from || !to || !status || !param) return fail(1, "jg_nr_set_branches: bad argument");
    if (int rc = set_device(h)) return rc;
    std::vector<int> f(nb), t(nb);
    for (int64_t k = 0; k < nb; ++k) {
        if (from[k] < 1 || from[k] > h->n || to[k] < 1 || to[k] > h->n) return fail(1, "jg_nr_set_branches: bus index out of range");
        f[k] = (int)(from[k] - 1); t[k] = (int)(to[k] - 1);
    }
    NR_HIP(hipStreamSynchronize(h->stream));
    hipFree(h->d_bfrom); hipFree(h->d_bto); hipFree(h->d_bstatus); hipFree(h->d_bparam);
    h->d_bfrom = h->d_bto = nullptr; h->d_bstatus = nullptr; h->d_bparam = nullptr;
    std::string err;
    if (jg::upload(&h->d_bfrom, f, err, h->stream) || jg::upload(&h->d_bto, t, err, h->stream) ||
        jg::upload(&h->d_bstatus, std::vector<signed char>(status, status + nb), err, h->stream) ||
        jg::upload(&h->d_bparam, std::vector<double>(param, param + nb * 16), err, h->stream))
        return fail(2, err);
    if (!h->d_outage) {
        NR_HIP(hipMalloc((void**)&h->d_outage, (size_t)h->ld * sizeof(int)));
        NR_HIP(jg::sync_fill(h->d_outage, 0, (size_t)h->ld * sizeof(int), h->stream));
    }
    if ((int)nb != h->nb) {                                      //  the screen's partial maxima are sized by the branch count and a rating
        hipFree(h->d_screen); hipFree(h->d_screc); hipFree(h->d_rating);   // belongs to the branch table it was installed for: both go with the old table
        h->d_screen = h->d_screc = h->d_rating = nullptr;
    }
    h->nb = (int)nb;
    return 0;
}

int jg_nr_set_outage_labels(jg_nr* h, const int64_t* label) {
    if (!h || !label) return fail(1, "jg_nr_set_outage_labels: bad argument");
    if (!h->d_outage) return fail(1, "jg_nr_set_outage_labels: call jg_nr_set_branches first");
    if (int rc = set_device(h)) return rc;
    std::vector<int> v(h->ld, 0);
    for (int b = 0; b < h->batch; ++b) {
        if (label[b] < 0 || label[b] > h->nb) return fail(1, "jg_nr_set_outage_labels: label out of range");
        v[b] = (int)label[b];
    }
    NR_HIP(hipStreamSynchronize(h->stream));
    NR_HIP(jg::sync_copy(h->d_outage, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    return 0;
}

static int post_staging(jg_nr* h, size_t bytes) {
    if (bytes <= h->post_bytes) return 0;
    hipFree(h->d_post); h->d_post = nullptr; h->post_bytes = 0;
    NR_HIP(hipMalloc((void**)&h->d_post, bytes));
    h->post_bytes = bytes;
    return 0;
}

// device [rows][ld][2] -> host [batch][rows][2]
static int get_pairs(jg_nr* h, const double* src, double* dst, size_t rows) {
    std::vector<double> t(rows * h->ld * 2);
    NR_HIP(jg::sync_copy(t.data(), src, t.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    for (int b = 0; b < h->batch; ++b)
        for (size_t r = 0; r < rows; ++r) { dst[((size_t)b * rows + r) * 2] = t[(r * h->ld + b) * 2]; dst[((size_t)b * rows + r) * 2 + 1] = t[(r * h->ld + b) * 2 + 1]; }
    return 0;
}

int jg_nr_branch_quantities(jg_nr* h, double* from_pq, double* to_pq, double* series_pq, double* charging_pq,
                            double* from_i, double* to_i, double* series_i) {
    if (!h) return fail(1, "jg_nr_branch_quantities: bad argument");
    if (!h->d_bparam) return fail(1, "jg_nr_branch_quantities: call jg_nr_set_branches first");
    if (int rc = set_device(h)) return rc;
    double* host[7] = {from_pq, to_pq, series_pq, charging_pq, from_i, to_i, series_i};
    const size_t one = (size_t)h->nb * h->ld * 2 * sizeof(double);
    int want = 0;
    for (double* p : host) want += p != nullptr;
    if (!want) return 0;
    if (int rc = post_staging(h, one * want)) return rc;
    double* dev[7];
    int q = 0;
    for (int k = 0; k < 7; ++k) dev[k] = host[k] ? h->d_post + (size_t)(q++) * (one / sizeof(double)) : nullptr;
    BranchArgs a{h->d_bfrom, h->d_bto, h->d_bstatus, h->d_bparam, h->d_vm, h->d_va, h->d_outage,
                 dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], h->nb, h->ld, h->batch};
    hipLaunchKernelGGL(k_branch_quantities, dim3((h->nb + 15) / 16, h->ld / 64), dim3(64, 16), 0, h->stream, a);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    for (int k = 0; k < 7; ++k)
        if (host[k]) if (int rc = get_pairs(h, dev[k], host[k], (size_t)h->nb)) return rc;
    return 0;
}

int jg_nr_bus_injection(jg_nr* h, double* inj_pq) {
    if (!h || !inj_pq) return fail(1, "jg_nr_bus_injection: bad argument");
    if (int rc = set_device(h)) return rc;
    if (int rc = post_staging(h, (size_t)h->n * h->ld * 2 * sizeof(double))) return rc;
    launch_assemble(h, jg::GroupSel{}, false, h->d_post);          // the mismatch pass of the assembly kernel: one Ybus row walk, patches applied
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    h->jac_valid = false;
    return get_pairs(h, h->d_post, inj_pq, (size_t)h->n);
}

int jg_nr_set_screen(jg_nr* h, const double* rating) {
    if (!h) return fail(1, "jg_nr_set_screen: bad argument");
    if (!h->d_bparam) return fail(1, "jg_nr_set_screen: call jg_nr_set_branches first");
    if (int rc = set_device(h)) return rc;
    hipFree(h->d_rating); h->d_rating = nullptr;
    if (rating) {
        for (int k = 0; k < h->nb; ++k) if (!(rating[k] >= 0.0)) return fail(1, "jg_nr_set_screen: a rating is negative or not a number (0 = no limit)");
        std::string err;
        if (jg::upload(&h->d_rating, std::vector<double>(rating, rating + h->nb), err, h->stream)) return fail(2, err);
    }
    return 0;
}

static int screen_launch(jg_nr* h) {
    if (!h->d_bparam) return fail(1, "jg_nr_screen: call jg_nr_set_branches first");
    const int bchunks = (h->nb + SCREEN_BR - 1) / SCREEN_BR, vchunks = (h->n + SCREEN_BUS - 1) / SCREEN_BUS;
    if (!h->d_screen) {
        NR_HIP(hipMalloc((void**)&h->d_screen, (size_t)(bchunks + vchunks) * 4 * h->ld * sizeof(double)));
        NR_HIP(hipMalloc((void**)&h->d_screc, (size_t)h->ld * 10 * sizeof(double)));
    }
    ScreenArgs a{h->d_bfrom, h->d_bto, h->d_bstatus, h->d_bparam, h->d_rating, h->d_vm, h->d_va, h->d_outage, h->d_screen, h->d_iters, h->d_status,
                 h->d_screc, h->nb, h->n, h->ld, h->batch, bchunks, vchunks};
    hipLaunchKernelGGL(k_screen_partial, dim3(bchunks + vchunks, h->ld / 64), dim3(64, 16), 0, h->stream, a);
    hipLaunchKernelGGL(k_screen_final, dim3(h->ld / 64), dim3(64), 0, h->stream, a);
    NR_HIP(hipGetLastError());
    return 0;
}

int jg_nr_screen(jg_nr* h, double* rec) {
    if (!h || !rec) return fail(1, "jg_nr_screen: bad argument");
    if (int rc = set_device(h)) return rc;
    if (int rc = screen_launch(h)) return rc;
    NR_HIP(jg::sync_copy(rec, h->d_screc, (size_t)h->batch * 10 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int jg_nr_screen_device(jg_nr* h, double* rec_dev) {
    if (!h || !rec_dev) return fail(1, "jg_nr_screen_device: bad argument");
    if (int rc = set_device(h)) return rc;
    if (int rc = screen_launch(h)) return rc;
    NR_HIP(hipMemcpyAsync(rec_dev, h->d_screc, (size_t)h->batch * 10 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

// rows of a summary record from a range of lanes (the stragglers a pool finished: jg_nr_pack_rows_device's counterpart)
__global__ void k_screen_rows(const double* screc, const int* rows, double* dst, int lane0, int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * 10) return;
    const int i = t / 10, c = t - i * 10;
    dst[(size_t)rows[i] * 10 + c] = screc[(size_t)(lane0 + i) * 10 + c];
}

int jg_nr_screen_rows_device(jg_nr* h, double* rec_dev, int64_t lane0, int64_t count, const int32_t* rows) {
    if (!h || !rec_dev || !rows || lane0 < 0 || count < 0 || lane0 + count > h->batch) return fail(1, "jg_nr_screen_rows_device: bad argument");
    if (count == 0) return 0;
    if (int rc = set_device(h)) return rc;
    if (int rc = screen_launch(h)) return rc;
    NR_HIP(hipMemcpyAsync(h->d_itmp, rows, (size_t)count * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_screen_rows, dim3((unsigned)((count * 10 + 255) / 256)), dim3(256), 0, h->stream, h->d_screc, h->d_itmp, rec_dev, (int)lane0, (int)count);
    NR_HIP(hipGetLastError());
    NR_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int jg_nr_time_kernel(jg_nr* h, int kernel, int reps, double* mean_ms) {
    if (!h || reps < 1 || !mean_ms || kernel < 0 || kernel > 5) return fail(1, "jg_nr_time_kernel: bad argument");
    if (kernel == 4 && (!h->base || !h->d_cw)) return fail(1, "jg_nr_time_kernel: kernel 4 (first iteration on the shared base factor) needs an attached base");
    if (h->fast && kernel < 2) return fail(1, "jg_nr_time_kernel: assembly / factorisation timing would overwrite the constant factor of a fast Newton-Raphson handle");
    if (int rc = set_device(h)) return rc;
    BranchArgs ba{};
    if (kernel == 3) {                                           // power!/current! branch kernel, all seven outputs
        if (!h->d_bparam) return fail(1, "jg_nr_time_kernel: call jg_nr_set_branches first");
        const size_t one = (size_t)h->nb * h->ld * 2;
        if (int rc = post_staging(h, one * 7 * sizeof(double))) return rc;
        ba = BranchArgs{h->d_bfrom, h->d_bto, h->d_bstatus, h->d_bparam, h->d_vm, h->d_va, h->d_outage, h->d_post, h->d_post + one,
                        h->d_post + 2 * one, h->d_post + 3 * one, h->d_post + 4 * one, h->d_post + 5 * one, h->d_post + 6 * one,
                        h->nb, h->ld, h->batch};
    }
    hipEvent_t e0, e1;
    NR_HIP(hipEventCreate(&e0));
    NR_HIP(hipEventCreate(&e1));
    jg::StateUpdate none{};
    NR_HIP(hipStreamSynchronize(h->stream));
    {
        NR_HIP(hipEventRecord(e0, h->stream));
        for (int r = 0; r < reps; ++r) {
            if (kernel == 0) launch_assemble(h, jg::GroupSel{}, true, nullptr, 0, nullptr, true);     // as the iteration runs it
            else if (kernel == 1) { if (int rc = h->eng.factor(h->stream, nullptr, h->d_F, jg::GroupSel{}, h->eng.plan->S.prefactor != 0)) return fail(rc, h->eng.error); }
            else if (kernel == 3) hipLaunchKernelGGL(k_branch_quantities, dim3((h->nb + 15) / 16, h->ld / 64), dim3(64, 16), 0, h->stream, ba);
            else if (kernel == 4) {                              // the linear step of a compensated first iteration: correction + sweep pair on the shared factor (no state update)
                const jg::CompBase& cb = h->base->cb;
                jg::CompFixArgs fa{h->d_ppos, h->d_pdg, h->d_pdb, h->mp, cb.rowptr, cb.colm, cb.posrow, cb.rowtype, cb.v0, cb.th0, cb.y0, cb.Zc,
                                   h->d_F, h->eng.status, nullptr, h->ld, h->batch};
                if (h->mp > 0) jg::launch_comp_fix(fa, h->stream);
                if (int rc = cb.solve(cb.split, h->stream, h->d_F, h->d_cw, h->d_inc, h->ld, h->batch, none, jg::GroupSel{})) return fail(rc, "shared-factor sweep failed");
            }
            else if (kernel == 5) launch_assemble(h, jg::GroupSel{}, false);      // the mismatch-only pass of a compensated start
            else { if (int rc = h->eng.backsolve(h->stream, h->d_inc, none, jg::GroupSel{})) return fail(rc, h->eng.error); }
        }
        NR_HIP(hipEventRecord(e1, h->stream));
    }
    NR_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    NR_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (kernel <= 1) h->jac_valid = false;                       // the factor storage no longer holds the plain Jacobian (kernel 0 runs the
                                                                 // prefactoring assembly: leaf diagonal blocks leave as 2x2 LU factors)
    *mean_ms = (double)ms / reps;
    return 0;
}

}  // extern "C"
