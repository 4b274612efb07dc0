// jg_comp.hip -- solves with ONE shared factor for a whole batch of right-hand sides, and the per-scenario rank-4 correction that turns such a
// solve into the first Newton step of an N-1 screen (jg_comp.hpp has the algebra and the reference lines it replaces).
//
// What is different from the batched engine (jg_engine.hip): the factor VALUES are the same for every scenario, so they come through the scalar
// cache (one s_load_dwordx8 per 2x2 block, already multiplied by the pivot block: a term is four multiply-adds with a scalar operand) and the only
// vector traffic is the right-hand sides -- 16 bytes per (row, scenario).  The factor being constant also allows what a refactorising iteration
// cannot afford: the sequential top of the elimination tree is ONE dense product with the explicit inverse of its Schur complement.
#include "jg_comp.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace jg {

namespace {

typedef int RecS __attribute__((ext_vector_type(16)));
typedef const RecS __attribute__((address_space(4)))* RecPtr;
typedef int SegS __attribute__((ext_vector_type(8)));
typedef const SegS __attribute__((address_space(4)))* SegPtr;
typedef const double __attribute__((address_space(4)))* CDbl;      // wave-uniform values: scalar loads
typedef const int __attribute__((address_space(4)))* CInt;
__device__ __forceinline__ RecS load_rec(const Rec* base, size_t index) { return ((RecPtr)base)[index]; }

// y = D^-1 r on the stored 2x2 LU {1/u11, u12, l (+ 4 if the rows were swapped), 1/u22} (jg_engine.hpp: diag_lu)
__device__ __forceinline__ void dsolve(double d00, double d01, double d10, double d11, double r1, double r2, double& y1, double& y2) {
    const bool sw = d10 > 2.0;
    const double l = sw ? d10 - 4.0 : d10;
    const double a = sw ? r2 : r1, b = sw ? r1 : r2;
    y2 = (b - l * a) * d11;
    y1 = (a - d01 * y2) * d00;
}

struct CSweepArgs {
    const Rec* rec; const Segment* seg;
    const double* Mc;              // compact premultiplied factor [entries][4]
    const double* rhs;             // forward: [n][ld][2], original (bus) order
    double* W;                     // [n + top rows][ld][2], pivot order
    double* out;                   // backward: [n][ld][2], bus order
    GroupSel sel; StateUpdate upd;
    int ld, seg_begin, lanes;
    int s0_base, s0_nchunks, s0_wpi, s0_rpw;
};

// one record: up to COMP_T terms  acc -= M(entry) * W[row]; every vector operand requested before the first use (jg_engine.hpp: gload16)
__device__ __forceinline__ void csweep_record(const CSweepArgs& a, const RecS& r, size_t ld, unsigned off, double& a0, double& a1) {
    static_assert(COMP_T == 6, "operand list of the wait");
    const int nt = r[3];
    d2v w[COMP_T];
#pragma unroll
    for (int t = 0; t < COMP_T; ++t) asm volatile("" : "=v"(w[t]));     // (no instruction)
#pragma unroll
    for (int t = 0; t < COMP_T; ++t)
        if (t < nt) gload16(w[t], (const char*)a.W + (size_t)r[5 + 2 * t] * ld * 16, off);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]));
#pragma unroll
    for (int t = 0; t < COMP_T; ++t)
        if (t < nt) {
            CDbl m = (CDbl)a.Mc + (size_t)r[4 + 2 * t] * 4;
            a0 = fma(-m[1], w[t].y, fma(-m[0], w[t].x, a0));
            a1 = fma(-m[3], w[t].y, fma(-m[2], w[t].x, a1));
        }
}

// One dependency level of a sweep: a wave = one row of 64 scenarios; `wpi` waves share a long row (partial sums meet in LDS, fixed order).
// 8-wave workgroups, two per 16-wave chunk of the tables (as k_bwd_level8).
template <bool BWD>
__global__ __launch_bounds__(512, 4) void k_csweep(CSweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [8][64] double2
    int base = a.s0_base, nchunks = a.s0_nchunks, wpi = a.s0_wpi, rpw = a.s0_rpw;
    if (blockIdx.y != 0) {
        const SegS sg = ((SegPtr)a.seg)[a.seg_begin + blockIdx.y];
        base = sg[0]; nchunks = sg[1]; wpi = sg[2]; rpw = sg[3];
    }
    int grp, bx;
    if (!map_block(a.sel, a.ld, nchunks * 2, grp, bx)) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)min(grp * 64 + lane, a.lanes - 1);
    const unsigned off = (unsigned)b * 16u;
    const size_t ri = (size_t)base + ((size_t)bx * 8 + wave) * rpw;
    const RecS first = load_rec(a.rec, ri);
    const int sub = wave & (wpi - 1);
    const int k = first[0], src = first[1], dg = first[2];
    double a0 = 0.0, a1 = 0.0;
    int fl = 0; bool act = true; double va = 0.0, vm = 0.0;
    if (k >= 0) {
        if (sub == 0) {
            if (BWD) {
                if (a.upd.va) {                                   // the state-update operands travel with the row's first loads
                    act = a.upd.active ? (a.upd.active[b] != 0) : true;
                    fl = uniform((int)a.upd.flags[src]);
                    va = a.upd.va[(size_t)src * ld + b]; vm = a.upd.vm[(size_t)src * ld + b];
                }
                const double2 y = load_vec(a.W, (size_t)k, b, ld);
                CDbl d = (CDbl)a.Mc + (size_t)dg * 4;
                dsolve(d[0], d[1], d[2], d[3], y.x, y.y, a0, a1);
            } else {
                const double2 f = load_vec(a.rhs, (size_t)src, b, ld);
                a0 = f.x; a1 = f.y;
            }
        }
        RecS cur = first;
        for (int j = 1; j < rpw; ++j) {
            const RecS nxt = load_rec(a.rec, ri + j);
            csweep_record(a, cur, ld, off, a0, a1);
            cur = nxt;
        }
        csweep_record(a, cur, ld, off, a0, a1);
        if (wpi > 1 && sub != 0) ((double2*)red)[(size_t)wave * 64 + lane] = double2{a0, a1};
    }
    if (wpi > 1) __syncthreads();
    if (k >= 0 && sub == 0) {
        for (int w = 1; w < wpi; ++w) { const double2 p = ((const double2*)red)[(size_t)(wave + w) * 64 + lane]; a0 += p.x; a1 += p.y; }
        store_vec(a.W, (size_t)k, b, ld, a0, a1);
        if (BWD) {
            if (!a.upd.va || act) store_vec(a.out, (size_t)src, b, ld, a0, a1);     // a finished scenario keeps its last increment
            if (a.upd.va) {
                if (act && (fl & 1)) a.upd.va[(size_t)src * ld + b] = va + a.upd.sign * a0;
                if (act && (fl & 2)) a.upd.vm[(size_t)src * ld + b] = vm + a.upd.sign * a1;
            }
        }
    }
}

// ---- the dense top: x_T = Sinv * yt_T --------------------------------------------------------------------------------------------------
// A GEMM [2 n_top x 2 n_top] x [2 n_top x scenarios] on the matrix cores (v_mfma_f64_16x16x4_f64: the one place of this path where a dense contraction IS
// the work -- 1 GFLOP per 512 scenarios at 502 top pivots; a first version fed Sinv through scalar loads and ran at 5 % of the f64 peak, bound by the
// latency of eight dependent s_loads per k step).  Workgroup = 16 rows of Sinv x the 64 scenarios of a lane group; its CTOP_WAVES waves split K and meet
// in LDS.  Sinv is stored in FRAGMENT order (k_take_sinv): [row block][k step][64 lanes], lane l = A(row l & 15, k l >> 4) of the instruction, so a
// fragment is one coalesced 512-byte load; the B fragment of scenarios 16 j .. 16 j + 15 is y(k l >> 4, scenario l & 15) read from the batch-minor scratch.
// D: lane l holds rows (l >> 4) + 4 r, r < 4, of column l & 15 (cdna_hip_programming.md: the f64 form has its own map).
constexpr int CTOP_WAVES = 4, CTOP_KPAD = 4 * CTOP_WAVES;     // K is padded to whole k steps of every wave
struct CTopArgs {
    const double* Sinv; int n_ks;                                // k steps of 4 (padded K / 4)
    const int* piv; const int* bus;
    double* W; double* out; GroupSel sel; StateUpdate upd;
    int ld, lanes, n, n_top, n_rb;
};
typedef double d4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64 * CTOP_WAVES) void k_ctop(CTopArgs a) {
    __shared__ double part[CTOP_WAVES][4][4][64];                // [wave][scenario block][register][lane]
    int grp, rb;
    if (!map_block(a.sel, a.ld, a.n_rb, grp, rb)) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const int kq = lane >> 4, col = lane & 15;
    // B operand: k = 4 ks + kq = component (kq & 1) of top row 2 ks + (kq >> 1)
    size_t boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = (size_t)min(grp * 64 + 16 * j + col, a.lanes - 1) * 2 + (kq & 1);
    const double* y = a.W + ((size_t)a.n + (kq >> 1)) * ld * 2;
    const double* sa = a.Sinv + ((size_t)rb * a.n_ks) * 64 + lane;
    d4v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = d4v{0.0, 0.0, 0.0, 0.0};
    const int per = a.n_ks / CTOP_WAVES;
    const int k0 = wave * per, k1 = k0 + per;
    constexpr int U = 4;                                         // k steps whose 5 U loads are in flight together (two waves per SIMD hide little)
    for (int ks = k0; ks < k1; ks += U) {
        double av[U], bv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = min(ks + u, k1 - 1);                   // (a repeated step is not accumulated)
            av[u] = sa[(size_t)kk * 64];
            const double* yr = y + (size_t)(2 * kk) * ld * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[u][j] = yr[boff[j]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (ks + u < k1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u][j], acc[j], 0, 0, 0);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][j][r][lane] = acc[j][r];
    __syncthreads();
    // wave j finishes scenario block j: row 16 rb + kq + 4 r = component (kq & 1) of top pivot 8 rb + (kq >> 1) + 2 r
    const int j = wave;
    const size_t b = (size_t)min(grp * 64 + 16 * j + col, a.lanes - 1);
    const bool act = a.upd.va ? (a.upd.active ? (a.upd.active[b] != 0) : true) : true;
    const int c = kq & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = 8 * rb + (kq >> 1) + 2 * r;
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < CTOP_WAVES; ++w) x += part[w][j][r][lane];
        if (t >= a.n_top) continue;
        const int k = a.piv[t], bus = a.bus[t];
        a.W[((size_t)k * ld + b) * 2 + c] = x;
        if (!a.upd.va || act) a.out[((size_t)bus * ld + b) * 2 + c] = x;
        if (a.upd.va && act) {
            const int fl = (int)a.upd.flags[bus];
            double* st = c == 0 ? a.upd.va : a.upd.vm;
            if (fl & (1 << c)) st[(size_t)bus * ld + b] += a.upd.sign * x;
        }
    }
}

// ---- set-up kernels (once per base state) --------------------------------------------------------------------------------------------------
// Mc from the plain factor of scenario 0 of an engine storage: below the diagonal Lh(i,k) D(k)^-1 (right solve on the stored 2x2 LU, as the
// factorisation tasks stage it), above it D(i)^-1 U(i,j), on it the LU itself
__global__ void k_comp_pack(const double* X, int ldx, const int* e_row, const int* e_col, const int* diag, double* Mc, int n_entries) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    auto blk = [&](int id, double (&v)[4]) {
        const double* p = X + (size_t)id * 4 * ldx;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[(size_t)2 * ldx]; v[3] = p[(size_t)2 * ldx + 1];
    };
    const int r = e_row[e], c = e_col[e];
    double x[4], d[4], o[4];
    blk(e, x);
    if (r == c) { o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; o[3] = x[3]; }
    else if (r > c) {                                            // x D^-1, row by row: z U_d = x, y L_d = z, columns swapped back
        blk(diag[c], d);
        const bool sw = d[2] > 2.0;
        const double l = sw ? d[2] - 4.0 : d[2];
        const double z1 = x[0] * d[0], w1 = x[2] * d[0];
        const double z2 = (x[1] - z1 * d[1]) * d[3], w2 = (x[3] - w1 * d[1]) * d[3];
        const double y1 = z1 - z2 * l, v1 = w1 - w2 * l;
        if (sw) { o[0] = z2; o[1] = y1; o[2] = w2; o[3] = v1; } else { o[0] = y1; o[1] = z2; o[2] = v1; o[3] = w2; }
    } else {                                                     // D^-1 x, column by column
        blk(diag[r], d);
        dsolve(d[0], d[1], d[2], d[3], x[0], x[2], o[0], o[2]);
        dsolve(d[0], d[1], d[2], d[3], x[1], x[3], o[1], o[3]);
    }
    double* m = Mc + (size_t)e * 4;
    m[0] = o[0]; m[1] = o[1]; m[2] = o[2]; m[3] = o[3];
}

// unit right-hand sides: scenario b of the batch gets e_(row[b], comp[b]) (row < 0: zero vector); rhs must be zero before
__global__ void k_unit_rhs(double* rhs, const int* row, const int* comp, int ld, int count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count || row[b] < 0) return;
    rhs[((size_t)row[b] * ld + b) * 2 + comp[b]] = 1.0;
}
__global__ void k_unit_clear(double* rhs, const int* row, const int* comp, int ld, int count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count || row[b] < 0) return;
    rhs[((size_t)row[b] * ld + b) * 2 + comp[b]] = 0.0;
}
// x_(top pivot i, component c) of scenario b = Sinv(row 2 i + c, column col0 + b), stored in the fragment order k_ctop reads (W in pivot order)
__global__ void k_take_sinv(const double* W, const int* piv, double* Sinv, int n_ks, int n_top, int ld, int col0, int count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (b >= count || i >= n_top) return;
    const double2 x = *(const double2*)(W + ((size_t)piv[i] * ld + b) * 2);
    const int kc = col0 + b, ks = kc >> 2, kq = kc & 3;
    for (int c = 0; c < 2; ++c) {
        const int r = 2 * i + c, rb = r >> 4, rr = r & 15;
        Sinv[((size_t)rb * n_ks + ks) * 64 + kq * 16 + rr] = c ? x.y : x.x;
    }
}
// scenario b solved J_0 x = e_(bus[b], comp[b]): x holds column (bus, comp) of J_0^-1.  Its rows over the Ybus neighbours j of the bus are the
// blocks Z(j, bus)(:, comp), stored at the row-CSR position of (j, bus) = the transposed position of (bus, j)
__global__ void k_take_z(const double* out, const int* bus, const int* comp, const int* rowptr, const int* colm, const int* tpos, double* Zc, int ld, int count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count || bus[b] < 0) return;
    const int a = bus[b], c = comp[b];
    for (int p = rowptr[a]; p < rowptr[a + 1]; ++p) {
        const int j = colm[p] & 0xffffff;
        const double2 x = *(const double2*)(out + ((size_t)j * ld + b) * 2);
        double* z = Zc + (size_t)tpos[p] * 4;
        z[c] = x.x; z[2 + c] = x.y;
    }
}

// ---- the per-scenario correction --------------------------------------------------------------------------------------------------------------
// One thread per scenario (a few hundred flops on <= 4 Ybus edits).  The edits of a scenario touch the rows / columns of at most two buses a, b (the
// host checked it: jg_nr_patch_ybus*); M = J_s - J_0 at the start state follows from the edits alone -- every term of equations.jl:105-144 is linear
// in G_ij, B_ij -- and so does f_s - f_0 (acPowerFlow.jl:676-680).
struct Fix4 { double m[4][4]; };
__global__ __launch_bounds__(64) void k_comp_fix(CompFixArgs a) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.lanes) return;
    if (a.active && !a.active[b]) return;
    int pos[8]; double dg[8], db[8];
    int ne = 0;
    for (int m = 0; m < a.mp; ++m) {
        const int p = a.ppos[(size_t)m * a.ld + b];
        if (p < 0) continue;
        pos[ne] = p; dg[ne] = a.pdg[(size_t)m * a.ld + b]; db[ne] = a.pdb[(size_t)m * a.ld + b]; ++ne;
    }
    if (ne == 0) return;                                         // base grid: x = J_0^-1 f_s as it stands
    int ba = a.posrow[pos[0]], bb = -1;
    for (int e = 0; e < ne; ++e) {
        const int i = a.posrow[pos[e]], j = a.colm[pos[e]] & 0xffffff;
        if (i != ba && bb < 0) bb = i;
        if (j != ba && bb < 0) bb = j;
    }
    const bool two = bb >= 0;
    if (!two) bb = ba;
    const int bus[2] = {ba, bb};
    double v[2], th[2]; int ty[2];
    for (int s = 0; s < 2; ++s) { v[s] = a.v0[bus[s]]; th[s] = a.th0[bus[s]]; ty[s] = a.rowtype[bus[s]] & 3; }
    double M[4][4] = {{0.0}}, ds1[2] = {0.0, 0.0}, ds2[2] = {0.0, 0.0}, dgii[2] = {0.0, 0.0}, dbii[2] = {0.0, 0.0};
    for (int e = 0; e < ne; ++e) {
        const int i = a.posrow[pos[e]], cm = a.colm[pos[e]], j = cm & 0xffffff, mk = cm >> 24;
        const int si = i == ba ? 0 : 1, sj = j == ba ? 0 : 1;
        double sn, cs;
        sincos(th[si] - th[sj], &sn, &cs);
        const double ac = dg[e] * cs + db[e] * sn, ad = dg[e] * sn - db[e] * cs;
        ds1[si] += v[sj] * ac; ds2[si] += v[sj] * ad;
        if (i == j) { dgii[si] = dg[e]; dbii[si] = db[e]; continue; }
        const double vi = v[si], vj = v[sj];
        M[2 * si][2 * sj] += (mk & 1) ? vi * vj * ad : 0.0;          // dP_i/dtheta_j   equations.jl:109-111
        M[2 * si][2 * sj + 1] += (mk & 2) ? vi * ac : 0.0;           // dP_i/dV_j       equations.jl:117-119
        M[2 * si + 1][2 * sj] += (mk & 4) ? -(vi * vj) * ac : 0.0;   // dQ_i/dtheta_j   equations.jl:134-136
        M[2 * si + 1][2 * sj + 1] += (mk & 8) ? vi * ad : 0.0;       // dQ_i/dV_j       equations.jl:142-144
    }
    double df[4] = {0.0, 0.0, 0.0, 0.0};
    for (int s = 0; s < (two ? 2 : 1); ++s) {
        const double vi = v[s];
        double d00 = -vi * ds2[s] - dbii[s] * (vi * vi), d01 = ds1[s] + dgii[s] * vi;      // equations.jl:105-107, 113-115
        double d10 = vi * ds1[s] - dgii[s] * (vi * vi), d11 = ds2[s] - dbii[s] * vi;       // equations.jl:130-132, 138-140
        double fp = vi * ds1[s], fq = vi * ds2[s];                                          // acPowerFlow.jl:676, 679
        if (ty[s] == 3) { d00 = d01 = d10 = d11 = 0.0; fp = fq = 0.0; }                     // identity rows of the padded Jacobian do not move
        else if (ty[s] == 2) { d01 = d10 = d11 = 0.0; fq = 0.0; }
        M[2 * s][2 * s] += d00; M[2 * s][2 * s + 1] += d01; M[2 * s + 1][2 * s] += d10; M[2 * s + 1][2 * s + 1] += d11;
        df[2 * s] = fp; df[2 * s + 1] = fq;
    }
    // S = F' J_0^-1 E: the blocks of J_0^-1 at the positions (a,a), (a,b), (b,a), (b,b) of the Ybus pattern
    double S[4][4] = {{0.0}};
    for (int si = 0; si < (two ? 2 : 1); ++si)
        for (int p = a.rowptr[bus[si]]; p < a.rowptr[bus[si] + 1]; ++p) {
            const int j = a.colm[p] & 0xffffff;
            for (int sj = 0; sj < (two ? 2 : 1); ++sj)
                if (j == bus[sj]) {
                    const double* z = a.Zc + (size_t)p * 4;
                    S[2 * si][2 * sj] = z[0]; S[2 * si][2 * sj + 1] = z[1]; S[2 * si + 1][2 * sj] = z[2]; S[2 * si + 1][2 * sj + 1] = z[3];
                }
        }
    // w = (J_0^-1 f_s)_{ab} = y0_{ab} + S df;   (I + S M) u = w;   c = M u
    double w[4], A[4][4];
    for (int r = 0; r < 4; ++r) {
        const int s = r >> 1;
        double t = (two || s == 0) ? a.y0[2 * bus[s] + (r & 1)] : 0.0;
        for (int c = 0; c < 4; ++c) t += S[r][c] * df[c];
        w[r] = t;
        for (int c = 0; c < 4; ++c) {
            double q = r == c ? 1.0 : 0.0;
            for (int k = 0; k < 4; ++k) q += S[r][k] * M[k][c];
            A[r][c] = q;
        }
    }
    // 4 x 4 LU with partial pivoting; a pivot that cancels to rounding level against what its row started from = the outage islands a part of
    // the grid (J_s singular): the scenario is marked like a zero pivot of the batched factorisation (PIVOT_EPS, jg_engine.hpp)
    double ref[4];
    for (int r = 0; r < 4; ++r) { ref[r] = 0.0; for (int c = 0; c < 4; ++c) ref[r] = fmax(ref[r], fabs(A[r][c])); }
    bool bad = false;
    for (int k = 0; k < 4; ++k) {
        int piv = k;
        for (int r = k + 1; r < 4; ++r) if (fabs(A[r][k]) > fabs(A[piv][k])) piv = r;
        if (piv != k) {
            for (int c = 0; c < 4; ++c) { const double t = A[k][c]; A[k][c] = A[piv][c]; A[piv][c] = t; }
            { const double t = w[k]; w[k] = w[piv]; w[piv] = t; }
            { const double t = ref[k]; ref[k] = ref[piv]; ref[piv] = t; }
        }
        if (!(fabs(A[k][k]) > PIVOT_EPS * ref[k])) bad = true;
        const double ip = 1.0 / A[k][k];
        for (int r = k + 1; r < 4; ++r) {
            const double l = A[r][k] * ip;
            for (int c = k + 1; c < 4; ++c) A[r][c] -= l * A[k][c];
            w[r] -= l * w[k];
        }
    }
    double u[4];
    for (int k = 3; k >= 0; --k) {
        double t = w[k];
        for (int c = k + 1; c < 4; ++c) t -= A[k][c] * u[c];
        u[k] = t / A[k][k];
    }
    if (bad || !(fabs(u[0]) < 1.0e300) || !(fabs(u[1]) < 1.0e300) || !(fabs(u[2]) < 1.0e300) || !(fabs(u[3]) < 1.0e300)) { atomicOr(a.lu_status + b, 4); return; }
    for (int s = 0; s < (two ? 2 : 1); ++s) {
        double c0 = 0.0, c1 = 0.0;
        for (int k = 0; k < 4; ++k) { c0 += M[2 * s][k] * u[k]; c1 += M[2 * s + 1][k] * u[k]; }
        double* f = a.F + ((size_t)bus[s] * a.ld + b) * 2;
        f[0] -= c0; f[1] -= c1;
    }
}

}  // namespace

void launch_comp_fix(const CompFixArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_comp_fix, dim3((unsigned)((a.lanes + 63) / 64)), dim3(64), 0, st, a);
}

int CompSweep::upload_tables(const BlockSymbolic& S, int top_cap, hipStream_t st, std::string& err) {
    build_comp_tables(S, top_cap, T);
    std::vector<int> bus(T.top.size());
    for (size_t i = 0; i < T.top.size(); ++i) bus[i] = S.perm[T.top[i]];
    if (upload(&fwd_rec, T.fwd_rec, err, st) || upload(&fwd_seg, T.fwd_seg, err, st) || upload(&bwd_rec, T.bwd_rec, err, st) || upload(&bwd_seg, T.bwd_seg, err, st) ||
        upload(&top_piv, T.top, err, st) || upload(&top_bus, bus, err, st)) return 2;
    auto launches = [](const std::vector<Segment>& segs, std::vector<DevLaunch>& out) {
        out.clear();
        size_t s = 0;
        while (s < segs.size()) {
            DevLaunch d{};
            d.seg_begin = (int)s;
            while (true) {
                d.nseg++; d.grid = std::max(d.grid, segs[s].nchunks); d.wpi_max = std::max(d.wpi_max, segs[s].wpi);
                if (segs[s++].last) break;
            }
            d.seg_end = (int)s;
            out.push_back(d);
        }
    };
    launches(T.fwd_seg, fwd);
    launches(T.bwd_seg, bwd);
    return 0;
}

void CompSweep::free_tables() {
    hipFree(fwd_rec); hipFree(fwd_seg); hipFree(bwd_rec); hipFree(bwd_seg); hipFree(top_piv); hipFree(top_bus);
    fwd_rec = bwd_rec = nullptr; fwd_seg = bwd_seg = nullptr; top_piv = top_bus = nullptr;
}

int CompBase::solve(const CompSweep& sw, hipStream_t st, const double* rhs, double* Wc, double* out, int ld, int lanes, const StateUpdate& upd, const GroupSel& sel) const {
    CSweepArgs a{sw.fwd_rec, sw.fwd_seg, Mc, rhs, Wc, out, sel, StateUpdate{}, ld, 0, lanes, 0, 0, 1, 1};
    for (const DevLaunch& L : sw.fwd) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = sw.T.fwd_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL((k_csweep<false>), dim3(grid_blocks(ld / 64, (long long)L.grid * 2), L.nseg), dim3(64, 8), 8 * 64 * sizeof(double2), st, a);
    }
    if (sw.T.n_top > 0) {
        const int n_rb = (2 * sw.T.n_top + 15) / 16;
        CTopArgs t{Sinv, lds, sw.top_piv, sw.top_bus, Wc, out, sel, upd, ld, lanes, n, sw.T.n_top, n_rb};
        hipLaunchKernelGGL(k_ctop, dim3(grid_blocks(ld / 64, n_rb)), dim3(64, CTOP_WAVES), 0, st, t);
    }
    a.rec = sw.bwd_rec; a.seg = sw.bwd_seg; a.upd = upd;
    for (const DevLaunch& L : sw.bwd) {
        a.seg_begin = L.seg_begin;
        { const Segment& g = sw.T.bwd_seg[L.seg_begin]; a.s0_base = g.rec_base; a.s0_nchunks = g.nchunks; a.s0_wpi = g.wpi; a.s0_rpw = g.rpw; }
        hipLaunchKernelGGL((k_csweep<true>), dim3(grid_blocks(ld / 64, (long long)L.grid * 2), L.nseg), dim3(64, 8), 8 * 64 * sizeof(double2), st, a);
    }
    if (hipGetLastError() != hipSuccess) return 2;
    return 0;
}

#define CB_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess) { error = std::string(#expr) + ": " + hipGetErrorString(err__); return 2; } \
    } while (0)

int CompBase::create(const BlockSymbolic& S, const double* Xsrc, int ldx, int top_cap, hipStream_t st) {
    if (S.symmetric) { error = "the shared-factor solve is built for unsymmetric plans (the Newton-Raphson Jacobian)"; return 1; }
    n = S.n; n_entries = S.n_entries;
    CB_HIP(hipGetDevice(&device));
    if (full.upload_tables(S, -1, st, error)) return 2;
    if (split.upload_tables(S, top_cap, st, error)) return 2;
    // the compact factor
    {
        int* d_row = nullptr; int* d_col = nullptr; int* d_diag = nullptr;
        if (upload(&d_row, S.e_row, error, st) || upload(&d_col, S.e_col, error, st) || upload(&d_diag, S.diag, error, st)) { hipFree(d_row); hipFree(d_col); hipFree(d_diag); return 2; }
        hipError_t e = hipMalloc((void**)&Mc, (size_t)n_entries * 4 * sizeof(double));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_comp_pack, dim3((n_entries + 255) / 256), dim3(256), 0, st, Xsrc, ldx, (const int*)d_row, (const int*)d_col, (const int*)d_diag, Mc, n_entries);
            e = hipStreamSynchronize(st);
        }
        hipFree(d_row); hipFree(d_col); hipFree(d_diag);
        CB_HIP(e);
    }
    // Sinv: unit right-hand sides on the top rows through the level-only tables (x_T = S^-1 r_T when r vanishes below the top)
    const int nt = split.T.n_top;
    if (nt > 0) {
        const int n_rb = (2 * nt + 15) / 16;
        lds = (2 * nt + CTOP_KPAD - 1) / CTOP_KPAD * (CTOP_KPAD / 4);        // k steps of 4, whole shares of the waves that split K (jg_comp.hpp: lds = k steps)
        const size_t sbytes = (size_t)n_rb * lds * 64 * sizeof(double);
        CB_HIP(hipMalloc((void**)&Sinv, sbytes));
        CB_HIP(sync_fill(Sinv, 0, sbytes, st));
        const int ldb = 512;
        double* rhs = nullptr; double* W = nullptr; double* out = nullptr; int* d_rowi = nullptr; int* d_comp = nullptr;
        const size_t vec = (size_t)n * ldb * 2 * sizeof(double);
        hipError_t e = hipMalloc((void**)&rhs, vec);
        if (e == hipSuccess) e = hipMalloc((void**)&W, vec);
        if (e == hipSuccess) e = hipMalloc((void**)&out, vec);
        if (e == hipSuccess) e = hipMalloc((void**)&d_rowi, ldb * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void**)&d_comp, ldb * sizeof(int));
        if (e == hipSuccess) e = sync_fill(rhs, 0, vec, st);
        std::vector<int> hr(ldb), hc(ldb);
        for (int c0 = 0; e == hipSuccess && c0 < 2 * nt; c0 += ldb) {
            const int cnt = std::min(ldb, 2 * nt - c0);
            for (int b = 0; b < ldb; ++b) { const int c = c0 + std::min(b, cnt - 1); hr[b] = b < cnt ? S.perm[split.T.top[c >> 1]] : -1; hc[b] = c & 1; }
            e = hipMemcpyAsync(d_rowi, hr.data(), ldb * sizeof(int), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(d_comp, hc.data(), ldb * sizeof(int), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(k_unit_rhs, dim3((ldb + 255) / 256), dim3(256), 0, st, rhs, (const int*)d_rowi, (const int*)d_comp, ldb, ldb);
            if (solve(full, st, rhs, W, out, ldb, ldb, StateUpdate{}, GroupSel{})) { e = hipErrorUnknown; break; }
            hipLaunchKernelGGL(k_take_sinv, dim3((cnt + 63) / 64, nt), dim3(64), 0, st, (const double*)W, (const int*)split.top_piv, Sinv, lds, nt, ldb, c0, cnt);
            hipLaunchKernelGGL(k_unit_clear, dim3((ldb + 255) / 256), dim3(256), 0, st, rhs, (const int*)d_rowi, (const int*)d_comp, ldb, ldb);
            e = hipStreamSynchronize(st);                        // (hr / hc are reused by the next batch)
        }
        hipFree(rhs); hipFree(W); hipFree(out); hipFree(d_rowi); hipFree(d_comp);
        CB_HIP(e);
    }
    return 0;
}

void CompBase::destroy() {
    full.free_tables(); split.free_tables();
    hipFree(Mc); hipFree(Sinv); hipFree(Zc); hipFree(v0); hipFree(th0); hipFree(f0); hipFree(y0); hipFree(p0); hipFree(q0);
    hipFree(rowptr); hipFree(colm); hipFree(posrow); hipFree(tpos); hipFree(rowtype);
    Mc = Sinv = Zc = v0 = th0 = f0 = y0 = p0 = q0 = nullptr;
    rowptr = colm = posrow = tpos = rowtype = nullptr;
}

// J_0^-1 on the Ybus pattern: one solve per (bus, component) with the split tables, 512 columns per batch
int comp_form_z(CompBase& B, hipStream_t st) {
    const int n = B.n, ldb = 512;
    std::string& error = B.error;
    CB_HIP(hipMalloc((void**)&B.Zc, (size_t)B.nnz * 4 * sizeof(double)));
    CB_HIP(sync_fill(B.Zc, 0, (size_t)B.nnz * 4 * sizeof(double), st));
    double* rhs = nullptr; double* W = nullptr; double* out = nullptr; int* d_bus = nullptr; int* d_comp = nullptr;
    const size_t vec = (size_t)n * ldb * 2 * sizeof(double);
    hipError_t e = hipMalloc((void**)&rhs, vec);
    if (e == hipSuccess) e = hipMalloc((void**)&W, B.scratch_rows(B.split) * ldb * 2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&out, vec);
    if (e == hipSuccess) e = hipMalloc((void**)&d_bus, ldb * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&d_comp, ldb * sizeof(int));
    if (e == hipSuccess) e = sync_fill(rhs, 0, vec, st);
    if (e == hipSuccess) e = sync_fill(W, 0, B.scratch_rows(B.split) * ldb * 2 * sizeof(double), st);      // the spare top row stays zero
    std::vector<int> hb(ldb), hc(ldb);
    for (int c0 = 0; e == hipSuccess && c0 < 2 * n; c0 += ldb) {
        const int cnt = std::min(ldb, 2 * n - c0);
        for (int b = 0; b < ldb; ++b) { hb[b] = b < cnt ? (c0 + b) >> 1 : -1; hc[b] = (c0 + b) & 1; }
        e = hipMemcpyAsync(d_bus, hb.data(), ldb * sizeof(int), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_comp, hc.data(), ldb * sizeof(int), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_unit_rhs, dim3((ldb + 255) / 256), dim3(256), 0, st, rhs, (const int*)d_bus, (const int*)d_comp, ldb, ldb);
        if (B.solve(B.split, st, rhs, W, out, ldb, ldb, StateUpdate{}, GroupSel{})) { e = hipErrorUnknown; break; }
        hipLaunchKernelGGL(k_take_z, dim3((ldb + 255) / 256), dim3(256), 0, st, (const double*)out, (const int*)d_bus, (const int*)d_comp, (const int*)B.rowptr, (const int*)B.colm,
                           (const int*)B.tpos, B.Zc, ldb, ldb);
        hipLaunchKernelGGL(k_unit_clear, dim3((ldb + 255) / 256), dim3(256), 0, st, rhs, (const int*)d_bus, (const int*)d_comp, ldb, ldb);
        e = hipStreamSynchronize(st);
    }
    hipFree(rhs); hipFree(W); hipFree(out); hipFree(d_bus); hipFree(d_comp);
    CB_HIP(e);
    return 0;
}

}  // namespace jg
