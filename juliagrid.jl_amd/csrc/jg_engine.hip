// jg_engine.hip -- device kernels of the batched block-sparse LU engine (gfx950, wave64).
//
// Factorisation A = Lh * inv(D) * U (Lh unscaled, D kept as 2x2 LU factors) with the forward elimination of the right-hand
// side fused into the same launches, then one backward sweep.  Every launch replays one dependency
// level of the static schedule (jg_symbolic): blockDim = (64 lanes = 64 scenarios, W waves),
// blockIdx.y = 64-scenario group.  `wpi` waves cooperate on one item: its update list is dealt out
// round-robin, partial sums meet in LDS and are added in a fixed order (run-to-run deterministic).
// All structural data is wave-uniform: one 32-byte descriptor per item through the scalar cache;
// the vector memory pipe only moves 512-byte contiguous value segments.
#include "jg_engine.hpp"

#include <algorithm>

namespace jg {

namespace {

struct FactArgs {
    const ItemDesc* desc; const int* ta; const int* td; const int* tb;
    const double* A; const double* rhs; double* X; double* W; int* status; const int* group_active;
    int item_begin, item_end, wpi, rounds, ld;
};

struct BwdArgs {
    const ItemDesc* desc; const int* u_ent; const int* u_col;
    const double* X; double* W; double* out; const int* group_active;
    StateUpdate upd;
    int item_begin, item_end, wpi, rounds, ld;
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct Blk { double v00, v01, v10, v11; };

__device__ __forceinline__ Blk load_blk(const double* p, size_t ld) { return Blk{p[0], p[ld], p[2 * ld], p[3 * ld]}; }

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

template <int UNROLL>
__device__ __forceinline__ void lu_terms(const FactArgs& a, int t0, int t1, int stride, size_t b, size_t ld, Blk& c) {
    int t = t0;
    for (; t + (UNROLL - 1) * stride < t1; t += UNROLL * stride) {
        Blk l[UNROLL], d[UNROLL], u[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            l[k] = load_blk(a.X + (size_t)uniform(a.ta[t + k * stride]) * 4 * ld + b, ld);
            d[k] = load_blk(a.X + (size_t)uniform(a.td[t + k * stride]) * 4 * ld + b, ld);
            u[k] = load_blk(a.X + (size_t)uniform(a.tb[t + k * stride]) * 4 * ld + b, ld);
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) term3(c, l[k], d[k], u[k]);
    }
    for (; t < t1; t += stride) {
        const Blk l = load_blk(a.X + (size_t)uniform(a.ta[t]) * 4 * ld + b, ld);
        const Blk d = load_blk(a.X + (size_t)uniform(a.td[t]) * 4 * ld + b, ld);
        const Blk u = load_blk(a.X + (size_t)uniform(a.tb[t]) * 4 * ld + b, ld);
        term3(c, l, d, u);
    }
}

// y -= Lh(a) * Dinv(d) * y_c
template <int UNROLL>
__device__ __forceinline__ void rhs_terms(const FactArgs& a, int t0, int t1, int stride, size_t b, size_t ld, double& y0, double& y1) {
    int t = t0;
    for (; t + (UNROLL - 1) * stride < t1; t += UNROLL * stride) {
        Blk l[UNROLL], d[UNROLL]; double w0[UNROLL], w1[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            l[k] = load_blk(a.X + (size_t)uniform(a.ta[t + k * stride]) * 4 * ld + b, ld);
            d[k] = load_blk(a.X + (size_t)uniform(a.td[t + k * stride]) * 4 * ld + b, ld);
            const double* pw = a.W + (size_t)uniform(a.tb[t + k * stride]) * 2 * ld + b;
            w0[k] = pw[0]; w1[k] = pw[ld];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            double z0, z1;
            dsolve(d[k], w0[k], w1[k], z0, z1);
            y0 -= l[k].v00 * z0 + l[k].v01 * z1;
            y1 -= l[k].v10 * z0 + l[k].v11 * z1;
        }
    }
    for (; t < t1; t += stride) {
        const Blk l = load_blk(a.X + (size_t)uniform(a.ta[t]) * 4 * ld + b, ld);
        const Blk d = load_blk(a.X + (size_t)uniform(a.td[t]) * 4 * ld + b, ld);
        const double* pw = a.W + (size_t)uniform(a.tb[t]) * 2 * ld + b;
        const double w0 = pw[0], w1 = pw[ld];
        double z0, z1;
        dsolve(d, w0, w1, z0, z1);
        y0 -= l.v00 * z0 + l.v01 * z1;
        y1 -= l.v10 * z0 + l.v11 * z1;
    }
}

__device__ __forceinline__ void fact_finish(const FactArgs& a, int kind, int id, size_t b, size_t ld, const Blk& c) {
    if (kind == 3) {
        a.W[((size_t)id * 2) * ld + b] = c.v00;
        a.W[((size_t)id * 2 + 1) * ld + b] = c.v01;
        return;
    }
    double* q = a.X + (size_t)id * 4 * ld + b;
    if (kind == 2) {                            // diagonal block: 2x2 LU with in-block partial pivoting
        const bool sw = fabs(c.v10) > fabs(c.v00);
        const double u11 = sw ? c.v10 : c.v00, u12 = sw ? c.v11 : c.v01;
        const double o21 = sw ? c.v00 : c.v10, o22 = sw ? c.v01 : c.v11;
        const double iu11 = 1.0 / u11;
        const double l = o21 * iu11;
        const double u22 = o22 - l * u12;
        const double iu22 = 1.0 / u22;
        if (!(fabs(u11) > 0.0) || !(fabs(u22) > 0.0) || !(fabs(iu11) < 1.0e300) || !(fabs(iu22) < 1.0e300)) atomicOr(a.status + b, 4);
        q[0] = iu11; q[ld] = u12; q[2 * ld] = sw ? l + 4.0 : l; q[3 * ld] = iu22;
    } else {
        q[0] = c.v00; q[ld] = c.v01; q[2 * ld] = c.v10; q[3 * ld] = c.v11;
    }
}

// One dependency level of  A = Lh inv(D) U  and  y = (Lh inv(D))^-1 rhs.
template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 1024 : 256) void k_fact(FactArgs a) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // SPLIT: [W][4][64]
    if (a.group_active && !a.group_active[blockIdx.y]) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int wpi = SPLIT ? a.wpi : 1;
    const int slots = blockDim.y / wpi;
    const int slot = wave / wpi, sub = wave - slot * wpi;
    for (int r = 0; r < a.rounds; ++r) {
        const int idx = a.item_begin + (blockIdx.x * a.rounds + r) * slots + slot;
        const bool valid = idx < a.item_end;
        Blk c{0.0, 0.0, 0.0, 0.0};
        int kind = 0, id = 0;
        if (valid) {
            const ItemDesc* dp = a.desc + idx;
            kind = uniform(dp->kind); id = uniform(dp->id);
            const int src = uniform(dp->src), t0 = uniform(dp->t0), t1 = uniform(dp->t1);
            if (kind == 3) {
                if (sub == 0) { c.v00 = a.rhs[((size_t)src * 2) * ld + b]; c.v01 = a.rhs[((size_t)src * 2 + 1) * ld + b]; }
                rhs_terms<4>(a, t0 + sub, t1, wpi, b, ld, c.v00, c.v01);
            } else {
                if (sub == 0 && src >= 0) c = load_blk(a.A + (size_t)src * 4 * ld + b, ld);
                lu_terms<3>(a, t0 + sub, t1, wpi, b, ld, c);
            }
            if (SPLIT && sub != 0) {
                double* q = red + (size_t)wave * 256 + lane;
                q[0] = c.v00; q[64] = c.v01; q[128] = c.v10; q[192] = c.v11;
            }
        }
        if (SPLIT) __syncthreads();
        if (valid && sub == 0) {
            if (SPLIT)
                for (int w = 1; w < wpi; ++w) {
                    const double* q = red + (size_t)(wave + w) * 256 + lane;
                    c.v00 += q[0]; c.v01 += q[64]; c.v10 += q[128]; c.v11 += q[192];
                }
            fact_finish(a, kind, id, b, ld, c);
        }
        if (SPLIT && r + 1 < a.rounds) __syncthreads();
    }
}

template <int UNROLL>
__device__ __forceinline__ void row_terms(const BwdArgs& a, int p0, int p1, int stride, size_t b, size_t ld, double& y0, double& y1) {
    int p = p0;
    for (; p + (UNROLL - 1) * stride < p1; p += UNROLL * stride) {
        Blk m[UNROLL]; double w0[UNROLL], w1[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            m[k] = load_blk(a.X + (size_t)uniform(a.u_ent[p + k * stride]) * 4 * ld + b, ld);
            const double* pw = a.W + (size_t)uniform(a.u_col[p + k * stride]) * 2 * ld + b;
            w0[k] = pw[0]; w1[k] = pw[ld];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            y0 -= m[k].v00 * w0[k] + m[k].v01 * w1[k];
            y1 -= m[k].v10 * w0[k] + m[k].v11 * w1[k];
        }
    }
    for (; p < p1; p += stride) {
        const Blk m = load_blk(a.X + (size_t)uniform(a.u_ent[p]) * 4 * ld + b, ld);
        const double* pw = a.W + (size_t)uniform(a.u_col[p]) * 2 * ld + b;
        const double w0 = pw[0], w1 = pw[ld];
        y0 -= m.v00 * w0 + m.v01 * w1;
        y1 -= m.v10 * w0 + m.v11 * w1;
    }
}

// One backward item (pivot row) handled by `wpi` cooperating waves; contains the two workgroup barriers of
// the LDS reduction when SPLIT.
template <bool SPLIT>
__device__ __forceinline__ void bwd_item(const BwdArgs& a, double* red, int idx, bool valid, int wave, int sub, int wpi,
                                         size_t b, size_t ld, bool act, int lane) {
    double y0 = 0.0, y1 = 0.0;
    int k = 0, bus = 0, dg = 0;
    if (valid) {
        const ItemDesc* dp = a.desc + idx;
        k = uniform(dp->id); bus = uniform(dp->src); dg = uniform(dp->aux);
        const int p0 = uniform(dp->t0), p1 = uniform(dp->t1);
        if (sub == 0) { y0 = a.W[((size_t)k * 2) * ld + b]; y1 = a.W[((size_t)k * 2 + 1) * ld + b]; }
        row_terms<4>(a, p0 + sub, p1, wpi, b, ld, y0, y1);
        if (SPLIT && sub != 0) { red[(size_t)wave * 128 + lane] = y0; red[(size_t)wave * 128 + 64 + lane] = y1; }
    }
    if (SPLIT) __syncthreads();
    if (valid && sub == 0) {
        if (SPLIT)
            for (int w = 1; w < wpi; ++w) { y0 += red[(size_t)(wave + w) * 128 + lane]; y1 += red[(size_t)(wave + w) * 128 + 64 + lane]; }
        const Blk d = load_blk(a.X + (size_t)dg * 4 * ld + b, ld);
        double x0, x1;
        dsolve(d, y0, y1, x0, x1);
        a.W[((size_t)k * 2) * ld + b] = x0;
        a.W[((size_t)k * 2 + 1) * ld + b] = x1;
        a.out[((size_t)bus * 2) * ld + b] = x0;
        a.out[((size_t)bus * 2 + 1) * ld + b] = x1;
        if (a.upd.va) {
            const int fl = uniform((int)a.upd.flags[bus]);
            if (act && (fl & 1)) a.upd.va[(size_t)bus * ld + b] += a.upd.sign * x0;
            if (act && (fl & 2)) a.upd.vm[(size_t)bus * ld + b] += a.upd.sign * x1;
        }
    }
}

// Backward sweep: x_k = Dinv_k (y_k - sum_c U(k,c) x_c), scattered to original order; optional fused
// state update (Newton-Raphson: V/theta -= increment on active scenarios).  One dependency level per launch.
template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 1024 : 256) void k_bwd(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // SPLIT: [W][2][64]
    if (a.group_active && !a.group_active[blockIdx.y]) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int wpi = SPLIT ? a.wpi : 1;
    const int slots = blockDim.y / wpi;
    const int slot = wave / wpi, sub = wave - slot * wpi;
    const bool act = a.upd.active ? (a.upd.active[b] != 0) : true;
    for (int r = 0; r < a.rounds; ++r) {
        const int idx = a.item_begin + (blockIdx.x * a.rounds + r) * slots + slot;
        bwd_item<SPLIT>(a, red, idx, idx < a.item_end, wave, sub, wpi, b, ld, act, lane);
        if (SPLIT && r + 1 < a.rounds) __syncthreads();
    }
}

// The narrow leading levels of the backward sweep (the dense tail of the elimination order: 1-4 rows per
// level) in ONE launch: one 16-wave workgroup per scenario group walks the levels as barrier-separated steps.
__global__ __launch_bounds__(1024) void k_bwd_fused(BwdArgs a, const int* steps, int n_steps) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [16][2][64]
    if (a.group_active && !a.group_active[blockIdx.y]) return;
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const bool act = a.upd.active ? (a.upd.active[b] != 0) : true;
    for (int s = 0; s < n_steps; ++s) {
        const int ib = uniform(steps[3 * s]), ie = uniform(steps[3 * s + 1]), wpi = uniform(steps[3 * s + 2]);
        const int slots = 16 / wpi;
        const int slot = wave / wpi, sub = wave - slot * wpi;
        for (int base = ib; base < ie; base += slots) {
            const int idx = base + slot;
            bwd_item<true>(a, red, idx, idx < ie, wave, sub, wpi, b, ld, act, lane);
            __syncthreads();                       // LDS reuse + makes this level's x visible to the next step
        }
    }
}

void flatten(const Schedule& s, std::vector<DevLaunch>& out, std::vector<int>* step_table = nullptr) {
    out.clear();
    for (const Launch& L : s.launches) {
        DevLaunch d;
        if (L.fused && step_table) {
            d.item_begin = L.item_begin; d.item_end = L.item_end; d.waves = L.waves; d.wpi = 1; d.rounds = 1; d.grid = 1;
            d.fused = 1; d.step0 = (int)step_table->size() / 3;
            for (int st = s.task_ptr[L.task_begin]; st < s.task_ptr[L.task_begin + 1]; ++st) {
                step_table->push_back(s.step_ptr[st]); step_table->push_back(s.step_ptr[st + 1]); step_table->push_back(s.step_wpi[st]);
            }
            d.n_steps = (int)step_table->size() / 3 - d.step0;
            out.push_back(d);
            continue;
        }
        d.item_begin = L.item_begin; d.item_end = L.item_end;
        d.waves = L.waves; d.wpi = L.wpi;
        const int slots = L.waves / L.wpi;
        d.rounds = std::max(1, L.chunk / slots);
        const int per_wg = slots * d.rounds;
        d.grid = (L.item_end - L.item_begin + per_wg - 1) / per_wg;
        out.push_back(d);
    }
}

}  // namespace

int Engine::create(int n, const int* rowptr, const int* col, int ld_, int policy) {
    if (ld_ <= 0 || ld_ % 64) { error = "batch leading dimension must be a positive multiple of 64"; return 1; }
    if (analyze(n, rowptr, col, policy, S)) { error = "block pattern must be structurally symmetric with a full diagonal"; return 1; }
    ld = ld_;
    const int nE = S.n_entries;
    // terms: LU terms, then rhs-row terms (Lh entry, diagonal of the column, column pivot)
    std::vector<int> va(S.t_a), vd(S.t_d), vb(S.t_b);
    const int rhs_base = (int)va.size();
    for (int r = 0; r < n; ++r)
        for (int p = S.l_ptr[r]; p < S.l_ptr[r + 1]; ++p) { va.push_back(S.l_ent[p]); vd.push_back(S.diag[S.l_col[p]]); vb.push_back(S.l_col[p]); }
    std::vector<ItemDesc> fd(S.fact.items.size());
    for (size_t i = 0; i < fd.size(); ++i) {
        const int it = S.fact.items[i];
        ItemDesc d{};
        if (it < nE) {
            d.kind = S.e_row[it] == S.e_col[it] ? 2 : (S.e_row[it] > S.e_col[it] ? 1 : 0);
            d.id = it; d.src = S.e_src[it]; d.t0 = S.t_ptr[it]; d.t1 = S.t_ptr[it + 1];
        } else {
            const int k = it - nE;
            d.kind = 3; d.id = k; d.src = S.perm[k]; d.t0 = rhs_base + S.l_ptr[k]; d.t1 = rhs_base + S.l_ptr[k + 1];
        }
        fd[i] = d;
    }
    std::vector<ItemDesc> bd(S.bwd.items.size());
    for (size_t i = 0; i < bd.size(); ++i) {
        const int k = S.bwd.items[i];
        ItemDesc d{};
        d.kind = 4; d.id = k; d.src = S.perm[k]; d.t0 = S.u_ptr[k]; d.t1 = S.u_ptr[k + 1]; d.aux = S.diag[k];
        bd[i] = d;
    }
    flatten(S.fact, fact);
    std::vector<int> bsteps;
    flatten(S.bwd, bwd, &bsteps);
    if (upload(&bwd_steps, bsteps, error)) return 2;
    if (upload(&fact_desc, fd, error) || upload(&bwd_desc, bd, error) || upload(&ta, va, error) || upload(&td, vd, error) ||
        upload(&tb, vb, error) || upload(&u_ent, S.u_ent, error) || upload(&u_col, S.u_col, error))
        return 2;
    JG_HIP(hipMalloc((void**)&X, factor_bytes()));
    JG_HIP(hipMemset(X, 0, factor_bytes()));
    JG_HIP(hipMalloc((void**)&W, (size_t)n * 2 * ld * sizeof(double)));
    JG_HIP(hipMemset(W, 0, (size_t)n * 2 * ld * sizeof(double)));
    JG_HIP(hipMalloc((void**)&status, (size_t)ld * sizeof(int)));
    JG_HIP(hipMemset(status, 0, (size_t)ld * sizeof(int)));
    return 0;
}

void Engine::destroy() {
    hipFree(fact_desc); hipFree(bwd_desc); hipFree(ta); hipFree(td); hipFree(tb); hipFree(u_ent); hipFree(u_col); hipFree(bwd_steps); bwd_steps = nullptr;
    hipFree(X); hipFree(W); hipFree(status);
    fact_desc = bwd_desc = nullptr;
    ta = td = tb = u_ent = u_col = status = nullptr;
    X = W = nullptr;
}

int Engine::factor(hipStream_t st, const double* A, const double* rhs, const int* group_active) {
    FactArgs a{fact_desc, ta, td, tb, A, rhs, X, W, status, group_active, 0, 0, 1, 1, ld};
    for (const DevLaunch& L : fact) {
        a.item_begin = L.item_begin; a.item_end = L.item_end; a.wpi = L.wpi; a.rounds = L.rounds;
        dim3 grid(L.grid, ld / 64), block(64, L.waves);
        if (L.wpi == 1) hipLaunchKernelGGL(k_fact<false>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(k_fact<true>, grid, block, (size_t)L.waves * 256 * sizeof(double), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

int Engine::backsolve(hipStream_t st, double* out, const StateUpdate& upd, const int* group_active) {
    BwdArgs a{bwd_desc, u_ent, u_col, X, W, out, group_active, upd, 0, 0, 1, 1, ld};
    for (const DevLaunch& L : bwd) {
        a.item_begin = L.item_begin; a.item_end = L.item_end; a.wpi = L.wpi; a.rounds = L.rounds;
        dim3 grid(L.grid, ld / 64), block(64, L.waves);
        if (L.fused) { hipLaunchKernelGGL(k_bwd_fused, grid, block, (size_t)16 * 128 * sizeof(double), st, a, bwd_steps + 3 * L.step0, L.n_steps); continue; }
        if (L.wpi == 1) hipLaunchKernelGGL(k_bwd<false>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(k_bwd<true>, grid, block, (size_t)L.waves * 128 * sizeof(double), st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

}  // namespace jg
