// jg_engine.hip -- device kernels of the batched block-sparse LU engine (gfx950, wave64).
//
// Every kernel runs a static schedule: launch -> tasks (one workgroup each) -> steps (separated by
// a workgroup barrier) -> items.  blockDim = (64 lanes = 64 scenarios, W waves); blockIdx.y picks the
// 64-scenario group.  All structural indices are wave-uniform, so they are forced into SGPRs
// (readfirstlane) and fetched through the scalar cache; the vector memory pipe only moves
// 512-byte contiguous value segments.
#include "jg_engine.hpp"

#include <algorithm>

namespace jg {

namespace {

struct LuArgs {
    const int* task_ptr; const int* step_ptr; const int* items;
    const int* e_src; const int* e_diag; const int* t_ptr; const int* t_a; const int* t_b;
    const double* A; double* X; int* status;
    int task0; int ld;
};

struct SolveArgs {
    const int* task_ptr; const int* step_ptr; const int* items;
    const int* r_ptr; const int* r_ent; const int* r_col;   // L rows (fwd) or U rows (bwd)
    const int* diag; const int* perm;
    const double* X; double* W; const double* rhs; double* out;
    StateUpdate upd;
    int task0; int ld;
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// X(e) -= sum over terms L(a) * U(b);  diagonal -> store inverse;  lower -> scale by Dinv(col).
__global__ __launch_bounds__(256) void k_lu(LuArgs a) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const int W = blockDim.y;
    const int task = a.task0 + blockIdx.x;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int s0 = a.task_ptr[task], s1 = a.task_ptr[task + 1];
    for (int s = s0; s < s1; ++s) {
        const int i0 = a.step_ptr[s], i1 = a.step_ptr[s + 1];
        for (int idx = i0 + wave; idx < i1; idx += W) {
            const int e = uniform(a.items[idx]);
            const int src = uniform(a.e_src[e]);
            double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
            if (src >= 0) {
                const double* p = a.A + (size_t)src * 4 * ld + b;
                c00 = p[0]; c01 = p[ld]; c10 = p[2 * ld]; c11 = p[3 * ld];
            }
            const int t0 = uniform(a.t_ptr[e]), t1 = uniform(a.t_ptr[e + 1]);
#pragma unroll 2
            for (int t = t0; t < t1; ++t) {
                const double* pl = a.X + (size_t)uniform(a.t_a[t]) * 4 * ld + b;
                const double* pu = a.X + (size_t)uniform(a.t_b[t]) * 4 * ld + b;
                const double l00 = pl[0], l01 = pl[ld], l10 = pl[2 * ld], l11 = pl[3 * ld];
                const double u00 = pu[0], u01 = pu[ld], u10 = pu[2 * ld], u11 = pu[3 * ld];
                c00 -= l00 * u00 + l01 * u10;
                c01 -= l00 * u01 + l01 * u11;
                c10 -= l10 * u00 + l11 * u10;
                c11 -= l10 * u01 + l11 * u11;
            }
            const int kind = uniform(a.e_diag[e]);
            double* q = a.X + (size_t)e * 4 * ld + b;
            if (kind == -2) {                       // diagonal block: keep the inverse
                const double det = c00 * c11 - c01 * c10;
                const double r = 1.0 / det;
                if (!(fabs(det) > 0.0) || !(fabs(r) < 1.0e300)) atomicOr(a.status + b, 4);
                q[0] = c11 * r; q[ld] = -c01 * r; q[2 * ld] = -c10 * r; q[3 * ld] = c00 * r;
            } else if (kind >= 0) {                 // lower: L = acc * inv(U_jj)
                const double* d = a.X + (size_t)kind * 4 * ld + b;
                const double d00 = d[0], d01 = d[ld], d10 = d[2 * ld], d11 = d[3 * ld];
                q[0] = c00 * d00 + c01 * d10; q[ld] = c00 * d01 + c01 * d11;
                q[2 * ld] = c10 * d00 + c11 * d10; q[3 * ld] = c10 * d01 + c11 * d11;
            } else {                                // upper
                q[0] = c00; q[ld] = c01; q[2 * ld] = c10; q[3 * ld] = c11;
            }
        }
        if (s + 1 < s1) __syncthreads();
    }
}

// Forward substitution with unit-lower L:  W_k = rhs_{perm k} - sum_c L(k,c) W_c
__global__ __launch_bounds__(256) void k_fwd(SolveArgs a) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const int W = blockDim.y;
    const int task = a.task0 + blockIdx.x;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int s0 = a.task_ptr[task], s1 = a.task_ptr[task + 1];
    for (int s = s0; s < s1; ++s) {
        const int i0 = a.step_ptr[s], i1 = a.step_ptr[s + 1];
        for (int idx = i0 + wave; idx < i1; idx += W) {
            const int k = uniform(a.items[idx]);
            const int bus = uniform(a.perm[k]);
            double y0 = a.rhs[((size_t)bus * 2) * ld + b], y1 = a.rhs[((size_t)bus * 2 + 1) * ld + b];
            const int p0 = uniform(a.r_ptr[k]), p1 = uniform(a.r_ptr[k + 1]);
#pragma unroll 2
            for (int p = p0; p < p1; ++p) {
                const double* pl = a.X + (size_t)uniform(a.r_ent[p]) * 4 * ld + b;
                const double* pw = a.W + (size_t)uniform(a.r_col[p]) * 2 * ld + b;
                const double w0 = pw[0], w1 = pw[ld];
                y0 -= pl[0] * w0 + pl[ld] * w1;
                y1 -= pl[2 * ld] * w0 + pl[3 * ld] * w1;
            }
            a.W[((size_t)k * 2) * ld + b] = y0;
            a.W[((size_t)k * 2 + 1) * ld + b] = y1;
        }
        if (s + 1 < s1) __syncthreads();
    }
}

// Backward substitution: x_k = Dinv_k (W_k - sum_c U(k,c) x_c); scatter to original order; optional
// fused state update (NR: V/theta -= increment on active scenarios).
__global__ __launch_bounds__(256) void k_bwd(SolveArgs a) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const int W = blockDim.y;
    const int task = a.task0 + blockIdx.x;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int s0 = a.task_ptr[task], s1 = a.task_ptr[task + 1];
    const bool act = a.upd.active ? (a.upd.active[b] != 0) : true;
    for (int s = s0; s < s1; ++s) {
        const int i0 = a.step_ptr[s], i1 = a.step_ptr[s + 1];
        for (int idx = i0 + wave; idx < i1; idx += W) {
            const int k = uniform(a.items[idx]);
            double y0 = a.W[((size_t)k * 2) * ld + b], y1 = a.W[((size_t)k * 2 + 1) * ld + b];
            const int p0 = uniform(a.r_ptr[k]), p1 = uniform(a.r_ptr[k + 1]);
#pragma unroll 2
            for (int p = p0; p < p1; ++p) {
                const double* pu = a.X + (size_t)uniform(a.r_ent[p]) * 4 * ld + b;
                const double* pw = a.W + (size_t)uniform(a.r_col[p]) * 2 * ld + b;
                const double w0 = pw[0], w1 = pw[ld];
                y0 -= pu[0] * w0 + pu[ld] * w1;
                y1 -= pu[2 * ld] * w0 + pu[3 * ld] * w1;
            }
            const double* d = a.X + (size_t)uniform(a.diag[k]) * 4 * ld + b;
            const double x0 = d[0] * y0 + d[ld] * y1;
            const double x1 = d[2 * ld] * y0 + d[3 * ld] * y1;
            a.W[((size_t)k * 2) * ld + b] = x0;
            a.W[((size_t)k * 2 + 1) * ld + b] = x1;
            const int bus = uniform(a.perm[k]);
            a.out[((size_t)bus * 2) * ld + b] = x0;
            a.out[((size_t)bus * 2 + 1) * ld + b] = x1;
            if (a.upd.va) {
                const int fl = uniform((int)a.upd.flags[bus]);
                if (act && (fl & 1)) a.upd.va[(size_t)bus * ld + b] += a.upd.sign * x0;
                if (act && (fl & 2)) a.upd.vm[(size_t)bus * ld + b] += a.upd.sign * x1;
            }
        }
        if (s + 1 < s1) __syncthreads();
    }
}

int upload_schedule(const Schedule& s, DevSchedule& d, std::string& err) {
    d.launches = s.launches;
    if (upload(&d.task_ptr, s.task_ptr, err)) return 2;
    if (upload(&d.step_ptr, s.step_ptr, err)) return 2;
    if (upload(&d.items, s.items, err)) return 2;
    return 0;
}

void free_schedule(DevSchedule& d) {
    hipFree(d.task_ptr); hipFree(d.step_ptr); hipFree(d.items);
    d = DevSchedule();
}

}  // namespace

int Engine::create(int n, const int* rowptr, const int* col, int ld_, int policy) {
    if (ld_ <= 0 || ld_ % 64) { error = "batch leading dimension must be a positive multiple of 64"; return 1; }
    if (analyze(n, rowptr, col, policy, S)) { error = "block pattern must be structurally symmetric with a full diagonal"; return 1; }
    ld = ld_;
    std::vector<int> kind(S.n_entries);
    for (int e = 0; e < S.n_entries; ++e) kind[e] = S.e_row[e] == S.e_col[e] ? -2 : (S.e_row[e] > S.e_col[e] ? S.e_diag[e] : -1);
    if (upload(&e_src, S.e_src, error) || upload(&e_diag, kind, error) || upload(&t_ptr, S.t_ptr, error) ||
        upload(&t_a, S.t_a, error) || upload(&t_b, S.t_b, error) || upload(&l_ptr, S.l_ptr, error) ||
        upload(&l_ent, S.l_ent, error) || upload(&l_col, S.l_col, error) || upload(&u_ptr, S.u_ptr, error) ||
        upload(&u_ent, S.u_ent, error) || upload(&u_col, S.u_col, error) || upload(&diag, S.diag, error) ||
        upload(&perm, S.perm, error))
        return 2;
    if (upload_schedule(S.lu, lu, error) || upload_schedule(S.fwd, fwd, error) || upload_schedule(S.bwd, bwd, error)) return 2;
    JG_HIP(hipMalloc((void**)&X, factor_bytes()));
    JG_HIP(hipMemset(X, 0, factor_bytes()));
    JG_HIP(hipMalloc((void**)&W, (size_t)n * 2 * ld * sizeof(double)));
    JG_HIP(hipMemset(W, 0, (size_t)n * 2 * ld * sizeof(double)));
    JG_HIP(hipMalloc((void**)&status, (size_t)ld * sizeof(int)));
    JG_HIP(hipMemset(status, 0, (size_t)ld * sizeof(int)));
    return 0;
}

void Engine::destroy() {
    hipFree(e_src); hipFree(e_diag); hipFree(t_ptr); hipFree(t_a); hipFree(t_b);
    hipFree(l_ptr); hipFree(l_ent); hipFree(l_col); hipFree(u_ptr); hipFree(u_ent); hipFree(u_col);
    hipFree(diag); hipFree(perm); hipFree(X); hipFree(W); hipFree(status);
    free_schedule(lu); free_schedule(fwd); free_schedule(bwd);
    e_src = e_diag = t_ptr = t_a = t_b = l_ptr = l_ent = l_col = u_ptr = u_ent = u_col = diag = perm = status = nullptr;
    X = W = nullptr;
}

int Engine::factor(hipStream_t st, const double* A) {
    LuArgs a{lu.task_ptr, lu.step_ptr, lu.items, e_src, e_diag, t_ptr, t_a, t_b, A, X, status, 0, ld};
    for (const Launch& L : lu.launches) {
        a.task0 = L.task_begin;
        dim3 grid(L.task_end - L.task_begin, ld / 64), block(64, L.waves);
        hipLaunchKernelGGL(k_lu, grid, block, 0, st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

int Engine::solve(hipStream_t st, const double* rhs, double* out, const StateUpdate& upd) {
    SolveArgs a{fwd.task_ptr, fwd.step_ptr, fwd.items, l_ptr, l_ent, l_col, diag, perm, X, W, rhs, out, upd, 0, ld};
    for (const Launch& L : fwd.launches) {
        a.task0 = L.task_begin;
        dim3 grid(L.task_end - L.task_begin, ld / 64), block(64, L.waves);
        hipLaunchKernelGGL(k_fwd, grid, block, 0, st, a);
    }
    a.task_ptr = bwd.task_ptr; a.step_ptr = bwd.step_ptr; a.items = bwd.items;
    a.r_ptr = u_ptr; a.r_ent = u_ent; a.r_col = u_col;
    for (const Launch& L : bwd.launches) {
        a.task0 = L.task_begin;
        dim3 grid(L.task_end - L.task_begin, ld / 64), block(64, L.waves);
        hipLaunchKernelGGL(k_bwd, grid, block, 0, st, a);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

}  // namespace jg
