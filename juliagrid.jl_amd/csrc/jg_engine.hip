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

struct Blk { double v00, v01, v10, v11; };

__device__ __forceinline__ Blk load_blk(const double* p, size_t ld) { return Blk{p[0], p[ld], p[2 * ld], p[3 * ld]}; }

// acc -= L(a) * U(b) over terms t0, t0+stride, ... < t1 ; four terms (64 loads of 512 B per wave) in flight
__device__ __forceinline__ void lu_terms(const LuArgs& a, int t0, int t1, int stride, size_t b, size_t ld, Blk& c) {
    int t = t0;
    for (; t + 3 * stride < t1; t += 4 * stride) {
        Blk l[4], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            l[k] = load_blk(a.X + (size_t)uniform(a.t_a[t + k * stride]) * 4 * ld + b, ld);
            u[k] = load_blk(a.X + (size_t)uniform(a.t_b[t + k * stride]) * 4 * ld + b, ld);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c.v00 -= l[k].v00 * u[k].v00 + l[k].v01 * u[k].v10;
            c.v01 -= l[k].v00 * u[k].v01 + l[k].v01 * u[k].v11;
            c.v10 -= l[k].v10 * u[k].v00 + l[k].v11 * u[k].v10;
            c.v11 -= l[k].v10 * u[k].v01 + l[k].v11 * u[k].v11;
        }
    }
    for (; t < t1; t += stride) {
        const Blk l = load_blk(a.X + (size_t)uniform(a.t_a[t]) * 4 * ld + b, ld);
        const Blk u = load_blk(a.X + (size_t)uniform(a.t_b[t]) * 4 * ld + b, ld);
        c.v00 -= l.v00 * u.v00 + l.v01 * u.v10;
        c.v01 -= l.v00 * u.v01 + l.v01 * u.v11;
        c.v10 -= l.v10 * u.v00 + l.v11 * u.v10;
        c.v11 -= l.v10 * u.v01 + l.v11 * u.v11;
    }
}

// diagonal -> store inverse;  lower -> scale by Dinv(col);  upper -> store
__device__ __forceinline__ void lu_finish(const LuArgs& a, int e, size_t b, size_t ld, const Blk& c) {
    const int kind = uniform(a.e_diag[e]);
    double* q = a.X + (size_t)e * 4 * ld + b;
    if (kind == -2) {
        const double det = c.v00 * c.v11 - c.v01 * c.v10;
        const double r = 1.0 / det;
        if (!(fabs(det) > 0.0) || !(fabs(r) < 1.0e300)) atomicOr(a.status + b, 4);
        q[0] = c.v11 * r; q[ld] = -c.v01 * r; q[2 * ld] = -c.v10 * r; q[3 * ld] = c.v00 * r;
    } else if (kind >= 0) {
        const Blk d = load_blk(a.X + (size_t)kind * 4 * ld + b, ld);
        q[0] = c.v00 * d.v00 + c.v01 * d.v10; q[ld] = c.v00 * d.v01 + c.v01 * d.v11;
        q[2 * ld] = c.v10 * d.v00 + c.v11 * d.v10; q[3 * ld] = c.v10 * d.v01 + c.v11 * d.v11;
    } else {
        q[0] = c.v00; q[ld] = c.v01; q[2 * ld] = c.v10; q[3 * ld] = c.v11;
    }
}

__device__ __forceinline__ Blk lu_source(const LuArgs& a, int e, size_t b, size_t ld) {
    const int src = uniform(a.e_src[e]);
    if (src < 0) return Blk{0.0, 0.0, 0.0, 0.0};
    return load_blk(a.A + (size_t)src * 4 * ld + b, ld);
}

// One wave per item; a task may chain several barrier-separated steps.
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
            Blk c = lu_source(a, e, b, ld);
            lu_terms(a, uniform(a.t_ptr[e]), uniform(a.t_ptr[e + 1]), 1, b, ld, c);
            lu_finish(a, e, b, ld, c);
        }
        if (s + 1 < s1) __syncthreads();
    }
}

// `wpi` waves share one item: wave `sub` takes terms sub, sub+wpi, ...; partial sums meet in LDS and
// are added in a fixed order (run-to-run deterministic).  Single-step tasks of blockDim.y/wpi items.
__global__ __launch_bounds__(1024) void k_lu_split(LuArgs a, int wpi) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [W][4][64]
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const int task = a.task0 + blockIdx.x;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int s = a.task_ptr[task];
    const int i0 = a.step_ptr[s], i1 = a.step_ptr[s + 1];
    const int slot = wave / wpi, sub = wave - slot * wpi;
    const int idx = i0 + slot;
    const bool valid = idx < i1;
    int e = 0;
    Blk c{0.0, 0.0, 0.0, 0.0};
    if (valid) {
        e = uniform(a.items[idx]);
        if (sub == 0) c = lu_source(a, e, b, ld);
        lu_terms(a, uniform(a.t_ptr[e]) + sub, uniform(a.t_ptr[e + 1]), wpi, b, ld, c);
        if (sub != 0) {
            double* r = red + (size_t)wave * 256 + lane;
            r[0] = c.v00; r[64] = c.v01; r[128] = c.v10; r[192] = c.v11;
        }
    }
    __syncthreads();
    if (valid && sub == 0) {
        for (int w = 1; w < wpi; ++w) {
            const double* r = red + (size_t)(wave + w) * 256 + lane;
            c.v00 += r[0]; c.v01 += r[64]; c.v10 += r[128]; c.v11 += r[192];
        }
        lu_finish(a, e, b, ld, c);
    }
}

// y -= sum over row entries X(ent) * W(col), entries p0, p0+stride, ... < p1; four in flight
__device__ __forceinline__ void row_terms(const SolveArgs& a, int p0, int p1, int stride, size_t b, size_t ld, double& y0, double& y1) {
    int p = p0;
    for (; p + 3 * stride < p1; p += 4 * stride) {
        Blk m[4]; double w0[4], w1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = load_blk(a.X + (size_t)uniform(a.r_ent[p + k * stride]) * 4 * ld + b, ld);
            const double* pw = a.W + (size_t)uniform(a.r_col[p + k * stride]) * 2 * ld + b;
            w0[k] = pw[0]; w1[k] = pw[ld];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            y0 -= m[k].v00 * w0[k] + m[k].v01 * w1[k];
            y1 -= m[k].v10 * w0[k] + m[k].v11 * w1[k];
        }
    }
    for (; p < p1; p += stride) {
        const Blk m = load_blk(a.X + (size_t)uniform(a.r_ent[p]) * 4 * ld + b, ld);
        const double* pw = a.W + (size_t)uniform(a.r_col[p]) * 2 * ld + b;
        const double w0 = pw[0], w1 = pw[ld];
        y0 -= m.v00 * w0 + m.v01 * w1;
        y1 -= m.v10 * w0 + m.v11 * w1;
    }
}

__device__ __forceinline__ void bwd_finish(const SolveArgs& a, int k, size_t b, size_t ld, double y0, double y1, bool act) {
    const Blk d = load_blk(a.X + (size_t)uniform(a.diag[k]) * 4 * ld + b, ld);
    const double x0 = d.v00 * y0 + d.v01 * y1;
    const double x1 = d.v10 * y0 + d.v11 * y1;
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

// Forward substitution with unit-lower L:  W_k = rhs_{perm k} - sum_c L(k,c) W_c         (BWD = false)
// Backward substitution: x_k = Dinv_k (W_k - sum_c U(k,c) x_c), scatter to original order, optional
// fused state update (NR: V/theta -= increment on active scenarios)                      (BWD = true)
template <bool BWD>
__global__ __launch_bounds__(256) void k_tri(SolveArgs a) {
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const int W = blockDim.y;
    const int task = a.task0 + blockIdx.x;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int s0 = a.task_ptr[task], s1 = a.task_ptr[task + 1];
    const bool act = (BWD && a.upd.active) ? (a.upd.active[b] != 0) : true;
    for (int s = s0; s < s1; ++s) {
        const int i0 = a.step_ptr[s], i1 = a.step_ptr[s + 1];
        for (int idx = i0 + wave; idx < i1; idx += W) {
            const int k = uniform(a.items[idx]);
            double y0, y1;
            if (BWD) {
                y0 = a.W[((size_t)k * 2) * ld + b]; y1 = a.W[((size_t)k * 2 + 1) * ld + b];
            } else {
                const int bus = uniform(a.perm[k]);
                y0 = a.rhs[((size_t)bus * 2) * ld + b]; y1 = a.rhs[((size_t)bus * 2 + 1) * ld + b];
            }
            row_terms(a, uniform(a.r_ptr[k]), uniform(a.r_ptr[k + 1]), 1, b, ld, y0, y1);
            if (BWD) bwd_finish(a, k, b, ld, y0, y1, act);
            else { a.W[((size_t)k * 2) * ld + b] = y0; a.W[((size_t)k * 2 + 1) * ld + b] = y1; }
        }
        if (s + 1 < s1) __syncthreads();
    }
}

template <bool BWD>
__global__ __launch_bounds__(1024) void k_tri_split(SolveArgs a, int wpi) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [W][2][64]
    const int lane = threadIdx.x;
    const int wave = uniform(threadIdx.y);
    const int task = a.task0 + blockIdx.x;
    const size_t ld = (size_t)a.ld;
    const size_t b = (size_t)blockIdx.y * 64 + lane;
    const int s = a.task_ptr[task];
    const int i0 = a.step_ptr[s], i1 = a.step_ptr[s + 1];
    const int slot = wave / wpi, sub = wave - slot * wpi;
    const int idx = i0 + slot;
    const bool valid = idx < i1;
    const bool act = (BWD && a.upd.active) ? (a.upd.active[b] != 0) : true;
    int k = 0;
    double y0 = 0.0, y1 = 0.0;
    if (valid) {
        k = uniform(a.items[idx]);
        if (sub == 0) {
            if (BWD) { y0 = a.W[((size_t)k * 2) * ld + b]; y1 = a.W[((size_t)k * 2 + 1) * ld + b]; }
            else {
                const int bus = uniform(a.perm[k]);
                y0 = a.rhs[((size_t)bus * 2) * ld + b]; y1 = a.rhs[((size_t)bus * 2 + 1) * ld + b];
            }
        }
        row_terms(a, uniform(a.r_ptr[k]) + sub, uniform(a.r_ptr[k + 1]), wpi, b, ld, y0, y1);
        if (sub != 0) { red[(size_t)wave * 128 + lane] = y0; red[(size_t)wave * 128 + 64 + lane] = y1; }
    }
    __syncthreads();
    if (valid && sub == 0) {
        for (int w = 1; w < wpi; ++w) { y0 += red[(size_t)(wave + w) * 128 + lane]; y1 += red[(size_t)(wave + w) * 128 + 64 + lane]; }
        if (BWD) bwd_finish(a, k, b, ld, y0, y1, act);
        else { a.W[((size_t)k * 2) * ld + b] = y0; a.W[((size_t)k * 2 + 1) * ld + b] = y1; }
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
        if (L.wpi == 1) hipLaunchKernelGGL(k_lu, grid, block, 0, st, a);
        else hipLaunchKernelGGL(k_lu_split, grid, block, (size_t)L.waves * 256 * sizeof(double), st, a, L.wpi);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

int Engine::solve(hipStream_t st, const double* rhs, double* out, const StateUpdate& upd) {
    SolveArgs a{fwd.task_ptr, fwd.step_ptr, fwd.items, l_ptr, l_ent, l_col, diag, perm, X, W, rhs, out, upd, 0, ld};
    for (const Launch& L : fwd.launches) {
        a.task0 = L.task_begin;
        dim3 grid(L.task_end - L.task_begin, ld / 64), block(64, L.waves);
        if (L.wpi == 1) hipLaunchKernelGGL(k_tri<false>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(k_tri_split<false>, grid, block, (size_t)L.waves * 128 * sizeof(double), st, a, L.wpi);
    }
    a.task_ptr = bwd.task_ptr; a.step_ptr = bwd.step_ptr; a.items = bwd.items;
    a.r_ptr = u_ptr; a.r_ent = u_ent; a.r_col = u_col;
    for (const Launch& L : bwd.launches) {
        a.task0 = L.task_begin;
        dim3 grid(L.task_end - L.task_begin, ld / 64), block(64, L.waves);
        if (L.wpi == 1) hipLaunchKernelGGL(k_tri<true>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(k_tri_split<true>, grid, block, (size_t)L.waves * 128 * sizeof(double), st, a, L.wpi);
    }
    JG_HIP(hipGetLastError());
    return 0;
}

}  // namespace jg
