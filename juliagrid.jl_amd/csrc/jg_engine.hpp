// jg_engine.hpp -- batched block-sparse LU / triangular-solve engine on the device (gfx950).
//
// Numeric half of the factorization seam of the reference
// (/root/reference/src/backend/utility.jl:478-484 `lu!`/`klu!`, :576-586 `ldiv!`), for B
// independent scenarios at once.  Layout is batch-minor ("structure of scenarios"):
//     value(entry e, component c, scenario b) = X[(e*4 + c) * ld + b]
// so the 64 lanes of a wavefront work on the same structural block of 64 scenarios: every load
// and store is a contiguous 512-byte segment, no divergence, one shared symbolic structure.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "jg_symbolic.hpp"

namespace jg {

struct DevSchedule {
    int* task_ptr = nullptr;
    int* step_ptr = nullptr;
    int* items = nullptr;
    std::vector<Launch> launches;
};

// Optional state update fused into the backward solve (NR: x <- x - dx, masked by bus flags).
struct StateUpdate {
    double* va = nullptr;          // [n][ld]
    double* vm = nullptr;          // [n][ld]
    const signed char* flags = nullptr;  // [n] bit0: angle is a state, bit1: magnitude is a state
    const int* active = nullptr;   // [ld] 1 = apply update for this scenario
    double sign = 0.0;             // -1 (Newton-Raphson) / +1 (Gauss-Newton)
};

struct Engine {
    BlockSymbolic S;
    int ld = 0;                    // padded batch (multiple of 64)
    int* e_src = nullptr;
    int* e_diag = nullptr;         // >=0 lower (diag entry of its column), -1 upper, -2 diagonal
    int* t_ptr = nullptr;
    int* t_a = nullptr;
    int* t_b = nullptr;
    int* l_ptr = nullptr; int* l_ent = nullptr; int* l_col = nullptr;
    int* u_ptr = nullptr; int* u_ent = nullptr; int* u_col = nullptr;
    int* diag = nullptr;
    int* perm = nullptr;
    double* X = nullptr;           // factor values [n_entries][4][ld]
    double* W = nullptr;           // solve workspace [n][2][ld], pivot order
    int* status = nullptr;         // [ld] bit 2 set on zero / non-finite pivot
    DevSchedule lu, fwd, bwd;
    std::string error;

    int create(int n, const int* rowptr, const int* col, int ld_, int policy);
    void destroy();
    // A: block values in the caller's CSR order [nnz][4][ld]
    int factor(hipStream_t st, const double* A);
    // rhs, out: [n][2][ld] in original block order. out receives the solution.
    int solve(hipStream_t st, const double* rhs, double* out, const StateUpdate& upd);
    size_t factor_bytes() const { return (size_t)S.n_entries * 4 * ld * sizeof(double); }
};

#define JG_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess) {                                                        \
            this->error = std::string(#expr) + ": " + hipGetErrorString(err__);           \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

template <class T>
int upload(T** dst, const std::vector<T>& src, std::string& err) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e == hipSuccess && !src.empty()) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) { err = std::string("upload: ") + hipGetErrorString(e); return 2; }
    return 0;
}

}  // namespace jg
