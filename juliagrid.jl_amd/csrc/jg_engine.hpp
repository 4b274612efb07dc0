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

void set_last_error(const std::string& msg);   // thread-local text behind jg_last_error()

// 32-byte item descriptor, fetched with one scalar load per item (schedule order).
struct ItemDesc {
    int kind;      // 0 upper U(k,j), 1 lower Lh(i,k) (unscaled), 2 diagonal (stores its 2x2 LU factors), 3 rhs row y_k
    int id;        // entry id, or pivot k for rhs rows / backward rows
    int src;       // block index in the caller's CSR (-1 = fill-in); original block index (bus) for rows
    int t0, t1;    // term range in (ta, td, tb)
    int aux;       // backward rows: entry id of the diagonal
    int pad0, pad1;
};

struct DevLaunch {
    int item_begin, item_end;   // range in the descriptor array
    int waves, wpi, rounds;     // blockDim.y, waves per item, items per slot
    int grid;                   // workgroups along x
    int fused = 0, n_steps = 0, step0 = 0;   // fused launch: step table rows [step0, step0 + n_steps) of {begin, end, wpi}
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
    ItemDesc* fact_desc = nullptr; // factorisation + fused forward elimination, schedule order
    ItemDesc* bwd_desc = nullptr;
    int* ta = nullptr; int* td = nullptr; int* tb = nullptr;   // LU terms followed by rhs-row terms
    int* u_ent = nullptr; int* u_col = nullptr;
    int* bwd_steps = nullptr;     // {item_begin, item_end, wpi} per step of fused backward launches
    double* X = nullptr;           // factor values [n_entries][4][ld]: U, unscaled Lh, factored diagonal blocks
    double* W = nullptr;           // [n][2][ld] pivot order: y after factor(), x after backsolve()
    int* status = nullptr;         // [ld] bit 2 set on zero / non-finite pivot
    std::vector<DevLaunch> fact, bwd;
    std::string error;

    int create(int n, const int* rowptr, const int* col, int ld_, int policy);
    void destroy();
    // A: block values in the caller's CSR order [nnz][4][ld]; rhs: [n][2][ld] original block order.
    // Computes A = Lh inv(D) U and y = (Lh inv(D))^-1 rhs in the same launches.
    // group_active (nullable): [ld/64] flags, workgroups of an inactive 64-scenario group exit at once.
    int factor(hipStream_t st, const double* A, const double* rhs, const int* group_active);
    // x = U^-1 D y, scattered to original order into out [n][2][ld]; optional fused state update.
    int backsolve(hipStream_t st, double* out, const StateUpdate& upd, const int* group_active);
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
