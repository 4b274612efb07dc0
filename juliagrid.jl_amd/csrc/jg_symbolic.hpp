// jg_symbolic.hpp -- host-side symbolic analysis for the batched block-sparse LU engine.
//
// Replaces (as a whole) the symbolic half of the reference's third-party factorizations
// (`lu`/`klu`/`ldlt` first call, /root/reference/src/backend/utility.jl:470-476, 486-492, 534-540):
// fill-reducing ordering, fill pattern, and -- new here -- a static dependency schedule that the
// device numeric kernels replay every iteration (the `lu!`/`klu!` path, utility.jl:478-484).
//
// The matrix is an n x n BLOCK matrix with a structurally symmetric pattern (Ybus pattern for the
// Newton-Raphson Jacobian, pattern of H'H for the Gauss-Newton gain); every block is 2x2
// ((theta_i, V_i) per bus).  No numerical pivoting between blocks: the bus-block Jacobian and the
// SPD gain matrix are factorised with a static pivot order (min degree); 2x2 diagonal blocks are
// inverted exactly.  Integer work only -- nothing here touches floating point.
#pragma once
#include <cstdint>
#include <vector>

namespace jg {

struct Launch {
    int task_begin = 0, task_end = 0;   // tasks [begin,end) -> one workgroup column each
    int waves = 4;                      // waves per workgroup for this launch
    int wpi = 1;                        // waves cooperating on ONE item (its update terms are split across
                                        // them and reduced through LDS); wpi > 1 => single-step tasks
    int chunk = 1;                      // items per task (the last task of a launch may hold fewer)
    int fused = 0;                      // 1: ONE task (one workgroup per scenario group) walks several narrow
                                        //    dependency levels as barrier-separated steps (step_wpi per step)
    int item_begin = 0, item_end = 0;   // the launch's contiguous range in Schedule::items
};

// A schedule = launches -> tasks (one workgroup each) -> steps (barrier separated) -> items.
struct Schedule {
    std::vector<Launch> launches;
    std::vector<int> task_ptr;          // steps of task t: [task_ptr[t], task_ptr[t+1])
    std::vector<int> step_ptr;          // items of step s: [step_ptr[s], step_ptr[s+1])
    std::vector<int> items;             // entry ids (LU) or pivot ids (solves)
    std::vector<int> step_wpi;          // waves per item of every step (1 unless the step belongs to a fused launch)
    int n_levels = 0;
};

struct BlockSymbolic {
    int n = 0;
    std::vector<int> perm;              // perm[k]  = original block index eliminated k-th
    std::vector<int> iperm;             // iperm[i] = pivot position of original index i
    // Factor entries (pattern of L+D+U in pivot numbering), ids are row-major over (row, sorted col)
    int n_entries = 0;
    std::vector<int> row_ptr;           // [n+1] entries of pivot-row r
    std::vector<int> e_col;             // column (pivot numbering)
    std::vector<int> e_row;
    std::vector<int> e_src;             // index into the caller's block CSR (original order), -1 = fill
    std::vector<int> e_diag;            // for lower entries (row>col): entry id of D(col,col); else -1
    std::vector<int> diag;              // [n] entry id of D(k,k)
    // Factorisation A = Lh * inv(D) * U with UNSCALED lower blocks Lh (so a pivot's diagonal inverse,
    // its U row and its Lh column all become final in the same dependency level):
    //   entry e accumulates  - X[t_a] * X[t_d] * X[t_b]   (Lh(i,k) * Dinv(k) * U(k,j))  for t in [t_ptr[e], t_ptr[e+1])
    std::vector<int> t_ptr, t_a, t_d, t_b;
    std::vector<int> e_level;           // dependency level of each entry (1-based)
    // Row lists (pivot numbering): strictly-lower and strictly-upper entries of pivot row k
    std::vector<int> l_ptr, l_ent, l_col;   // forward elimination fused into the factorisation:
                                            //   y_k = f_k - sum_c Lh(k,c) Dinv(c) y_c      (item id n_entries + k)
    std::vector<int> u_ptr, u_ent, u_col;   // x_k = Dinv_k (y_k - sum_c U(k,c) x_c)
    std::vector<int> y_level, bwd_level;
    Schedule fact, bwd;                 // fact items: [0,n_entries) entries, [n_entries, n_entries+n) rhs rows
    long long n_terms = 0;
};

// pattern: CSR (rowptr[n+1], col[nnz]) 0-based, must contain the diagonal and be structurally
// symmetric. policy: 0 = one launch per dependency level (baseline), 1 = subtree tasks + level tail.
// Returns 0, or 1 on a malformed pattern.
int analyze(int n, const int* rowptr, const int* col, int policy, BlockSymbolic& out);

}  // namespace jg
