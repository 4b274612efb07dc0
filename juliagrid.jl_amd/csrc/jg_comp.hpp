// jg_comp.hpp -- the first Newton iteration of a common-start batch on ONE shared factor (compensation method).
//
// Reference counterpart: the user loop of an N-1 screen, /root/reference/src/powerSystem/branch.jl:453-459 (`updateBranch!(analysis; label,
// status = 0)`), followed by `powerFlow!` whose first `solve!` factorises (lu!, acPowerFlow.jl:890-897) a Jacobian that differs from the base
// case's in the 2 x 2 block rows / columns of the two buses of the branch:  J_s = J_0 + E M_s F'  (E = F = the <= 4 unit columns of those
// buses, M_s = the change of the <= 4 blocks).  With every scenario starting from ONE state the batch therefore needs ONE factorisation:
//     x_s = J_0^-1 (f_s - E c_s),    c_s = M_s u_s,    (I + S_s M_s) u_s = F' J_0^-1 f_s,    S_s = F' J_0^-1 E
// -- S_s (4 x 4) are entries of J_0^-1 on the Ybus pattern (formed once per base state), F' J_0^-1 f_s = (J_0^-1 f_0)_{ab} + S_s (f_s - f_0)_{ab}
// because an outage moves the mismatch of its two buses only, and x_s is ONE sweep pair with the factor VALUES wave-uniform (scalar loads of a
// 2.2 MB compact factor; only the right-hand sides are batch-minor).  A scenario whose 4 x 4 system is singular is an islanding outage (status 3).
// Iterations >= 2 refactorise as before (the states have separated).
#pragma once
#include "jg_engine.hpp"

namespace jg {

struct CompSweep {                       // device tables of one CompTables (jg_symbolic.hpp)
    CompTables T;
    Rec* fwd_rec = nullptr; Segment* fwd_seg = nullptr; Rec* bwd_rec = nullptr; Segment* bwd_seg = nullptr;
    int* top_piv = nullptr; int* top_bus = nullptr;      // [n_top] pivot, original index
    std::vector<DevLaunch> fwd, bwd;
    int upload_tables(const BlockSymbolic& S, int top_cap, hipStream_t st, std::string& err);
    void free_tables();
};

struct CompBase {
    int n = 0, nnz = 0, n_entries = 0, device = 0;
    CompSweep full, split;               // every pivot a level item (bootstrap) / bottom levels + dense top
    double* Mc = nullptr;                // [n_entries][4]: Lh(i,k) D(k)^-1 below, D(i)^-1 U(i,j) above, the 2x2 LU of D(k) on the diagonal
    double* Sinv = nullptr; int lds = 0; // dense inverse of the top's Schur complement in MFMA fragment order [row blocks of 16][lds k steps of 4][64] (jg_comp.hip: k_ctop)
    double* Zc = nullptr;                // [nnz][4]: block (unknowns of bus i, equations of bus j) of J_0^-1 at row-CSR position (i, j)
    double* v0 = nullptr; double* th0 = nullptr; double* f0 = nullptr; double* y0 = nullptr;    // start state [n], base mismatch / J_0^-1 f_0 [n][2]
    double* p0 = nullptr; double* q0 = nullptr;                                                   // injections of the base [n]
    int* rowptr = nullptr; int* colm = nullptr; int* posrow = nullptr; int* tpos = nullptr; int* rowtype = nullptr;   // Ybus row-CSR (borrowed from the creator's copies: owned here)
    std::string error;

    // Xsrc: factor of the base Jacobian (plain rows) in an engine storage of leading dimension ldx, scenario 0.  st: stream of the creator.
    int create(const BlockSymbolic& S, const double* Xsrc, int ldx, int top_cap, hipStream_t st);
    void destroy();
    // x = J_0^-1 rhs for the scenarios of `sel`: rhs [n][ld][2] bus order, Wc scratch [(n + n_top_pad)][ld][2], out [n][ld][2] bus order; optional fused state update
    int solve(const CompSweep& sw, hipStream_t st, const double* rhs, double* Wc, double* out, int ld, int lanes, const StateUpdate& upd, const GroupSel& sel) const;
    size_t scratch_rows(const CompSweep& sw) const { return (size_t)n + (size_t)((sw.T.n_top + 7) & ~7); }      // the top's partial rows, padded (zero) to the k steps k_ctop reads: 16 values = 8 rows
};

// per-scenario correction of the right-hand side: F[a], F[b] -= c_s (see above); marks singular scenarios in lu_status (bit 2)
struct CompFixArgs {
    const int* ppos; const double* pdg; const double* pdb; int mp;     // the scenario's Ybus edits (jg_nr_patch_ybus: row-CSR position, dG, dB)
    const int* rowptr; const int* colm; const int* posrow; const int* rowtype;
    const double* v0; const double* th0; const double* y0; const double* Zc;
    double* F; int* lu_status; const int* active;
    int ld, lanes;
};
void launch_comp_fix(const CompFixArgs& a, hipStream_t st);
// J_0^-1 on the Ybus pattern (B.Zc) by one shared-factor solve per (bus, component), 512 columns per batch; needs B.rowptr / colm / tpos / nnz
int comp_form_z(CompBase& B, hipStream_t st);

}  // namespace jg
