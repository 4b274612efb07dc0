// jg_plan_api.cpp -- device-free access to the symbolic analysis (C ABI, see include/jgrid.h).
// Lets the CPU test-suite replay the static LU / solve schedules in numpy and check every
// dependency (a race detector for the schedule) without a GPU.  No numeric code here.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jgrid.h"
#include "jg_symbolic.hpp"

struct jg_plan {
    jg::BlockSymbolic S;
};

extern "C" {

int jg_plan_create(jg_plan** out, int64_t n, const int32_t* rowptr, const int32_t* col, int64_t policy) {
    if (!out || !rowptr || !col || n < 1) return 1;
    jg_plan* p = new jg_plan();
    if (jg::analyze((int)n, rowptr, col, (long long)policy, p->S)) { delete p; return 1; }
    *out = p;
    return 0;
}

void jg_plan_destroy(jg_plan* p) { delete p; }

// which: 0 perm, 1 e_row, 2 e_col, 3 e_src, 4 t_ptr, 5 t_a, 6 t_b, 7 e_level, 8 e_diag, 9 diag,
//        10 l_ptr, 11 l_ent, 12 l_col, 13 u_ptr, 14 u_ent, 15 u_col,
//        16 t_d, 17 y_level
//        60 fact segments (x8), 61 fact wave records (x16), 62 bwd segments, 63 bwd records, 64 src_entry (device replay tables)
//        18 bwd_level (row-wise), 19 chain_level (level of the backward chain / row a pivot belongs to), 65 backward chain task data, 66 / 67 forward-only segments / records,
//        68 / 69 selected-inverse segments / records
//        70 top-task headers (x16), 71 top-task data, 72 top launches (x8: task_begin, ntasks, class, level, grouped, wg_begin, nwg, -),
//        78 workgroup map of the grouped launches (task << 8 | scenario block),
//        73 task of each pivot (-1: bottom), 74 {top_level, stack doubles per scenario (low 31 bits), top terms (low 31 bits), stack doubles per interleave class x3,
//        Jordan plan (0 / 1), blocks of Jordan rows behind the factor entries, factorisation tables are TASKS (0 / 1: policy bit 50, jg_symbolic.hpp),
//        rounds a task is filled up to, operands staged by the tasks, terms of task items in three-operand form}, 79 / 80 backward segments / records over Jordan rows
//        81 - 85 the factorisation of a single instance below the top (plans with policy bit 60, jg_symbolic.hpp: SINGLE_FACT_LEVELS): records (x16), first record of the
//        bottom items, [workgroups][levels + 1] ranges of them, first record of the partial items, {ok, workgroups, levels, items per workgroup}
//        90 - 97 the backward sweep of a single instance (jg_symbolic.hpp: SingleTables): {ok, top rows, top levels, bottom rows, bottom workgroups, bottom levels, rows per
//        workgroup, terms-as-lanes form granted, most terms / rows of a level}, t_row (x4), t_ptr (x2), t_term, t_level, b_wg (x2), b_row (x6), b_term (x2), 98 - 100 t_jb (x2), t_cslot, t_toff
// out == NULL returns the length.
int64_t jg_plan_export(jg_plan* p, int which, int32_t* out, int64_t cap) {
    if (!p) return -1;
    if (which == 68 || which == 69) jg::build_selected_inverse(p->S);     // built on demand
    const jg::BlockSymbolic& S = p->S;
    std::vector<int> tmp;
    const std::vector<int>* v = nullptr;
    switch (which) {
        case 0: v = &S.perm; break;   case 1: v = &S.e_row; break;  case 2: v = &S.e_col; break;
        case 3: v = &S.e_src; break;  case 4: v = &S.t_ptr; break;  case 5: v = &S.t_a; break;
        case 6: v = &S.t_b; break;    case 7: v = &S.e_level; break; case 8: v = &S.e_diag; break;
        case 9: v = &S.diag; break;   case 10: v = &S.l_ptr; break; case 11: v = &S.l_ent; break;
        case 12: v = &S.l_col; break; case 13: v = &S.u_ptr; break; case 14: v = &S.u_ent; break;
        case 15: v = &S.u_col; break;  case 16: v = &S.t_d; break;  case 17: v = &S.y_level; break;
        case 18: v = &S.bwd_level; break;
        case 60: tmp.assign((const int*)S.fact_seg.data(), (const int*)S.fact_seg.data() + S.fact_seg.size() * 8); v = &tmp; break;
        case 61: tmp.assign((const int*)S.fact_rec.data(), (const int*)S.fact_rec.data() + S.fact_rec.size() * 16); v = &tmp; break;
        case 62: tmp.assign((const int*)S.bwd_seg.data(), (const int*)S.bwd_seg.data() + S.bwd_seg.size() * 8); v = &tmp; break;
        case 63: tmp.assign((const int*)S.bwd_rec.data(), (const int*)S.bwd_rec.data() + S.bwd_rec.size() * 16); v = &tmp; break;
        case 64: v = &S.src_entry; break;
        case 65: v = &S.bwd_chain; break;
        case 66: tmp.assign((const int*)S.fwd_seg.data(), (const int*)S.fwd_seg.data() + S.fwd_seg.size() * 8); v = &tmp; break;
        case 67: tmp.assign((const int*)S.fwd_rec.data(), (const int*)S.fwd_rec.data() + S.fwd_rec.size() * 16); v = &tmp; break;
        case 19: v = &S.chain_level; break;
        case 68: tmp.assign((const int*)S.sel_seg.data(), (const int*)S.sel_seg.data() + S.sel_seg.size() * 8); v = &tmp; break;
        case 69: tmp.assign((const int*)S.sel_rec.data(), (const int*)S.sel_rec.data() + S.sel_rec.size() * 16); v = &tmp; break;
        case 70: tmp.assign((const int*)S.top_task.data(), (const int*)S.top_task.data() + S.top_task.size() * 16); v = &tmp; break;
        case 71: v = &S.top_data; break;
        case 72: tmp.assign((const int*)S.top_launch.data(), (const int*)S.top_launch.data() + S.top_launch.size() * 8); v = &tmp; break;
        case 78: v = &S.top_wgmap; break;
        case 73: v = &S.top_task_of; break;
        case 74: tmp = {S.top_level, (int)(S.top_stack & 0x7fffffff), (int)(S.top_terms & 0x7fffffff), (int)S.top_stack_cls[0], (int)S.top_stack_cls[1], (int)S.top_stack_cls[2], S.jordan, S.n_jordan,
                        S.fact_tasks, S.task_rounds, (int)S.n_staged, (int)S.n_direct_terms}; v = &tmp; break;
        case 79: tmp.assign((const int*)S.bwdj_seg.data(), (const int*)S.bwdj_seg.data() + S.bwdj_seg.size() * 8); v = &tmp; break;
        case 80: tmp.assign((const int*)S.bwdj_rec.data(), (const int*)S.bwdj_rec.data() + S.bwdj_rec.size() * 16); v = &tmp; break;
        case 81: tmp.assign((const int*)S.f_rec.data(), (const int*)S.f_rec.data() + S.f_rec.size() * 16); v = &tmp; break;
        case 82: v = &S.f1_first; break;
        case 83: v = &S.f1_wg; break;
        case 84: v = &S.f2_first; break;
        case 85: tmp = {S.single_fact_ok ? 1 : 0, S.n_f1_wg, jg::SINGLE_FACT_LEVELS, jg::SINGLE_FACT_ITEMS}; v = &tmp; break;
        case 90: case 91: case 92: case 93: case 94: case 95: case 96: case 97: case 98: case 99: case 100: {
            jg::SingleTables T;
            jg::build_single_tables(S, T);
            switch (which) {
                case 90: tmp = {T.ok ? 1 : 0, T.n_top, T.n_top_levels, T.n_bottom, T.n_wg, T.b_levels, jg::SINGLE_BOTTOM_ROWS, T.flat_ok ? 1 : 0, T.max_level_terms, T.max_level_rows}; break;
                case 91: tmp = T.t_row; break;  case 92: tmp = T.t_ptr; break;  case 93: tmp = T.t_term; break;  case 94: tmp = T.t_level; break;
                case 95: tmp = T.b_wg; break;   case 96: tmp = T.b_row; break;  case 97: tmp = T.b_term; break;
                case 98: tmp = T.t_jb; break;   case 99: tmp = T.t_cslot; break; default: tmp = T.t_toff; break;
            }
            v = &tmp; break;
        }
        case 75: tmp.assign(S.pre_pivot.begin(), S.pre_pivot.end()); v = &tmp; break;
        case 76: tmp.assign((const int*)S.pre_seg.data(), (const int*)S.pre_seg.data() + S.pre_seg.size() * 8); v = &tmp; break;
        case 77: tmp.assign((const int*)S.pre_rec.data(), (const int*)S.pre_rec.data() + S.pre_rec.size() * 16); v = &tmp; break;
        default: return -1;
    }
    if (!out) return (int64_t)v->size();
    if ((int64_t)v->size() > cap) return -1;
    if (!v->empty()) std::memcpy(out, v->data(), v->size() * sizeof(int));     // (an empty vector may hand out a null pointer: UBSan, tools/asan_plan.sh)
    return (int64_t)v->size();
}

// Tables of the shared-factor solve (jg_symbolic.hpp: CompTables) for a dense top of at most top_cap pivots (< 0: none).
// which: 0 {n_top, split level, forward levels, backward levels}, 1 top pivots, 2 / 3 forward segments (x8) / records (x16), 4 / 5 backward segments / records.
int64_t jg_plan_comp_export(jg_plan* p, int64_t top_cap, int which, int32_t* out, int64_t cap) {
    if (!p || which < 0 || which > 5) return -1;
    jg::CompTables T;
    jg::build_comp_tables(p->S, (int)top_cap, T);
    std::vector<int> tmp;
    switch (which) {
        case 0: tmp = {T.n_top, T.split, T.n_fwd_levels, T.n_bwd_levels}; break;
        case 1: tmp = T.top; break;
        case 2: tmp.assign((const int*)T.fwd_seg.data(), (const int*)T.fwd_seg.data() + T.fwd_seg.size() * 8); break;
        case 3: tmp.assign((const int*)T.fwd_rec.data(), (const int*)T.fwd_rec.data() + T.fwd_rec.size() * 16); break;
        case 4: tmp.assign((const int*)T.bwd_seg.data(), (const int*)T.bwd_seg.data() + T.bwd_seg.size() * 8); break;
        default: tmp.assign((const int*)T.bwd_rec.data(), (const int*)T.bwd_rec.data() + T.bwd_rec.size() * 16); break;
    }
    if (!out) return (int64_t)tmp.size();
    if ((int64_t)tmp.size() > cap) return -1;
    if (!tmp.empty()) std::memcpy(out, tmp.data(), tmp.size() * sizeof(int));
    return (int64_t)tmp.size();
}

}  // extern "C"
