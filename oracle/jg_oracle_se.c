/*
 * jg_oracle_se.c -- CPU ORACLE for the Gauss-Newton WLS state-estimation path (test infrastructure,
 * NOT product code; see the header of jg_oracle.c for the rules).
 *
 * Restates, in plain C, of mcosovic/JuliaGrid.jl v0.6.2 (paths relative to /root/reference):
 *   acWLS                 src/stateEstimation/acStateEstimation.jl:77-259   (+ index builders :1130-1238)
 *   normalEquation!       src/stateEstimation/acStateEstimation.jl:261-583
 *   increment!{Normal}    src/stateEstimation/acStateEstimation.jl:878-904  (+ sparse.jl:155-188)
 *   solve!                src/stateEstimation/acStateEstimation.jl:1035-1047
 *   stateEstimation!      src/stateEstimation/acStateEstimation.jl:1286-1329
 *   measurement functions src/backend/equations.jl:20-60, 147-573, PMU precision :576-677, objective :689-698
 *   varianceSquare/if2exp src/measurement/utility.jl:115-129
 *   exact measurement values (power!/current! per element) src/postprocessing/acAnalysis.jl:838-931
 *
 * Parity pin: the reference's known-answer rule (test/stateEstimation/analysis.jl:27-298 through
 * test/utility/utility.jl:282-290): noise-free measurements generated from a converged power flow
 * make the WLS estimate equal the power-flow state to 1e-10 -- tests/test_oracle_se.py checks it per
 * measurement family on the golden cases; plus the analytic PMU precision checks of
 * test/stateEstimation/analysis.jl:300-346 and the squared-current variance rule (:173-200).
 * The gain-matrix solve uses the KLU-style LU of jg_oracle.c (SuiteSparse is not in the tree).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* from jg_oracle.c */
typedef struct jgo_lu jgo_lu;
jgo_lu *jgo_lu_create(i64 n);
void jgo_lu_destroy(jgo_lu *F);
int jgo_lu_factor(jgo_lu *F, const i64 *Ap, const i64 *Ai, const double *Ax);
int jgo_lu_refactor(jgo_lu *F, const i64 *Ap, const i64 *Ai, const double *Ax);
void jgo_lu_solve(jgo_lu *F, const double *b, double *x);

/* ------------------------------------------------------------------------------------------- */
/* Exact per-element quantities (postprocessing/acAnalysis.jl:838-931), used to synthesise      */
/* noise-free measurement sets independently of the product's host code.                        */
/* out per branch k (8): Pij Qij Pji Qji |Iij| ang(Iij) |Iji| ang(Iji); per bus (2): Pi Qi      */
/* twoport = the array produced by jgo_ac_model: {y, yff, yft, ytt, ytf} (re,im)                */
/* ------------------------------------------------------------------------------------------- */
void jgo_exact_quantities(i64 n, i64 nb, const i64 *from, const i64 *to, const int8_t *status,
                          const double *twoport, const i64 *colptr, const i64 *rowval,
                          const double *ytre, const double *ytim, const double *vm, const double *va,
                          double *branch_out, double *bus_out) {
    for (i64 k = 0; k < nb; k++) {
        double *o = branch_out + 8 * k;
        for (int c = 0; c < 8; c++) o[c] = 0.0;
        if (status[k] != 1) continue;
        const double *t = twoport + 10 * k;
        i64 i = from[k] - 1, j = to[k] - 1;
        double vir = vm[i] * cos(va[i]), vii = vm[i] * sin(va[i]);
        double vjr = vm[j] * cos(va[j]), vji = vm[j] * sin(va[j]);
        /* Iij = Vi*yff + Vj*yft ; Iji = Vi*ytf + Vj*ytt   (:915-921) */
        double ifr = vir * t[2] - vii * t[3] + vjr * t[4] - vji * t[5];
        double ifi = vir * t[3] + vii * t[2] + vjr * t[5] + vji * t[4];
        double itr = vir * t[8] - vii * t[9] + vjr * t[6] - vji * t[7];
        double iti = vir * t[9] + vii * t[8] + vjr * t[7] + vji * t[6];
        /* Sij = Vi conj(Iij) (:893-899) */
        o[0] = vir * ifr + vii * ifi; o[1] = vii * ifr - vir * ifi;
        o[2] = vjr * itr + vji * iti; o[3] = vji * itr - vjr * iti;
        o[4] = hypot(ifr, ifi); o[5] = atan2(ifi, ifr);
        o[6] = hypot(itr, iti); o[7] = atan2(iti, itr);
    }
    for (i64 i = 0; i < n; i++) {                                   /* Ii, PiQi (:867-891) */
        double ir = 0.0, ii = 0.0;
        for (i64 p = colptr[i] - 1; p < colptr[i + 1] - 1; p++) {
            i64 k = rowval[p] - 1;
            double vr = vm[k] * cos(va[k]), vi = vm[k] * sin(va[k]);
            ir += ytre[p] * vr - ytim[p] * vi;
            ii += ytre[p] * vi + ytim[p] * vr;
        }
        double vr = vm[i] * cos(va[i]), vi = vm[i] * sin(va[i]);
        bus_out[2 * i] = ir * vr + ii * vi;                         /* Re(conj(I) V) */
        bus_out[2 * i + 1] = ir * vi - ii * vr;                     /* Im(conj(I) V) */
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Gauss-Newton state                                                                           */
/* ------------------------------------------------------------------------------------------- */
typedef struct { double A, B, C, D; } PiModel;
typedef struct { double Vi, Vj, s, c, si, sj, ci, cj; } StateModel;

typedef struct {
    i64 n, nb, slack, m, nnzH, nW;
    /* system */
    i64 *colptr, *rowval; double *yre, *yim, *ytre, *ytim;
    i64 *from, *to; double *adm_re, *adm_im, *bg, *bb, *tap, *shift;
    /* model (acWLS) */
    int8_t *type; i64 *index; i64 range[6];
    double *mean, *wdiag, *woff;        /* woff[r] = W[r, r+1] (0 when none) */
    i64 *hcolptr, *hrowval; double *hval;
    double *residual, *increment, *vm, *va;
    double objective; i64 iteration;
    /* gain */
    i64 *gcolptr, *growval; double *gval; i64 gnnz;
    i64 *gmap_a, *gmap_b, *gmap_pos; double *gmap_w_is_off; i64 ngmap;   /* contribution list (built once) */
    jgo_lu *F; int factored;
} jgo_gn;

static void *xmalloc(size_t s) { void *p = malloc(s ? s : 1); if (!p) abort(); return p; }
#define DUPA(dst, src, cnt, T) do { dst = (T *)xmalloc((size_t)(cnt) * sizeof(T)); memcpy(dst, src, (size_t)(cnt) * sizeof(T)); } while (0)

/* stored position in a CSC column (sparse.jl:104-121) */
static i64 stored(const i64 *colptr, const i64 *rowval, i64 row, i64 col) {
    i64 lo = colptr[col] - 1, hi = colptr[col + 1] - 2;
    while (lo <= hi) { i64 mid = (lo + hi) >> 1; i64 r = rowval[mid] - 1; if (r < row) lo = mid + 1; else if (r > row) hi = mid - 1; else return mid; }
    return -1;
}

/*
 * Device table, one record per meter in the reference's concatenation order
 * (voltmeters, ammeters, wattmeters, varmeters, PMUs):
 *   kind 1..5, loc 0 bus / 1 from / 2 to, index (1-based bus or branch),
 *   mean1/var1/status1 (magnitude or the single quantity), mean2/var2/status2 (PMU angle),
 *   flags bit0 square, bit1 polar, bit2 correlated.
 */
jgo_gn *jgo_gn_create(i64 n, i64 nb, const i64 *colptr, const i64 *rowval, const double *yre, const double *yim,
                      const double *ytre, const double *ytim, const i64 *from, const i64 *to,
                      const double *twoport, const double *bg, const double *bb, const double *tap, const double *shift,
                      i64 slack, i64 ndev, const int8_t *kind, const int8_t *loc, const i64 *index,
                      const double *mean1, const double *var1, const int8_t *status1,
                      const double *mean2, const double *var2, const int8_t *status2, const int8_t *flags,
                      const double *vm0, const double *va0) {
    jgo_gn *h = (jgo_gn *)calloc(1, sizeof(jgo_gn));
    i64 nnz = colptr[n] - 1;
    h->n = n; h->nb = nb; h->slack = slack;
    DUPA(h->colptr, colptr, n + 1, i64); DUPA(h->rowval, rowval, nnz, i64);
    DUPA(h->yre, yre, nnz, double); DUPA(h->yim, yim, nnz, double); DUPA(h->ytre, ytre, nnz, double); DUPA(h->ytim, ytim, nnz, double);
    DUPA(h->from, from, nb, i64); DUPA(h->to, to, nb, i64);
    h->adm_re = (double *)xmalloc((size_t)nb * 8); h->adm_im = (double *)xmalloc((size_t)nb * 8);
    for (i64 k = 0; k < nb; k++) { h->adm_re[k] = twoport[10 * k]; h->adm_im[k] = twoport[10 * k + 1]; }
    DUPA(h->bg, bg, nb, double); DUPA(h->bb, bb, nb, double); DUPA(h->tap, tap, nb, double); DUPA(h->shift, shift, nb, double);
    DUPA(h->vm, vm0, n, double); DUPA(h->va, va0, n, double);       /* acStateEstimation.jl:52-55 */

    /* ---- acWLS (:77-259) ---- */
    i64 total = 0;
    for (i64 d = 0; d < ndev; d++) total += kind[d] == 5 ? 2 : 1;   /* :90 */
    h->m = total;
    h->type = (int8_t *)calloc((size_t)total + 1, 1); h->index = (i64 *)calloc((size_t)total + 1, sizeof(i64));
    h->mean = (double *)calloc((size_t)total + 1, 8); h->wdiag = (double *)calloc((size_t)total + 1, 8); h->woff = (double *)calloc((size_t)total + 1, 8);
    h->residual = (double *)calloc((size_t)total + 1, 8); h->increment = (double *)calloc(2 * (size_t)n, 8);
    /* triplets */
    i64 cap = 0;
    for (i64 d = 0; d < ndev; d++) {
        i64 deg2 = 0;
        if ((kind[d] == 3 || kind[d] == 4) && loc[d] == 0) deg2 = 2 * (colptr[index[d]] - colptr[index[d] - 1]);
        cap += deg2 + 8;
    }
    i64 *tr = (i64 *)xmalloc((size_t)cap * 8), *tc = (i64 *)xmalloc((size_t)cap * 8); double *tv = (double *)xmalloc((size_t)cap * 8);
    i64 cnt = 0, row = 0;
    for (int f = 0; f < 6; f++) h->range[f] = 1;
#define ONE(col_, code_, st_, bus_) do { h->type[row] = (int8_t)((st_) * (code_)); h->index[row] = (bus_); tr[cnt] = row; tc[cnt] = (col_); tv[cnt] = (double)(st_); cnt++; row++; } while (0)
#define TWO(code_, st_, bus_) do { h->type[row] = (int8_t)((st_) * (code_)); h->index[row] = (bus_); tr[cnt] = row; tc[cnt] = (bus_) - 1; tv[cnt] = 0.0; cnt++; tr[cnt] = row; tc[cnt] = (bus_) - 1 + n; tv[cnt] = 0.0; cnt++; row++; } while (0)
#define FOUR(code1_, code2_, st_, isfrom_, br_) do { h->index[row] = (br_); h->type[row] = (int8_t)((st_) * ((isfrom_) ? (code1_) : (code2_))); \
        i64 f_ = from[(br_) - 1] - 1, t_ = to[(br_) - 1] - 1; \
        tr[cnt] = row; tc[cnt] = f_; tv[cnt++] = 0.0; tr[cnt] = row; tc[cnt] = t_; tv[cnt++] = 0.0; \
        tr[cnt] = row; tc[cnt] = f_ + n; tv[cnt++] = 0.0; tr[cnt] = row; tc[cnt] = t_ + n; tv[cnt++] = 0.0; row++; } while (0)
#define NTH(code_, st_, bus_) do { h->type[row] = (int8_t)((st_) * (code_)); h->index[row] = (bus_); \
        for (i64 p_ = colptr[(bus_) - 1] - 1; p_ < colptr[(bus_)] - 1; p_++) { tr[cnt] = row; tc[cnt] = rowval[p_] - 1; tv[cnt++] = 0.0; tr[cnt] = row; tc[cnt] = rowval[p_] - 1 + n; tv[cnt++] = 0.0; } row++; } while (0)
    int fam = 1;
    for (i64 d = 0; d < ndev; d++) {
        while (fam < kind[d]) { h->range[fam] = row + 1; fam++; }   /* range[2..5] = first row of the next family */
        int8_t st = status1[d];
        int sq = flags[d] & 1, polar = (flags[d] >> 1) & 1, corr = (flags[d] >> 2) & 1;
        i64 k = index[d];
        if (kind[d] == 1) {                                          /* :135-141 */
            h->mean[row] = st * mean1[d]; h->wdiag[row] = 1.0 / var1[d];
            ONE(k - 1 + n, 1, st, k);
        } else if (kind[d] == 2) {                                   /* :144-157 */
            h->mean[row] = st * (sq ? mean1[d] * mean1[d] : mean1[d]);
            h->wdiag[row] = 1.0 / (sq ? 4.0 * mean1[d] * mean1[d] * var1[d] : var1[d]);   /* varianceSquare */
            if (sq) FOUR(4, 5, st, loc[d] == 1, k); else FOUR(2, 3, st, loc[d] == 1, k);
        } else if (kind[d] == 3) {                                   /* :160-171 */
            h->mean[row] = st * mean1[d]; h->wdiag[row] = 1.0 / var1[d];
            if (loc[d] == 0) NTH(6, st, k); else FOUR(7, 8, st, loc[d] == 1, k);
        } else if (kind[d] == 4) {                                   /* :174-185 */
            h->mean[row] = st * mean1[d]; h->wdiag[row] = 1.0 / var1[d];
            if (loc[d] == 0) NTH(9, st, k); else FOUR(10, 11, st, loc[d] == 1, k);
        } else {                                                     /* PMU :188-235 */
            int8_t sm = status1[d], sa = status2[d];
            if (polar) {
                h->mean[row] = sm * (sq ? mean1[d] * mean1[d] : mean1[d]);
                h->wdiag[row] = 1.0 / (sq ? 4.0 * mean1[d] * mean1[d] * var1[d] : var1[d]);
                h->mean[row + 1] = sa * mean2[d]; h->wdiag[row + 1] = 1.0 / var2[d];
                if (loc[d] == 0) { ONE(k - 1 + n, 12, sm, k); ONE(k - 1, 13, sa, k); }
                else {
                    if (sq) FOUR(4, 5, sm, loc[d] == 1, k); else FOUR(2, 3, sm, loc[d] == 1, k);
                    FOUR(14, 15, sa, loc[d] == 1, k);
                }
            } else {
                double s = sin(mean2[d]), c = cos(mean2[d]);
                int8_t stt = (int8_t)(sm * sa);
                h->mean[row] = stt * mean1[d] * c; h->mean[row + 1] = stt * mean1[d] * s;
                /* variancePmu (equations.jl:576-588) */
                double vre = var1[d] * c * c + var2[d] * (mean1[d] * s) * (mean1[d] * s);
                double vim = var1[d] * s * s + var2[d] * (mean1[d] * c) * (mean1[d] * c);
                if (corr) {                                          /* covariancePmu + precision! (:591-666) */
                    double L1i = 1.0 / sqrt(vre);
                    double L2 = s * c * (var1[d] - var2[d] * mean1[d] * mean1[d]) * L1i;
                    double L3i2 = 1.0 / (vim - L2 * L2);
                    double off = (-L2 * L1i) * L3i2;
                    h->woff[row] = off;
                    h->wdiag[row] = (L1i - L2 * off) * L1i;
                    h->wdiag[row + 1] = L3i2;
                } else { h->wdiag[row] = 1.0 / vre; h->wdiag[row + 1] = 1.0 / vim; }
                if (loc[d] == 0) { TWO(16, stt, k); TWO(17, stt, k); }
                else { FOUR(18, 19, stt, loc[d] == 1, k); FOUR(20, 21, stt, loc[d] == 1, k); }
            }
        }
    }
    while (fam < 5) { h->range[fam] = row + 1; fam++; }
    h->range[5] = row + 1;
    /* sparse(row, col, val, total, 2n) (:238): CSC, rows sorted inside a column, duplicates summed */
    i64 ncol = 2 * n;
    h->hcolptr = (i64 *)calloc((size_t)ncol + 1, 8);
    for (i64 t = 0; t < cnt; t++) h->hcolptr[tc[t] + 1]++;
    h->hcolptr[0] = 1;
    for (i64 c = 0; c < ncol; c++) h->hcolptr[c + 1] += h->hcolptr[c];
    h->nnzH = h->hcolptr[ncol] - 1;
    h->hrowval = (i64 *)xmalloc((size_t)h->nnzH * 8); h->hval = (double *)xmalloc((size_t)h->nnzH * 8);
    {
        i64 *fill = (i64 *)xmalloc((size_t)ncol * 8);
        for (i64 c = 0; c < ncol; c++) fill[c] = h->hcolptr[c] - 1;
        for (i64 t = 0; t < cnt; t++) { i64 p = fill[tc[t]]++; h->hrowval[p] = tr[t] + 1; h->hval[p] = tv[t]; }   /* rows already ascending */
        free(fill);
    }
    free(tr); free(tc); free(tv);
    h->F = jgo_lu_create(2 * n);
    return h;
}

void jgo_gn_destroy(jgo_gn *h) {
    if (!h) return;
    free(h->colptr); free(h->rowval); free(h->yre); free(h->yim); free(h->ytre); free(h->ytim); free(h->from); free(h->to);
    free(h->adm_re); free(h->adm_im); free(h->bg); free(h->bb); free(h->tap); free(h->shift);
    free(h->type); free(h->index); free(h->mean); free(h->wdiag); free(h->woff); free(h->hcolptr); free(h->hrowval); free(h->hval);
    free(h->residual); free(h->increment); free(h->vm); free(h->va);
    free(h->gcolptr); free(h->growval); free(h->gval); free(h->gmap_a); free(h->gmap_b); free(h->gmap_pos); free(h->gmap_w_is_off);
    jgo_lu_destroy(h->F); free(h);
}

i64 jgo_gn_rows(jgo_gn *h) { return h->m; }
i64 jgo_gn_nnz(jgo_gn *h) { return h->nnzH; }
i64 jgo_gn_iteration(jgo_gn *h) { return h->iteration; }
double jgo_gn_objective(jgo_gn *h) { return h->objective; }
void jgo_gn_get_model(jgo_gn *h, int8_t *type, i64 *index, i64 *range, double *mean, double *wdiag, double *woff,
                      i64 *hcolptr, i64 *hrowval) {
    memcpy(type, h->type, (size_t)h->m); memcpy(index, h->index, (size_t)h->m * 8); memcpy(range, h->range, 48);
    memcpy(mean, h->mean, (size_t)h->m * 8); memcpy(wdiag, h->wdiag, (size_t)h->m * 8); memcpy(woff, h->woff, (size_t)h->m * 8);
    memcpy(hcolptr, h->hcolptr, (2 * (size_t)h->n + 1) * 8); memcpy(hrowval, h->hrowval, (size_t)h->nnzH * 8);
}
void jgo_gn_get_vectors(jgo_gn *h, double *hval, double *residual, double *increment, double *vm, double *va) {
    if (hval) memcpy(hval, h->hval, (size_t)h->nnzH * 8);
    if (residual) memcpy(residual, h->residual, (size_t)h->m * 8);
    if (increment) memcpy(increment, h->increment, 2 * (size_t)h->n * 8);
    if (vm) memcpy(vm, h->vm, (size_t)h->n * 8);
    if (va) memcpy(va, h->va, (size_t)h->n * 8);
}
void jgo_gn_set_voltage(jgo_gn *h, const double *vm, const double *va) { memcpy(h->vm, vm, (size_t)h->n * 8); memcpy(h->va, va, (size_t)h->n * 8); }
void jgo_gn_set_mean(jgo_gn *h, const double *mean) { memcpy(h->mean, mean, (size_t)h->m * 8); }

/* ---- coefficients and states (equations.jl:20-60, 147-156, 183-192, 215-224, 247-256, 279-291, 334-346, 389-399, 426-436) */
static StateModel st_ij(const jgo_gn *h, i64 k) {        /* ViVjthetaijState */
    i64 i = h->from[k] - 1, j = h->to[k] - 1; StateModel e; memset(&e, 0, sizeof e);
    double th = h->va[i] - h->va[j] - h->shift[k];
    e.Vi = h->vm[i]; e.Vj = h->vm[j]; e.s = sin(th); e.c = cos(th); return e;
}
static StateModel st_ithj(const jgo_gn *h, i64 k) {      /* ViVjthetaithetajState */
    i64 i = h->from[k] - 1, j = h->to[k] - 1; StateModel e; memset(&e, 0, sizeof e);
    e.Vi = h->vm[i]; e.Vj = h->vm[j]; e.si = sin(h->va[i]); e.ci = cos(h->va[i]);
    e.sj = sin(h->va[j] + h->shift[k]); e.cj = cos(h->va[j] + h->shift[k]); return e;
}
static StateModel st_jthi(const jgo_gn *h, i64 k) {      /* VjVithetajthetaiState */
    i64 i = h->from[k] - 1, j = h->to[k] - 1; StateModel e; memset(&e, 0, sizeof e);
    e.Vi = h->vm[i]; e.Vj = h->vm[j]; e.si = sin(h->va[i] - h->shift[k]); e.ci = cos(h->va[i] - h->shift[k]);
    e.sj = sin(h->va[j]); e.cj = cos(h->va[j]); return e;
}
static PiModel co_Pij(const jgo_gn *h, i64 k) { double ti = 1.0 / h->tap[k]; PiModel p = {ti * ti * (h->adm_re[k] + 0.5 * h->bg[k]), ti * h->adm_re[k], ti * h->adm_im[k], 0}; return p; }
static PiModel co_Pji(const jgo_gn *h, i64 k) { double ti = 1.0 / h->tap[k]; PiModel p = {h->adm_re[k] + 0.5 * h->bg[k], ti * h->adm_re[k], ti * h->adm_im[k], 0}; return p; }
static PiModel co_Qij(const jgo_gn *h, i64 k) { double ti = 1.0 / h->tap[k]; PiModel p = {ti * ti * (h->adm_im[k] + 0.5 * h->bb[k]), ti * h->adm_re[k], ti * h->adm_im[k], 0}; return p; }
static PiModel co_Qji(const jgo_gn *h, i64 k) { double ti = 1.0 / h->tap[k]; PiModel p = {h->adm_im[k] + 0.5 * h->bb[k], ti * h->adm_re[k], ti * h->adm_im[k], 0}; return p; }
static PiModel co_Iij(const jgo_gn *h, i64 k) {
    double g = h->adm_re[k], b = h->adm_im[k], gs = 0.5 * h->bg[k], bs = 0.5 * h->bb[k], ti = 1.0 / h->tap[k];
    double t2 = ti * ti;
    PiModel p = {t2 * t2 * ((g + gs) * (g + gs) + (b + bs) * (b + bs)), t2 * (g * g + b * b), t2 * ti * (g * (g + gs) + b * (b + bs)), t2 * ti * (g * bs - b * gs)};
    return p;
}
static PiModel co_Iji(const jgo_gn *h, i64 k) {
    double g = h->adm_re[k], b = h->adm_im[k], gs = 0.5 * h->bg[k], bs = 0.5 * h->bb[k], ti = 1.0 / h->tap[k];
    PiModel p = {ti * ti * (g * g + b * b), (g + gs) * (g + gs) + (b + bs) * (b + bs), ti * (g * (g + gs) + b * (b + bs)), ti * (g * bs - gs * b)};
    return p;
}
static PiModel co_psiij(const jgo_gn *h, i64 k) { double ti = 1.0 / h->tap[k]; PiModel p = {ti * ti * (h->adm_re[k] + 0.5 * h->bg[k]), ti * ti * (h->adm_im[k] + 0.5 * h->bb[k]), ti * h->adm_re[k], ti * h->adm_im[k]}; return p; }
static PiModel co_psiji(const jgo_gn *h, i64 k) { double ti = 1.0 / h->tap[k]; PiModel p = {h->adm_re[k] + 0.5 * h->bg[k], h->adm_im[k] + 0.5 * h->bb[k], ti * h->adm_re[k], ti * h->adm_im[k]}; return p; }

static double ReIij(PiModel p, StateModel e) { return (p.A * e.ci - p.B * e.si) * e.Vi - (p.C * e.cj - p.D * e.sj) * e.Vj; }
static double ImIij(PiModel p, StateModel e) { return (p.A * e.si + p.B * e.ci) * e.Vi - (p.C * e.sj + p.D * e.cj) * e.Vj; }
static double ReIji(PiModel p, StateModel e) { return (p.A * e.cj - p.B * e.sj) * e.Vj - (p.C * e.ci - p.D * e.si) * e.Vi; }
static double ImIji(PiModel p, StateModel e) { return (p.A * e.sj + p.B * e.cj) * e.Vj - (p.C * e.si + p.D * e.ci) * e.Vi; }

static void seobj1(jgo_gn *h, i64 r) { h->objective += h->residual[r] * h->residual[r] * h->wdiag[r]; }                  /* :689-692 */
static void seobj2(jgo_gn *h, i64 r, i64 r2) { h->objective += h->residual[r] * h->residual[r] * h->wdiag[r] + 2.0 * h->residual[r] * h->residual[r2] * h->woff[r2]; }  /* :694-698 */

/* normalEquation! (acStateEstimation.jl:261-583): column-major walk over the theta columns of H */
void jgo_gn_normal_equation(jgo_gn *h) {
    i64 n = h->n;
    const double *vm = h->vm, *va = h->va;
    h->objective = 0.0;
    for (i64 col = 0; col < n; col++) {
        i64 cok = col + n;
        for (i64 lin = h->hcolptr[col] - 1; lin < h->hcolptr[col + 1] - 1; lin++) {
            i64 row = h->hrowval[lin] - 1;
            int ty = h->type[row];
            if (ty == 0) continue;
            i64 idx = h->index[row] - 1;                             /* bus or branch, 0-based */
            i64 lin2 = stored(h->hcolptr, h->hrowval, row, cok);     /* jcb[row, cok] */
            double *Jt = &h->hval[lin], *Jv = &h->hval[lin2];
            double r = 0.0; int own = 0;
            if (ty == 6 || ty == 9) {
                if (col == idx) {
                    double cT = 0.0, cV = 0.0;
                    for (i64 q = h->colptr[col] - 1; q < h->colptr[col + 1] - 1; q++) {
                        i64 j = h->rowval[q] - 1;
                        double G = h->ytre[q], B = h->ytim[q], th = va[col] - va[j], s = sin(th), c = cos(th);
                        if (ty == 6) { cT += vm[j] * (G * s - B * c); cV += vm[j] * (G * c + B * s); }
                        else { cT += vm[j] * (G * c + B * s); cV += vm[j] * (G * s - B * c); }
                    }
                    i64 pd = stored(h->colptr, h->rowval, col, col);
                    double Gii = h->yre[pd], Bii = h->yim[pd];
                    r = h->mean[row] - vm[col] * cV; own = 1;
                    if (ty == 6) { *Jt = vm[col] * (-cT) - Bii * vm[col] * vm[col]; *Jv = cV + Gii * vm[col]; }      /* :295-296 */
                    else { *Jt = vm[col] * cT - Gii * vm[col] * vm[col]; *Jv = cV - Bii * vm[col]; }                 /* :349-350 */
                } else {
                    i64 p = stored(h->colptr, h->rowval, idx, col);  /* nodalMatrix[idx, col] */
                    double G = h->yre[p], B = h->yim[p], th = va[idx] - va[col], s = sin(th), c = cos(th);
                    if (ty == 6) { *Jt = vm[idx] * vm[col] * (G * s - B * c); *Jv = vm[idx] * (G * c + B * s); }
                    else { *Jt = -vm[idx] * vm[col] * (G * c + B * s); *Jv = vm[idx] * (G * s - B * c); }
                }
            } else if (ty == 16 || ty == 17) {
                double s = sin(va[idx]), c = cos(va[idx]);
                if (ty == 16) { r = h->mean[row] - vm[idx] * c; *Jt = -vm[idx] * s; *Jv = c; h->residual[row] = r; seobj1(h, row); }
                else { r = h->mean[row] - vm[idx] * s; *Jt = vm[idx] * c; *Jv = s; h->residual[row] = r; seobj2(h, row, row - 1); }
                continue;
            } else {
                int isfrom = (col == h->from[idx] - 1);
                StateModel e = st_ij(h, idx);
                double ti = 0.0, vi_ = 0.0, tj = 0.0, vj_ = 0.0, hv = 0.0;   /* partials wrt theta_i, V_i, theta_j, V_j, value */
                int second = 0;
                switch (ty) {
                case 7: { PiModel p = co_Pij(h, idx);
                    hv = p.A * e.Vi * e.Vi - (p.B * e.c + p.C * e.s) * e.Vi * e.Vj;
                    ti = (p.B * e.s - p.C * e.c) * e.Vi * e.Vj; vi_ = 2 * p.A * e.Vi - (p.B * e.c + p.C * e.s) * e.Vj;
                    tj = -ti; vj_ = -(p.B * e.c + p.C * e.s) * e.Vi; break; }
                case 8: { PiModel p = co_Pji(h, idx);
                    hv = p.A * e.Vj * e.Vj - (p.B * e.c - p.C * e.s) * e.Vi * e.Vj;
                    ti = (p.B * e.s + p.C * e.c) * e.Vi * e.Vj; vi_ = (-p.B * e.c + p.C * e.s) * e.Vj;
                    tj = -ti; vj_ = 2 * p.A * e.Vj - (p.B * e.c - p.C * e.s) * e.Vi; break; }
                case 10: { PiModel p = co_Qij(h, idx);
                    hv = -p.A * e.Vi * e.Vi - (p.B * e.s - p.C * e.c) * e.Vi * e.Vj;
                    ti = -(p.B * e.c + p.C * e.s) * e.Vi * e.Vj; vi_ = -2 * p.A * e.Vi - (p.B * e.s - p.C * e.c) * e.Vj;
                    tj = -ti; vj_ = -(p.B * e.s - p.C * e.c) * e.Vi; break; }
                case 11: { PiModel p = co_Qji(h, idx);
                    hv = -p.A * e.Vj * e.Vj + (p.B * e.s + p.C * e.c) * e.Vi * e.Vj;
                    ti = (p.B * e.c - p.C * e.s) * e.Vi * e.Vj; vi_ = (p.B * e.s + p.C * e.c) * e.Vj;
                    tj = -ti; vj_ = -2 * p.A * e.Vj + (p.B * e.s + p.C * e.c) * e.Vi; break; }
                case 2: { PiModel p = co_Iij(h, idx);
                    double Iinv = 1.0 / sqrt(p.A * e.Vi * e.Vi + p.B * e.Vj * e.Vj - 2 * e.Vi * e.Vj * (p.C * e.c - p.D * e.s));
                    hv = 1.0 / Iinv;
                    ti = Iinv * (p.C * e.s + p.D * e.c) * e.Vi * e.Vj; vi_ = Iinv * (p.A * e.Vi - (p.C * e.c - p.D * e.s) * e.Vj);
                    tj = -ti; vj_ = Iinv * (p.B * e.Vj - (p.C * e.c - p.D * e.s) * e.Vi); break; }
                case 4: { PiModel p = co_Iij(h, idx);
                    hv = p.A * e.Vi * e.Vi + p.B * e.Vj * e.Vj - 2 * e.Vi * e.Vj * (p.C * e.c - p.D * e.s);
                    ti = 2 * (p.C * e.s + p.D * e.c) * e.Vi * e.Vj; vi_ = 2 * (p.A * e.Vi - (p.C * e.c - p.D * e.s) * e.Vj);
                    tj = -ti; vj_ = 2 * (p.B * e.Vj - (p.C * e.c - p.D * e.s) * e.Vi); break; }
                case 3: { PiModel p = co_Iji(h, idx);
                    double Iinv = 1.0 / sqrt(p.A * e.Vi * e.Vi + p.B * e.Vj * e.Vj - 2 * e.Vi * e.Vj * (p.C * e.c + p.D * e.s));
                    hv = 1.0 / Iinv;
                    ti = Iinv * (p.C * e.s - p.D * e.c) * e.Vi * e.Vj; vi_ = Iinv * (p.A * e.Vi - (p.C * e.c + p.D * e.s) * e.Vj);
                    tj = -ti; vj_ = Iinv * (p.B * e.Vj - (p.C * e.c + p.D * e.s) * e.Vi); break; }
                case 5: { PiModel p = co_Iji(h, idx);
                    hv = p.A * e.Vi * e.Vi + p.B * e.Vj * e.Vj - 2 * e.Vi * e.Vj * (p.C * e.c + p.D * e.s);
                    ti = 2 * (p.C * e.s - p.D * e.c) * e.Vi * e.Vj; vi_ = 2 * (p.A * e.Vi - (p.C * e.c + p.D * e.s) * e.Vj);
                    tj = -ti; vj_ = 2 * (p.B * e.Vj - (p.C * e.c + p.D * e.s) * e.Vi); break; }
                case 14: { PiModel pp = co_psiij(h, idx); StateModel ee = st_ithj(h, idx);      /* :466-483 */
                    double re = ReIij(pp, ee), im = ImIij(pp, ee), Iinv2 = 1.0 / (re * re + im * im);
                    PiModel p = co_Iij(h, idx);
                    hv = atan2(im, re);
                    ti = Iinv2 * (p.A * e.Vi * e.Vi - (p.C * e.c - p.D * e.s) * e.Vi * e.Vj); vi_ = -Iinv2 * (p.C * e.s + p.D * e.c) * e.Vj;
                    tj = Iinv2 * (p.B * e.Vj * e.Vj - (p.C * e.c - p.D * e.s) * e.Vi * e.Vj); vj_ = Iinv2 * (p.C * e.s + p.D * e.c) * e.Vi; break; }
                case 15: { PiModel pp = co_psiji(h, idx); StateModel ee = st_jthi(h, idx);      /* :485-502 */
                    double re = ReIji(pp, ee), im = ImIji(pp, ee), Iinv2 = 1.0 / (re * re + im * im);
                    PiModel p = co_Iji(h, idx);
                    hv = atan2(im, re);
                    ti = Iinv2 * (p.A * e.Vi * e.Vi - (p.C * e.c + p.D * e.s) * e.Vi * e.Vj); vi_ = -Iinv2 * (p.C * e.s - p.D * e.c) * e.Vj;
                    tj = Iinv2 * (p.B * e.Vj * e.Vj - (p.C * e.c + p.D * e.s) * e.Vi * e.Vj); vj_ = Iinv2 * (p.C * e.s - p.D * e.c) * e.Vi; break; }
                case 18: { PiModel p = co_psiij(h, idx); StateModel ee = st_ithj(h, idx);
                    hv = ReIij(p, ee); ti = -(p.A * ee.si + p.B * ee.ci) * ee.Vi; vi_ = p.A * ee.ci - p.B * ee.si;
                    tj = (p.C * ee.sj + p.D * ee.cj) * ee.Vj; vj_ = -p.C * ee.cj + p.D * ee.sj; break; }
                case 20: { PiModel p = co_psiij(h, idx); StateModel ee = st_ithj(h, idx); second = 1;
                    hv = ImIij(p, ee); ti = (p.A * ee.ci - p.B * ee.si) * ee.Vi; vi_ = p.A * ee.si + p.B * ee.ci;
                    tj = (-p.C * ee.cj + p.D * ee.sj) * ee.Vj; vj_ = -p.C * ee.sj - p.D * ee.cj; break; }
                case 19: { PiModel p = co_psiji(h, idx); StateModel ee = st_jthi(h, idx);
                    hv = ReIji(p, ee); ti = (p.C * ee.si + p.D * ee.ci) * ee.Vi; vi_ = -p.C * ee.ci + p.D * ee.si;
                    tj = -(p.A * ee.sj + p.B * ee.cj) * ee.Vj; vj_ = p.A * ee.cj - p.B * ee.sj; break; }
                case 21: { PiModel p = co_psiji(h, idx); StateModel ee = st_jthi(h, idx); second = 1;
                    hv = ImIji(p, ee); ti = (-p.C * ee.ci + p.D * ee.si) * ee.Vi; vi_ = -p.C * ee.si - p.D * ee.ci;
                    tj = (p.A * ee.cj - p.B * ee.sj) * ee.Vj; vj_ = p.A * ee.sj + p.B * ee.cj; break; }
                default: continue;
                }
                if (isfrom) {
                    h->residual[row] = h->mean[row] - hv;
                    if (second) seobj2(h, row, row - 1); else seobj1(h, row);
                    *Jt = ti; *Jv = vi_;
                } else { *Jt = tj; *Jv = vj_; }
                continue;
            }
            if (own) { h->residual[row] = r; seobj1(h, row); }
        }
    }
    for (i64 row = h->range[0] - 1; row < h->range[1] - 1; row++)                    /* :567-572 */
        if (h->type[row] == 1) { h->residual[row] = h->mean[row] - vm[h->index[row] - 1]; seobj1(h, row); }
    for (i64 row = h->range[4] - 1; row < h->range[5] - 1; row++) {                  /* :574-582 */
        if (h->type[row] == 12) { h->residual[row] = h->mean[row] - vm[h->index[row] - 1]; seobj1(h, row); }
        else if (h->type[row] == 13) { h->residual[row] = h->mean[row] - va[h->index[row] - 1]; seobj1(h, row); }
    }
}

/* G = H' W H pattern + contribution list, built once: for every measurement row, every pair of its
 * stored columns (and the cross terms of correlated pairs) adds into one gain entry. */
static void build_gain_pattern(jgo_gn *h) {
    i64 n2 = 2 * h->n, m = h->m;
    /* row-wise view of H */
    i64 *rptr = (i64 *)calloc((size_t)m + 1, 8);
    for (i64 p = 0; p < h->nnzH; p++) rptr[h->hrowval[p]]++;
    for (i64 r = 0; r < m; r++) rptr[r + 1] += rptr[r];
    i64 *rcol = (i64 *)xmalloc((size_t)h->nnzH * 8), *rpos = (i64 *)xmalloc((size_t)h->nnzH * 8), *fill = (i64 *)xmalloc((size_t)m * 8);
    memcpy(fill, rptr, (size_t)m * 8);
    for (i64 c = 0; c < n2; c++) for (i64 p = h->hcolptr[c] - 1; p < h->hcolptr[c + 1] - 1; p++) { i64 r = h->hrowval[p] - 1; rcol[fill[r]] = c; rpos[fill[r]] = p; fill[r]++; }
    i64 cap = 0;
    for (i64 r = 0; r < m; r++) { i64 d = rptr[r + 1] - rptr[r]; cap += d * d; if (h->woff[r] != 0.0) { i64 d2 = rptr[r + 2] - rptr[r + 1]; cap += 2 * d * d2; } }
    cap += 1;
    i64 *ga = (i64 *)xmalloc((size_t)cap * 8), *gb = (i64 *)xmalloc((size_t)cap * 8), *gr = (i64 *)xmalloc((size_t)cap * 8), *gc = (i64 *)xmalloc((size_t)cap * 8);
    double *gw = (double *)xmalloc((size_t)cap * 8);   /* weight row index (>=0 diag of row, encoded) */
    i64 k = 0;
    for (i64 r = 0; r < m; r++) {
        for (i64 a = rptr[r]; a < rptr[r + 1]; a++) for (i64 b = rptr[r]; b < rptr[r + 1]; b++) { ga[k] = rpos[a]; gb[k] = rpos[b]; gr[k] = rcol[a]; gc[k] = rcol[b]; gw[k] = (double)r; k++; }
        if (h->woff[r] != 0.0)
            for (i64 a = rptr[r]; a < rptr[r + 1]; a++) for (i64 b = rptr[r + 1]; b < rptr[r + 2]; b++) {
                ga[k] = rpos[a]; gb[k] = rpos[b]; gr[k] = rcol[a]; gc[k] = rcol[b]; gw[k] = -(double)(r + 1); k++;
                ga[k] = rpos[b]; gb[k] = rpos[a]; gr[k] = rcol[b]; gc[k] = rcol[a]; gw[k] = -(double)(r + 1); k++;
            }
    }
    h->ngmap = k;
    /* slack diagonal must exist */
    i64 sl = h->slack - 1;
    /* CSC pattern of G */
    i64 *cnt = (i64 *)calloc((size_t)n2 + 1, 8);
    for (i64 t = 0; t < k; t++) cnt[gc[t] + 1]++;
    cnt[sl + 1]++;
    i64 *start = (i64 *)xmalloc(((size_t)n2 + 1) * 8); start[0] = 0;
    for (i64 c = 0; c < n2; c++) start[c + 1] = start[c] + cnt[c + 1];
    i64 tot = start[n2];
    i64 *rows = (i64 *)xmalloc((size_t)tot * 8), *f2 = (i64 *)xmalloc((size_t)n2 * 8);
    memcpy(f2, start, (size_t)n2 * 8);
    for (i64 t = 0; t < k; t++) rows[f2[gc[t]]++] = gr[t];
    rows[f2[sl]++] = sl;
    h->gcolptr = (i64 *)xmalloc(((size_t)n2 + 1) * 8);
    i64 *uniq = (i64 *)xmalloc((size_t)tot * 8); i64 u = 0;
    h->gcolptr[0] = 0;
    for (i64 c = 0; c < n2; c++) {
        i64 lo = start[c], hi = start[c + 1];
        /* sort rows of this column (small lists) */
        for (i64 a = lo + 1; a < hi; a++) { i64 v = rows[a], b = a - 1; while (b >= lo && rows[b] > v) { rows[b + 1] = rows[b]; b--; } rows[b + 1] = v; }
        for (i64 a = lo; a < hi; a++) if (a == lo || rows[a] != rows[a - 1]) uniq[u++] = rows[a];
        h->gcolptr[c + 1] = u;
    }
    h->gnnz = u;
    h->growval = (i64 *)xmalloc((size_t)u * 8); memcpy(h->growval, uniq, (size_t)u * 8);
    h->gval = (double *)calloc((size_t)u, 8);
    h->gmap_a = ga; h->gmap_b = gb; h->gmap_w_is_off = gw;
    h->gmap_pos = (i64 *)xmalloc((size_t)k * 8);
    for (i64 t = 0; t < k; t++) {
        i64 lo = h->gcolptr[gc[t]], hi = h->gcolptr[gc[t] + 1] - 1, pos = -1;
        while (lo <= hi) { i64 mid = (lo + hi) >> 1; if (h->growval[mid] < gr[t]) lo = mid + 1; else if (h->growval[mid] > gr[t]) hi = mid - 1; else { pos = mid; break; } }
        h->gmap_pos[t] = pos;
    }
    free(rptr); free(rcol); free(rpos); free(fill); free(gr); free(gc); free(cnt); free(start); free(rows); free(f2); free(uniq);
}

/* increment!{Normal} (acStateEstimation.jl:878-904). Returns max |increment|; rc: 0 ok, 3 singular. */
int jgo_gn_increment(jgo_gn *h, double *maxinc) {
    i64 n2 = 2 * h->n, sl = h->slack - 1;
    jgo_gn_normal_equation(h);                                                       /* :883 */
    if (!h->gcolptr) build_gain_pattern(h);
    /* removeColumn(H, slack) (:885, sparse.jl:155-163) */
    i64 c0 = h->hcolptr[sl] - 1, c1 = h->hcolptr[sl + 1] - 1;
    double *saved = (double *)xmalloc((size_t)(c1 - c0 + 1) * 8);
    for (i64 p = c0; p < c1; p++) { saved[p - c0] = h->hval[p]; h->hval[p] = 0.0; }
    /* gain = H' W H (:887-888) */
    memset(h->gval, 0, (size_t)h->gnnz * 8);
    for (i64 t = 0; t < h->ngmap; t++) {
        double wq = h->gmap_w_is_off[t];
        double w = wq >= 0 ? h->wdiag[(i64)wq] : h->woff[(i64)(-wq) - 1];
        h->gval[h->gmap_pos[t]] += h->hval[h->gmap_a[t]] * w * h->hval[h->gmap_b[t]];
    }
    {   /* gain[slack, slack] = 1 (:889) */
        i64 lo = h->gcolptr[sl], hi = h->gcolptr[sl + 1] - 1;
        while (lo <= hi) { i64 mid = (lo + hi) >> 1; if (h->growval[mid] < sl) lo = mid + 1; else if (h->growval[mid] > sl) hi = mid - 1; else { h->gval[mid] = 1.0; break; } }
    }
    /* rhs = H' W r (:898) */
    double *rhs = (double *)calloc((size_t)n2, 8);
    double *wr = (double *)calloc((size_t)h->m + 1, 8);
    for (i64 r = 0; r < h->m; r++) {
        wr[r] += h->wdiag[r] * h->residual[r];
        if (h->woff[r] != 0.0) { wr[r] += h->woff[r] * h->residual[r + 1]; wr[r + 1] += h->woff[r] * h->residual[r]; }
    }
    for (i64 c = 0; c < n2; c++) { double s = 0.0; for (i64 p = h->hcolptr[c] - 1; p < h->hcolptr[c + 1] - 1; p++) s += h->hval[p] * wr[h->hrowval[p] - 1]; rhs[c] = s; }
    int rc = h->factored ? jgo_lu_refactor(h->F, h->gcolptr, h->growval, h->gval)    /* :891-896 */
                         : jgo_lu_factor(h->F, h->gcolptr, h->growval, h->gval);
    if (!rc) {
        h->factored = 1;
        jgo_lu_solve(h->F, rhs, h->increment);
        h->increment[sl] = 0.0;                                                      /* :899 */
    }
    for (i64 p = c0; p < c1; p++) h->hval[p] = saved[p - c0];                        /* restoreColumn! :901 */
    free(saved); free(rhs); free(wr);
    if (rc) return rc;
    double mx = 0.0;
    for (i64 i = 0; i < n2; i++) { double a = fabs(h->increment[i]); if (a > mx || a != a) mx = a; }
    *maxinc = mx;
    return 0;
}

/* solve! (:1035-1047) */
void jgo_gn_solve(jgo_gn *h) {
    for (i64 i = 0; i < h->n; i++) { h->va[i] += h->increment[i]; h->vm[i] += h->increment[i + h->n]; }
    h->iteration++;
}

/* stateEstimation! (:1286-1329). status 0 converged, 1 iteration limit, 3 singular gain. */
int jgo_gn_state_estimation(jgo_gn *h, i64 maxit, double tol, double *history, i64 *nhist) {
    h->iteration = 0;
    i64 nh = 0; int status = 1;
    for (i64 iter = 0; iter <= maxit; iter++) {
        double mx = 0.0;
        int rc = jgo_gn_increment(h, &mx);
        if (rc) { status = rc; break; }
        if (history) history[nh] = mx;
        nh++;
        if (mx < tol) { status = 0; break; }
        if (h->iteration == maxit) { status = 1; break; }
        jgo_gn_solve(h);
    }
    if (nhist) *nhist = nh;
    return status;
}
