/*
 * jg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference algorithm for the Newton-Raphson AC power-flow path of
 * mcosovic/JuliaGrid.jl v0.6.2.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (juliagrid.jl_amd/ + libjgrid_hip.so) never does.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the reference's own MATPOWER
 * goldens (test/data/results.h5: case14test NR = 7 iterations, case30test NR = 4, V/theta vectors)
 * exported to tests/golden/results_*.npz by tools/make_fixtures.py.
 *
 * Third-party arithmetic absent from /root/reference: the reference factorises the Jacobian with
 * SuiteSparse (UMFPACK `lu`/`lu!` by default, KLU.jl 0.6 `klu`/`klu!`; call sites
 * src/backend/utility.jl:470-516, 576-586).  The sparse LU below restates the PUBLISHED KLU
 * algorithm (Davis & Palamadai Natarajan, ACM TOMS 37(3), 2010): fill-reducing column pre-ordering
 * (minimum degree on A+A'), left-looking Gilbert-Peierls numeric factorisation with threshold
 * partial pivoting that prefers the diagonal (tol 1e-3), and `klu_refactor`-style numeric
 * refactorisation that re-uses pattern and pivot order.  L/U factors are parity-unpinned in the
 * reference (no test inspects them); converged states and iteration counts are what is pinned.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * All indices crossing this API are 1-based int64, exactly as the Julia containers hold them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* ------------------------------------------------------------------------------------------- */
/* PF0: acModel!  (src/powerSystem/model.jl:23-78, CSC builder src/backend/sparse.jl:2-101)     */
/* ------------------------------------------------------------------------------------------- */

/* 1/(r+jx) by Smith's algorithm (what `1 / impedance` amounts to, model.jl:54-55). */
static void cinv(double r, double x, double *re, double *im) {
    if (fabs(r) >= fabs(x)) {
        double q = x / r, d = r + x * q;
        *re = 1.0 / d; *im = -q / d;
    } else {
        double q = r / x, d = r * q + x;
        *re = q / d; *im = -1.0 / d;
    }
}

/* Two-port parameters of one in-service branch (model.jl:54-64). out = {y, yff, yft, ytt, ytf} re/im pairs */
static void two_port(double r, double x, double g, double b, double tap, double shift, double *o) {
    double yre, yim; cinv(r, x, &yre, &yim);
    double tinv = 1.0 / tap;
    double tre = tinv * cos(-shift), tim = tinv * sin(-shift);     /* turnsRatioInv * cis(-shift) */
    double ttre = yre + 0.5 * g, ttim = yim + 0.5 * b;             /* nodalToTo   model.jl:61 */
    double t2 = tinv * tinv;
    o[0] = yre; o[1] = yim;
    o[2] = t2 * ttre; o[3] = t2 * ttim;                            /* nodalFromFrom :62 */
    /* nodalFromTo = -conj(t) * y   :63 */
    o[4] = -(tre * yre + tim * yim); o[5] = -(tre * yim - tim * yre);
    o[6] = ttre; o[7] = ttim;
    /* nodalToFrom = -t * y   :64 */
    o[8] = -(tre * yre - tim * yim); o[9] = -(tre * yim + tim * yre);
}

/*
 * Builds Ybus (CSC, sorted rows, duplicates summed in insertion order, stored zeros for
 * out-of-service branches), the transpose value array (same pattern, value p = Y[col,row]) and the
 * per-branch two-port arrays twoport[nb*10] = {y, yff, yft, ytt, ytf}.
 * colptr[n+1], rowval/yre/yim/ytre/ytim sized n+2*nb by the caller.  Returns nnz.
 */
i64 jgo_ac_model(i64 n, i64 nb, const i64 *from, const i64 *to, const int8_t *status,
                 const double *r, const double *x, const double *g, const double *b,
                 const double *tap, const double *shift, const double *gs, const double *bs,
                 i64 *colptr, i64 *rowval, double *yre, double *yim, double *ytre, double *ytim,
                 double *twoport) {
    i64 *deg = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    for (i64 i = 0; i < n; i++) deg[i] = 1;                         /* model.jl:36-40 */
    for (i64 k = 0; k < nb; k++) { deg[from[k] - 1]++; deg[to[k] - 1]++; }
    i64 *start = (i64 *)malloc(((size_t)n + 1) * sizeof(i64));
    start[0] = 0;
    for (i64 i = 0; i < n; i++) start[i + 1] = start[i] + deg[i];
    i64 cap = start[n];
    i64 *fill = (i64 *)malloc((size_t)n * sizeof(i64));
    i64 *brow = (i64 *)malloc((size_t)cap * sizeof(i64));
    double *bre = (double *)malloc((size_t)cap * sizeof(double));
    double *bim = (double *)malloc((size_t)cap * sizeof(double));
    i64 *diag = (i64 *)malloc((size_t)n * sizeof(i64));
    for (i64 i = 0; i < n; i++) {                                   /* model.jl:44-47 */
        fill[i] = start[i];
        diag[i] = fill[i];
        brow[fill[i]] = i; bre[fill[i]] = gs[i]; bim[fill[i]] = bs[i]; fill[i]++;
    }
    memset(twoport, 0, (size_t)nb * 10 * sizeof(double));
    for (i64 k = 0; k < nb; k++) {                                  /* model.jl:49-72 */
        i64 f = from[k] - 1, t = to[k] - 1;
        double *o = twoport + 10 * k;
        if (status[k] == 1) {
            two_port(r[k], x[k], g[k], b[k], tap[k], shift[k], o);
            bre[diag[f]] += o[2]; bim[diag[f]] += o[3];
            bre[diag[t]] += o[6]; bim[diag[t]] += o[7];
        }
        /* addEntry!(from,to) goes to column `to`; addEntry!(to,from) to column `from` */
        brow[fill[t]] = f; bre[fill[t]] = o[4]; bim[fill[t]] = o[5]; fill[t]++;
        brow[fill[f]] = t; bre[fill[f]] = o[8]; bim[fill[f]] = o[9]; fill[f]++;
    }
    /* canonicalize! (sparse.jl:43-95): stable insertion sort per column, sum duplicates */
    i64 count = 0;
    colptr[0] = 1;
    for (i64 c = 0; c < n; c++) {
        i64 lo = start[c], hi = start[c + 1];
        for (i64 j = lo + 1; j < hi; j++) {
            i64 rw = brow[j]; double vr = bre[j], vi = bim[j];
            i64 k = j - 1;
            while (k >= lo && brow[k] > rw) { brow[k + 1] = brow[k]; bre[k + 1] = bre[k]; bim[k + 1] = bim[k]; k--; }
            brow[k + 1] = rw; bre[k + 1] = vr; bim[k + 1] = vi;
        }
        i64 p = lo;
        while (p < hi) {
            i64 rw = brow[p]; double vr = bre[p], vi = bim[p];
            p++;
            while (p < hi && brow[p] == rw) { vr += bre[p]; vi += bim[p]; p++; }
            rowval[count] = rw + 1; yre[count] = vr; yim[count] = vi; count++;
        }
        colptr[c + 1] = count + 1;
    }
    /* nodalMatrixTranspose = copy(transpose(Y)) (model.jl:75); symmetric pattern -> same colptr/rowval */
    for (i64 c = 0; c < n; c++)
        for (i64 p = colptr[c] - 1; p < colptr[c + 1] - 1; p++) {
            i64 rw = rowval[p] - 1;                                  /* entry (rw, c); find (c, rw) in column rw */
            i64 lo = colptr[rw] - 1, hi = colptr[rw + 1] - 2;
            i64 q = -1;
            while (lo <= hi) { i64 m = (lo + hi) >> 1; if (rowval[m] - 1 < c) lo = m + 1; else if (rowval[m] - 1 > c) hi = m - 1; else { q = m; break; } }
            ytre[p] = yre[q]; ytim[p] = yim[q];
        }
    free(deg); free(start); free(fill); free(brow); free(bre); free(bim); free(diag);
    return count;
}

/* ------------------------------------------------------------------------------------------- */
/* PF1: initializeACPowerFlow + changeSlackBus!  (src/powerFlow/acPowerFlow.jl:1312-1358)      */
/* ------------------------------------------------------------------------------------------- */
/*
 * first_gen[i] = 1-based index of the first in-service generator listed for bus i (0 = none),
 * i.e. bus.supply.generator[i][1] (load.jl:271-277).  Mutates type/slack like the reference.
 * Returns the (possibly relocated) slack, or 0 for errorSlackDefinition().
 */
i64 jgo_initialize_ac_power_flow(i64 n, int8_t *type, i64 slack, const i64 *first_gen,
                                 const double *gen_vm, const double *bus_vm, const double *bus_va,
                                 double *vm, double *va) {
    for (i64 i = 0; i < n; i++) { vm[i] = bus_vm[i]; va[i] = bus_va[i]; }   /* :1315-1316 */
    for (i64 i = 0; i < n; i++) {
        if (!first_gen[i] && type[i] == 2) type[i] = 1;                       /* :1319-1322 */
        if (first_gen[i] && type[i] != 1) vm[i] = gen_vm[first_gen[i] - 1];   /* :1323-1325 */
    }
    if (!first_gen[slack - 1]) {                                              /* :1335-1357 */
        type[slack - 1] = 1;
        for (i64 i = 0; i < n; i++)
            if (type[i] == 2 && first_gen[i]) { type[i] = 3; slack = i + 1; break; }
        if (type[slack - 1] == 1) return 0;
    }
    return slack;
}

/* ------------------------------------------------------------------------------------------- */
/* PF2: newtonJacobian  (src/powerFlow/acPowerFlow.jl:89-175) -- integer only, bit-exact        */
/* ------------------------------------------------------------------------------------------- */
/* pass jrowval == NULL to size: returns nnz(J); *dim = dimJ.  jcolptr sized 2n. */
i64 jgo_newton_jacobian(i64 n, const i64 *colptr, const i64 *rowval, const int8_t *type, i64 slack,
                        i64 *pq, i64 *pvpq, i64 *pcount, i64 *dim, i64 *jcolptr, i64 *jrowval) {
    i64 pvpqNum = 0, pqNum = 0;
    for (i64 i = 0; i < n; i++) {                                   /* :93-106 */
        pq[i] = 0; pvpq[i] = 0;
        if (type[i] == 1) { pqNum++; pq[i] = pqNum + n - 1; }
        if (type[i] != 3) { pvpqNum++; pvpq[i] = pvpqNum; }
    }
    i64 dimJ = n + pqNum - 1;                                        /* :108 */
    *dim = dimJ;
    i64 *colcount = (i64 *)calloc((size_t)dimJ + 1, sizeof(i64));
    i64 *qcount = (i64 *)calloc((size_t)n, sizeof(i64));
    for (i64 i = 0; i < n; i++) pcount[i] = 0;
    for (i64 i = 0; i < n; i++) {                                   /* :113-130 */
        if (i + 1 == slack) continue;
        for (i64 p = colptr[i] - 1; p < colptr[i + 1] - 1; p++) {
            int8_t tr = type[rowval[p] - 1];
            if (tr != 3) pcount[i]++;
            if (tr == 1) qcount[i]++;
        }
        colcount[pvpq[i] - 1] = pcount[i] + qcount[i];
        if (type[i] == 1) colcount[pq[i] - 1] = pcount[i] + qcount[i];
    }
    jcolptr[0] = 1;                                                  /* :132-136 */
    for (i64 c = 0; c < dimJ; c++) jcolptr[c + 1] = jcolptr[c] + colcount[c];
    i64 nnzJ = jcolptr[dimJ] - 1;
    if (jrowval) {
        for (i64 i = 0; i < n; i++) {                               /* :142-172 */
            if (i + 1 == slack) continue;
            int isPQ = type[i] == 1;
            i64 pA = jcolptr[pvpq[i] - 1], qA = pA + pcount[i];
            i64 pM = isPQ ? jcolptr[pq[i] - 1] : 0, qM = isPQ ? pM + pcount[i] : 0;
            for (i64 p = colptr[i] - 1; p < colptr[i + 1] - 1; p++) {
                i64 row = rowval[p] - 1; int8_t tr = type[row];
                if (tr != 3) { jrowval[pA - 1] = pvpq[row]; pA++; if (isPQ) { jrowval[pM - 1] = pvpq[row]; pM++; } }
                if (tr == 1) { jrowval[qA - 1] = pq[row]; qA++; if (isPQ) { jrowval[qM - 1] = pq[row]; qM++; } }
            }
        }
    }
    free(colcount); free(qcount);
    return nnzJ;
}

/* ------------------------------------------------------------------------------------------- */
/* PF4: mismatch!  (src/powerFlow/acPowerFlow.jl:645-685; equations.jl:63-68, 78-103, 126-128)  */
/* ------------------------------------------------------------------------------------------- */
void jgo_mismatch(i64 n, const i64 *colptr, const i64 *rowval, const double *ytre, const double *ytim,
                  const int8_t *type, i64 slack, const i64 *pq, const i64 *pvpq,
                  const double *vm, const double *va, const double *p_supply, const double *q_supply,
                  const double *p_demand, const double *q_demand, double *mism, double *stop) {
    double stopP = 0.0, stopQ = 0.0;
    for (i64 i = 0; i < n; i++) {
        if (i + 1 == slack) continue;
        double cP = 0.0, cQ = 0.0;
        int isPQ = type[i] == 1;
        for (i64 p = colptr[i] - 1; p < colptr[i + 1] - 1; p++) {
            i64 row = rowval[p] - 1;
            double G = ytre[p], B = ytim[p];                        /* nodalMatrixTranspose.nzval[q] */
            double th = va[i] - va[row], s = sin(th), c = cos(th);
            cP += vm[row] * (G * c + B * s);                        /* PiQiSumPlus  */
            if (isPQ) cQ += vm[row] * (G * s - B * c);              /* PiQiSumMinus */
        }
        i64 k = pvpq[i] - 1;
        mism[k] = vm[i] * cP - p_supply[i] + p_demand[i];           /* :676 */
        if (fabs(mism[k]) > stopP) stopP = fabs(mism[k]);
        if (isPQ) {
            i64 q = pq[i] - 1;
            mism[q] = vm[i] * cQ - q_supply[i] + q_demand[i];       /* :679 */
            if (fabs(mism[q]) > stopQ) stopQ = fabs(mism[q]);
        }
    }
    stop[0] = stopP; stop[1] = stopQ;
}

/* ------------------------------------------------------------------------------------------- */
/* PF5: Jacobian fill of solve!  (src/powerFlow/acPowerFlow.jl:820-888; equations.jl:105-144)   */
/* ------------------------------------------------------------------------------------------- */
void jgo_jacobian_fill(i64 n, const i64 *colptr, const i64 *rowval, const double *yre, const double *yim,
                       const double *ytre, const double *ytim, const int8_t *type, i64 slack,
                       const i64 *pq, const i64 *pvpq, const i64 *pcount, const i64 *jcolptr,
                       const double *vm, const double *va, double *nz) {
    for (i64 i = 0; i < n; i++) {
        if (i + 1 == slack) continue;
        int isPQ = type[i] == 1;
        i64 pA = jcolptr[pvpq[i] - 1] - 1, qA = pA + pcount[i];
        i64 pM = isPQ ? jcolptr[pq[i] - 1] - 1 : 0, qM = isPQ ? pM + pcount[i] : 0;
        for (i64 j = colptr[i] - 1; j < colptr[i + 1] - 1; j++) {
            i64 row = rowval[j] - 1; int8_t tr = type[row];
            if (tr == 3) continue;
            double G = yre[j], B = yim[j];                          /* nodalMatrix.nzval[j] = Y[row,i] */
            if (row != i) {
                double th = va[row] - va[i], s = sin(th), c = cos(th);
                nz[pA++] = vm[row] * vm[i] * (G * s - B * c);                      /* Pitheta_j */
                if (tr == 1) nz[qA++] = -vm[row] * vm[i] * (G * c + B * s);        /* Qitheta_j */
                if (isPQ) nz[pM++] = vm[row] * (G * c + B * s);                    /* PiVj */
                if (isPQ && tr == 1) nz[qM++] = vm[row] * (G * s - B * c);         /* QiVj */
            } else {
                double cT = 0.0, cV = 0.0;                                          /* :859-870 */
                for (i64 p = colptr[i] - 1; p < colptr[i + 1] - 1; p++) {
                    i64 q = rowval[p] - 1;
                    double Gk = ytre[p], Bk = ytim[p];
                    double th = va[i] - va[q], s = sin(th), c = cos(th);
                    cT += vm[q] * (Gk * s - Bk * c);
                    if (isPQ) cV += vm[q] * (Gk * c + Bk * s);
                }
                nz[pA++] = vm[i] * (-cT) - B * vm[i] * vm[i];                      /* Pitheta_i :872 */
                if (isPQ) nz[qA++] = vm[i] * cV - G * vm[i] * vm[i];               /* Qitheta_i :875 */
                if (isPQ) nz[pM++] = cV + G * vm[i];                               /* PiVi :879 */
                if (isPQ) nz[qM++] = cT - B * vm[i];                               /* QiVi :883 */
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* PF6: sparse LU with symbolic reuse -- restatement of the published KLU algorithm             */
/*      (reference call sites: src/backend/utility.jl:470-516 `lu`/`lu!`/`klu`/`klu!`, :576-586)*/
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    i64 n;
    i64 *q;            /* column pre-ordering: column k of the factor is column q[k] of A */
    i64 *pinv;         /* row i of A is row pinv[i] of the factor */
    i64 *Lp, *Li; double *Lx;   /* unit lower, column-compressed, rows in factor order (strict part) */
    i64 *Up, *Ui; double *Ux;   /* upper, strict part in TOPOLOGICAL order per column; */
    double *Ud;                 /* pivots */
    double *work; i64 *xi, *mark;
    i64 lcap, ucap;
    int factored;
} jgo_lu;

static void *xrealloc(void *p, size_t s) { void *q = realloc(p, s); if (!q) abort(); return q; }

/* minimum-degree ordering of the pattern of A+A' (elimination graph with explicit sets) */
static void min_degree(i64 n, const i64 *Ap, const i64 *Ai, i64 *perm) {
    i64 *len = (i64 *)calloc((size_t)n, sizeof(i64)), *cap = (i64 *)calloc((size_t)n, sizeof(i64));
    i64 **adj = (i64 **)calloc((size_t)n, sizeof(i64 *));
    /* build symmetric adjacency (no self loops, duplicates removed later by marker) */
    for (i64 c = 0; c < n; c++) for (i64 p = Ap[c]; p < Ap[c + 1]; p++) { i64 r = Ai[p]; if (r != c) { len[r]++; len[c]++; } }
    for (i64 i = 0; i < n; i++) { cap[i] = len[i] + 4; adj[i] = (i64 *)malloc((size_t)cap[i] * sizeof(i64)); len[i] = 0; }
    for (i64 c = 0; c < n; c++) for (i64 p = Ap[c]; p < Ap[c + 1]; p++) { i64 r = Ai[p]; if (r != c) { adj[r][len[r]++] = c; adj[c][len[c]++] = r; } }
    i64 *mark = (i64 *)malloc((size_t)n * sizeof(i64));
    for (i64 i = 0; i < n; i++) mark[i] = -1;
    for (i64 i = 0; i < n; i++) {              /* dedupe */
        i64 m = 0;
        for (i64 k = 0; k < len[i]; k++) { i64 u = adj[i][k]; if (mark[u] != i) { mark[u] = i; adj[i][m++] = u; } }
        len[i] = m;
    }
    for (i64 i = 0; i < n; i++) mark[i] = -1;
    i64 stamp = 0;
    char *done = (char *)calloc((size_t)n, 1);
    /* bucket lists by degree */
    i64 *head = (i64 *)malloc(((size_t)n + 1) * sizeof(i64)), *next = (i64 *)malloc((size_t)n * sizeof(i64)), *prev = (i64 *)malloc((size_t)n * sizeof(i64));
    i64 *deg = (i64 *)malloc((size_t)n * sizeof(i64));
    for (i64 d = 0; d <= n; d++) head[d] = -1;
#define BUCKET_INSERT(v) do { i64 d_ = deg[v]; next[v] = head[d_]; prev[v] = -1; if (head[d_] >= 0) prev[head[d_]] = v; head[d_] = v; } while (0)
#define BUCKET_REMOVE(v) do { if (prev[v] >= 0) next[prev[v]] = next[v]; else head[deg[v]] = next[v]; if (next[v] >= 0) prev[next[v]] = prev[v]; } while (0)
    for (i64 i = n - 1; i >= 0; i--) { deg[i] = len[i]; BUCKET_INSERT(i); }
    i64 mind = 0;
    for (i64 k = 0; k < n; k++) {
        while (head[mind] < 0) mind++;
        i64 v = head[mind];
        BUCKET_REMOVE(v);
        done[v] = 1; perm[k] = v;
        /* neighbours of v that are still alive */
        i64 m = 0;
        for (i64 t = 0; t < len[v]; t++) { i64 u = adj[v][t]; if (!done[u]) adj[v][m++] = u; }
        len[v] = m;
        for (i64 t = 0; t < m; t++) {
            i64 u = adj[v][t];
            BUCKET_REMOVE(u);
            /* adj[u] = (adj[u] \ {v, dead}) U (adj[v] \ {u}) */
            stamp++;
            i64 mm = 0;
            for (i64 s = 0; s < len[u]; s++) { i64 w = adj[u][s]; if (!done[w]) { mark[w] = stamp; adj[u][mm++] = w; } }
            len[u] = mm;
            for (i64 s = 0; s < m; s++) {
                i64 w = adj[v][s];
                if (w == u || mark[w] == stamp) continue;
                if (len[u] == cap[u]) { cap[u] = cap[u] * 2 + 4; adj[u] = (i64 *)xrealloc(adj[u], (size_t)cap[u] * sizeof(i64)); }
                adj[u][len[u]++] = w; mark[w] = stamp;
            }
            deg[u] = len[u];
            BUCKET_INSERT(u);
            if (deg[u] < mind) mind = deg[u];
        }
        free(adj[v]); adj[v] = NULL;
    }
#undef BUCKET_INSERT
#undef BUCKET_REMOVE
    free(len); free(cap); free(adj); free(mark); free(done); free(head); free(next); free(prev); free(deg);
}

jgo_lu *jgo_lu_create(i64 n) {
    jgo_lu *F = (jgo_lu *)calloc(1, sizeof(jgo_lu));
    F->n = n;
    F->q = (i64 *)malloc((size_t)n * sizeof(i64)); F->pinv = (i64 *)malloc((size_t)n * sizeof(i64));
    F->Lp = (i64 *)calloc((size_t)n + 1, sizeof(i64)); F->Up = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    F->Ud = (double *)calloc((size_t)n, sizeof(double));
    F->work = (double *)calloc((size_t)n, sizeof(double));
    F->xi = (i64 *)malloc(2 * (size_t)n * sizeof(i64)); F->mark = (i64 *)malloc((size_t)n * sizeof(i64));
    return F;
}

void jgo_lu_destroy(jgo_lu *F) {
    if (!F) return;
    free(F->q); free(F->pinv); free(F->Lp); free(F->Li); free(F->Lx); free(F->Up); free(F->Ui); free(F->Ux);
    free(F->Ud); free(F->work); free(F->xi); free(F->mark); free(F);
}

/* depth-first reach of column pattern in the graph of L (Gilbert-Peierls); returns top */
static i64 gp_reach(jgo_lu *F, i64 k, const i64 *Ap, const i64 *Ai, i64 col) {
    i64 n = F->n, top = n;
    i64 *xi = F->xi, *pstack = F->xi + n, *mark = F->mark;
    for (i64 p = Ap[col]; p < Ap[col + 1]; p++) {
        i64 j0 = Ai[p];
        if (mark[j0] == k) continue;
        i64 head = 0; xi[0] = j0;
        while (head >= 0) {
            i64 j = xi[head];
            i64 jnew = F->pinv[j];
            if (mark[j] != k) { mark[j] = k; pstack[head] = (jnew < 0) ? 0 : F->Lp[jnew]; }
            int done = 1;
            i64 p2 = (jnew < 0) ? 0 : F->Lp[jnew + 1];
            for (i64 pp = pstack[head]; pp < p2; pp++) {
                i64 i = F->Li[pp];
                if (mark[i] == k) continue;
                pstack[head] = pp + 1;
                xi[++head] = i; done = 0; break;
            }
            if (done) { head--; xi[--top] = j; }
        }
    }
    return top;
}

/* lu(A): ordering + Gilbert-Peierls with threshold partial pivoting (0-based CSC in). 0 ok, 3 singular */
int jgo_lu_factor(jgo_lu *F, const i64 *Ap, const i64 *Ai, const double *Ax) {
    i64 n = F->n;
    const double tol = 1e-3;
    min_degree(n, Ap, Ai, F->q);
    for (i64 i = 0; i < n; i++) { F->pinv[i] = -1; F->mark[i] = -1; F->work[i] = 0.0; }
    F->lcap = 4 * Ap[n] + n; F->ucap = 4 * Ap[n] + n;
    F->Li = (i64 *)xrealloc(F->Li, (size_t)F->lcap * sizeof(i64)); F->Lx = (double *)xrealloc(F->Lx, (size_t)F->lcap * sizeof(double));
    F->Ui = (i64 *)xrealloc(F->Ui, (size_t)F->ucap * sizeof(i64)); F->Ux = (double *)xrealloc(F->Ux, (size_t)F->ucap * sizeof(double));
    i64 lnz = 0, unz = 0;
    double *x = F->work;
    for (i64 k = 0; k < n; k++) {
        F->Lp[k] = lnz; F->Up[k] = unz;
        if (lnz + n > F->lcap) { F->lcap = 2 * F->lcap + n; F->Li = (i64 *)xrealloc(F->Li, (size_t)F->lcap * sizeof(i64)); F->Lx = (double *)xrealloc(F->Lx, (size_t)F->lcap * sizeof(double)); }
        if (unz + n > F->ucap) { F->ucap = 2 * F->ucap + n; F->Ui = (i64 *)xrealloc(F->Ui, (size_t)F->ucap * sizeof(i64)); F->Ux = (double *)xrealloc(F->Ux, (size_t)F->ucap * sizeof(double)); }
        i64 col = F->q[k];
        i64 top = gp_reach(F, k, Ap, Ai, col);
        for (i64 p = top; p < n; p++) x[F->xi[p]] = 0.0;
        for (i64 p = Ap[col]; p < Ap[col + 1]; p++) x[Ai[p]] = Ax[p];
        /* sparse triangular solve in topological order */
        for (i64 px = top; px < n; px++) {
            i64 j = F->xi[px], jnew = F->pinv[j];
            if (jnew < 0) continue;
            double xj = x[j];
            F->Ui[unz] = jnew; F->Ux[unz++] = xj;
            for (i64 p = F->Lp[jnew]; p < F->Lp[jnew + 1]; p++) x[F->Li[p]] -= F->Lx[p] * xj;
        }
        /* pivot search among non-pivotal rows; prefer the diagonal (row == col) */
        double amax = -1.0; i64 ipiv = -1;
        for (i64 px = top; px < n; px++) {
            i64 i = F->xi[px];
            if (F->pinv[i] >= 0) continue;
            double a = fabs(x[i]);
            if (a > amax) { amax = a; ipiv = i; }
        }
        if (ipiv < 0 || amax <= 0.0 || amax != amax) return 3;
        if (F->pinv[col] < 0 && F->mark[col] == k && fabs(x[col]) >= tol * amax) ipiv = col;
        double pivot = x[ipiv];
        F->Ud[k] = pivot;
        F->pinv[ipiv] = k;
        for (i64 px = top; px < n; px++) {
            i64 i = F->xi[px];
            if (F->pinv[i] < 0) { F->Li[lnz] = i; F->Lx[lnz++] = x[i] / pivot; }
            x[i] = 0.0;
        }
    }
    F->Lp[n] = lnz; F->Up[n] = unz;
    for (i64 p = 0; p < lnz; p++) F->Li[p] = F->pinv[F->Li[p]];     /* rows -> factor order */
    F->factored = 1;
    return 0;
}

/* lu!(F, A): numeric refactorisation, same pattern and pivot order (klu_refactor). */
int jgo_lu_refactor(jgo_lu *F, const i64 *Ap, const i64 *Ai, const double *Ax) {
    i64 n = F->n; double *x = F->work;
    if (!F->factored) return jgo_lu_factor(F, Ap, Ai, Ax);
    for (i64 k = 0; k < n; k++) {
        i64 col = F->q[k];
        for (i64 p = Ap[col]; p < Ap[col + 1]; p++) x[F->pinv[Ai[p]]] = Ax[p];
        for (i64 p = F->Up[k]; p < F->Up[k + 1]; p++) {
            i64 j = F->Ui[p]; double xj = x[j];
            F->Ux[p] = xj; x[j] = 0.0;
            for (i64 pp = F->Lp[j]; pp < F->Lp[j + 1]; pp++) x[F->Li[pp]] -= F->Lx[pp] * xj;
        }
        double pivot = x[k];
        if (pivot == 0.0 || pivot != pivot) return 3;
        F->Ud[k] = pivot; x[k] = 0.0;
        for (i64 p = F->Lp[k]; p < F->Lp[k + 1]; p++) { i64 i = F->Li[p]; F->Lx[p] = x[i] / pivot; x[i] = 0.0; }
    }
    return 0;
}

/* ldiv!(x, F, b) */
void jgo_lu_solve(jgo_lu *F, const double *b, double *xout) {
    i64 n = F->n; double *y = F->work;
    for (i64 i = 0; i < n; i++) y[F->pinv[i]] = b[i];
    for (i64 k = 0; k < n; k++) { double yk = y[k]; for (i64 p = F->Lp[k]; p < F->Lp[k + 1]; p++) y[F->Li[p]] -= F->Lx[p] * yk; }
    for (i64 k = n - 1; k >= 0; k--) {
        y[k] /= F->Ud[k];
        double yk = y[k];
        for (i64 p = F->Up[k]; p < F->Up[k + 1]; p++) y[F->Ui[p]] -= F->Ux[p] * yk;
    }
    for (i64 k = 0; k < n; k++) { xout[F->q[k]] = y[k]; y[k] = 0.0; }
}

i64 jgo_lu_nnz(jgo_lu *F) { return F->Lp[F->n] + F->Up[F->n] + F->n; }

/* ------------------------------------------------------------------------------------------- */
/* PF3/PF5-PF8: NewtonRaphson state, solve!, powerFlow!                                         */
/* (src/definition/analysis.jl:154-164; acPowerFlow.jl:793-911, 1389-1433)                      */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    i64 n, slack, dim, nnzJ;
    i64 *colptr, *rowval; double *yre, *yim, *ytre, *ytim;   /* Ybus + transpose values (copied) */
    int8_t *type;
    i64 *pq, *pvpq, *pcount, *jcolptr, *jrowval;
    i64 *jcp0, *jri0;                                        /* 0-based copies for the LU */
    double *jnz, *mism, *incr, *vm, *va;
    double *ps, *qs, *pd, *qd;
    jgo_lu *F;
    i64 iteration;
} jgo_nr;

jgo_nr *jgo_nr_create(i64 n, const i64 *colptr, const i64 *rowval, const double *yre, const double *yim,
                      const double *ytre, const double *ytim, const int8_t *type, i64 slack) {
    jgo_nr *h = (jgo_nr *)calloc(1, sizeof(jgo_nr));
    i64 nnz = colptr[n] - 1;
    h->n = n; h->slack = slack;
#define DUP(dst, src, cnt, T) do { h->dst = (T *)malloc((size_t)(cnt) * sizeof(T)); memcpy(h->dst, src, (size_t)(cnt) * sizeof(T)); } while (0)
    DUP(colptr, colptr, n + 1, i64); DUP(rowval, rowval, nnz, i64);
    DUP(yre, yre, nnz, double); DUP(yim, yim, nnz, double); DUP(ytre, ytre, nnz, double); DUP(ytim, ytim, nnz, double);
    DUP(type, type, n, int8_t);
#undef DUP
    h->pq = (i64 *)malloc((size_t)n * sizeof(i64)); h->pvpq = (i64 *)malloc((size_t)n * sizeof(i64)); h->pcount = (i64 *)malloc((size_t)n * sizeof(i64));
    h->jcolptr = (i64 *)malloc(2 * (size_t)n * sizeof(i64) + 8);
    h->nnzJ = jgo_newton_jacobian(n, colptr, rowval, type, slack, h->pq, h->pvpq, h->pcount, &h->dim, h->jcolptr, NULL);
    h->jrowval = (i64 *)malloc((size_t)h->nnzJ * sizeof(i64) + 8);
    jgo_newton_jacobian(n, colptr, rowval, type, slack, h->pq, h->pvpq, h->pcount, &h->dim, h->jcolptr, h->jrowval);
    h->jcp0 = (i64 *)malloc(((size_t)h->dim + 1) * sizeof(i64)); h->jri0 = (i64 *)malloc((size_t)h->nnzJ * sizeof(i64) + 8);
    for (i64 c = 0; c <= h->dim; c++) h->jcp0[c] = h->jcolptr[c] - 1;
    for (i64 p = 0; p < h->nnzJ; p++) h->jri0[p] = h->jrowval[p] - 1;
    h->jnz = (double *)calloc((size_t)h->nnzJ + 1, sizeof(double));
    h->mism = (double *)calloc((size_t)h->dim + 1, sizeof(double)); h->incr = (double *)calloc((size_t)h->dim + 1, sizeof(double));
    h->vm = (double *)calloc((size_t)n, sizeof(double)); h->va = (double *)calloc((size_t)n, sizeof(double));
    h->ps = (double *)calloc((size_t)n, sizeof(double)); h->qs = (double *)calloc((size_t)n, sizeof(double));
    h->pd = (double *)calloc((size_t)n, sizeof(double)); h->qd = (double *)calloc((size_t)n, sizeof(double));
    h->F = jgo_lu_create(h->dim);
    return h;
}

void jgo_nr_destroy(jgo_nr *h) {
    if (!h) return;
    free(h->colptr); free(h->rowval); free(h->yre); free(h->yim); free(h->ytre); free(h->ytim); free(h->type);
    free(h->pq); free(h->pvpq); free(h->pcount); free(h->jcolptr); free(h->jrowval); free(h->jcp0); free(h->jri0);
    free(h->jnz); free(h->mism); free(h->incr); free(h->vm); free(h->va); free(h->ps); free(h->qs); free(h->pd); free(h->qd);
    jgo_lu_destroy(h->F); free(h);
}

i64 jgo_nr_dim(jgo_nr *h) { return h->dim; }
i64 jgo_nr_nnz(jgo_nr *h) { return h->nnzJ; }
i64 jgo_nr_iteration(jgo_nr *h) { return h->iteration; }
i64 jgo_nr_lu_nnz(jgo_nr *h) { return h->F->factored ? jgo_lu_nnz(h->F) : 0; }

void jgo_nr_set_power(jgo_nr *h, const double *ps, const double *qs, const double *pd, const double *qd) {
    size_t s = (size_t)h->n * sizeof(double);
    memcpy(h->ps, ps, s); memcpy(h->qs, qs, s); memcpy(h->pd, pd, s); memcpy(h->qd, qd, s);
}
void jgo_nr_set_voltage(jgo_nr *h, const double *vm, const double *va) {
    memcpy(h->vm, vm, (size_t)h->n * sizeof(double)); memcpy(h->va, va, (size_t)h->n * sizeof(double));
}
void jgo_nr_get_voltage(jgo_nr *h, double *vm, double *va) {
    memcpy(vm, h->vm, (size_t)h->n * sizeof(double)); memcpy(va, h->va, (size_t)h->n * sizeof(double));
}
/* in-place Ybus edit (what acNodalUpdate! does to both value arrays, model.jl:93-101): 0-based pointer */
void jgo_nr_add_ybus(jgo_nr *h, i64 ptr, double dre, double dim_) {
    h->yre[ptr] += dre; h->yim[ptr] += dim_;
    /* the transposed entry */
    i64 c = 0; while (h->colptr[c + 1] - 1 <= ptr) c++;
    i64 r = h->rowval[ptr] - 1;
    for (i64 p = h->colptr[r] - 1; p < h->colptr[r + 1] - 1; p++) if (h->rowval[p] - 1 == c) { h->ytre[p] += dre; h->ytim[p] += dim_; break; }
}
void jgo_nr_get_maps(jgo_nr *h, i64 *pq, i64 *pvpq, i64 *pcount, i64 *jcolptr, i64 *jrowval) {
    memcpy(pq, h->pq, (size_t)h->n * sizeof(i64)); memcpy(pvpq, h->pvpq, (size_t)h->n * sizeof(i64));
    memcpy(pcount, h->pcount, (size_t)h->n * sizeof(i64));
    memcpy(jcolptr, h->jcolptr, ((size_t)h->dim + 1) * sizeof(i64)); memcpy(jrowval, h->jrowval, (size_t)h->nnzJ * sizeof(i64));
}
void jgo_nr_get_vectors(jgo_nr *h, double *jnz, double *mism, double *incr) {
    if (jnz) memcpy(jnz, h->jnz, (size_t)h->nnzJ * sizeof(double));
    if (mism) memcpy(mism, h->mism, (size_t)h->dim * sizeof(double));
    if (incr) memcpy(incr, h->incr, (size_t)h->dim * sizeof(double));
}

void jgo_nr_mismatch(jgo_nr *h, double *stop) {
    jgo_mismatch(h->n, h->colptr, h->rowval, h->ytre, h->ytim, h->type, h->slack, h->pq, h->pvpq,
                 h->vm, h->va, h->ps, h->qs, h->pd, h->qd, h->mism, stop);
}

/* solve!: fill, factor (first) / refactor (later), solve, update (acPowerFlow.jl:793-911) */
int jgo_nr_solve(jgo_nr *h) {
    jgo_jacobian_fill(h->n, h->colptr, h->rowval, h->yre, h->yim, h->ytre, h->ytim, h->type, h->slack,
                      h->pq, h->pvpq, h->pcount, h->jcolptr, h->vm, h->va, h->jnz);
    int rc = h->F->factored ? jgo_lu_refactor(h->F, h->jcp0, h->jri0, h->jnz)      /* :890-895 */
                            : jgo_lu_factor(h->F, h->jcp0, h->jri0, h->jnz);
    if (rc) return rc;
    jgo_lu_solve(h->F, h->mism, h->incr);                                          /* :897 */
    for (i64 i = 0; i < h->n; i++) {                                               /* :899-906 */
        if (h->type[i] == 1) h->vm[i] = h->vm[i] - h->incr[h->pq[i] - 1];
        if (i + 1 != h->slack) h->va[i] = h->va[i] - h->incr[h->pvpq[i] - 1];
    }
    h->iteration++;                                                                /* :908 */
    return 0;
}

/* powerFlow! (acPowerFlow.jl:1389-1433). status: 0 converged, 1 max iterations, 3 singular.
 * history (optional, 2*(maxit+1)) receives (delP, delQ) of every mismatch evaluation. */
int jgo_nr_power_flow(jgo_nr *h, i64 maxit, double tol, double *history, i64 *nhist) {
    h->iteration = 0;                                                              /* :1401 */
    i64 nh = 0; int status = 1;
    for (i64 iter = 0; iter <= maxit; iter++) {                                    /* :1406 */
        double stop[2];
        jgo_nr_mismatch(h, stop);
        if (history) { history[2 * nh] = stop[0]; history[2 * nh + 1] = stop[1]; }
        nh++;
        if (stop[0] < tol && stop[1] < tol) { status = 0; break; }                 /* :1410 */
        if (h->iteration == maxit) { status = 1; break; }                          /* :1414 */
        int rc = jgo_nr_solve(h);
        if (rc) { status = rc; break; }
    }
    if (nhist) *nhist = nh;
    return status;
}
