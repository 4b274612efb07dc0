/* gn_from_file.c -- the Gauss-Newton half of the C ABI of libjgrid_hip.so driven from plain C (no Python, no torch): what a
 * ccall binding does for gaussNewton(monitoring) + stateEstimation!(analysis)
 * (src/stateEstimation/acStateEstimation.jl:43-75, 878-904, 1035-1047, 1286-1329), both ways the reference can be driven:
 * the fused loop (jg_gn_run) and the caller's own loop over increment! / solve! (jg_gn_increment, jg_gn_solve).
 *
 *   gcc -O2 -I include bindings/c/gn_from_file.c -L juliagrid.jl_amd -ljgrid_hip -Wl,-rpath,$PWD/juliagrid.jl_amd -o gn_from_file
 *   ./gn_from_file model.bin result.bin [batch]
 *
 * model.bin (little endian): int64 n, nnz, nb, slack, m, n_corr, max_iter; double tol; then
 *   int64 colptr[n+1], rowval[nnz]; double y_reim[2 nnz], yt_reim[2 nnz]            (system.model.ac.nodalMatrix / Transpose)
 *   int64 from[nb], to[nb]; double branch_param[nb][6]
 *   int8 code[m], status[m]; int64 index[m]; int64 corr_row[max(n_corr,1)]
 *   double mean[m], wdiag[m], woff[max(n_corr,1)]; double vm[n], va[n]                 (start point)
 * result.bin: int64 rc, iterations (fused), status, iterations (own loop); double vm[n], va[n] (fused), vm[n], va[n] (own loop),
 *   increment[2n] (last increment of the own loop)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jgrid.h"

static void* rd(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s model.bin result.bin [batch]\n", argv[0]); return 2; }
    const int64_t batch = argc > 3 ? atoll(argv[3]) : 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t hdr[7]; double tol;
    if (fread(hdr, 8, 7, f) != 7 || fread(&tol, 8, 1, f) != 1) { fprintf(stderr, "bad header\n"); return 3; }
    const int64_t n = hdr[0], nnz = hdr[1], nb = hdr[2], slack = hdr[3], m = hdr[4], ncorr = hdr[5], max_iter = hdr[6];
    const int64_t nc1 = ncorr > 0 ? ncorr : 1;
    int64_t* colptr = rd(f, (size_t)(n + 1) * 8);
    int64_t* rowval = rd(f, (size_t)nnz * 8);
    double* y = rd(f, (size_t)nnz * 16);
    double* yt = rd(f, (size_t)nnz * 16);
    int64_t* from = rd(f, (size_t)nb * 8);
    int64_t* to = rd(f, (size_t)nb * 8);
    double* bp = rd(f, (size_t)nb * 48);
    int8_t* code = rd(f, (size_t)m);
    int8_t* status = rd(f, (size_t)m);
    int64_t* index = rd(f, (size_t)m * 8);
    int64_t* corr = rd(f, (size_t)nc1 * 8);
    double* mean = rd(f, (size_t)m * 8);
    double* wdiag = rd(f, (size_t)m * 8);
    double* woff = rd(f, (size_t)nc1 * 8);
    double* vm = rd(f, (size_t)n * 8);
    double* va = rd(f, (size_t)n * 8);
    fclose(f);

    jg_gn* h = NULL;
    int rc = jg_gn_create(&h, n, colptr, rowval, y, yt, nb, from, to, bp, slack, m, code, status, index, ncorr, corr, batch, 0);
    int32_t* iters = calloc((size_t)batch, 4);
    int32_t* st = calloc((size_t)batch, 4);
    double* v1 = calloc((size_t)(batch * n * 2), 8);
    double* v2 = calloc((size_t)(batch * n * 2), 8);
    double* inc = calloc((size_t)(batch * n * 2), 8);
    double* maxinc = calloc((size_t)batch, 8);
    if (!rc) rc = jg_gn_set_measurement(h, mean, wdiag, woff, 0, 0);       /* stride 0: one set of readings for every scenario */
    /* 1. the fused loop */
    if (!rc) rc = jg_gn_set_voltage(h, vm, va, 0);
    if (!rc) rc = jg_gn_run(h, max_iter, tol, iters, st);
    if (!rc) rc = jg_gn_get_voltage(h, v1, v1 + batch * n);
    /* 2. the caller's loop, statement by statement stateEstimation! of the reference (acStateEstimation.jl:1303-1316) */
    int64_t own = 0;
    if (!rc) rc = jg_gn_set_voltage(h, vm, va, 0);
    for (int64_t it = 0; it <= max_iter && !rc; ++it) {
        rc = jg_gn_increment(h, maxinc);
        if (rc) break;
        if (maxinc[0] < tol) break;
        if (own == max_iter) break;
        rc = jg_gn_solve(h);
        ++own;
    }
    if (!rc) rc = jg_gn_get_voltage(h, v2, v2 + batch * n);
    if (!rc) rc = jg_gn_get_increment(h, inc);
    if (rc) fprintf(stderr, "libjgrid_hip: code %d: %s\n", rc, jg_last_error());
    for (int64_t b = 1; b < batch && !rc; ++b)                              /* identical scenarios: identical answers, bit for bit */
        if (iters[b] != iters[0] || memcmp(v1 + b * n, v1, (size_t)n * 8) || memcmp(v1 + (batch + b) * n, v1 + batch * n, (size_t)n * 8)) rc = 100;
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    const int64_t out[4] = {rc, iters[0], st[0], own};
    fwrite(out, 8, 4, o);
    fwrite(v1, 8, (size_t)n, o);
    fwrite(v1 + batch * n, 8, (size_t)n, o);
    fwrite(v2, 8, (size_t)n, o);
    fwrite(v2 + batch * n, 8, (size_t)n, o);
    fwrite(inc, 8, (size_t)(2 * n), o);
    fclose(o);
    if (h) jg_gn_destroy(h);
    printf("rc %d iterations %d (fused) %d (own loop) status %d\n", rc, (int)iters[0], (int)own, (int)st[0]);
    return rc ? 1 : 0;
}
