/* nr_from_file.c -- the C ABI of libjgrid_hip.so driven from plain C (no Python, no torch): what a cgo / ccall / JNI binding
 * of the reference would do for newtonRaphson(system) + powerFlow!(analysis) (src/powerFlow/acPowerFlow.jl:39-87, 1389-1433).
 *
 *   gcc -O2 -I include bindings/c/nr_from_file.c -L juliagrid.jl_amd -ljgrid_hip -Wl,-rpath,$PWD/juliagrid.jl_amd -o nr_from_file
 *   ./nr_from_file model.bin result.bin [batch]
 *
 * model.bin (little endian): int64 n, nnz, slack, max_iter; double tol; then
 *   int64 colptr[n+1], rowval[nnz] (1-based, system.model.ac.nodalMatrix); double y_reim[2 nnz], yt_reim[2 nnz];
 *   int8 type[n]; double p_inj[n], q_inj[n], vm[n], va[n]
 * result.bin: int64 rc, iterations, status; double vm[n], va[n]   (scenario 0; every scenario of a batch is the same problem)
 */
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
    int64_t hdr[4]; double tol;
    if (fread(hdr, 8, 4, f) != 4 || fread(&tol, 8, 1, f) != 1) { fprintf(stderr, "bad header\n"); return 3; }
    const int64_t n = hdr[0], nnz = hdr[1], slack = hdr[2], max_iter = hdr[3];
    int64_t* colptr = rd(f, (size_t)(n + 1) * 8);
    int64_t* rowval = rd(f, (size_t)nnz * 8);
    double* y = rd(f, (size_t)nnz * 16);
    double* yt = rd(f, (size_t)nnz * 16);
    int8_t* type = rd(f, (size_t)n);
    double* p = rd(f, (size_t)n * 8);
    double* q = rd(f, (size_t)n * 8);
    double* vm = rd(f, (size_t)n * 8);
    double* va = rd(f, (size_t)n * 8);
    fclose(f);

    jg_nr* h = NULL;
    int rc = jg_nr_create(&h, n, colptr, rowval, y, yt, type, slack, batch, 0, 0);
    int32_t* iters = calloc((size_t)batch, 4);
    int32_t* status = calloc((size_t)batch, 4);
    double* ovm = calloc((size_t)(batch * n), 8);
    double* ova = calloc((size_t)(batch * n), 8);
    if (!rc) rc = jg_nr_set_injection(h, p, q, 0);           /* stride 0: one vector for every scenario */
    if (!rc) rc = jg_nr_set_voltage(h, vm, va, 0);
    if (!rc) rc = jg_nr_run(h, max_iter, tol, iters, status);
    if (!rc) rc = jg_nr_get_voltage(h, ovm, ova);
    if (rc) fprintf(stderr, "libjgrid_hip: code %d: %s\n", rc, jg_last_error());
    for (int64_t b = 1; b < batch && !rc; ++b)               /* identical scenarios must give identical answers, bit for bit */
        if (iters[b] != iters[0] || memcmp(ovm + b * n, ovm, (size_t)n * 8) || memcmp(ova + b * n, ova, (size_t)n * 8)) rc = 100;
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    const int64_t out[3] = {rc, iters[0], status[0]};
    fwrite(out, 8, 3, o);
    fwrite(ovm, 8, (size_t)n, o);
    fwrite(ova, 8, (size_t)n, o);
    fclose(o);
    if (h) jg_nr_destroy(h);
    printf("rc %d iterations %d status %d\n", rc, (int)iters[0], (int)status[0]);
    return rc ? 1 : 0;
}
