/* A plain-C client of the drop-in boundary (include/misslap.h): no Python, no torch, only the C ABI.
 * Reads a problem from a binary file written by the test (int64 nnz, int32 maximize, int32 loc[nnz][2],
 * double val[nnz]), solves it through misslap_create / misslap_solve / misslap_destroy, checks the matching
 * with misslap_hopcroft_karp first, solves three more handles of the same problem in lockstep (misslap_solve_batch,
 * meta through a SHORTER caller-side struct), and writes "its nreductions n_assigned obj_f64 ..." + the assignment as text.
 * Build: gcc -std=c11 -I include tests/cabi_client.c -L sslap_amd -lmisslap -Wl,-rpath,$PWD/sslap_amd */
#include <inttypes.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "misslap.h"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t nnz;
    int32_t maximize;
    if (fread(&nnz, sizeof nnz, 1, f) != 1 || fread(&maximize, sizeof maximize, 1, f) != 1) return 2;
    int32_t *loc = malloc(sizeof(int32_t) * 2 * (size_t)nnz);
    double *val = malloc(sizeof(double) * (size_t)nnz);
    if (fread(loc, sizeof(int32_t) * 2, (size_t)nnz, f) != (size_t)nnz || fread(val, sizeof(double), (size_t)nnz, f) != (size_t)nnz)
        return 2;
    fclose(f);
    if (misslap_abi_version() != MISSLAP_ABI_VERSION) return 3;

    int32_t n_rows = 0, n_cols = 0;
    for (int64_t k = 0; k < nnz; ++k) {
        if (loc[2 * k] + 1 > n_rows) n_rows = loc[2 * k] + 1;
        if (loc[2 * k + 1] + 1 > n_cols) n_cols = loc[2 * k + 1] + 1;
    }
    int32_t card = -1;
    if (misslap_hopcroft_karp(loc, nnz, n_rows, n_cols, &card, NULL, NULL) != MISSLAP_OK) {
        fprintf(stderr, "hopcroft_karp: %s\n", misslap_last_error());
        return 4;
    }

    misslap_options opt;
    memset(&opt, 0, sizeof opt);
    opt.struct_size = (int32_t)sizeof opt;
    opt.maximize = maximize;
    opt.max_iter = 100000000;
    opt.tail_threshold = -1; /* library default */
    misslap_solver *h = NULL;
    if (misslap_create(&h, nnz, loc, val, &opt) != MISSLAP_OK) {
        fprintf(stderr, "create: %s\n", misslap_last_error());
        return 5;
    }
    int64_t n = 0, m = 0, e = 0;
    misslap_dims(h, &n, &m, &e);
    int32_t *sol = malloc(sizeof(int32_t) * (size_t)n);
    misslap_meta meta;
    meta.struct_size = (int32_t)sizeof meta; /* ABI 2: the library writes at most this many bytes */
    if (misslap_solve(h, sol, &meta) != MISSLAP_OK) {
        fprintf(stderr, "solve: %s\n", misslap_last_error());
        return 6;
    }
    /* the batched entry point: three more handles of the same problem, solved on shared launches.  Each meta is a
     * caller-side struct of its own size (here: the fields up to solve_ms) -- the library writes struct_size bytes */
    enum { kBatch = 3 };
    struct short_meta {
        misslap_meta m; /* only the first struct_size bytes are the library's */
        char guard[16];
    } bm[kBatch];
    misslap_solver *bh[kBatch] = {0};
    int32_t *bsol[kBatch];
    misslap_meta *bmp[kBatch];
    const int32_t short_size = (int32_t)offsetof(misslap_meta, edges_scanned); /* the smallest the library accepts */
    for (int k = 0; k < kBatch; ++k) {
        if (misslap_create(&bh[k], nnz, loc, val, &opt) != MISSLAP_OK) return 7;
        bsol[k] = malloc(sizeof(int32_t) * (size_t)n);
        memset(&bm[k], 0x5a, sizeof bm[k]);
        bm[k].m.struct_size = short_size;
        bmp[k] = &bm[k].m;
    }
    misslap_batch_info info;
    if (misslap_solve_batch(bh, kBatch, bsol, bmp, 0, &info) != MISSLAP_OK) {
        fprintf(stderr, "solve_batch: %s\n", misslap_last_error());
        return 8;
    }
    int batch_ok = info.groups == 1 && info.launches_issued < info.calls_recorded;
    for (int k = 0; k < kBatch; ++k) {
        batch_ok = batch_ok && memcmp(bsol[k], sol, sizeof(int32_t) * (size_t)n) == 0 && bm[k].m.its == meta.its &&
                   bm[k].m.obj_f64 == meta.obj_f64;
        const unsigned char *raw = (const unsigned char *)&bm[k].m;
        for (size_t q = (size_t)short_size; q < sizeof bm[k].m; ++q) batch_ok = batch_ok && raw[q] == 0x5a; /* untouched */
        misslap_destroy(bh[k]);
        free(bsol[k]);
    }
    FILE *o = fopen(argv[2], "w");
    fprintf(o, "%" PRId64 " %d %" PRId64 " %.17g %d %" PRId64 " %" PRId64 " %d\n", meta.its, meta.nreductions,
            meta.n_assigned, meta.obj_f64, card, n, m, batch_ok);
    for (int64_t i = 0; i < n; ++i) fprintf(o, "%d\n", sol[i]);
    fclose(o);
    misslap_destroy(h);
    free(sol);
    free(loc);
    free(val);
    return 0;
}
