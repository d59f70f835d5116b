/*
 * tools/sim/auction_sim.c -- ANALYSIS TOOLING (CPU only; neither product nor oracle).
 *
 * A fast sequential model of the reference's Jacobi auction (same rounds, same tie rules as
 * oracle/auction_oracle.c, but with an O(#winners) assignment phase) that carries experiment hooks the
 * checker must not: a per-person candidate-cache model and a histogram of round sizes.  Its own `its` /
 * sha256(sol) are compared with the golden fixture by tools/sim/run_sim.py, so a model that drifts from
 * the reference is noticed.
 *
 * Candidate cache model: a person keeps up to C of its edges (col, cost) plus a bound tau such that every
 * edge NOT in the cache had value (cost - price) <= tau when the cache was built.  Prices only rise, so
 * the bound holds forever.  A later bid is answered from the cache alone ("hit") iff, at current prices,
 * the cached best value v1 > tau and the cached second-best v2 >= tau: then no uncached edge can be the
 * best (ties excluded by the strict test) and none can exceed the second best.
 *   policy 0: cache = the C largest values, tau = the (C+1)-th largest (exact top-C)
 *   policy 1: cache = all edges with value >= t for a threshold t found by a bounded bisection between the
 *             row's minimum and second-best value so that the count lands in [C/2, C]; tau = t
 *             (what a wavefront can build with a handful of ballots)
 *
 * Bidder trace: with SIM_TRACE=<file> and SIM_TRACE_THR=<K> in the environment the persons bidding in rounds with
 * at most K bidders are written as int32, rounds separated by -1 (tools/tail_reuse.py reads it).
 *
 * On-chip working set of the small rounds: with SIM_LDS=<K> every bid of a round with at most K bidders looks its
 * person's line up in direct-mapped caches of 128 .. 1024 lines (index = person mod size) and each candidate's price
 * record in direct-mapped caches of 1024 .. 16384 records (index = object mod size); the hit rates are printed per
 * mode ("lds_sim").  The caches are write-through in the design they stand for, so a hit is a tag match.
 *
 * In-wavefront speculation gate (round 5): with SIM_SPEC=1 every person remembers the RUNNER-UP object of its last bid.
 * In the rounds with K <= 2 (the chain / duo rounds of the tail kernel) the model asks, per bidder: is the object it bids
 * on now the remembered runner-up (then the next bidder -- the owner of that object -- could have been fetched and
 * evaluated one round ahead), does that object have an owner at all (the chain goes on), and is the object among the
 * three best of the owner's row at the OLD prices (then the owner's top-2 after the price update follow from its top-3
 * and the one patched value).  Printed to stderr.
 *
 * usage: auction_sim <input.bin> C policy build_thr use_thr
 *   input.bin: int64 nnz, int32 maximize, int32 loc[nnz][2], double val[nnz]
 *   build_thr: caches are (re)built on a miss in rounds with K <= build_thr
 *   use_thr:   hits / misses are counted in rounds with K <= use_thr
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int cmp_desc(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return x < y ? 1 : x > y ? -1 : 0;
}

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s input.bin C policy build_thr use_thr\n", argv[0]);
        return 2;
    }
    const int C = atoi(argv[2]), policy = atoi(argv[3]), build_thr = atoi(argv[4]), use_thr = atoi(argv[5]);
    const int cmin = getenv("SIM_CMIN") ? atoi(getenv("SIM_CMIN")) : (C + 1) / 2;
    /* SIM_REFRESH=A: a hit that leaves fewer than A cached candidates at or above tau rebuilds the cache right away
     * ("background" rebuild by an idle wavefront: counted separately, the bid itself stays a hit) */
    const int refresh = getenv("SIM_REFRESH") ? atoi(getenv("SIM_REFRESH")) : 0;
    int64_t bg_builds = 0;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t nnz;
    int32_t maximize;
    if (fread(&nnz, 8, 1, f) != 1 || fread(&maximize, 4, 1, f) != 1) return 2;
    int32_t *loc = malloc(sizeof(int32_t) * 2 * nnz);
    double *val = malloc(sizeof(double) * nnz);
    if (fread(loc, 8, nnz, f) != (size_t)nnz || fread(val, 8, nnz, f) != (size_t)nnz) return 2;
    fclose(f);
    int N = loc[2 * (nnz - 1)] + 1, M = 0;
    for (int64_t k = 0; k < nnz; ++k)
        if (loc[2 * k + 1] + 1 > M) M = loc[2 * k + 1] + 1;
    int *row_ptr = calloc(N + 1, sizeof(int)), *col = malloc(sizeof(int) * nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        row_ptr[loc[2 * k] + 1]++;
        col[k] = loc[2 * k + 1];
        if (!maximize) val[k] = -val[k];
    }
    for (int i = 0; i < N; ++i) row_ptr[i + 1] += row_ptr[i];
    double maxabs = 0;
    for (int64_t k = 0; k < nnz; ++k)
        if (fabs(val[k]) > maxabs) maxabs = fabs(val[k]);
    float eps = (float)((double)(float)maxabs / 2.0), target = (float)(1.0 / (double)N), theta = (float)0.15;
    double *p = calloc(M, sizeof(double));
    int *p2o = malloc(sizeof(int) * N), *o2p = malloc(sizeof(int) * M), *U = malloc(sizeof(int) * N);
    int *pos = malloc(sizeof(int) * N);
    for (int i = 0; i < N; ++i) p2o[i] = -1, U[i] = i, pos[i] = i;
    for (int j = 0; j < M; ++j) o2p[j] = -1;
    double *best_bid = malloc(sizeof(double) * M);
    int *best_n = malloc(sizeof(int) * M);
    for (int j = 0; j < M; ++j) best_bid[j] = -1.0, best_n[j] = -1;
    int *bobj = malloc(sizeof(int) * N), *touched = malloc(sizeof(int) * N);
    double *bbid = malloc(sizeof(double) * N);
    /* cache */
    int *c_col = malloc(sizeof(int) * (size_t)N * C);
    double *c_cost = malloc(sizeof(double) * (size_t)N * C), *c_tau = malloc(sizeof(double) * N);
    unsigned char *c_n = calloc(N, 1), *c_valid = calloc(N, 1);
    double *tmpv = malloc(sizeof(double) * (size_t)(M < 1 << 20 ? 1 << 20 : M));
    /* stats by mode: 0 chain (K=1), 1 pair, 2 team (3..16), 3 block (17..64), 4 (65..512), 5 (513..2048), 6 bigger */
    int64_t rounds[7] = {0}, bids[7] = {0}, hits[7] = {0}, allhit[7] = {0}, bad = 0, builds = 0;
    int64_t phase_rounds[7] = {0};
    int64_t khist[33] = {0};  /* rounds with exactly K bidders, K <= 32 */
    int64_t its = 0;
    int K = N, nred = 0;
    /* direct-mapped on-chip caches of the small rounds (see the header) */
    const int lds_thr = getenv("SIM_LDS") ? atoi(getenv("SIM_LDS")) : 0;
    enum { NL = 4, NR = 5 };
    const int lsz[NL] = {128, 256, 512, 1024}, rsz[NR] = {1024, 2048, 4096, 8192, 16384};
    int *ltag[NL], *rtag[NR];
    for (int a = 0; a < NL; ++a) ltag[a] = malloc(sizeof(int) * lsz[a]), memset(ltag[a], 0xff, sizeof(int) * lsz[a]);
    for (int a = 0; a < NR; ++a) rtag[a] = malloc(sizeof(int) * rsz[a]), memset(rtag[a], 0xff, sizeof(int) * rsz[a]);
    int64_t lds_bids[7] = {0}, lds_lhit[7][NL] = {{0}}, lds_recs[7] = {0}, lds_rhit[7][NR] = {{0}}, lds_allrec[7][NR] = {{0}};
    const int f32_mode = getenv("SIM_F32") ? atoi(getenv("SIM_F32")) : 0;
    const int f32_min_K = (int)(0.3 * N);
    int64_t f32_hist[17] = {0}, f32_bids = 0, f32_round_over4 = 0;
    const int spec_mode = getenv("SIM_SPEC") ? atoi(getenv("SIM_SPEC")) : 0;
    int *ru_obj = malloc(sizeof(int) * N);
    for (int i = 0; i < N; ++i) ru_obj[i] = -1;
    /* [K - 1][.]: bids, with a remembered runner-up, prediction right, ... and the object has an owner (chain goes on),
     * ... and the object is in the owner's top-3 at the old prices; rounds, rounds with EVERY bidder fully predicted */
    int64_t sp[2][5] = {{0}}, sp_rounds[2] = {0}, sp_rounds_ok[2] = {0};
    FILE *trace = getenv("SIM_TRACE") ? fopen(getenv("SIM_TRACE"), "wb") : NULL;
    const int trace_thr = getenv("SIM_TRACE_THR") ? atoi(getenv("SIM_TRACE_THR")) : 256;
    for (;;) {
        if (trace && K <= trace_thr) {
            const int sep = -1;
            fwrite(U, sizeof(int), K, trace);
            fwrite(&sep, sizeof(int), 1, trace);
        }
        const int mode = K == 1 ? 0 : K == 2 ? 1 : K <= 16 ? 2 : K <= 64 ? 3 : K <= 512 ? 4 : K <= 2048 ? 5 : 6;
        const int use = K <= use_thr, build = K <= build_thr;
        int round_hits = 0, round_full = 0;
        for (int n = 0; n < K; ++n) {
            const int i = U[n], s = row_ptr[i], e = row_ptr[i + 1];
            double vbest = -INFINITY, wi = -INFINITY, costbest = 0;
            int jbest = 0, jsecond = -1;
            for (int g = s; g < e; ++g) {
                const double v = val[g] - p[col[g]];
                if (v >= vbest || g == s) jsecond = jbest, jbest = col[g], wi = vbest, vbest = v, costbest = val[g];
                else if (v > wi) wi = v, jsecond = col[g];
            }
            if (spec_mode && K <= 2) {
                int64_t *c = sp[K - 1];
                c[0]++;
                int full = 0;
                if (ru_obj[i] >= 0) {
                    c[1]++;
                    if (ru_obj[i] == jbest) {
                        c[2]++;
                        const int nb = o2p[jbest];
                        if (nb >= 0) {
                            c[3]++;
                            /* rank of jbest in the owner's row at the old prices (ties count against it) */
                            double vj = -INFINITY;
                            for (int g = row_ptr[nb]; g < row_ptr[nb + 1]; ++g)
                                if (col[g] == jbest) vj = val[g] - p[jbest];
                            int above = 0;
                            for (int g = row_ptr[nb]; g < row_ptr[nb + 1]; ++g) above += (val[g] - p[col[g]]) > vj;
                            if (above <= 2) c[4]++, full = 1;
                        }
                    }
                }
                round_full += full;
            }
            if (spec_mode) ru_obj[i] = jsecond;
            if (f32_mode && K >= f32_min_K) { /* how many edges a single-precision filter could not tell from the top two */
                float b32 = -INFINITY, w32 = -INFINITY;
                double pmax = 0, vmax = 0;
                for (int g = s; g < e; ++g) {
                    const float v = (float)val[g] - (float)p[col[g]];
                    if (v >= b32) w32 = b32, b32 = v;
                    else if (v > w32) w32 = v;
                    if (fabs(p[col[g]]) > pmax) pmax = fabs(p[col[g]]);
                    if (fabs(val[g]) > vmax) vmax = fabs(val[g]);
                }
                const double delta = (pmax + 2.0 * (pmax + vmax)) * 0x1p-24; /* rounding of the price + of the difference, generous */
                int cnt = 0;
                for (int g = s; g < e; ++g) cnt += (double)((float)val[g] - (float)p[col[g]]) >= (double)w32 - 2.0 * delta;
                f32_hist[cnt < 2 ? 2 : cnt > 16 ? 16 : cnt]++;
                f32_bids++;
                if (cnt > 4) f32_round_over4++;
            }
            if (C > 0 && (use || build)) {
                int hit = 0, alive = 0;
                if (c_valid[i]) {
                    double vc = -INFINITY, wc = -INFINITY;
                    for (int k = 0; k < c_n[i]; ++k) {
                        const double v = c_cost[(size_t)i * C + k] - p[c_col[(size_t)i * C + k]];
                        alive += v >= c_tau[i];
                        if (v >= vc) wc = vc, vc = v;
                        else if (v > wc) wc = v;
                    }
                    if (vc > c_tau[i] && wc >= c_tau[i]) {
                        hit = 1;
                        if (vc != vbest || wc != wi) bad++;
                    }
                }
                if (use) {
                    bids[mode]++;
                    hits[mode] += hit;
                    round_hits += hit;
                }
                if (lds_thr && K <= lds_thr && c_valid[i]) {
                    lds_bids[mode]++;
                    for (int a = 0; a < NL; ++a) {
                        lds_lhit[mode][a] += ltag[a][i % lsz[a]] == i;
                        ltag[a][i % lsz[a]] = i;
                    }
                    lds_recs[mode] += c_n[i];
                    for (int a = 0; a < NR; ++a) {
                        int all = 1;
                        for (int k = 0; k < c_n[i]; ++k) {
                            const int cj = c_col[(size_t)i * C + k];
                            const int h1 = rtag[a][cj % rsz[a]] == cj;
                            lds_rhit[mode][a] += h1;
                            all &= h1;
                            rtag[a][cj % rsz[a]] = cj;
                        }
                        lds_allrec[mode][a] += all;
                    }
                }
                const int bg = hit && build && alive < refresh;
                bg_builds += bg;
                if ((!hit || bg) && build) {
                    builds += !bg;
                    const int len = e - s;
                    for (int g = s; g < e; ++g) tmpv[g - s] = val[g] - p[col[g]];
                    double t;
                    if (policy == 0) {
                        qsort(tmpv, len, sizeof(double), cmp_desc);
                        t = len > C ? tmpv[C] : -INFINITY; /* bound = (C+1)-th value; cache = values > bound ... */
                        /* ties with the bound are left out of the cache (they are <= tau) */
                    } else {
                        /* bisection on the threshold: count(v >= t) in [C/2, C] */
                        double lo = INFINITY, hi = wi; /* count(v >= hi) >= 2 */
                        for (int k = 0; k < len; ++k)
                            if (tmpv[k] < lo) lo = tmpv[k];
                        t = hi;
                        int cnt_hi = 0;
                        for (int k = 0; k < len; ++k) cnt_hi += tmpv[k] >= hi;
                        if (cnt_hi > C || !(lo < hi)) t = INFINITY; /* too many ties at the top: no cache */
                        else if (len <= C) t = -INFINITY;
                        else {
                            for (int it = 0; it < 12; ++it) {
                                const double mid = 0.5 * (lo + hi);
                                int cnt = 0;
                                for (int k = 0; k < len; ++k) cnt += tmpv[k] >= mid;
                                if (cnt > C) lo = mid;
                                else {
                                    hi = mid;
                                    t = mid;
                                    if (cnt >= cmin) break;
                                }
                            }
                        }
                    }
                    int cn = 0;
                    if (t != INFINITY) {
                        for (int g = s; g < e && cn < C; ++g) {
                            const double v = val[g] - p[col[g]];
                            if (policy == 0 ? v > t : v >= t) c_col[(size_t)i * C + cn] = col[g], c_cost[(size_t)i * C + cn] = val[g], cn++;
                        }
                        c_n[i] = (unsigned char)cn;
                        c_tau[i] = t;
                        c_valid[i] = 1;
                    } else
                        c_valid[i] = 0;
                }
            }
            bobj[n] = jbest;
            bbid[n] = (costbest - wi) + (double)eps;
        }
        if (use) {
            rounds[mode]++;
            allhit[mode] += round_hits == K;
        }
        if (spec_mode && K <= 2) sp_rounds[K - 1]++, sp_rounds_ok[K - 1] += round_full == K;
        phase_rounds[mode]++;
        if (K <= 32) khist[K]++;
        /* resolve */
        int nt = 0;
        for (int n = 0; n < K; ++n) {
            const int j = bobj[n];
            if (bbid[n] > best_bid[j]) {
                if (best_n[j] == -1) touched[nt++] = j;
                best_bid[j] = bbid[n];
                best_n[j] = n;
            }
        }
        /* assign */
        int Kn = K;
        for (int k = 0; k < nt; ++k) {
            const int j = touched[k], n = best_n[j], i = U[n];
            p[j] = best_bid[j];
            const int prev = o2p[j];
            if (prev != -1) p2o[prev] = -1, U[n] = prev;
            else U[n] = -1, Kn--;
            p2o[i] = j;
            o2p[j] = i;
            best_bid[j] = -1.0;
            best_n[j] = -1;
        }
        /* push_all_left */
        {
            int r = Kn;
            for (int l = 0; l < Kn; ++l)
                if (U[l] == -1) {
                    while (U[r] == -1) r++;
                    U[l] = U[r];
                    U[r] = -1;
                }
        }
        K = Kn;
        its++;
        if (K == 0) {
            /* eCE at target */
            int ok = 1;
            for (int i = 0; i < N && ok; ++i) {
                const int j = p2o[i];
                double cc = 0;
                for (int g = row_ptr[i]; g < row_ptr[i + 1]; ++g)
                    if (col[g] == j) cc = val[g];
                const double lhs = cc - p[j] + 1e-7;
                for (int g = row_ptr[i]; g < row_ptr[i + 1]; ++g)
                    if (lhs < (val[g] - p[col[g]]) - (double)target) {
                        ok = 0;
                        break;
                    }
            }
            if (ok || eps < target) break;
            eps = eps * theta;
            for (int i = 0; i < N; ++i) p2o[i] = -1, U[i] = i;
            for (int j = 0; j < M; ++j) o2p[j] = -1;
            K = N;
            nred++;
        }
    }
    if (trace) fclose(trace);
    /* FNV of sol for a cheap cross-check, plus its */
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < N; ++i) h = (h ^ (uint64_t)(uint32_t)p2o[i]) * 1099511628211ull;
    printf("{\"its\": %lld, \"nreductions\": %d, \"sol_fnv\": \"%016llx\", \"C\": %d, \"policy\": %d, \"build_thr\": %d, "
           "\"use_thr\": %d, \"inconsistent\": %lld, \"builds\": %lld, \"bg_builds\": %lld,\n \"modes\": [",
           (long long)its, nred, (unsigned long long)h, C, policy, build_thr, use_thr, (long long)bad, (long long)builds,
           (long long)bg_builds);
    const char *names[7] = {"K=1", "K=2", "K=3..16", "K=17..64", "K=65..512", "K=513..2048", "K>2048"};
    for (int m = 0; m < 7; ++m)
        printf("%s{\"mode\": \"%s\", \"rounds_all\": %lld, \"rounds\": %lld, \"bids\": %lld, \"hit_rate\": %.4f, "
               "\"all_hit_rounds\": %.4f}",
               m ? ",\n  " : "", names[m], (long long)phase_rounds[m], (long long)rounds[m], (long long)bids[m],
               bids[m] ? (double)hits[m] / bids[m] : 0.0, rounds[m] ? (double)allhit[m] / rounds[m] : 0.0);
    if (lds_thr) {
        printf("],\n \"lds_sim\": [");
        for (int m = 0; m < 7; ++m) {
            if (!lds_bids[m]) continue;
            printf("%s{\"mode\": \"%s\", \"bids\": %lld, \"line_hit\": {", m ? ",\n  " : "", names[m], (long long)lds_bids[m]);
            for (int a = 0; a < NL; ++a) printf("%s\"%d\": %.4f", a ? ", " : "", lsz[a], (double)lds_lhit[m][a] / lds_bids[m]);
            printf("}, \"record_hit\": {");
            for (int a = 0; a < NR; ++a) printf("%s\"%d\": %.4f", a ? ", " : "", rsz[a], lds_recs[m] ? (double)lds_rhit[m][a] / lds_recs[m] : 0.0);
            printf("}, \"all_records_of_a_bid_hit\": {");
            for (int a = 0; a < NR; ++a) printf("%s\"%d\": %.4f", a ? ", " : "", rsz[a], (double)lds_allrec[m][a] / lds_bids[m]);
            printf("}}");
        }
    }
    if (f32_mode) {
        fprintf(stderr, "f32 filter, bids of rounds with K >= 0.3 N: %lld; edges within the margin of the second-best (2, 3, ..., 16+): ", (long long)f32_bids);
        for (int k = 2; k <= 16; ++k) fprintf(stderr, "%.5f ", (double)f32_hist[k] / (double)(f32_bids ? f32_bids : 1));
        fprintf(stderr, "; more than 4: %.5f\n", (double)f32_round_over4 / (double)(f32_bids ? f32_bids : 1));
    }
    if (spec_mode)
        for (int k = 0; k < 2; ++k)
            fprintf(stderr, "speculation gate, rounds with K = %d: %lld rounds, %lld bids; runner-up remembered %.4f; the bid goes to the "
                    "remembered runner-up %.4f; ... and that object has an owner (the chain goes on) %.4f; ... and it is among "
                    "the owner's three best at the old prices %.4f; rounds in which EVERY bidder is fully predicted %.4f\n",
                    k + 1, (long long)sp_rounds[k], (long long)sp[k][0], (double)sp[k][1] / (double)(sp[k][0] ? sp[k][0] : 1),
                    (double)sp[k][2] / (double)(sp[k][0] ? sp[k][0] : 1), (double)sp[k][3] / (double)(sp[k][0] ? sp[k][0] : 1),
                    (double)sp[k][4] / (double)(sp[k][0] ? sp[k][0] : 1),
                    (double)sp_rounds_ok[k] / (double)(sp_rounds[k] ? sp_rounds[k] : 1));
    printf("],\n \"rounds_by_K\": [");
    for (int k = 1; k <= 32; ++k) printf("%s%lld", k > 1 ? ", " : "", (long long)khist[k]);
    printf("]}\n");
    /* sol to a file for the sha256 check */
    if (argc > 6) {
        FILE *o = fopen(argv[6], "wb");
        fwrite(p2o, sizeof(int), N, o);
        fclose(o);
    }
    return 0;
}
