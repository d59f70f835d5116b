/*
 * oracle/auction_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A sequential, single-threaded plain-C restatement of the reference's
 * epsilon-scaling Jacobi auction (OllieBoyne/sslap v0.2.5, sslap/auction_.pyx).
 * It exists only as the parity checker for the HIP path and as the "port"
 * CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library
 * (sslap_amd/csrc) never links or calls it.
 *
 * Parity pin: this restatement is checked bit-for-bit (sol, its, nreductions,
 * prices, objective) against outputs of the real reference (Cython 3.2.9 build
 * of /root/reference, run in the build container) by tests/golden/make_golden.py;
 * the resulting vectors are committed under tests/golden/ and re-checked by
 * tests/test_oracle_golden.py everywhere.
 *
 * Every function cites the reference lines it follows.  Semantics that matter
 * (SURVEY.md section 5 quirk 8): prices, values and bids are double; eps,
 * target_eps and theta are float; `1/N` is true division (Cython 3).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORACLE_API __attribute__((visibility("default")))

typedef struct oracle_solver {
    /* auction_.pyx:167-200 (cdef class AuctionSolver fields) */
    int64_t num_rows, num_cols, nnz;
    double *p;
    int *i_starts_stops; /* row_ptr, N+1 */
    int *j_counts;       /* N */
    int *flat_j;         /* nnz */
    double *val;         /* borrowed from the caller, sign-flipped in place for 'min' (:237) */
    int *person_to_object, *object_to_person;
    float eps, target_eps, theta;
    int maximize;
    int nits, nreductions;
    int64_t max_iter;
    double *best_bids;
    int *best_bidders;
    int num_unassigned;
    int *unassigned_people, *person_to_assignment_idx;
    float start_eps;
    /* instrumentation (not in the reference) */
    uint64_t edges_scanned, bids_made;
    double t_bid, t_total;
    int time_phases;
    int assign_by_bidders; /* 0 (default): the reference's O(M) walk; 1: the same assignments, O(#bids) -- see bid_and_assign */
} oracle_solver;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* auction_.pyx:66-72 */
static int *fill_int(size_t n, int v) {
    int *out = (int *)malloc((n ? n : 1) * sizeof(int));
    for (size_t i = 0; i < n; ++i) out[i] = v;
    return out;
}
/* auction_.pyx:77-82 */
static double *fill_float(size_t n, double v) {
    double *out = (double *)malloc((n ? n : 1) * sizeof(double));
    for (size_t i = 0; i < n; ++i) out[i] = v;
    return out;
}
/* auction_.pyx:101-107 */
static int *arange(size_t n) {
    int *out = (int *)malloc((n ? n : 1) * sizeof(int));
    for (size_t i = 0; i < n; ++i) out[i] = (int)i;
    return out;
}

/* auction_.pyx:33-48 cumulative_idxs: row id advances by ONE whenever the row
 * value exceeds the running id (rows must be sorted, gap-free). */
static int *cumulative_idxs(const int *loc, size_t nnz, size_t N) {
    int *out = (int *)malloc((N + 2) * sizeof(int));
    int value = -1;
    size_t i = 0;
    for (i = 0; i < nnz; ++i) {
        if (loc[2 * i] > value) {
            value += 1;
            out[value] = (int)i;
        }
    }
    out[value + 1] = (int)nnz; /* ':47' writes i+1 with i = last index */
    return out;
}

/* auction_.pyx:137-162 push_all_left */
static void push_all_left(int *data, int *mapper, int num_ints, size_t size) {
    if (num_ints == 0) return;
    int left_track = 0, right_track = num_ints;
    while (left_track < num_ints) {
        if (data[left_track] == -1) {
            while (data[right_track] == -1 && (size_t)right_track < size) right_track += 1;
            int i = data[right_track];
            data[left_track] = i;
            data[right_track] = -1;
            mapper[i] = left_track;
        }
        left_track += 1;
    }
}

/* auction_.pyx:202-265 AuctionSolver.__init__.
 * loc: int32[nnz][2] row-sorted; val: double[nnz], MUTATED for minimise (:237).
 * eps_start: the float the adapters pass (:571 / :617). */
ORACLE_API oracle_solver *oracle_create(int64_t nnz, const int32_t *loc, double *val, int maximize,
                                        float eps_start, int64_t max_iter) {
    oracle_solver *s = (oracle_solver *)calloc(1, sizeof(oracle_solver));
    int rmax = -1, cmax = -1;
    for (int64_t i = 0; i < nnz; ++i) { /* :209-210 */
        if (loc[2 * i] > rmax) rmax = loc[2 * i];
        if (loc[2 * i + 1] > cmax) cmax = loc[2 * i + 1];
    }
    size_t N = (size_t)rmax + 1, M = (size_t)cmax + 1;
    s->num_rows = (int64_t)N;
    s->num_cols = (int64_t)M;
    s->nnz = nnz;
    s->maximize = maximize;
    s->nits = 0;
    s->nreductions = 0;
    s->max_iter = max_iter;
    s->p = fill_float(M, 0.0);                          /* :220 */
    s->i_starts_stops = cumulative_idxs(loc, (size_t)nnz, N); /* :223 */
    s->j_counts = (int *)malloc((N ? N : 1) * sizeof(int));   /* :226 diff */
    for (size_t i = 0; i < N; ++i) s->j_counts[i] = s->i_starts_stops[i + 1] - s->i_starts_stops[i];
    s->flat_j = (int *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int)); /* :229 */
    for (int64_t i = 0; i < nnz; ++i) s->flat_j[i] = loc[2 * i + 1];
    s->person_to_object = fill_int(N, -1); /* :231 */
    s->object_to_person = fill_int(M, -1); /* :232 */
    if (!maximize) /* :236-237 mult_ndarray_by(val, -1) */
        for (int64_t i = 0; i < nnz; ++i) val[i] = val[i] * -1;
    s->val = val;
    /* :242-243 C = max |a_ij| held in a C float; max_val :123-134 */
    double maxval = -INFINITY;
    for (int64_t i = 0; i < nnz; ++i) {
        double v = fabs(val[i]);
        if (v > maxval) maxval = v;
    }
    float C = (float)maxval;
    s->eps = (float)((double)C / 2.0);          /* :246 (true division) */
    s->target_eps = (float)(1.0 / (double)N);   /* :247 */
    s->theta = (float)0.15;                     /* :248 */
    if (eps_start > 0) s->eps = eps_start;      /* :251-252 */
    s->best_bids = fill_float(M, -1.0);         /* :255 */
    s->best_bidders = fill_int(M, -1);          /* :256 */
    s->num_unassigned = (int)N;                 /* :259 */
    s->unassigned_people = arange(N);           /* :260 */
    s->person_to_assignment_idx = arange(N);    /* :261 */
    s->start_eps = s->eps;                      /* :264 meta['start_eps'] (rounded in Python) */
    return s;
}

/* auction_.pyx:313-430 bid_and_assign */
static void bid_and_assign(oracle_solver *s) {
    size_t N = (size_t)s->num_cols; /* sic :318 */
    size_t num_bidders = (size_t)s->num_unassigned;
    int *unassigned_people = s->unassigned_people;
    int *person_to_assignment_idx = s->person_to_assignment_idx;
    /* :323-325 fresh scratch every round (the reference never frees it; we do) */
    int *bidders = fill_int(num_bidders, -1);
    int *objects_bidded = fill_int(num_bidders, -1);
    double *bids = fill_float(num_bidders, -1.0);
    double *p = s->p;
    const int *j_counts = s->j_counts, *i_starts_stops = s->i_starts_stops, *flat_j = s->flat_j;
    const double *val = s->val;
    int *person_to_object = s->person_to_object, *object_to_person = s->object_to_person;
    const double eps = (double)s->eps; /* float promoted in ':360' */

    double t0 = s->time_phases ? now_s() : 0.0;
    /* BIDDING PHASE :339-365 */
    for (size_t nbidder = 0; nbidder < num_bidders; ++nbidder) {
        int i = unassigned_people[nbidder];
        size_t num_objects = (size_t)j_counts[i];
        size_t start = (size_t)i_starts_stops[i];
        double vbest = -INFINITY, wi = -INFINITY, costbest = 0.0;
        int jbest = 0;
        for (size_t idx = 0; idx < num_objects; ++idx) {
            size_t glob_idx = start + idx;
            int j = flat_j[glob_idx];
            double cost = val[glob_idx];
            double vi = cost - p[j];
            if (vi >= vbest || idx == 0) { /* :351 ">=": the LAST maximum wins */
                jbest = j;
                wi = vbest;
                vbest = vi;
                costbest = cost;
            } else if (vi > wi) {
                wi = vi;
            }
        }
        double bbest = costbest - wi + eps; /* :360 (left-to-right) */
        bidders[nbidder] = i;
        bids[nbidder] = bbest;
        objects_bidded[nbidder] = jbest;
        s->edges_scanned += num_objects;
    }
    s->bids_made += num_bidders;
    if (s->time_phases) s->t_bid += now_s() - t0;

    /* RESOLVE :367-385 (strict '>' => earliest bidder in list order wins ties) */
    double *best_bids = s->best_bids;
    int *best_bidders = s->best_bidders;
    size_t num_successful_bids = 0;
    for (size_t n = 0; n < num_bidders; ++n) {
        int i = bidders[n];
        double bid_val = bids[n];
        size_t jbid = (size_t)objects_bidded[n];
        if (bid_val > best_bids[jbid]) {
            if (best_bidders[jbid] == -1) num_successful_bids += 1;
            best_bids[jbid] = bid_val;
            best_bidders[jbid] = i;
        }
    }

    /* ASSIGNMENT PHASE :388-427: walk over ALL objects with early break.
     * assign_by_bidders (bench.py's "optimised" CPU figure; never the parity oracle): the same
     * loop body for the objects that received a bid, found through the round's bidders instead of a walk over all M
     * objects.  Every write below touches only the winner i, its object j and j's previous owner, all distinct between
     * winners, so the visiting order does not change the result (tests/test_oracle_golden.py checks every fixture). */
    size_t people_to_unassign_ctr = 0, people_to_assign_ctr = 0;
    int bid_ctr = 0;
    const int by_bidders = s->assign_by_bidders;
    const int walk_n = by_bidders ? (int)num_bidders : (int)s->num_cols;
    for (int w = 0; w < walk_n; ++w) {
        const int j = by_bidders ? objects_bidded[w] : w;
        int i = best_bidders[j];
        if (i != -1) {
            p[j] = best_bids[j];
            int assignment_idx = person_to_assignment_idx[i];
            int prev_i = object_to_person[j];
            if (prev_i != -1) {
                people_to_unassign_ctr += 1;
                person_to_object[prev_i] = -1;
                person_to_assignment_idx[i] = -1;
                person_to_assignment_idx[prev_i] = assignment_idx;
                unassigned_people[assignment_idx] = prev_i;
            } else {
                unassigned_people[assignment_idx] = -1;
                person_to_assignment_idx[i] = -1;
            }
            people_to_assign_ctr += 1;
            person_to_object[i] = j;
            object_to_person[j] = i;
            best_bidders[j] = -1;
            best_bids[j] = -1;
            bid_ctr += 1;
            if ((size_t)bid_ctr >= num_successful_bids) break;
        }
    }
    s->num_unassigned += (int)people_to_unassign_ctr - (int)people_to_assign_ctr; /* :429 */
    push_all_left(unassigned_people, person_to_assignment_idx, s->num_unassigned, N); /* :430 */
    free(bidders);
    free(objects_bidded);
    free(bids);
}

/* auction_.pyx:443-485 eCE_satisfied(eps) with tol = 1e-7 (:16), on plain arrays: the loop of the reference for ANY
 * state with everybody assigned (tests put the GPU's eCE pass next to it at every end of an eps-phase) */
ORACLE_API int oracle_ece_arrays(int64_t num_rows, const int *i_starts_stops, const int *flat_j, const double *val,
                                 const double *p, const int *person_to_object, float eps_f) {
    const double tol = 1e-7;
    const double eps = (double)eps_f;
    double choice_cost = 0.0; /* NOT reset per row: the reference declares it once (:450) */
    for (size_t i = 0; i < (size_t)num_rows; ++i) {
        size_t start = (size_t)i_starts_stops[i];
        size_t num_objects = (size_t)i_starts_stops[i + 1] - start;
        size_t j = (size_t)person_to_object[i];
        for (size_t idx = 0; idx < num_objects; ++idx) { /* :467-471 (last match) */
            size_t g = start + idx;
            if ((size_t)flat_j[g] == j) choice_cost = val[g];
        }
        double LHS = choice_cost - p[j] + tol; /* :475 */
        for (size_t idx = 0; idx < num_objects; ++idx) {
            size_t g = start + idx;
            double v = val[g] - p[flat_j[g]];
            if (LHS < v - eps) return 0; /* :482 */
        }
    }
    return 1;
}
static int eCE_satisfied(const oracle_solver *s, float eps_f) {
    if (s->num_unassigned > 0) return 0; /* :446-447 */
    return oracle_ece_arrays(s->num_rows, s->i_starts_stops, s->flat_j, s->val, s->p, s->person_to_object, eps_f);
}

/* auction_.pyx:433-439 */
static int is_optimal(const oracle_solver *s) {
    if (s->num_unassigned > 0) return 0;
    return eCE_satisfied(s, s->target_eps);
}
/* auction_.pyx:308-309 */
static int terminate_(const oracle_solver *s) {
    return ((int64_t)s->nits >= s->max_iter) || ((s->num_unassigned == 0) && is_optimal(s));
}

/* auction_.pyx:489-523 get_obj: double accumulator in row order (callers cast to float) */
ORACLE_API double oracle_objective(const oracle_solver *s) {
    double obj = 0;
    for (size_t i = 0; i < (size_t)s->num_rows; ++i) {
        int j = s->person_to_object[i];
        if (j == -1) continue;
        size_t start = (size_t)s->i_starts_stops[i];
        for (size_t idx = 0; idx < (size_t)s->j_counts[i]; ++idx) {
            size_t g = start + idx;
            if (s->flat_j[g] == j) {
                if (s->maximize) obj += s->val[g];
                else obj -= s->val[g];
            }
        }
    }
    return obj;
}

/* One pass of the body of the `while True` loop of solve(), auction_.pyx:271-292.
 * Returns 1 when the loop breaks. */
ORACLE_API int oracle_step(oracle_solver *s) {
    bid_and_assign(s);
    s->nits += 1;
    if (terminate_(s)) return 1;
    if (s->num_unassigned == 0) {
        if (s->eps < s->target_eps) return 1; /* :280 float compare */
        s->eps = s->eps * s->theta;           /* :283 float multiply */
        size_t N = (size_t)s->num_rows;
        for (size_t i = 0; i < N; ++i) s->person_to_object[i] = -1;                 /* :286 */
        for (size_t j = 0; j < (size_t)s->num_cols; ++j) s->object_to_person[j] = -1; /* :287 */
        s->num_unassigned = (int)N;                                                 /* :288 */
        free(s->unassigned_people);
        free(s->person_to_assignment_idx);
        s->unassigned_people = arange(N);        /* :289 */
        s->person_to_assignment_idx = arange(N); /* :290 */
        s->nreductions += 1;                     /* :292 */
    }
    return 0;
}

/* auction_.pyx:268-306 solve() */
ORACLE_API void oracle_solve(oracle_solver *s) {
    double t0 = now_s();
    for (;;)
        if (oracle_step(s)) break;
    s->t_total += now_s() - t0;
}

typedef struct oracle_meta {
    float start_eps, final_eps, target_eps;
    int eCE, soln_found;
    int its, nreductions, n_assigned, num_unassigned;
    float obj_f32;  /* get_obj() returns a C float (:489) */
    double obj_f64;
    uint64_t edges_scanned, bids_made;
    double t_bid, t_total;
    int64_t num_rows, num_cols;
} oracle_meta;

/* auction_.pyx:297-304 */
ORACLE_API void oracle_get_meta(const oracle_solver *s, oracle_meta *m) {
    m->start_eps = s->start_eps;
    m->final_eps = s->eps;
    m->target_eps = s->target_eps;
    m->eCE = eCE_satisfied(s, s->target_eps);
    m->soln_found = is_optimal(s);
    m->its = s->nits;
    m->nreductions = s->nreductions;
    m->n_assigned = (int)s->num_rows - s->num_unassigned;
    m->num_unassigned = s->num_unassigned;
    m->obj_f64 = oracle_objective(s);
    m->obj_f32 = (float)m->obj_f64;
    m->edges_scanned = s->edges_scanned;
    m->bids_made = s->bids_made;
    m->t_bid = s->t_bid;
    m->t_total = s->t_total;
    m->num_rows = s->num_rows;
    m->num_cols = s->num_cols;
}

ORACLE_API void oracle_set_timing(oracle_solver *s, int on) { s->time_phases = on; }
ORACLE_API void oracle_set_assign_by_bidders(oracle_solver *s, int on) { s->assign_by_bidders = on; }
ORACLE_API const int *oracle_person_to_object(const oracle_solver *s) { return s->person_to_object; }
ORACLE_API const int *oracle_object_to_person(const oracle_solver *s) { return s->object_to_person; }
ORACLE_API const double *oracle_prices(const oracle_solver *s) { return s->p; }
ORACLE_API const int *oracle_unassigned(const oracle_solver *s) { return s->unassigned_people; }
ORACLE_API const int *oracle_row_ptr(const oracle_solver *s) { return s->i_starts_stops; }

ORACLE_API void oracle_destroy(oracle_solver *s) {
    if (!s) return;
    free(s->p);
    free(s->i_starts_stops);
    free(s->j_counts);
    free(s->flat_j);
    free(s->person_to_object);
    free(s->object_to_person);
    free(s->best_bids);
    free(s->best_bidders);
    free(s->unassigned_people);
    free(s->person_to_assignment_idx);
    free(s);
}

/* auction_.pyx:546-557 _from_matrix dense scan: row-major, keep v >= 0.
 * Returns the number of valid entries; loc_out/val_out sized rows*cols by the caller. */
ORACLE_API int64_t oracle_dense_to_coo(const double *mat, int64_t rows, int64_t cols, int32_t *loc_out,
                                       double *val_out) {
    int64_t ctr = 0;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) {
            double v = mat[r * cols + c];
            if (v >= 0) {
                loc_out[2 * ctr] = (int32_t)r;
                loc_out[2 * ctr + 1] = (int32_t)c;
                val_out[ctr] = v;
                ctr++;
            }
        }
    return ctr;
}
