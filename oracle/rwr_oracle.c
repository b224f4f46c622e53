/*
 * CPU oracle #2: flat-array C restatement of the reference's RWR hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- the product library (recommendersystems_amd/csrc)
 * never links, loads or calls this file.  It is used by tests/, by
 * __graft_entry__.smoke() as the checker, and by bench.py's cpu_baseline leg
 * (kind "port": the reference is C# and cannot be built in this image).
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no tests or golden vectors
 * (SURVEY.md section 4).  Pinned by the hand-derived KATs of SURVEY.md section 8c
 * and by bitwise agreement with the independent literal Python restatement
 * (oracle/rwr_oracle.py) on randomised graphs (tests/test_oracle.py).
 *
 * Build:  gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC  (oracle/Makefile).
 * -ffp-contract=off is mandatory: the .NET x64 JIT emits mulsd/addsd, never FMA.
 *
 * Same algorithm shape as the reference: push-style scatter over out-links, one
 * seed at a time, sequential per seed.  Citations are relative to /root/reference/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { EDGE_UNDEFINED = 0, EDGE_LIKE = 1 };   /* Recommender.cs:5 */
enum { NODE_ITEM = 2 };                       /* Recommender.cs:4 */

/* Graph.buildGraph, Recommenders/RWRBased/Graph.cs:51-88.
 * w_norm[e] = w[e] / (left-to-right sum of the explicit weights of e's source);
 * UNDEFINED links get w_norm = 0 and are skipped by everything below.
 * dangling[i] = 1 when node i has no explicit out-link (graph[i] == null, :53,:64,:86). */
int rwr_oracle_build(int32_t n, const int64_t *rowptr, const int32_t *dst, const uint8_t *etype,
                     const double *w, double *w_norm, uint8_t *dangling)
{
    (void)dst;
    for (int32_t i = 0; i < n; i++) {
        int64_t b = rowptr[i], e = rowptr[i + 1];
        int64_t n_explicit = 0;
        for (int64_t p = b; p < e; p++)                 /* :57-61 */
            if (etype[p] != EDGE_UNDEFINED) n_explicit++;
        for (int64_t p = b; p < e; p++) w_norm[p] = 0.0;
        dangling[i] = (n_explicit == 0);
        if (n_explicit > 0) {                           /* :64 */
            double sum = 0;                             /* :70 */
            for (int64_t p = b; p < e; p++)             /* :71-77 */
                if (etype[p] != EDGE_UNDEFINED) sum += w[p];
            for (int64_t p = b; p < e; p++)             /* :80-81 */
                if (etype[p] != EDGE_UNDEFINED) w_norm[p] = w[p] / sum;
        }
    }
    return 0;
}

/* Model.deliverRanks, Model.cs:76-100.  restart_idx/restart_val list the
 * non-zero entries of restart[] (skipping zeros is bit-identical: the skipped
 * addends are +0.0 onto a non-negative accumulator); with
 * dense != 0 the O(n^2) loops of :92-93/:96-97 run literally over restart_full. */
static void deliver(int32_t n, const int64_t *rowptr, const int32_t *dst, const uint8_t *etype,
                    const double *w_norm, const uint8_t *dangling, double d,
                    const double *rank, double *next,
                    int32_t n_rs, const int32_t *restart_idx, const double *restart_val,
                    int dense, const double *restart_full)
{
    for (int32_t i = 0; i < n; i++) {                    /* :78 */
        if (!dangling[i]) {                              /* :80 */
            double rw = (1 - d) * rank[i];               /* :84 */
            for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++)   /* :85-88 */
                if (etype[p] != EDGE_UNDEFINED) next[dst[p]] += rw * w_norm[p];
            double rr = rank[i] - rw;                    /* :91 */
            if (dense) for (int32_t r = 0; r < n; r++) next[r] += rr * restart_full[r];   /* :92-93 */
            else for (int32_t q = 0; q < n_rs; q++) next[restart_idx[q]] += rr * restart_val[q];
        } else {
            double x = rank[i];
            if (dense) for (int32_t r = 0; r < n; r++) next[r] += x * restart_full[r];    /* :96-97 */
            else for (int32_t q = 0; q < n_rs; q++) next[restart_idx[q]] += x * restart_val[q];
        }
    }
}

/* Model ctor + run: Model.cs:14-31 (seed < 0, global), :33-50 (personalised),
 * :52-73 (run).  mode 0: value = iteration count; mode 1: value = threshold;
 * mode 2: threshold = (1/DBL_MAX)*n (Model.cs:53).  Returns #deliverRanks calls,
 * or -1 on bad arguments / out of memory.  max_iter bounds modes 1 and 2 (0 = unbounded). */
int64_t rwr_oracle_model_run(int32_t n, const int64_t *rowptr, const int32_t *dst, const uint8_t *etype,
                             const double *w_norm, const uint8_t *dangling, double d, int32_t seed,
                             int32_t mode, double value, int32_t dense, int64_t max_iter,
                             double *rank_out)
{
    if (n <= 0 || seed >= n) return -1;
    double *rank = (double *)malloc(sizeof(double) * n);
    double *next = (double *)calloc(n, sizeof(double));
    double *restart = (double *)calloc(n, sizeof(double));
    int32_t *ridx = (int32_t *)malloc(sizeof(int32_t) * n);
    double *rval = (double *)malloc(sizeof(double) * n);
    if (!rank || !next || !restart || !ridx || !rval) { free(rank); free(next); free(restart); free(ridx); free(rval); return -1; }
    int32_t n_rs = 0;
    if (seed < 0) {
        for (int32_t i = 0; i < n; i++) { rank[i] = 1.0; restart[i] = 1.0 / n; }       /* :23-30 */
    } else {
        for (int32_t i = 0; i < n; i++) { rank[i] = (i == seed) ? (double)n : 0; restart[i] = (i == seed) ? 1.0 : 0; }  /* :42-49 */
    }
    for (int32_t i = 0; i < n; i++) if (restart[i] != 0) { ridx[n_rs] = i; rval[n_rs] = restart[i]; n_rs++; }

    int64_t it = 0;
    if (mode == 0) {
        int64_t T = (int64_t)value;
        for (; it < T; it++) {                                                          /* :69-72 */
            deliver(n, rowptr, dst, etype, w_norm, dangling, d, rank, next, n_rs, ridx, rval, dense, restart);
            for (int32_t i = 0; i < n; i++) { rank[i] = next[i]; next[i] = 0; }         /* :104-107 */
        }
    } else {
        double thr = (mode == 2) ? (1 / DBL_MAX) * n : value;                           /* :53 */
        for (;;) {                                                                      /* :58-65 */
            deliver(n, rowptr, dst, etype, w_norm, dangling, d, rank, next, n_rs, ridx, rval, dense, restart);
            it++;
            double diff = 0;                                                            /* :111-113 */
            for (int32_t i = 0; i < n; i++)
                diff += (rank[i] > next[i]) ? (rank[i] - next[i]) : (next[i] - rank[i]);
            int conv = diff < thr;
            for (int32_t i = 0; i < n; i++) { rank[i] = next[i]; next[i] = 0; }
            if (conv) break;
            if (max_iter > 0 && it >= max_iter) break;
        }
    }
    memcpy(rank_out, rank, sizeof(double) * n);
    free(rank); free(next); free(restart); free(ridx); free(rval);
    return it;
}

typedef struct { int64_t id; double score; } kv_t;

/* Recommender.cs:35-38: score descending, then id descending. */
static int cmp_kv(const void *a, const void *b)
{
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    if (x->id > y->id) return -1;
    if (x->id < y->id) return 1;
    return 0;
}

/* Recommender.Recommendation, Recommender.cs:14-40 (+ top-N overload :42-51).
 * d crosses as float and is widened (Recommender.cs:14,16 -> Model.cs:33).
 * out_id/out_score must hold min(top_n, #items) entries (all items if top_n <= 0).
 * rank_out (optional, n doubles) receives the final rank vector.
 * Returns the number of entries written, or -1. */
int64_t rwr_oracle_recommend(int32_t n, const int64_t *node_id, const uint8_t *node_type,
                             const int64_t *rowptr, const int32_t *dst, const uint8_t *etype,
                             const double *w_norm, const uint8_t *dangling,
                             int32_t seed, float d_f, int32_t n_iter, int64_t top_n,
                             int64_t *out_id, double *out_score, double *rank_out)
{
    if (seed < 0 || seed >= n) return -1;
    double d = (double)d_f;
    double *rank = (double *)malloc(sizeof(double) * n);
    uint8_t *excl = (uint8_t *)calloc(n, 1);
    kv_t *rec = (kv_t *)malloc(sizeof(kv_t) * (size_t)n);
    if (!rank || !excl || !rec) { free(rank); free(excl); free(rec); return -1; }
    if (rwr_oracle_model_run(n, rowptr, dst, etype, w_norm, dangling, d, seed, 0, (double)n_iter, 0, 0, rank) < 0) {
        free(rank); free(excl); free(rec); return -1;
    }
    for (int64_t p = rowptr[seed]; p < rowptr[seed + 1]; p++)       /* :20-24, raw list */
        if (etype[p] == EDGE_LIKE) excl[dst[p]] = 1;
    int64_t m = 0;
    for (int32_t i = 0; i < n; i++)                                 /* :27-31 */
        if (node_type[i] == NODE_ITEM && !excl[i]) { rec[m].id = node_id[i]; rec[m].score = rank[i]; m++; }
    qsort(rec, (size_t)m, sizeof(kv_t), cmp_kv);                    /* :35-38 (total order: ids unique) */
    int64_t cnt = (top_n > 0 && top_n < m) ? top_n : m;             /* :42-51 */
    for (int64_t q = 0; q < cnt; q++) { out_id[q] = rec[q].id; out_score[q] = rec[q].score; }
    if (rank_out) memcpy(rank_out, rank, sizeof(double) * n);
    free(rank); free(excl); free(rec);
    return cnt;
}

/* CPU baseline driver: one seed per host thread (the reference's own parallelism
 * is one thread per graph, Program.cs:11,61-66).  top_n must be > 0.
 * out_id/out_score are K x top_n; counts[k] receives the entries written. */
int rwr_oracle_recommend_batch(int32_t n, const int64_t *node_id, const uint8_t *node_type,
                               const int64_t *rowptr, const int32_t *dst, const uint8_t *etype,
                               const double *w_norm, const uint8_t *dangling,
                               const int32_t *seeds, int32_t K, float d_f, int32_t n_iter, int32_t top_n,
                               int32_t n_threads, int64_t *out_id, double *out_score, int32_t *counts)
{
    int bad = 0;
    if (top_n <= 0) return -1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t k = 0; k < K; k++) {
        int64_t c = rwr_oracle_recommend(n, node_id, node_type, rowptr, dst, etype, w_norm, dangling,
                                         seeds[k], d_f, n_iter, top_n,
                                         out_id + (size_t)k * top_n, out_score + (size_t)k * top_n, NULL);
        if (c < 0) {
#pragma omp atomic write
            bad = 1;
            counts[k] = 0;
        } else counts[k] = (int32_t)c;
    }
    return bad ? -1 : 0;
}

static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* Experiment.cs:121-128: hits and running-precision sum of a ranked id list against the test set. */
int rwr_oracle_evaluate(const int64_t *ranked_ids, int64_t count, const int64_t *test_ids, int64_t n_test,
                        int64_t *hits_out, double *sum_precision_out)
{
    int64_t *ts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_test > 0 ? n_test : 1));
    if (!ts) return -1;
    memcpy(ts, test_ids, sizeof(int64_t) * (size_t)n_test);
    qsort(ts, (size_t)n_test, sizeof(int64_t), cmp_i64);
    int64_t nHits = 0;
    double sumPrecision = 0;
    for (int64_t i = 0; i < count; i++) {                                   /* :123 */
        if (bsearch(&ranked_ids[i], ts, (size_t)n_test, sizeof(int64_t), cmp_i64)) {   /* :124 testSet.Contains */
            nHits += 1;
            sumPrecision += (double)nHits / (double)(i + 1);                /* :126 */
        }
    }
    free(ts);
    *hits_out = nHits;
    *sum_precision_out = sumPrecision;
    return 0;
}

int rwr_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
