/*
 * librwr -- MI355X-native Random-Walk-with-Restart engine, C-ABI boundary.
 *
 * This header is the drop-in boundary for the public surface of the reference's
 * namespace Recommenders.RWRBased (ChangUk/RecommenderSystems).  The reference is
 * 100 % managed C# with no FFI of its own; a thin C# shim (INTEGRATION.md,
 * csharp/Recommenders/RWRBased/) keeps the reference's public types and P/Invokes
 * exactly the entry points declared here.  Every entry point cites the reference
 * interface it replaces as file:line relative to the reference root.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary;
 *   - every function returns an int32 status (RWR_OK == 0); rwr_last_error() returns a
 *     thread-local UTF-8 message for the last failing call on the calling thread
 *     (the shim turns statuses into the .NET exceptions the reference would throw);
 *   - all pointers are HOST pointers unless a parameter says otherwise;
 *   - every entry point makes the graph's device current for the call and restores the calling
 *     thread's current HIP device before it returns;
 *   - handles are independent: distinct rwr_graph handles may be used concurrently from
 *     distinct host threads (the reference runs up to 10 worker threads, each with its
 *     own Graph/Recommender: Program.cs:11,61-66; Experiment.cs:71,104,108).  One handle
 *     must not be used from two threads at once;
 *   - there is NO CPU fallback: every compute entry point fails with RWR_E_NO_DEVICE
 *     when no gfx950 device is usable.
 */
#ifndef RWR_H
#define RWR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RWR_VERSION_STRING "0.3.0"

/* status codes */
enum {
    RWR_OK            = 0,
    RWR_E_INVALID     = 1,   /* bad argument (null pointer, negative size, malformed CSR)        */
    RWR_E_RANGE       = 2,   /* node index out of range  -> ArgumentOutOfRangeException/KeyNotFound */
    RWR_E_NO_DEVICE   = 3,   /* no usable HIP device                                             */
    RWR_E_HIP         = 4,   /* a HIP runtime call failed (message has the HIP error string)     */
    RWR_E_NOMEM       = 5,   /* host or device allocation failed                                 */
    RWR_E_CAPACITY    = 6,   /* caller's output buffer too small (needed size returned)          */
    RWR_E_UNSUPPORTED = 7    /* valid request this build does not implement                      */
};

/* enum NodeType { UNDEFINED, USER, ITEM, ETC }                     -- Recommender.cs:4 */
enum { RWR_NODE_UNDEFINED = 0, RWR_NODE_USER = 1, RWR_NODE_ITEM = 2, RWR_NODE_ETC = 3 };
/* enum EdgeType { UNDEFINED, LIKE, FRIENDSHIP, FOLLOW, MENTION, AUTHORSHIP, PURCHASE, ETC }
 *                                                                  -- Recommender.cs:5 */
enum { RWR_EDGE_UNDEFINED = 0, RWR_EDGE_LIKE = 1, RWR_EDGE_FRIENDSHIP = 2, RWR_EDGE_FOLLOW = 3,
       RWR_EDGE_MENTION = 4, RWR_EDGE_AUTHORSHIP = 5, RWR_EDGE_PURCHASE = 6, RWR_EDGE_ETC = 7 };

/* arithmetic mode of the power iteration.  There is one: every rank value is bitwise what
 * Model.deliverRanks (Model.cs:76-100) computes -- same addends, same order, no FMA contraction -- and
 * every ranked list is the reference's (Recommender.cs:35-38).  (Value 1 was RWR_MODE_FAST, a
 * re-associated mode without a ranking guarantee; it was removed in ABI 3 and rwr_graph_create
 * now rejects it with RWR_E_INVALID.) */
enum { RWR_MODE_EXACT = 0 };

/* rwr_model_run stop rule -- Model.run(int) / run(double) / run(): Model.cs:68-73,57-66,52-55 */
enum { RWR_RUN_ITERATIONS = 0, RWR_RUN_THRESHOLD = 1, RWR_RUN_DEFAULT_THRESHOLD = 2 };

typedef struct rwr_graph rwr_graph;   /* opaque: device-resident graph + workspaces + streams */

typedef struct rwr_opts {
    int32_t struct_size;     /* = sizeof(rwr_opts); lets the struct grow compatibly            */
    int32_t device;          /* HIP device ordinal; -1 = env RWR_DEVICE, else current device   */
    int32_t mode;            /* RWR_MODE_EXACT; -1 = default (EXACT)                           */
    int32_t tile_seeds;      /* seeds per rank-matrix tile (lanes per row): 1,2,4,8,16,32,64;
                                0 = auto                                                        */
    int32_t tile_group;      /* tiles iterated together (grid.y of the SpMM launch); 0 = auto  */
    int32_t profile;         /* 1 = bracket every kernel phase with HIP events (rwr_get_stats)  */
    int64_t workspace_bytes; /* cap on the batch workspace (rank matrices + sort buffers);
                                0 = auto (a fraction of free HBM)                               */
    int32_t seed_row_kernel; /* EXACT mode, how the seed's own row (the n-term restart chain,
                                Model.cs:91-93,96-97) is folded -- every choice is bitwise equal:
                                0 = auto (env RWR_CHAIN, else by batch size), 1 = sequential fold
                                beside the SpMM, 2 = parallel binade scan, 3 = simple one-lane loop */
    int32_t reserved0;
} rwr_opts;

/* accumulated since creation or the last rwr_reset_stats(); *_ms are device times measured
 * with HIP events on the library's own streams (only when opts.profile != 0) */
typedef struct rwr_stats {
    int32_t struct_size;
    int32_t n;               /* nodes                                                           */
    int64_t nnz_raw;         /* links handed over (incl. UNDEFINED)                             */
    int64_t nnz;             /* explicit links (type != UNDEFINED) = entries of P               */
    int32_t uniform;         /* 1 when every row's explicit raw weights are equal               */
    int32_t tile_seeds;      /* resolved seeds per tile of the last batch                       */
    int32_t tile_group;      /* resolved tiles per launch of the last batch                     */
    int32_t mode;
    double  build_ms;        /* device-side Graph.buildGraph + transpose (one-off)              */
    double  spmm_ms;         /* sum of SpMM launch durations                                    */
    int64_t spmm_launches;
    int64_t spmm_seed_steps; /* sum over launches of (seeds in the launch) -- one unit = one
                                power-iteration step of one seed                                */
    double  chain_ms;        /* seed-row kernels (sequential fold or binade scan)                */
    int64_t chain_launches;
    double  rank_ms;         /* exclusion mask + top-k / sort + gather                          */
    double  iterate_wall_ms; /* device time from the first to the last event of the iterate
                                phase (SpMM and seed-row kernels overlapped)                    */
    double  total_wall_ms;   /* host wall time inside rwr_recommend* calls                      */
    int64_t seeds_done;
    int64_t chain_redo_blocks; /* binade scan: blocks the carry had to redo row by row (binade
                                  crossings + mispredicted binades), summed over steps and seeds */
    /* the DENSE launches among spmm_launches: every row of the matrix is walked and every entry's
     * source row gathered (no frontier bitmap, no skipped rows).  Only these are priced with the full
     * algorithmic byte count by bench.py's roofline; the first iterations of a run, which skip the rows
     * that are still exactly zero, are not */
    double  spmm_dense_ms;
    int64_t spmm_dense_launches;
    int64_t spmm_dense_seed_steps;
    int32_t uniform_path;    /* 1 when the value-free matrix path serves this graph: every row's weights are
                                equal (unit raw weights, DataLoader.cs:293-294), so fl(fl((1-d) rank[i]) * weight)
                                (Model.cs:84,87) is ONE double per source node and step; the kernels gather it and
                                read no per-entry value (4 instead of 12 matrix bytes per entry) -- bitwise equal */
    int32_t reserved1;
} rwr_stats;

/* ---- library ------------------------------------------------------------------------- */

const char *rwr_version(void);
/* number of usable gfx950 devices (0 when none); never fails */
int32_t rwr_device_count(void);
/* message of the last failure on this thread ("" if none) */
const char *rwr_last_error(void);

/* ---- Graph ---------------------------------------------------------------------------
 * Replaces  new Graph(nodes, edges) + Graph.buildGraph()   (Graph.cs:45-49, 51-88; called
 * at Experiment.cs:104-105) and Graph.size() (Graph.cs:91-93).
 *
 * Input is the RAW link list exactly as the host built it, flattened in dictionary/list
 * order:  node i (0..n-1, the keys of Dictionary<int,Node>, Graph.cs:39) has
 * id node_id[i] / type node_type[i] (struct Node, Graph.cs:4-17) and out-links
 * rowptr[i]..rowptr[i+1]-1, each (dst, etype, w) = struct ForwardLink
 * {targetNode, type, weight} (Graph.cs:19-35) in List<ForwardLink> order
 * (Graph.cs:40).  Links are NOT filtered or normalised by the caller: the library drops
 * type == UNDEFINED links, sums the remaining weights of a source left to right and
 * divides (Graph.cs:57-81) on the device, then builds the in-neighbour (transposed) CSR
 * with entries ordered (source asc, list position asc) -- the addend order of
 * Model.deliverRanks.  A node without explicit links is dangling (Graph.cs:53,64,86).
 * A key missing from the edge dictionary is passed as an empty list.
 * The arrays are copied; the caller may free them after the call returns. */
int32_t rwr_graph_create(int32_t n, const int64_t *node_id, const uint8_t *node_type,
                         const int64_t *rowptr /* n+1 */, const int32_t *dst, const uint8_t *etype,
                         const double *w, const rwr_opts *opts /* may be NULL */, rwr_graph **out);
int32_t rwr_graph_destroy(rwr_graph *g);
/* Incremental rebuild.  The harness builds an almost identical graph for every fold and
 * methodology (new DataLoader + new Graph + buildGraph per fold, Experiment.cs:69-77,104-105),
 * and three methodologies differ from their base only by relabelling FRIENDSHIP links to
 * UNDEFINED (Experiment.cs:84-101).  When node list, rowptr and dst are unchanged, only the
 * links whose type or weight differ need to cross the boundary: link_index[q] is a position
 * in the flattened raw list given to rwr_graph_create (distinct positions), etype[q] / w[q]
 * its new type / raw weight (either array may be NULL = unchanged).  The library patches its
 * resident raw lists and re-runs Graph.buildGraph (Graph.cs:51-88) and the transpose on the
 * device; the resulting state is bit for bit what rwr_graph_create would build from the
 * patched lists.  count == 0 just rebuilds.  Not to be called concurrently with other calls
 * on the same handle.  Argument errors (RWR_E_INVALID / RWR_E_RANGE) are detected before
 * anything is patched and leave the graph as it was; a failure AFTER the patch (out of device
 * memory during the rebuild, a HIP error) invalidates the handle: every later call on it
 * fails with RWR_E_INVALID until it is destroyed. */
int32_t rwr_graph_update_links(rwr_graph *g, int64_t count, const int64_t *link_index,
                               const uint8_t *etype /* may be NULL */, const double *w /* may be NULL */);
/* Graph.size() (Graph.cs:91-93) plus link counts; any out pointer may be NULL */
int32_t rwr_graph_size(const rwr_graph *g, int32_t *n, int64_t *nnz_raw, int64_t *nnz_explicit);
/* Backs the public field Graph.graph (Graph.cs:43): w_out[e] = normalised weight of raw
 * link e (0 for UNDEFINED links, which the reference leaves out of graph[i]);
 * dangling_out[i] = 1 when graph[i] == null.  Either pointer may be NULL. */
int32_t rwr_graph_get_normalized(rwr_graph *g, double *w_out /* nnz_raw */, uint8_t *dangling_out /* n */);

/* ---- Recommender ---------------------------------------------------------------------
 * Replaces Recommender.Recommendation(int idxTargetUser, float dampingFactor, int nIteration)
 * (Recommender.cs:14-40) and the topN overload (Recommender.cs:42-51):
 *   personalised Model (Model.cs:33-50), run(nIteration) (Model.cs:68-73), exclusion of the
 *   seed's raw LIKE out-links (Recommender.cs:20-24), candidates = ITEM nodes not excluded
 *   (:27-31), sorted by score desc then id desc (:35-38).
 * d crosses as float and is widened natively, as Recommender.cs:16 -> Model.cs:33 does.
 * top_n <= 0 returns the whole list (the reference's Count == topN test never fires).
 * Domain: raw weights >= 0 with a positive finite sum per node (what the reference's loader
 * produces) and 0 <= d <= 1; on a graph with a negative / NaN weight or a zero row sum, or
 * for a damping factor outside [0, 1] (ranks would go negative), the Recommendation entries
 * fail with RWR_E_UNSUPPORTED (rwr_model_run still reproduces the reference there).
 * in:  *inout_count = capacity of out_id/out_score;  out: entries written.  If the list
 * needs more room the call fails with RWR_E_CAPACITY and *inout_count = required size
 * (n always suffices). */
int32_t rwr_recommend(rwr_graph *g, int32_t seed, float d, int32_t n_iter, int32_t top_n,
                      int64_t *out_id, double *out_score, int64_t *inout_count);

/* Recommendation + the evaluation the harness runs on its result (TweetRecommender/Experiment.cs:109,121-128):
 * walks the FULL ranked list and returns  n_hits = #{ranked ids in test_ids}  and
 * sum_precision = sum over hits, in rank order, of (double)hitsSoFar / (position + 1)  -- the harness then reports
 * n_hits and sum_precision / n_hits (Experiment.cs:131-138).  The list itself never leaves the device.
 * list_len (optional) receives the length of the ranked list. */
int32_t rwr_recommend_eval(rwr_graph *g, int32_t seed, float d, int32_t n_iter, const int64_t *test_ids,
                           int64_t n_test, int64_t *n_hits, double *sum_precision, int64_t *list_len);

/* The same for K seeds of one graph, each with its own test set: test set k = test_ids[test_ptr[k] .. test_ptr[k+1])
 * (CSR form, test_ptr has K + 1 entries).  n_hits / sum_precision / list_len (optional) have K entries; entry k is
 * exactly what rwr_recommend_eval(seeds[k], ..., test set k) returns.  The K full ranked lists are produced by the
 * batched iteration and evaluated where they lie (one workgroup per seed); nothing but 16 bytes per seed comes back. */
int32_t rwr_recommend_eval_batch(rwr_graph *g, const int32_t *seeds, int32_t K, float d, int32_t n_iter,
                                 const int64_t *test_ptr, const int64_t *test_ids, int64_t *n_hits,
                                 double *sum_precision, int64_t *list_len);

/* ---- Many small graphs at once ---------------------------------------------------------
 * The reference's own workload (TweetRecommender/Experiment.cs:64-134, ten threads of it: Program.cs:11) is thousands of
 * ego-network-sized graphs, each built once, asked ONE Recommendation(seed, d, nIteration) and evaluated against its fold's
 * test set.  Entry g of the batch is exactly
 *     rwr_graph_create(graphs[g]) ; rwr_recommend_eval(seeds[g], d, n_iter, test set g) ; rwr_graph_destroy
 * -- the same arrays, the same arithmetic, bit for bit -- but graphs that fit the one-launch build and the one-launch call
 * (up to 6144 nodes / 4096 items / 65536 links) share ONE build launch, ONE iteration launch and ONE evaluation launch for
 * the whole batch (a workgroup per graph) and three synchronisations in total; larger graphs of the batch take the calls
 * above one by one.  test set g = test_ids[test_ptr[g] .. test_ptr[g+1]); n_hits / sum_precision / list_len (optional)
 * have `count` entries.  opts as for rwr_graph_create (device, mode). */
typedef struct rwr_graph_desc {
    int32_t n_nodes;
    int32_t reserved0;
    const int64_t *node_id;        /* [n_nodes] */
    const uint8_t *node_type;      /* [n_nodes] */
    const int64_t *rowptr;         /* [n_nodes + 1] */
    const int32_t *dst;            /* [rowptr[n_nodes]] */
    const uint8_t *etype;
    const double *w;
} rwr_graph_desc;
int32_t rwr_eval_graphs(int32_t count, const rwr_graph_desc *graphs, const int32_t *seeds, float d, int32_t n_iter,
                        const int64_t *test_ptr, const int64_t *test_ids, const rwr_opts *opts, int64_t *n_hits,
                        double *sum_precision, int64_t *list_len);

/* Batch entry (an addition: the reference creates a fresh Model per call, Recommender.cs:16,
 * so seeds are independent and batching is semantically free).  top_n must be >= 1.
 * ids/scores are K x top_n row-major; counts[k] = entries valid in row k (the rest of the
 * row is id 0 / score 0).  Result k is exactly rwr_recommend(seeds[k], d, n_iter, top_n). */
int32_t rwr_recommend_batch(rwr_graph *g, const int32_t *seeds, int32_t K, float d, int32_t n_iter,
                            int32_t top_n, int64_t *ids, double *scores, int32_t *counts);

/* ---- Model ---------------------------------------------------------------------------
 * Backs the public class Model: ctor (Model.cs:33-50 personalised, seed >= 0;
 * Model.cs:14-31 global, seed == -1), run(int) / run(double) / run() (Model.cs:68-73,
 * 57-66, 52-55 incl. checkConvergence :110-115).  d is already a double here
 * (Model.cs:33 takes double).  rank_out receives Model.rank (n doubles) after the run;
 * iters_out (optional) the number of deliverRanks() calls made. */
int32_t rwr_model_run(rwr_graph *g, int32_t seed, double d, int32_t run_mode, double value,
                      double *rank_out, int64_t *iters_out);
/* ONE Model.deliverRanks() (Model.cs:76-100) on a rank vector held by the caller: next_rank = what the reference would
 * leave in nextRank (which updateRanks() has zeroed before, Model.cs:103-108) for the given rank.  Backs the public
 * step-by-step API -- deliverRanks / updateRanks / checkConvergence (Model.cs:76,103,110) -- for a host that drives the
 * loop itself; updateRanks and checkConvergence are array operations the shim does on its own copies.  seed >= 0:
 * personalised restart vector e_seed (bitwise in EXACT mode, for any rank vector); seed == -1: the global model's
 * uniform restart (tolerance parity, as rwr_model_run).  rank and next_rank hold n doubles and may not alias. */
int32_t rwr_model_deliver(rwr_graph *g, int32_t seed, double d, const double *rank, double *next_rank);

/* ---- row-partitioned mode (graphs beyond one GPU; BASELINE.json config 5) ------------
 * An ADDITION: the reference has no distributed mode.  The transition matrix is partitioned by SOURCE rows
 * (the reference's native layout: graph[i] = out-links of i, Graph.cs:43): rank r owns the contiguous node slab
 * [slab_lo, slab_hi) and creates its rwr_graph from the FULL node arrays but with out-links only for its own
 * rows (rowptr flat outside the slab).  Every rank keeps the full rank matrix x[n][G] (G = tile width,
 * K <= 64 seeds); one power-iteration step is
 *     rwr_part_local_step    y = (1-d) * P_slab^T x   over ALL n rows,  r[k] = restart mass of the slab's rows
 *     (caller)               all-reduce(sum) of y and r over the ranks  -- RCCL over xGMI, torch.distributed
 *     rwr_part_finish_step   y[seed_k][k] += r[k]      (Model.cs:91-93,96-97)
 * after which y is the next x.  dev_* are DEVICE pointers owned by the caller (n*G, n*G and G doubles).
 * Partial sums re-associate across ranks: parity with the single-GPU result is to tolerance (scores
 * within 1e-6; identical top-k lists in the tests), never bitwise.  rwr_part_rank ranks the seeds whose row
 * this rank owns (only the owner holds the seed's raw LIKE links for the exclusion list,
 * Recommender.cs:20-24) and reports counts[k] = -1 for the others. */
int32_t rwr_part_begin(rwr_graph *g, int32_t slab_lo, int32_t slab_hi, const int32_t *seeds, int32_t K, double d,
                       void *dev_x, int32_t *tile_seeds_out);
/* The preferred step: ONE asynchronous call per iteration, enqueued on the CALLER's HIP stream (`stream` = a hipStream_t,
 * e.g. the stream torch.distributed / RCCL is ordered after; NULL = the device's default stream, which is what
 * torch.cuda.current_stream() denotes unless the caller switched streams) with no host synchronisation:
 *     y = (1-d) * P_slab^T x  over ALL n rows,  then  y[seed_k][k] += restart mass of the slab's rows  (Model.cs:91-93,96-97)
 * so that the sum over the ranks of y is the complete next rank matrix.  A rank only ever READS its own slab of x (its
 * graph holds out-links of its own rows only), so the exchange is a REDUCE-SCATTER of y by slabs -- half the bytes of an
 * all-reduce, and no second collective for the restart scalars; only the last step needs every row everywhere (ranking)
 * and uses an all-reduce.  recommendersystems_amd/partitioned.py is the host logic. */
int32_t rwr_part_step(rwr_graph *g, const void *dev_x, void *dev_y, void *stream);
int32_t rwr_part_local_step(rwr_graph *g, const void *dev_x, void *dev_y, void *dev_r);
int32_t rwr_part_finish_step(rwr_graph *g, void *dev_y, const void *dev_r);
int32_t rwr_part_rank(rwr_graph *g, void *dev_x, int32_t top_n, int64_t *ids, double *scores, int32_t *counts);

/* ---- measurement ---------------------------------------------------------------------- */
int32_t rwr_get_stats(rwr_graph *g, rwr_stats *out);
int32_t rwr_reset_stats(rwr_graph *g);

#ifdef __cplusplus
}
#endif
#endif /* RWR_H */
