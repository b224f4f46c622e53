// Device-resident graph and batch workspace (internal).
#pragma once
#include <cstdlib>

#include "common.h"

struct rwr_graph {
    int32_t device = 0;
    int32_t n = 0;
    int64_t nnz_raw = 0;     // links handed over
    int64_t nnz = 0;         // explicit links = entries of the transition matrix
    int32_t n_items = 0;     // nodes of type ITEM
    uint64_t id_key_top = 0; // largest ITEM id (order-preserving unsigned form) and the bits of the id range: sort keys of item_order
    int32_t id_key_bits = 64;
    int32_t uniform = 0;     // every row's explicit raw weights equal
    // Value-free matrix path (uniform && nonneg, RWR_VALUE_FREE != 0).  Every out-link of source i then carries the SAME
    // normalised weight w_src[i] (Graph.cs:79-81 divides equal raw weights by one sum), so the product Model.cs:87 adds,
    //     fl( fl((1-d) * rank[i]) * weight ),
    // is one and the same double for all links of i: it is formed once per node and step (z[i], written by the kernel
    // that produced rank[i]) and the SpMM / SpMV gather z and read NO per-entry value -- 4 instead of 12 matrix bytes per
    // entry and one add per entry, bit for bit the reference's sums.  in_w is not even built on such graphs
    // (ensure_in_w materialises it for the few entry points that still take the weighted kernels).
    int32_t vf = 0;
    int32_t nonneg = 1;      // every normalised weight is a finite number >= 0 (raw weights >= 0, row sums in (0, inf)): ranks stay
                             // >= 0, which the zero-skipping frontier paths and the binade scan rely on; otherwise the general kernels run
    int32_t max_in_deg = 0;
    int32_t staged = 0;      // ego-network-sized graph: raw arrays arrived through the pinned staging buffer (build.hip)
    int32_t stage_pending = 0;   // ... and the device-side copy of that buffer has not been unpacked into the arrays yet
    int32_t poisoned = 0;    // a failed incremental rebuild left raw and derived arrays out of step: every entry point refuses
    // rows of row_order (in-degree descending) with in-degree >= 128 / >= 32 / >= 4: lane-width bins of the K = 1 vector SpMV
    int32_t bin_end[3] = {0, 0, 0};
    int32_t bin_huge = 0;    // rows with in-degree >= 2048
    rwr_opts opts{};

    // node SoA (struct Node, Graph.cs:4-17)
    rwr::DevBuf<int64_t> node_id;
    rwr::DevBuf<uint8_t> node_type;
    // raw out-links in list order (struct ForwardLink, Graph.cs:19-35); kept for the
    // exclusion list, which reads the RAW list (Recommender.cs:20-24)
    rwr::DevBuf<int64_t> rowptr;
    rwr::DevBuf<int32_t> dst;
    rwr::DevBuf<uint8_t> etype;
    rwr::DevBuf<double> w_raw;        // raw weights as handed over (kept for incremental rebuilds)
    rwr::DevBuf<double> w_norm_raw;   // Graph.graph weights per raw link (0 for UNDEFINED)
    std::vector<int64_t> h_rowptr;    // host copy (seed validation, exclusion sizing)
    std::vector<uint8_t> h_dangling;  // host copy of dangling[] (dangling seeds are answered without iterating)
    std::vector<int64_t> h_in_ptr;    // host copy of in_ptr (sizing of the seeds' in-link term buffers)
    // transposed (in-neighbour) CSR of the normalised matrix, entries ordered
    // (source asc, list position asc) = addend order of Model.deliverRanks
    rwr::DevBuf<int64_t> in_ptr;
    rwr::DevBuf<int32_t> in_src;
    rwr::DevBuf<double> in_w;
    rwr::DevBuf<double> w_src;        // uniform graphs: the one normalised weight of source i
    rwr::DevBuf<uint8_t> dangling;    // graph[i] == null (Graph.cs:53,86)
    rwr::DevBuf<int32_t> row_order;   // destination rows by in-degree descending (stable)
    // single-seed SpMV: ITEM rows first, then the rest, each by in-degree descending.  An item's in-links come from users and
    // a user's mostly from items, so each phase gathers from one region of the rank vector and the XCDs' L2s are not split
    // between the two regions; x_rows[p] = rows of phase p, x_bins[p][0..2] = how many of them have >= 128 / 32 / 4 in-links
    rwr::DevBuf<int32_t> row_order_x;
    int32_t x_rows[2] = {0, 0};
    int32_t x_bins[2][3] = {{0, 0, 0}, {0, 0, 0}};
    int32_t x_hub[2] = {0, 0};        // rows of phase p with in-degree >= hub_t (hub rows of the exact single-seed SpMV)
    int32_t bin_hub = 0, hub_t = 2048; // ... and of the one-phase order; the threshold (RWR_HUB_T)
    std::vector<uint8_t> h_is_item;
    rwr::DevBuf<int32_t> item_order;  // ITEM rows by id descending
    rwr::DevBuf<int32_t> item_rows;   // ITEM rows by row index ascending

    // single-seed SpMV of value-free graphs as a source-block sweep with z staged through LDS (sweep.hip): 0 = not decided
    // yet, 1 = tables built, -1 = this graph does not qualify; re-decided after every (re)build
    int32_t sw_state = 0, sw_B = 0, sw_BN = 0, sw_K = 0, sw_nwg = 0, sw_wpg = 0, sw_hub0 = 0;
    int32_t sw_rows = 0;              // rows the sweep serves (all non-hub rows, or -- sw_partial -- the non-hub ITEM rows only)
    int32_t sw_partial = 0;           // 1: phase 0 (ITEM rows) by the sweep, phase 1 by the row-binned kernel beside it
    rwr::DevBuf<uint32_t> sw_wgblk;   // [workgroup][block]: does any row of the workgroup read that block (else its refill is skipped)
    int64_t sw_words = 0;             // 64-bit words of the entry stream (4 block-local 16-bit indices each)
    rwr::DevBuf<int32_t> sw_order;    // rows in sweep order (row_order_x without its hub rows)
    rwr::DevBuf<uint4> sw_meta;       // [wave][block]: first word-row of the chunk, piece lengths of the wave's slots
    rwr::DevBuf<uint2> sw_ent;        // the matrix in sweep order

    // batch workspace (lazily sized)
    rwr::DevBuf<double> X, Y;         // rank matrices [tile][n][G]
    rwr::DevBuf<int32_t> sm_tab;      // small.hip: places of the seed row's addends (per call)
    void *sm_pin = nullptr;           // small.hip: pinned host buffer the one-launch kernel writes the ranked list into
    void *sm_stage = nullptr;         // build.hip: pinned staging buffer of an ego-network-sized graph's upload / read-back
    // a handle built as part of a batch (multi.hip): its staging slot and read-back area lie in the batch's pinned arena, the
    // device-side landing area in the batch's one buffer (one H2D copy for all graphs); none of them is owned by the handle
    uint8_t *sm_out = nullptr;
    const uint8_t *stage_dev_ext = nullptr;
    int32_t borrowed = 0;             // streams / events / pinned buffers belong to the batch context, not to this handle
    rwr::DevBuf<uint8_t> d_stage;     // ... and its device-side landing area
    int32_t sm_pin_count = -1;        // >= 0: the last single-seed call left its list (that many entries) in sm_pin
    rwr::DevBuf<double> Z0, Z1;       // value-free path: z = ((1-d) x) * w_src of the current / next ranks, same layout
    rwr::DevBuf<int32_t> d_seeds;     // [tile][G], -1 = padding lane
    rwr::DevBuf<int32_t> d_slot_k;    // [tile][G]: batch position of the seed in this slot (-1 = padding)
    rwr::DevBuf<double> d_part;       // partial sums of the deterministic tree reductions (global model, row-partitioned restart mass)
    rwr::DevBuf<unsigned int> d_gate;   // exact mode: check-in counter of the resident chain workgroups
    rwr::DevBuf<uint32_t> d_nz;       // [2][tile][ceil(n/32)] bitmaps: row of X / Y has a non-zero (first iterations)
    rwr::DevBuf<int64_t> d_evoff;     // exact mode: per seed slot, offset of its in-link terms in d_evterm
    rwr::DevBuf<double> d_evterm;     // exact mode: ((1-d) x_src) * w of every link INTO a seed, list order
    // exact mode, parallel seed-row chain (chain_scan.hip): per (tile, block, seed) approximate block sum,
    // predicted biased exponent, parity-function pair; per (slot, block) offset into the seed's in-link list
    rwr::DevBuf<double> cs_approx;
    rwr::DevBuf<int32_t> cs_e;
    rwr::DevBuf<long long> cs_d0, cs_d1;
    rwr::DevBuf<double> cs_side;      // single seed: CS_SIDE_WORDS words per block (chain_scan.hip)
    rwr::DevBuf<double> cs_mx;        // single seed: scratch row of the exact-start block's addend sequence
    rwr::DevBuf<int32_t> cs_lnk;
    rwr::DevBuf<int32_t> cs_lnk0;     // all-zero link table + slot for chain_scan_sum (checkConvergence)
    rwr::DevBuf<double> cs_diff;      // |rank - nextRank| per node (checkConvergence)
    rwr::DevBuf<unsigned long long> cs_redo;   // blocks redone by the carry kernel (binade crossings + mispredictions)
    rwr::DevBuf<uint64_t> keys, keys_alt;
    rwr::DevBuf<uint32_t> vals, vals_alt;
    rwr::DevBuf<uint8_t> sort_temp;
    rwr::DevBuf<int64_t> d_out_id;
    rwr::DevBuf<double> d_out_score;
    rwr::DevBuf<int32_t> d_counts;

    // row-partitioned mode (rwr_part_*)
    int32_t part_lo = 0, part_hi = 0, part_G = 0, part_K = 0;
    int32_t part_steps = 0;           // slab steps since rwr_part_begin (the first ones mark the frontier, like the seed path)
    double part_c1 = 0;
    std::vector<int32_t> part_seeds;

    hipStream_t stream = nullptr, stream2 = nullptr;
    hipStream_t stream3 = nullptr;    // hub rows of the exact single-seed SpMV, beside the binned kernel
    hipEvent_t ev_h0 = nullptr, ev_h1 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;   // profiling pairs

    rwr_stats stats{};
    std::vector<uint8_t> spmm_ev_dense;   // per profiled SpMM launch: 1 = dense (no frontier skipping)
};

namespace rwr {

// Node count from which a graph counts as "beyond the L2s" (its rank vector no longer fits them): two-phase row order of
// the single-seed SpMV, one more frontier iteration, 32-seed tiles.
// RWR_BIG_N overrides the default of 2 M so that small test graphs can reach the same code paths.
inline int32_t spmv_big_n()
{
    static const int32_t v = [] { const char *e = getenv("RWR_BIG_N"); return e ? (int32_t)atol(e) : (int32_t)2000000; }();
    return v;
}

int32_t graph_build(rwr_graph *g, const int64_t *node_id, const uint8_t *node_type, const int64_t *rowptr,
                    const int32_t *dst, const uint8_t *etype, const double *w);

int32_t graph_update_links(rwr_graph *g, int64_t count, const int64_t *idx, const uint8_t *etype, const double *w);
// value-free graphs: materialise in_w (= w_src[in_src]) for an entry point that runs the weighted kernels
int32_t ensure_in_w(rwr_graph *g);
// many ego-network-sized graphs at once (rwr_eval_graphs): one build launch, one call launch, one evaluation launch
bool graph_fits_small_build(int32_t n, int64_t m);
void multi_pins_acquire();    // a set of pinned buffers for this thread's rwr_eval_graphs call (from a process-wide pool) ...
void multi_pins_release();    // ... and back
int32_t multi_pinned(int slot, size_t bytes, void **out);   // buffer number `slot` (0..5) of the acquired set, at least `bytes` long
int32_t graphs_build_multi(rwr_graph **gs, int32_t count, const rwr_graph_desc *descs, const int32_t *caller_index, hipStream_t s);
int32_t recommend_small_multi(rwr_graph **gs, const int32_t *seeds, int32_t count, double d, int32_t n_iter, hipStream_t s,
                              rwr::DevBuf<uint8_t> &args_keep);
int32_t eval_ranked_multi(rwr_graph **gs, int32_t count, const int64_t *test_ptr_host, const int64_t *test_sorted_host,
                          int64_t *n_hits, double *sum_precision, int64_t *list_len, hipStream_t s);

// runs the power iteration for K seeds and leaves, per seed, the ranked list
// (mode 0: top-k into host arrays; mode 1: full rank vector of one seed)
int32_t recommend_batch(rwr_graph *g, const int32_t *seeds, int32_t K, double d, int32_t n_iter, int32_t top_n,
                        int64_t *ids, double *scores, int32_t *counts, int64_t row_stride);
int32_t eval_ranked(rwr_graph *g, int32_t cnt, const int64_t *test_sorted_host, int64_t n_test, int64_t *n_hits,
                    double *sum_precision);
int32_t eval_ranked_batch(rwr_graph *g, int32_t K, int64_t row_stride, const int64_t *test_ptr_host,
                          const int64_t *test_sorted_host, int64_t *n_hits, double *sum_precision);
int32_t part_begin(rwr_graph *g, int32_t lo, int32_t hi, const int32_t *seeds, int32_t K, double d, double *x,
                   int32_t *G_out);
int32_t part_local_step(rwr_graph *g, const double *x, double *y, double *r);
int32_t part_step(rwr_graph *g, const double *x, double *y, hipStream_t stream);
int32_t part_finish_step(rwr_graph *g, double *y, const double *r);
int32_t part_rank(rwr_graph *g, double *x, int32_t top_n, int64_t *ids, double *scores, int32_t *counts);
int32_t model_run(rwr_graph *g, int32_t seed, double d, int32_t run_mode, double value, double *rank_out,
                  int64_t *iters_out);
int32_t model_deliver(rwr_graph *g, int32_t seed, double d, const double *rank_in, double *next_out);
// spmv.hip: single-seed SpMV (list-order sums, rows binned by in-degree)
// (zin != nullptr: value-free form -- gathers zin, reads no per-entry value; zout (may be nullptr) receives the next z)
// (hub_scan: every addend is known to be >= 0 and finite -- weights, ranks and 1-d -- so that rows of >= 2048 in-links may
//  be summed by the exact parallel reduction of pf.h instead of one add at a time)
void launch_spmv_exact(rwr_graph *g, const double *X, double *Y, const int32_t *seeds, double c1, int skip,
                       const uint32_t *act, uint32_t *nz_out, hipStream_t s, const double *zin = nullptr,
                       double *zout = nullptr, bool hub_scan = false);
// sweep.hip: single-seed SpMV of value-free graphs, z staged through LDS source block by source block (bitwise the same sums)
int32_t sweep_prepare(rwr_graph *g);
bool sweep_ready(const rwr_graph *g);
void launch_sweep(rwr_graph *g, const double *zin, double *Y, double *zout, const int32_t *seeds, int skip, double c1,
                  hipStream_t s);
// small.hip: ego-network-sized graphs, one single-seed Recommendation as ONE kernel launch (bitwise the EXACT path's result)
int32_t small_pin_ensure(rwr_graph *g);
void *small_pin_scratch(rwr_graph *g);   // the score half of the pinned result buffer (SM_MAX_ITEMS 8-byte words), as scratch
int64_t small_pin_words();
bool small_path_ok(const rwr_graph *g);
bool small_path_seed_ok(const rwr_graph *g, int32_t seed);
const int64_t *small_pin_ids(const rwr_graph *g);      // the list of the last recommend_small call, in pinned host memory
const double *small_pin_scores(const rwr_graph *g);
int32_t recommend_small(rwr_graph *g, int32_t seed, double d, int32_t n_iter, int32_t top_n, int64_t *ids, double *scores,
                        int32_t *count);
// chain_scan.hip: the exact seed-row chain as a parallel binade scan
int32_t chain_scan_prepare(rwr_graph *g, int G, int tg, const int32_t *d_seeds, hipStream_t s);
int32_t chain_scan_step(rwr_graph *g, int G, int tg, const double *X, double *Y, const int32_t *d_seeds,
                        const int64_t *d_evoff, double c1, uint32_t *nz_out, hipStream_t s, const double *zterms = nullptr,
                        double *zout = nullptr);
bool chain_scan_self_contained(int G);   // G == 1: no k_seed_terms / k_seed_z launches around the step
int32_t chain_scan_collect(rwr_graph *g, hipStream_t s);
int32_t chain_scan_sum(rwr_graph *g, const double *D, double *out, hipStream_t s);   // exact sequential sum of n addends >= 0


}  // namespace rwr
