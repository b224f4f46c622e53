// Ranking stage of Recommender.Recommendation (Recommender.cs:27-38, 42-51):
// candidates = ITEM nodes not LIKEd by the seed, ordered by (score desc, id desc).
//
// The ITEM rows are kept pre-sorted by id descending (graph build), so a STABLE sort on
// the score alone yields the reference's total order (ids are unique among items:
// DataLoader.cs:51-58, so List.Sort's instability cannot show).  Scores are mapped to
// order-preserving 64-bit keys; non-candidates (excluded by k_exclude: score -1, or a
// padding lane) get the largest key and sort to the end.
#include "engine.h"

namespace rwr {

template <int G>
__global__ __launch_bounds__(256) void k_rank_keys(int32_t n_items, const int32_t *__restrict__ item_order,
                                                   const double *__restrict__ X,
                                                   const int32_t *__restrict__ seeds,
                                                   uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    // thread = (item slot q, seed k); consecutive threads walk k first: X reads coalesce
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t q = t / G;
    const int k = (int)(t % G);
    if (q >= n_items) return;
    const int32_t row = item_order[q];
    const double s = X[(size_t)row * G + k];
    const bool cand = (seeds[k] >= 0) && (s >= 0.0);
    keys[(size_t)k * n_items + q] = cand ? ~f64_orderable(s) : ~0ull;
    vals[(size_t)k * n_items + q] = (uint32_t)q;
}

// counts[k] = number of candidates = first position holding the sentinel key
__global__ void k_rank_count(int32_t n_items, int G, const uint64_t *__restrict__ keys, int32_t *__restrict__ counts)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G) return;
    const uint64_t *kk = keys + (size_t)k * n_items;
    int32_t lo = 0, hi = n_items;
    while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (kk[mid] == ~0ull) hi = mid;
        else lo = mid + 1;
    }
    counts[k] = lo;
}

__global__ __launch_bounds__(256) void k_rank_emit(int32_t n_items, int G, int32_t n_real, int32_t top_n,
                                                   const uint32_t *__restrict__ vals,
                                                   const int32_t *__restrict__ cand_counts,
                                                   const int32_t *__restrict__ item_order,
                                                   const int64_t *__restrict__ node_id,
                                                   const double *__restrict__ X, int64_t *__restrict__ out_id,
                                                   double *__restrict__ out_score, int32_t *__restrict__ out_counts)
{
    const int k = blockIdx.y;
    if (k >= n_real) return;
    int32_t cnt = cand_counts[k];
    if (cnt > top_n) cnt = top_n;
    const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) out_counts[k] = cnt;
    if (q >= cnt) return;
    const int32_t row = item_order[vals[(size_t)k * n_items + q]];
    out_id[(size_t)k * top_n + q] = node_id[row];
    out_score[(size_t)k * top_n + q] = X[(size_t)row * G + k];
}

#define RWR_DISPATCH_G(G, CALL)                          \
    switch (G) {                                         \
        case 1: { constexpr int GG = 1; CALL; } break;   \
        case 2: { constexpr int GG = 2; CALL; } break;   \
        case 4: { constexpr int GG = 4; CALL; } break;   \
        case 8: { constexpr int GG = 8; CALL; } break;   \
        case 16: { constexpr int GG = 16; CALL; } break; \
        case 32: { constexpr int GG = 32; CALL; } break; \
        default: { constexpr int GG = 64; CALL; } break; \
    }

int32_t rank_tile(rwr_graph *g, int G, int tile_in_group, int64_t first_seed_slot, int32_t n_real, int32_t top_n,
                  const double *X, const int32_t *d_seeds_tile, hipStream_t s)
{
    (void)tile_in_group;
    const int32_t m = g->n_items;
    if (m == 0 || n_real <= 0) return RWR_OK;
    const size_t tot = (size_t)G * m;
    RWR_TRY(g->keys.ensure(tot));
    RWR_TRY(g->keys_alt.ensure(tot));
    RWR_TRY(g->vals.ensure(tot));
    RWR_TRY(g->vals_alt.ensure(tot));
    RWR_TRY(g->sort_temp.ensure(radix_sort_temp_bytes((size_t)m, G) + (size_t)G * sizeof(int32_t)));
    RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_rank_keys<GG>, dim3(cdiv(tot, 256)), dim3(256), 0, s, m,
                                         g->item_order.p, X, d_seeds_tile, g->keys.p, g->vals.p));
    RWR_HIP(hipGetLastError());
    bool alt = false;
    RWR_TRY(radix_sort_pairs<uint64_t>(g->keys.p, g->keys_alt.p, g->vals.p, g->vals_alt.p, (size_t)m, G, 64,
                                       g->sort_temp.p, s, &alt));
    const uint64_t *ks = alt ? g->keys_alt.p : g->keys.p;
    const uint32_t *vs = alt ? g->vals_alt.p : g->vals.p;
    int32_t *cand = (int32_t *)(g->sort_temp.p + radix_sort_temp_bytes((size_t)m, G));
    hipLaunchKernelGGL(k_rank_count, dim3(1), dim3(64), 0, s, m, G, ks, cand);
    int32_t width = top_n < m ? top_n : m;
    hipLaunchKernelGGL(k_rank_emit, dim3(cdiv((size_t)width, 256), n_real), dim3(256), 0, s, m, G, n_real, top_n, vs,
                       cand, g->item_order.p, g->node_id.p, X, g->d_out_id.p + (size_t)first_seed_slot * top_n,
                       g->d_out_score.p + (size_t)first_seed_slot * top_n, g->d_counts.p + first_seed_slot);
    RWR_HIP(hipGetLastError());
    return RWR_OK;
}

}  // namespace rwr
