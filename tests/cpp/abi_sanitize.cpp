// Host-side sanitizer run of the C-ABI (ASan + UBSan on the HOST code of librwr; the device code is not instrumented):
// a loader-shaped random graph driven through every entry point, including the error paths.
//   hipcc ... -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined  (see tools/asan_run.sh)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

#include <unistd.h>

#include "../../include/rwr.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int32_t s_ = (call);                                                                     \
        if (s_ != RWR_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, s_, rwr_last_error()); return 1; } \
    } while (0)
#define EXPECT_FAIL(call, code)                                                                  \
    do {                                                                                         \
        int32_t s_ = (call);                                                                     \
        if (s_ != (code)) { std::fprintf(stderr, "%s -> %d, expected %d\n", #call, s_, (int)(code)); return 1; } \
    } while (0)

int main()
{
    if (rwr_device_count() < 1) { std::fprintf(stderr, "no gfx950 device\n"); return 2; }
    std::mt19937_64 rng(12345);
    for (int round = 0; round < 6; ++round) {
        const int U = 50 + (int)(rng() % 3000), I = 20 + (int)(rng() % 6000), n = U + I;
        std::vector<std::vector<int32_t>> out(n);
        const int likes = (int)(rng() % (6u * n));
        for (int e = 0; e < likes; ++e) {
            const int u = (int)((rng() % U) * (rng() % U) / U), v = U + (int)((rng() % I) * (rng() % I) / I);
            out[u].push_back(v);
            out[v].push_back(u);
        }
        std::vector<int64_t> id(n), rowptr(n + 1, 0);
        std::vector<uint8_t> type(n);
        for (int i = 0; i < n; ++i) { id[i] = (int64_t)i * 3 - 7; type[i] = i < U ? RWR_NODE_USER : RWR_NODE_ITEM; rowptr[i + 1] = rowptr[i] + (int64_t)out[i].size(); }
        const int64_t m = rowptr[n];
        std::vector<int32_t> dst((size_t)m);
        std::vector<uint8_t> et((size_t)m);
        std::vector<double> w((size_t)m);
        int64_t p = 0;
        for (int i = 0; i < n; ++i)
            for (int32_t t : out[i]) { dst[p] = t; et[p] = (rng() % 20 == 0) ? RWR_EDGE_UNDEFINED : RWR_EDGE_LIKE; w[p] = (round & 1) ? 1.0 : 1.0 + (double)(rng() % 4); ++p; }   // (odd rounds: unit weights = the value-free path)
        rwr_opts o{};
        o.struct_size = sizeof(o);
        o.device = -1;
        o.mode = RWR_MODE_EXACT;
        o.profile = round & 1;
        o.seed_row_kernel = round % 3;
        rwr_graph *g = nullptr;
        CHECK(rwr_graph_create(n, id.data(), type.data(), rowptr.data(), dst.data(), et.data(), w.data(), &o, &g));
        int32_t gn = 0; int64_t raw = 0, nnz = 0;
        CHECK(rwr_graph_size(g, &gn, &raw, &nnz));
        std::vector<double> wn((size_t)(m ? m : 1)); std::vector<uint8_t> dg(n);
        CHECK(rwr_graph_get_normalized(g, wn.data(), dg.data()));
        // single seed: top-n, full list, too-small buffer
        std::vector<int64_t> ids(n); std::vector<double> sc(n);
        int64_t cnt = n;
        CHECK(rwr_recommend(g, 0, 0.15f, 6, 10, ids.data(), sc.data(), &cnt));
        cnt = n;
        CHECK(rwr_recommend(g, U / 2, 0.15f, 6, 0, ids.data(), sc.data(), &cnt));
        const int64_t full = cnt;
        if (full > 1) { int64_t small = 1; EXPECT_FAIL(rwr_recommend(g, U / 2, 0.15f, 6, 0, ids.data(), sc.data(), &small), RWR_E_CAPACITY); }
        EXPECT_FAIL(rwr_recommend(g, n, 0.15f, 6, 0, ids.data(), sc.data(), &cnt), RWR_E_RANGE);
        cnt = n;
        EXPECT_FAIL(rwr_recommend(g, 0, 1.5f, 6, 0, ids.data(), sc.data(), &cnt), RWR_E_UNSUPPORTED);   // damping factor outside [0, 1]
        // batch
        const int K = 1 + (int)(rng() % 70), top = 1 + (int)(rng() % 1200);
        std::vector<int32_t> seeds(K), counts(K);
        for (auto &s : seeds) s = (int32_t)(rng() % U);
        std::vector<int64_t> bid((size_t)K * top); std::vector<double> bsc((size_t)K * top);
        CHECK(rwr_recommend_batch(g, seeds.data(), K, 0.15f, 5, top, bid.data(), bsc.data(), counts.data()));
        // evaluation, model run (iterations / threshold), single deliver step
        int64_t hits = 0, ll = 0; double sp = 0;
        std::vector<int64_t> test{ids[0], id[n - 1], 123456789};
        CHECK(rwr_recommend_eval(g, U / 2, 0.15f, 6, test.data(), (int64_t)test.size(), &hits, &sp, &ll));
        std::vector<double> rank(n), next(n);
        int64_t iters = 0;
        {   // K seeds with K test sets (CSR), then the same graph three times through the many-graphs entry point
            const int32_t es[3] = {0, U / 2, U - 1};
            const int64_t tp[4] = {0, (int64_t)test.size() / 2, (int64_t)test.size() / 2, (int64_t)test.size()};
            int64_t h3[3], l3[3]; double s3[3];
            CHECK(rwr_recommend_eval_batch(g, es, 3, 0.15f, 6, tp, test.data(), h3, s3, l3));
            rwr_graph_desc dd[3];
            for (auto &D : dd) D = rwr_graph_desc{n, 0, id.data(), type.data(), rowptr.data(), dst.data(), et.data(), w.data()};
            int64_t h4[3], l4[3]; double s4[3];
            CHECK(rwr_eval_graphs(3, dd, es, 0.15f, 6, tp, test.data(), &o, h4, s4, l4));
            for (int q = 0; q < 3; ++q)
                if (h3[q] != h4[q] || l3[q] != l4[q] || std::memcmp(&s3[q], &s4[q], 8) != 0) { std::fprintf(stderr, "eval_graphs != eval_batch (%d)\n", q); return 1; }
            const int32_t bad_seed[3] = {0, n, 0};
            EXPECT_FAIL(rwr_eval_graphs(3, dd, bad_seed, 0.15f, 6, tp, test.data(), &o, h4, s4, l4), RWR_E_RANGE);
        }
        CHECK(rwr_model_run(g, 1, 0.15, RWR_RUN_ITERATIONS, 4, rank.data(), &iters));
        CHECK(rwr_model_run(g, 1, 0.15, RWR_RUN_THRESHOLD, 1e-3, rank.data(), &iters));
        CHECK(rwr_model_run(g, -1, 0.15, RWR_RUN_ITERATIONS, 3, rank.data(), &iters));
        CHECK(rwr_model_deliver(g, 1, 0.15, rank.data(), next.data()));
        CHECK(rwr_model_deliver(g, -1, 0.15, rank.data(), next.data()));
        EXPECT_FAIL(rwr_model_deliver(g, 1, 0.15, rank.data(), rank.data()), RWR_E_INVALID);
        // incremental rebuild + error path
        if (m > 0) {
            std::vector<int64_t> idx; std::vector<uint8_t> nt; std::vector<double> nw;
            for (int64_t q = 0; q < m; q += 1 + (int64_t)(rng() % 50)) { idx.push_back(q); nt.push_back((uint8_t)(rng() % 3)); nw.push_back(2.0); }
            CHECK(rwr_graph_update_links(g, (int64_t)idx.size(), idx.data(), nt.data(), nw.data()));
            CHECK(rwr_graph_update_links(g, (int64_t)idx.size(), idx.data(), nullptr, nw.data()));
            int64_t badi = m;
            EXPECT_FAIL(rwr_graph_update_links(g, 1, &badi, nt.data(), nullptr), RWR_E_RANGE);
            CHECK(rwr_recommend_batch(g, seeds.data(), K, 0.15f, 5, top, bid.data(), bsc.data(), counts.data()));
        }
        rwr_stats st{};
        st.struct_size = sizeof(st);
        CHECK(rwr_get_stats(g, &st));
        CHECK(rwr_reset_stats(g));
        CHECK(rwr_graph_destroy(g));
    }
    std::puts("abi_sanitize: ok");
    std::fflush(stdout);
    _exit(0);   // (skip the HSA runtime's exit-time teardown, which trips ROCm's ASan device allocator: not this library's code)
}
