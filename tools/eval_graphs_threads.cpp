// Host-thread scaling of rwr_eval_graphs without an interpreter in the way: T threads, each handing its own G synthetic
// ego-network-sized graphs to one call, R times (third argument 1: the same graphs one per call -- create, recommend_eval, destroy).  Build and run on the GPU box (tools/eval_graphs_threads.sh):
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/eval_graphs_threads.cpp -o /tmp/egt recommendersystems_amd/librwr.so -Wl,-rpath,$PWD/recommendersystems_amd
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "rwr.h"

struct HostGraph {
    std::vector<int64_t> node_id, rowptr;
    std::vector<uint8_t> node_type, etype;
    std::vector<int32_t> dst;
    std::vector<double> w;
    std::vector<int64_t> test;
};

static HostGraph make_graph(unsigned seed)
{
    std::mt19937 rng(seed);
    const int users = 100 + (int)(rng() % 200), items = 600 + (int)(rng() % 1500), n = users + items;
    const int likes = 1500 + (int)(rng() % 4500);
    std::vector<std::vector<std::pair<int, int>>> adj((size_t)n);     // (target, type)
    for (int q = 0; q < likes; ++q) {
        const int u = (int)(rng() % users), it = users + (int)(rng() % items);
        adj[u].push_back({it, RWR_EDGE_LIKE});
        adj[it].push_back({u, RWR_EDGE_LIKE});
    }
    for (int q = 0; q < 3 * users; ++q) {
        const int a = (int)(rng() % users), b = (int)(rng() % users);
        if (a != b) adj[a].push_back({b, RWR_EDGE_FRIENDSHIP});
    }
    HostGraph g;
    g.node_id.resize(n); g.node_type.resize(n); g.rowptr.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) { g.node_id[i] = 1000 + 7 * (int64_t)i; g.node_type[i] = i < users ? RWR_NODE_USER : RWR_NODE_ITEM; }
    for (int i = 0; i < n; ++i) {
        g.rowptr[i + 1] = g.rowptr[i] + (int64_t)adj[i].size();
        for (auto &e : adj[i]) { g.dst.push_back(e.first); g.etype.push_back((uint8_t)e.second); g.w.push_back(1.0); }
    }
    for (int q = 0; q < 30; ++q) g.test.push_back(g.node_id[users + (int)(rng() % items)]);
    return g;
}

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 19, R = argc > 2 ? atoi(argv[2]) : 40;
    const bool single = argc > 3 && atoi(argv[3]) != 0;   // 1: one graph per call (create + recommend_eval + destroy), the unmodified harness's way
    const int maxT = 16;
    std::vector<HostGraph> graphs;
    for (int i = 0; i < maxT * G; ++i) graphs.push_back(make_graph(1000u + (unsigned)i));
    for (int T : {1, 2, 4, 8, 10, 16}) {
        auto work = [&](int t, int reps) {
            std::vector<rwr_graph_desc> d((size_t)G);
            std::vector<int32_t> seeds((size_t)G, 0);
            std::vector<int64_t> tp((size_t)G + 1, 0), tids, hits((size_t)G), len((size_t)G);
            std::vector<double> sp((size_t)G);
            for (int q = 0; q < G; ++q) {
                HostGraph &h = graphs[(size_t)t * G + q];
                d[q] = rwr_graph_desc{(int32_t)h.node_id.size(), 0, h.node_id.data(), h.node_type.data(), h.rowptr.data(), h.dst.data(),
                                      h.etype.data(), h.w.data()};
                tids.insert(tids.end(), h.test.begin(), h.test.end());
                tp[(size_t)q + 1] = (int64_t)tids.size();
            }
            for (int r = 0; r < reps; ++r) {
                if (single) {
                    for (int q = 0; q < G; ++q) {
                        rwr_graph *h = nullptr;
                        int32_t rc = rwr_graph_create(d[q].n_nodes, d[q].node_id, d[q].node_type, d[q].rowptr, d[q].dst, d[q].etype, d[q].w, nullptr, &h);
                        if (rc == RWR_OK)
                            rc = rwr_recommend_eval(h, 0, 0.15f, 10, tids.data() + tp[q], tp[(size_t)q + 1] - tp[q], &hits[q], &sp[q], &len[q]);
                        if (rc != RWR_OK) { fprintf(stderr, "single-graph calls: %s\n", rwr_last_error()); exit(1); }
                        rwr_graph_destroy(h);
                    }
                    continue;
                }
                const int32_t rc = rwr_eval_graphs(G, d.data(), seeds.data(), 0.15f, 10, tp.data(), tids.data(), nullptr, hits.data(),
                                                   sp.data(), len.data());
                if (rc != RWR_OK) { fprintf(stderr, "rwr_eval_graphs: %s\n", rwr_last_error()); exit(1); }
            }
        };
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(work, t, 2);      // warm-up: pools, pinned sets, streams
            for (auto &x : th) x.join();
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t, R);
        for (auto &x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"threads\": %d, \"graphs_per_call\": %d, \"graphs_per_s\": %.0f, \"ms_per_call\": %.3f}\n", T, single ? 1 : G,
               (double)T * G * R / dt, 1e3 * dt / R / (single ? G : 1));
        fflush(stdout);
    }
    return 0;
}
