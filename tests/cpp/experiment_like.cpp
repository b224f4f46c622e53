// The caller pattern of TweetRecommender/Experiment.cs:104-128 written against the C++ mirror of the reference's
// classes: build nodes/edges, (optionally) relabel a link UNDEFINED, Graph + buildGraph, Recommender.Recommendation,
// walk the list for hits.  Checks the two hand-derived KATs of SURVEY.md section 8c.  Exit code 0 = all good.
#include <cmath>
#include <cstdio>
#include <set>

#include "../../include/recommenders/rwr_based.hpp"

using namespace Recommenders::RWRBased;

#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } \
    } while (0)

int main()
{
    if (rwr_device_count() < 1) { std::fprintf(stderr, "no gfx950 device\n"); return 2; }
    {   // KAT-1: bipartite path, no dangling node
        std::map<int, Node> nodes{{0, Node(10, NodeType::USER)}, {1, Node(11, NodeType::ITEM)},
                                  {2, Node(12, NodeType::USER)}, {3, Node(13, NodeType::ITEM)}};
        std::map<int, std::vector<ForwardLink>> edges;
        auto like = [&](int a, int b) { edges[a].emplace_back(b, EdgeType::LIKE, 1.0); edges[b].emplace_back(a, EdgeType::LIKE, 1.0); };
        like(0, 1); like(1, 2); like(2, 3);
        Graph graph(nodes, edges);
        graph.buildGraph();
        auto norm = graph.graph();
        EXPECT(norm[1]->size() == 2 && (*norm[1])[0].weight == 0.5);
        Model model(graph, 0.5, 0);
        model.run(3);
        EXPECT(model.rank[0] == 2.25 && model.rank[1] == 1.375 && model.rank[2] == 0.25 && model.rank[3] == 0.125);
        Model stepwise(graph, 0.5, 0);                           // the same three iterations, driven step by step
        stepwise.deliverRanks();
        EXPECT(stepwise.nextRank[0] == 2.0 && stepwise.nextRank[1] == 2.0 && !stepwise.checkConvergence(3.9) && stepwise.checkConvergence(4.1));
        stepwise.updateRanks();
        stepwise.run(2);                                         // continues from the advanced state
        EXPECT(stepwise.rank == model.rank);
        Recommender recommender(graph);
        auto recommendation = recommender.Recommendation(0, 0.5f, 3);
        EXPECT(recommendation.size() == 1 && recommendation[0].first == 13 && recommendation[0].second == 0.125);
        std::set<int64_t> testSet{13};                           // Experiment.cs:121-128
        int nHits = 0;
        double sumPrecision = 0;
        for (size_t i = 0; i < recommendation.size(); i++)
            if (testSet.count(recommendation[i].first)) { nHits += 1; sumPrecision += (double)nHits / (i + 1); }
        EXPECT(nHits == 1 && sumPrecision == 1.0);
        bool threw = false;
        try { recommender.Recommendation(7, 0.5f, 3); } catch (const std::out_of_range &) { threw = true; }
        EXPECT(threw);                                           // KeyNotFoundException analogue
    }
    {   // KAT-2: dangling node; a link relabelled UNDEFINED stays in the raw list but leaves the walk (Experiment.cs:96)
        std::map<int, Node> nodes{{0, Node(7, NodeType::USER)}, {1, Node(9, NodeType::ITEM)}};
        std::map<int, std::vector<ForwardLink>> edges;
        edges[0].emplace_back(1, EdgeType::LIKE, 1.0);
        edges[1].emplace_back(0, EdgeType::FRIENDSHIP, 1.0);
        edges[1][0].type = EdgeType::UNDEFINED;
        Graph graph(nodes, edges);
        graph.buildGraph();
        EXPECT(graph.graph()[1] == nullptr);
        Model model(graph, 0.5, 0);
        model.run(3);
        EXPECT(model.rank[0] == 1.25 && model.rank[1] == 0.75);
        EXPECT(Recommender(graph).Recommendation(0, 0.5f, 3).empty());
    }
    {   // buildGraph() again after an in-place relabel (Experiment.cs:84-101): the mirror sends only the changed link;
        // KAT-1's path graph with the 2-3 like cut off becomes  0 <-> 1 <-> 2,  node 3 dangling
        std::map<int, Node> nodes{{0, Node(10, NodeType::USER)}, {1, Node(11, NodeType::ITEM)},
                                  {2, Node(12, NodeType::USER)}, {3, Node(13, NodeType::ITEM)}};
        std::map<int, std::vector<ForwardLink>> edges;
        auto like = [&](int a, int b) { edges[a].emplace_back(b, EdgeType::LIKE, 1.0); edges[b].emplace_back(a, EdgeType::LIKE, 1.0); };
        like(0, 1); like(1, 2); like(2, 3);
        Graph graph(nodes, edges);
        graph.buildGraph();
        rwr_graph *before = graph.handle();
        for (auto &l : graph.edges[2]) if (l.targetNode == 3) l.type = EdgeType::UNDEFINED;
        for (auto &l : graph.edges[3]) if (l.targetNode == 2) l.type = EdgeType::UNDEFINED;
        graph.buildGraph();
        EXPECT(graph.handle() == before);                        // patched in place, not re-created
        auto norm = graph.graph();
        EXPECT(norm[3] == nullptr && norm[2]->size() == 1 && (*norm[2])[0].weight == 1.0);
        Model model(graph, 0.5, 0);
        model.run(2);                                            // [4,0,0,0] -> [2,2,0,0] -> [2.5,1,0.5,0]
        EXPECT(model.rank[0] == 2.5 && model.rank[1] == 1.0 && model.rank[2] == 0.5 && model.rank[3] == 0.0);
    }
    std::puts("experiment_like: ok");
    return 0;
}
