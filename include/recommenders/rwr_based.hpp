// C++ host mirror of the reference's public surface `namespace Recommenders.RWRBased`
// (Recommenders/RWRBased/{Graph,Model,Recommender}.cs) over the C-ABI of include/rwr.h.
//
// The reference is C# and no C# toolchain exists in the build image, so this header (and the Python mirror
// recommendersystems_amd/rwr_based.py) are the host sides that are actually compiled and exercised; the C# shim
// under csharp/ binds the very same entry points.  Same names, argument meaning and error behaviour as the C#:
//   KeyNotFoundException / ArgumentOutOfRangeException  ->  std::out_of_range
//   any other failure                                   ->  std::runtime_error (with rwr_last_error())
// Header-only; link with librwr.so.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../rwr.h"

namespace Recommenders {
namespace RWRBased {

enum class NodeType : uint8_t { UNDEFINED, USER, ITEM, ETC };                                        // Recommender.cs:4
enum class EdgeType : uint8_t { UNDEFINED, LIKE, FRIENDSHIP, FOLLOW, MENTION, AUTHORSHIP, PURCHASE, ETC };  // :5

struct Node {                                                  // Graph.cs:4-17
    int64_t id = 0;
    NodeType type = NodeType::UNDEFINED;
    Node() = default;
    explicit Node(int64_t id_) : id(id_) {}
    Node(int64_t id_, NodeType t) : id(id_), type(t) {}
};

struct ForwardLink {                                           // Graph.cs:19-35 (public mutable fields)
    int targetNode = 0;
    EdgeType type = EdgeType::UNDEFINED;
    double weight = 0;
    ForwardLink() = default;
    ForwardLink(int t, double w) : targetNode(t), weight(w) {}
    ForwardLink(int t, EdgeType ty, double w) : targetNode(t), type(ty), weight(w) {}
};

inline void check(int32_t status)
{
    if (status == RWR_OK) return;
    const std::string msg = rwr_last_error();
    if (status == RWR_E_RANGE) throw std::out_of_range(msg);
    throw std::runtime_error("librwr status " + std::to_string(status) + ": " + msg);
}

class Graph {                                                  // Graph.cs:37-94
public:
    std::map<int, Node> nodes;                                 // Dictionary<int, Node>, keys 0..n-1
    std::map<int, std::vector<ForwardLink>> edges;             // Dictionary<int, List<ForwardLink>>

    Graph(std::map<int, Node> n, std::map<int, std::vector<ForwardLink>> e) : nodes(std::move(n)), edges(std::move(e)) {}
    ~Graph() { rwr_graph_destroy(h_); }
    Graph(const Graph &) = delete;
    Graph &operator=(const Graph &) = delete;

    void buildGraph()                                          // Graph.cs:51-88 -> rwr_graph_create
    {
        const int n = (int)nodes.size();
        std::vector<int64_t> id(n), rowptr(n + 1, 0);
        std::vector<uint8_t> type(n);
        for (int i = 0; i < n; ++i) {
            const Node &nd = nodes.at(i);
            id[i] = nd.id;
            type[i] = (uint8_t)nd.type;
            auto it = edges.find(i);
            rowptr[i + 1] = rowptr[i] + (it == edges.end() ? 0 : (int64_t)it->second.size());
        }
        dst_.assign((size_t)rowptr[n], 0);
        etype_.assign((size_t)rowptr[n], 0);
        std::vector<double> w((size_t)rowptr[n]);
        size_t e = 0;
        for (int i = 0; i < n; ++i) {
            auto it = edges.find(i);
            if (it == edges.end()) continue;
            for (const ForwardLink &l : it->second) { dst_[e] = l.targetNode; etype_[e] = (uint8_t)l.type; w[e] = l.weight; ++e; }
        }
        // buildGraph() again after the host changed link types / weights in place (Experiment.cs:84-101 style): same nodes,
        // list lengths and targets => only the changed links cross the boundary (rwr_graph_update_links)
        if (h_ && id == sent_id_ && type == sent_type_ && rowptr == rowptr_ && dst_ == sent_dst_) {
            std::vector<int64_t> idx;
            std::vector<uint8_t> nt;
            std::vector<double> nw;
            for (size_t p = 0; p < w.size(); ++p)
                if (etype_[p] != sent_etype_[p] || std::memcmp(&w[p], &sent_w_[p], sizeof(double)) != 0) {
                    idx.push_back((int64_t)p); nt.push_back(etype_[p]); nw.push_back(w[p]);
                }
            check(rwr_graph_update_links(h_, (int64_t)idx.size(), idx.data(), nt.data(), nw.data()));
        } else {
            rwr_graph_destroy(h_);
            h_ = nullptr;
            check(rwr_graph_create(n, id.data(), type.data(), rowptr.data(), dst_.data(), etype_.data(), w.data(), nullptr, &h_));
        }
        rowptr_ = rowptr;
        sent_id_ = id; sent_type_ = type; sent_dst_ = dst_; sent_etype_ = etype_; sent_w_ = w;
    }

    // the public field Graph.graph (Graph.cs:43): normalised explicit links per node; empty optional == null
    std::map<int, std::unique_ptr<std::vector<ForwardLink>>> graph()
    {
        const int n = (int)nodes.size();
        std::vector<double> wn(dst_.size() ? dst_.size() : 1);
        std::vector<uint8_t> dg(n);
        check(rwr_graph_get_normalized(handle(), wn.data(), dg.data()));
        std::map<int, std::unique_ptr<std::vector<ForwardLink>>> out;
        for (int i = 0; i < n; ++i) {
            if (dg[i]) { out[i] = nullptr; continue; }
            auto v = std::make_unique<std::vector<ForwardLink>>();
            for (int64_t p = rowptr_[i]; p < rowptr_[i + 1]; ++p)
                if (etype_[p] != 0) v->emplace_back(dst_[p], (EdgeType)etype_[p], wn[p]);
            out[i] = std::move(v);
        }
        return out;
    }

    int size() const { return (int)nodes.size(); }             // Graph.cs:91-93
    rwr_graph *handle() const
    {
        if (!h_) throw std::runtime_error("Graph.buildGraph() has not been called");
        return h_;
    }

private:
    rwr_graph *h_ = nullptr;
    std::vector<int64_t> rowptr_;
    std::vector<int32_t> dst_;
    std::vector<uint8_t> etype_;
    // what the device currently holds (for the incremental rebuild)
    std::vector<int64_t> sent_id_;
    std::vector<uint8_t> sent_type_, sent_etype_;
    std::vector<int32_t> sent_dst_;
    std::vector<double> sent_w_;
};

class Model {                                                  // Model.cs:5-116
public:
    Graph &graph;
    std::vector<double> rank, nextRank, restart;
    int nNodes;
    double dampingFactor;

    Model(Graph &g, double d) : graph(g), nNodes(g.size()), dampingFactor(d), seed_(-1)              // :14-31
    {
        rank.assign(nNodes, 1.0);
        nextRank.assign(nNodes, 0.0);
        restart.assign(nNodes, 1.0 / nNodes);
    }
    Model(Graph &g, double d, int targetNode) : graph(g), nNodes(g.size()), dampingFactor(d), seed_(targetNode)   // :33-50
    {
        rank.assign(nNodes, 0.0);
        nextRank.assign(nNodes, 0.0);
        restart.assign(nNodes, 0.0);
        if (targetNode >= 0 && targetNode < nNodes) { rank[targetNode] = nNodes; restart[targetNode] = 1.0; }
    }
    void run() { run_(RWR_RUN_DEFAULT_THRESHOLD, 0); }         // :52-55
    void run(double threshold) { run_(RWR_RUN_THRESHOLD, threshold); }   // :57-66
    void run(int nIterations) { run_(RWR_RUN_ITERATIONS, nIterations); } // :68-73
    int64_t iterations = 0;

    // the reference's public single steps
    void deliverRanks()                                        // :76-100 -> rwr_model_deliver
    {
        for (double v : nextRank)
            if (v != 0.0) throw std::logic_error("deliverRanks() on a non-zero nextRank: call updateRanks() first");
        check(rwr_model_deliver(graph.handle(), seed_, dampingFactor, rank.data(), nextRank.data()));
    }
    void updateRanks()                                         // :103-108
    {
        for (int i = 0; i < nNodes; ++i) { rank[i] = nextRank[i]; nextRank[i] = 0.0; }
    }
    bool checkConvergence(double threshold) const              // :110-115
    {
        double diff = 0.0;
        for (int i = 0; i < nNodes; ++i) diff += rank[i] > nextRank[i] ? rank[i] - nextRank[i] : nextRank[i] - rank[i];
        return diff < threshold;
    }

private:
    int seed_;
    bool ctor_state_() const
    {
        for (int i = 0; i < nNodes; ++i) {
            if (nextRank[i] != 0.0) return false;
            const double expect = seed_ < 0 ? 1.0 : (i == seed_ ? (double)nNodes : 0.0);
            if (rank[i] != expect) return false;
        }
        return true;
    }
    void run_(int mode, double value)
    {
        if (ctor_state_()) {                                   // the whole loop stays on the device
            check(rwr_model_run(graph.handle(), seed_, dampingFactor, mode, value, rank.data(), &iterations));
            nextRank.assign(nNodes, 0.0);
            return;
        }
        // an already advanced model: the reference's run() continues from the current rank (Model.cs:57-73)
        iterations = 0;
        if (mode == RWR_RUN_ITERATIONS) {
            for (int64_t k = 0; k < (int64_t)value; ++k) { deliverRanks(); updateRanks(); ++iterations; }
            return;
        }
        const double threshold = mode == RWR_RUN_DEFAULT_THRESHOLD ? (1 / 1.7976931348623157e308) * nNodes : value;
        for (;;) {
            deliverRanks();
            ++iterations;
            const bool done = checkConvergence(threshold);
            updateRanks();
            if (done) return;
        }
    }
};

class Recommender {                                            // Recommender.cs:7-52
public:
    explicit Recommender(Graph &g) : graph_(g) {}

    std::vector<std::pair<int64_t, double>> Recommendation(int idxTargetUser, float dampingFactor, int nIteration,
                                                           int topN = 0)                     // :14-40, :42-51
    {
        if (!graph_.edges.count(idxTargetUser)) throw std::out_of_range("KeyNotFound: graph.edges[idxTargetUser]");   // :21
        int64_t count = graph_.size();
        std::vector<int64_t> ids((size_t)count);
        std::vector<double> scores((size_t)count);
        check(rwr_recommend(graph_.handle(), idxTargetUser, dampingFactor, nIteration, topN, ids.data(), scores.data(),
                            &count));
        std::vector<std::pair<int64_t, double>> out((size_t)count);
        for (int64_t i = 0; i < count; ++i) out[i] = {ids[i], scores[i]};
        return out;
    }

private:
    Graph &graph_;
};

}  // namespace RWRBased
}  // namespace Recommenders
