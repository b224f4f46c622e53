// Drop-in replacement of Recommenders/RWRBased/Recommender.cs: same enums (same member order = same integer
// values, they cross the C-ABI as bytes) and the same two Recommendation overloads.
using System.Collections.Generic;

namespace Recommenders.RWRBased {
    public enum NodeType { UNDEFINED, USER, ITEM, ETC }
    public enum EdgeType { UNDEFINED, LIKE, FRIENDSHIP, FOLLOW, MENTION, AUTHORSHIP, PURCHASE, ETC }

    public class Recommender {
        private Graph graph;
        public Recommender(Graph graph) { this.graph = graph; }

        public List<KeyValuePair<long, double>> Recommendation(int idxTargetUser, float dampingFactor, int nIteration) {
            return Recommendation(idxTargetUser, dampingFactor, nIteration, 0);
        }

        public List<KeyValuePair<long, double>> Recommendation(int idxTargetUser, float dampingFactor, int nIteration, int topN) {
            // the reference reads graph.edges[idxTargetUser] and throws KeyNotFoundException when absent
            if (!graph.edges.ContainsKey(idxTargetUser)) throw new KeyNotFoundException();
            long count = graph.size();
            var ids = new long[count]; var scores = new double[count];
            Native.Check(Native.rwr_recommend(graph.handle, idxTargetUser, dampingFactor, nIteration, topN, ids, scores, ref count));
            var result = new List<KeyValuePair<long, double>>((int)count);
            for (long i = 0; i < count; i++) result.Add(new KeyValuePair<long, double>(ids[i], scores[i]));
            return result;
        }

        // addition: many seeds in one call (results identical to calling Recommendation per seed)
        public List<KeyValuePair<long, double>>[] RecommendationBatch(int[] seeds, float dampingFactor, int nIteration, int topN) {
            if (topN < 1) throw new System.ArgumentOutOfRangeException("topN", "RecommendationBatch needs topN >= 1 (rwr_recommend_batch)");
            int K = seeds.Length;
            var ids = new long[(long)K * topN]; var scores = new double[(long)K * topN]; var counts = new int[K];
            Native.Check(Native.rwr_recommend_batch(graph.handle, seeds, K, dampingFactor, nIteration, topN, ids, scores, counts));
            var all = new List<KeyValuePair<long, double>>[K];
            for (int k = 0; k < K; k++) {
                all[k] = new List<KeyValuePair<long, double>>(counts[k]);
                for (int q = 0; q < counts[k]; q++) all[k].Add(new KeyValuePair<long, double>(ids[(long)k * topN + q], scores[(long)k * topN + q]));
            }
            return all;
        }
    }
}
