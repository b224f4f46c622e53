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

        // addition: the harness's loop body (Experiment.cs:69-134) for MANY graphs in one call -- graphs[k].buildGraph(),
        // Recommendation(seeds[k], d, nIteration) and the hits / sum-of-precisions walk over testSets[k] -- without building the
        // Graph objects one by one.  hits[k] and sumPrecision[k] are exactly what the loop computes for graph k.
        public static void EvaluateGraphs(Graph[] graphs, int[] seeds, float dampingFactor, int nIteration, HashSet<long>[] testSets,
                                          out long[] hits, out double[] sumPrecision) {
            int K = graphs.Length;
            var descs = new RwrGraphDesc[K];
            var pins = new List<System.Runtime.InteropServices.GCHandle>();
            System.Func<object, System.IntPtr> pin = a => {
                var h = System.Runtime.InteropServices.GCHandle.Alloc(a, System.Runtime.InteropServices.GCHandleType.Pinned);
                pins.Add(h);
                return h.AddrOfPinnedObject();
            };
            var testPtr = new long[K + 1];
            var testIds = new List<long>();
            try {
                for (int k = 0; k < K; k++) {
                    long[] nodeId; byte[] nodeType; long[] rowptr; int[] dst; byte[] etype; double[] w;
                    graphs[k].Flatten(out nodeId, out nodeType, out rowptr, out dst, out etype, out w);
                    descs[k] = new RwrGraphDesc { n_nodes = nodeId.Length, reserved0 = 0, node_id = pin(nodeId), node_type = pin(nodeType),
                                                  rowptr = pin(rowptr), dst = pin(dst), etype = pin(etype), w = pin(w) };
                    testIds.AddRange(testSets[k]);
                    testPtr[k + 1] = testIds.Count;
                }
                hits = new long[K]; sumPrecision = new double[K];
                var opts = new RwrOpts { struct_size = System.Runtime.InteropServices.Marshal.SizeOf(typeof(RwrOpts)), device = -1, mode = -1 };
                Native.Check(Native.rwr_eval_graphs(K, descs, seeds, dampingFactor, nIteration, testPtr, testIds.ToArray(), ref opts, hits,
                                                    sumPrecision, null));
            } finally {
                foreach (var h in pins) h.Free();
            }
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
