// Drop-in replacement of Recommenders/RWRBased/Model.cs: same public fields and methods; run*() execute on
// the GPU through rwr_model_run and leave the result in `rank` exactly as the reference does.
namespace Recommenders.RWRBased {
    public class Model {
        public Graph graph;
        public double[] rank;
        public double[] nextRank;
        public int nNodes;
        public double dampingFactor;
        public double[] restart;
        int seed = -1;

        public Model(Graph graph, double dampingFactor) {
            this.graph = graph; nNodes = graph.size(); this.dampingFactor = dampingFactor;
            rank = new double[nNodes]; nextRank = new double[nNodes]; restart = new double[nNodes];
            for (int i = 0; i < nNodes; i++) { rank[i] = 1d; restart[i] = 1d / nNodes; }
        }

        public Model(Graph graph, double dampingFactor, int targetNode) {
            this.graph = graph; nNodes = graph.size(); this.dampingFactor = dampingFactor; seed = targetNode;
            rank = new double[nNodes]; nextRank = new double[nNodes]; restart = new double[nNodes];
            for (int i = 0; i < nNodes; i++) { rank[i] = (i == targetNode) ? nNodes : 0; restart[i] = (i == targetNode) ? 1d : 0; }
        }

        void Run(int mode, double value) {
            long iters;
            Native.Check(Native.rwr_model_run(graph.handle, seed, dampingFactor, mode, value, rank, out iters));
            for (int i = 0; i < nNodes; i++) nextRank[i] = 0;
        }
        public void run() { Run(2, 0); }
        public void run(double threshold) { Run(1, threshold); }
        public void run(int nIterations) { Run(0, nIterations); }
        // single-step methods of the reference: one iteration == run(1)
        public void deliverRanks() { throw new System.NotSupportedException("use run(int): deliverRanks+updateRanks are fused on the GPU"); }
        public void updateRanks() { }
        public bool checkConvergence(double threshold) { throw new System.NotSupportedException("use run(double)"); }
    }
}
