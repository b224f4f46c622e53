// Drop-in replacement of Recommenders/RWRBased/Model.cs: same public fields and methods; run*() execute on
// the GPU through rwr_model_run and leave the result in `rank` exactly as the reference does; deliverRanks() is one
// propagation on the GPU (rwr_model_deliver), updateRanks()/checkConvergence() are the reference's array loops.
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

        // rank / nextRank are still what the constructor left => the whole loop can stay on the device
        bool CtorState() {
            for (int i = 0; i < nNodes; i++) {
                if (nextRank[i] != 0) return false;
                double expect = seed < 0 ? 1d : (i == seed ? nNodes : 0d);
                if (rank[i] != expect) return false;
            }
            return true;
        }

        void Run(int mode, double value) {
            CheckRestart();
            if (CtorState()) {
                long iters;
                Native.Check(Native.rwr_model_run(graph.handle, seed, dampingFactor, mode, value, rank, out iters));
                for (int i = 0; i < nNodes; i++) nextRank[i] = 0;
                return;
            }
            // an already advanced model: the reference's run() continues from the current rank (Model.cs:57-73)
            if (mode == 0) {
                for (int n = 0; n < (int)value; n++) { deliverRanks(); updateRanks(); }
                return;
            }
            double threshold = mode == 2 ? (1 / double.MaxValue) * nNodes : value;
            while (true) {
                deliverRanks();
                if (checkConvergence(threshold)) { updateRanks(); return; }
                updateRanks();
            }
        }
        public void run() { Run(2, 0); }
        public void run(double threshold) { Run(1, threshold); }
        public void run(int nIterations) { Run(0, nIterations); }

        // `restart` is a public field of the reference, but only the constructors' restart vectors have a device kernel
        // (one-hot at the seed / uniform 1/n): a host-edited vector is refused, as the Python mirror does
        void CheckRestart() {
            for (int i = 0; i < nNodes; i++) {
                double expect = seed < 0 ? 1d / nNodes : (i == seed ? 1d : 0d);
                if (restart[i] != expect)
                    throw new System.NotSupportedException("Model.restart was modified: only the constructors' restart vectors are supported");
            }
        }

        // the reference's public single steps (Model.cs:76,103,110)
        public void deliverRanks() {
            CheckRestart();
            for (int i = 0; i < nNodes; i++)
                if (nextRank[i] != 0)
                    throw new System.InvalidOperationException("deliverRanks() on a non-zero nextRank: call updateRanks() first");
            Native.Check(Native.rwr_model_deliver(graph.handle, seed, dampingFactor, rank, nextRank));
        }
        public void updateRanks() {
            for (int i = 0; i < nNodes; i++) { rank[i] = nextRank[i]; nextRank[i] = 0; }
        }
        public bool checkConvergence(double threshold) {
            double diff = 0;
            for (int i = 0; i < nNodes; i++) diff += System.Math.Abs(rank[i] - nextRank[i]);
            return diff < threshold;
        }
    }
}
