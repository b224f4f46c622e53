// The two value types of the reference's public surface (Recommenders/RWRBased/Graph.cs:4-35), field for field: the
// unmodified host constructs them and reads/writes their public fields (DataLoader.cs:40,53,66,73,424;
// Experiment.cs:90-97), so names, field order and constructors are part of the drop-in contract.
namespace Recommenders.RWRBased {
    public struct Node {
        public long id;
        public NodeType type;
        public Node(long id) { this.id = id; this.type = NodeType.UNDEFINED; }
        public Node(long id, NodeType type) { this.id = id; this.type = type; }
    }

    public struct ForwardLink {
        public int targetNode;
        public EdgeType type;
        public double weight;
        public ForwardLink(int targetNode, double weight) { this.targetNode = targetNode; this.type = EdgeType.UNDEFINED; this.weight = weight; }
        public ForwardLink(int targetNode, EdgeType type, double weight) { this.targetNode = targetNode; this.type = type; this.weight = weight; }
    }

}
