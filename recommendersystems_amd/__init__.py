"""MI355X-native Random-Walk-with-Restart engine behind the public surface of the
reference's ``Recommenders.RWRBased`` (ChangUk/RecommenderSystems).

  csrc/          hand-written HIP kernels (gfx950) + the C-ABI of include/rwr.h -> librwr.so
  _lib.py        ctypes binding of that C-ABI
  rwr_based.py   host-side mirror of Graph / Model / Recommender / Node / ForwardLink / enums
  synth.py       deterministic integer-only synthetic bipartite graphs
"""
from .rwr_based import EdgeType, EvaluateGraphs, ForwardLink, Graph, Model, Node, NodeType, Recommender  # noqa: F401
from ._lib import RwrError  # noqa: F401

__all__ = ["EdgeType", "EvaluateGraphs", "ForwardLink", "Graph", "Model", "Node", "NodeType", "Recommender", "RwrError"]
