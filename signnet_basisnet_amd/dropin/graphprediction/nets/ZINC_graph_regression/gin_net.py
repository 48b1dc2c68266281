"""`from nets.ZINC_graph_regression.gin_net import GINNet` (GraphPrediction/nets/ZINC_graph_regression/load_net.py)."""
from signnet_basisnet_amd.dgl_nets import GINNet  # noqa: F401
