"""`from nets.ZINC_graph_regression.gatedgcn_net import GatedGCNNet` (GraphPrediction/nets/ZINC_graph_regression/load_net.py)."""
from signnet_basisnet_amd.dgl_nets import GatedGCNNet  # noqa: F401
