"""`from nets.ZINC_graph_regression.pna_net import PNANet` (GraphPrediction/nets/ZINC_graph_regression/load_net.py)."""
from signnet_basisnet_amd.dgl_nets import PNANet  # noqa: F401
