"""`from nets.ZINC_graph_regression.gat_net import GATNet` (GraphPrediction/nets/ZINC_graph_regression/load_net.py)."""
from signnet_basisnet_amd.dgl_nets import GATNet  # noqa: F401
