"""`from nets.ZINC_graph_regression.transformer_net import TransformerNet` (GraphPrediction/nets/ZINC_graph_regression/load_net.py)."""
from signnet_basisnet_amd.dgl_nets import TransformerNet  # noqa: F401
