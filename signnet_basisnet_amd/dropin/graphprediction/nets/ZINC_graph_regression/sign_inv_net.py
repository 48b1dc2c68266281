"""`from nets.ZINC_graph_regression.sign_inv_net import get_sign_inv_net` (used by every *_net.py of the DGL tree)."""
from signnet_basisnet_amd.dgl_deepsigns import get_sign_inv_net  # noqa: F401
