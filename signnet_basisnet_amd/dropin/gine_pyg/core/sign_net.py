"""`from core.sign_net import SignNetGNN` (GINESignNetPyG/train/zinc.py:5) -> the HIP module, GINESignNetPyG semantics."""
from signnet_basisnet_amd.pyg import SignNetGNN as _Impl


class SignNetGNN(_Impl):
    def __init__(self, node_feat, edge_feat, n_hid, n_out, nl_signnet, nl_gnn):
        super().__init__(node_feat, edge_feat, n_hid, n_out, nl_signnet, nl_gnn, variant="gine")
