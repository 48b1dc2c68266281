"""`from sign_net.sign_net import SignNetGNN` (Alchemy/main_alchemy.py:21) -> the HIP module, Alchemy semantics."""
from signnet_basisnet_amd.pyg import SignNetGNN as _Impl


class SignNetGNN(_Impl):
    def __init__(self, node_feat, edge_feat, n_hid, n_out, nl_signnet, nl_gnn, nl_rho=4, ignore_eigval=False, gnn_type="GINEConv"):
        super().__init__(node_feat, edge_feat, n_hid, n_out, nl_signnet, nl_gnn, nl_rho, ignore_eigval, gnn_type, variant="alchemy")
