"""net_params of the reference's shipped sign-invariant ZINC configurations (GraphPrediction/configs/*/*_ZINC_LapPE_signinv_GIN*.json:
hidden width, depth, heads / towers, k, readout, pe_aggregate; sign_inv_net = GINDeepSigns / MaskedGINDeepSigns with 8 layers) as the
constructor dictionaries of `dgl_nets` — used by the full-size parity tests and by `bench.py --workload dgl`."""

COMMON = dict(num_atom_type=28, num_bond_type=4, in_feat_dropout=0.0, dropout=0.0, batch_norm=True, residual=True, edge_feat=True,
              pe_init="lap_pe", lap_method="sign_inv", lap_lspe=False, use_lapeig_loss=False, lambda_loss=1, alpha_loss=1e-4,
              sign_inv_net="gin", sign_inv_layers=8, sign_inv_activation="relu", phi_out_dim=4)
SHIPPED = {
    "gin": dict(cls="GINNet", hidden_dim=95, out_dim=95, L=16, readout="mean", pos_enc_dim=8, pe_aggregate="concat"),
    "gatedgcn": dict(cls="GatedGCNNet", hidden_dim=68, out_dim=68, L=16, readout="mean", pos_enc_dim=8, pe_aggregate="concat"),
    "gat": dict(cls="GATNet", hidden_dim=59, out_dim=59, L=8, n_heads=4, readout="mean", pos_enc_dim=8, pe_aggregate="concat"),
    "pna": dict(cls="PNANet", hidden_dim=70, out_dim=70, L=16, readout="sum", pos_enc_dim=8, pe_aggregate="add", graph_norm=True,
                aggregators="mean max min std", scalers="identity amplification attenuation", towers=5, divide_input_first=True,
                divide_input_last=True, edge_dim=40, pretrans_layers=1, posttrans_layers=1, gru=False,
                avg_d=dict(lin=2.2, exp=0.6, log=1.1)),
    "transformer": dict(cls="TransformerNet", hidden_dim=64, out_dim=64, L=10, n_heads=8, readout="sum", pos_enc_dim=16,
                        pe_aggregate="concat", full_graph=False, layer_norm=True),
    # the three shipped MASKED configs (k = 37 = every eigenvector of the largest ZINC graph, sign_inv_net = masked_gin):
    # configs/gatedgcn/GatedGCN_ZINC_LapPE_signinv_GIN_mask.json, pna/PNA_ZINC_LapPE_signinv_GIN_mask.json,
    # transformer/Transformer_ZINC_LapPE_signinv_GIN_masked.json
    "gatedgcn_mask": dict(cls="GatedGCNNet", hidden_dim=67, out_dim=67, L=16, readout="mean", pos_enc_dim=37, pe_aggregate="concat",
                          sign_inv_net="masked_gin", phi_out_dim=67),
    "pna_mask": dict(cls="PNANet", hidden_dim=70, out_dim=70, L=16, readout="sum", pos_enc_dim=37, pe_aggregate="concat", graph_norm=True,
                     aggregators="mean max min std", scalers="identity amplification attenuation", towers=5, divide_input_first=True,
                     divide_input_last=True, edge_dim=40, pretrans_layers=1, posttrans_layers=1, gru=False,
                     avg_d=dict(lin=2.2, exp=0.6, log=1.1), sign_inv_net="masked_gin", phi_out_dim=70),
    "transformer_mask": dict(cls="TransformerNet", hidden_dim=56, out_dim=56, L=10, n_heads=8, readout="sum", pos_enc_dim=37,
                             pe_aggregate="concat", full_graph=False, layer_norm=True, sign_inv_net="masked_gin", phi_out_dim=16),
}


def net_params(name, device):
    """(class name, constructor dictionary) of the shipped configuration `name`."""
    c = dict(SHIPPED[name])
    cls = c.pop("cls")
    p = dict(COMMON, device=str(device))
    p.update(c)
    return cls, p
