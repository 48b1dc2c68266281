"""`from layers.deepsigns import GINDeepSigns, MaskedGINDeepSigns` (GraphPrediction/nets/.../sign_inv_net.py:1)."""
from signnet_basisnet_amd.dgl_deepsigns import GIN, MLP, GINDeepSigns, MaskedGINDeepSigns  # noqa: F401
