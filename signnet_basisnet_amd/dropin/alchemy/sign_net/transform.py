"""`from sign_net.transform import EVDTransform` (Alchemy/main_alchemy.py:22)."""
from signnet_basisnet_amd.transform import BatchEVDTransform, EVDTransform, evd_laplacian  # noqa: F401
