"""`from core.transform import EVDTransform` (GINESignNetPyG/train/zinc.py:6)."""
from signnet_basisnet_amd.transform import BatchEVDTransform, EVDTransform, evd_laplacian  # noqa: F401
