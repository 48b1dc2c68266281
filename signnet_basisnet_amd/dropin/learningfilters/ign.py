"""`from ign import IGN2to1` (LearningFilters/signbasisnet.py:7)."""
from signnet_basisnet_amd.basisnet import IGN2to1, layer_1_to_1, layer_2_to_1  # noqa: F401
