"""`from signbasisnet import SignPlus, IGNBasisInv, IGNShared` (LearningFilters/training.py)."""
from signnet_basisnet_amd.basisnet import IGN2to1, IGNBasisInv, IGNShared, SignPlus  # noqa: F401
