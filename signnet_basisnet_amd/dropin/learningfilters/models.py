"""`from models import MLP, EqDeepSetsEncoder, Transformer` (LearningFilters/training.py:9).  The graph-convolution baselines of
that import line (ChebNet, BernNet, GcnNet, GatNet, ARMANet, GPRNet) are not part of the sign / basis invariant path."""
from signnet_basisnet_amd.basisnet import EqDeepSetsEncoder  # noqa: F401
from signnet_basisnet_amd.learning_filters import MLP, Transformer  # noqa: F401
