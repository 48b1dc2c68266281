"""`from models import ChebNet,BernNet,GcnNet,GatNet,ARMANet,GPRNet,MLP,EqDeepSetsEncoder, Transformer`
(LearningFilters/training.py:9).

`MLP`, `EqDeepSetsEncoder` and `Transformer` — the base models the sign / basis invariant features feed — are the HIP modules.
The six spectral graph-convolution baselines of that import line are competitors of the path, not part of it: they resolve,
lazily, to the reference's own `models.py` found later on `sys.path` (which needs torch_geometric, as it always did); without
a reference tree on the path they are placeholders that raise on construction, so the import line itself always succeeds.
"""
import os as _os

from signnet_basisnet_amd.basisnet import EqDeepSetsEncoder  # noqa: F401
from signnet_basisnet_amd.learning_filters import MLP, Transformer  # noqa: F401

BASELINES = ("ChebNet", "BernNet", "GcnNet", "GatNet", "ARMANet", "GPRNet")
_HERE = _os.path.dirname(_os.path.abspath(__file__))


def _placeholder(name, why):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            f"{name} is a spectral baseline of LearningFilters/models.py, outside the sign/basis-invariant path; {why}")
    return type(name, (), {"__init__": __init__, "__doc__": f"placeholder for the reference's {name}"})


def __getattr__(name):
    if name not in BASELINES:
        raise AttributeError(f"module 'models' has no attribute {name!r}")
    from signnet_basisnet_amd.dropin import reference_module
    try:
        ref = reference_module("models", _HERE)
        why = "no reference models.py found on sys.path"
    except ImportError as e:            # the tree is there but torch_geometric (its dependency) is not
        ref, why = None, f"the reference's own models.py does not import here ({e})"
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    return _placeholder(name, why)
