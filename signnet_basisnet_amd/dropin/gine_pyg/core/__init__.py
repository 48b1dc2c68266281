"""Shim portion of the reference's regular package `core` (GINESignNetPyG/core/__init__.py, empty): this directory answers
`core.sign_net` and `core.transform`; `core.config`, `core.train`, `core.model`, `core.log`, `core.model_utils.*`
(GINESignNetPyG/train/zinc.py:2-4) are found in the tree's own `core/` directory later on `sys.path`."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
