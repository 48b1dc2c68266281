"""Shim portion of the reference's regular package `sign_net` (Alchemy/sign_net/__init__.py, empty): this directory answers
`sign_net.sign_net` and `sign_net.transform`; every other submodule (`sign_net.model`, `sign_net.model_utils.*`) is found in the
tree's own `sign_net/` directory later on `sys.path`."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
