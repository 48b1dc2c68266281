"""Drop-in binding of the reference's entry scripts to the HIP modules.

The reference is pure Python: its "FFI" for this path is the set of dotted module names its entry scripts import
(`Alchemy/main_alchemy.py:20-22`, `GINESignNetPyG/train/zinc.py:2-6`, `GraphPrediction/main_ZINC_graph_regression.py:47`
through `nets/ZINC_graph_regression/load_net.py:6-10`, `LearningFilters/training.py:8-10`).  Two ways to rebind them, both
leaving every file of the reference tree untouched and every OTHER module of the tree (`core.config`, `core.train`,
`core.model`, `layers.mlp_readout_layer`, `nets.ZINC_graph_regression.load_net`, `utils`, `data.*`, `train.*`) importing from the
tree as before:

1. `install(tree)` puts a meta-path finder in front of the import system that answers exactly the names of `ALIASES[tree]` with the
   HIP modules.  It does not depend on the order of `sys.path` (main_alchemy.py:5-6 inserts '.' and '..' at position 0 itself, so
   no PYTHONPATH entry can win there), and it is what the runner uses:

       cd <reference>/GraphPrediction && python -m signnet_basisnet_amd.dropin.run main_ZINC_graph_regression.py --config ...

2. The shim directories next to this file (`alchemy/`, `gine_pyg/`, `graphprediction/`, `learningfilters/`): put ONE of them
   ahead of the tree on `sys.path`.  `sign_net/` and `core/` are regular packages in the reference, so the shim's `__init__.py`
   extends `__path__` over the tree's package of the same name; `layers/` and `nets/` are namespace packages in the reference, so
   the shim has no `__init__.py` there and the portions merge, the shim's modules first.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

_PKG = __name__

# tree -> {dotted name the reference imports: module of this package that provides it}
ALIASES = {
    "alchemy": {
        "sign_net.sign_net": _PKG + ".alchemy.sign_net.sign_net",
        "sign_net.transform": _PKG + ".alchemy.sign_net.transform",
    },
    "gine_pyg": {
        "core.sign_net": _PKG + ".gine_pyg.core.sign_net",
        "core.transform": _PKG + ".gine_pyg.core.transform",
    },
    "graphprediction": {
        "layers.deepsigns": _PKG + ".graphprediction.layers.deepsigns",
        **{"nets.ZINC_graph_regression." + m: _PKG + ".graphprediction.nets.ZINC_graph_regression." + m
           for m in ("sign_inv_net", "gin_net", "gatedgcn_net", "pna_net", "transformer_net", "gat_net")},
    },
    "learningfilters": {
        "models": _PKG + ".learningfilters.models",
        "signbasisnet": _PKG + ".learningfilters.signbasisnet",
        "ign": _PKG + ".learningfilters.ign",
    },
}

# entry script (basename) -> tree, for the runner
SCRIPTS = {
    "main_alchemy.py": "alchemy",
    "zinc.py": "gine_pyg",
    "main_ZINC_graph_regression.py": "graphprediction",
    "training.py": "learningfilters",
}


def shim_dir(tree: str) -> str:
    if tree not in ALIASES:
        raise KeyError(f"unknown reference tree {tree!r}; one of {sorted(ALIASES)}")
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), tree)


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target: str):
        self.target = target

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        impl = importlib.import_module(self.target)
        keep = {"__name__", "__spec__", "__loader__", "__package__", "__path__", "__file__", "__cached__"}
        for k, v in vars(impl).items():
            if k not in keep:
                module.__dict__[k] = v
        module.__dict__["__signnet_hip__"] = self.target
        module.__dict__.setdefault("__file__", getattr(impl, "__file__", None))


class AliasFinder(importlib.abc.MetaPathFinder):
    """Answers the names of `ALIASES[tree]` and nothing else; every other import goes on to the normal finders."""

    def __init__(self, tree: str):
        self.tree = tree
        self.table = dict(ALIASES[tree])

    def find_spec(self, fullname, path=None, target=None):
        impl = self.table.get(fullname)
        if impl is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(impl), origin=impl)


def install(tree: str) -> AliasFinder:
    """Route the reference's module names of `tree` to the HIP modules.  Idempotent per tree."""
    shim_dir(tree)
    for f in sys.meta_path:
        if isinstance(f, AliasFinder) and f.tree == tree:
            return f
    finder = AliasFinder(tree)
    sys.meta_path.insert(0, finder)
    for name in finder.table:           # a module imported before install() would otherwise stay bound to the tree's file
        sys.modules.pop(name, None)
    return finder


def uninstall(tree: str | None = None) -> None:
    for f in list(sys.meta_path):
        if isinstance(f, AliasFinder) and (tree is None or f.tree == tree):
            sys.meta_path.remove(f)
            for name in f.table:
                sys.modules.pop(name, None)


def reference_module(name: str, exclude_dir: str):
    """The module `name` as the NEXT location on `sys.path` (not `exclude_dir`) provides it, loaded under a private name.

    Used by `learningfilters/models.py` to hand the spectral baselines of `training.py:9` (ChebNet, BernNet, ...) on to the
    reference's own `models.py`: they are competitors of the path, not part of it, and stay the reference's code.
    """
    private = "_signnet_reference_" + name.replace(".", "_")
    if private in sys.modules:
        return sys.modules[private]
    exclude = os.path.realpath(exclude_dir)
    search = [p for p in sys.path if os.path.realpath(p or os.getcwd()) != exclude]
    spec = importlib.machinery.PathFinder.find_spec(name, search)
    if spec is None or spec.loader is None or not spec.origin or os.path.realpath(os.path.dirname(spec.origin)) == exclude:
        return None
    spec = importlib.util.spec_from_file_location(private, spec.origin)
    module = importlib.util.module_from_spec(spec)
    sys.modules[private] = module
    try:
        spec.loader.exec_module(module)
    except BaseException:
        sys.modules.pop(private, None)
        raise
    return module
