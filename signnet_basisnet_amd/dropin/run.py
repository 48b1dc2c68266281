"""Run one of the reference's entry scripts, unchanged, on the HIP modules.

    cd <reference>/Alchemy          && python -m signnet_basisnet_amd.dropin.run main_alchemy.py
    cd <reference>/GINESignNetPyG   && python -m signnet_basisnet_amd.dropin.run train/zinc.py model.gnn_type GINEConv ...
    cd <reference>/GraphPrediction  && python -m signnet_basisnet_amd.dropin.run main_ZINC_graph_regression.py --config configs/...
    cd <reference>/LearningFilters  && python -m signnet_basisnet_amd.dropin.run training.py --net DS --use_eig --lap_method basis_inv

The script is executed with `runpy` as `__main__` with its own `sys.argv`; `sys.path[0]` is the script's directory as under
`python script.py` (the working directory stays where the caller is: the scripts open `data/...`, `configs/...` relative to it,
and GINESignNetPyG's README runs `python -m train.zinc` from the tree root, which is why '' stays on the path too).  The only
difference to a plain run is the finder of `dropin.install(tree)`: the names listed in `dropin.ALIASES[tree]` come from this
package, everything else from the tree.  `--tree` overrides the guess from the script's file name.
"""
from __future__ import annotations

import os
import runpy
import sys

from . import ALIASES, SCRIPTS, install


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    tree = None
    if argv and argv[0] == "--tree":
        if len(argv) < 2:
            raise SystemExit("--tree needs a value: " + ", ".join(sorted(ALIASES)))
        tree, argv = argv[1], argv[2:]
    elif argv and argv[0].startswith("--tree="):
        tree, argv = argv[0].split("=", 1)[1], argv[1:]
    if not argv:
        raise SystemExit("usage: python -m signnet_basisnet_amd.dropin.run [--tree T] <entry script> [script arguments ...]\n"
                         "trees: " + ", ".join(f"{t} ({s})" for s, t in SCRIPTS.items()))
    script = argv[0]
    if not os.path.isfile(script):
        raise SystemExit(f"{script}: no such entry script (run from the reference tree, as the script's README says)")
    if tree is None:
        tree = SCRIPTS.get(os.path.basename(script))
        if tree is None:
            raise SystemExit(f"cannot tell the reference tree from {os.path.basename(script)!r}; pass --tree " + "|".join(sorted(ALIASES)))
    install(tree)
    script_dir = os.path.dirname(os.path.abspath(script))
    for p in (os.getcwd(), script_dir):          # script_dir ends up first, as under `python script.py`
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
