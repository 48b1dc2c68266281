"""Boundary (b): the reference's entry scripts run UNCHANGED against the HIP modules.

The reference tree does not travel, so the tests build a SYNTHETIC tree in tmp_path with the reference's package structure and
the import statements of its entry scripts (`GINESignNetPyG/train/zinc.py:2-6`, `GraphPrediction/main_ZINC_graph_regression.py:47`
+ `nets/ZINC_graph_regression/load_net.py:6-10`, `LearningFilters/training.py:8-10`, `Alchemy/main_alchemy.py:5-6,20-22`); every
module of the synthetic tree carries `ORIGIN = "tree"`.  Each script reports where every imported name came from.  Two bindings
are exercised in fresh interpreters: the shim directory first on `sys.path`, and the runner
`python -m signnet_basisnet_amd.dropin.run <script>` (meta-path finder; independent of path order).
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, "signnet_basisnet_amd", "dropin")

REPORT = '''
import json as _json
def _origin(obj):
    mod = getattr(obj, "__module__", None) or getattr(obj, "__name__", "")
    m = __import__("sys").modules.get(mod)
    if mod.startswith("signnet_basisnet_amd") or any(
            (getattr(c, "__module__", "") or "").startswith("signnet_basisnet_amd") for c in getattr(obj, "__mro__", ())):
        return "hip"            # the shim's thin subclasses (ctor adapters) count through their base class
    return getattr(m, "ORIGIN", "?:" + mod)
print("REPORT " + _json.dumps({k: _origin(v) for k, v in dict(globals()).items()
                               if not k.startswith("_") and k not in ("sys", "torch", "json")}))
'''


def _write(root, files):
    for rel, body in files.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(textwrap.dedent(body))


def _cls(*names):
    return 'ORIGIN = "tree"\n' + "".join(f"class {n}:\n    pass\n" for n in names)


def _fn(*names):
    return 'ORIGIN = "tree"\n' + "".join(f"def {n}(*a, **k):\n    return None\n" for n in names)


def make_gine_tree(root):
    _write(root, {
        "core/__init__.py": "",
        "core/config.py": 'ORIGIN = "tree"\nclass _C(dict):\n    __module__ = __name__\ncfg = _C()\n' + "def update_cfg(c):\n    return c\n",
        "core/log.py": _fn("config_logger"),
        "core/train.py": "from core.log import config_logger\n" + _fn("run"),
        "core/model_utils/__init__.py": "",
        "core/model_utils/elements.py": _cls("MLP", "DiscreteEncoder"),
        "core/model.py": "from core.model_utils.elements import MLP, DiscreteEncoder\n" + _cls("GNN"),
        "core/sign_net.py": _cls("SignNetGNN"),
        "core/transform.py": _cls("EVDTransform"),
        # the import block of GINESignNetPyG/train/zinc.py:1-6
        "train/zinc.py": "import torch\nfrom core.config import cfg, update_cfg\nfrom core.train import run\nfrom core.model import GNN\n"
                         "from core.sign_net import SignNetGNN\nfrom core.transform import EVDTransform\n" + REPORT,
    })
    return "train/zinc.py", {"cfg": "tree", "update_cfg": "tree", "run": "tree", "GNN": "tree", "SignNetGNN": "hip", "EVDTransform": "hip"}


def make_graphprediction_tree(root):
    nets = ("gatedgcn_net:GatedGCNNet", "gin_net:GINNet", "gat_net:GATNet", "pna_net:PNANet", "transformer_net:TransformerNet")
    files = {
        # namespace packages, as in the reference: no __init__.py under layers/ nets/ nets/ZINC_graph_regression/ data/ train/
        "layers/mlp_readout_layer.py": _cls("MLPReadout"),
        "layers/gatedgcn_layer.py": _cls("GatedGCNLayer"),
        "layers/deepsigns.py": _cls("GINDeepSigns", "MaskedGINDeepSigns"),
        "nets/ZINC_graph_regression/sign_inv_net.py": "from layers.deepsigns import GINDeepSigns, MaskedGINDeepSigns\n" + _fn("get_sign_inv_net"),
        # load_net.py:6-10 + the dispatch function of :27-36
        "nets/ZINC_graph_regression/load_net.py": "".join(
            f"from nets.ZINC_graph_regression.{m} import {c}\n" for m, c in (n.split(":") for n in nets))
        + 'ORIGIN = "tree"\ndef gnn_model(name, net_params):\n    return {"GatedGCN": GatedGCNNet, "GIN": GINNet, "GAT": GATNet, "PNA": PNANet, "Transformer": TransformerNet}[name]\n',
        "data/data.py": _fn("LoadData"),
        "train/train_ZINC_graph_regression.py": "from train.metrics import MAE\n" + _fn("train_epoch_sparse", "evaluate_network_sparse"),
        "train/metrics.py": _fn("MAE"),
        # main_ZINC_graph_regression.py:47-48 (+ the trainer import of :103)
        "main_ZINC_graph_regression.py": "from nets.ZINC_graph_regression.load_net import gnn_model\nfrom data.data import LoadData\n"
                                         "from train.train_ZINC_graph_regression import train_epoch_sparse\n"
                                         "from layers.mlp_readout_layer import MLPReadout\n"
                                         "from layers.deepsigns import GINDeepSigns\n"
                                         "from nets.ZINC_graph_regression.sign_inv_net import get_sign_inv_net\n"
                                         + "".join(f"{k} = gnn_model('{k}', None)\n" for k in ("GatedGCN", "GIN", "GAT", "PNA", "Transformer")) + REPORT,
    }
    for n in nets:
        m, c = n.split(":")
        files[f"nets/ZINC_graph_regression/{m}.py"] = "from layers.mlp_readout_layer import MLPReadout\nfrom .sign_inv_net import get_sign_inv_net\n" + _cls(c)
    _write(root, files)
    return "main_ZINC_graph_regression.py", {
        "gnn_model": "tree", "LoadData": "tree", "train_epoch_sparse": "tree", "MLPReadout": "tree", "GINDeepSigns": "hip",
        "get_sign_inv_net": "hip", "GatedGCN": "hip", "GIN": "hip", "GAT": "hip", "PNA": "hip", "Transformer": "hip"}


def make_learningfilters_tree(root):
    base = ("ChebNet", "BernNet", "GcnNet", "GatNet", "ARMANet", "GPRNet")
    _write(root, {
        "utils.py": _fn("filtering", "TwoDGrid", "data_to_eig"),
        "models.py": _cls(*base, "MLP", "EqDeepSetsEncoder", "Transformer"),
        "ign.py": _cls("IGN2to1"),
        "signbasisnet.py": "from ign import IGN2to1\n" + _cls("SignPlus", "IGNBasisInv", "IGNShared"),
        # training.py:8-10
        "training.py": "from utils import filtering, TwoDGrid, data_to_eig\n"
                       "from models import ChebNet,BernNet,GcnNet,GatNet,ARMANet,GPRNet,MLP,EqDeepSetsEncoder, Transformer\n"
                       "from signbasisnet import SignPlus, IGNBasisInv, IGNShared\n" + REPORT,
    })
    exp = {n: "tree" for n in base + ("filtering", "TwoDGrid", "data_to_eig")}
    exp.update({n: "hip" for n in ("MLP", "EqDeepSetsEncoder", "Transformer", "SignPlus", "IGNBasisInv", "IGNShared")})
    return "training.py", exp


def make_alchemy_tree(root):
    _write(root, {
        "sign_net/__init__.py": "",
        "sign_net/model.py": "from sign_net.model_utils.elements import MLP\n" + _cls("GNN"),
        "sign_net/model_utils/__init__.py": "",
        "sign_net/model_utils/elements.py": _cls("MLP"),
        "sign_net/sign_net.py": _cls("SignNetGNN"),
        "sign_net/transform.py": _cls("EVDTransform"),
        "baseline_gin.py": _cls("NetGINE"),
        # main_alchemy.py:3-6,20-22: the script puts '.' and '..' at the head of sys.path ITSELF
        "main_alchemy.py": "import sys\nsys.path.insert(0, '..')\nsys.path.insert(0, '.')\nfrom baseline_gin import NetGINE\n"
                           "from sign_net.transform import EVDTransform\nfrom sign_net.sign_net import SignNetGNN\nfrom sign_net.model import GNN\n" + REPORT,
    })
    return "main_alchemy.py", {"NetGINE": "tree", "EVDTransform": "hip", "SignNetGNN": "hip", "GNN": "tree"}


TREES = {"gine_pyg": make_gine_tree, "graphprediction": make_graphprediction_tree,
         "learningfilters": make_learningfilters_tree, "alchemy": make_alchemy_tree}


def _run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("REPORT ")]
    assert line, out.stdout + out.stderr
    return json.loads(line[-1][len("REPORT "):])


@pytest.mark.parametrize("tree", sorted(TREES))
def test_runner_executes_the_unchanged_entry_script_on_the_hip_modules(tree, tmp_path):
    script, expected = TREES[tree](str(tmp_path))
    got = _run([sys.executable, "-m", "signnet_basisnet_amd.dropin.run", script], cwd=str(tmp_path))
    assert {k: got.get(k) for k in expected} == expected


@pytest.mark.parametrize("tree", ["gine_pyg", "graphprediction", "learningfilters"])
def test_shim_directory_first_on_sys_path_merges_with_the_tree(tree, tmp_path):
    """The documented `sys.path = [dropin/<tree>, <reference>/<tree>]` recipe: sibling modules still import from the tree."""
    script, expected = TREES[tree](str(tmp_path))
    code = (f"import sys, runpy; sys.path[:0] = [{os.path.join(DROPIN, tree)!r}, {str(tmp_path)!r}]; "
            f"runpy.run_path({script!r}, run_name='__main__')")
    got = _run([sys.executable, "-c", code], cwd=str(tmp_path))
    assert {k: got.get(k) for k in expected} == expected


def test_alchemy_shim_directory_resolves_the_two_modules_and_leaves_the_rest(tmp_path):
    make_alchemy_tree(str(tmp_path))
    code = (f"import sys; sys.path[:0] = [{os.path.join(DROPIN, 'alchemy')!r}, {str(tmp_path)!r}]\n"
            "from sign_net.sign_net import SignNetGNN\nfrom sign_net.transform import EVDTransform\nfrom sign_net.model import GNN\n"
            "from baseline_gin import NetGINE\n" + REPORT)
    got = _run([sys.executable, "-c", code], cwd=str(tmp_path))
    assert {k: got[k] for k in ("SignNetGNN", "EVDTransform", "GNN", "NetGINE")} == {
        "SignNetGNN": "hip", "EVDTransform": "hip", "GNN": "tree", "NetGINE": "tree"}


def test_learningfilters_baselines_without_a_tree_import_and_raise_on_construction(tmp_path):
    code = (f"import sys; sys.path.insert(0, {os.path.join(DROPIN, 'learningfilters')!r})\n"
            "from models import ChebNet,BernNet,GcnNet,GatNet,ARMANet,GPRNet,MLP,EqDeepSetsEncoder, Transformer\n"
            "try:\n    ChebNet()\n    print('REPORT ' + '{\"raised\": false}')\n"
            "except NotImplementedError as e:\n    print('REPORT ' + '{\"raised\": true}')\n")
    assert _run([sys.executable, "-c", code], cwd=str(tmp_path)) == {"raised": True}


def test_install_is_scoped_to_the_listed_names():
    import signnet_basisnet_amd.dropin as D
    f = D.AliasFinder("gine_pyg")
    assert f.find_spec("core.config") is None and f.find_spec("core") is None and f.find_spec("torch") is None
    assert f.find_spec("core.sign_net") is not None and f.find_spec("core.transform") is not None
    for tree, table in D.ALIASES.items():
        assert os.path.isdir(D.shim_dir(tree))
        for name, impl in table.items():        # every alias has a shim file at the same dotted path
            assert os.path.isfile(os.path.join(D.shim_dir(tree), *name.split(".")) + ".py"), (tree, name)
            assert impl.endswith(name)
    with pytest.raises(KeyError):
        D.shim_dir("nope")


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")
def test_real_reference_siblings_import_next_to_the_hip_modules():
    """In the build container: the reference's OWN `core.model` / `load_net.py` / `layers.mlp_readout_layer` import beside the HIP
    modules (third-party graph libraries through the stand-ins of tests/golden/ref_shim)."""
    shim = os.path.join(REPO, "tests", "golden", "ref_shim")
    code = textwrap.dedent(f"""
        import sys
        sys.path[:0] = [{shim!r}]
        import signnet_basisnet_amd.dropin as D
        tree = sys.argv[1]
        D.install(tree)
        out = {{}}
        if tree == "gine_pyg":
            sys.path.insert(0, {REF + '/GINESignNetPyG'!r})
            from core.model import GNN
            from core.sign_net import SignNetGNN
            import core.model_utils.elements as E
            out = dict(GNN=GNN.__module__, GNN_file=sys.modules[GNN.__module__].__file__, SignNetGNN=SignNetGNN.__module__,
                       elements=E.__file__)
        elif tree == "graphprediction":
            sys.path.insert(0, {REF + '/GraphPrediction'!r})
            from nets.ZINC_graph_regression.load_net import gnn_model
            from layers.mlp_readout_layer import MLPReadout
            import nets.ZINC_graph_regression.load_net as L
            out = dict(load_net=L.__file__, MLPReadout=sys.modules[MLPReadout.__module__].__file__,
                       nets={{k: getattr(L, k + "Net").__module__ for k in ("GatedGCN", "GIN", "GAT", "PNA", "Transformer")}})
        import json; print("REPORT " + json.dumps(out))
    """)
    got = _run([sys.executable, "-c", code, "gine_pyg"], cwd=REPO)
    assert got["GNN_file"].startswith(REF) and got["elements"].startswith(REF)
    assert got["SignNetGNN"].startswith("signnet_basisnet_amd")
    got = _run([sys.executable, "-c", code, "graphprediction"], cwd=REPO)
    assert got["load_net"].startswith(REF) and got["MLPReadout"].startswith(REF)
    assert all(v.startswith("signnet_basisnet_amd") for v in got["nets"].values()), got
