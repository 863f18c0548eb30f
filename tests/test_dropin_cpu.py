"""The drop-in recipe of INTEGRATION.md section A, executed: the import lines of the reference's own scripts
(scripts/main_predict.py:23-30, scripts/main_train.py:13-19) must resolve with this build ahead of the reference --
``mst.models.{dino,resnet,base_model}`` from the build, ``mst.data`` / ``mst.utils`` / ``mst.models.utils.functions`` from
the reference.  Third-party packages the image lacks (torchio, monai, torchvision, ...) are replaced by empty stand-ins:
only name resolution is under test, nothing of them runs.  Needs the reference checkout (skipped on the GPU box)."""
import json
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "new-vit_amd"
REF = Path("/root/reference")

pytestmark = pytest.mark.skipif(not (REF / "scripts" / "main_predict.py").exists(), reason="reference checkout not present")

_DRIVER = textwrap.dedent(r'''
    import importlib.abc, importlib.machinery, importlib.util, json, re, sys, types
    BUILD, REF, MODE = sys.argv[1], sys.argv[2], sys.argv[3]

    # third-party roots the two scripts and the reference modules they pull in import (environment.yaml:9-25)
    CANDIDATES = {"torchio", "monai", "torchvision", "seaborn", "pytorch_lightning", "torchmetrics", "h5py", "SimpleITK",
                  "nibabel", "matplotlib", "sklearn", "pandas", "tqdm", "transformers", "einops", "scipy"}

    class _AnyMeta(type):
        def __getattr__(cls, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _AnyMeta(name, (), {})
        def __call__(cls, *a, **k):
            return cls
    class _Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _AnyMeta(name, (), {})
    class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        """An importable-but-empty stand-in for every third-party root that is absent from this image."""
        def __init__(self):
            self.roots = set()
        def find_spec(self, fullname, path=None, target=None):
            root = fullname.split(".")[0]
            if root == "mst":
                return None
            if root not in self.roots:
                if fullname != root or root not in CANDIDATES:
                    return None
                for f in sys.meta_path:
                    if f is not self and getattr(f, "find_spec", None) and f.find_spec(fullname, None) is not None:
                        return None                      # really installed: leave it alone
                self.roots.add(root)
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        def create_module(self, spec):
            m = _Stub(spec.name)
            m.__path__ = []
            return m
        def exec_module(self, module):
            pass

    def import_lines(script):
        out = []
        for line in open(script).read().splitlines():
            if re.match(r"^(def|class|if __name__)\b", line):
                break
            if re.match(r"^(from\s+\S+\s+import\s+.+|import\s+.+)$", line.strip()) and not line.startswith((" ", "\t")):
                out.append(line.strip())
        return out

    if MODE == "pythonpath":            # PYTHONPATH=new-vit_amd:<reference>  (main_train.py leaves sys.path alone)
        sys.path[:0] = [BUILD, REF]
    else:                               # main_predict.py:11-12 pushes its project root to the FRONT: mst_dropin.install() is needed
        sys.path[:0] = [REF, BUILD]
        import mst_dropin
        mst_dropin.install()
    import mst.models                    # the build's (pytorch_lightning absent -> its stand-in base class)
    sys.meta_path.append(_StubFinder())  # behind the real finders: only what is missing becomes a stand-in

    executed = []
    for script in ("main_predict.py", "main_train.py"):
        for line in import_lines(REF + "/scripts/" + script):
            if line.startswith(("print", "sys.path")):
                continue
            exec(line, {})
            executed.append(line)
    import mst, mst.models.dino, mst.models.resnet, mst.models.base_model, mst.data, mst.utils.roc_curve
    import mst.models.utils.functions, mst.data.datamodules, mst.data.datasets.dataset_3d_lidc
    from mst.models.dino import DinoV2ClassifierSlice
    from mst.models import DinoV2ClassifierSlice as D2
    print(json.dumps({
        "executed": executed,
        "cls_module_file": sys.modules[DinoV2ClassifierSlice.__module__].__file__,
        "same_class": D2 is DinoV2ClassifierSlice,
        "files": {n: sys.modules[n].__file__ for n in (
            "mst", "mst.models", "mst.models.dino", "mst.models.resnet", "mst.models.base_model", "mst.data",
            "mst.data.datamodules", "mst.data.datasets.dataset_3d_lidc", "mst.utils.roc_curve", "mst.models.utils.functions")},
    }))
''')


@pytest.mark.parametrize("mode", ["pythonpath", "dropin"])
def test_reference_script_imports_resolve(mode, tmp_path):
    drv = tmp_path / "drv.py"
    drv.write_text(_DRIVER)
    r = subprocess.run([sys.executable, str(drv), str(BUILD), str(REF), mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    ex = "\n".join(info["executed"])
    for needle in ("from mst.data.datasets.dataset_3d_lidc import LIDC_Dataset3D", "from mst.data.datamodules import DataModule",
                   "from mst.models.resnet import ResNet, ResNetSliceTrans",
                   "from mst.models.dino import DinoV2ClassifierSlice, DinoV3ClassifierSlice",
                   "from mst.utils.roc_curve import plot_roc_curve, cm2acc, cm2x",
                   "from mst.models.utils.functions import tensor2image, tensor_cam2image, minmax_norm, one_hot",
                   "from pytorch_lightning.trainer import Trainer"):
        assert needle in ex, f"import line not executed: {needle}"
    f = info["files"]
    for name in ("mst", "mst.models", "mst.models.dino", "mst.models.resnet", "mst.models.base_model"):
        assert f[name].startswith(str(BUILD)), (name, f[name])
    for name in ("mst.data", "mst.data.datamodules", "mst.data.datasets.dataset_3d_lidc", "mst.utils.roc_curve",
                 "mst.models.utils.functions"):
        assert f[name].startswith(str(REF)), (name, f[name])
    assert info["cls_module_file"].startswith(str(BUILD)) and info["same_class"]


def test_dropin_runner_runs_a_script(tmp_path):
    """python -m mst_dropin <script>: the script sees the build's classes even after pushing the reference to sys.path[0]."""
    script = tmp_path / "probe.py"
    script.write_text(textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {str(REF)!r})            # what scripts/main_predict.py:11-12 does
        from mst.models.dino import DinoV2ClassifierSlice
        import mst.models.dino as d
        assert __name__ == "__main__" and sys.argv[1:] == ["--flag", "7"], sys.argv
        print("ORIGIN", d.__file__)
    '''))
    r = subprocess.run([sys.executable, "-m", "mst_dropin", str(script), "--flag", "7"], capture_output=True, text=True,
                       env={**__import__("os").environ, "PYTHONPATH": str(BUILD)}, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"ORIGIN {BUILD}" in r.stdout
