"""Make this build's ``mst.models.{dino,resnet,base_model}`` win over a reference checkout's, whatever ``sys.path`` says.

``PYTHONPATH=new-vit_amd`` is enough for callers that leave ``sys.path`` alone (scripts/main_train.py).
scripts/main_predict.py:11-12 however pushes its own project root to ``sys.path[0]``, so a plain path
order would resolve ``mst`` to the reference again.  ``install()`` puts a meta-path finder in front of
the path machinery that serves every module this build HAS under ``mst`` from this build; modules it does
not have (``mst.data``, ``mst.utils``, ``mst.models.utils.functions``, ``mst.models.extern``) are found through
the packages' ``__path__``, which the build's ``__init__``s extend with the reference's directories.

    PYTHONPATH=/path/to/new-vit_amd python -m mst_dropin /path/to/reference/scripts/main_predict.py --run_folder ...

or, inside a program:  ``import mst_dropin; mst_dropin.install()`` before the first ``import mst``.
"""
from __future__ import annotations

import importlib.abc
import importlib.util
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent


class BuildFinder(importlib.abc.MetaPathFinder):
    def __init__(self, root: Path = ROOT):
        self.root = Path(root)

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "mst" and not fullname.startswith("mst."):
            return None
        rel = self.root.joinpath(*fullname.split("."))
        init = rel / "__init__.py"
        if init.is_file():
            return importlib.util.spec_from_file_location(fullname, init, submodule_search_locations=[str(rel)])
        mod = rel.with_suffix(".py")
        if mod.is_file():
            return importlib.util.spec_from_file_location(fullname, mod)
        return None          # not part of this build: the parent package's (extended) __path__ decides


def install(root: Path = ROOT) -> BuildFinder:
    for f in sys.meta_path:
        if isinstance(f, BuildFinder) and f.root == Path(root):
            return f
    # an ``mst`` imported earlier from somewhere else would keep serving its own sub-modules
    for name in [n for n in sys.modules if n == "mst" or n.startswith("mst.")]:
        origin = getattr(sys.modules[name], "__file__", None) or ""
        if not origin.startswith(str(root)):
            del sys.modules[name]
    finder = BuildFinder(root)
    sys.meta_path.insert(0, finder)
    if str(root) not in sys.path:
        sys.path.append(str(root))     # extend_path() of the build's packages scans sys.path for the reference's
    return finder


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m mst_dropin <script.py> [script arguments ...]")
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
