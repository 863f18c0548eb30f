"""Import surface of the reference's ``mst.models`` package (mst/models/__init__.py:1-2; scripts/main_train.py:18-19 and
scripts/main_predict.py:27-28 import from here and from the sub-modules).

``DinoV2ClassifierSlice`` is the MI355X hot path; ``ResNet`` / ``ResNetSliceTrans`` run their inference forward on HIP kernels too
(models/resnet.py); ``DinoV3ClassifierSlice`` exists so that the scripts' imports and ``isinstance`` dispatch keep working (it raises
on construction: out of scope).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # mst.models.extern etc. fall through to a reference checkout (mst/__init__.py)

from .base_model import BasicClassifier, BasicModel
from .dino import DinoV2ClassifierSlice, DinoV3ClassifierSlice
from .resnet import ResNet, ResNetSliceTrans

__all__ = ["BasicModel", "BasicClassifier", "DinoV2ClassifierSlice", "DinoV3ClassifierSlice", "ResNet", "ResNetSliceTrans"]
