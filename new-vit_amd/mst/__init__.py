"""MI355X-native Medical Slice Transformer hot path (drop-in for the reference's ``mst`` package).

Import paths mirror the reference (``mst.models.dino.DinoV2ClassifierSlice`` ...); the compute runs
in hand-written HIP kernels behind the C ABI of ``include/mst_hip.h`` (``mst.hip``).

Only the hot path lives here.  Everything else the reference's scripts import from ``mst``
(``mst.data``, ``mst.utils``, ``mst.models.utils.functions``: scripts/main_predict.py:23-30,
scripts/main_train.py:13-19) stays the reference's own code: when a second ``mst`` package is found
later on ``sys.path`` its directory is appended to this package's ``__path__``, so those sub-modules
resolve there while ``mst.models.{dino,resnet,base_model}`` stay this build's.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
__version__ = "0.2.0"
