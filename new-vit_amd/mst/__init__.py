"""MI355X-native Medical Slice Transformer hot path (drop-in for the reference's ``mst`` package).

Import paths mirror the reference (``mst.models.dino.DinoV2ClassifierSlice`` ...); the compute runs
in hand-written HIP kernels behind the C ABI of ``include/mst_hip.h`` (``mst.hip``).
"""
__version__ = "0.1.0"
