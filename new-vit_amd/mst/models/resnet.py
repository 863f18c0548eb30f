"""Placeholders for the reference's ResNet models (mst/models/resnet.py).

`ResNet` / `ResNetSliceTrans` are a "next" row of the scope contract: their
torchvision / MONAI backbones are not part of the reference tree and have no HIP path yet.  The
classes exist so that `from mst.models.resnet import ResNet, ResNetSliceTrans` and the
`isinstance` dispatch of scripts/main_predict.py:136-143 keep working; constructing one fails loudly.
"""
from .base_model import BasicClassifier


class ResNet(BasicClassifier):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "mst.models.resnet.ResNet has no MI355X-native implementation yet (SURVEY.md 8f-2); "
            "only DinoV2ClassifierSlice is built on the HIP path")


class ResNetSliceTrans(BasicClassifier):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "mst.models.resnet.ResNetSliceTrans has no MI355X-native implementation yet (SURVEY.md 8f-2); "
            "only DinoV2ClassifierSlice is built on the HIP path")
