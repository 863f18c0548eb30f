"""Same import surface as the reference's mst/models/__init__.py:1-2."""
from .resnet import ResNet, ResNetSliceTrans
from .dino import DinoV2ClassifierSlice
