"""models/_utils.py:180-193 and models/segmentation.py:30-84 of the reference."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch
from torch import nn

from .. import ops
from .layers import Conv2d, init_block_nhwc


class IntermediateLayerGetter(nn.ModuleDict):
    """Keeps the children of `model` up to the last requested layer (segmentation.py:55-84)."""

    def __init__(self, model: nn.Module, return_layers: Dict[str, str]) -> None:
        if not set(return_layers).issubset([name for name, _ in model.named_children()]):
            raise ValueError("return_layers are not present in model")
        orig = return_layers
        remaining = {str(k): str(v) for k, v in return_layers.items()}
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class _SimpleSegmentationModel_iekd(nn.Module):
    """Only __init__ matters on the path: it swaps backbone.conv1 for a 1-channel 7x7 stride-1
    pad-2 conv with bias (_utils.py:192).  Global_and_Local pulls the sub-modules out by name
    (ours.py:1725-1735) and never calls this forward; it is kept for API completeness."""

    def __init__(self, backbone: nn.Module, classifier: nn.Module, aux_classifier: Optional[nn.Module] = None) -> None:
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier
        self.backbone.conv1 = Conv2d(1, 64, kernel_size=7, stride=1, padding=2)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        hw = x.shape[-2:]
        bb = self.backbone
        f = ops.from_nhwc(init_block_nhwc(ops.to_nhwc(x), bb["conv1"], bb["bn1"], bb["maxpool"]))     # one launch under no_grad evaluation
        for name in ("layer1", "layer2", "layer3", "layer4"):
            f = bb[name](f)
        logits = self.classifier(f)
        out = OrderedDict()
        out["out"] = ops.bilinear_up(ops.to_nhwc(logits), int(hw[0]), int(hw[1]))
        out["x_layer4"] = f
        return out
