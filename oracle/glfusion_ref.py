"""TEST INFRASTRUCTURE ONLY -- the oracle for the GL-Fusion hot path.

A plain PyTorch-CPU fp32 restatement of the reference's forward/backward hot path
(SURVEY.md section 8a, rows a1-a10).  It is the checker for the HIP engine in
gl-fusion_amd/; it is never the thing measured (except as bench.py's reported
``cpu_baseline``) and never shipped on the product path.

Pinning status
--------------
* a1, a2 (conv1 swap), a4, a5, a6, a7, a9, a10: PINNED by tests/golden/*.npz, which
  were generated in the build container by importing the reference's own
  ``models/ours.py`` (tests/golden/make_golden.py) and executing it on CPU.
* a3 (ResNet-50): the BLOCK is pinned -- tests/golden/bottleneck_ref.npz comes from the reference's own in-tree Bottleneck
  (models/resnet.py:43-79, the block torchvision's ResNet-50 is made of; identity and stride-2 + downsample forms, train
  forward / gradients / running statistics / eval forward, fp32 and fp64) and ``Bottleneck`` below matches it to 1e-6.
  STILL UNPINNED: torchvision's dilation rule of ``_make_layer`` (block 0 of a dilated stage keeps the previous dilation,
  padding = dilation).  The reference takes the trunk from torchvision==0.9.1 (requirements.txt:25; call site
  models/segmentation.py:205-207), which is neither vendored under /root/reference nor installed here, and the in-tree block
  has no dilation argument; ``ResNet50Trunk`` restates the published v0.9.1 rule.  The golden generator plugs this same
  class in as the torchvision stand-in, so the end-to-end fixtures pin the wiring around the trunk, not that rule.
* f3: all eight fixture-able variants (incl. model19, Global_and_Local_CPS) are checked against fixtures of the reference's
  own classes on the CPU side (tests/test_oracle_golden.py).
* f4 (prepare_clip): PARITY UNPINNED -- monai / nibabel are absent; the transform chain is restated from
  datasets/loader.py:460-498.

All file:line citations are relative to /root/reference/GLfusion/.
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from typing import Dict, List, Sequence

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# a3: ResNet-50 (torchvision 0.9.1 `resnet50(replace_stride_with_dilation=[F,T,T])`)
# --------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    """1x1 -> BN -> ReLU -> 3x3(stride, dilation) -> BN -> ReLU -> 1x1 -> BN -> (+id) -> ReLU.

    Structure pinned in-tree only by models/resnet.py:43-79; stride lives on the 3x3
    (the "v1.5" variant torchvision ships)."""

    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1,
                 downsample: nn.Module | None = None, dilation: int = 1) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)


class ResNet50Trunk(nn.Module):
    """Child order conv1,bn1,relu,maxpool,layer1..4,avgpool,fc matters: the reference's
    IntermediateLayerGetter walks named_children() in order (models/segmentation.py:67-72)."""

    def __init__(self, replace_stride_with_dilation: Sequence[bool] = (False, True, True),
                 num_classes: int = 1000) -> None:
        super().__init__()
        self.inplanes = 64
        self.dilation = 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, 3, 1, False)
        self.layer2 = self._stage(128, 4, 2, replace_stride_with_dilation[0])
        self.layer3 = self._stage(256, 6, 2, replace_stride_with_dilation[1])
        self.layer4 = self._stage(512, 3, 2, replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _stage(self, planes: int, blocks: int, stride: int, dilate: bool) -> nn.Sequential:
        prev_dilation = self.dilation
        if dilate:                       # stride traded for dilation
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, down, prev_dilation)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes, dilation=self.dilation) for _ in range(1, blocks)]
        return nn.Sequential(*mods)


def resnet50(pretrained: bool = False, progress: bool = True, **kw) -> ResNet50Trunk:
    """Drop-in for torchvision.models.resnet.resnet50; `pretrained` is ignored (no network)."""
    return ResNet50Trunk(kw.get("replace_stride_with_dilation", (False, False, False)))


# --------------------------------------------------------------------------------------
# a4: DeepLabHead / ASPP (models/deeplabv3.py:102-166)
# --------------------------------------------------------------------------------------
class _ConvBNReLU(nn.Sequential):
    def __init__(self, cin: int, cout: int, k: int, dilation: int = 1) -> None:
        pad = 0 if k == 1 else dilation
        super().__init__(nn.Conv2d(cin, cout, k, padding=pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())


class ASPPPooling(nn.Sequential):
    """deeplabv3.py:123-135 -- global average -> 1x1 -> BN -> ReLU -> bilinear from 1x1
    (a broadcast)."""

    def __init__(self, cin: int, cout: int) -> None:
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        hw = x.shape[-2:]
        for m in self:
            x = m(x)
        return F.interpolate(x, size=hw, mode="bilinear", align_corners=False)


class ASPP(nn.Module):
    """deeplabv3.py:138-166."""

    def __init__(self, in_channels: int, atrous_rates: Sequence[int], out_channels: int = 256) -> None:
        super().__init__()
        branches: List[nn.Module] = [_ConvBNReLU(in_channels, out_channels, 1)]
        branches += [_ConvBNReLU(in_channels, out_channels, 3, r) for r in atrous_rates]
        branches.append(ASPPPooling(in_channels, out_channels))
        self.convs = nn.ModuleList(branches)
        self.project = nn.Sequential(
            nn.Conv2d(len(branches) * out_channels, out_channels, 1, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(), nn.Dropout(0.5))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.project(torch.cat([b(x) for b in self.convs], dim=1))


class DeepLabHead(nn.Sequential):
    """deeplabv3.py:102-110."""

    def __init__(self, in_channels: int, num_classes: int) -> None:
        super().__init__(ASPP(in_channels, [12, 24, 36]),
                         nn.Conv2d(256, 256, 3, padding=1, bias=False),
                         nn.BatchNorm2d(256), nn.ReLU(),
                         nn.Conv2d(256, num_classes, 1))


# --------------------------------------------------------------------------------------
# a10: the template factory (models/segmentation.py:484-500, 197-244; _utils.py:180-193)
# --------------------------------------------------------------------------------------
class _Template(nn.Module):
    """Stand-in for DeepLabV3_iekd: .backbone is a ModuleDict of the trunk children up to
    layer4 (IntermediateLayerGetter, segmentation.py:55-72) and conv1 is swapped for a
    1-channel 7x7 stride-1 pad-2 conv WITH bias (_utils.py:192)."""

    def __init__(self, num_classes: int) -> None:
        super().__init__()
        trunk = ResNet50Trunk((False, True, True))
        kept = OrderedDict()
        for name, child in trunk.named_children():
            kept[name] = child
            if name == "layer4":
                break
        self.backbone = nn.ModuleDict(kept)
        self.classifier = DeepLabHead(2048, num_classes)
        self.backbone["conv1"] = nn.Conv2d(1, 64, kernel_size=7, stride=1, padding=2)


def deeplabv3_resnet50_iekd(pretrained: bool = False, progress: bool = True,
                            num_classes: int = 21, aux_loss=None, **kw) -> _Template:
    return _Template(num_classes)


# --------------------------------------------------------------------------------------
# a6: TPAVIModule (models/ours.py:770-917), modes 'dot' and 'embedded' only
# --------------------------------------------------------------------------------------
class TPAVIModule(nn.Module):
    def __init__(self, in_channels: int, inter_channels: int | None = None, mode: str = "dot",
                 dimension: int = 3, bn_layer: bool = True) -> None:
        super().__init__()
        if mode not in ("dot", "embedded"):
            raise ValueError("oracle restates modes 'dot' and 'embedded' only")
        if dimension != 3 or not bn_layer:
            raise ValueError("oracle restates dimension=3, bn_layer=True only")
        self.mode, self.dimension = mode, dimension
        self.in_channels = in_channels
        self.inter_channels = inter_channels or max(in_channels // 2, 1)
        self.align_channel = nn.Linear(128, in_channels)       # ours.py:796 (dead for this model)
        self.norm_layer = nn.LayerNorm(in_channels)            # ours.py:797
        ci = self.inter_channels
        self.g = nn.Conv3d(in_channels, ci, 1)                 # ours.py:816
        self.W_z = nn.Sequential(nn.Conv3d(ci, in_channels, 1), nn.BatchNorm3d(in_channels))
        nn.init.zeros_(self.W_z[1].weight)                     # ours.py:826-827
        nn.init.zeros_(self.W_z[1].bias)
        self.theta = nn.Conv3d(in_channels, ci, 1)             # ours.py:836
        self.phi = nn.Conv3d(in_channels, ci, 1)               # ours.py:837

    def forward(self, x: torch.Tensor, audio=None):
        if audio is not None:
            raise ValueError("audio branch (ours.py:855-861) is not on the path")
        n = x.size(0)
        ci = self.inter_channels
        g_x = self.g(x).view(n, ci, -1).transpose(1, 2)            # [n, L, ci]   ours.py:866-869
        th = self.theta(x).view(n, ci, -1).transpose(1, 2)         # [n, L, ci]   ours.py:878,880
        ph = self.phi(x).view(n, ci, -1)                           # [n, ci, L]   ours.py:879
        f = torch.matmul(th, ph)                                   # [n, L, L]    ours.py:881
        if self.mode == "embedded":
            f = F.softmax(f, dim=-1)                               # ours.py:896-897
        else:
            f = f / f.size(-1)                                     # ours.py:898-900
        y = torch.matmul(f, g_x)                                   # [n, L, ci]   ours.py:902
        y = y.transpose(1, 2).contiguous().view(n, ci, *x.shape[2:])
        z = self.W_z(y) + x                                        # ours.py:908-910
        z = self.norm_layer(z.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)   # ours.py:913-915
        return z, 0


# --------------------------------------------------------------------------------------
# a1: Global_and_Local (models/ours.py:1708-1843)
# --------------------------------------------------------------------------------------
class Global_and_Local(nn.Module):
    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"),
                 center_aware_weight: float = 20) -> None:
        super().__init__()
        self.view_num = list(view_num)
        self.test_view = list(test_view)
        self.center_aware_weight = center_aware_weight
        self.network = deeplabv3_resnet50_iekd(pretrained=False, aux_loss=False)   # ours.py:1716
        self.init_block = nn.ModuleDict()
        self.layer1, self.layer2 = nn.ModuleDict(), nn.ModuleDict()
        self.layer3, self.layer4 = nn.ModuleDict(), nn.ModuleDict()
        self.classifier, self.centerness = nn.ModuleDict(), nn.ModuleDict()
        bb = self.network.backbone
        for v in self.view_num:                                                    # ours.py:1724-1744
            self.init_block[v] = copy.deepcopy(nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"]))
            self.layer1[v] = copy.deepcopy(bb["layer1"])
            self.layer2[v] = copy.deepcopy(bb["layer2"])
            self.layer3[v] = copy.deepcopy(bb["layer3"])
            self.layer4[v] = copy.deepcopy(bb["layer4"])
            self.classifier[v] = copy.deepcopy(self.network.classifier)
            self.classifier[v][-1] = nn.Conv2d(256, 5, kernel_size=1)
            self.centerness[v] = copy.deepcopy(self.network.classifier)
            self.centerness[v][-1] = nn.Conv2d(256, 1, kernel_size=1)
        self.global_attn = TPAVIModule(2048, mode="dot")                           # ours.py:1746
        self.local_attn = TPAVIModule(2048, mode="dot")                            # ours.py:1747

    def encode(self, x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        f4 = {}
        for v in self.view_num:                                                    # ours.py:1795-1800
            f = self.init_block[v](x[v])
            f = self.layer1[v](f)
            f = self.layer2[v](f)
            f = self.layer3[v](f)
            f4[v] = self.layer4[v](f)
        return f4

    def backbone(self, x):                                                         # ours.py:1749-1773
        hw = x[self.view_num[0]].shape[-2:]
        f4 = self.encode(x)
        mask = {v: F.interpolate(self.classifier[v](f4[v]), size=hw, mode="bilinear", align_corners=False)
                for v in self.view_num}
        return mask, f4

    def attend(self, block, stacked):
        return block(stacked)[0]

    def fuse(self, v, g, l):
        return g + l                                                               # ours.py:1833-1834

    def forward(self, x: Dict[str, torch.Tensor]):
        return self._forward_from_f4(x, self.encode(x))

    def _forward_from_f4(self, x, f4):
        hw = x[self.view_num[0]].shape[-2:]
        f4_local, f4_bg = {}, {}
        for v in self.view_num:
            # ours.py:1802-1807: AdaptiveMaxPool3d((1,h,w)) on a 4-D tensor == max over the class channels
            s = torch.sigmoid(self.classifier[v](f4[v]))
            m = F.adaptive_max_pool3d(s, (1, s.shape[2], s.shape[3]))   # same op => same tie-breaking in backward
            c = torch.sigmoid(self.centerness[v](f4[v]))                           # ours.py:1809-1811
            a = torch.sigmoid(self.center_aware_weight * m * c)                    # ours.py:1814-1815
            f4_local[v] = f4[v] * a                                                # ours.py:1816
            if getattr(self, "global_on_background", False):
                f4_bg[v] = f4[v] * (torch.ones_like(a) - a)                        # ours.py:2966
        g_in = f4_bg if f4_bg else f4
        g_out = self.attend(self.global_attn, torch.stack([g_in[v] for v in self.view_num], dim=2))      # ours.py:1819-1821
        l_out = self.attend(self.local_attn, torch.stack([f4_local[v] for v in self.view_num], dim=2))   # ours.py:1826-1828
        f4_g = {v: g_out[:, :, i] for i, v in enumerate(self.view_num)}
        f4_l = {v: l_out[:, :, i] for i, v in enumerate(self.view_num)}
        mask, mask_bb = {}, {}
        for v in self.view_num:                                                    # ours.py:1833-1841
            fused = self.fuse(v, f4_g[v], f4_l[v])
            mask[v] = F.interpolate(self.classifier[v](fused), size=hw, mode="bilinear", align_corners=False)
            mask_bb[v] = F.interpolate(self.classifier[v](f4[v]), size=hw, mode="bilinear", align_corners=False)
        return mask, mask_bb, f4_g, f4_l


class Global_and_Local_cyc_nofusion(Global_and_Local):
    """ours.py:2628-2764: Global_and_Local returning (mask, mask_bb, f4, f4_local_fusion)."""

    def forward(self, x):
        f4 = self.encode(x)
        mask, mask_bb, _, f4_l = self._forward_from_f4(x, f4)
        return mask, mask_bb, f4, f4_l


class Foreground_and_Background(Global_and_Local):
    """ours.py:2887-3024: global block on f4*(1-a), local block on f4*a; returns (mask, mask_bb, f4_fusion, None)."""
    global_on_background = True

    def forward(self, x):
        mask, mask_bb, f4_g, f4_l = self._forward_from_f4(x, self.encode(x))
        return mask, mask_bb, {v: f4_g[v] + f4_l[v] for v in self.view_num}, None


class Global_and_Local_Temporal(Global_and_Local):
    """ours.py:1846-1997 with the is_video branch as it is spelled out ([T,C,V,h,w] -> [1,C,T*V,h,w]; the shipped code
    calls `tensor.shape(...)` there, ours.py:1962, and raises).  No reference fixture can exist for is_video=True; for
    is_video=False it is Global_and_Local."""

    def forward(self, x, is_video: bool = False):
        self._video = bool(is_video)
        try:
            return self._forward_from_f4(x, self.encode(x))
        finally:
            self._video = False

    def attend(self, block, stacked):                                              # stacked [T,C,V,h,w]
        if not getattr(self, "_video", False):
            return block(stacked)[0]
        t, c, v, h, w = stacked.shape
        folded = stacked.permute(1, 0, 2, 3, 4).reshape(c, t * v, h, w).unsqueeze(0)            # ours.py:1960-1962
        out = block(folded)[0]
        return out.squeeze(0).reshape(c, t, v, h, w).permute(1, 0, 2, 3, 4)                     # ours.py:1965


class Global_and_Local_conv_merge(Global_and_Local):
    """ours.py:2766-2886: fusion = merge[v](cat([global, local], dim=1)) with merge = Conv2d(4096, 2048, 1) + ReLU."""

    def __init__(self, view_num, test_view=("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__(view_num, test_view, center_aware_weight)
        g, l = self.global_attn, self.local_attn
        del self.global_attn, self.local_attn
        self.merge = nn.ModuleDict({v: nn.Sequential(nn.Conv2d(2048 * 2, 2048, kernel_size=1), nn.ReLU()) for v in self.view_num})
        self.global_attn, self.local_attn = g, l                                   # registration order of ours.py:2781-2806

    def fuse(self, v, g, l):
        return self.merge[v](torch.cat([g, l], dim=1))                             # ours.py:2860-2862


class Global_only(Global_and_Local):
    """ours.py:1999-2111: no local branch; the centerness heads are constructed but unused."""

    def __init__(self, view_num, test_view=("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__(view_num, test_view, center_aware_weight)
        del self.local_attn                                                        # ours.py:2040-2041

    def forward(self, x):
        hw = x[self.view_num[0]].shape[-2:]
        f4 = self.encode(x)                                                        # ours.py:2085-2090
        g_out, _ = self.global_attn(torch.stack([f4[v] for v in self.view_num], dim=2))        # ours.py:2093-2095
        f4_g = {v: g_out[:, :, i] for i, v in enumerate(self.view_num)}
        mask, mask_bb = {}, {}
        for v in self.view_num:                                                    # ours.py:2103-2108
            mask[v] = F.interpolate(self.classifier[v](f4_g[v].contiguous()), size=hw, mode="bilinear", align_corners=False)
            mask_bb[v] = F.interpolate(self.classifier[v](f4[v]), size=hw, mode="bilinear", align_corners=False)
        return mask, mask_bb, f4_g, None


class Global_only_cyc_nofusion(Global_and_Local):
    """ours.py:3026-3139: Global_only's forward returning (mask, mask_bb, f4, None); its constructor still registers
    BOTH fusion blocks (ours.py:3065-3066), so `local_attn.*` is in the state_dict and never used."""

    def forward(self, x):
        hw = x[self.view_num[0]].shape[-2:]
        f4 = self.encode(x)
        g_out, _ = self.global_attn(torch.stack([f4[v] for v in self.view_num], dim=2))
        mask, mask_bb = {}, {}
        for i, v in enumerate(self.view_num):
            mask[v] = F.interpolate(self.classifier[v](g_out[:, :, i].contiguous()), size=hw, mode="bilinear", align_corners=False)
            mask_bb[v] = F.interpolate(self.classifier[v](f4[v]), size=hw, mode="bilinear", align_corners=False)
        return mask, mask_bb, f4, None


class Local_only(Global_and_Local):
    """ours.py:2113-2249: no global branch; returns (mask, mask_bb, atten_map, local fusion features)."""

    def __init__(self, view_num, test_view=("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__(view_num, test_view, center_aware_weight)
        del self.global_attn                                                       # ours.py:2151

    def forward(self, x):
        hw = x[self.view_num[0]].shape[-2:]
        f4 = self.encode(x)
        atten, f4_local = {}, {}
        for v in self.view_num:                                                    # ours.py:2209-2223
            s = torch.sigmoid(self.classifier[v](f4[v]))
            m = F.adaptive_max_pool3d(s, (1, s.shape[2], s.shape[3]))
            c = torch.sigmoid(self.centerness[v](f4[v]))
            atten[v] = torch.sigmoid(self.center_aware_weight * m * c)
            f4_local[v] = f4[v] * atten[v]
        l_out, _ = self.local_attn(torch.stack([f4_local[v] for v in self.view_num], dim=2))   # ours.py:2226-2228
        f4_l = {v: l_out[:, :, i] for i, v in enumerate(self.view_num)}
        mask, mask_bb = {}, {}
        for v in self.view_num:                                                    # ours.py:2242-2247
            mask[v] = F.interpolate(self.classifier[v](f4_l[v].contiguous()), size=hw, mode="bilinear", align_corners=False)
            mask_bb[v] = F.interpolate(self.classifier[v](f4[v]), size=hw, mode="bilinear", align_corners=False)
        return mask, mask_bb, atten, f4_l


class model19(nn.Module):
    """ours.py:976-1041: per-view encoders + ONE fusion block named `non_local` (no centre-ness heads, no gate);
    returns (mask, mask_bb, f4, f4_fusion)."""

    def __init__(self, view_num: Sequence[str], local_attn: bool = False, test_view: Sequence[str] = ("1", "2", "3", "4")) -> None:
        super().__init__()
        self.outchannel_list = {"1": 2, "2": 1, "3": 2, "4": 4}
        self.view_num, self.test_view, self.local_attn = view_num, test_view, local_attn
        self.network = deeplabv3_resnet50_iekd(pretrained=False, aux_loss=False)
        self.init_block = nn.ModuleDict()
        self.layer1, self.layer2 = nn.ModuleDict(), nn.ModuleDict()
        self.layer3, self.layer4 = nn.ModuleDict(), nn.ModuleDict()
        self.classifier = nn.ModuleDict()
        bb = self.network.backbone
        for v in self.view_num:                                                    # ours.py:990-1006
            self.init_block[v] = copy.deepcopy(nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"]))
            self.layer1[v] = copy.deepcopy(bb["layer1"])
            self.layer2[v] = copy.deepcopy(bb["layer2"])
            self.layer3[v] = copy.deepcopy(bb["layer3"])
            self.layer4[v] = copy.deepcopy(bb["layer4"])
            self.classifier[v] = copy.deepcopy(self.network.classifier)
            self.classifier[v][-1] = nn.Conv2d(256, 5, kernel_size=1)
        self.non_local = TPAVIModule(2048, mode="dot")                             # ours.py:1008

    def forward(self, x):
        hw = x[self.view_num[0]].shape[-2:]
        f4 = {}
        for v in self.view_num:                                                    # ours.py:1024-1029
            f = self.init_block[v](x[v])
            f = self.layer1[v](f)
            f = self.layer2[v](f)
            f = self.layer3[v](f)
            f4[v] = self.layer4[v](f)
        out, _ = self.non_local(torch.stack([f4[v] for v in self.view_num], dim=2))               # ours.py:1030-1033
        f4_fusion = {v: out[:, :, i] for i, v in enumerate(self.view_num)}
        mask, mask_bb = {}, {}
        for v in self.view_num:                                                    # ours.py:1036-1041
            mask[v] = F.interpolate(self.classifier[v](f4_fusion[v].contiguous()), size=hw, mode="bilinear", align_corners=False)
            mask_bb[v] = F.interpolate(self.classifier[v](f4[v].contiguous()), size=hw, mode="bilinear", align_corners=False)
        return mask, mask_bb, f4, f4_fusion


class Global_and_Local_CPS(nn.Module):
    """ours.py:3141-3349 (cross pseudo supervision): TWO Global_and_Local networks over the same input.  Network 1 owns
    deep copies of the template per view; network 2's encoders ARE the template's modules, shared by every view
    (ours.py:3192-3202: no deepcopy), so `init_block_2.<v>.*` / `layer*_2.<v>.*` alias `network.backbone.*` in the
    state_dict, their gradients sum over the views and their BatchNorm statistics are updated once per view.  Neither
    network evaluates the backbone-only mask; returns (mask, mask_2, f4_global_fusion, f4_local_fusion) of network 1."""

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__()
        self.outchannel_list = {"1": 2, "2": 1, "3": 2, "4": 4}
        self.view_num, self.test_view, self.center_aware_weight = view_num, test_view, center_aware_weight
        self.network = deeplabv3_resnet50_iekd(pretrained=False, aux_loss=False)
        for sfx in ("_1", "_2"):                                                   # registration order of ours.py:3150-3165
            for name in ("init_block", "layer1", "layer2", "layer3", "layer4", "classifier", "centerness"):
                setattr(self, name + sfx, nn.ModuleDict())
        bb = self.network.backbone
        for v in self.view_num:                                                    # ours.py:3167-3187
            self.init_block_1[v] = copy.deepcopy(nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"]))
            for l in ("layer1", "layer2", "layer3", "layer4"):
                getattr(self, l + "_1")[v] = copy.deepcopy(bb[l])
            self.classifier_1[v] = copy.deepcopy(self.network.classifier)
            self.classifier_1[v][-1] = nn.Conv2d(256, 5, kernel_size=1)
            self.centerness_1[v] = copy.deepcopy(self.network.classifier)
            self.centerness_1[v][-1] = nn.Conv2d(256, 1, kernel_size=1)
        self.global_attn_1 = TPAVIModule(2048, mode="dot")
        self.local_attn_1 = TPAVIModule(2048, mode="dot")
        for v in self.view_num:                                                    # ours.py:3192-3212: shared, not copied
            self.init_block_2[v] = nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"])
            for l in ("layer1", "layer2", "layer3", "layer4"):
                getattr(self, l + "_2")[v] = bb[l]
            self.classifier_2[v] = copy.deepcopy(self.network.classifier)
            self.classifier_2[v][-1] = nn.Conv2d(256, 5, kernel_size=1)
            self.centerness_2[v] = copy.deepcopy(self.network.classifier)
            self.centerness_2[v][-1] = nn.Conv2d(256, 1, kernel_size=1)
        self.global_attn_2 = TPAVIModule(2048, mode="dot")
        self.local_attn_2 = TPAVIModule(2048, mode="dot")

    def _net(self, x, sfx: str):
        hw = x[self.view_num[0]].shape[-2:]
        g = lambda name: getattr(self, name + sfx)
        f4, f4_local = {}, {}
        for v in self.view_num:
            f = g("init_block")[v](x[v])
            for l in ("layer1", "layer2", "layer3", "layer4"):
                f = g(l)[v](f)
            f4[v] = f
        for v in self.view_num:
            s = torch.sigmoid(g("classifier")[v](f4[v]))
            m = F.adaptive_max_pool3d(s, (1, s.shape[2], s.shape[3]))
            c = torch.sigmoid(g("centerness")[v](f4[v]))
            f4_local[v] = f4[v] * torch.sigmoid(self.center_aware_weight * m * c)
        g_out, _ = g("global_attn")(torch.stack([f4[v] for v in self.view_num], dim=2))
        l_out, _ = g("local_attn")(torch.stack([f4_local[v] for v in self.view_num], dim=2))
        f4_g = {v: g_out[:, :, i] for i, v in enumerate(self.view_num)}
        f4_l = {v: l_out[:, :, i] for i, v in enumerate(self.view_num)}
        mask = {v: F.interpolate(g("classifier")[v]((f4_g[v] + f4_l[v]).contiguous()), size=hw, mode="bilinear", align_corners=False)
                for v in self.view_num}
        return mask, f4_g, f4_l

    def forward(self, x):
        mask, f4_g, f4_l = self._net(x, "_1")
        mask_2, _, _ = self._net(x, "_2")
        return mask, mask_2, f4_g, f4_l


# --------------------------------------------------------------------------------------
# a8 / a9: the caller's step and metrics (main.py:87,202-243 and main.py:800-815)
# --------------------------------------------------------------------------------------
def overlap_metrics(gt: torch.Tensor, pred: torch.Tensor, eps: float = 1e-5):
    """main.py:800-815. Returns (pixel_acc, dice, precision, specificity, recall)."""
    o = pred.reshape(-1).float()
    t = gt.reshape(-1).float()
    tp = torch.sum(o * t)
    fp = torch.sum(o * (1 - t))
    fn = torch.sum((1 - o) * t)
    tn = torch.sum((1 - o) * (1 - t))
    return ((tp + tn) / (tp + tn + fp + fn + eps), 2 * tp / (2 * tp + fp + fn + eps),
            tp / (tp + fp + eps), tn / (tn + fp + eps), tp / (tp + fn + eps))


def binarize(logits: torch.Tensor) -> torch.Tensor:
    """main.py:250,385: torch.where(sigmoid(x) > 0.5, 1, 0)."""
    return torch.where(torch.sigmoid(logits) > 0.5, 1, 0)


def seg_loss(model: nn.Module, imgs: Dict[str, torch.Tensor], masks: Dict[str, torch.Tensor],
             views: Sequence[str] | None = None) -> torch.Tensor:
    """main.py:207-211: sum over views of BCEWithLogitsLoss(reduction='sum') on output 0."""
    pred = model(imgs)[0]
    bce = nn.BCEWithLogitsLoss(reduction="sum")
    return sum(bce(pred[v], masks[v]) for v in (views or list(pred.keys())))


def train_step(model: nn.Module, imgs, masks) -> float:
    """forward -> sum_v BCE-sum -> backward (no optimizer step), as the metric defines."""
    for p in model.parameters():
        p.grad = None
    loss = seg_loss(model, imgs, masks)
    loss.backward()
    return float(loss.detach())


# --------------------------------------------------------------------------------------
# Deterministic, RNG-free fills shared by the golden generator, the tests and smoke()
# --------------------------------------------------------------------------------------
def _hash_unit(n: int, salt: int) -> np.ndarray:
    """n values in [0,1) from an exact 64-bit integer hash of (index, salt); bit-identical
    on every platform (no libm)."""
    x = np.arange(n, dtype=np.uint64) + np.uint64((salt * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def closed_form_tensor(shape, salt: int, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = _hash_unit(n, salt)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(tuple(shape))


@torch.no_grad()
def closed_form_fill(module: nn.Module, salt: int = 0) -> None:
    """Fill state_dict() in key order with a closed-form rule so that fixtures need not
    store weights.  conv/linear weights: centred uniform with He-like scale; 1-D `weight`
    (norm gammas, incl. the zero-initialised W_z BN gamma so attention is live): 1 +- 0.1
    (0.5 +- 0.05 for the BNs feeding a residual join);
    biases / betas / running_mean: +-0.05; running_var: 1..1.25."""
    for k, (name, t) in enumerate(module.state_dict().items()):
        s = salt * 100003 + k + 1
        if name.endswith("num_batches_tracked"):
            t.zero_()
        elif name.endswith("running_var"):
            t.copy_(closed_form_tensor(t.shape, s, 1.0, 1.25))
        elif name.endswith("running_mean"):
            t.copy_(closed_form_tensor(t.shape, s, -0.05, 0.05))
        elif t.dim() >= 2:
            fan_in = int(np.prod(t.shape[1:]))
            b = float(np.sqrt(6.0 / fan_in))          # uniform(-b, b): var = 2/fan_in
            t.copy_(closed_form_tensor(t.shape, s, -b, b))
        elif name.endswith("bn3.weight") or name.endswith("downsample.1.weight"):
            # small gains on the two summands of every residual join keep the trunk's activations O(1)
            # through 16 blocks in eval() (running stats ~ N(0,1) do not renormalise), as in a trained net
            t.copy_(closed_form_tensor(t.shape, s, 0.45, 0.55))
        elif name.endswith("weight"):
            t.copy_(closed_form_tensor(t.shape, s, 0.9, 1.1))
        else:
            t.copy_(closed_form_tensor(t.shape, s, -0.05, 0.05))


@torch.no_grad()
def kinkfree_fill(module: nn.Module, salt: int = 0, offset: float = 6.0) -> None:
    """closed_form_fill, then every BatchNorm2d gets beta_c = +offset (even c) / -offset (odd c) and gamma in 1 +- 0.1.
    In train() a BatchNorm output is gamma * z + beta with z of exactly zero mean and unit variance per channel, so every
    ReLU input sits at +-offset +- ~1: no pre-activation comes within rounding distance of zero, no ReLU mask can differ
    between two floating-point evaluations (fp32 / fp64 / split-fp16), and channels of the same index keep the same sign
    across a residual join (identity and bn3 branch are both dead, or both alive).  Gradients of a step on such weights
    are smooth functions of the arithmetic: the reference's own fp32-vs-fp64 deviation is ~1e-6, which lets the parity
    tests gate every parameter gradient tightly instead of allowing for flipped masks."""
    closed_form_fill(module, salt)
    for k, m in enumerate(module.modules()):
        if isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            sign = torch.where(torch.arange(c) % 2 == 0, 1.0, -1.0)
            m.bias.copy_(sign * offset + closed_form_tensor((c,), salt * 7919 + k, -0.05, 0.05))
            m.weight.copy_(closed_form_tensor((c,), salt * 7927 + k, 0.9, 1.1))
        elif isinstance(m, nn.Conv2d) and m.out_channels == 1 and m.bias is not None:
            # the 1-channel centre-ness logit: small weights and bias -3 => c = sigmoid(.) ~ 0.05, which keeps the local
            # gate sigmoid(20 m c) (ours.py:1814) away from saturation: gradients do reach the centerness heads through it
            m.weight.mul_(0.02)
            m.bias.fill_(-3.0)


def varied_images(views: Sequence[str], n: int, h: int = 112, w: int = 112, salt: int = 13):
    """Frames that differ strongly from each other (per-frame gain 0.25 .. 1 and a per-frame gradient) so that statistics
    taken ACROSS frames -- the ASPP pooled branch normalises N frame averages per channel -- are well-conditioned."""
    out = {}
    yy = torch.linspace(0.0, 1.0, h).view(1, 1, h, 1)
    xx = torch.linspace(0.0, 1.0, w).view(1, 1, 1, w)
    for i, v in enumerate(views):
        base = closed_form_tensor((n, 1, h, w), salt * 1000 + i)
        f = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1)
        gain = 0.25 + 0.75 * ((f * 5 + i * 3) % n) / max(n - 1, 1)
        ramp = 0.5 * (torch.cos(f * 1.7 + i) * yy + torch.sin(f * 2.3 + i) * xx)
        out[v] = (gain * base + 0.25 * ramp + 0.25).clamp_(0.0, 1.0).contiguous()
    return out


def closed_form_images(views: Sequence[str], n: int, h: int = 112, w: int = 112, salt: int = 7):
    return {v: closed_form_tensor((n, 1, h, w), salt * 1000 + i) for i, v in enumerate(views)}


def closed_form_targets(views: Sequence[str], n: int, c: int = 5, h: int = 112, w: int = 112,
                        salt: int = 11, p: float = 0.3):
    return {v: (closed_form_tensor((n, c, h, w), salt * 1000 + i) < p).float()
            for i, v in enumerate(views)}


# ------------------------------------------------------------------------------------------------
# temporal cycle-consistency loss: Trainer.seg_cycle / dense_seg_cycle, main.py:650-798
# ------------------------------------------------------------------------------------------------
def _cycle_logits(feat: torch.Tensor, target_region: int, cyc_off: int, chunk_size: int, temperature: float, start: int):
    """The logits `q_similarity_averaged` of main.py:650-711 for one start frame, in index form.

    feat [T, F]: frames [0, R) are queries, [R, T) keys (main.py:651-653).  A chunk of `chunk_size` consecutive query
    frames starting at `start` is matched against every chunk of consecutive key frames (squared distance summed
    over the chunk, main.py:665-676), the matches are soft-maxed (main.py:678-679) into one weighted key chunk
    (main.py:684-691), which is then matched back against the chunks of query frames [cyc_off, R) (main.py:695-709)."""
    R, c, off = target_region, chunk_size, cyc_off
    T, F = feat.shape
    kn = T - R
    q_all, q_cyc, key = feat[:R], feat[off:R], feat[R:]
    query = q_all[start:start + c]                                            # main.py:659
    j = torch.arange(c)
    n_beta = kn - (c + off) + 1
    d = ((key[:, None, :] - query[None, :, :]) ** 2).sum(-1)                  # [kn, c]   main.py:665-667
    rows = (torch.arange(n_beta)[:, None] + j[None, :]) % kn                  # main.py:670-674 (first n_beta rows)
    sim = -d[rows, j[None, :]].sum(1)                                         # main.py:675-676
    beta = torch.softmax(sim / F / c * temperature, dim=0)                    # main.py:678-679
    rows_b = (torch.arange(off, kn - c + 1)[:, None] + j[None, :]) % kn       # main.py:684-688: chunks off .. kn-c
    weighted = (beta[:, None, None] * key[rows_b]).sum(0)                     # [c, F]   main.py:690-691
    rc = R - off
    n_out = rc - c + 1
    qd = ((q_cyc[:, None, :] - weighted[None, :, :]) ** 2).sum(-1)            # [rc, c]  main.py:695-697
    rows_q = (torch.arange(n_out)[:, None] + j[None, :]) % rc                 # main.py:699-703
    return -qd[rows_q, j[None, :]].sum(1) / F / c * temperature              # main.py:706-709


def seg_cycle(feat: torch.Tensor, target_region: int = 16, cyc_off: int = 2, chunk_size: int = 3, temperature: float = 10,
              start: int = 0) -> torch.Tensor:
    """main.py:650-718 with the random start frame (np.random.choice, main.py:655) as an explicit argument."""
    z = _cycle_logits(feat, target_region, cyc_off, chunk_size, temperature, start)
    target = torch.zeros_like(z)
    target[start] = 1.0                                                       # main.py:656
    return F.binary_cross_entropy_with_logits(z, target)                      # main.py:716 (mean)


def dense_seg_cycle(feat: torch.Tensor, target_region: int = 16, cyc_off: int = 2, chunk_size: int = 3, temperature: float = 10,
                    soft_label: bool = False, is_overlap: bool = True) -> torch.Tensor:
    """main.py:720-798: every start frame (stride chunk_size when not overlapping); the sum is divided by the number
    of POSSIBLE starts whatever the stride (main.py:798)."""
    n = target_region - (chunk_size + cyc_off) + 1
    total = 0
    for start in range(0, n, 1 if is_overlap else chunk_size):                # main.py:730
        z = _cycle_logits(feat, target_region, cyc_off, chunk_size, temperature, start)
        target = torch.zeros_like(z)
        target[start] = 1.0
        if soft_label:                                                        # main.py:790-791
            target = torch.where(target == 1, torch.full_like(z, 0.8), torch.full_like(z, 0.2 / (n - 1)))
        total = total + F.binary_cross_entropy_with_logits(z, target)
    return total / n


# ------------------------------------------------------------------------------------------------
# data path + evaluation harness: datasets/loader.py:298-330, 358-414, 460-498; main.py:484-543
# ------------------------------------------------------------------------------------------------
def mask_to_allclass(masks: torch.Tensor, view: str) -> torch.Tensor:
    """loader.py:358-414: the per-view part masks [k, ...] re-ordered into 5 class channels."""
    out = torch.zeros(5, *masks.shape[1:])
    if view in ("1", "3"):
        out[1] = masks[1]; out[3] = masks[0]
    elif view == "2":
        out[4] = masks[0]
    elif view == "4":
        out[0] = masks[2]; out[1] = masks[3]; out[2] = masks[1]; out[3] = masks[0]
    else:
        raise KeyError(view)
    return out


def prepare_clip(images: torch.Tensor, labels: torch.Tensor, view: str, crop_offset=None, labelled: bool = True):
    """The sample path of Seg_PAHDataset.__getitem__ for one view in torch CPU ops: AddChannel, Resized((144, 144, T),
    mode='nearest') == F.interpolate(mode='nearest') on the two spatial axes (the frame axis keeps its size), crop 112 x 112
    at `crop_offset` (centre when None: CenterSpatialCropd, loader.py:489), part masks (loader.py:298-316) ->
    mask_to_allclass, images / 255 (loader.py:327); then the frame reshape of main.py:495-499: [1,H,W,T] -> [T,1,H,W].
    images / labels: [H0, W0, T].  Returns (frames [T,1,112,112], masks [T,5,112,112])."""
    t = images.shape[-1]
    img = F.interpolate(images.permute(2, 0, 1).unsqueeze(1).float(), size=(144, 144), mode="nearest")     # [T,1,144,144]
    lab = F.interpolate(labels.permute(2, 0, 1).unsqueeze(1).float(), size=(144, 144), mode="nearest")
    oy, ox = (16, 16) if crop_offset is None else crop_offset
    img = img[:, :, oy:oy + 112, ox:ox + 112]
    lab = lab[:, 0, oy:oy + 112, ox:ox + 112]                                                               # [T,112,112]
    k = {"1": 2, "2": 1, "3": 2, "4": 4}[view]
    parts = torch.stack([torch.where(lab == c, 1, 0) for c in range(1, k + 1)], dim=0).float()             # loader.py:298-316
    masks = mask_to_allclass(parts, view).permute(1, 0, 2, 3).contiguous()                                  # [T,5,112,112]
    return (img / 255.0 if labelled else img).contiguous(), masks


def eval_metrics(all_pred: torch.Tensor, all_mask: torch.Tensor):
    """main.py:519 and 537-543: overlap metrics of the whole view, then Dice per part channel."""
    whole = overlap_metrics(all_mask, binarize(all_pred))
    parts = [overlap_metrics(all_mask[:, c], binarize(all_pred[:, c]))[1] for c in range(all_pred.shape[1])]
    return [float(x) for x in whole], [float(x) for x in parts]


def set_dropout(module: nn.Module, p: float) -> None:
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = p
