"""Global_and_Local / TPAVIModule of the reference (GLfusion/models/ours.py:1708-1843, 770-917)
with identical constructor signatures, return contracts and state_dict keys, every forward
running on the MI355X HIP engine (no torch compute kernels on the path)."""
from __future__ import annotations

import copy
from typing import Dict, Sequence

import os

import torch
from torch import nn

from .. import ops
from ..fusion import tpavi_forward
from .layers import Conv2d, ReLU, conv_bn_act, init_block_nhwc
from .segmentation import deeplabv3_resnet50_iekd


# the two fusion blocks on side streams of their own (GLF_FUSION_STREAMS=0: one after the other on the current stream)
_FUSION_STREAMS = os.environ.get("GLF_FUSION_STREAMS", "1") != "0"
# classifier and centerness head of a view on streams of their own (inside the view's section)
_HEAD_STREAMS = os.environ.get("GLF_HEAD_STREAMS", "0") != "0"
_HEAD_ORDER = os.environ.get("GLF_HEAD_ORDER", "0") != "0"
# the global fusion block beside the views' head sections instead of beside the local block
_EARLY_GLOBAL = os.environ.get("GLF_EARLY_GLOBAL", "0") != "0"

class TPAVIModule(nn.Module):
    """ours.py:770-917.  Built modes: 'dot' (shipped) and 'embedded' (softmax); dimension=3,
    bn_layer=True.  forward(x [N,C,V,h,w]) -> (z [N,C,V,h,w], audio_temp=0)."""

    def __init__(self, in_channels: int, inter_channels=None, mode: str = "dot", dimension: int = 3, bn_layer: bool = True) -> None:
        super().__init__()
        if mode not in ("gaussian", "embedded", "dot", "concatenate"):
            raise ValueError("`mode` must be one of `gaussian`, `embedded`, `dot` or `concatenate`")
        if mode not in ("dot", "embedded") or dimension != 3 or not bn_layer:
            raise NotImplementedError("glfusion_amd builds TPAVIModule for mode in {'dot','embedded'}, dimension=3, bn_layer=True")
        self.mode, self.dimension = mode, dimension
        self.in_channels = in_channels
        self.inter_channels = inter_channels
        if self.inter_channels is None:
            self.inter_channels = in_channels // 2 or 1
        ci = self.inter_channels
        # registration order == the reference's (state_dict key order): align_channel, norm_layer, g, W_z, theta, phi
        self.align_channel = nn.Linear(128, in_channels)        # dead for this model (audio branch)
        self.norm_layer = nn.LayerNorm(in_channels)
        self.g = nn.Conv3d(in_channels, ci, kernel_size=1)
        self.W_z = nn.Sequential(nn.Conv3d(ci, in_channels, kernel_size=1), nn.BatchNorm3d(in_channels))
        nn.init.constant_(self.W_z[1].weight, 0)
        nn.init.constant_(self.W_z[1].bias, 0)
        self.theta = nn.Conv3d(in_channels, ci, kernel_size=1)
        self.phi = nn.Conv3d(in_channels, ci, kernel_size=1)

    def forward_nvhwc(self, x5: torch.Tensor) -> torch.Tensor:
        """x5: [N, V, h, w, C] channels-last."""
        return tpavi_forward(x5, self)

    def forward(self, x: torch.Tensor, audio=None):
        if audio is not None:
            raise NotImplementedError("the audio branch (ours.py:855-861) is not on the path")
        ops._chk(x, "TPAVI input")
        x5 = x.permute(0, 2, 3, 4, 1)
        x5 = x5 if x5.is_contiguous() else x5.contiguous()
        return self.forward_nvhwc(x5).permute(0, 4, 1, 2, 3), 0


class _PerViewNetworks(nn.Module):
    """The constructor the reference repeats in every multi-view variant (ours.py:1709-1744, 2000-2037, 2114-2149):
    per view a copy of the stem, layer1-4, a 5-class `classifier` head and a 1-class `centerness` head, all cut out of
    one `network` template (which stays registered, so its dead parameters are part of the state_dict)."""

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__()
        self.outchannel_list = {"1": 2, "2": 1, "3": 2, "4": 4}
        self.view_num = view_num
        self.test_view = test_view
        self.center_aware_weight = center_aware_weight
        self.network = deeplabv3_resnet50_iekd(pretrained=False, aux_loss=False)
        self.init_block = nn.ModuleDict()
        self.layer1 = nn.ModuleDict()
        self.layer2 = nn.ModuleDict()
        self.layer3 = nn.ModuleDict()
        self.layer4 = nn.ModuleDict()
        self.classifier = nn.ModuleDict()
        self.centerness = nn.ModuleDict()
        bb = self.network.backbone
        for view in self.view_num:
            self.init_block[view] = copy.deepcopy(nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"]))
            self.layer1[view] = copy.deepcopy(bb["layer1"])
            self.layer2[view] = copy.deepcopy(bb["layer2"])
            self.layer3[view] = copy.deepcopy(bb["layer3"])
            self.layer4[view] = copy.deepcopy(bb["layer4"])
            self.classifier[view] = copy.deepcopy(self.network.classifier)
            last = self.network.classifier[-1]
            self.classifier[view][-1] = Conv2d(last.in_channels, 5, kernel_size=last.kernel_size)
            self.centerness[view] = copy.deepcopy(self.network.classifier)
            self.centerness[view][-1] = Conv2d(last.in_channels, 1, kernel_size=last.kernel_size)

    # -- encoder (ours.py:1795-1800) -------------------------------------------------------
    def _encode_view(self, view: str, xv: torch.Tensor) -> torch.Tensor:
        blk = self.init_block[view]
        f = init_block_nhwc(ops.to_nhwc(xv), blk[0], blk[1], blk[3])
        f = self.layer1[view].forward_nhwc(f)
        f = self.layer2[view].forward_nhwc(f, sole_reader=True)        # each stage output has one reader: the next stage
        f = self.layer3[view].forward_nhwc(f, sole_reader=True)
        return self.layer4[view].forward_nhwc(f, sole_reader=True)

    def _encode(self, x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        views = list(self.view_num)
        outs = ops.parallel_sections([lambda v=v: self._encode_view(v, x[v]) for v in views])   # views are independent
        return dict(zip(views, outs))

    def backbone(self, x):
        """ours.py:1749-1773: per-view encoder + classifier, no fusion."""
        hw = x[self.view_num[0]].shape[-2:]
        f4 = self._encode(x)
        mask = {v: ops.bilinear_up(self.classifier[v].forward_nhwc(f4[v]), int(hw[0]), int(hw[1])) for v in self.view_num}
        return mask, {v: ops.from_nhwc(f) for v, f in f4.items()}



class Global_and_Local(_PerViewNetworks):
    """ours.py:1708-1843."""
    _third_output_is_f4 = False
    _global_gets_background = False

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__(view_num, test_view, center_aware_weight)
        self.global_attn = TPAVIModule(in_channels=2048, mode="dot")
        self.local_attn = TPAVIModule(in_channels=2048, mode="dot")

    def _attend(self, block, stacked):
        """One fusion block over the stacked views [N,V,h,w,C] (ours.py:1819-1830): attention among the V*h*w positions
        of each frame."""
        return block.forward_nvhwc(stacked)

    def _fuse(self, g_out, l_out):
        """f4_fusion[v] = global + local (ours.py:1833-1834); returns a function (view index, view) -> [N,h,w,C]."""
        fused = ops.add_views(g_out, l_out)
        return lambda i, v: fused[i]

    def forward(self, x: Dict[str, torch.Tensor]):
        views = list(self.view_num)
        hw = x[views[0]].shape[-2:]
        ho, wo = int(hw[0]), int(hw[1])
        ops.begin_step(x[views[0]].device)

        # per view: encoder, then M_cls, M_ctr and the gated local features (ours.py:1795-1816)
        def view_section(v):
            f = self._encode_view(v, x[v])
            fa, fb, fc, fg, *raw = ops.fan_out(f, 5 if self._third_output_is_f4 else 4)   # classifier / centerness / gate / global fusion
            if _HEAD_STREAMS:            # the two heads of a view are independent chains of ~25 kernels each
                (cls, again), ctr = ops.parallel_sections([lambda: self.classifier[v].forward_nhwc_shared(fa),
                                                           lambda: self.centerness[v].forward_nhwc(fb)])
            elif _HEAD_ORDER and views.index(v) % 2 == 1:
                # the views' chains are copies of each other and run in lockstep -- contraction phases and streaming phases line
                # up across the three streams; every other view evaluates its two (independent) heads in the opposite order
                ctr = self.centerness[v].forward_nhwc(fb)
                cls, again = self.classifier[v].forward_nhwc_shared(fa)
            else:
                cls, again = self.classifier[v].forward_nhwc_shared(fa)     # `again`: the mask_bb call below, same input
                ctr = self.centerness[v].forward_nhwc(fb)
            gated = ops.local_gate(cls, ctr, fc, self.center_aware_weight)
            if self._global_gets_background:                            # Foreground_and_Background (ours.py:2966)
                g1, g2 = ops.fan_out(gated, 2)
                fg, gated = ops.axpby(fg, g1, 1.0, -1.0), g2            # f4 * (1 - a) = f4 - f4 * a
            return again, fg, gated, (raw[0] if raw else None)

        if _EARLY_GLOBAL and not self._global_gets_background:
            # The global fusion block needs only the encoders' outputs: it runs NEXT TO the views' head + gate sections (four
            # independent chains) instead of after them next to the local block.  The fusion blocks are the part of the step
            # with the least to overlap with -- two chains of large contractions whose streaming kernels (stack / split /
            # statistics / LayerNorm tail) ran with nothing but their twin beside them (profiles/r03_timeline_step.txt).
            k = 5 if self._third_output_is_f4 else 4
            enc = ops.parallel_sections([lambda v=v: ops.fan_out(self._encode_view(v, x[v]), k) for v in views])

            def heads_gate(v, parts):
                ops.use_here(*parts)
                fa, fb, fc = parts[0], parts[1], parts[2]
                cls, again = self.classifier[v].forward_nhwc_shared(fa)
                ctr = self.centerness[v].forward_nhwc(fb)
                return again, ops.local_gate(cls, ctr, fc, self.center_aware_weight)

            def global_block():
                fg = [e[3] for e in enc]
                ops.use_here(*fg)
                return self._attend(self.global_attn, ops.stack_views(fg))
            res = ops.parallel_sections([lambda v=v, e=e: heads_gate(v, e) for v, e in zip(views, enc)] + [global_block])
            g_out = res[-1]
            cls_again = {v: r[0] for v, r in zip(views, res)}
            f4_local = {v: r[1] for v, r in zip(views, res)}
            secs = [(None, None, None, (e[4] if k == 5 else None)) for e in enc]
            l_out = self._attend(self.local_attn, ops.stack_views([f4_local[v] for v in views]))
        else:
            secs = ops.parallel_sections([lambda v=v: view_section(v) for v in views])
            cls_again = {v: s[0] for v, s in zip(views, secs)}
            f4_glob = {v: s[1] for v, s in zip(views, secs)}
            f4_local = {v: s[2] for v, s in zip(views, secs)}
            # global / local cross-view fusion (ours.py:1819-1830): two independent blocks
            fusion_jobs = [
                lambda: self._attend(self.global_attn, ops.stack_views([f4_glob[v] for v in views])),    # [N,V,h,w,C]
                lambda: self._attend(self.local_attn, ops.stack_views([f4_local[v] for v in views]))]
            g_out, l_out = ops.parallel_sections(fusion_jobs) if _FUSION_STREAMS else [j() for j in fusion_jobs]
        fused = self._fuse(g_out, l_out)                                                        # ours.py:1833-1834

        def head_section(i, v):       # same order per view as the reference: fused mask first, backbone mask second
            m = ops.bilinear_up(self.classifier[v].forward_nhwc(fused(i, v)), ho, wo)           # ours.py:1837-1838
            mb = ops.bilinear_up(self.classifier[v].forward_nhwc_replay(cls_again[v]), ho, wo)  # ours.py:1840-1841, on f4[v]
            return m, mb

        heads = ops.parallel_sections([lambda i=i, v=v: head_section(i, v) for i, v in enumerate(views)])
        mask, mask_bb, f4_g, f4_l = {}, {}, {}, {}
        for i, v in enumerate(views):
            f4_g[v] = g_out[:, i].permute(0, 3, 1, 2)       # == global_conv_feat[:, :, i, :, :]
            f4_l[v] = l_out[:, i].permute(0, 3, 1, 2)
            f4_g[v]._glf_stack = (g_out, i)                 # lets ops.pooled_fusion_features pool the block once
            f4_l[v]._glf_stack = (l_out, i)
            mask[v], mask_bb[v] = heads[i]
        if self._third_output_is_f4:                                   # Global_and_Local_cyc_nofusion (ours.py:2764)
            return mask, mask_bb, {v: ops.from_nhwc(s[3]) for v, s in zip(views, secs)}, f4_l
        return mask, mask_bb, f4_g, f4_l


class Global_and_Local_cyc_nofusion(Global_and_Local):
    """ours.py:2628-2764: the same network; returns the un-fused layer4 features as third output (the cycle loss of that
    experiment is taken on them): (mask, mask_bb, f4, f4_local_fusion)."""
    _third_output_is_f4 = True


class Foreground_and_Background(Global_and_Local):
    """ours.py:2887-3024: the gate splits f4 into foreground f4*a (local fusion block) and background f4*(1-a) (global
    fusion block, instead of the un-gated f4); returns (mask, mask_bb, f4_fusion, None) with f4_fusion = the sum of the
    two blocks' outputs."""
    _global_gets_background = True

    def forward(self, x: Dict[str, torch.Tensor]):
        mask, mask_bb, f4_g, f4_l = super().forward(x)
        g_out, l_out = f4_g[self.view_num[0]]._glf_stack[0], f4_l[self.view_num[0]]._glf_stack[0]
        fused = ops.add_views(g_out, l_out)                              # ours.py:3013-3014 (returned as features)
        return mask, mask_bb, {v: ops.from_nhwc(fused[i]) for i, v in enumerate(self.view_num)}, None


class Global_and_Local_Temporal(Global_and_Local):
    """ours.py:1846-1997: forward(x, is_video).  With is_video the T frames of a clip are folded into the attention
    axis -- one "frame" of T*V*h*w positions (L = 37 632 at T = 16) -- so every position attends across views AND time.
    As shipped that branch cannot run (`tensor.shape(c, t*v, h, w)` at ours.py:1962 / 1975 calls a torch.Size); built
    here is what it spells out: [T,C,V,h,w] -> [1,C,T*V,h,w] -> block -> back.  In channels-last terms the stacked
    [T,V,h,w,C] tensor is already in that order, so it is a view; the re-associated dot attention makes the long axis
    free (M = phi^T g / L is still [Ci,Ci]; BatchNorm3d / LayerNorm see the same rows as before)."""

    def forward(self, x: Dict[str, torch.Tensor], is_video: bool = False):
        self._is_video = bool(is_video)
        try:
            return super().forward(x)
        finally:
            self._is_video = False

    def _attend(self, block, stacked):
        if not getattr(self, "_is_video", False):
            return block.forward_nvhwc(stacked)
        t, v, h, w, c = stacked.shape
        return block.forward_nvhwc(stacked.reshape(1, t * v, h, w, c)).reshape(t, v, h, w, c)


class Global_and_Local_conv_merge(Global_and_Local):
    """ours.py:2766-2886: the two fusion outputs are merged by a learned per-view `merge` = Conv2d(4096, 2048, 1) + ReLU
    over their channel concatenation instead of being added.  The concatenation is never materialised (the 1x1 conv
    runs as two accumulated K = 2048 contractions over the two fusion blocks' per-view slices)."""

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        _PerViewNetworks.__init__(self, view_num, test_view, center_aware_weight)
        self.merge = nn.ModuleDict()                                    # registered before the fusion blocks, as in the reference
        for view in self.view_num:
            self.merge[view] = nn.Sequential(Conv2d(2048 * 2, 2048, kernel_size=1), ReLU())
        self.global_attn = TPAVIModule(in_channels=2048, mode="dot")
        self.local_attn = TPAVIModule(in_channels=2048, mode="dot")

    def _fuse(self, g_out, l_out):
        gs, ls = ops.split_views(g_out), ops.split_views(l_out)

        def fused(i, v):                                                # ours.py:2860-2862: cat(dim=1) -> merge
            conv = self.merge[v][0]
            return ops.relu(ops.conv1x1_cat(conv.weight, [gs[i], ls[i]], conv.bias))
        return fused


class Global_only(_PerViewNetworks):
    """Ablation without the local branch (ours.py:1999-2111): mask = classifier(global fusion), mask_bb =
    classifier(f4); the centerness heads exist (state_dict) but are not evaluated.  Returns
    (mask, mask_bb, f4_global_fusion, None)."""
    _third_output_is_f4 = False

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__(view_num, test_view, center_aware_weight)
        self.global_attn = TPAVIModule(in_channels=2048, mode="dot")

    def forward(self, x: Dict[str, torch.Tensor]):
        views = list(self.view_num)
        ops.begin_step(x[views[0]].device)
        hw = x[views[0]].shape[-2:]
        ho, wo = int(hw[0]), int(hw[1])

        def view_section(v):                                 # ours.py:2085-2090
            return ops.fan_out(self._encode_view(v, x[v]), 3 if self._third_output_is_f4 else 2)   # global fusion / mask_bb (/ raw f4)

        secs = ops.parallel_sections([lambda v=v: view_section(v) for v in views])
        g_out = self.global_attn.forward_nvhwc(ops.stack_views([s[0] for s in secs]))            # ours.py:2093-2097
        per_view = ops.split_views(g_out)

        def head_section(i, v):                               # ours.py:2103-2108: fused mask first, backbone mask second
            m = ops.bilinear_up(self.classifier[v].forward_nhwc(per_view[i]), ho, wo)
            mb = ops.bilinear_up(self.classifier[v].forward_nhwc(secs[i][1]), ho, wo)
            return m, mb

        heads = ops.parallel_sections([lambda i=i, v=v: head_section(i, v) for i, v in enumerate(views)])
        mask, mask_bb, f4_g = {}, {}, {}
        for i, v in enumerate(views):
            f4_g[v] = g_out[:, i].permute(0, 3, 1, 2)
            f4_g[v]._glf_stack = (g_out, i)
            mask[v], mask_bb[v] = heads[i]
        if self._third_output_is_f4:                                   # Global_only_cyc_nofusion (ours.py:3139)
            return mask, mask_bb, {v: ops.from_nhwc(s[2]) for v, s in zip(views, secs)}, None
        return mask, mask_bb, f4_g, None


class Global_only_cyc_nofusion(Global_only):
    """ours.py:3026-3139: Global_only returning the un-fused layer4 features as third output: (mask, mask_bb, f4, None).
    Its constructor registers BOTH fusion blocks (ours.py:3065-3066): `local_attn.*` is in the state_dict, unused."""
    _third_output_is_f4 = True

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__(view_num, test_view, center_aware_weight)
        self.local_attn = TPAVIModule(in_channels=2048, mode="dot")


class Local_only(_PerViewNetworks):
    """Ablation without the global branch (ours.py:2113-2249): the centre-aware gate, local fusion only.  Returns
    (mask, mask_bb, atten_map, f4_fusion) with atten_map[v] = sigmoid(w * max_c sigmoid(cls) * sigmoid(ctr)) [N,1,h,w]
    (returned detached: nothing on the reference's path differentiates through the returned map) and
    f4_fusion[v] the local-fusion features."""

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__(view_num, test_view, center_aware_weight)
        self.local_attn = TPAVIModule(in_channels=2048, mode="dot")

    def forward(self, x: Dict[str, torch.Tensor]):
        views = list(self.view_num)
        ops.begin_step(x[views[0]].device)
        hw = x[views[0]].shape[-2:]
        ho, wo = int(hw[0]), int(hw[1])

        def view_section(v):                                  # ours.py:2202-2223
            f = self._encode_view(v, x[v])
            fa, fb, fc = ops.fan_out(f, 3)                    # classifier / centerness / gate
            cls, again = self.classifier[v].forward_nhwc_shared(fa)       # `again`: the mask_bb call on the same f4
            ctr = self.centerness[v].forward_nhwc(fb)
            gated, amap = ops.local_gate_with_map(cls, ctr, fc, self.center_aware_weight)
            return again, gated, amap

        secs = ops.parallel_sections([lambda v=v: view_section(v) for v in views])
        l_out = self.local_attn.forward_nvhwc(ops.stack_views([s[1] for s in secs]))             # ours.py:2226-2231
        per_view = ops.split_views(l_out)

        def head_section(i, v):                               # ours.py:2242-2247
            m = ops.bilinear_up(self.classifier[v].forward_nhwc(per_view[i]), ho, wo)
            mb = ops.bilinear_up(self.classifier[v].forward_nhwc_replay(secs[i][0]), ho, wo)
            return m, mb

        heads = ops.parallel_sections([lambda i=i, v=v: head_section(i, v) for i, v in enumerate(views)])
        mask, mask_bb, atten, f4_l = {}, {}, {}, {}
        for i, v in enumerate(views):
            f4_l[v] = l_out[:, i].permute(0, 3, 1, 2)
            f4_l[v]._glf_stack = (l_out, i)
            atten[v] = secs[i][2]
            mask[v], mask_bb[v] = heads[i]
        return mask, mask_bb, atten, f4_l


class model19(nn.Module):
    """ours.py:976-1041: per-view encoders and classifier heads (no centre-ness heads, no gate) around ONE fusion block
    registered as `non_local`; returns (mask, mask_bb, f4, f4_fusion)."""

    def __init__(self, view_num: Sequence[str], local_attn: bool = False, test_view: Sequence[str] = ("1", "2", "3", "4")) -> None:
        super().__init__()
        self.outchannel_list = {"1": 2, "2": 1, "3": 2, "4": 4}
        self.view_num, self.test_view, self.local_attn = view_num, test_view, local_attn
        self.network = deeplabv3_resnet50_iekd(pretrained=False, aux_loss=False)
        self.init_block = nn.ModuleDict()
        self.layer1 = nn.ModuleDict()
        self.layer2 = nn.ModuleDict()
        self.layer3 = nn.ModuleDict()
        self.layer4 = nn.ModuleDict()
        self.classifier = nn.ModuleDict()
        bb = self.network.backbone
        for view in self.view_num:                                       # ours.py:990-1006
            self.init_block[view] = copy.deepcopy(nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"]))
            self.layer1[view] = copy.deepcopy(bb["layer1"])
            self.layer2[view] = copy.deepcopy(bb["layer2"])
            self.layer3[view] = copy.deepcopy(bb["layer3"])
            self.layer4[view] = copy.deepcopy(bb["layer4"])
            self.classifier[view] = copy.deepcopy(self.network.classifier)
            last = self.network.classifier[-1]
            self.classifier[view][-1] = Conv2d(last.in_channels, 5, kernel_size=last.kernel_size)
        self.non_local = TPAVIModule(in_channels=2048, mode="dot")      # ours.py:1008

    _encode_view = _PerViewNetworks._encode_view

    def forward(self, x: Dict[str, torch.Tensor]):
        views = list(self.view_num)
        ops.begin_step(x[views[0]].device)
        hw = x[views[0]].shape[-2:]
        ho, wo = int(hw[0]), int(hw[1])
        secs = ops.parallel_sections([lambda v=v: ops.fan_out(self._encode_view(v, x[v]), 3) for v in views])   # fusion / mask_bb / returned f4
        out = self.non_local.forward_nvhwc(ops.stack_views([s[0] for s in secs]))                               # ours.py:1030-1033
        per_view = ops.split_views(out)

        def head_section(i, v):                                          # ours.py:1036-1041
            m = ops.bilinear_up(self.classifier[v].forward_nhwc(per_view[i]), ho, wo)
            mb = ops.bilinear_up(self.classifier[v].forward_nhwc(secs[i][1]), ho, wo)
            return m, mb

        heads = ops.parallel_sections([lambda i=i, v=v: head_section(i, v) for i, v in enumerate(views)])
        mask, mask_bb, f4, f4_fusion = {}, {}, {}, {}
        for i, v in enumerate(views):
            mask[v], mask_bb[v] = heads[i]
            f4[v] = ops.from_nhwc(secs[i][2])
            f4_fusion[v] = out[:, i].permute(0, 3, 1, 2)
            f4_fusion[v]._glf_stack = (out, i)
        return mask, mask_bb, f4, f4_fusion


class Global_and_Local_CPS(nn.Module):
    """ours.py:3141-3349 (cross pseudo supervision): two Global_and_Local networks over the same input.  Network 1 owns
    per-view deep copies of the template; network 2's encoders ARE the template's modules, shared by all views
    (ours.py:3192-3202 registers them without deepcopy): `init_block_2.<v>.*` / `layer*_2.<v>.*` alias
    `network.backbone.*` in the state_dict, gradients sum over the views, and the shared BatchNorm layers update their
    running statistics once per view -- so network 2's encoders run view after view on one stream, in the reference's
    order.  Neither network evaluates the backbone-only mask.  Returns (mask, mask_2, f4_global_fusion, f4_local_fusion)."""

    def __init__(self, view_num: Sequence[str], test_view: Sequence[str] = ("1", "2", "3", "4"), center_aware_weight: float = 20) -> None:
        super().__init__()
        self.outchannel_list = {"1": 2, "2": 1, "3": 2, "4": 4}
        self.view_num, self.test_view, self.center_aware_weight = view_num, test_view, center_aware_weight
        self.network = deeplabv3_resnet50_iekd(pretrained=False, aux_loss=False)
        for sfx in ("_1", "_2"):                                         # registration order of ours.py:3150-3165
            for name in ("init_block", "layer1", "layer2", "layer3", "layer4", "classifier", "centerness"):
                setattr(self, name + sfx, nn.ModuleDict())
        bb = self.network.backbone
        last = self.network.classifier[-1]
        for view in self.view_num:                                       # ours.py:3167-3187
            self.init_block_1[view] = copy.deepcopy(nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"]))
            for l in ("layer1", "layer2", "layer3", "layer4"):
                getattr(self, l + "_1")[view] = copy.deepcopy(bb[l])
            self.classifier_1[view] = copy.deepcopy(self.network.classifier)
            self.classifier_1[view][-1] = Conv2d(last.in_channels, 5, kernel_size=last.kernel_size)
            self.centerness_1[view] = copy.deepcopy(self.network.classifier)
            self.centerness_1[view][-1] = Conv2d(last.in_channels, 1, kernel_size=last.kernel_size)
        self.global_attn_1 = TPAVIModule(in_channels=2048, mode="dot")
        self.local_attn_1 = TPAVIModule(in_channels=2048, mode="dot")
        for view in self.view_num:                                       # ours.py:3192-3212: shared with the template, not copied
            self.init_block_2[view] = nn.Sequential(bb["conv1"], bb["bn1"], bb["relu"], bb["maxpool"])
            for l in ("layer1", "layer2", "layer3", "layer4"):
                getattr(self, l + "_2")[view] = bb[l]
            self.classifier_2[view] = copy.deepcopy(self.network.classifier)
            self.classifier_2[view][-1] = Conv2d(last.in_channels, 5, kernel_size=last.kernel_size)
            self.centerness_2[view] = copy.deepcopy(self.network.classifier)
            self.centerness_2[view][-1] = Conv2d(last.in_channels, 1, kernel_size=last.kernel_size)
        self.global_attn_2 = TPAVIModule(in_channels=2048, mode="dot")
        self.local_attn_2 = TPAVIModule(in_channels=2048, mode="dot")

    def _encode(self, sfx: str, view: str, xv: torch.Tensor) -> torch.Tensor:
        g = lambda name: getattr(self, name + sfx)[view]
        blk = g("init_block")
        f = init_block_nhwc(ops.to_nhwc(xv), blk[0], blk[1], blk[3])
        for l in ("layer1", "layer2", "layer3", "layer4"):
            f = g(l).forward_nhwc(f, sole_reader=l != "layer1")         # a stage output has one reader: the next stage
        return f

    def _net(self, x, sfx: str, shared_encoder: bool):
        views = list(self.view_num)
        hw = x[views[0]].shape[-2:]
        ho, wo = int(hw[0]), int(hw[1])
        g = lambda name: getattr(self, name + sfx)

        def heads_and_gate(v, f):
            fa, fb, fc, fg = ops.fan_out(f, 4)                          # classifier / centerness / gate / global fusion
            cls = g("classifier")[v].forward_nhwc(fa)
            ctr = g("centerness")[v].forward_nhwc(fb)
            return fg, ops.local_gate(cls, ctr, fc, self.center_aware_weight)

        if shared_encoder:               # one set of modules for every view: sequential, the reference's view order
            f4 = [self._encode(sfx, v, x[v]) for v in views]
            secs = ops.parallel_sections([lambda v=v, f=f: heads_and_gate(v, f) for v, f in zip(views, f4)])
        else:
            secs = ops.parallel_sections([lambda v=v: heads_and_gate(v, self._encode(sfx, v, x[v])) for v in views])
        g_out, l_out = ops.parallel_sections([
            lambda: g("global_attn").forward_nvhwc(ops.stack_views([s[0] for s in secs])),
            lambda: g("local_attn").forward_nvhwc(ops.stack_views([s[1] for s in secs]))])
        fused = ops.add_views(g_out, l_out)
        masks = ops.parallel_sections([lambda i=i, v=v: ops.bilinear_up(g("classifier")[v].forward_nhwc(fused[i]), ho, wo)
                                       for i, v in enumerate(views)])
        return dict(zip(views, masks)), g_out, l_out

    def forward(self, x: Dict[str, torch.Tensor]):
        ops.begin_step(x[self.view_num[0]].device)
        mask, g_out, l_out = self._net(x, "_1", shared_encoder=False)
        mask_2, _, _ = self._net(x, "_2", shared_encoder=True)
        f4_g, f4_l = {}, {}
        for i, v in enumerate(self.view_num):
            f4_g[v] = g_out[:, i].permute(0, 3, 1, 2)
            f4_l[v] = l_out[:, i].permute(0, 3, 1, 2)
            f4_g[v]._glf_stack = (g_out, i)
            f4_l[v]._glf_stack = (l_out, i)
        return mask, mask_2, f4_g, f4_l
