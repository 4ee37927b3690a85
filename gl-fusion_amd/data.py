"""GPU side of the reference's data path (datasets/loader.py:190-498).

The reference loads a NIfTI volume per (patient, view) with nibabel, runs a MONAI transform chain on the CPU
(AddChannel -> Resized(144, 144, 'nearest') -> RandSpatialCrop / CenterSpatialCrop(112, 112) -> EnsureType), turns the
label map into per-view part masks re-ordered into 5 class channels (loader.py:298-316, 358-414) and scales the image by
1/255 (loader.py:325-327).  Here the raw volume goes to the GPU once and ONE kernel (glf_prepare_frames) writes the model's
input layout directly: frames [T,1,112,112] and masks [T,5,112,112] -- the `[1,1,H,W,T] -> permute -> reshape(-1,1,H,W)`
of main.py:361-365 / 495-499 included.  NIfTI files themselves are not shipped with the reference (absolute paths on the
authors' machine); `SyntheticPatients` produces volumes with the same tensor contract.
(The RandFlipd objects of loader.py:468-474 are constructed but never put into the Compose: no flip is applied.)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch

from ._lib import check, lib
from .ops import _chk, _contig, _p, _stream

# class id (1..4) -> channel of the 5-channel mask, per view: loader.py:298-316 (which parts a view shows) composed with
# mask_to_allclass (loader.py:358-414)
CLASS_TO_CHANNEL: Dict[str, Tuple[int, int, int, int]] = {
    "1": (3, 1, -1, -1),        # parasternal long axis: LV -> 3, RV -> 1
    "2": (4, -1, -1, -1),       # pulmonary artery long axis: PA -> 4
    "3": (3, 1, -1, -1),        # LV short axis: LV -> 3, RV -> 1
    "4": (3, 2, 0, 1),          # apical four chamber: LV -> 3, LA -> 2, RA -> 0, RV -> 1
}
RESIZE, CROP = 144, 112         # loader.py:462-463


def prepare_frames(images: Optional[torch.Tensor], labels: Optional[torch.Tensor], view: str, train: bool = False,
                   crop_offset: Optional[Tuple[int, int]] = None, labelled: bool = True):
    """images / labels: raw volumes [H0, W0, T] (or [H0, W0] for a single frame) on the GPU, float32.
    Returns (frames [T,1,112,112], masks [T,5,112,112]); either is None when its input is.
    Eval: centre crop (loader.py:489).  Train: a random crop window (loader.py:480) -- drawn here on the host unless
    `crop_offset` = (y, x) is given.  Unlabelled clips (is_unlab, loader.py:325) keep raw grey levels (no / 255)."""
    if view not in CLASS_TO_CHANNEL:
        raise KeyError(f"glfusion_amd.data: no part table for view {view!r} (loader.py:298-316 defines views 1-4)")
    ref = images if images is not None else labels
    if ref is None:
        raise RuntimeError("prepare_frames: nothing to prepare")
    if ref.dim() == 2:
        images = images.unsqueeze(-1) if images is not None else None
        labels = labels.unsqueeze(-1) if labels is not None else None
        ref = images if images is not None else labels
    h0, w0, t = ref.shape
    if crop_offset is None:
        if train:
            oy = int(torch.randint(0, RESIZE - CROP + 1, ()).item())
            ox = int(torch.randint(0, RESIZE - CROP + 1, ()).item())
        else:
            oy = ox = (RESIZE - CROP) // 2
    else:
        oy, ox = int(crop_offset[0]), int(crop_offset[1])
    dev = ref.device
    frames = masks = None
    if images is not None:
        images = _contig(_chk(images, "image volume"))
        frames = torch.empty(t, 1, CROP, CROP, dtype=torch.float32, device=dev)
    if labels is not None:
        labels = _contig(_chk(labels, "label volume"))
        if labels.shape != ref.shape:
            raise RuntimeError("prepare_frames: image and label volumes differ in shape")
        masks = torch.empty(t, 5, CROP, CROP, dtype=torch.float32, device=dev)
    table = (C.c_int * 4)(*CLASS_TO_CHANNEL[view])
    check(lib.glf_prepare_frames(_p(images), _p(labels), _p(frames), _p(masks), h0, w0, t, RESIZE, CROP, CROP, oy, ox, table,
                                 255.0 if labelled else 1.0, _stream()), "prepare_frames")
    return frames, masks


def part_overlap_counts(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """int64 [C, 4] = (tp, fp, fn, tn) of sigmoid(logits) > 0.5 against target per class channel of [N,C,H,W] tensors
    (the per-part metrics of main.py:537-543; their sum over C is main.py:519's whole-view count)."""
    logits, target = _contig(_chk(logits, "logits")), _contig(_chk(target, "target"))
    if logits.shape != target.shape or logits.dim() != 4:
        raise RuntimeError("part_overlap_counts: [N,C,H,W] logits and targets of one shape expected")
    n, c, h, w = logits.shape
    counts = torch.empty(c, 4, dtype=torch.int64, device=logits.device)
    check(lib.glf_overlap_counts_nchw(_p(logits), _p(target), _p(counts), n, c, h * w, _stream()), "overlap_counts_nchw")
    return counts


class SyntheticPatients:
    """Stand-in for Seg_PAHDataset's files: per patient and view a raw grey-level volume [H0, W0, T] in [0, 255] and a
    label volume with class ids 0..k (k parts of that view, loader.py:298-316), deterministic per (seed, patient, view).
    The parts are ellipses that drift over the frames, so Dice and the per-part tables are non-trivial."""

    PARTS = {"1": 2, "2": 1, "3": 2, "4": 4}

    def __init__(self, views: Sequence[str], n_patients: int, clip_length: int = 40, h0: int = 200, w0: int = 160, device="cuda", seed: int = 0):
        self.views, self.n, self.t, self.h0, self.w0, self.device, self.seed = list(views), n_patients, clip_length, h0, w0, device, seed

    def __len__(self) -> int:
        return self.n

    def volume(self, patient: int, view: str):
        g = torch.Generator().manual_seed(self.seed * 1000003 + patient * 101 + int(view))
        yy = torch.arange(self.h0, dtype=torch.float32).view(-1, 1, 1)
        xx = torch.arange(self.w0, dtype=torch.float32).view(1, -1, 1)
        tt = torch.arange(self.t, dtype=torch.float32).view(1, 1, -1)
        lab = torch.zeros(self.h0, self.w0, self.t)
        img = torch.rand(self.h0, self.w0, self.t, generator=g) * 60.0
        for k in range(1, self.PARTS[view] + 1):
            cy, cx = (torch.rand(2, generator=g) * 0.5 + 0.25).tolist()
            ry, rx = (torch.rand(2, generator=g) * 0.12 + 0.08).tolist()
            inside = (((yy - (cy + 0.02 * torch.sin(tt / 5.0)) * self.h0) / (ry * self.h0)) ** 2
                      + ((xx - (cx + 0.02 * torch.cos(tt / 7.0)) * self.w0) / (rx * self.w0)) ** 2) <= 1.0
            lab = torch.where(inside, torch.full_like(lab, float(k)), lab)
            img = torch.where(inside, img + 40.0 * k, img)
        return img.clamp_(0, 255).floor_().to(self.device), lab.to(self.device)

    def __iter__(self) -> Iterator:
        for p in range(self.n):
            yield {v: self.volume(p, v) for v in self.views}
