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


# ------------------------------------------------------------------------------------------------------------
# host side of the loader: the Dataset contract of datasets/loader.py:190-340
# ------------------------------------------------------------------------------------------------------------
def crop_offsets(rs, size=(RESIZE, RESIZE), crop=(CROP, CROP)) -> Tuple[int, ...]:
    """Top-left corner of a RandSpatialCropd(roi_size=crop, random_size=False) window on an image of `size`, drawn from the
    numpy RandomState `rs` the way MONAI draws it (monai.data.utils.get_random_patch: one `rs.randint(low=0, high=ms - ps + 1)`
    per spatial dimension, in order, only where the image is larger than the patch) -- so a caller that seeds the transform's
    RandomState like the reference (loader.py:478-484; MONAI's Randomizable.set_random_state) gets the reference's windows.
    MONAI is not installed here: restated from its published source, parity unpinned."""
    return tuple(int(rs.randint(low=0, high=ms - ps + 1)) if ms > ps else 0 for ms, ps in zip(size, crop))


class SegPAHDataset:
    """Seg_PAHDataset (datasets/loader.py:190-340) with the file access factored out: `infos[id]` is a dict with
    'dataset_name', 'views_images' and 'views_labels', whose per-view entries are NIfTI-1 paths (what the reference stores
    there, loader.py:233-234; read by glfusion_amd.nifti), raw volumes [H0, W0, T] (numpy / torch) or zero-argument callables
    returning them.
    Everything else follows the reference: the train / val / test split of the id list (loader.py:210-217), 4 samples per
    patient and epoch in train mode (loader.py:289, 335-338), the labelled-frame selection `input_select` (loader.py:429-458:
    Python's `random` module, so `random.seed` reproduces the reference's picks), the transform chain and the part masks --
    the last two on the GPU in one kernel (prepare_frames).  Returns (images, masks, index) in the reference's layouts:
    images [1,112,112] or [1,112,112,T] scaled by 1/255, masks [5,112,112(,T)]."""

    def __init__(self, infos: dict, root=None, is_train: bool = True, data_list=None, set_select=("rmyy",), view_num=("2",),
                 single_frame: bool = True, clip_length: int = 32, seg_parts: bool = True, require_id: bool = False,
                 device="cuda", crop_seed=None):
        import random
        import numpy as np
        if len(view_num) != 1:
            raise ValueError("SegPAHDataset: one view per dataset object, as in the reference (main.py:112-121 builds one per view)")
        self.is_train, self.view_num, self.single_frame = is_train, list(view_num), single_frame
        self.clip_length, self.seg_parts, self.require_id, self.device = clip_length, seg_parts, require_id, device
        self.data_dict = {k: v for k, v in infos.items() if v["dataset_name"] in set_select}          # get_dict, loader.py:416-427
        self.id_list = list(self.data_dict.keys())
        if data_list is not None:
            self.id_list = list(data_list)
        elif is_train:                                                                             # loader.py:213-217
            self.train_list = random.sample(self.id_list, int(len(self.id_list) * 0.8))
            rest = list(set(self.id_list).difference(set(self.train_list)))
            self.valid_list = random.sample(rest, int(len(rest) * 0.5))
            self.test_list = list(set(self.id_list).difference(set(self.train_list)).difference(set(self.valid_list)))
            self.id_list = self.train_list
        self.R = np.random.RandomState(crop_seed)            # the RandSpatialCropd's own RandomState (MONAI Randomizable)

    def __len__(self) -> int:
        return len(self.id_list) * 4 if self.is_train else len(self.id_list)

    @staticmethod
    def _volume(entry):
        import os
        import numpy as np
        if isinstance(entry, (str, os.PathLike)):                # the reference's case: a NIfTI path (loader.py:233-234)
            from . import nifti
            entry = nifti.read(entry)
        v = entry() if callable(entry) else entry
        return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))

    def input_select(self, images: torch.Tensor, masks: torch.Tensor):
        """loader.py:429-458.  Frames whose label map holds more than 100 labelled pixels are candidates; one is drawn with
        random.choice; a clip of clip_length - 1 frames around it (start drawn with random.randint) in clip mode."""
        import random
        if masks.dim() >= 3:
            per_frame = masks.sum(dim=(0, 1))
            cand = torch.nonzero(per_frame > 100).flatten().tolist()
            if not cand:
                raise RuntimeError("SegPAHDataset: no labelled frame in this volume (the reference's random.choice raises here too)")
            index = random.choice(cand)
            if self.single_frame:
                return images[:, :, index], masks[..., index], index
            if masks.shape[-1] == 3:
                return images[:, :, 1:2].repeat(1, 1, self.clip_length), masks[:, :, 1:2].repeat(1, 1, self.clip_length), index
            r_index = random.randint(0, index if index < self.clip_length - 1 else self.clip_length - 1)
            start = index - r_index
            end = start + self.clip_length - 1
            return images[:, :, start:end], masks[..., start:end], r_index
        if self.single_frame:
            return images, masks, 0
        return images.unsqueeze(-1).repeat(1, 1, self.clip_length), masks.unsqueeze(-1).repeat(1, 1, self.clip_length), 0

    def __getitem__(self, index: int):
        view = self.view_num[0]
        pid = self.id_list[index // 4] if self.is_train else self.id_list[index]
        entry = self.data_dict[pid]
        img_e, lab_e = entry["views_images"].get(view), entry["views_labels"].get(view)
        if img_e is None or lab_e is None:                       # loader.py:262-279: a patient without this view -> zeros
            shape = (1, CROP, CROP) if self.single_frame else (1, CROP, CROP, self.clip_length)
            imgs = torch.zeros(shape, device=self.device)
            masks = torch.zeros((5,) + shape[1:], device=self.device)
            return (imgs, masks, 0, pid) if self.require_id else (imgs, masks, 0)
        images, labels = self._volume(img_e).float(), self._volume(lab_e).float()
        images, labels, idx = self.input_select(images, labels)
        offs = crop_offsets(self.R) if self.is_train else None
        frames, masks = prepare_frames(images.to(self.device), labels.to(self.device), view, train=self.is_train, crop_offset=offs)
        if not self.seg_parts:                                   # loader.py:321: any part -> one foreground channel
            masks = (masks.sum(dim=1, keepdim=True) > 0).float()
        if self.single_frame:
            imgs, masks = frames[0], masks[0]                    # [1,112,112], [5,112,112]
        else:
            imgs, masks = frames.permute(1, 2, 3, 0), masks.permute(1, 2, 3, 0)      # [1,112,112,T], [5,112,112,T] (views)
        return (imgs, masks, idx, pid) if self.require_id else (imgs, masks, idx)


def synthetic_infos(views: Sequence[str], n_patients: int, clip_length: int = 40, device="cpu", seed: int = 0, fold: str = "0") -> dict:
    """An `infos` dict in the reference's shape (np.load('./infos/...npy') of main.py:283) whose volume entries are lazy
    SyntheticPatients volumes: ids '<fold>_<k>' like the reference's ('0_0', '0_2' = validation, the rest = test, main.py:288-289)."""
    sp = SyntheticPatients(views, n_patients, clip_length, device=device, seed=seed)
    infos = {}
    for k in range(n_patients):
        infos[f"{fold}_{k}"] = {"dataset_name": "rmyy", "fold": fold,
                                "views_images": {v: (lambda k=k, v=v: sp.volume(k, v)[0]) for v in views},
                                "views_labels": {v: (lambda k=k, v=v: sp.volume(k, v)[1]) for v in views}}
    return infos
