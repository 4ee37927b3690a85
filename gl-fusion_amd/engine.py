"""Caller-side harness with the reference Trainer's surface (main.py:63-883) for the hot path:
`Trainer(config)`, `.train(is_backbone, is_cycle)`, `.eval(net_path, is_fuse, raw_data)`,
`_calculate_overlap_metrics`, `save()`.  The reference's data layer needs NIfTI files that are not
shipped (absolute paths on the authors' machine), so loaders here produce synthetic clips with
the reference's tensor contract: images [N,1,112,112] in [0,1], masks [N,5,112,112] in {0,1}
(datasets/loader.py:298-330).  The temporal cycle loss (main.py:650-798) is SURVEY row f1 (next), so
`is_cycle=True` is accepted and ignored with a notice.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, Tuple

import torch

from . import ops
from .ddp import GradAllReducer, all_reduce_counts
from .optim import Adam
from .models import Global_and_Local


class SyntheticClips:
    """Per-view iterator of (img, mask) batches on the device, deterministic per (seed, rank)."""

    def __init__(self, views, frames: int, h: int, w: int, device, seed: int = 1234, length: int = 8):
        self.views, self.frames, self.h, self.w, self.device, self.length = list(views), frames, h, w, device, length
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def __len__(self) -> int:
        return self.length

    def batch(self) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        imgs = {v: torch.rand(self.frames, 1, self.h, self.w, device=self.device, generator=self.gen) for v in self.views}
        masks = {v: (torch.rand(self.frames, 5, self.h, self.w, device=self.device, generator=self.gen) < 0.3).float() for v in self.views}
        return imgs, masks

    def __iter__(self) -> Iterator:
        for _ in range(self.length):
            yield self.batch()


class Trainer:
    def __init__(self, config: dict):
        self.config = config
        tr = config["train"]
        self.view_num = tr["view_num"]
        self.test_view = tr.get("test_view", self.view_num)
        self.device = tr.get("device", torch.device("cuda", 0))
        if not torch.cuda.is_available():
            raise RuntimeError("glfusion_amd.engine.Trainer needs an MI355X: the HIP engine has no CPU fallback")
        self.print_val = tr.get("global_rank", 0) == 0                       # main.py:92
        self.model = Global_and_Local(view_num=self.view_num).to(self.device)  # main.py:150
        opt = config["net"]["opt"]
        if opt.get("opt_name", "Adam") != "Adam":
            raise NotImplementedError("glfusion_amd: only the Adam branch of main.py:158-165 is built (the shipped config)")
        self.optimizer = Adam(self.model.parameters(), lr=opt["lr"], weight_decay=opt["weight_decay"])    # main.py:162-165, fused
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=tr["num_epochs"])   # main.py:168
        self.reducer = GradAllReducer(self.model)
        self.reducer.broadcast_parameters(0)
        frames = tr["batch_size"] * tr.get("frames_per_clip", 1)
        self.loader = SyntheticClips(self.view_num, frames, 112, 112, self.device, seed=1234 + tr.get("global_rank", 0),
                                     length=tr.get("iters_per_epoch", 4))

    def train_step(self, imgs, masks) -> torch.Tensor:
        """main.py:202-243 without the cycle term: forward, sum_v BCE-sum, backward, Adam step."""
        pred_frames, _, _, _ = self.model(imgs)
        loss = None
        for view in self.test_view:
            l = ops.bce_with_logits_sum(pred_frames[view], masks[view])
            loss = l if loss is None else loss + l
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.reducer.finalize()
        self.optimizer.step()
        return loss.detach(), pred_frames

    def train(self, is_backbone: bool = False, is_cycle: bool = True):
        if is_cycle and self.print_val:
            print("[glfusion_amd] temporal cycle loss (main.py:650-798) is not on the built path yet: training on seg_loss only")
        for epoch in range(self.config["train"]["num_epochs"]):
            self.model.train()
            for imgs, masks in self.loader:
                loss, pred = self.train_step(imgs, masks)
            self.scheduler.step()
            if self.print_val:
                dice = {v: self._calculate_overlap_metrics(masks[v], pred[v].detach())[1] for v in self.test_view}
                print(f"epoch {epoch}: loss {float(loss):.2f} dice {dice}")
            self.save(epoch)

    @torch.no_grad()
    def eval(self, net_path: str = None, is_fuse: bool = True, raw_data: bool = True):
        """main.py:417-543 shape contract: clip [1,1,H,W,T] -> [T,1,H,W] frames, sigmoid > 0.5, overlap metrics."""
        if net_path and os.path.exists(net_path):
            self.model.load_state_dict(torch.load(net_path, map_location=self.device)["network"], strict=True)
        self.model.eval()
        out = {}
        imgs, masks = self.loader.batch()
        mask, _, _, _ = self.model(imgs)
        for v in self.test_view:
            out[v] = self._calculate_overlap_metrics(masks[v], mask[v])
        return out

    def _calculate_overlap_metrics(self, gt, logits, eps: float = 1e-5):
        """main.py:800-815 on pred = (sigmoid(logits) > 0.5); counters reduced over ranks."""
        counts = all_reduce_counts(ops.overlap_counts(logits, gt))
        return ops.overlap_metrics_from_counts(counts, eps)

    def save(self, epoch: int):
        """main.py:857-872: {'network': state_dict} -> save_dir/net_%05d.pth + latest.ckpt."""
        if not self.print_val:
            return
        d = self.config["train"]["save_dir"]
        os.makedirs(d, exist_ok=True)
        torch.save({"network": self.model.state_dict()}, os.path.join(d, "net_%05d.pth" % epoch))
        with open(os.path.join(d, "latest.ckpt"), "w") as f:
            f.write(str(epoch))
