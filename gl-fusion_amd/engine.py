"""Caller-side harness with the reference Trainer's surface (main.py:63-883) for the hot path:
`Trainer(config)`, `.train(is_backbone, is_cycle)`, `.eval(net_path, is_fuse, raw_data)`,
`_calculate_overlap_metrics`, `save()`.  The reference's data layer needs NIfTI files that are not
shipped (absolute paths on the authors' machine), so loaders here produce synthetic clips with
the reference's tensor contract: images [N,1,112,112] in [0,1], masks [N,5,112,112] in {0,1}
(datasets/loader.py:298-330); the unlabelled "video" loader of the cycle term yields clips of `clip_length`
frames per view (main.py:213-218).  `is_cycle=True` adds the temporal cycle-consistency loss of main.py:213-237
(second forward on the video clip, pooled global-fusion features, seg_cycle / dense_seg_cycle, total = seg + 1e-2 cyc).
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


class StepGraph:
    """One training step -- forward, loss, backward of a FIXED-shape batch -- recorded once as a hipGraph and replayed.

    The eager step issues ~3 200 kernel launches from Python (autograd Functions + ctypes), ~230 ms of host work per 264 ms
    step at config 2: the launch thread, not the GPU, bounds any further speed-up.  A replay costs one hipGraphLaunch.

        graph = StepGraph(step_fn, params)      # step_fn(): zero-arg closure over STATIC input tensors; returns the loss
        loss = graph.replay()                   # every kernel of the step runs again; loss / .grad tensors are static

    What makes the step capture-safe (none of it changes results):
      * weight-derived images are rebuilt by ops.refresh_weights() INSIDE the captured work (static buffers, four launches),
        so a replay after an optimizer step sees the new parameters;
      * Dropout masks depend on a device step counter the captured work advances first thing (ops.advance_step);
      * the slot pools of operand maxima / BatchNorm sums are re-created inside the capture (their zero fill is replayed);
      * the independent sections of a forward use pairwise different side streams (ops.SECTIONS_DISTINCT; re-using one side
        stream in two fork/join groups of a capture crashes hipStreamEndCapture on ROCm 7.2);
      * nothing in the step synchronises the host or allocates outside torch's allocator.
    Gradient all-reduce stays OUTSIDE the graph (GradAllReducer in deferred mode: the hooks' bucket copies are captured, the
    collectives are launched by finalize() after each replay).
    Update parameters only in place (the fused Adam does): the graph holds their addresses."""

    def __init__(self, step_fn, params, warmup: int = 2, reducer=None, refresh_weights: bool = True):
        self.params = [p for p in params]
        if not self.params:
            raise RuntimeError("StepGraph: no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("StepGraph needs CUDA(HIP) parameters: the engine has no CPU fallback")
        self.device = dev
        self.reducer = reducer
        if reducer is not None:
            reducer.deferred = True
        prev_distinct, ops.SECTIONS_DISTINCT = ops.SECTIONS_DISTINCT, True
        self._prev_distinct = prev_distinct
        # a conv's weight gradient on a side stream is a fork nested inside a section: not capturable on ROCm 7.2 (see above)
        prev_wgrad, ops.WGRAD_STREAM = ops.WGRAD_STREAM, False

        self.wtable = None                         # frozen job table of THIS model's weight images (made after the warm-up)

        def body():
            ops.advance_step(dev)
            if refresh_weights:
                ops.refresh_weights(self.wtable)   # warm-up: the registry-wide refresh; captured: this model's images only
            for p in self.params:
                p.grad = None
            return step_fn()

        self.stream = torch.cuda.Stream(device=dev)            # warm-up and capture share it (library rings, pools, allocator)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(self.stream):
                for _ in range(max(1, warmup)):                    # lazily-created state (caches, job table, rings) settles here
                    out = body()
                    if reducer is not None:
                        reducer.finalize()
                del out
                torch.cuda.synchronize(dev)
                for p in self.params:
                    p.grad = None
                ops.reset_capture_pools()
                if refresh_weights:
                    # the captured refresh replays against a table of its own: it names only this model's images and keeps their
                    # buffers alive, so nothing another model (or the garbage collector) does to the registry can invalidate it
                    self.wtable = ops.freeze_weight_table(self.params)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.out = body()
        except BaseException:
            # a failed warm-up / capture must not leave the process in capture configuration: the reducer back to
            # immediate mode, the section / weight-gradient stream policies back to what they were
            if reducer is not None:
                reducer.deferred = False
            ops.SECTIONS_DISTINCT = prev_distinct
            raise
        finally:
            ops.WGRAD_STREAM = prev_wgrad
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        self.grads = {p: p.grad for p in self.params if p.grad is not None}      # static tensors the replays write

    def replay(self):
        """Run the recorded step on the current stream.  Returns step_fn()'s (static) result."""
        self.graph.replay()
        if self.reducer is not None and self.reducer.world > 1:
            self.reducer.finalize()                  # re-points .grad at the reduced bucket slices
        else:
            for p, g in self.grads.items():          # an optimizer's zero_grad(set_to_none=True) may have dropped them
                p.grad = g
        return self.out

    def release(self) -> None:
        """Free the graph and its private memory pool."""
        ops.SECTIONS_DISTINCT = self._prev_distinct
        if self.reducer is not None:
            self.reducer.deferred = False
        self.graph = None
        self.out = None
        self.grads = {}
        self.wtable = None


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
        # config['train']['precision'] (no reference counterpart): "f32" | "bf16x6" | "f16x3" | "f16" | "bf16" (16-bit activation storage,
        # BASELINE.json configs[2] / [4]); absent = whatever glfusion_amd.ops is set to
        if tr.get("precision"):
            ops.set_precision(tr["precision"])
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
        self.dense_cyc = bool(tr.get("dense_cyc", False))                      # main.py:228
        # config['train']['graph']: replay the (cycle-free) training step from one hipGraph instead of issuing ~3 400 launches
        self.use_graph = bool(tr.get("graph", False))
        self._graph = None
        self.video_loader = SyntheticClips(self.view_num, int(tr.get("clip_length", 40)), 112, 112, self.device,
                                           seed=4321 + tr.get("global_rank", 0), length=tr.get("iters_per_epoch", 4))

    def cycle_loss(self, video: Dict[str, torch.Tensor]) -> torch.Tensor:
        """main.py:213-235: second forward on an unlabelled clip [T,1,H,W] per view, global-fusion features summed
        over (h, w), seg_cycle (or dense_seg_cycle) per view with target_region 16, cyc_off 2, chunk_size 3,
        temperature 10."""
        _, _, feat_out, _ = self.model(video)
        feats = ops.pooled_fusion_features(feat_out)
        total = None
        for view in self.view_num:
            if self.dense_cyc:
                l = ops.dense_seg_cycle(feats[view], 16, 2, 3, 10, soft_label=False, is_overlap=True)
            else:
                l = ops.seg_cycle(feats[view], 16, 2, 3, 10)
            total = l if total is None else total + l
        return total

    def _graph_step(self, imgs, masks):
        """The segmentation step (no cycle term) replayed from ONE hipGraph (StepGraph): the batch is copied into static input
        buffers, the recorded forward + loss + backward (+ weight-image refresh) runs, Adam updates the parameters in place.
        Recorded at the first call; the batch shape must not change afterwards."""
        if self._graph is None:
            self._static = ({v: torch.empty_like(t) for v, t in imgs.items()}, {v: torch.empty_like(t) for v, t in masks.items()})
            s_imgs, s_masks = self._static
            holder = {}

            def core():
                pred, _, _, _ = self.model(s_imgs)
                holder["pred"] = pred
                loss = None
                for view in self.test_view:
                    l = ops.bce_with_logits_sum(pred[view], s_masks[view])
                    loss = l if loss is None else loss + l
                loss.backward()
                return loss.detach()
            for v in imgs:
                s_imgs[v].copy_(imgs[v]); s_masks[v].copy_(masks[v])
            self._graph = StepGraph(core, [p for p in self.model.parameters()], warmup=1, reducer=self.reducer if self.reducer.world > 1 else None)
            self._graph_pred = holder["pred"]
        s_imgs, s_masks = self._static
        for v in imgs:
            if imgs[v].shape != s_imgs[v].shape:
                raise RuntimeError("glfusion_amd.engine: a graph-replayed step needs batches of one shape (config['train']['graph'] = False for ragged ones)")
            s_imgs[v].copy_(imgs[v]); s_masks[v].copy_(masks[v])
        loss = self._graph.replay()
        self.optimizer.step()
        return loss, self._graph_pred

    def train_step(self, imgs, masks, video=None) -> torch.Tensor:
        """main.py:202-243: forward, sum_v BCE-sum (+ 1e-2 x cycle loss on `video`), backward, Adam step."""
        if video is None and self.use_graph:
            return self._graph_step(imgs, masks)
        pred_frames, _, _, _ = self.model(imgs)
        loss = None
        for view in self.test_view:
            l = ops.bce_with_logits_sum(pred_frames[view], masks[view])
            loss = l if loss is None else loss + l
        if video is not None:
            loss = loss + 1e-2 * self.cycle_loss(video)                        # main.py:237
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.reducer.finalize()
        self.optimizer.step()
        return loss.detach(), pred_frames

    def train(self, is_backbone: bool = False, is_cycle: bool = True):
        for epoch in range(self.config["train"]["num_epochs"]):
            self.model.train()
            for imgs, masks in self.loader:
                video = self.video_loader.batch()[0] if is_cycle else None
                loss, pred = self.train_step(imgs, masks, video)
            self.scheduler.step()
            # the epoch's Dice sums its counters over ranks: a collective, so EVERY rank computes it (only rank 0 prints)
            dice = {v: self._calculate_overlap_metrics(masks[v], pred[v].detach())[1] for v in self.test_view}
            if self.print_val:
                print(f"epoch {epoch}: loss {float(loss):.2f} dice {dice}")
                if self.config["train"].get("validate_every_epoch", True):
                    # main.py:259-274: validation runs on the printing rank alone, over every clip of the splits, with NO collective
                    # (the other ranks have already moved on to the next epoch's gradient all-reduces and wait there)
                    self.validation_and_test(net_root=None, is_fuse=True, raw_data=True, reduce=False)
            self.save(epoch)

    @torch.no_grad()
    def eval(self, net_path: str = None, is_fuse: bool = True, raw_data: bool = True, patients=None):
        """main.py:417-543.  Per patient and view a raw clip volume goes through the data path (data.prepare_frames: the
        loader's resize / centre crop / part masks / 255 and the `[1,1,H,W,T] -> [T,1,H,W]` reshape of main.py:495-499),
        the model predicts all frames of the clip (the fused mask, or the backbone mask when is_fuse is False, main.py:
        504-506), BCE-sum losses accumulate per view (main.py:510-512) and the overlap counters accumulate over patients
        -- tp / fp / fn / tn are additive, so the reference's concatenation of every prediction (main.py:514-516) is not
        needed.  Returns {view: (pixel_acc, dice, precision, specificity, recall)} (main.py:519); the per-part Dice of
        main.py:537-543 and the losses are left in `self.eval_report`."""
        from .data import SyntheticPatients, part_overlap_counts
        from . import data as _data
        if net_path and os.path.exists(net_path):
            self.model.load_state_dict(torch.load(net_path, map_location=self.device)["network"], strict=True)
        self.model.eval()
        if patients is None:
            patients = SyntheticPatients(self.view_num, int(self.config["train"].get("eval_patients", 2)),
                                         int(self.config["train"].get("clip_length", 40)), device=self.device, seed=77)
        counts = {v: torch.zeros(5, 4, dtype=torch.int64, device=self.device) for v in self.test_view}
        loss4view = {v: 0.0 for v in self.test_view}
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        for k, sample in enumerate(patients):
            if k % world != rank:                               # every rank scores its own share; counters and losses are summed below
                continue
            imgs, masks = {}, {}
            for v in self.view_num:
                imgs[v], masks[v] = _data.prepare_frames(sample[v][0], sample[v][1], v, train=False)
            out = self.model(imgs)
            pred = out[0] if is_fuse else out[1]
            for v in self.test_view:
                counts[v] += part_overlap_counts(pred[v], masks[v])
                loss4view[v] += float(ops.bce_with_logits_sum(pred[v], masks[v]))
        result, part_dice = {}, {}
        if world > 1:
            lt = torch.tensor([loss4view[v] for v in self.test_view], dtype=torch.float64, device=self.device)
            dist.all_reduce(lt)
            loss4view = {v: float(x) for v, x in zip(self.test_view, lt.tolist())}
        for v in self.test_view:
            per_part = all_reduce_counts(counts[v])
            result[v] = ops.overlap_metrics_from_counts(per_part.sum(dim=0))
            part_dice[v] = [ops.overlap_metrics_from_counts(per_part[c])[1] for c in range(5)]
        self.eval_report = {"loss": loss4view, "part_dice": part_dice}
        if self.print_val:
            for v in self.test_view:
                print(f"validation view {v}: loss {loss4view[v]:.4f} pixel-acc {result[v][0]:.4f} dice {result[v][1]:.4f} "
                      f"precision {result[v][2]:.4f} specificity {result[v][3]:.4f} recall {result[v][4]:.4f}; part dice "
                      + " ".join(f"{d:.4f}" for d in part_dice[v]))
        return result

    @torch.no_grad()
    def validation_and_test(self, net_root: str = None, is_fuse: bool = True, raw_data: bool = True, infos: dict = None,
                            val_list=("0_0", "0_2"), test_list=("0_1", "0_3", "0_4", "0_5", "0_6", "0_7", "0_8", "0_9"), first_scored: int = 50,
                            reduce: bool = True):
        """main.py:279-415.  Two splits of the test infos -- 'Inner-val' (ids 0_0, 0_2) and 'Inner-test' (the other eight) --
        evaluated clip by clip (batch 1, all frames of the clip as the model's batch, main.py:361-365): per view the BCE-sum loss,
        pixel accuracy / Dice / precision / specificity / recall over ALL frames of the split and the per-part Dice
        (main.py:385-407).  net_root None (the call at the end of every training epoch, main.py:274): the current weights, returns
        the validation Dice averaged over views (main.py:409-410).  With net_root: every checkpoint net_%05d.pth found there is
        loaded and scored, and the best validation epoch from `first_scored` on is reported (main.py:412-415: epochs 50+).
        `infos`: the reference loads ./infos/test_infos.npy (NIfTI paths, not shipped); default = synthetic volumes with the same ids.
        reduce: sum the overlap counters over ranks (every rank must then make this call); False for a call made by one rank alone."""
        from .data import SegPAHDataset, synthetic_infos, part_overlap_counts
        clip = 40 if raw_data else int(self.config["train"].get("clip_length", 40))                       # main.py:307-311
        if infos is None:
            infos = synthetic_infos(self.view_num, 10, clip, device="cpu", seed=91)
        splits = {"Inner-val": [i for i in val_list if i in infos], "Inner-test": [i for i in test_list if i in infos]}
        sets = {name: {v: SegPAHDataset(infos, is_train=False, data_list=ids, view_num=[v], single_frame=False, clip_length=clip + 1,
                                        seg_parts=True, device=self.device) for v in self.view_num} for name, ids in splits.items()}

        def score(tag):
            self.model.eval()
            report = {}
            for name, per_view in sets.items():
                counts = {v: torch.zeros(5, 4, dtype=torch.int64, device=self.device) for v in self.test_view}
                loss4view = {v: 0.0 for v in self.test_view}
                for i in range(len(per_view[self.view_num[0]])):
                    imgs, masks = {}, {}
                    for v in self.view_num:
                        img, mask, _ = per_view[v][i]                          # [1,112,112,T], [5,112,112,T]
                        imgs[v] = img.permute(3, 0, 1, 2).contiguous()         # main.py:361-365: frames become the batch
                        masks[v] = mask.permute(3, 0, 1, 2).contiguous()
                    out = self.model(imgs)
                    pred = out[0] if is_fuse else out[1]
                    for v in self.test_view:
                        counts[v] += part_overlap_counts(pred[v], masks[v])
                        loss4view[v] += float(ops.bce_with_logits_sum(pred[v], masks[v]))
                res = {}
                for v in self.test_view:
                    # reduce=False: this rank scored every clip itself (the per-epoch call from train(), rank 0 only): no collective
                    per_part = all_reduce_counts(counts[v]) if reduce else counts[v]
                    res[v] = {"metrics": ops.overlap_metrics_from_counts(per_part.sum(dim=0)), "loss": loss4view[v],
                              "part_dice": [ops.overlap_metrics_from_counts(per_part[c])[1] for c in range(5)]}
                    if self.print_val:
                        m = res[v]["metrics"]
                        print(f"------Validation Result . {name} for view{v} {tag}------ Loss : {loss4view[v]:.4f} Pixel Acc : {m[0]:.4f} "
                              f"Dice : {m[1]:.4f} Precision : {m[2]:.4f} Specificity : {m[3]:.4f} Recall : {m[4]:.4f}; part dice "
                              + " ".join(f"{d:.4f}" for d in res[v]["part_dice"]))
                report[name] = res
            val = report["Inner-val"]
            report["val_dice"] = sum(val[v]["metrics"][1] for v in self.test_view) / max(len(self.test_view), 1)
            return report

        if net_root is None:
            self.validation_report = score("")
            return self.validation_report["val_dice"]
        dices = []
        for epoch in range(100):                                                # main.py:317
            path = os.path.join(net_root, "net_%05d.pth" % epoch)
            if not os.path.exists(path):
                break
            self.model.load_state_dict(torch.load(path, map_location=self.device)["network"], strict=True)
            dices.append(score(f"(epoch {epoch})")["val_dice"])
        scored = dices[first_scored:] if len(dices) > first_scored else dices
        off = first_scored if len(dices) > first_scored else 0
        best = max(range(len(scored)), key=lambda k: scored[k]) + off if scored else None
        if self.print_val and best is not None:
            print(f"best val epoch:{best},best val dice:{dices[best]}")
        return best, dices

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
            f.write("%05d\n" % epoch)                              # main.py:869: `echo 00005 > latest.ckpt` (zero-padded, newline)
