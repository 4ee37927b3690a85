"""Multi-GPU data parallelism for the hot path: one process per GPU, clips (frames) sharded
across ranks, ONE exchange step per iteration -- an all-reduce of the parameter gradients
(RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" on CPU for tests).

Replaces the reference's single-process nn.DataParallel (main.py:155), which re-broadcasts
897 MB of parameters every forward and reduces gradients onto GPU 0.  Semantics kept:
  * the caller's loss is a SUM over samples (BCEWithLogitsLoss(reduction='sum'), main.py:87),
    so gradients are SUMMED over ranks (average=False) -- identical to DataParallel's
    reduce-add of replica gradients for a batch split across devices;
  * BatchNorm statistics stay per-rank (DataParallel replicas use local statistics too; the
    reference has no SyncBN).
Gradients are packed into flat buckets in reverse registration order (~ backward order); a
bucket is all-reduced asynchronously as soon as its last gradient has been accumulated, so
communication overlaps the rest of backward.  Parameters that never receive gradients on
this path (the dead `network.*` template and `*.align_channel.*`) are excluded up front.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def default_ignore(name: str) -> bool:
    return name.startswith("network.") or ".align_channel." in name or name.startswith("align_channel.")


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        n = sum(p.numel() for p in params)
        p0 = params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.pending = len(params)
        self.fired = [False] * len(params)   # whose post-accumulate hook ran this step
        self.fired_capture = None            # deferred mode: the hooks that ran while the step was RECORDED (they do not run on a replay)
        self.work = None
        self.events = []        # one per gradient copied on a GPU stream (the model runs its views on side streams)


class GradAllReducer:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 48.0, average: bool = False,
                 ignore: Optional[Callable[[str], bool]] = default_ignore, process_group=None, in_place: bool = True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and not (ignore and ignore(n))]
        self.names = [n for n, _ in named]
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets: List[_Bucket] = []
        cur, size = [], 0
        for _, p in reversed(named):
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > cap:
                self.buckets.append(_Bucket(cur))
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
        # Gradient producers of the engine (ops.grad_out: conv / linear weight gradients, BatchNorm gamma / beta) write
        # straight into the parameter's bucket slice, so the hook below finds the gradient already in place and the 736 MB
        # copy into the buckets is gone.  (Only on a multi-rank job: a single rank keeps private gradient tensors.)
        self._slots = []
        if self.world > 1 and in_place:
            from . import ops
            for b in self.buckets:
                for i, p in enumerate(b.params):
                    ops.register_grad_slot(p, b.flat, b.offsets[i])
                    self._slots.append(p)
        self.in_place_elems = 0        # gradient elements found already inside their bucket slice / copied into it (cumulative)
        self.copied_elems = 0
        self.overlap_log = []          # per launched bucket: (bucket index, gradients in place when its collective was enqueued, total)
        self._fired_count = 0
        self.allreduce_ms = None       # deferred mode: GPU time of the last finalize()'s collectives (timed with events)
        self._first_launch_ev = None   # eager mode: event recorded when the step's first collective was enqueued
        self._eager_events = None      # eager mode: (first enqueue, end of backward, last collective done) of the latest finalize()
        self._handles = []
        # deferred mode (engine.StepGraph): the hooks only copy gradients into the buckets -- work a hipGraph capture can
        # record -- and every collective is launched by finalize(), after backward (after the graph replay), on the current
        # stream.  No overlap with backward, nothing but plain eager all_reduce calls on persistent flat tensors.
        self.deferred = False
        if self.world > 1:
            for p in self._where:
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    # -- one-off: make every rank start from rank 0's parameters and buffers ----------------
    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world == 1:
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src, group=self.group)

    def _launch(self, b: _Bucket) -> None:
        if b.events:            # the collective is ordered after the current stream: make that stream wait for every copy
            cur = torch.cuda.current_stream(b.flat.device)
            for e in b.events:
                cur.wait_event(e)
            b.events = []
        self.overlap_log.append((self.buckets.index(b), self._fired_count, len(self._where)))
        if self._first_launch_ev is None and b.flat.is_cuda:
            self._first_launch_ev = torch.cuda.Event(enable_timing=True)
            self._first_launch_ev.record(torch.cuda.current_stream(b.flat.device))
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _hook(self, p: torch.nn.Parameter) -> None:
        b, i = self._where[p]
        off = b.offsets[i]
        self._fired_count += 1
        g = p.grad
        if not (g.is_contiguous() and g.data_ptr() == b.flat.data_ptr() + off * b.flat.element_size()):
            b.flat[off:off + p.numel()].copy_(g.reshape(-1))        # produced elsewhere (TPAVI projections, shared parameters ...)
            self.copied_elems += p.numel()
        else:
            self.in_place_elems += p.numel()
        if self.deferred:
            b.fired[i] = True            # the end-of-backward stream join orders the copy before finalize()
            return
        if b.flat.is_cuda:
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(b.flat.device))
            b.events.append(e)
        b.pending -= 1
        b.fired[i] = True
        if b.pending == 0:
            self._launch(b)

    def finalize(self) -> None:
        """Call once after backward(): launches buckets that did not fill, waits for the collectives and re-points the
        .grad of every parameter that RECEIVED a gradient this step at its reduced bucket slice.  A parameter whose hook
        never fired (unused in this step's graph -- the same set on every rank, the graph being the same) contributes
        zeros to the collective and keeps `.grad` as it was (None after zero_grad(set_to_none=True)): the optimizer
        skips it exactly as it does on one GPU, so results do not depend on the world size."""
        if self.world == 1:
            return
        self._fired_count = 0
        if self._slots:
            from . import ops
            ops.release_grad_slots(self._slots)
        if self.deferred:
            # The hooks run while the step is RECORDED (warm-up + capture), not on a replay: which parameters take part is fixed at
            # the first finalize() after the capture.  Slices of parameters that never fire must contribute zeros -- the collective is
            # in place, so whatever an earlier (eager or warm-up) step left there would be re-reduced on every replay: they are
            # zero-filled before the collectives.
            for b in self.buckets:
                if b.fired_capture is None or any(b.fired):
                    b.fired_capture = [a or c for a, c in zip(b.fired, b.fired_capture or [False] * len(b.params))]
                b.fired = [False] * len(b.params)
                b.pending = len(b.params)
                b.work = None
                b.events = []
                for i, p in enumerate(b.params):
                    if not b.fired_capture[i]:
                        off = b.offsets[i]
                        b.flat[off:off + p.numel()].zero_()
            ev = None
            if self.buckets and self.buckets[0].flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            works = [dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]
            for b, w in zip(self.buckets, works):
                w.wait()
            if ev is not None:
                ev[1].record()
                self._last_events = ev
            for b in self.buckets:
                if self.average:
                    b.flat.div_(self.world)
                for i, p in enumerate(b.params):
                    if b.fired_capture[i]:
                        off = b.offsets[i]
                        p.grad = b.flat[off:off + p.numel()].view_as(p)
            return
        for b in self.buckets:
            b.fired_capture = None           # (back in eager mode: a later capture starts from its own record)
        self.overlap_log = self.overlap_log[-len(self.buckets):]
        cuda = bool(self.buckets) and self.buckets[0].flat.is_cuda
        ev_bwd = None
        if cuda:
            # everything backward enqueued on this stream is ahead of this event: what the step waits for AFTER it is communication
            ev_bwd = torch.cuda.Event(enable_timing=True)
            ev_bwd.record()
        for b in self.buckets:
            if b.work is None:
                for i, p in enumerate(b.params):
                    if not b.fired[i]:
                        off = b.offsets[i]
                        b.flat[off:off + p.numel()].zero_()
                self._launch(b)
        for b in self.buckets:
            b.work.wait()                  # (GPU: the CURRENT stream -- the one the optimizer runs on -- waits for the collective)
            if self.average:
                b.flat.div_(self.world)
            for i, p in enumerate(b.params):
                if b.fired[i]:
                    off = b.offsets[i]
                    p.grad = b.flat[off:off + p.numel()].view_as(p)
            b.pending = len(b.params)
            b.fired = [False] * len(b.params)
            b.work = None
        if cuda:
            ev_done = torch.cuda.Event(enable_timing=True)
            ev_done.record()
            self._eager_events = (self._first_launch_ev, ev_bwd, ev_done)
        self._first_launch_ev = None

    def eager_comm_ms(self):
        """Eager (immediate) mode, after a finalize(): (comm_window_ms, exposed_comm_ms) on the GPU clock -- the time from the
        enqueue of the step's FIRST bucket collective to the completion of the last one, and the part of it that lies AFTER the
        last backward kernel (what the step actually waits for; 0 when every collective finished under backward).  Synchronises."""
        ev = self._eager_events
        if ev is None or ev[0] is None:
            return None
        ev[2].synchronize()
        return ev[0].elapsed_time(ev[2]), ev[1].elapsed_time(ev[2])

    def last_allreduce_ms(self) -> Optional[float]:
        """GPU time between the first collective's enqueue and the last one's completion in the latest deferred finalize()
        (synchronises the device)."""
        ev = getattr(self, "_last_events", None)
        if ev is None:
            return None
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
        if self._slots:
            from . import ops
            ops.unregister_grad_slots(self._slots)
            self._slots = []


def shard_frames(n_total: int, rank: int, world: int):
    """[start, end) of the frames (clips x T) this rank owns: contiguous, equal shares (weak scaling keeps
    the per-rank share fixed; cross-view attention couples the V views of one frame, so all views of a
    frame stay on one rank)."""
    if n_total % world != 0:
        raise ValueError(f"{n_total} frames do not split evenly over {world} ranks")
    per = n_total // world
    return rank * per, (rank + 1) * per


def all_reduce_counts(counts: torch.Tensor, process_group=None) -> torch.Tensor:
    """Sum the 4 overlap counters (tp, fp, fn, tn) over ranks for a global Dice in eval."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=process_group)
    return counts
