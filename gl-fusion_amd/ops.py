"""torch.autograd.Function wrappers over the C ABI (include/glfusion.h).

Every function here launches hand-written gfx950 kernels from libglfusion_hip.so on
PyTorch's current HIP stream.  PyTorch is used for device memory (torch.empty / zeros),
views and the autograd graph only.  Inputs that are not CUDA float32 tensors raise: there
is no CPU or eager fallback on the product path.

Internal layout: activations are channels-last, carried as contiguous [N, H, W, C]
tensors (== the row-major matrix [N*H*W, C] the kernels see).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import GemmParams, WeightJob, WJ_AMAX, WJ_COPY, WJ_PACK, WJ_PASSES, WJ_PASS_OF, WJ_TAP_MAJOR, WJ_TAP_MAJOR_T, WJ_TRANSPOSE, WJ_ZERO, check, lib


# ----------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------
def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(
            f"glfusion_amd: {name} must be a CUDA(HIP) float32 tensor (got "
            f"{getattr(t, 'device', None)}, {getattr(t, 'dtype', None)}). The engine has no CPU fallback.")
    return t


def _contig(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _ws(rows: int, c: int, dev) -> torch.Tensor:
    return torch.empty(int(lib.glf_bn_workspace(rows, c)), dtype=torch.float64, device=dev)


def zeros(*shape, dtype=torch.float32, device=None) -> torch.Tensor:
    """Zero-filled device tensor through the library (glf_zero on the current stream), for accumulation targets."""
    t = torch.empty(*shape, dtype=dtype, device=device)
    if t.numel():
        check(lib.glf_zero(_p(t), t.numel() * t.element_size(), _stream()), "zero")
    return t


def zero_(t: torch.Tensor) -> torch.Tensor:
    """In-place zero of a DENSE tensor (contiguous storage span) through the library."""
    if not t.is_contiguous():
        raise RuntimeError("glfusion_amd: zero_ needs a contiguous tensor")
    if t.numel():
        check(lib.glf_zero(_p(t), t.numel() * t.element_size(), _stream()), "zero")
    return t


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """NCHW-shaped tensor (any strides) -> contiguous [N,H,W,C]; free for channels_last / C == 1."""
    _chk(x, "input")
    if x.dim() != 4:
        raise RuntimeError(f"expected a 4-D NCHW tensor, got shape {tuple(x.shape)}")
    n, c, h, w = x.shape
    if c == 1:
        return _contig(x).reshape(n, h, w, 1)
    xp = x.permute(0, 2, 3, 1)
    return xp if xp.is_contiguous() else xp.contiguous()


def from_nhwc(y: torch.Tensor) -> torch.Tensor:
    """[N,H,W,C] -> NCHW-shaped view with channels_last strides (zero-copy)."""
    return y.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------
# contraction front-end
# ----------------------------------------------------------------------------------------
def _axis_valid_fwd(hd: int, hs: int, k: int, stride: int, pad: int, dil: int) -> List[bool]:
    return [any(0 <= y * stride - pad + t * dil < hs for y in range(hd)) for t in range(k)]


def _axis_valid_bwd(hd: int, hs: int, k: int, stride: int, pad: int, dil: int) -> List[bool]:
    out = []
    for t in range(k):
        ok = False
        for y in range(hd):
            v = y + pad - t * dil
            if v >= 0 and v % stride == 0 and v // stride < hs:
                ok = True
                break
        out.append(ok)
    return out


_mask_cache = {}


def tap_mask(gather: int, hd: int, wd: int, hs: int, ws: int, kh: int, kw: int, stride: int, pad: int, dil: int) -> int:
    key = (gather, hd, wd, hs, ws, kh, kw, stride, pad, dil)
    m = _mask_cache.get(key)
    if m is None:
        f = _axis_valid_fwd if gather == 1 else _axis_valid_bwd
        vy, vx = f(hd, hs, kh, stride, pad, dil), f(wd, ws, kw, stride, pad, dil)
        m = 0
        for ky in range(kh):
            for kx in range(kw):
                if vy[ky] and vx[kx]:
                    m |= 1 << (ky * kw + kx)
        _mask_cache[key] = m
    return m


# contraction precision of the calls this module issues (index into PRECISIONS); passed PER CALL through
# glf_gemm_params.precision, so nothing process-wide is touched in the library
_PREC = [0]
_S16 = [False]             # 16-bit storage mode ("bf16" precision): see glfusion_amd.ops16


def s16() -> bool:
    return _S16[0]


def act_dtype() -> torch.dtype:
    """dtype of the activations the engine keeps in HBM under the current precision."""
    return torch.bfloat16 if _S16[0] else torch.float32


def _is16(t) -> bool:
    return isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16

TWO_STAGE_SPLITK = os.environ.get("GLF_TWO_STAGE", "1") != "0"


def tn_needs_zero(split: int) -> bool:
    """True when a split-K TN call sums its slices with float atomics (C must then be zero-filled by the caller)."""
    return split > 1 and not TWO_STAGE_SPLITK


# bench.py sets this to a list to time every contraction launch with HIP events on the launch stream
PROFILER = None
KERNEL_NAMES = {("nt", False): "gemm_rows_kernel<0,false>", ("nt", True): "gemm_rows_kernel<0,true>",
                ("nn", False): "gemm_rows_kernel<1,false>", ("nn", True): "gemm_rows_kernel<1,true>",
                ("tn", False): "gemm_tn_kernel<false>", ("tn", True): "gemm_tn_kernel<true>"}


def gemm(mode: str, A: torch.Tensor, B: torch.Tensor, Cm: torch.Tensor, *, M: int, N: int, K: int,
         lda: int, ldb: int, ldc: int, bias: Optional[torch.Tensor] = None, taps: int = 1, mask: int = 1,
         tap_stride_b: int = 0, gather: int = 0, geo: Optional[Tuple[int, ...]] = None, batch: int = 1,
         bsa: int = 0, bsb: int = 0, bsc: int = 0, alpha: float = 1.0, accumulate: bool = False,
         split: int = 1, rect: bool = False, amax_a: Optional[torch.Tensor] = None,
         amax_b: Optional[torch.Tensor] = None, amax_c: Optional[torch.Tensor] = None,
         colstats: Optional[torch.Tensor] = None, a_packed: bool = False, b_packed: bool = False,
         colmax: Optional[torch.Tensor] = None) -> None:
    """mode in {'nt','nn','tn'}; geo = (n_img, hs, ws, hd, wd, kh, kw, stride, pad, dil).
    amax_a / amax_b: device scalars bounding max|A| / max|B| (f16x3 precision only; None = measured by the library);
    amax_c: a slot from amax_slot() that receives max|C written| (ignored by rect / split > 1 / non-f16x3 calls -- pass
    it only to calls that store C directly).
    a_packed / b_packed: A / B is the packed pre-split image packed_of() made with the same amax_a / amax_b."""
    p = GemmParams()
    p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, lda, ldb, ldc
    p.taps, p.tap_mask, p.tap_stride_b, p.gather = taps, mask, tap_stride_b, gather
    if geo is not None:
        (p.n_img, p.hs, p.ws, p.hd, p.wd, p.kh, p.kw, p.stride, p.pad, p.dil) = geo
    else:
        (p.n_img, p.hs, p.ws, p.hd, p.wd, p.kh, p.kw, p.stride, p.pad, p.dil) = (1, 1, 1, 1, 1, 1, 1, 1, 0, 1)
    p.batch, p.batch_stride_a, p.batch_stride_b, p.batch_stride_c = batch, bsa, bsb, bsc
    p.alpha, p.accumulate, p.split, p.rect = alpha, int(accumulate), split, int(rect)
    p.amax_a, p.amax_b, p.amax_c = _p(amax_a), _p(amax_b), _p(amax_c)
    p.colstats = _p(colstats)                  # zero-filled float64 [2, N]: column sums of C and C^2 (f16x3 NT only)
    p.colmax = _p(colmax)                      # zero-filled float32 [N]: column maxima of |C| (with colstats)
    p.precision = _PREC[0] + 1
    p.a_presplit, p.b_presplit = int(a_packed), int(b_packed)
    ws = None
    if mode == "tn" and split > 1 and TWO_STAGE_SPLITK:
        # two-stage reduction: the slices store partial sums, a second kernel adds them in a fixed order -- no atomics, no
        # zero-filled C, bitwise reproducible gradients
        nbytes = int(lib.glf_gemm_tn_workspace_bytes(C.byref(p)))
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=Cm.device)
        p.workspace, p.workspace_bytes = _p(ws), nbytes
    prof = PROFILER
    if prof is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    if mode == "nt":
        check(lib.glf_gemm_nt(_p(A), _p(B), _p(bias), _p(Cm), C.byref(p), _stream()), "gemm_nt")
    elif mode == "nn":
        check(lib.glf_gemm_nn(_p(A), _p(B), _p(bias), _p(Cm), C.byref(p), _stream()), "gemm_nn")
    elif mode == "tn":
        check(lib.glf_gemm_tn(_p(A), _p(B), _p(Cm), C.byref(p), _stream()), "gemm_tn")
    else:
        raise ValueError(mode)
    if prof is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        # dense, reference-equivalent FLOPs of this launch (every tap, padding included), the FLOPs of the taps the
        # host-side mask keeps, and the ALGORITHMIC bytes: every operand element read once, every result written once
        kept_taps = bin(mask).count("1")
        dense = 2.0 * M * N * K * taps * batch
        src_rows = (geo[0] * geo[1] * geo[2]) if geo else None
        # per-tap rectangle / region launches contract only the in-range part of every kept tap: the EXECUTED work is the
        # kept taps' work times that fraction (ASPP rate 12: 0.51, rate 24: 0.18)
        in_range = rect_fraction(gather, geo[3], geo[4], geo[1], geo[2], geo[5], geo[6], geo[8], geo[9], mask) if (rect and geo and gather) else 1.0
        if mode == "tn":
            abytes = 4.0 * batch * (K * M + (src_rows if src_rows else K) * N + M * N * kept_taps)
        else:
            abytes = 4.0 * batch * ((src_rows if src_rows else M) * K + N * K * kept_taps + M * N * (2 if accumulate else 1))
        prof.append((KERNEL_NAMES[(mode, gather != 0)], dense, dense * kept_taps / taps * in_range, ev0, ev1,
                     (M, N, K, taps, kept_taps, batch, split, geo[8] if geo else 0, geo[9] if geo else 0), abytes))


# ----------------------------------------------------------------------------------------
# weight-derived images: caches + the multi-tensor refresh (include/glfusion.h: glf_weights_refresh)
# ----------------------------------------------------------------------------------------
# Everything derived from a parameter -- its max magnitude (f16x3 / f16), the tap-major re-layouts of a k x k weight, the
# transposed copy of a 1x1 / linear weight, the stacked theta | phi | g operand, and the packed pre-split images of those --
# is cached against the parameter's version counter and rebuilt after an optimizer step.  One image at a time that is a
# launch per image (~1 000 per update for the 3-view model); every image therefore also registers itself as a JOB of a device
# table, and refresh_weights() recomputes all registered images in four launches.  An image keeps its buffer for the life of
# its parameter (a miss recomputes IN PLACE), so the table -- and a hipGraph that captured a refresh -- stay valid.
class _WImage:
    __slots__ = ("key", "owner", "kind", "src", "dst", "amax", "dims", "version", "extra", "__weakref__")


class _WeightRegistry:
    ARENA = 1 << 14

    def __init__(self, dev):
        self.dev = dev
        self.images = {}                   # key -> _WImage
        self.dirty = True
        self.arena = None                  # float32 [ARENA]: one slot per measured parameter
        self.free = []
        self.table = None                  # (device table, pass_first, pass_count, pass_wgs, n_jobs)

    def slot(self) -> torch.Tensor:
        if self.arena is None:
            self.arena = zeros(self.ARENA, device=self.dev)
            self.free = list(range(self.ARENA - 1, -1, -1))
        if not self.free:
            raise RuntimeError("glfusion_amd: weight amax arena exhausted (more than 16384 live parameters on one device)")
        i = self.free.pop()
        return self.arena[i:i + 1]

    def drop(self, key) -> None:
        try:
            im = self.images.pop(key, None)
        except (AttributeError, TypeError):      # interpreter shutdown: module globals are already gone
            return
        if im is not None:
            self.dirty = True
            if im.kind == 1 and self.arena is not None and im.amax is not None:        # 1 == WJ_AMAX (a literal: see drop at shutdown)
                self.free.append(int((im.amax.data_ptr() - self.arena.data_ptr()) // 4))

    def build(self, only=None):
        """Plan the job table of the registered images and copy it to the device.  only = None: all of them, installed as the
        registry's own table (rebuilt whenever an image is added or dropped; its refresh zero-fills the whole amax arena).
        only = a set of owner ids: a FROZEN table of those owners' images, returned to the caller together with strong
        references to every buffer it names -- what a captured hipGraph replays against (engine.StepGraph), valid however
        the registry changes afterwards; it zeroes exactly its own amax slots (GLF_WJ_ZERO jobs)."""
        ims = [im for im in self.images.values() if only is None or _image_belongs(im, only)]
        jobs = []                                  # (kind, src ptr, dst ptr, amax ptr, dims)
        for im in ims:
            jobs.append((im.kind, im.src.data_ptr(), im.dst.data_ptr() if im.dst is not None else 0,
                         im.amax.data_ptr() if im.amax is not None else 0, im.dims))
            if only is not None and im.kind == WJ_AMAX:
                jobs.append((WJ_ZERO, 0, im.amax.data_ptr(), 0, (1, 0, 0)))
        jobs.sort(key=lambda j: WJ_PASS_OF[j[0]])
        n = len(jobs)
        arr = (WeightJob * max(n, 1))()
        for j, (kind, src, dst, am, dims) in zip(arr, jobs):
            j.src, j.dst, j.amax, j.kind, j.pass_ = src, dst, am, kind, WJ_PASS_OF[kind]
            j.d0, j.d1, j.d2 = dims
        pf, pc, pw = (C.c_int * WJ_PASSES)(), (C.c_int * WJ_PASSES)(), (C.c_int64 * WJ_PASSES)()
        check(lib.glf_weights_plan(C.cast(arr, C.c_void_p), n, C.cast(pf, C.c_void_p), C.cast(pc, C.c_void_p), C.cast(pw, C.c_void_p)), "weights_plan")
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        table = (host.to(self.dev), pf, pc, pw, n)
        if only is not None:
            return table + ([(im.src, im.dst, im.amax) for im in ims], self.arena)
        self.table = table
        self.dirty = False
        return None

    def refresh(self, frozen=None) -> None:
        if frozen is not None:                     # a frozen table: no bookkeeping (its owner re-stamps nothing; eager callers
            tab, pf, pc, pw, n = frozen[:5]        # of the same parameters find their caches stale and rebuild in place)
            if n:
                check(lib.glf_weights_refresh(_p(tab), C.cast(pf, C.c_void_p), C.cast(pc, C.c_void_p), C.cast(pw, C.c_void_p),
                                              None, 0, _stream()), "weights_refresh")
            return
        if torch.cuda.is_current_stream_capturing() and self.dirty:
            raise RuntimeError("glfusion_amd: the weight-image table changed inside a graph capture (freeze a table first)")
        if self.dirty:
            self.build()
        tab, pf, pc, pw, n = self.table
        if n:
            check(lib.glf_weights_refresh(_p(tab), C.cast(pf, C.c_void_p), C.cast(pc, C.c_void_p), C.cast(pw, C.c_void_p),
                                          _p(self.arena), self.ARENA if self.arena is not None else 0, _stream()), "weights_refresh")
        for im in self.images.values():
            o = im.owner()
            if o is not None:
                im.version = _wversion(o, im)
                if im.kind == WJ_AMAX:
                    o._glf_amax = (o._version, o.data_ptr(), im.amax)


def _image_belongs(im, owner_ids) -> bool:
    """True when the image derives from one of the given parameters: directly, or through a stacked operand assembled from
    them (fusion._qkv_weights marks it with `_glf_sources`)."""
    o = im.owner()
    if o is None:
        return False
    if id(o) in owner_ids:
        return True
    srcs = getattr(o, "_glf_sources", None)
    return srcs is not None and any(r() is not None and id(r()) in owner_ids for r in srcs)


_wreg = {}


def _registry(dev) -> _WeightRegistry:
    r = _wreg.get(dev)
    if r is None:
        r = _wreg[dev] = _WeightRegistry(dev)
    return r


def _wversion(owner, im=None):
    """Version stamp of an image owner: the tensor's in-place counter, or -- for a dense operand ASSEMBLED from several
    parameters (fusion._qkv_weights sets `_glf_version_fn` on it) -- the combined stamp of its sources."""
    fn = getattr(owner, "_glf_version_fn", None)
    return fn() if fn is not None else owner._version


def _wimage(owner: torch.Tensor, tag: str, kind: int, src: torch.Tensor, dims, make_dst, amax: Optional[torch.Tensor] = None):
    """The cached image (owner, tag): (image, fresh) with fresh = False when the cached contents are current.  A stale or new
    image must be (re)computed by the caller into image.dst -- the buffer of a stale image is reused."""
    reg = _registry(owner.device)
    key = (id(owner), tag)
    im = reg.images.get(key)
    ver = _wversion(owner)
    if im is not None and im.owner() is owner and im.src.data_ptr() == src.data_ptr() and im.dims == tuple(dims) and \
            (im.amax is amax or kind == WJ_AMAX):
        if im.version == ver:
            return im, False
        im.version = ver
        return im, True
    if im is not None:
        reg.drop(key)
    im = _WImage()
    im.key, im.kind, im.src, im.dims, im.version, im.extra = key, kind, src, tuple(dims), ver, None
    im.owner = weakref.ref(owner, lambda _r, k=key, r=reg: r.drop(k))
    im.amax = reg.slot() if kind == WJ_AMAX else amax
    im.dst = make_dst() if make_dst is not None else None
    reg.images[key] = im
    reg.dirty = True
    return im, True


def refresh_weights(frozen=None) -> None:
    """Recompute every registered weight-derived image from the current parameter values (four launches per device on the
    current stream) and mark the caches current.  Call after an optimizer step wrote the parameters (glfusion_amd.optim.Adam
    does); anything not registered yet is still rebuilt lazily by its cache.  frozen: a table from freeze_weight_table() --
    only that table's images are recomputed (no cache bookkeeping): the form a captured step replays."""
    if frozen is not None:
        for dev, tab in frozen.items():
            _registry(dev).refresh(tab)
        return
    for reg in list(_wreg.values()):
        if reg.images:
            reg.refresh()


def reset_weight_images() -> None:
    """Forget every registered weight-derived image (they are rebuilt lazily by their caches): after a change of precision the
    images of the previous one would otherwise keep being refreshed with every update."""
    for reg in _wreg.values():
        for key in list(reg.images):
            reg.drop(key)


def freeze_weight_table(params) -> dict:
    """{device: frozen job table} of every image currently registered for the given parameters (run a step first so that
    they exist).  The tables hold their buffers alive; pass the dict to refresh_weights(frozen=...)."""
    ids = {id(p) for p in params}
    out = {}
    for dev, reg in _wreg.items():
        tab = reg.build(only=ids)
        if tab[4]:
            out[dev] = tab
    return out


def _is_weight(t: torch.Tensor) -> bool:
    return isinstance(t, torch.nn.Parameter) or hasattr(t, "_glf_version_fn")


def weight_amax(t: torch.Tensor) -> Optional[torch.Tensor]:
    """amax_of for a parameter (or a dense operand assembled from parameters): the scalar lives in the registry's arena."""
    if _PREC[0] < 2 or not t.is_contiguous():
        return None
    im, fresh = _wimage(t, "amax", WJ_AMAX, t, (t.numel(), 0, 0), None)
    if fresh:
        zero_(im.amax)
        check(lib.glf_amax(_p(t), 1, t.numel(), t.numel(), _p(im.amax), _stream()), "amax")
    return im.amax


def amax_of(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Device scalar max|t| for the f16x3 contraction kernels (None under the other precisions, where it is not
    used).  Measured once per tensor and version and remembered on the tensor object; a permutation of the
    elements (weight re-layouts) has the same maximum, so callers pass the owning parameter.  Only call it on
    tensors whose contents are final (raw kernel writes do not bump torch's version counter)."""
    if t is None or _PREC[0] < 2:
        return None
    if _is_weight(t) and t.is_contiguous():
        return weight_amax(t)
    hit = getattr(t, "_glf_amax", None)
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr():
        return hit[2]
    out = torch.empty(1, dtype=torch.float32, device=t.device)
    if t.is_contiguous():
        check(lib.glf_amax(_p(t), 1, t.numel(), t.numel(), _p(out), _stream()), "amax")
    elif t.dim() == 2 and t.stride(1) == 1:
        check(lib.glf_amax(_p(t), t.shape[0], t.shape[1], t.stride(0), _p(out), _stream()), "amax")
    else:
        return None                      # the library measures the operand region itself
    try:
        t._glf_amax = (t._version, t.data_ptr(), out)
    except AttributeError:
        pass
    return out


_amax_pool = {}


def amax_slot(dev) -> Optional[torch.Tensor]:
    """A zeroed device float for a kernel that reports max|output| as a by-product (None unless f16x3 is
    active).  Slots come from a pre-zeroed pool per stream (one fill kernel per 4096 slots); a used-up pool stays
    alive through the slices that reference it."""
    if _PREC[0] < 2:
        return None
    key = torch.cuda.current_stream().cuda_stream          # the fill kernel and the users must share a stream
    pool = _amax_pool.get(key)
    if pool is None or pool[1] >= 4096 or pool[0].device != dev:
        pool = _amax_pool[key] = [zeros(4096, device=dev), 0]
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


_stats_pool = {}
_STATS_POOL_DOUBLES = 1 << 19


def stats_slot(c: int, dev) -> torch.Tensor:
    """A zeroed float64 [2, c] for a contraction's fused column statistics (glf_gemm_params.colstats), carved out of a
    per-stream pool that is zeroed in bulk (one fill per ~500 k doubles instead of one tiny fill per conv); a used-up pool
    stays alive through the slices that reference it."""
    key = torch.cuda.current_stream().cuda_stream
    need = 2 * c
    pool = _stats_pool.get(key)
    if pool is None or pool[1] + need > pool[0].numel() or pool[0].device != dev:
        pool = _stats_pool[key] = [zeros(max(_STATS_POOL_DOUBLES, need), dtype=torch.float64, device=dev), 0]
    i = pool[1]
    pool[1] = i + need
    return pool[0][i:i + need].view(2, c)


def bnbwd_slot(c: int, dev) -> torch.Tensor:
    """Zeroed scratch of glf_bn_bwd's two-launch form (fused_sums): 2 c doubles of sums followed by 2 c floats of maxima, carved out of
    the pooled, bulk-zeroed statistics buffer (stats_slot)."""
    return stats_slot((3 * c + 1) // 2, dev).view(-1)


FUSED_BN_BWD = os.environ.get("GLF_FUSED_BN_BWD", "1") != "0"      # BatchNorm backward in two launches (atomics) instead of three


_colmax_pool: dict = {}


def colmax_slot(c: int, dev) -> torch.Tensor:
    """A zeroed float32 [c] for a contraction's per-column maxima (glf_gemm_params.colmax), pooled like stats_slot."""
    key = torch.cuda.current_stream().cuda_stream
    pool = _colmax_pool.get(key)
    if pool is None or pool[1] + c > pool[0].numel() or pool[0].device != dev:
        pool = _colmax_pool[key] = [zeros(max(1 << 16, c), dtype=torch.float32, device=dev), 0]
    i = pool[1]
    pool[1] = i + c
    return pool[0][i:i + c]


def amax_bound(t: torch.Tensor, srcs: Sequence[Optional[torch.Tensor]], scale: float = 1.0, sum_: bool = False) -> None:
    """Attach to t an upper bound of its maximum derived from the maxima of the tensors it was made from (glf_amax_combine: one
    tiny launch per pair, no pass over the data): scale * max(srcs) or scale * sum(srcs).  Does nothing unless every source's
    maximum is known (t is then measured on first use, as before)."""
    if _PREC[0] < 2:
        return
    ams = [getattr(x, "_glf_amax", None) for x in srcs]
    ams = [h[2] if (h is not None and h[0] == x._version and h[1] == x.data_ptr()) else None for h, x in zip(ams, srcs)]
    if not ams or any(a is None for a in ams):
        return
    out = amax_slot(t.device)
    if sum_:
        if len(ams) != 2:
            raise ValueError("amax_bound: sum_ takes exactly two sources")
        check(lib.glf_amax_combine(_p(ams[0]), _p(ams[1]), float(scale), 1, _p(out), _stream()), "amax_combine")
    else:
        for i in range(0, len(ams), 2):
            check(lib.glf_amax_combine(_p(ams[i]), _p(ams[i + 1]) if i + 1 < len(ams) else None, float(scale), 0, _p(out), _stream()), "amax_combine")
    set_amax(t, out)


def set_amax(t: torch.Tensor, amax: Optional[torch.Tensor]) -> None:
    """Attach a maximum produced as a by-product of the kernel that wrote t."""
    if amax is not None:
        t._glf_amax = (t._version, t.data_ptr(), amax)


_TN_TARGET = int(os.environ.get("GLF_TN_TARGET", "0"))      # 0: per precision (below)
_TN_ROUND = os.environ.get("GLF_TN_ROUND", "1") != "0"


def _tn_split(rows: int, m: int, n: int, ntaps: int, batch: int = 1) -> int:
    """Reduction slices for the TN (wgrad) kernel: fill ~4 waves of 512 resident workgroups,
    keep >= 16 K-tiles (512 rows) per slice."""
    tiles = ((m + 127) // 128) * ((n + 127) // 128) * max(ntaps, 1) * batch
    # workgroups to aim for: 2048 128x128 tiles on the fp32 / bf16x6 kernels (two per CU); the f16x3 kernel has 256-wide
    # tiles, one workgroup per CU, and shares the chip with other streams: 1024 measured best (293.5 vs 299.2 ms / step)
    target = _TN_TARGET or (1024 if _PREC[0] >= 2 else 2048)
    want = max(1, (target + tiles - 1) // tiles)
    cap = max(1, rows // 512)
    hi = 65535 // max(batch, 1)
    if _TN_ROUND and _PREC[0] >= 2:
        # among the slice counts around the target, the one whose workgroups (256-wide tiles, one per CU) fill whole
        # rounds of the 256 CUs best
        wg = ((m + 255) // 256) * ((n + 127) // 128) * max(ntaps, 1) * batch
        best, best_score = want, -1.0
        for s in range(max(1, want // 2), max(1, min(want * 2, cap, hi)) + 1):
            b = wg * s
            score = b / (((b + 255) // 256) * 256.0) - 0.004 * s
            if score > best_score:
                best, best_score = s, score
        want = best
    return int(max(1, min(want, cap, hi)))


# ----------------------------------------------------------------------------------------
# gradient slots: a parameter's gradient produced directly inside its all-reduce bucket (glfusion_amd.ddp)
# ----------------------------------------------------------------------------------------
_grad_slots = {}


def register_grad_slot(param: torch.Tensor, flat: torch.Tensor, offset: int) -> None:
    _grad_slots[id(param)] = [weakref.ref(param), flat, int(offset), False]


def unregister_grad_slots(params) -> None:
    for p in params:
        _grad_slots.pop(id(p), None)


def release_grad_slots(params) -> None:
    """End of a step: every slot may be handed out again."""
    for p in params:
        s = _grad_slots.get(id(p))
        if s is not None:
            s[3] = False


def grad_out(param: Optional[torch.Tensor], shape, device) -> torch.Tensor:
    """The tensor a backward kernel writes `param`'s gradient into: the parameter's slice of its all-reduce bucket when a
    GradAllReducer registered one and nobody took it yet this step (a parameter used several times per forward gets ONE
    in-bucket gradient -- autograd sums the others onto it), a fresh tensor otherwise.  The view is new every time, so
    autograd's AccumulateGrad can adopt it as .grad without a copy."""
    if param is not None and _grad_slots:
        s = _grad_slots.get(id(param))
        if s is not None and s[0]() is param and not s[3]:
            n = 1
            for d in shape:
                n *= int(d)
            if n == param.numel() and s[1].device == device and not torch.cuda.is_current_stream_capturing():
                s[3] = True
                return s[1][s[2]:s[2] + n].view(tuple(shape))
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)


def wgrad_split(rows_o: int, frac: float, cout: int, cin: int, ntap: int, rect: bool) -> int:
    """Reduction slices of a convolution's weight gradient (mirror of conv_api.hip's plan).  A slice is ceil(rows / split) rows
    of EVERY tap's reduction; in rect mode (ASPP: most taps reach only a small rectangle of the map) a tap uses as many slices
    as its rectangle is long, so the count is scaled up by the in-range fraction to keep the launch at its target size -- and
    all workgroups reduce the same number of rows (before: every tap was cut into `split` pieces of its own rectangle, the
    centre tap's workgroups ran 3 x (rate 12) to 50 x (rate 24) longer than the corner taps')."""
    split = _tn_split(max(512, int(rows_o * frac)) if rect else rows_o, cout, cin, ntap)
    if rect:
        # (measured on the 2048 -> 256 ASPP convs, profiles/r03_aspp_wgrad_probe.txt: half of split / frac is the sweet spot --
        # rate 12: 7 slices 1.01 ms, 14 slices 1.22 ms; rate 24: 19 slices 0.48 ms, 39 slices 0.50 ms, 78 slices 0.57 ms)
        s2 = max(split, int(split / (2.0 * max(frac, 0.02)) + 0.999))
        split = max(1, min(s2, max(1, rows_o // 512), 65535))
    return split


_ones_cache = {}


def _ones4(dev) -> torch.Tensor:
    t = _ones_cache.get(dev)
    if t is None:
        t = _ones_cache[dev] = torch.ones(4, dtype=torch.float32, device=dev)      # once per process
    return t


def colsum(dy2d: torch.Tensor, rows: int, c: int) -> torch.Tensor:
    """db[c] = sum_r dy[r][c] (bias gradients)."""
    db = torch.empty(c, dtype=torch.float32, device=dy2d.device)
    if c % 4 == 0:
        check(lib.glf_colsum(_p(dy2d), c, _p(db), rows, c, _p(_ws(rows, c, dy2d.device)), _stream()), "colsum")
    else:
        # tiny channel counts (the 5- and 1-channel heads): reduce with the TN contraction against a
        # broadcast 1.0 (row stride 0)
        one = _ones4(dy2d.device)
        sp = _tn_split(rows, c, 1, 1)
        if tn_needs_zero(sp):
            zero_(db)
        gemm("tn", dy2d, one, db, M=c, N=1, K=rows, lda=c, ldb=0, ldc=1, split=sp)
    return db


def split_mode() -> bool:
    """True when contractions run on a split (bf16x6 / f16x3) kernel family, which has NT and TN forms only."""
    return _PREC[0] >= 1


def transpose2d(x: torch.Tensor, rows: int, cols: int, batch: int = 1) -> torch.Tensor:
    """[batch][rows][cols] -> [batch][cols][rows] (fresh tensor)."""
    out = torch.empty(batch * rows * cols, dtype=torch.float32, device=x.device)
    check(lib.glf_transpose2d(_p(x), _p(out), rows, cols, batch, _stream()), "transpose2d")
    return out


# Pre-split operands (split-fp16 precisions): an operand is split into its fp16 pieces ONCE, into a packed image of the same
# size and strides, instead of in the staging path of every tile of every launch that reads it.
PRESPLIT = os.environ.get("GLF_PRESPLIT", "1") != "0"
# An activation / gradient tensor is worth its split pass (one read + one write of the tensor) only if the contractions that
# read it re-split it often enough: the pass costs ~ rows x K x 8 bytes of HBM traffic, the NT kernel saves ~14 % and the
# weight-gradient kernel ~15 % of a time proportional to rows x K x (output columns x taps).  Break-even measured near
# columns x taps = 1000.  (Weights are always pre-split: once per update, cached.)
PRESPLIT_MIN_COLS = int(os.environ.get("GLF_PRESPLIT_MIN_COLS", "1024"))
WGRAD_STREAM = os.environ.get("GLF_WGRAD_STREAM", "0") != "0"      # a conv's weight gradient on a side stream of its dgrad
_wgrad_streams = {}


def presplit_ok(t: torch.Tensor, amax: Optional[torch.Tensor]) -> bool:
    return (PRESPLIT and _PREC[0] >= 2 and amax is not None and t.dim() >= 1 and t.shape[-1] % 4 == 0
            and t.is_contiguous() and t.data_ptr() % 16 == 0)


def packed_of(t: torch.Tensor, amax: torch.Tensor) -> torch.Tensor:
    """The packed pre-split image of a contiguous fp32 tensor whose rows (last dim) hold a multiple of 4 elements: a tensor of
    the same shape (declared fp32, holding fp16 pairs) to pass as A / B with a_packed / b_packed and the SAME amax."""
    cols = t.shape[-1]
    out = torch.empty_like(t)
    check(lib.glf_split_f16_packed(_p(t), t.numel() // cols, cols, cols, _p(amax), _p(out), cols, _stream()), "split_f16_packed")
    return out


def weight_packed(layout: torch.Tensor, owner: torch.Tensor, tag: str, amax: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """packed_of(layout) for a dense weight layout derived from the parameter `owner`, made once per weight update (cached
    against the owner's version counter and the amax scalar it was scaled with).  None when not applicable."""
    if not presplit_ok(layout, amax):
        return None
    im, fresh = _wimage(owner, "pk:" + tag, WJ_PACK, layout, (layout.numel(), 0, 0), lambda: torch.empty_like(layout), amax)
    if fresh:
        cols = layout.shape[-1]
        check(lib.glf_split_f16_packed(_p(layout), layout.numel() // cols, cols, cols, _p(amax), _p(im.dst), cols, _stream()), "split_f16_packed")
    return im.dst


def packed_hit(t: torch.Tensor, amax: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The pre-split image already made of t (by an earlier consumer of this tensor object, or of another alias handed out by
    fan_out), or None."""
    hit = getattr(t, "_glf_packed", None)
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr() and hit[2] is amax:
        pk, made_on = hit[3], (hit[4] if len(hit) > 4 else None)
        if made_on is not None:
            cur = torch.cuda.current_stream()
            if made_on.cuda_stream != cur.cuda_stream:   # a consumer on another side stream: order it behind the split pass
                cur.wait_stream(made_on)
                pk.record_stream(cur)
        return pk
    share = getattr(t, "_glf_pack_share", None)          # the aliases of one fan_out share one image
    if share is not None and share[0] is not None and share[0][0] == t.data_ptr() and share[0][1] is amax:
        pk, made_on = share[0][2], share[0][3]
        cur = torch.cuda.current_stream()
        if made_on.cuda_stream != cur.cuda_stream:       # the aliases' consumers may sit on different side streams
            cur.wait_stream(made_on)
            pk.record_stream(cur)
        return pk
    return None


# A pre-split image kept from the forward pass to the backward pass is a second copy of that activation.  At C2 that is 33 GB
# on top of 56 GB; at the 5-view 224^2 shape it took the step from 180 GB to 264 GB of the 288 GB and the allocator started
# thrashing (1.16 s -> 4.0 s per step).  So images are RETAINED (on the tensor object, for the weight gradient) only while the
# device has room: while the caching allocator has RESERVED less than this fraction of the device memory (reserved, not
# allocated: with one block pool per stream the reserved figure runs far ahead -- at that shape 205 GB allocated already
# thrashed).  Above it an image lives for the launches of one autograd node: 1.02 s per step there, against 1.12 s without
# any pre-split.
PRESPLIT_KEEP_FRAC = float(os.environ.get("GLF_PRESPLIT_KEEP_FRAC", "0.5"))
PRESPLIT_OFF_FRAC = float(os.environ.get("GLF_PRESPLIT_OFF_FRAC", "0.6"))
_dev_total = {}
_retain_off = {}
_retain_step = {}


def retain_ok(dev) -> bool:
    """Retained images sit in memory across the peak of the step (end of forward), so the decision cannot be made tensor by
    tensor: once the allocator has been seen above PRESPLIT_OFF_FRAC of the device, retention is off for the rest of the
    process (the workload does not fit with second copies; the first step pays for finding out)."""
    hit = _retain_step.get(dev)
    if hit is not None:                       # decided at the first pack of this step (begin_step clears it): no allocator
        return hit                            # queries per conv, and the step does not change its mind half-way
    tot = _dev_total.get(dev)
    if tot is None:
        tot = _dev_total[dev] = torch.cuda.get_device_properties(dev).total_memory
    off = _retain_off.get(dev, 0)
    if off:
        if off == 1 and torch.cuda.memory_allocated(dev) < 0.3 * tot and not torch.cuda.is_current_stream_capturing():
            # first pack of a later step: the images retained before the switch are gone -- hand the pools they inflated
            # back to the device once (a later step needs less than the one that found out)
            torch.cuda.empty_cache()
            _retain_off[dev] = 2
        _retain_step[dev] = False
        return False
    r = torch.cuda.memory_reserved(dev)
    if r > PRESPLIT_OFF_FRAC * tot:
        _retain_off[dev] = 1                  # this step keeps what it already retained; the next ones retain nothing
        _retain_step[dev] = False
        return False
    ok = r < PRESPLIT_KEEP_FRAC * tot
    if ok:
        _retain_step[dev] = True              # a "no" below the off threshold is re-examined at the next pack
    return ok


def act_packed(t: torch.Tensor, amax: Optional[torch.Tensor], retain: Optional[bool] = None) -> Optional[torch.Tensor]:
    """packed_of(t) for an activation / gradient tensor, remembered on the tensor object (like its amax) so that every
    contraction that reads it -- a conv input in the forward of each consumer and again in its weight gradient, an output
    gradient in dgrad and wgrad -- shares one split pass.  retain: keep the image alive with t (default: while the device
    has memory to spare, see retain_ok; gradients pass True: they die with their autograd node).  None when not applicable."""
    if not presplit_ok(t, amax):
        return None
    pk = packed_hit(t, amax)
    if pk is not None:
        return pk
    pk = packed_of(t, amax)
    if retain is None:
        retain = retain_ok(t.device)
    if retain:
        try:
            t._glf_packed = (t._version, t.data_ptr(), amax, pk, torch.cuda.current_stream() if STREAMS else None)
        except AttributeError:
            pass
        share = getattr(t, "_glf_pack_share", None)
        if share is not None:
            share[0] = (t.data_ptr(), amax, pk, torch.cuda.current_stream())
    return pk


def nt_presplit_ok(K: int, lda: int, ldb: int) -> bool:
    """Mirror of the library's eligibility test for the aligned split-fp16 NT kernel (gemm_f16s.hip f16s_rows_ok)."""
    return _PREC[0] >= 2 and K % 32 == 0 and 32 <= K <= (1 << 18) and lda % 4 == 0 and ldb % 4 == 0


def tn_presplit_ok(M: int, N: int, lda: int, ldb: int) -> bool:
    """Same for the TN kernel (f16s_tn_ok)."""
    return _PREC[0] >= 2 and M % 4 == 0 and N % 4 == 0 and 4 <= M <= (1 << 18) and 4 <= N <= (1 << 18) and lda % 4 == 0 and ldb % 4 == 0


def pick(t: torch.Tensor, packed: Optional[torch.Tensor], ok: bool):
    """(operand, packed flag) for a gemm() call."""
    return (packed, True) if (ok and packed is not None) else (t, False)


def weight_T(w2d: torch.Tensor, owner: torch.Tensor) -> torch.Tensor:
    """Transposed copy [cols][rows] of a 2-D weight view, cached against the owning parameter's version."""
    rows, cols = w2d.shape
    src = _contig(w2d)
    im, fresh = _wimage(owner, "T2", WJ_TRANSPOSE, src, (rows, cols, 0),
                        lambda: torch.empty(cols, rows, dtype=torch.float32, device=owner.device))
    if fresh:
        check(lib.glf_transpose2d(_p(src), _p(im.dst), rows, cols, 1, _stream()), "transpose2d")
    return im.dst


def tap_major_T(weight: torch.Tensor) -> torch.Tensor:
    """[tap][Cin][Cout] re-layout (dgrad as an NT contraction), cached like tap_major."""
    co, ci, kh, kw = weight.shape
    src = _contig(weight.detach())
    taps = kh * kw
    mk = lambda: torch.empty(taps, ci, co, dtype=torch.float32, device=weight.device)
    if taps == 1:                      # a plain transpose [co][ci] -> [ci][co]
        im, fresh = _wimage(weight, "tapT", WJ_TRANSPOSE, src, (co, ci, 0), mk)
        if fresh:
            check(lib.glf_transpose2d(_p(src), _p(im.dst), co, ci, 1, _stream()), "transpose2d")
        return im.dst
    im, fresh = _wimage(weight, "tapT", WJ_TAP_MAJOR_T, src, (co, ci, taps), mk)
    if fresh:
        check(lib.glf_oihw_to_tap_major_t(_p(src), _p(im.dst), co, ci, taps, _stream()), "oihw_to_tap_major_t")
    return im.dst


def tap_major(weight: torch.Tensor) -> torch.Tensor:
    """OIHW -> [tap][Cout][Cin], recomputed only when the parameter changes (a view for 1x1 weights)."""
    co, ci, kh, kw = weight.shape
    if kh * kw == 1:
        return _contig(weight.detach()).view(1, co, ci)
    src = _contig(weight.detach())
    im, fresh = _wimage(weight, "tap", WJ_TAP_MAJOR, src, (co, ci, kh * kw),
                        lambda: torch.empty(kh * kw, co, ci, dtype=torch.float32, device=weight.device))
    if fresh:
        check(lib.glf_oihw_to_tap_major(_p(src), _p(im.dst), co, ci, kh * kw, _stream()), "oihw_to_tap_major")
    return im.dst


# ----------------------------------------------------------------------------------------
# conv2d (NHWC, implicit GEMM)
# ----------------------------------------------------------------------------------------
_rect_cache = {}


def rect_fraction(gather: int, hd: int, wd: int, hs: int, ws: int, kh: int, kw: int, pad: int, dil: int, mask: int) -> float:
    """sum over kept taps of (in-range rectangle area) / (kept taps x full area), stride 1 (mirrors tap_rect
    in gemm_f32.hip).  Small values = most tap work is padding => the tap-parallel rect mode pays."""
    key = (gather, hd, wd, hs, ws, kh, kw, pad, dil, mask)
    f = _rect_cache.get(key)
    if f is None:
        tot, n = 0, 0
        for t in range(kh * kw):
            if not (mask >> t) & 1:
                continue
            ky, kx = divmod(t, kw)
            oy = pad - ky * dil if gather == 1 else ky * dil - pad
            ox = pad - kx * dil if gather == 1 else kx * dil - pad
            rh = max(0, min(hs + oy, hd) - max(oy, 0))
            rw = max(0, min(ws + ox, wd) - max(ox, 0))
            tot += rh * rw
            n += 1
        f = tot / max(1, n * hd * wd)
        _rect_cache[key] = f
    return f


# use rect mode when less than this fraction of the kept taps' work is in range (per conv pass)
# The rectangle GEMMs sum their taps with float atomics; dgrad writes the widest outputs (Cin = 2048 columns for
# ASPP) over the shortest reductions (Cout = 256), so once the MFMA work is cheap (f16x3) the atomics cost more
# than the padding work they avoid unless most of it is padding: measured on the ASPP shapes, rate 12 (51 % in
# range) is faster dense, rate 24 (18 %) faster as rectangles.
RECT_THRESHOLD = {"fwd": 0.8, "dgrad": 0.8, "wgrad": 0.8, "dgrad_f16x3": 0.35, "region": 0.8}


def region_mode(taps: int, kh: int, stride: int, pad: int, dil: int, h: int, w: int, ho: int, wo: int, k: int, frac: float) -> bool:
    """rect = 2 of glf_gemm_nt (f16x3 kernels): a 3x3 stride-1 "same" conv (pad == dil) is cut into <= 9 rectangles of
    output pixels with a constant set of in-range taps -- no padding work, no atomics, no zero fill.  Used for the
    dgrad of the ASPP rate-12 / 24 convs (2048-column outputs): 14.4 -> 11.2 ms and 7.6 -> 6.7 ms per step against
    dense / per-tap rectangles with atomics.  Measured NOT to pay elsewhere: the forward of the same convs (256-column
    outputs, 392 workgroups) is faster as per-tap rectangles -- the regions' blocks are fewer and of uneven length
    (4 / 6 / 9 taps) -- and for dilations 1-4 the 5-18 % of padding work saved is less than the extra partial tiles
    and the per-element pixel arithmetic of the epilogue cost."""
    return (os.environ.get("GLF_REGION", "1") != "0" and _PREC[0] >= 2 and taps == 9 and kh == 3 and stride == 1 and pad == dil and h == ho and w == wo
            and k % 32 == 0 and frac < RECT_THRESHOLD["region"])


def _rect_thr(which: str) -> float:
    if which == "dgrad" and _PREC[0] >= 2:
        return RECT_THRESHOLD["dgrad_f16x3"]
    return RECT_THRESHOLD[which]


def _conv_out(h: int, k: int, stride: int, pad: int, dil: int) -> int:
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


class Conv2dFn(Function):
    """F.conv2d on [N,H,W,Cin] with torch-layout weights [Cout,Cin,kh,kw] (groups = 1)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride: int, pad: int, dil: int, colstats=None):
        _chk(x, "conv input"); _chk(weight, "conv weight")
        x = _contig(x)
        n, h, w, cin = x.shape
        cout, cin_w, kh, kw = weight.shape
        if cin_w != cin:
            raise RuntimeError(f"conv2d: input has {cin} channels, weight expects {cin_w}")
        ho, wo = _conv_out(h, kh, stride, pad, dil), _conv_out(w, kw, stride, pad, dil)
        if ho <= 0 or wo <= 0:
            raise RuntimeError("conv2d: empty output")
        taps = kh * kw
        wt = tap_major(weight)
        y = torch.empty(n, ho, wo, cout, dtype=torch.float32, device=x.device)
        plain = taps == 1 and stride == 1 and pad == 0
        geo = (n, h, w, ho, wo, kh, kw, stride, pad, dil)
        mask = 1 if plain else tap_mask(1, ho, wo, h, w, kh, kw, stride, pad, dil)
        rect = (not plain and taps > 1 and stride == 1 and bias is None and bin(mask).count("1") > 1
                and rect_fraction(1, ho, wo, h, w, kh, kw, pad, dil, mask) < _rect_thr("fwd"))
        if rect:
            zero_(y)
            if colstats is not None:
                raise RuntimeError("conv2d: fused column statistics are not available for a conv evaluated as per-tap rectangles")
        am_w, am_x = amax_of(weight), amax_of(x)
        x_pk = packed_only(x)                # the producing BatchNorm wrote the activation as a packed image (no fp32 form exists)
        # <= 64 output rows (the ASPP pooled branch: one row per frame): glf_gemm_nt's skinny kernel takes plain fp32 operands
        skinny = plain and n * ho * wo <= 64 and cout >= 64 and cin % 16 == 0 and not x_pk and colstats is None
        ok = nt_presplit_ok(cin, cin, cin) and not skinny
        if x_pk and not (ok and am_x is not None and takes_packed_input(weight)):
            raise RuntimeError("glfusion_amd: a packed-only activation reached a convolution that cannot consume it")
        ok_x = ok and (cout * bin(mask).count("1") >= PRESPLIT_MIN_COLS or packed_hit(x, am_x) is not None)
        xa, pa = (x, True) if x_pk else pick(x, act_packed(x, am_x) if ok_x else None, ok_x)
        wb, pb = pick(wt, weight_packed(wt, weight, "w", am_w) if ok else None, ok)
        # exact fp32 (the strict-precision leg): per-tap rectangles ONE TAP PER LAUNCH.  In one launch the taps of an output pixel meet in
        # float atomics in whatever order the workgroups finish; the order-dependent last bits of the ASPP outputs are harmless in the
        # forward pass (loss to 3e-8) and came back from this model's backward pass (BatchNorm over the N per-frame averages of the
        # pooled branch) as 2-8e-3 run-to-run changes of whole gradient tensors (profiles/r04_exact_leg_spread.txt).  Tap after tap every
        # element gets its addends in tap order: reproducible, and the rectangles' short fp32 chains are kept (a dense evaluation is a
        # 9 K-long chain and moved the train-mode logits past 1e-4).
        masks = [1 << t_ for t_ in range(taps) if (mask >> t_) & 1] if (rect and _PREC[0] == 0) else [mask]
        for mk in masks:
            gemm("nt", xa, wb, y, M=n * ho * wo, N=cout, K=cin, lda=cin, ldb=cin, ldc=cout, bias=bias,
                 taps=taps, mask=mk, tap_stride_b=cout * cin, gather=0 if plain else 1, geo=None if plain else geo, rect=rect,
                 amax_a=am_x, amax_b=am_w, colstats=colstats, colmax=getattr(colstats, "_glf_colmax", None), a_packed=pa, b_packed=pb)
        ctx.save_for_backward(x, wt)
        ctx.x_packed = (xa, am_x) if (pa and (x_pk or packed_hit(x, am_x) is not None)) else None      # retained: the weight gradient reads the same image
        ctx.join = getattr(x, "_glf_join", None) if plain else None
        ctx.weight_ref = weight            # for the cached [tap][Cin][Cout] layout of the split-bf16 dgrad
        ctx.cfg = (n, h, w, cin, cout, kh, kw, ho, wo, stride, pad, dil, plain, bias is not None, tuple(weight.shape))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        n, h, w, cin, cout, kh, kw, ho, wo, stride, pad, dil, plain, has_bias, wshape = ctx.cfg
        dy = _contig(dy)
        taps = kh * kw
        rows_o = n * ho * wo
        dx = dw = db = None
        am_dy, am_w = amax_of(dy), amax_of(ctx.weight_ref)
        dy_pk = packed_only(dy)              # BatchNorm backward wrote the gradient as a packed pre-split image (no fp32 form exists)
        if dy_pk and (has_bias or am_dy is None or not (split_mode() and cout % 32 == 0)):
            raise RuntimeError("glfusion_amd: a packed-only gradient reached a convolution that cannot consume it")
        def dgrad():
            dx = None
            if True:
                mask = 1 if plain else tap_mask(2, h, w, ho, wo, kh, kw, stride, pad, dil)
                if mask == 0:
                    dx = zeros(x.shape, device=x.device)
                else:
                    frac = 1.0 if plain or stride != 1 else rect_fraction(2, h, w, ho, wo, kh, kw, pad, dil, mask)
                    if not plain and bin(mask).count("1") > 1 and region_mode(taps, kh, stride, pad, dil, ho, wo, h, w, cout, frac):
                        rect = 2
                    else:
                        rect = int(not plain and taps > 1 and stride == 1 and bin(mask).count("1") > 1 and frac < _rect_thr("dgrad"))
                    parked = None
                    if ctx.join is not None:              # the shortcut's gradient is waiting: add this dgrad onto it in the epilogue
                        parked, ctx.join.parked = ctx.join.parked, None
                        if parked is None:
                            raise RuntimeError("glfusion_amd: gradient join reached before the shortcut's gradient was produced")
                    acc = parked is not None
                    if acc:
                        dx = parked
                    else:
                        dx = zeros(x.shape, device=x.device) if rect == 1 else torch.empty_like(x)
                    if split_mode() and cout % 32 == 0:
                        # dgrad as NT on the split-bf16 kernels: B_tap[n = ci][k = co]
                        am_dx = amax_slot(dx.device) if acc else None
                        wT = tap_major_T(ctx.weight_ref)
                        ok = nt_presplit_ok(cout, cout, cout)
                        ok_dy = ok and cin * bin(mask).count("1") >= PRESPLIT_MIN_COLS
                        da, pa = (dy, True) if dy_pk else pick(dy, act_packed(dy, am_dy, True) if ok_dy else None, ok_dy)
                        wb, pb = pick(wT, weight_packed(wT, ctx.weight_ref, "wT", am_w) if ok else None, ok)
                        gemm("nt", da, wb, dx, M=n * h * w, N=cin, K=cout, lda=cout, ldb=cout, ldc=cin,
                             taps=taps, mask=mask, tap_stride_b=cout * cin, gather=0 if plain else 2,
                             geo=None if plain else (n, ho, wo, h, w, kh, kw, stride, pad, dil), rect=rect,
                             amax_a=am_dy, amax_b=am_w, accumulate=acc, amax_c=am_dx, a_packed=pa, b_packed=pb)
                        if acc:
                            dx._glf_amax = None
                            set_amax(dx, am_dx)            # the maximum of the SUM, from the accumulating epilogue
                    else:
                        gemm("nn", dy, wt, dx, M=n * h * w, N=cin, K=cout, lda=cout, ldb=cin, ldc=cin, taps=taps, mask=mask,
                             tap_stride_b=cout * cin, gather=0 if plain else 2,
                             geo=None if plain else (n, ho, wo, h, w, kh, kw, stride, pad, dil), rect=rect, accumulate=acc)
                        if acc:
                            dx._glf_amax = None
            return dx

        def wgrad():
            dw = None
            if ctx.needs_input_grad[1]:
                mask = 1 if plain else tap_mask(1, ho, wo, h, w, kh, kw, stride, pad, dil)
                ntap = bin(mask).count("1")
                rect = (not plain and taps > 1 and stride == 1 and ntap > 1
                        and rect_fraction(1, ho, wo, h, w, kh, kw, pad, dil, mask) < _rect_thr("wgrad"))
                frac = rect_fraction(1, ho, wo, h, w, kh, kw, pad, dil, mask) if rect else 1.0
                split = wgrad_split(rows_o, frac, cout, cin, ntap, rect)
                full = mask == (1 << taps) - 1
                if taps == 1 and full and not tn_needs_zero(split):
                    dwt = grad_out(ctx.weight_ref, (1, cout, cin), x.device)       # a 1x1 weight's gradient is the contraction's output
                else:
                    dwt = (zeros if (tn_needs_zero(split) or not full) else torch.empty)(taps, cout, cin, dtype=torch.float32, device=x.device)
                ok = tn_presplit_ok(cout, cin, cout, cin)
                am_x = ctx.x_packed[1] if ctx.x_packed is not None else amax_of(x)
                # dy: the image dgrad made (or one worth making for this kernel alone); x: the image the forward made, if any
                ok_dy = ok and (packed_hit(dy, am_dy) is not None or cin * ntap >= PRESPLIT_MIN_COLS)
                da, pa = (dy, True) if dy_pk else pick(dy, act_packed(dy, am_dy, True) if ok_dy else None, ok_dy)
                xb, pb = pick(x, ctx.x_packed[0] if ctx.x_packed is not None else None, ok)
                gemm("tn", da, xb, dwt, M=cout, N=cin, K=rows_o, lda=cout, ldb=cin, ldc=cin, taps=taps, mask=mask,
                     tap_stride_b=cout * cin, gather=0 if plain else 1,
                     geo=None if plain else (n, h, w, ho, wo, kh, kw, stride, pad, dil), split=split, rect=rect,
                     amax_a=am_dy, amax_b=am_x, a_packed=pa, b_packed=pb)
                if taps == 1:
                    dw = dwt.view(wshape)
                else:
                    dw = grad_out(ctx.weight_ref, wshape, x.device)
                    check(lib.glf_tap_major_to_oihw(_p(dwt), _p(dw), cout, cin, taps, _stream()), "tap_major_to_oihw")
            return dw

        if WGRAD_STREAM and STREAMS and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # dgrad and wgrad of one conv are independent and read the same dy: the weight gradient goes to a side stream of
            # the stream this node runs on (its workgroups fill the partial last round of the dgrad kernel and vice versa)
            if not dy_pk and presplit_ok(dy, am_dy) and (cin * taps >= PRESPLIT_MIN_COLS):
                act_packed(dy, am_dy, True)                  # the shared image is made before the fork, on this stream
            cur = torch.cuda.current_stream()
            side = _wgrad_streams.get(cur.cuda_stream)
            if side is None:
                side = _wgrad_streams[cur.cuda_stream] = torch.cuda.Stream(device=dy.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                dw = wgrad()
            dx = dgrad()
            cur.wait_stream(side)
            if dw is not None:
                dw.record_stream(cur)
        else:
            if ctx.needs_input_grad[0]:
                dx = dgrad()
            if ctx.needs_input_grad[1]:
                dw = wgrad()
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy, rows_o, cout)
        return dx, dw, db, None, None, None, None


def conv2d(x, weight, bias=None, stride: int = 1, pad: int = 0, dil: int = 1, colstats=None):
    if _is16(x):
        from . import ops16
        return ops16.conv2d(x, weight, bias, stride, pad, dil, colstats)
    return Conv2dFn.apply(x, weight, bias, stride, pad, dil, colstats)


def conv_stats_fusable(weight, stride: int, pad: int, dil: int, h: int, w: int, dtype=None) -> bool:
    """True when conv2d(..., colstats=) is honoured: f16x3 kernels (Cin % 32 == 0, Cout % 4 == 0) and the conv is not
    one that runs as per-tap rectangles with atomics (ASPP rate 12 / 24 forward)."""
    if _S16[0]:                          # the 16-bit kernels always honour colstats (region mode stores every element once)
        return dtype == torch.bfloat16 and weight.shape[1] % 64 == 0 and weight.shape[0] % 64 == 0
    if _PREC[0] < 2:
        return False
    cout, cin, kh, kw = weight.shape
    if cin % 32 != 0 or cout % 4 != 0:
        return False
    taps = kh * kw
    if taps == 1 and stride == 1 and pad == 0:
        return True
    ho, wo = _conv_out(h, kh, stride, pad, dil), _conv_out(w, kw, stride, pad, dil)
    if ho <= 0 or wo <= 0:
        return False
    mask = tap_mask(1, ho, wo, h, w, kh, kw, stride, pad, dil)
    rect = (taps > 1 and stride == 1 and bin(mask).count("1") > 1
            and rect_fraction(1, ho, wo, h, w, kh, kw, pad, dil, mask) < _rect_thr("fwd"))
    return not rect


class ConvCatFn(Function):
    """1x1 conv over the channel-concatenation of several [N,H,W,Ck] inputs WITHOUT materialising
    the concat (ASPP project, deeplabv3.py:153-165): y = sum_k x_k @ W[:, slice_k]^T."""

    @staticmethod
    def forward(ctx, weight, bias, *xs):
        _chk(weight, "weight")
        cout, ctot = weight.shape[0], weight.shape[1]
        w2 = _contig(weight.detach()).view(cout, ctot)
        rows = xs[0].numel() // xs[0].shape[-1]
        # inputs that ARE the column slices of one [..., ctot] buffer (ops.output_into): one K = ctot contraction
        t0 = _chk(xs[0], "input")
        offs = [0]
        for t in xs:
            offs.append(offs[-1] + t.shape[-1])
        ctx.cat = (offs[-1] == ctot and not t0.is_contiguous() and t0.stride(-1) == 1 and t0.stride(-2) == ctot
                   and all(_chk(t, "input").stride() == t0.stride() and t.shape[:-1] == t0.shape[:-1]
                           and t.data_ptr() == t0.data_ptr() + 4 * o for t, o in zip(xs, offs)))
        if ctx.cat:
            y = torch.empty(*t0.shape[:-1], cout, dtype=torch.float32, device=t0.device)
            am_w = amax_of(weight)
            ok = nt_presplit_ok(ctot, ctot, ctot)
            wb, pb = pick(w2, weight_packed(w2, weight, "w", am_w) if ok else None, ok)
            gemm("nt", t0, wb, y, M=rows, N=cout, K=ctot, lda=ctot, ldb=ctot, ldc=cout, bias=bias,
                 amax_a=amax_of(t0), amax_b=am_w, b_packed=pb)
            ctx.save_for_backward(w2, *xs)
            ctx.wshape = tuple(weight.shape)
            ctx.weight_ref = weight
            ctx.has_bias = bias is not None
            return y
        xs = [_contig(_chk(t, "input")) for t in xs]
        y = torch.empty(*xs[0].shape[:-1], cout, dtype=torch.float32, device=xs[0].device)
        off = 0
        for i, t in enumerate(xs):
            ck = t.shape[-1]
            gemm("nt", t, w2[:, off:], y, M=rows, N=cout, K=ck, lda=ck, ldb=ctot, ldc=cout, accumulate=i > 0,
                 bias=bias if i == 0 else None, amax_a=amax_of(t), amax_b=amax_of(weight))
            off += ck
        if off != ctot:
            raise RuntimeError(f"conv_cat: inputs have {off} channels in total, weight expects {ctot}")
        ctx.save_for_backward(w2, *xs)
        ctx.wshape = tuple(weight.shape)
        ctx.weight_ref = weight
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        w2, *xs = ctx.saved_tensors
        dy = _contig(dy)
        cout, ctot = w2.shape
        rows = dy.numel() // cout
        split = _tn_split(rows, cout, xs[0].shape[-1], 1)
        dw = None
        if ctx.needs_input_grad[0]:
            dw = (zeros if tn_needs_zero(split) else torch.empty)(cout, ctot, dtype=torch.float32, device=dy.device)
        db = colsum(dy, rows, cout) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        grads = []
        off = 0
        am_dy, am_w = amax_of(dy), amax_of(ctx.weight_ref)
        dy_pk = packed_only(dy)
        if dy_pk and (not ctx.cat or ctx.has_bias or am_dy is None or not (split_mode() and cout % 32 == 0)):
            raise RuntimeError("glfusion_amd: a packed-only gradient reached a projection that cannot consume it")
        if ctx.cat:
            t0 = xs[0]
            if any(ctx.needs_input_grad[2:]):
                dcat = torch.empty(*t0.shape[:-1], ctot, dtype=torch.float32, device=dy.device)
                am_dc = amax_slot(dy.device)
                if split_mode() and cout % 32 == 0:
                    w2T = weight_T(w2, ctx.weight_ref)
                    ok = nt_presplit_ok(cout, cout, cout)
                    da, pa = (dy, True) if dy_pk else pick(dy, act_packed(dy, am_dy, True) if ok else None, ok)
                    wb, pb = pick(w2T, weight_packed(w2T, ctx.weight_ref, "T2", am_w) if ok else None, ok)
                    gemm("nt", da, wb, dcat, M=rows, N=ctot, K=cout, lda=cout, ldb=cout, ldc=ctot,
                         amax_a=am_dy, amax_b=am_w, amax_c=am_dc, a_packed=pa, b_packed=pb)
                else:
                    am_dc = None
                    gemm("nn", dy, w2, dcat, M=rows, N=ctot, K=cout, lda=cout, ldb=ctot, ldc=ctot)
                for t in xs:
                    g = dcat[..., off:off + t.shape[-1]]
                    set_amax(g, am_dc)
                    grads.append(g)
                    off += t.shape[-1]
            else:
                grads = [None] * len(xs)
            if dw is not None:
                split = _tn_split(rows, cout, ctot, 1)
                if tn_needs_zero(split):
                    zero_(dw)
                ok = tn_presplit_ok(cout, ctot, cout, ctot)
                da, pa = (dy, True) if dy_pk else pick(dy, act_packed(dy, am_dy, True) if ok else None, ok)
                gemm("tn", da, t0, dw, M=cout, N=ctot, K=rows, lda=cout, ldb=ctot, ldc=ctot, split=split,
                     amax_a=am_dy, amax_b=amax_of(t0), a_packed=pa)
            return (dw.view(ctx.wshape) if dw is not None else None, db, *grads)
        for i, t in enumerate(xs):
            ck = t.shape[-1]
            if ctx.needs_input_grad[2 + i]:
                dx = torch.empty_like(t)
                if split_mode() and cout % 32 == 0:
                    wT = weight_T(w2, ctx.weight_ref)                     # [ctot][cout]
                    gemm("nt", dy, wT[off:], dx, M=rows, N=ck, K=cout, lda=cout, ldb=cout, ldc=ck, amax_a=am_dy, amax_b=am_w)
                else:
                    gemm("nn", dy, w2[:, off:], dx, M=rows, N=ck, K=cout, lda=cout, ldb=ctot, ldc=ck)
                grads.append(dx)
            else:
                grads.append(None)
            if dw is not None:
                gemm("tn", dy, t, dw[:, off:], M=cout, N=ck, K=rows, lda=cout, ldb=ck, ldc=ctot, split=split,
                     amax_a=am_dy, amax_b=amax_of(t))
            off += ck
        return (dw.view(ctx.wshape) if dw is not None else None, db, *grads)


def conv1x1_cat(weight, xs: Sequence[torch.Tensor], bias=None, colstats=None):
    if _is16(xs[0]):
        from . import ops16
        return ops16.conv1x1_cat(weight, xs, bias, colstats)
    if colstats is not None:
        raise RuntimeError("glfusion_amd: conv1x1_cat honours colstats in 16-bit storage mode only")
    return ConvCatFn.apply(weight, bias, *xs)


# ----------------------------------------------------------------------------------------
# stem: Conv2d(1, 64, 7, stride 1, pad p) + bias   (models/_utils.py:192)
# ----------------------------------------------------------------------------------------
class StemFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad: int):
        _chk(x, "stem input"); _chk(weight, "stem weight")
        x = _contig(x)                                  # [N,H,W,1]
        n, h, w, _ = x.shape
        cout = weight.shape[0]
        y = torch.empty(n, h + 2 * pad - 6, w + 2 * pad - 6, cout, dtype=torch.float32, device=x.device)
        check(lib.glf_stem7x7_fwd(_p(x), _p(_contig(weight.detach())), _p(bias), _p(y), n, h, w, cout, pad, _stream()), "stem7x7_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (n, h, w, cout, pad, tuple(weight.shape), bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, h, w, cout, pad, wshape, has_bias = ctx.cfg
        if ctx.needs_input_grad[0]:
            raise RuntimeError("glfusion_amd: gradient w.r.t. the input image is not on the path (stem dgrad not built)")
        dy = _contig(dy)
        dw = torch.empty(wshape, dtype=torch.float32, device=dy.device)
        db = torch.empty(cout, dtype=torch.float32, device=dy.device) if has_bias else None
        part = torch.empty(int(lib.glf_stem7x7_wgrad_workspace(n, h, w, cout, pad)), dtype=torch.float32, device=dy.device)
        check(lib.glf_stem7x7_wgrad(_p(x), _p(dy), _p(dw), _p(db), _p(part), n, h, w, cout, pad, _stream()), "stem7x7_wgrad")
        return None, dw, db, None


def stem7x7(x, weight, bias, pad: int):
    if _S16[0]:                          # fp32 image in, bf16 conv output: the entry into the 16-bit domain
        from . import ops16
        return ops16.stem7x7(x, weight, bias, pad)
    return StemFn.apply(x, weight, bias, pad)


FUSED_STEM = os.environ.get("GLF_FUSED_STEM", "1") != "0"


def stem_bn_relu_pool(x, weight, bias, running_mean, running_var, gamma, beta, eps: float, pad: int):
    """Inference-mode init_block in one launch (glf_stem7x7_bn_relu_pool): maxpool3x3s2(relu(bn_eval(conv7x7(x) + bias))) on
    [N,H,W,1] -> [N,Hp,Wp,64].  Forward only (no autograd node): for no_grad evaluation."""
    _chk(x, "stem input"); _chk(weight, "stem weight")
    x = _contig(x)
    n, h, w, _ = x.shape
    cout = weight.shape[0]
    ho, wo = h + 2 * pad - 6, w + 2 * pad - 6
    dev = x.device
    mean = torch.empty(cout, dtype=torch.float32, device=dev)
    invstd = torch.empty(cout, dtype=torch.float32, device=dev)
    check(lib.glf_bn_eval_coeffs(_p(running_mean), _p(running_var), eps, _p(mean), _p(invstd), cout, _stream()), "bn_eval_coeffs")
    y = torch.empty(n, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, cout, dtype=torch.float32, device=dev)
    am = amax_slot(dev)
    check(lib.glf_stem7x7_bn_relu_pool(_p(x), _p(_contig(weight.detach())), _p(bias), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(y),
                                       n, h, w, cout, pad, _p(am), _stream()), "stem7x7_bn_relu_pool")
    set_amax(y, am)
    return y


# ----------------------------------------------------------------------------------------
# writing a branch's last kernel straight into a column slice of a wider buffer (ASPP: no concat, ONE projection GEMM)
# ----------------------------------------------------------------------------------------
_OUT_VIEW = [None]


class output_into:
    """Within the block, the next BatchNorm-apply / broadcast whose output has `view`'s shape writes into `view` (a
    [..., C] column slice of a wider channels-last buffer: row stride = the buffer's channel count) instead of a
    fresh tensor, and reports its maximum into `amax` (a slot shared by all writers of the buffer)."""

    def __init__(self, view: torch.Tensor, amax: Optional[torch.Tensor] = None):
        self.item = (view, amax)

    def __enter__(self):
        _OUT_VIEW[0] = self.item
        return self

    def __exit__(self, *exc):
        _OUT_VIEW[0] = None
        return False


def _take_out(shape, device):
    """(output tensor, row stride, shared amax slot or None) for a [..., C] result of `shape`."""
    item = _OUT_VIEW[0]
    if item is not None and tuple(item[0].shape) == tuple(shape) and item[0].stride(-1) == 1:
        _OUT_VIEW[0] = None
        return item[0], int(item[0].stride(-2)), item[1]
    return torch.empty(tuple(shape), dtype=torch.float32, device=device), int(shape[-1]), None


def _rows_view(t: torch.Tensor):
    """(tensor, row stride) for reading a [..., C] tensor as rows: contiguous, or a column slice of a wider
    channels-last buffer (uniform row stride); anything else is copied."""
    if t.is_contiguous():
        return t, int(t.shape[-1])
    if t.dim() >= 2 and t.stride(-1) == 1:
        ld = int(t.stride(-2))
        ok = all(t.stride(d) == t.stride(d + 1) * t.shape[d + 1] for d in range(t.dim() - 2))
        if ok and ld >= t.shape[-1]:
            return t, ld
    t = t.contiguous()
    return t, int(t.shape[-1])


# ----------------------------------------------------------------------------------------
# BatchNorm (+ residual, + ReLU)
# ----------------------------------------------------------------------------------------
_last_bn = [None]          # (mean, invstd, rows) of the most recent BatchNormActFn.forward (read by BN_TAP)
FUSE_BN_FINALIZE = os.environ.get("GLF_FUSE_BN_FINALIZE", "1") != "0"     # statistics finished inside the apply kernel
RELU_MASK_BYTES = os.environ.get("GLF_RELU_MASK_BYTES", "1") != "0"       # BN(+residual)+ReLU keeps sign bytes, not y, for backward


class BatchNormActFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, nbt, training: bool,
                momentum: float, eps: float, relu: bool, sums=None, packed_grad: bool = False, packed_out: bool = False):
        _chk(x, "bn input"); _chk(gamma, "bn weight"); _chk(beta, "bn bias")
        x = _contig(x)
        c = x.shape[-1]
        rows = x.numel() // c
        dev = x.device
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        fused_stats = training and sums is not None and c <= 4096 and FUSE_BN_FINALIZE
        if fused_stats:
            pass                                 # finished inside the apply kernel below (one launch)
        elif training and sums is not None:      # (sum x, sum x^2) came out of the producing contraction's epilogue
            check(lib.glf_bn_stats_from_sums(_p(sums), rows, c, eps, momentum, _p(mean), _p(invstd), _p(running_mean),
                                             _p(running_var), _p(nbt), _stream()), "bn_stats_from_sums")
        elif training:
            check(lib.glf_bn_stats(_p(x), c, rows, c, eps, momentum, _p(mean), _p(invstd), _p(running_mean), _p(running_var),
                                   _p(nbt), _p(_ws(rows, c, dev)), _stream()), "bn_stats")
        else:
            if running_mean is None or running_var is None:
                raise RuntimeError("batch_norm in eval mode needs running statistics")
            check(lib.glf_bn_eval_coeffs(_p(running_mean), _p(running_var), eps, _p(mean), _p(invstd), c, _stream()), "bn_eval_coeffs")
        if residual is not None:
            residual = _contig(_chk(residual, "bn residual"))
        # packed_out: the only reader is a convolution that takes its input as a packed pre-split image -- write that, once,
        # scaled by a bound of max |y| derived from the conv epilogue's per-channel maxima (no fp32 y, no split pass)
        colmax = getattr(sums, "_glf_colmax", None) if sums is not None else None
        packed = (bool(packed_out) and fused_stats and residual is None and colmax is not None and _PREC[0] >= 2
                  and PACKED_ACTS and _OUT_VIEW[0] is None)
        y, ldy, shared = _take_out(x.shape, dev)
        am = shared if shared is not None else amax_slot(dev)
        packed = packed and am is not None
        # relu + residual: the backward needs the sign of the forward output -- kept as one byte per four channels instead of y
        # (grad mode is always OFF inside Function.forward: before round 4 this read torch.is_grad_enabled() and the sign bytes were
        # never produced -- the backward fell back to re-reading y; what says that a backward may follow is needs_input_grad)
        need_mask = relu and residual is not None and any(ctx.needs_input_grad[:4]) and RELU_MASK_BYTES
        mask = torch.empty(rows * (c // 4), dtype=torch.uint8, device=dev) if need_mask else None
        if fused_stats:
            check(lib.glf_bn_apply_from_sums(_p(x), c, _p(residual), c, _p(y), ldy, _p(sums), rows, c, eps, momentum, _p(gamma), _p(beta),
                                             _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(nbt), int(relu), _p(am),
                                             _p(mask), _p(colmax) if packed else None, _stream()), "bn_apply_from_sums")
        else:
            check(lib.glf_bn_apply(_p(x), c, _p(residual), c, _p(y), ldy, _p(mean), _p(invstd), _p(gamma), _p(beta), rows, c,
                                   int(relu), _p(am), _p(mask), _stream()), "bn_apply")
        set_amax(y, am)
        if packed:
            y._glf_packed_only = True
        # without a residual the ReLU mask is recomputed from x in backward (sign of the same expression): y is not kept
        ctx.save_for_backward(x, mask if need_mask else (y if (relu and residual is not None) else None), mean, invstd, gamma, beta if relu else None)
        ctx.has_mask = need_mask
        ctx.cfg = (rows, c, relu, training, residual is not None, ldy)
        ctx.packed_grad = bool(packed_grad) and _PREC[0] >= 2
        ctx.param_refs = (gamma, beta)
        ctx.join = getattr(residual, "_glf_join", None) if residual is not None else None
        _last_bn[0] = (mean, invstd, rows)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, mean, invstd, gamma, beta = ctx.saved_tensors
        rows, c, relu, training, has_res, ldy = ctx.cfg
        dy2 = getattr(dy, "_glf_addend", None)    # fan_out(lazy=True): the two gradients of a block input arrive unsummed
        lddy2 = 0
        dy, lddy = _rows_view(dy)                 # may be a column slice of the concat-free projection's gradient
        if dy2 is not None:
            if dy2.shape != dy.shape:
                raise RuntimeError("glfusion_amd: the two addends of a lazy fan-in gradient differ in shape")
            dy2, lddy2 = _rows_view(dy2)
        dev = dy.device
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and ctx.needs_input_grad[3]) else None
        dgamma = grad_out(ctx.param_refs[0], (c,), dev)
        dbeta = grad_out(ctx.param_refs[1], (c,), dev)
        am = amax_slot(dev)
        # packed: the producing conv reads this gradient only through its dgrad / wgrad contractions -- write it ONCE, as the
        # packed pre-split image they want (scaled by a bound of its maximum the reduction pass provides), instead of fp32
        # followed by a split pass
        packed = ctx.packed_grad and _PREC[0] >= 2 and am is not None and PACKED_GRADS
        mask = y if ctx.has_mask else None
        fused = bnbwd_slot(c, dev) if (FUSED_BN_BWD and c <= 4096) else None
        check(lib.glf_bn_bwd(_p(dy), lddy, _p(x), c, None if ctx.has_mask else _p(y), ldy, _p(mean), _p(invstd), _p(gamma), _p(beta),
                             _p(dx), c, _p(dres), c, _p(dgamma), _p(dbeta), rows, c, int(relu), int(training),
                             None if fused is not None else _p(_ws(rows, c, dev)), _p(am), int(packed), _p(mask), _p(dy2), lddy2, _p(fused), _stream()), "bn_bwd")
        set_amax(dx, am)
        if packed:
            dx._glf_packed_only = True
        if ctx.join is not None and dres is not None:
            ctx.join.parked, dres = dres, None       # handed to the block's first conv, whose dgrad accumulates onto it
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None


# When a list, every train-mode batch_norm_act call appends (module, batch mean, batch invstd, rows): enough to
# replay the running-statistics update of a layer whose output is being reused instead of recomputed.
BN_TAP = None


def replay_bn_updates(records) -> None:
    """Apply again the running-stat updates recorded by BN_TAP (same batch statistics): what a second forward of the
    same modules over the same input would have done to running_mean / running_var / num_batches_tracked."""
    for bn, mean, invstd, rows in records:
        if not bn.track_running_stats or bn.running_mean is None:
            continue
        check(lib.glf_bn_replay_running(_p(mean), _p(invstd), int(rows), mean.numel(), float(bn.eps), float(bn.momentum),
                                        _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), _stream()), "bn_replay_running")


# BatchNorm backward hands the gradient of a conv output to that conv as a packed pre-split image (no fp32 copy, no split pass)
PACKED_GRADS = os.environ.get("GLF_PACKED_GRADS", "1") != "0"


def takes_packed_grad(weight: torch.Tensor) -> bool:
    """True when Conv2dFn / ConvCatFn backward can consume the gradient of a conv's output as a packed-only image: both its
    dgrad (NT, K = Cout) and its weight gradient (TN, M = Cout, N = Cin) run on the aligned split-fp16 kernels."""
    if not PACKED_GRADS or _PREC[0] < 2 or not PRESPLIT:
        return False
    cout, cin = weight.shape[0], weight.shape[1]
    return cout % 32 == 0 and nt_presplit_ok(cout, cout, cout) and tn_presplit_ok(cout, cin, cout, cin)


# BatchNorm forward hands an activation whose only reader is a convolution to it as a packed pre-split image
PACKED_ACTS = os.environ.get("GLF_PACKED_ACTS", "1") != "0"


def takes_packed_input(weight: torch.Tensor) -> bool:
    """True when Conv2dFn can consume its INPUT as a packed-only image: its forward (NT, K = Cin) and its weight gradient
    (TN, N = Cin) both run on the aligned split-fp16 kernels."""
    if not PACKED_ACTS or _PREC[0] < 2 or not PRESPLIT:
        return False
    cout, cin = weight.shape[0], weight.shape[1]
    return cin % 32 == 0 and nt_presplit_ok(cin, cin, cin) and tn_presplit_ok(cout, cin, cout, cin)


def packed_only(t: torch.Tensor) -> bool:
    return bool(getattr(t, "_glf_packed_only", False))


def batch_norm_act(x, bn: torch.nn.modules.batchnorm._BatchNorm, relu: bool, residual=None, sums=None, packed_grad: bool = False,
                   packed_out: bool = False):
    """nn.BatchNorm{2,3}d semantics (train: batch stats + running update; eval: running stats),
    optionally fused with a residual add and ReLU."""
    training = bn.training or bn.running_mean is None
    if training and x.numel() // x.shape[-1] <= 1:
        # same refusal (and wording) as torch.nn.functional.batch_norm in train mode
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {list(from_nhwc(x).shape) if x.dim() == 4 else list(x.shape)}")
    momentum = 0.0 if bn.momentum is None else float(bn.momentum)
    if bn.momentum is None and training and bn.track_running_stats:
        raise RuntimeError("glfusion_amd: cumulative-average BatchNorm (momentum=None) is not built")
    track = training and bn.track_running_stats
    if _is16(x):
        from . import ops16
        y = ops16.BatchNormAct16Fn.apply(x, bn.weight, bn.bias, residual,
                                         bn.running_mean if (track or not training) else None,
                                         bn.running_var if (track or not training) else None,
                                         bn.num_batches_tracked if track else None,
                                         training, momentum, float(bn.eps), relu, sums if training else None)
        if BN_TAP is not None and track:
            BN_TAP.append((bn,) + _last_bn[0])
        return y
    y = BatchNormActFn.apply(x, bn.weight, bn.bias, residual,
                             bn.running_mean if (track or not training) else None,
                             bn.running_var if (track or not training) else None,
                             bn.num_batches_tracked if track else None,
                             training, momentum, float(bn.eps), relu, sums if training else None, packed_grad, packed_out)
    if BN_TAP is not None and track:
        BN_TAP.append((bn,) + _last_bn[0])
    return y


# ----------------------------------------------------------------------------------------
# ReLU / dropout / pooling
# ----------------------------------------------------------------------------------------
class ReluFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _contig(_chk(x, "relu input"))
        y = torch.empty_like(x)
        check(lib.glf_relu_fwd(_p(x), _p(y), x.numel(), _stream()), "relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _contig(dy)
        dx = torch.empty_like(dy)
        check(lib.glf_relu_bwd(_p(dy), _p(y), _p(dx), dy.numel(), _stream()), "relu_bwd")
        return dx


def relu(x):
    if _is16(x):
        from . import ops16
        return ops16.Relu16Fn.apply(x)
    return ReluFn.apply(x)


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p: float, seed: int):
        x = _contig(_chk(x, "dropout input"))
        y = torch.empty_like(x)
        check(lib.glf_dropout(_p(x), _p(y), x.numel(), p, seed, _p(step_counter(x.device)), _stream()), "dropout")
        amax_bound(y, [x], scale=1.0 / (1.0 - p))          # kept elements are scaled by 1 / (1 - p)
        ctx.cfg = (p, seed)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        p, seed = ctx.cfg
        dy = _contig(dy)
        dx = torch.empty_like(dy)
        check(lib.glf_dropout(_p(dy), _p(dx), dy.numel(), p, seed, _p(step_counter(dy.device)), _stream()), "dropout_bwd")
        return dx, None, None


def dropout(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        raise RuntimeError("dropout p must be < 1")
    seed = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62).item())   # host RNG: no device sync
    if _is16(x):
        from . import ops16
        return ops16.Dropout16Fn.apply(x, float(p), seed)
    return DropoutFn.apply(x, float(p), seed)


class MaxPool3x3s2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _contig(_chk(x, "maxpool input"))
        n, h, w, c = x.shape
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        y = torch.empty(n, ho, wo, c, dtype=torch.float32, device=x.device)
        idx = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=x.device)
        check(lib.glf_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), n, h, w, c, _stream()), "maxpool_fwd")
        amax_bound(y, [x])                                  # a window maximum never exceeds the input's largest magnitude
        ctx.save_for_backward(idx)
        ctx.cfg = (n, h, w, c)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, h, w, c = ctx.cfg
        dy = _contig(dy)
        dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dy.device)
        check(lib.glf_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), n, h, w, c, _stream()), "maxpool_bwd")
        return dx


def maxpool3x3s2(x):
    if _is16(x):
        from . import ops16
        return ops16.MaxPool16Fn.apply(x)
    return MaxPool3x3s2Fn.apply(x)


class AvgPoolFn(Function):
    """AdaptiveAvgPool2d(1): [N,H,W,C] -> [N,1,1,C]."""

    @staticmethod
    def forward(ctx, x):
        x = _contig(_chk(x, "avgpool input"))
        n, h, w, c = x.shape
        y = torch.empty(n, 1, 1, c, dtype=torch.float32, device=x.device)
        check(lib.glf_avgpool_fwd(_p(x), _p(y), n, h * w, c, _stream()), "avgpool_fwd")
        ctx.cfg = (n, h, w, c)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, h, w, c = ctx.cfg
        dy = _contig(dy)
        dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dy.device)
        check(lib.glf_bcast_rows_scaled(_p(dy), _p(dx), c, 1.0 / (h * w), n, h * w, c, _stream()), "avgpool_bwd")
        return dx


def global_avgpool(x):
    if _is16(x):
        from . import ops16
        return ops16.AvgPool16Fn.apply(x)
    return AvgPoolFn.apply(x)


class BroadcastFn(Function):
    """bilinear up-sampling from a 1x1 map == broadcast: [N,1,1,C] -> [N,H,W,C]."""

    @staticmethod
    def forward(ctx, x, h: int, w: int):
        x = _contig(_chk(x, "broadcast input"))
        n, c = x.shape[0], x.shape[-1]
        y, ldy, shared = _take_out((n, h, w, c), x.device)
        check(lib.glf_bcast_rows_fwd(_p(x), _p(y), ldy, n, h * w, c, _stream()), "bcast_rows")
        if shared is not None:                   # the broadcast repeats x: its maximum is the source's
            src = amax_of(x)
            if src is not None:
                torch.maximum(shared, src, out=shared)
            set_amax(y, shared)
        ctx.cfg = (n, h, w, c)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, h, w, c = ctx.cfg
        dy, lddy = _rows_view(dy)
        dx = torch.empty(n, 1, 1, c, dtype=torch.float32, device=dy.device)
        check(lib.glf_sum_rows_fwd(_p(dy), lddy, _p(dx), 1.0, n, h * w, c, _stream()), "bcast_rows_bwd")
        return dx, None, None


def broadcast_hw(x, h: int, w: int):
    if _is16(x) or _S16[0]:             # 16-bit storage: the pooled branch arrives in fp32 and is broadcast into the bf16 buffer
        from . import ops16
        return ops16.Broadcast16Fn.apply(x, h, w)
    return BroadcastFn.apply(x, h, w)


class FanOutFn(Function):
    """Identity with k outputs; backward sums the k incoming gradients in ONE pass (k reads + 1 write) instead of
    autograd's pairwise accumulation (3 (k-1) tensor passes through torch's add kernel)."""

    @staticmethod
    def forward(ctx, x, k: int, lazy: bool = False):
        ctx.k = k
        # lazy only when x really is a batch_norm_act output: its backward is the one node that understands `_glf_addend`
        ctx.lazy = bool(lazy) and k == 2 and LAZY_FAN_IN and type(x.grad_fn).__name__ in ("BatchNormActFnBackward", "BatchNormAct16FnBackward")
        outs = tuple(x.view_as(x) for _ in range(k))
        am = amax_of(x)                       # one measurement (or the producer's by-product) serves every alias
        share = [None]                        # ... and so does one pre-split image, whichever consumer makes it first
        for t in outs:
            set_amax(t, am)
            t._glf_pack_share = share
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        live = [_contig(d) for d in dys if d is not None]
        if not live:
            return None, None, None
        if len(live) == 1:
            return live[0], None, None
        if ctx.lazy and live[0].shape == live[1].shape and not packed_only(live[0]) and not packed_only(live[1]):
            # the producer of x is a BatchNorm(+residual) whose backward adds the two while reading them (glf_bn_bwd dy2): the
            # caller of fan_out(lazy=True) guarantees that node is the ONLY reader of this gradient
            a = live[0]
            a._glf_addend = live[1]
            return a, None, None
        if _is16(live[0]):
            from . import ops16
            return ops16.add_n16(live), None, None
        out = torch.empty_like(live[0])
        n = out.numel()
        if n % 4 != 0 or any(d.shape != out.shape for d in live) or len(live) > 8:
            raise RuntimeError("fan_out: gradients must share one shape with numel % 4 == 0 (<= 8 branches)")
        arr = (C.c_void_p * len(live))(*[d.data_ptr() for d in live])
        check(lib.glf_add_n(arr, len(live), _p(out), n, _stream()), "add_n")
        return out, None, None


# The two gradients of a residual block's input go to the previous block's last BatchNorm unsummed (glf_bn_bwd adds them while reading)
LAZY_FAN_IN = os.environ.get("GLF_LAZY_FAN_IN", "1") != "0"


def fan_out(x: torch.Tensor, k: int, lazy: bool = False):
    """k aliases of x for k consumers (use each exactly once).  lazy (k == 2): the caller guarantees that x is the output of a
    batch_norm_act call and that this fan_out is its ONLY reader -- the backward then hands the two incoming gradients to that
    BatchNorm's backward as a pair (first + `_glf_addend`) instead of summing them in a pass of its own."""
    if k <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return tuple(x for _ in range(max(k, 1)))
    return FanOutFn.apply(x, k, lazy)


class GradJoin:
    """Meeting point of the two gradients of a residual block's input (identity shortcut + first conv).  The shortcut's
    gradient -- produced by the BatchNorm(+residual) backward, early in the block's backward -- is parked here instead of
    being returned; the first conv's dgrad, the last kernel of the block's backward, ACCUMULATES into it in its epilogue
    (C += result) and returns the total.  The separate two-input add (two reads + one write of the block input per block)
    is gone.  Attached to the two aliases by `join_gradients`."""
    __slots__ = ("parked",)

    def __init__(self):
        self.parked = None


# Measured (C2 step, A/B in one run): 298.9 ms with the join against 291.3 ms without -- the accumulating epilogue sits on
# the backward's critical chain (the block's last dgrad) while the add it replaces overlaps with other streams.  Off.
GRAD_JOIN = os.environ.get("GLF_GRAD_JOIN", "0") != "0"


def join_gradients(conv_input: torch.Tensor, shortcut: torch.Tensor) -> None:
    """Mark two fan_out aliases of one tensor: `shortcut` will be used as the residual of a BatchNorm, `conv_input` as the
    input of a 1x1 stride-1 convolution whose dgrad then absorbs the shortcut's gradient (see GradJoin)."""
    if GRAD_JOIN and torch.is_grad_enabled() and conv_input.requires_grad:
        j = GradJoin()
        conv_input._glf_join = j
        shortcut._glf_join = j


# ----------------------------------------------------------------------------------------
# independent sections on side streams
# ----------------------------------------------------------------------------------------
# The per-view encoders / heads and the two fusion blocks are independent chains.  Each kernel of a chain either
# fills the chip with MFMA work in whole "rounds" of 256 workgroups (leaving the last round partly idle) or is a
# short HBM-bound pass; running the chains on separate HIP streams lets the hardware fill one chain's idle CUs
# with another chain's workgroups.  Backward nodes run on the stream their forward ran on (autograd does that), so
# the overlap carries over.  GLF_STREAMS=0 serialises everything on the current stream.
STREAMS = os.environ.get("GLF_STREAMS", "1") != "0"
N_SIDE_STREAMS = int(os.environ.get("GLF_SIDE_STREAMS", "24"))
if STREAMS and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    # parameters' AccumulateGrad nodes live on the stream that first touched them (the default one when a gradient
    # hook keeps them alive); gradients produced on a side stream are synchronised into it, which is intended
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
_side = {}


def _walk_tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _walk_tensors(v)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            yield from _walk_tensors(v)


# GLF_SECTIONS_DISTINCT=1: the stream counter restarts only at begin_step() (the models call it at the top of forward), so
# successive top-level parallel_sections calls of one forward get DIFFERENT side streams instead of re-using 0, 1, 2 ...
SECTIONS_DISTINCT = os.environ.get("GLF_SECTIONS_DISTINCT", "0") != "0"


_step_counters = {}


def step_counter(dev) -> torch.Tensor:
    """Device uint64 (held as int64) that advance_step() moves once per captured training step; the dropout kernels mix it into their
    seed, so a step replayed from a hipGraph (whose launches carry fixed seed arguments) still draws fresh masks."""
    t = _step_counters.get(dev)
    if t is None:
        t = _step_counters[dev] = zeros(1, dtype=torch.int64, device=dev)
    return t


def begin_step(dev=None) -> None:
    """Top of a model forward: restart the side-stream assignment and decide ONCE whether this step retains pre-split images
    from forward to backward (see retain_ok)."""
    for pool in _side.values():
        if pool[2] == 0:
            pool[1] = 0
    _retain_step.clear()


def advance_step(dev) -> None:
    """Advance the device step counter (one tiny launch).  A captured training step calls it ONCE, first thing, whatever the
    number of forwards it contains -- forward and backward of a Dropout must read the same value.  Eager steps need not call
    it: their dropout seeds are drawn on the host per call."""
    check(lib.glf_counter_add(_p(step_counter(dev)), 1, _stream()), "counter_add")


def reset_capture_pools() -> None:
    """Before a hipGraph capture: drop the pre-zeroed slot pools so that the pools the captured step uses are created -- and
    zero-filled -- INSIDE the capture (a replay then starts from zeroed maxima / statistics like an eager step does)."""
    _amax_pool.clear()
    _stats_pool.clear()
    _colmax_pool.clear()


def use_here(*tensors) -> None:
    """Tell the caching allocator that these tensors (allocated on another stream) are read on the CURRENT stream: their
    memory must not be handed out again before this stream's work on them is done."""
    if not STREAMS:
        return
    cur = torch.cuda.current_stream()
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(cur)


def parallel_sections(fns):
    """Run the callables as independent sections, section i on side stream i, and join them on the current
    stream.  Returns their results in order."""
    if not STREAMS or len(fns) <= 1 or not torch.cuda.is_available():
        return [f() for f in fns]
    cur = torch.cuda.current_stream()
    dev = cur.device
    pool = _side.setdefault(dev, [[], 0, 0])          # streams, next index, nesting depth
    if not pool[0]:
        pool[0] = [torch.cuda.Stream(device=dev) for _ in range(N_SIDE_STREAMS)]
    # Streams are handed out in call order, restarting with every top-level call: nested calls get streams of their
    # own and a given section lands on the SAME stream every step (the caching allocator keeps one pool per stream;
    # a wandering assignment would re-allocate every activation).  A stream shared by two sections only adds an
    # ordering between them.
    if pool[2] == 0 and not SECTIONS_DISTINCT:
        pool[1] = 0
    pool[2] += 1
    used = []
    for _ in fns:
        st = pool[0][pool[1] % N_SIDE_STREAMS]
        pool[1] += 1
        if st.cuda_stream == cur.cuda_stream:
            st = pool[0][pool[1] % N_SIDE_STREAMS]
            pool[1] += 1
        used.append(st)
    outs = []
    try:
        for s, f in zip(used, fns):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(f())
    finally:
        pool[2] -= 1
    for s, out in zip(used, outs):
        cur.wait_stream(s)
        for t in _walk_tensors(out):
            if t.is_cuda:
                t.record_stream(cur)       # allocated on s, consumed on cur: keep the allocator from reusing it early
    return outs


# ----------------------------------------------------------------------------------------
# local gate (ours.py:1802-1816)
# ----------------------------------------------------------------------------------------
def _gate_forward(ctx, cls, ctr, f, weight: float):
    cls, ctr, f = _contig(_chk(cls, "cls")), _contig(_chk(ctr, "ctr")), _contig(_chk(f, "f4"))
    c = f.shape[-1]
    rows = f.numel() // c
    ncls = cls.shape[-1]
    y = torch.empty_like(f)
    a = torch.empty(rows, dtype=torch.float32, device=f.device)
    am = torch.empty(rows, dtype=torch.int32, device=f.device)
    check(lib.glf_gate_fwd(_p(cls), ncls, _p(ctr), _p(f), _p(y), _p(a), _p(am), weight, rows, c, _stream()), "gate_fwd")
    amax_bound(y, [f])                                      # y = f * a with a in (0, 1)
    ctx.save_for_backward(cls, ctr, f, a, am)
    ctx.cfg = (rows, c, ncls, weight)
    return y, a


class GateFn(Function):
    @staticmethod
    def forward(ctx, cls, ctr, f, weight: float):
        return _gate_forward(ctx, cls, ctr, f, weight)[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        cls, ctr, f, a, am = ctx.saved_tensors
        rows, c, ncls, weight = ctx.cfg
        dy = _contig(dy)
        df = torch.empty_like(f)
        dcls = torch.empty_like(cls)
        dctr = torch.empty_like(ctr)
        check(lib.glf_gate_bwd(_p(dy), _p(f), _p(cls), ncls, _p(ctr), _p(a), _p(am), weight, _p(df), _p(dcls), _p(dctr),
                               rows, c, _stream()), "gate_bwd")
        return dcls, dctr, df, None


def local_gate(cls_logits, ctr_logits, f4, weight: float):
    if _is16(f4):
        from . import ops16
        return ops16.Gate16Fn.apply(cls_logits, ctr_logits, f4, float(weight))[0]
    return GateFn.apply(cls_logits, ctr_logits, f4, float(weight))


class AxpbyFn(Function):
    """out = a*x + b*y on tensors of one shape."""

    @staticmethod
    def forward(ctx, x, y, a: float, b: float):
        x, y = _contig(_chk(x, "x")), _contig(_chk(y, "y"))
        if x.shape != y.shape:
            raise RuntimeError("axpby: shapes differ")
        out = torch.empty_like(x)
        check(lib.glf_axpby(_p(x), _p(y), _p(out), a, b, x.numel(), _stream()), "axpby")
        ctx.ab = (a, b)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        a, b = ctx.ab
        d = _contig(d)
        zero = d                                   # a*d + 0*d
        dx = dy = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(d)
            check(lib.glf_axpby(_p(d), _p(zero), _p(dx), a, 0.0, d.numel(), _stream()), "axpby_bwd")
        if ctx.needs_input_grad[1]:
            dy = torch.empty_like(d)
            check(lib.glf_axpby(_p(d), _p(zero), _p(dy), b, 0.0, d.numel(), _stream()), "axpby_bwd")
        return dx, dy, None, None


def axpby(x, y, a: float, b: float):
    if _is16(x):
        from . import ops16
        return ops16.Axpby16Fn.apply(x, y, float(a), float(b))
    return AxpbyFn.apply(x, y, float(a), float(b))


class GateMapFn(GateFn):
    """GateFn that also hands out the gate map a = sigmoid(w * max_c sigmoid(cls) * sigmoid(ctr)) as [N,1,h,w]
    (Local_only returns it as `atten_map`, ours.py:2221-2222, 2249); the map is a non-differentiable output."""

    @staticmethod
    def forward(ctx, cls, ctr, f, weight: float):
        y, a = _gate_forward(ctx, cls, ctr, f, weight)
        n, h, w = f.shape[0], f.shape[1], f.shape[2]
        amap = a.view(n, 1, h, w).clone()
        ctx.mark_non_differentiable(amap)
        return y, amap

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _damap):
        return GateFn.backward(ctx, dy)


def local_gate_with_map(cls_logits, ctr_logits, f4, weight: float):
    if _is16(f4):
        from . import ops16
        y, a = ops16.Gate16Fn.apply(cls_logits, ctr_logits, f4, float(weight))
        n, h, w = f4.shape[0], f4.shape[1], f4.shape[2]
        return y, a.view(n, 1, h, w).clone()
    return GateMapFn.apply(cls_logits, ctr_logits, f4, float(weight))


# ----------------------------------------------------------------------------------------
# view stacking / slicing for the fusion block
# ----------------------------------------------------------------------------------------
class StackViewsFn(Function):
    """[N,h,w,C] x V -> [N,V,h,w,C] (ours.py:1819-1820: unsqueeze(2) + cat(dim=2) in NCDHW terms)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_contig(_chk(t, "view feature")) for t in xs]
        n, h, w, c = xs[0].shape
        v = len(xs)
        out = torch.empty(n, v, h, w, c, dtype=torch.float32, device=xs[0].device)
        inner = h * w * c
        amax_bound(out, xs)                  # known BEFORE the copy: the largest of the views' maxima
        am = getattr(out, "_glf_amax", None)
        am = am[2] if am is not None else None
        if am is not None and presplit_ok(out, am) and retain_ok(out.device):
            # the stack is read by the fusion block's contractions through its packed image only: write it in the same pass
            pk = torch.empty_like(out)
            for i, t in enumerate(xs):
                check(lib.glf_copy_frames_split(_p(t), inner, _p(out[:, i]), _p(pk[:, i]), v * inner, n, inner, _p(am), _stream()), "stack_views")
            out._glf_packed = (out._version, out.data_ptr(), am, pk, torch.cuda.current_stream() if STREAMS else None)
        else:
            for i, t in enumerate(xs):
                check(lib.glf_copy_frames(_p(t), inner, _p(out[:, i]), v * inner, n, inner, _stream()), "stack_views")
        ctx.cfg = (n, v, h, w, c)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, v, h, w, c = ctx.cfg
        dy = _contig(dy)
        inner = h * w * c
        outs = []
        for i in range(v):
            g = torch.empty(n, h, w, c, dtype=torch.float32, device=dy.device)
            check(lib.glf_copy_frames(_p(dy[:, i]), v * inner, _p(g), inner, n, inner, _stream()), "unstack_views")
            outs.append(g)
        return tuple(outs)


def stack_views(xs: Sequence[torch.Tensor]):
    if _is16(xs[0]):
        from . import ops16
        return ops16.StackViews16Fn.apply(*xs)
    return StackViewsFn.apply(*xs)


class AddViewsFn(Function):
    """f4_fusion[v] = G[:, i] + L[:, i] for every view i of two [N,V,h,w,C] tensors
    (ours.py:1833-1834); returns V contiguous [N,h,w,C] tensors."""

    @staticmethod
    def forward(ctx, g, l):
        g, l = _contig(_chk(g, "global")), _contig(_chk(l, "local"))
        n, v, h, w, c = g.shape
        inner = h * w * c
        outs = []
        for i in range(v):
            out = torch.empty(n, h, w, c, dtype=torch.float32, device=g.device)
            check(lib.glf_add_frames(_p(g[:, i]), v * inner, _p(l[:, i]), v * inner, _p(out), inner, n, inner, _stream()), "add_views")
            amax_bound(out, [g, l], sum_=True)
            outs.append(out)
        ctx.cfg = (n, v, h, w, c)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        n, v, h, w, c = ctx.cfg
        inner = h * w * c
        dev = next(d.device for d in dys if d is not None)
        dg = (zeros if any(d is None for d in dys) else torch.empty)(n, v, h, w, c, dtype=torch.float32, device=dev)
        for i, d in enumerate(dys):
            if d is not None:
                check(lib.glf_copy_frames(_p(_contig(d)), inner, _p(dg[:, i]), v * inner, n, inner, _stream()), "add_views_bwd")
        return dg, dg


def add_views(g, l):
    if _is16(g):
        from . import ops16
        return ops16.AddViews16Fn.apply(g, l)
    return AddViewsFn.apply(g, l)


class SplitViewsFn(Function):
    """[N,V,h,w,C] -> V contiguous [N,h,w,C] tensors (`conv_feat[:, :, i]` + `.contiguous()` of ours.py:2097, 2104)."""

    @staticmethod
    def forward(ctx, g):
        g = _contig(_chk(g, "stacked views"))
        n, v, h, w, c = g.shape
        inner = h * w * c
        outs = []
        for i in range(v):
            out = torch.empty(n, h, w, c, dtype=torch.float32, device=g.device)
            check(lib.glf_copy_frames(_p(g[:, i]), v * inner, _p(out), inner, n, inner, _stream()), "split_views")
            outs.append(out)
        ctx.cfg = (n, v, h, w, c)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        n, v, h, w, c = ctx.cfg
        inner = h * w * c
        dev = next(d.device for d in dys if d is not None)
        dg = (zeros if any(d is None for d in dys) else torch.empty)(n, v, h, w, c, dtype=torch.float32, device=dev)
        for i, d in enumerate(dys):
            if d is not None:
                check(lib.glf_copy_frames(_p(_contig(d)), inner, _p(dg[:, i]), v * inner, n, inner, _stream()), "split_views_bwd")
        return dg


def split_views(g):
    if _is16(g):
        from . import ops16
        return ops16.SplitViews16Fn.apply(g)
    return SplitViewsFn.apply(g)


# ----------------------------------------------------------------------------------------
# bilinear up-sampling to NCHW logits, loss, metrics
# ----------------------------------------------------------------------------------------
class BilinearUpFn(Function):
    """F.interpolate(mode='bilinear', align_corners=False): [N,h,w,C] -> NCHW [N,C,H,W]."""

    @staticmethod
    def forward(ctx, x, ho: int, wo: int):
        x = _contig(_chk(x, "upsample input"))
        n, h, w, c = x.shape
        y = torch.empty(n, c, ho, wo, dtype=torch.float32, device=x.device)
        check(lib.glf_bilinear_up_fwd(_p(x), _p(y), n, h, w, c, ho, wo, _stream()), "bilinear_up_fwd")
        ctx.cfg = (n, h, w, c, ho, wo)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, h, w, c, ho, wo = ctx.cfg
        dy = _contig(_chk(dy, "upsample grad"))
        dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dy.device)
        check(lib.glf_bilinear_up_bwd(_p(dy), _p(dx), n, h, w, c, ho, wo, _stream()), "bilinear_up_bwd")
        return dx, None, None


def bilinear_up(x, ho: int, wo: int):
    if _is16(x):
        from . import ops16
        x = ops16.to_f32(x)
    return BilinearUpFn.apply(x, ho, wo)


class BceSumFn(Function):
    """nn.BCEWithLogitsLoss(reduction='sum') (main.py:87)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = _contig(_chk(logits, "logits")), _contig(_chk(target, "target"))
        if logits.shape != target.shape:
            raise RuntimeError("bce: logits/target shape mismatch")
        loss = torch.empty(1, dtype=torch.float64, device=logits.device)
        check(lib.glf_bce_logits_sum(_p(logits), _p(target), _p(loss), None, 1.0, None, logits.numel(), _stream()), "bce_logits_sum")
        ctx.save_for_backward(logits, target)
        return loss[0].float()

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        logits, target = ctx.saved_tensors
        dx = torch.empty_like(logits)
        scratch = torch.empty(1, dtype=torch.float64, device=logits.device)
        dl = _contig(dl.float())
        # dx = (sigmoid(x) - t) * upstream, the upstream scalar read on the device (no host sync)
        check(lib.glf_bce_logits_sum(_p(logits), _p(target), _p(scratch), _p(dx), 1.0, _p(dl), logits.numel(), _stream()), "bce_logits_bwd")
        return dx, None


def bce_with_logits_sum(logits, target):
    return BceSumFn.apply(logits, target)


def overlap_counts(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """int64 [tp, fp, fn, tn] of pred = sigmoid(logits) > 0.5 against target (main.py:800-815)."""
    logits, target = _contig(_chk(logits, "logits")), _contig(_chk(target, "target"))
    counts = torch.empty(4, dtype=torch.int64, device=logits.device)
    check(lib.glf_overlap_counts(_p(logits), _p(target), _p(counts), logits.numel(), _stream()), "overlap_counts")
    return counts


def overlap_metrics_from_counts(counts: torch.Tensor, eps: float = 1e-5):
    """(pixel_acc, dice, precision, specificity, recall) exactly as main.py:807-813."""
    tp, fp, fn, tn = [float(v) for v in counts.tolist()]
    return ((tp + tn) / (tp + tn + fp + fn + eps), (2 * tp) / (2 * tp + fp + fn + eps),
            tp / (tp + fp + eps), tn / (tn + fp + eps), tp / (tp + fn + eps))


# ----------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------
# temporal cycle-consistency loss (main.py:213-235, 650-798)
# ----------------------------------------------------------------------------------------
class SumHWFn(Function):
    """x [..., h, w, C] channels-last (any number of leading frame dims) -> sum over (h, w): [..., C].
    `f4_global_fusion[v].sum(dim=(2, 3))` of main.py:226."""

    @staticmethod
    def forward(ctx, x):
        x = _contig(_chk(x, "sum_hw input"))
        *lead, h, w, c = x.shape
        n = 1
        for d in lead:
            n *= d
        y = torch.empty(*lead, c, dtype=torch.float32, device=x.device)
        check(lib.glf_sum_rows_fwd(_p(x), c, _p(y), 1.0, n, h * w, c, _stream()), "sum_hw")
        ctx.cfg = (tuple(x.shape), n, h * w, c)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        shape, n, p, c = ctx.cfg
        dy = _contig(dy)
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        check(lib.glf_bcast_rows_scaled(_p(dy), _p(dx), c, 1.0, n, p, c, _stream()), "sum_hw_bwd")
        return dx


def sum_hw(x: torch.Tensor) -> torch.Tensor:
    if _is16(x):                          # the cycle loss works on fp32 features (main.py:226): one cast of the pooled block's input
        from . import ops16
        x = ops16.to_f32(_contig(x))
    return SumHWFn.apply(x)


def pooled_fusion_features(f4_fusion: dict) -> dict:
    """{view: [T, C]} = f4_fusion[view].sum(dim=(2, 3)) (main.py:226) for the dict Global_and_Local.forward returns.
    Its entries are per-view slices of ONE [T, V, h, w, C] block (the fusion module's output): the block is pooled
    once (one read, and one broadcast in backward) instead of V strided reductions through sliced views."""
    views = list(f4_fusion)
    base = getattr(f4_fusion[views[0]], "_glf_stack", None)
    if base is not None and all(getattr(f4_fusion[v], "_glf_stack", (None, -1))[0] is base[0] for v in views):
        pooled = sum_hw(base[0])                                   # [T, V, C]
        return {v: pooled[:, f4_fusion[v]._glf_stack[1]] for v in views}
    return {v: sum_hw(to_nhwc(f4_fusion[v])) for v in views}


class SegCycleFn(Function):
    @staticmethod
    def forward(ctx, feat, target_region: int, cyc_off: int, chunk_size: int, temperature: float, start0: int, n_starts: int,
                stride: int, weight: float, soft: bool):
        feat = _contig(_chk(feat, "cycle features"))
        if feat.dim() != 2:
            raise RuntimeError("seg_cycle: features must be [T, F]")
        t, f = feat.shape
        loss = torch.empty((), dtype=torch.float32, device=feat.device)
        dfeat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        check(lib.glf_seg_cycle(_p(feat), t, f, target_region, cyc_off, chunk_size, float(temperature), start0, n_starts, stride,
                                float(weight), int(soft), _p(loss), _p(dfeat), _stream()), "seg_cycle")
        ctx.save_for_backward(dfeat)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        (dfeat,) = ctx.saved_tensors
        out = torch.empty_like(dfeat)
        check(lib.glf_scale(_p(dfeat), _p(out), dfeat.numel(), 1.0, _p(_contig(dloss)), _stream()), "seg_cycle_bwd")
        return out, None, None, None, None, None, None, None, None, None


def seg_cycle(feat, target_region: int = 16, cyc_off: int = 2, chunk_size: int = 3, temperature: float = 10, start: int = None):
    """Trainer.seg_cycle (main.py:650-718).  `start` is the query start frame the reference draws with
    np.random.choice(target_region - (chunk_size + cyc_off) + 1) (main.py:655); None draws it the same way."""
    n = target_region - (chunk_size + cyc_off) + 1
    if start is None:
        import numpy as np
        start = int(np.random.choice(n))
    return SegCycleFn.apply(feat, target_region, cyc_off, chunk_size, temperature, int(start), 1, 1, 1.0, False)


def dense_seg_cycle(feat, target_region: int = 16, cyc_off: int = 2, chunk_size: int = 3, temperature: float = 10,
                    soft_label: bool = False, is_overlap: bool = True):
    """Trainer.dense_seg_cycle (main.py:720-798): every start frame, averaged over the number of possible starts."""
    n = target_region - (chunk_size + cyc_off) + 1
    stride = 1 if is_overlap else chunk_size
    return SegCycleFn.apply(feat, target_region, cyc_off, chunk_size, temperature, 0, (n + stride - 1) // stride, stride, 1.0 / n,
                            bool(soft_label))


# contraction precision: "f32" = exact fp32 MFMA, "bf16x6" = split-bf16 (6 MFMAs per product),
# "f16x3" = scaled split-fp16 (3 MFMAs per product).  Host-side setting handed to the library with every call.
# "bf16" = 16-bit STORAGE (BASELINE configs 3 / 5): bf16 activations / saved tensors / activation gradients in HBM, one bf16 MFMA
# per product (glfusion_amd.ops16); the few fp32 contractions left in that mode (5- / 1-channel head logits) use the exact kernels.
PRECISIONS = ("f32", "bf16x6", "f16x3", "f16", "bf16")
# ----------------------------------------------------------------------------------------
def set_precision(mode: str) -> None:
    i = PRECISIONS.index(mode)
    _S16[0] = mode == "bf16"
    _PREC[0] = 0 if _S16[0] else i


def get_precision() -> str:
    return "bf16" if _S16[0] else PRECISIONS[_PREC[0]]


class precision_scope:
    """`with ops.precision_scope("f32"): ...` -- forward AND backward of whatever runs inside must both happen inside
    the block (backward launches read the setting when they run)."""

    def __init__(self, mode: str):
        self.mode = mode

    def __enter__(self):
        self.prev = get_precision()
        set_precision(self.mode)
        return self

    def __exit__(self, *exc):
        set_precision(self.prev)
        return False
