"""CPU: the C-ABI library loads and exports every symbol include/glfusion.h declares; the
host package imports; the product path refuses CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

import glfusion_amd
from glfusion_amd import _lib


def test_header_parses_and_library_exports_every_symbol():
    protos = _lib.parse_header()
    text = open(_lib.HEADER).read()
    declared = set(re.findall(r"\b(glf_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    assert declared == set(protos), declared ^ set(protos)
    assert len(protos) >= 35
    assert os.path.exists(_lib.LIB_PATH), "build the engine first: make -C gl-fusion_amd/csrc"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"libglfusion_hip.so does not export {name}"
    dll.glf_abi_version.restype = ctypes.c_int
    assert dll.glf_abi_version() == 7
    # pure host-side queries work without a GPU
    dll.glf_bn_workspace.restype = ctypes.c_size_t
    dll.glf_bn_workspace.argtypes = [ctypes.c_int, ctypes.c_int]
    assert dll.glf_bn_workspace(50176, 2048) == 3 * 1024 * 2048 + 2 * 2048


def test_diagnostic_library_is_separate_and_exports_its_header():
    """include/glfusion_diag.h -> lib/libglfusion_diag.so: the measurement aids of bench.py live outside the product library."""
    root = os.path.dirname(os.path.dirname(_lib.HEADER))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "glfusion_diag.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(glf_[a-z0-9_]+)\s*\(", text))
    assert declared == {"glf_probe_mfma_f16"}
    diag = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libglfusion_diag.so"))
    product = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(diag, name) and not hasattr(product, name), name
    # the workspace queries of the single-call fusion block are host-only
    tp = _lib.TpaviParams(64, 2352, 2048, 1024, 1, 1e-5, 0.1, 1e-5)
    product.glf_s16_tpavi_workspace_bytes.restype = ctypes.c_size_t
    product.glf_s16_tpavi_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fwd, bwd = (product.glf_s16_tpavi_workspace_bytes(ctypes.byref(tp), k) for k in (0, 1))
    rows = 64 * 2352
    assert fwd == 2 * 2048 * 8 and bwd > rows * (2048 + 1024 + 3072) * 2
    bad = _lib.TpaviParams(0, 2352, 2048, 1024, 1, 1e-5, 0.1, 1e-5)
    assert product.glf_s16_tpavi_workspace_bytes(ctypes.byref(bad), 1) == 0


def test_gemm_params_struct_matches_header_layout():
    # field order / count of the ctypes mirror against the header text
    text = open(_lib.HEADER).read()
    body = re.search(r"typedef struct \{(.*?)\} glf_gemm_params;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("const float*", "ptr").split(None, 1)[1]
        names += [n.strip() for n in parts.split(",")]
    assert names == [f[0] for f in _lib.GemmParams._fields_]
    dll = ctypes.CDLL(_lib.LIB_PATH)
    dll.glf_sizeof_gemm_params.restype = ctypes.c_size_t
    assert ctypes.sizeof(_lib.GemmParams) == dll.glf_sizeof_gemm_params()
    dll.glf_sizeof_tpavi_params.restype = ctypes.c_size_t
    assert ctypes.sizeof(_lib.TpaviParams) == dll.glf_sizeof_tpavi_params()


def test_no_cpu_fallback():
    from glfusion_amd import ops
    from glfusion_amd.models import Global_and_Local
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 1, 1))
    m = Global_and_Local(["1"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"1": torch.zeros(1, 1, 112, 112)})


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gl-fusion_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_tap_masks():
    from glfusion_amd.ops import tap_mask
    full = (1 << 9) - 1
    assert tap_mask(1, 28, 28, 28, 28, 3, 3, 1, 1, 1) == full
    assert tap_mask(1, 28, 28, 28, 28, 3, 3, 1, 12, 12) == full
    assert tap_mask(1, 28, 28, 28, 28, 3, 3, 1, 24, 24) == full
    assert tap_mask(1, 28, 28, 28, 28, 3, 3, 1, 36, 36) == 1 << 4          # rate 36 on 28x28 == centre tap only
    assert tap_mask(2, 28, 28, 28, 28, 3, 3, 1, 36, 36) == 1 << 4
    assert tap_mask(1, 28, 28, 55, 55, 3, 3, 2, 1, 1) == full
    assert tap_mask(2, 55, 55, 28, 28, 3, 3, 2, 1, 1) == full


def test_adam_pointer_table_chunks():
    """Host logic of the fused optimizer: parameters are cut into rows of at most CHUNK elements, pointers advanced."""
    import numpy as np
    from glfusion_amd import optim
    rows = optim._chunk_rows([(1000, 2000, 3000, 4000, optim.CHUNK + 7), (16, 32, 48, 64, 5)])
    assert rows.tolist() == [[1000, 2000, 3000, 4000, optim.CHUNK],
                             [1000 + 4 * optim.CHUNK, 2000 + 4 * optim.CHUNK, 3000 + 4 * optim.CHUNK, 4000 + 4 * optim.CHUNK, 7],
                             [16, 32, 48, 64, 5]]


def test_conv_plan_matches_host_policy():
    """glf_conv2d_plan (the launch policy embedded in the C convolution entry points) against the policy the Python autograd
    wrappers use (ops.tap_mask / rect_fraction / region_mode / _tn_split): same kept taps, same mode, same reduction slices, for
    every conv geometry of the model (and a few ragged ones) under every precision.  No GPU needed: the plan launches nothing."""
    import ctypes as C
    from glfusion_amd import ops
    from glfusion_amd._lib import ConvParams, ConvPlan, lib
    convs = [  # n, h, w, cin, cout, k, stride, pad, dil
        (64, 55, 55, 64, 64, 3, 1, 1, 1), (64, 55, 55, 128, 128, 3, 2, 1, 1), (64, 55, 55, 256, 512, 1, 2, 0, 1),
        (64, 28, 28, 256, 256, 3, 1, 2, 2), (64, 28, 28, 512, 512, 3, 1, 4, 4), (64, 28, 28, 2048, 256, 3, 1, 12, 12),
        (64, 28, 28, 2048, 256, 3, 1, 24, 24), (64, 28, 28, 2048, 256, 3, 1, 36, 36), (64, 28, 28, 2048, 256, 1, 1, 0, 1),
        (64, 28, 28, 256, 5, 1, 1, 0, 1), (64, 1, 1, 2048, 256, 1, 1, 0, 1), (32, 56, 56, 2048, 256, 3, 1, 36, 36),
        (3, 15, 13, 32, 40, 3, 1, 1, 1), (2, 28, 28, 64, 288, 3, 1, 12, 12), (192, 28, 28, 1024, 2048, 1, 1, 0, 1)]
    for prec in ("f32", "bf16x6", "f16x3", "f16"):
        ops.set_precision(prec)
        try:
            for (n, h, w, cin, cout, k, stride, pad, dil) in convs:
                p = ConvParams()
                p.n, p.h, p.w, p.cin, p.cout, p.kh, p.kw, p.stride, p.pad, p.dil = n, h, w, cin, cout, k, k, stride, pad, dil
                p.precision = ops.PRECISIONS.index(prec) + 1
                ho, wo = ops._conv_out(h, k, stride, pad, dil), ops._conv_out(w, k, stride, pad, dil)
                taps, plain = k * k, (k == 1 and stride == 1 and pad == 0)
                pl = ConvPlan()
                # forward
                assert lib.glf_conv2d_plan(C.byref(p), 0, C.byref(pl)) == 0
                mask = 1 if plain else ops.tap_mask(1, ho, wo, h, w, k, k, stride, pad, dil)
                rect = (not plain and taps > 1 and stride == 1 and bin(mask).count("1") > 1
                        and ops.rect_fraction(1, ho, wo, h, w, k, k, pad, dil, mask) < ops._rect_thr("fwd"))
                assert (pl.ho, pl.wo, pl.tap_mask, pl.rect, pl.plain) == (ho, wo, mask, int(rect), int(plain)), (prec, "fwd", n, h, cin, cout, k, dil)
                from torch import empty
                wshape = empty(cout, cin, k, k, device="meta")
                assert bool(pl.colstats_ok) == ops.conv_stats_fusable(wshape, stride, pad, dil, h, w), (prec, "colstats", cin, cout, k, dil)
                # dgrad
                assert lib.glf_conv2d_plan(C.byref(p), 1, C.byref(pl)) == 0
                mask = 1 if plain else ops.tap_mask(2, h, w, ho, wo, k, k, stride, pad, dil)
                frac = 1.0 if plain or stride != 1 else ops.rect_fraction(2, h, w, ho, wo, k, k, pad, dil, mask)
                if mask and not plain and bin(mask).count("1") > 1 and ops.region_mode(taps, k, stride, pad, dil, ho, wo, h, w, cout, frac):
                    want = 2
                else:
                    want = int(not plain and taps > 1 and stride == 1 and bin(mask).count("1") > 1 and frac < ops._rect_thr("dgrad"))
                assert (pl.tap_mask, pl.rect) == (mask, want), (prec, "dgrad", n, h, cin, cout, k, dil)
                # wgrad
                assert lib.glf_conv2d_plan(C.byref(p), 2, C.byref(pl)) == 0
                mask = 1 if plain else ops.tap_mask(1, ho, wo, h, w, k, k, stride, pad, dil)
                ntap = bin(mask).count("1")
                rect = (not plain and taps > 1 and stride == 1 and ntap > 1
                        and ops.rect_fraction(1, ho, wo, h, w, k, k, pad, dil, mask) < ops._rect_thr("wgrad"))
                frac = ops.rect_fraction(1, ho, wo, h, w, k, k, pad, dil, mask) if rect else 1.0
                split = ops.wgrad_split(n * ho * wo, frac, cout, cin, ntap, rect)
                assert (pl.tap_mask, pl.rect, pl.split) == (mask, int(rect), split), (prec, "wgrad", n, h, cin, cout, k, dil, pl.split, split)
        finally:
            ops.set_precision("f32")


def test_weights_plan_bookkeeping():
    """glf_weights_plan is host-only: grid bookkeeping of the multi-tensor weight refresh (first workgroup of every job in
    its pass, jobs / workgroups per pass) and its refusals (unsorted passes, misaligned packed images)."""
    import ctypes as C
    from glfusion_amd._lib import WeightJob, WJ_AMAX, WJ_COPY, WJ_PACK, WJ_PASSES, WJ_TAP_MAJOR, WJ_TRANSPOSE, lib
    jobs = (WeightJob * 5)()
    spec = [(WJ_COPY, 0, (5000, 0, 0)), (WJ_AMAX, 1, (4096, 0, 0)), (WJ_AMAX, 1, (4097, 0, 0)), (WJ_TAP_MAJOR, 2, (64, 64, 9)),
            (WJ_PACK, 3, (8192, 0, 0))]
    for j, (kind, ps, dims) in zip(jobs, spec):
        j.src, j.dst, j.amax, j.kind, j.pass_ = 0x10000, 0x20000, 0x30000, kind, ps
        j.d0, j.d1, j.d2 = dims
    pf, pc, pw = (C.c_int * WJ_PASSES)(), (C.c_int * WJ_PASSES)(), (C.c_int64 * WJ_PASSES)()
    args = (C.cast(pf, C.c_void_p), C.cast(pc, C.c_void_p), C.cast(pw, C.c_void_p))
    assert lib.glf_weights_plan(C.cast(jobs, C.c_void_p), 5, *args) == 0
    assert list(pf) == [0, 1, 3, 4] and list(pc) == [1, 2, 1, 1]
    assert list(pw) == [2, 1 + 2, (64 * 64 * 9 + 4095) // 4096, 2]
    assert [j.first_wg for j in jobs] == [0, 0, 1, 0, 0]
    # a 65 x 33 transpose: 3 x 2 tiles of 32 x 32
    t = (WeightJob * 1)()
    t[0].src, t[0].dst, t[0].kind, t[0].pass_, t[0].d0, t[0].d1 = 0x10000, 0x20000, WJ_TRANSPOSE, 2, 65, 33
    assert lib.glf_weights_plan(C.cast(t, C.c_void_p), 1, *args) == 0 and list(pw)[2] == 6
    # refusals
    jobs[0].pass_ = 2                                  # not sorted by pass
    assert lib.glf_weights_plan(C.cast(jobs, C.c_void_p), 5, *args) != 0
    jobs[0].pass_ = 0
    jobs[4].src = 0x10004                              # packed image source not 16-byte aligned
    assert lib.glf_weights_plan(C.cast(jobs, C.c_void_p), 5, *args) != 0
    assert b"aligned" in lib.glf_last_error()


def test_dataset_host_logic_follows_the_loader():
    """The NIfTI-free Dataset shim (glfusion_amd.data.SegPAHDataset; datasets/loader.py:190-458) on the CPU side: id split,
    4 samples per patient and epoch in train mode, labelled-frame selection with Python's `random`, and crop windows drawn
    from a numpy RandomState exactly as MONAI's get_random_patch draws them."""
    import random
    import numpy as np
    import torch
    from glfusion_amd import data
    infos = data.synthetic_infos(["4"], 20, clip_length=12, device="cpu", seed=3)
    infos["x_0"] = dict(infos["0_0"], dataset_name="other")                  # filtered out by set_select
    random.seed(5)
    ds = data.SegPAHDataset(infos, is_train=True, view_num=["4"], single_frame=True, device="cpu", crop_seed=11)
    assert len(ds.train_list) == 16 and len(ds.valid_list) == 2 and len(ds.test_list) == 2 and len(ds) == 64
    assert not (set(ds.train_list) & set(ds.valid_list)) and not (set(ds.train_list) & set(ds.test_list)) and "x_0" not in ds.id_list
    ev = data.SegPAHDataset(infos, is_train=False, data_list=["0_1", "0_3"], view_num=["4"], single_frame=False, clip_length=8, device="cpu")
    assert len(ev) == 2
    # crop windows: numpy RandomState.randint(0, 144 - 112 + 1) per spatial dimension, in order; nothing drawn where sizes agree
    rs, rs2 = np.random.RandomState(11), np.random.RandomState(11)
    assert data.crop_offsets(rs) == (int(rs2.randint(0, 33)), int(rs2.randint(0, 33)))
    a = data.crop_offsets(rs, size=(144, 144, 40), crop=(112, 112, 40))
    assert a == (int(rs2.randint(0, 33)), int(rs2.randint(0, 33)), 0) and rs.randint(0, 1 << 30) == rs2.randint(0, 1 << 30)
    # input_select: only frames with > 100 labelled pixels are candidates; clip mode returns clip_length - 1 frames around one
    img = torch.zeros(30, 30, 10)
    lab = torch.zeros(30, 30, 10)
    lab[:15, :15, 4] = 1.0                                                    # 225 labelled pixels in frame 4 only
    lab[:5, :5, 7] = 2.0                                                      # 25: below the threshold
    random.seed(0)
    f_img, f_lab, idx = ds.input_select(img, lab)
    assert idx == 4 and tuple(f_lab.shape) == (30, 30) and float(f_lab.sum()) == 225.0
    random.seed(1)
    c_img, c_lab, r_idx = ev.input_select(img, lab)
    random.seed(1)
    assert random.choice([4]) == 4
    want_r = random.randint(0, 4)
    assert r_idx == want_r and c_lab.shape[-1] == min(7, 10 - (4 - want_r)) and float(c_lab[..., want_r].sum()) == 225.0
