"""ctypes binding of libglfusion_hip.so (the C ABI declared in include/glfusion.h).

The prototypes are generated from the header itself, so the Python side cannot drift from
the ABI.  There is NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (the product path never routes around the HIP engine).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Tuple

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(ROOT, "include", "glfusion.h")
LIB_PATH = os.environ.get("GLF_LIB_PATH") or os.path.join(_PKG, "lib", "libglfusion_hip.so")   # override: A/B of two builds


class GemmParams(C.Structure):
    """Mirror of glf_gemm_params (include/glfusion.h)."""
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("taps", C.c_int32), ("tap_mask", C.c_uint32),
        ("tap_stride_b", C.c_int64),
        ("gather", C.c_int32),
        ("n_img", C.c_int32), ("hs", C.c_int32), ("ws", C.c_int32), ("hd", C.c_int32), ("wd", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
        ("batch", C.c_int32),
        ("batch_stride_a", C.c_int64), ("batch_stride_b", C.c_int64), ("batch_stride_c", C.c_int64),
        ("alpha", C.c_float), ("accumulate", C.c_int32), ("split", C.c_int32), ("rect", C.c_int32),
        ("amax_a", C.c_void_p), ("amax_b", C.c_void_p), ("amax_c", C.c_void_p),
        ("colstats", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("a_presplit", C.c_int32), ("b_presplit", C.c_int32),
        ("precision", C.c_int32),
        ("colmax", C.c_void_p),
        ("c_dtype", C.c_int32), ("reserved0", C.c_int32),
    ]


class AttnParams(C.Structure):
    """Mirror of glf_attn_params (include/glfusion.h)."""
    _fields_ = [("frames", C.c_int32), ("L", C.c_int32), ("ci", C.c_int32),
                ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldy", C.c_int64), ("lddy", C.c_int64), ("ldd", C.c_int64)]


class TpaviParams(C.Structure):
    """Mirror of glf_tpavi_params (include/glfusion.h)."""
    _fields_ = [("n", C.c_int32), ("L", C.c_int32), ("c", C.c_int32), ("ci", C.c_int32), ("training", C.c_int32),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("ln_eps", C.c_float)]


class ConvParams(C.Structure):
    """Mirror of glf_conv_params (include/glfusion.h)."""
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("kh", C.c_int32),
                ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32), ("precision", C.c_int32),
                ("amax_x", C.c_void_p), ("amax_w", C.c_void_p), ("amax_dy", C.c_void_p), ("amax_out", C.c_void_p), ("colstats", C.c_void_p)]


class ConvPlan(C.Structure):
    """Mirror of glf_conv_plan (include/glfusion.h)."""
    _fields_ = [("ho", C.c_int32), ("wo", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("taps", C.c_int32),
                ("kept_taps", C.c_int32), ("tap_mask", C.c_uint32), ("plain", C.c_int32), ("rect", C.c_int32), ("split", C.c_int32),
                ("zero_fill", C.c_int32), ("colstats_ok", C.c_int32), ("workspace_bytes", C.c_int64)]


class WeightJob(C.Structure):
    """Mirror of glf_weight_job (include/glfusion.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("amax", C.c_void_p), ("kind", C.c_int32), ("pass_", C.c_int32),
                ("d0", C.c_int32), ("d1", C.c_int32), ("d2", C.c_int32), ("reserved", C.c_int32), ("first_wg", C.c_int64)]


WJ_COPY, WJ_AMAX, WJ_TAP_MAJOR, WJ_TAP_MAJOR_T, WJ_TRANSPOSE, WJ_PACK, WJ_ZERO, WJ_CVT_BF16, WJ_PASSES = 0, 1, 2, 3, 4, 5, 6, 7, 4
WJ_PASS_OF = {WJ_COPY: 0, WJ_ZERO: 0, WJ_AMAX: 1, WJ_TAP_MAJOR: 2, WJ_TAP_MAJOR_T: 2, WJ_TRANSPOSE: 2, WJ_PACK: 3, WJ_CVT_BF16: 3}


_SCALARS = {
    "int": C.c_int, "float": C.c_float, "double": C.c_double, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
    "size_t": C.c_size_t, "glf_stream_t": C.c_void_p, "void": None, "uint8_t": C.c_uint8, "uint32_t": C.c_uint32, "int32_t": C.c_int32,
}


def _ctype(decl: str):
    decl = re.sub(r"/\*.*?\*/", "", decl).strip()
    if decl == "void":
        return None
    if "glf_gemm_params" in decl:
        return C.POINTER(GemmParams)
    if "glf_attn_params" in decl:
        return C.POINTER(AttnParams)
    if "glf_tpavi_params" in decl:
        return C.POINTER(TpaviParams)
    if "glf_conv_params" in decl:
        return C.POINTER(ConvParams)
    if "glf_conv_plan" in decl:
        return C.POINTER(ConvPlan)
    if "*" in decl:
        return C.c_void_p
    t = decl.replace("const", "").split()
    base = t[0]
    if base not in _SCALARS:
        raise ValueError(f"unmapped C type in {decl!r}")
    return _SCALARS[base]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """{symbol: (restype, [argtypes])} for every function the header declares."""
    text = open(path).read()
    text_nc = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?m)^(const char\*|int|size_t)\s+(glf_\w+)\s*\(([^;{]*?)\)\s*;", text_nc, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = C.c_char_p if "char" in ret else (C.c_size_t if ret == "size_t" else C.c_int)
        argtypes = [a for a in (_ctype(x) for x in args.split(",")) if a is not None] if args.strip() else []
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self) -> None:
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"GL-Fusion HIP engine not built: {LIB_PATH} is missing. Build it with "
                f"`make -C {os.path.join(_PKG, 'csrc')}` (or __graft_entry__.build()). There is no CPU fallback.")
        # torch first: it brings its own HIP runtime (libamdhip64); loaded after ours, the process would hold two runtimes and the
        # library's would not see torch's device context ("no ROCm-capable device is detected" from the first glf_* call)
        import torch  # noqa: F401
        dll = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(dll, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        self._dll = dll
        return dll

    def __getattr__(self, name: str):
        if name.startswith("glf_"):
            return getattr(self.load(), name)
        raise AttributeError(name)


lib = _Lib()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.glf_last_error()
        raise RuntimeError(f"glfusion HIP call failed ({what}, status {rc}): {msg.decode() if msg else '?'}")
