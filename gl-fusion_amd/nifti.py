"""Host-side NIfTI-1 volume reader (datasets/loader.py:233-234).

The reference reads every (patient, view) volume with `np.array(nib.load(path).dataobj)`: the on-disk voxel array in its
stored dtype, first index fastest (a 2-D+t echo clip is W x H x T, e.g. 800 x 600 x 172), multiplied by scl_slope and
shifted by scl_inter only when the header carries a usable scaling (nibabel's ArrayProxy: slope finite and non-zero and
not the identity; the result is then float64).  nibabel is not installed in this image and cannot be (no network), so
this is a restatement of the published NIfTI-1.1 header layout (nifti1.h), not of nibabel's code: PARITY UNPINNED against
nibabel itself; pinned by byte-level fixtures the tests build straight from the header layout, independent of `write`.
Single-file `.nii` / `.nii.gz`, either byte order.  ANALYZE 7.5 pairs and NIfTI-2 are refused (the reference's data is
NIfTI-1 `.nii.gz`).
"""
from __future__ import annotations

import gzip
import os
import struct
from typing import Tuple

import numpy as np

# nifti1.h datatype codes -> numpy
_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_CODES = {np.dtype(v).newbyteorder("=").str[1:]: k for k, v in _DTYPES.items()}
HEADER_BYTES = 348


def _open_r(path):
    path = os.fspath(path)
    with open(path, "rb") as f:
        gz = f.read(2) == b"\x1f\x8b"          # by content, as nibabel does for a mis-named file
    return gzip.open(path, "rb") if gz else open(path, "rb")


def read_header(raw: bytes) -> dict:
    """Fields of a 348-byte NIfTI-1 header that locate and type the voxel array.  Byte order: the one in which
    sizeof_hdr reads 348 (nifti1.h's rule)."""
    if len(raw) < HEADER_BYTES:
        raise ValueError("nifti: file shorter than a NIfTI-1 header")
    for bo in ("<", ">"):
        if struct.unpack_from(bo + "i", raw, 0)[0] == HEADER_BYTES:
            break
    else:
        if struct.unpack_from("<i", raw, 0)[0] == 540 or struct.unpack_from(">i", raw, 0)[0] == 540:
            raise ValueError("nifti: NIfTI-2 files are not supported")
        raise ValueError("nifti: sizeof_hdr is not 348 in either byte order (not a NIfTI-1 file)")
    magic = raw[344:348]
    if magic not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"nifti: bad magic {magic!r} (ANALYZE 7.5 files are not supported)")
    if magic == b"ni1\0":
        raise ValueError("nifti: header/image pairs (.hdr + .img) are not supported; use a single-file .nii / .nii.gz")
    dim = struct.unpack_from(bo + "8h", raw, 40)
    if not 1 <= dim[0] <= 7:
        raise ValueError(f"nifti: dim[0] = {dim[0]} is outside 1..7")
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    if any(d < 1 for d in shape):
        raise ValueError(f"nifti: non-positive extent in dim = {dim}")
    datatype, bitpix = struct.unpack_from(bo + "2h", raw, 70)
    if datatype not in _DTYPES:
        raise ValueError(f"nifti: datatype code {datatype} is not supported")
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(bo)
    if bitpix != dt.itemsize * 8:
        raise ValueError(f"nifti: bitpix {bitpix} does not match datatype code {datatype}")
    vox_offset, slope, inter = struct.unpack_from(bo + "3f", raw, 108)
    return {"byteorder": bo, "shape": shape, "dtype": dt, "vox_offset": int(vox_offset) if vox_offset >= HEADER_BYTES else 352,
            "scl_slope": float(slope), "scl_inter": float(inter), "pixdim": struct.unpack_from(bo + "8f", raw, 76)}


def _scaling(h: dict) -> Tuple[float, float]:
    """(slope, inter) to apply, or (1, 0): nibabel ignores a slope that is 0 or not finite, and an intercept that is not finite."""
    s, i = h["scl_slope"], h["scl_inter"]
    if not np.isfinite(s) or s == 0.0:
        return 1.0, 0.0
    return s, (i if np.isfinite(i) else 0.0)


def read(path) -> np.ndarray:
    """The voxel array of a single-file NIfTI-1 volume, as `np.array(nib.load(path).dataobj)` returns it: shape dim[1..dim[0]],
    first index fastest on disk (returned Fortran-ordered, native byte order), stored dtype unless a scaling applies."""
    with _open_r(path) as f:
        raw = f.read(HEADER_BYTES)
        h = read_header(raw)
        skip = h["vox_offset"] - HEADER_BYTES
        if skip:
            f.read(skip)
        count = int(np.prod(h["shape"]))
        buf = f.read(count * h["dtype"].itemsize)
    if len(buf) != count * h["dtype"].itemsize:
        raise ValueError(f"nifti: {path}: voxel data truncated ({len(buf)} of {count * h['dtype'].itemsize} bytes)")
    arr = np.frombuffer(buf, dtype=h["dtype"]).reshape(h["shape"], order="F")
    arr = arr.astype(h["dtype"].newbyteorder("="), copy=False)
    slope, inter = _scaling(h)
    if (slope, inter) != (1.0, 0.0):
        arr = arr.astype(np.float64) * slope + inter
    return arr


def write(path, volume: np.ndarray, pixdim=(1.0, 1.0, 1.0)) -> None:
    """A minimal single-file NIfTI-1 volume (little-endian, no extensions, no scaling) holding `volume` -- for tests and for
    materialising SyntheticPatients on disk; the reference only reads."""
    v = np.asarray(volume)
    key = v.dtype.newbyteorder("=").str[1:]
    if key not in _CODES:
        raise ValueError(f"nifti: dtype {v.dtype} has no NIfTI-1 datatype code")
    if not 1 <= v.ndim <= 7:
        raise ValueError("nifti: 1 to 7 dimensions")
    hdr = bytearray(HEADER_BYTES)
    struct.pack_into("<i", hdr, 0, HEADER_BYTES)
    struct.pack_into("<8h", hdr, 40, v.ndim, *(list(v.shape) + [1] * (7 - v.ndim)))
    struct.pack_into("<2h", hdr, 70, _CODES[key], v.dtype.itemsize * 8)
    pd = [1.0] + [float(x) for x in pixdim][:v.ndim] + [1.0] * 7
    struct.pack_into("<8f", hdr, 76, *pd[:8])
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2                                   # xyzt_units: millimetres
    hdr[344:348] = b"n+1\0"
    with _open_w(path) as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")                       # extension flag: none
        f.write(np.asfortranarray(v.astype(v.dtype.newbyteorder("<"), copy=False)).tobytes(order="F"))


def _open_w(path):
    path = os.fspath(path)
    return gzip.open(path, "wb", compresslevel=1) if path.endswith(".gz") else open(path, "wb")
