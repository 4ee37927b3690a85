"""Host NIfTI-1 reader (datasets/loader.py:233-234 reads volumes with nibabel, which is absent here).

The fixtures are built byte by byte from the NIfTI-1.1 header layout (nifti1.h offsets, written out in `_raw_file` below),
NOT with glfusion_amd.nifti.write, so the reader is checked against the format and the writer against the reader.
"""
import gzip
import struct

import numpy as np
import pytest

from glfusion_amd import nifti


def _raw_file(arr: np.ndarray, code: int, bo: str = "<", slope: float = 0.0, inter: float = 0.0, vox_offset: float = 352.0,
              magic: bytes = b"n+1\0") -> bytes:
    """nifti1.h: sizeof_hdr @0 (int32 = 348), dim[8] @40 (int16), datatype @70, bitpix @72 (int16), pixdim[8] @76 (float32),
    vox_offset @108, scl_slope @112, scl_inter @116 (float32), magic @344; voxels follow at vox_offset, first index fastest."""
    h = bytearray(348)
    struct.pack_into(bo + "i", h, 0, 348)
    struct.pack_into(bo + "8h", h, 40, arr.ndim, *(list(arr.shape) + [1] * (7 - arr.ndim)))
    struct.pack_into(bo + "2h", h, 70, code, arr.dtype.itemsize * 8)
    struct.pack_into(bo + "8f", h, 76, 1, 0.5, 0.5, 1, 1, 1, 1, 1)
    struct.pack_into(bo + "3f", h, 108, vox_offset, slope, inter)
    h[344:348] = magic
    pad = b"\0" * (int(vox_offset) - 348)
    return bytes(h) + pad + arr.astype(arr.dtype.newbyteorder(bo)).tobytes(order="F")


@pytest.mark.parametrize("dtype,code", [("u1", 2), ("i2", 4), ("i4", 8), ("f4", 16), ("f8", 64), ("i1", 256), ("u2", 512)])
@pytest.mark.parametrize("bo", ["<", ">"])
def test_reads_every_datatype_in_both_byte_orders(tmp_path, dtype, code, bo):
    rng = np.random.default_rng(3)
    a = (rng.integers(0, 100, size=(7, 5, 3)) if dtype[0] in "ui" else rng.standard_normal((7, 5, 3))).astype(dtype)
    p = tmp_path / "v.nii"
    p.write_bytes(_raw_file(a, code, bo))
    got = nifti.read(p)
    assert got.dtype == np.dtype(dtype) and got.shape == a.shape and got.dtype.isnative
    np.testing.assert_array_equal(got, a)
    # element (i, j, k) sits at byte offset 352 + itemsize * (i + 7 j + 35 k): first index fastest
    raw = p.read_bytes()
    i, j, k = 4, 2, 1
    off = 352 + a.dtype.itemsize * (i + 7 * j + 35 * k)
    assert np.frombuffer(raw[off:off + a.dtype.itemsize], dtype=np.dtype(dtype).newbyteorder(bo))[0] == a[i, j, k]


def test_gzip_is_detected_by_content_and_echo_clip_shape(tmp_path):
    a = np.arange(8 * 6 * 4, dtype=np.uint8).reshape(8, 6, 4)          # W x H x T like the reference's 800 x 600 x 172
    p = tmp_path / "clip.nii.gz"
    p.write_bytes(gzip.compress(_raw_file(a, 2)))
    np.testing.assert_array_equal(nifti.read(p), a)
    q = tmp_path / "misnamed.nii"                                       # gzip bytes under a .nii name
    q.write_bytes(gzip.compress(_raw_file(a, 2)))
    np.testing.assert_array_equal(nifti.read(q), a)


def test_scaling_follows_the_header(tmp_path):
    a = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    p = tmp_path / "s.nii"
    p.write_bytes(_raw_file(a, 4, slope=0.5, inter=-3.0))
    got = nifti.read(p)
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, a.astype(np.float64) * 0.5 - 3.0)
    for slope, inter in ((0.0, 7.0), (float("nan"), 1.0), (1.0, 0.0)):     # unusable or identity scaling: stored values, stored dtype
        p.write_bytes(_raw_file(a, 4, slope=slope, inter=inter))
        got = nifti.read(p)
        assert got.dtype == np.int16
        np.testing.assert_array_equal(got, a)


def test_vox_offset_and_extensions_are_skipped(tmp_path):
    a = np.arange(30, dtype=np.float32).reshape(5, 6)
    p = tmp_path / "e.nii"
    p.write_bytes(_raw_file(a, 16, vox_offset=416.0))                   # 64 bytes of header extension before the voxels
    np.testing.assert_array_equal(nifti.read(p), a)


def test_refusals(tmp_path):
    a = np.zeros((2, 2), dtype=np.uint8)
    p = tmp_path / "bad.nii"
    p.write_bytes(b"\0" * 100)
    with pytest.raises(ValueError, match="shorter"):
        nifti.read(p)
    p.write_bytes(_raw_file(a, 2, magic=b"ni1\0"))
    with pytest.raises(ValueError, match="pairs"):
        nifti.read(p)
    p.write_bytes(_raw_file(a, 2, magic=b"\0\0\0\0"))
    with pytest.raises(ValueError, match="magic"):
        nifti.read(p)
    p.write_bytes(_raw_file(a, 128))                                    # RGB24: not a scalar volume
    with pytest.raises(ValueError, match="datatype"):
        nifti.read(p)
    p.write_bytes(_raw_file(a, 2)[:-1])
    with pytest.raises(ValueError, match="truncated"):
        nifti.read(p)
    raw = bytearray(_raw_file(a, 2))
    struct.pack_into("<i", raw, 0, 540)
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="NIfTI-2"):
        nifti.read(p)


@pytest.mark.parametrize("name", ["w.nii", "w.nii.gz"])
def test_writer_round_trip_and_layout(tmp_path, name):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 255, size=(9, 7, 5)).astype(np.uint8)
    p = tmp_path / name
    nifti.write(p, a)
    np.testing.assert_array_equal(nifti.read(p), a)
    raw = gzip.decompress(p.read_bytes()) if name.endswith(".gz") else p.read_bytes()
    assert raw == _raw_file(a, 2)[:76] + raw[76:108] + _raw_file(a, 2, slope=1.0)[108:123] + b"\x02" + _raw_file(a, 2)[124:]
    lab = rng.integers(0, 5, size=(9, 7, 5)).astype(np.float32)
    nifti.write(p, lab)
    got = nifti.read(p)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, lab)
