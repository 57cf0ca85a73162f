"""NPZ codec of the drop-in boundary -- host-only, runs
without a GPU. Checks o3dmi_npz_write / o3dmi_npz_read against numpy in both
directions and against the byte layout t::io::WriteNpz produces
(cpp/open3d/t/io/NumpyIO.cpp:157-205 header, :360-466 records), including the
known-answer vectors of the reference's own tests (cpp/tests/t/io/NumpyIO.cpp).
"""
import os
import struct
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def npz():
    import __graft_entry__ as ge
    ge.build()
    from open3d_amd import npz
    return npz


DTYPES = [np.float32, np.float64, np.int8, np.int16, np.int32, np.int64,
          np.uint8, np.uint16, np.uint32, np.uint64, np.bool_]


def _sample(rng):
    d = {}
    for i, dt in enumerate(DTYPES):
        shape = [(), (1,), (7,), (3, 4), (2, 3, 4), (0, 3), (5, 1, 2, 2)][i % 7]
        a = (rng.random(shape) * 200 - 100)
        d["a%d_%s" % (i, np.dtype(dt).name)] = \
            (a > 0) if dt is np.bool_ else a.astype(dt)
    return d


def test_numpy_reads_what_we_write(npz, tmp_path):
    rng = np.random.default_rng(0)
    want = _sample(rng)
    p = str(tmp_path / "w.npz")
    npz.write_npz(p, want)
    got = np.load(p)
    assert sorted(got.files) == sorted(want)
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
        assert got[k].tobytes() == v.tobytes(), k


def test_we_read_what_numpy_writes(npz, tmp_path):
    rng = np.random.default_rng(1)
    want = _sample(rng)
    for name, saver in (("s.npz", np.savez), ("c.npz", np.savez_compressed)):
        p = str(tmp_path / name)
        saver(p, **want)
        got = npz.read_npz(p)
        assert sorted(got) == sorted(want)
        for k, v in want.items():
            assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
            assert got[k].tobytes() == v.tobytes(), k


def test_round_trip_and_zip64_records(npz, tmp_path, monkeypatch):
    rng = np.random.default_rng(2)
    want = _sample(rng)
    want["big"] = rng.standard_normal((300, 16, 16, 16)).astype(np.float32)
    for force in (False, True):
        if force:
            monkeypatch.setenv("O3DMI_NPZ_FORCE_ZIP64", "1")
        p = str(tmp_path / ("r%d.npz" % force))
        npz.write_npz(p, want)
        for got in (npz.read_npz(p), dict(np.load(p))):
            for k, v in want.items():
                assert got[k].dtype == v.dtype and got[k].tobytes() == v.tobytes()
    monkeypatch.delenv("O3DMI_NPZ_FORCE_ZIP64")


def _reference_layout(arrays):
    """Bytes t::io::WriteNpz writes for `arrays` (list of (name, ndarray)),
    restated from NumpyIO.cpp:157-205,360-466."""
    out, central = b"", b""
    for name, a in arrays:
        if a.ndim == 0:
            shape = "()"
        elif a.ndim == 1:
            shape = "(%d,)" % a.shape[0]
        else:
            shape = "(" + ", ".join(str(s) for s in a.shape) + ")"
        kind = {"f": "f", "i": "i", "u": "u", "b": "b"}[a.dtype.kind]
        d = "{'descr': '<%s%d', 'fortran_order': False, 'shape': %s, }" % (
            kind, a.dtype.itemsize, shape)
        d += " " * (16 - (10 + len(d)) % 16 - 1) + "\n"
        npy = b"\x93NUMPY\x01\x00" + struct.pack("<H", len(d)) + d.encode()
        data = a.tobytes()
        crc = zlib.crc32(npy + data) & 0xFFFFFFFF
        var = (name + ".npy").encode()
        nbytes = len(npy) + len(data)
        local = b"PK" + struct.pack("<HHHHHHIIIHH", 0x0403, 20, 0, 0, 0, 0,
                                    crc, nbytes, nbytes, len(var), 0) + var
        central += b"PK" + struct.pack("<HH", 0x0201, 20) + local[4:30] + \
            struct.pack("<HHHII", 0, 0, 0, 0, len(out)) + var
        out += local + npy + data
    n = len(arrays)
    footer = b"PK" + struct.pack("<HHHHHIIH", 0x0605, 0, 0, n, n, len(central),
                                 len(out), 0)
    return out + central + footer


def test_bytes_equal_reference_writer_layout(npz, tmp_path):
    arrays = [("voxel_size", np.array([0.008], np.float32)),
              ("block_resolution", np.array([16], np.int64)),
              ("HIP:0", np.zeros((), np.uint8)),
              ("attr_name_tsdf", np.array([0], np.int32)),
              ("key", np.arange(12, dtype=np.int32).reshape(4, 3)),
              ("value_000", np.linspace(0, 1, 4 * 8, dtype=np.float32)
               .reshape(4, 2, 2, 2, 1))]
    p = str(tmp_path / "b.npz")
    npz.write_npz(p, dict(arrays))
    assert open(p, "rb").read() == _reference_layout(arrays)


def test_reference_known_answers(npz, tmp_path):
    """cpp/tests/t/io/NumpyIO.cpp: NpyIO / NpzIO fixtures -- a {2,2,2} ramp in
    every dtype, a 0-d scalar, {0}-, {0,0}- and {0,1,0}-shaped arrays."""
    t = {"t0": np.arange(8, dtype=np.float32).reshape(2, 2, 2),
         "t1": np.arange(8, dtype=np.int64).reshape(2, 2, 2),
         "scalar": np.array(3.14, np.float64),
         "e0": np.zeros((0,), np.float32), "e00": np.zeros((0, 0), np.int32),
         "e010": np.zeros((0, 1, 0), np.uint8)}
    p = str(tmp_path / "k.npz")
    npz.write_npz(p, t)
    a, b = npz.read_npz(p), np.load(p)
    for k, v in t.items():
        for got in (a[k], b[k]):
            assert got.shape == v.shape and got.dtype == v.dtype
            assert np.array_equal(got, v)
    npz.write_npz(str(tmp_path / "empty.npz"), {})
    assert npz.read_npz(str(tmp_path / "empty.npz")) == {}
    assert len(np.load(str(tmp_path / "empty.npz")).files) == 0


def test_error_paths(npz, tmp_path):
    with pytest.raises(RuntimeError, match="Failed to open"):
        npz.read_npz(str(tmp_path / "missing.npz"))
    p = str(tmp_path / "junk.npz")
    open(p, "wb").write(b"not a zip archive at all, just some bytes....")
    with pytest.raises(RuntimeError):
        npz.read_npz(p)
    f = str(tmp_path / "f.npz")
    np.savez(f, a=np.asfortranarray(np.arange(6.0).reshape(2, 3)))
    with pytest.raises(RuntimeError, match="Fortran"):
        npz.read_npz(f)
    with pytest.raises(ValueError, match="Unsupported dtype"):
        npz.write_npz(str(tmp_path / "c.npz"),
                      {"c": np.zeros(3, np.complex64)})
