"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol the public headers declare (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(o3dmi_[a-z0-9_]+)\s*\(", txt)) -
                  {"o3dmi_icp_callback_t", "o3dmi_allreduce_sum_t"})


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from open3d_amd import _lib
    return _lib


@pytest.mark.parametrize("header", ["o3d_mi355x.h", "o3d_mi355x_host.h"])
def test_every_declared_symbol_is_exported(built, header):
    names = _declared(header)
    assert len(names) >= 10
    so = ctypes.CDLL(built.SO_PATH)
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing
    unbound = [n for n in names if n not in built.PROTOTYPES]
    assert not unbound, "declared in header but not bound in _lib.py: %s" % unbound


def test_library_loads_and_reports_version(built):
    L = built.lib()
    assert L.o3dmi_abi_version() == 1
    assert L.o3dmi_status_string(0) == b"ok"
    assert b"No block is touched" in L.o3dmi_status_string(6)
    assert b"Singular 6x6" in L.o3dmi_status_string(5)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure; the product path must not reach it."""
    pkg = os.path.join(ROOT, "open3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "_oracle" not in txt and "oracle/" not in txt, f
                assert "libo3d_oracle" not in txt, f


def test_missing_extension_fails_loudly(built, monkeypatch):
    monkeypatch.setattr(built, "SO_PATH", "/nonexistent/libo3d_mi355x.so")
    monkeypatch.setattr(built, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU"):
        built.lib()


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers must compile as C99 (what a cgo /
    JNI / ctypes-cffi binding would feed to its C compiler)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "o3d_mi355x.h"\n#include "o3d_mi355x_host.h"\n'
                   "int main(void) { return (int)sizeof(o3dmi_odometry_result_t)"
                   " > 0 ? 0 : 1; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic",
                        "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr


def test_sliced_touch_wire_constants(built):
    """Pure functions of the sliced block touch (no GPU): a chunk is 16 launches
    worth of frames, at most 256 (one frame bit each in a 48-byte record)."""
    L = built.lib()
    assert L.o3dmi_vbg_slice_chunk_frames(12) == 192
    assert L.o3dmi_vbg_slice_chunk_frames(16) == 256
    assert L.o3dmi_vbg_slice_chunk_frames(100) == 256   # clamped to 16 frames
    assert L.o3dmi_vbg_slice_chunk_frames(0) == 16 * 8  # the default group
    assert L.o3dmi_vbg_slice_segment_bytes(None) == 0
