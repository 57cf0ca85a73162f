"""The reference-side binding (INTEGRATION.md sections 0-5) as a compiled
translation unit: integration/hip_backend.cpp is type-checked against the
reference's own headers (from /root/reference, over oracle/ref_shim's stand-in
Tensor) and include/o3d_mi355x.h, and linked against the built library.

Every `...HIP` function in it is static_assert-ed to have the type of the
per-device function it stands beside (DepthTouchCPU/CUDA, IntegrateCPU/CUDA,
RayCastCPU/CUDA, ComputePosePointToPlaneCPU/CUDA, TransformPointsCPU/CUDA), and
HIPHashBackend overrides the reference's DeviceHashBackend virtuals, so a
signature drift on either side fails here. CPU-only; skipped where the
reference tree is absent (the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/cpp"
SRC = os.path.join(ROOT, "integration", "hip_backend.cpp")
INC = ["-I", os.path.join(ROOT, "integration", "shim"),
       "-I", os.path.join(ROOT, "oracle", "ref_shim"), "-I", REF,
       "-I", os.path.join(ROOT, "include")]

pytestmark = pytest.mark.skipif(
    not os.path.isdir(os.path.join(REF, "open3d")) or
    shutil.which("g++") is None,
    reason="needs the reference tree and g++")


def test_binding_type_checks_against_the_reference_headers():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall"] + INC +
                       [SRC], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_a_drifted_signature_is_caught(tmp_path):
    """The static_asserts bite: DepthTouchHIP with one argument type changed
    no longer matches the dispatcher's per-device signature."""
    bad = tmp_path / "bad.cpp"
    text = open(SRC).read()
    assert "index_t stride) {" in text
    bad.write_text(text.replace("index_t stride) {", "int64_t stride) {", 1))
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only"] + INC +
                       [str(bad)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "DepthTouchHIP has the dispatcher's per-device signature" in r.stderr


def test_binding_links_against_the_library(tmp_path):
    lib_dir = os.path.join(ROOT, "open3d_amd", "lib")
    if not os.path.exists(os.path.join(lib_dir, "libo3d_mi355x.so")):
        pytest.skip("library not built")
    out = tmp_path / "libo3d_binding.so"
    r = subprocess.run(["g++", "-std=c++17", "-shared", "-fPIC"] + INC +
                       [SRC, "-o", str(out), "-Wl,--no-undefined",
                        "-L", lib_dir, "-lo3d_mi355x",
                        "-Wl,-rpath," + lib_dir],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    # every o3dmi_* symbol the binding needs is an export of the library
    nm = subprocess.run(["nm", "-D", "--undefined-only", str(out)],
                        capture_output=True, text=True).stdout
    need = {ln.split()[-1] for ln in nm.splitlines() if " o3dmi_" in ln}
    have = subprocess.run(
        ["nm", "-D", "--defined-only",
         os.path.join(lib_dir, "libo3d_mi355x.so")],
        capture_output=True, text=True).stdout
    have = {ln.split()[-1] for ln in have.splitlines()}
    assert len(need) >= 20, need
    assert need <= have, need - have
