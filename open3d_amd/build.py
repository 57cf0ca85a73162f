"""Builds the MI355X backend (hand-written HIP for gfx950) in-tree.

    python -m open3d_amd.build            # incremental
    python -m open3d_amd.build --force

hipcc cross-compiles for gfx950 without a GPU. The output
open3d_amd/lib/libo3d_mi355x.so is git-ignored but travels with the tree.

Flags: -ffp-contract=off (no FMA contraction) and HIP's default correctly
rounded float32 divide/sqrt keep the kernels' float arithmetic identical to
Open3D's CPU tensor path (SSE2, no FMA: cmake/Open3DSetGlobalProperties.cmake
sets no -march), which the bit-exact block-activation parity relies on.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
SO = os.path.join(LIBDIR, "libo3d_mi355x.so")

ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = [
    "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
]


def sources():
    out = []
    for dirpath, _, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith(".hip") or f.endswith(".cpp"):
                out.append(os.path.join(dirpath, f))
    return sorted(out)


def headers():
    out = [os.path.join(ROOT, "include", "o3d_mi355x.h"),
           os.path.join(ROOT, "include", "o3d_mi355x_host.h")]
    for dirpath, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith(".h"):
                out.append(os.path.join(dirpath, f))
    return out


def _obj(src):
    rel = os.path.relpath(src, CSRC).replace(os.sep, "_")
    return os.path.join(OBJDIR, rel + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(src):
    obj = _obj(src)
    cmd = [HIPCC] + CXXFLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) \
        + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" %
                           (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    hdrs = headers() + [os.path.abspath(__file__)]
    todo = [s for s in srcs if force or _stale(_obj(s), [s] + hdrs)]
    if verbose and todo:
        print("[open3d_amd.build] compiling %d file(s) for %s" %
              (len(todo), ARCH), flush=True)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(_compile, todo))
    objs = [_obj(s) for s in srcs]
    if force or todo or _stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + \
              ["-lz", "-o", SO]  # zlib: CRC-32 / inflate of the NPZ codec
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[open3d_amd.build] linked", SO, flush=True)
    return SO


def build_examples(verbose=True):
    """examples/*.cpp: plain C++ callers of the C ABI (no Python, no torch),
    linked against the library built above."""
    out = []
    exdir = os.path.join(ROOT, "examples")
    for f in sorted(os.listdir(exdir)) if os.path.isdir(exdir) else []:
        if not f.endswith(".cpp"):
            continue
        src = os.path.join(exdir, f)
        exe = src[:-4]
        if _stale(exe, [src, SO] + headers() +
                  [os.path.join(ROOT, "include", "o3d_mi355x_host.h")]):
            cmd = [HIPCC, "-O2", "-std=c++17", src,
                   "-I" + os.path.join(ROOT, "include"), "-L" + LIBDIR,
                   "-lo3d_mi355x", "-Wl,-rpath,$ORIGIN/../open3d_amd/lib",
                   "-o", exe]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("example build failed for %s:\n%s\n%s" %
                                   (src, r.stdout, r.stderr))
            if verbose:
                print("[open3d_amd.build] built", exe, flush=True)
        out.append(exe)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_examples()
