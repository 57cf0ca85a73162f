"""CPU study of the ray-cast march (no GPU): how long is a ray's chain of
DEPENDENT voxel samples, and how much of it is predictable?

Builds the tracking scene's grid with the oracle (touch + integrate of
`--frames` synthetic VGA frames, as tools/bench_raycast.py does on the GPU),
then replays the reference's march (VoxelBlockGridImpl.h:784-870) for every
pixel of the last pose in numpy and classifies each sample by the stride that
follows it:

  empty   block not allocated          -> t += block_size   (no voxel load)
  crawl   tsdf * sdf_trunc < voxel     -> t += voxel_size   (predictable)
  one     tsdf == 1                    -> t += sdf_trunc
  mid     anything else                -> t += tsdf * sdf_trunc

and counts the steps a ray needs when a window of D samples (t, t + voxel,
t + 2 voxel, ...) is taken at once during a crawl. The launch lasts as long as
its slowest wave, i.e. the per-8x8-tile maximum. 60 frames, 640x480: 6.7
samples per ray (2.0 empty, 0.8 crawl, 1.0 one, 1.9 mid) -- but the slowest
wave takes 58 voxel samples in a row, 54 of them crawl steps; 33 / 22 / 19
with windows of 2 / 4 / 8. That tail is what the kernel's cooperative march
(idle lanes of the wave sample ahead for the rays still crawling) removes;
O3DMI_RAYCAST_STEPS=1 prints the same distribution measured on the GPU.

The valid-pixel fraction is checked against the oracle's own ray cast.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc  # noqa: E402
import _scene as sc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--windows", default="1,2,4,8")
a = ap.parse_args()
W, H = 640, 480
VOX, RES, TM = np.float32(0.008), 16, 8.0
trunc = np.float32(VOX * TM)
bs = np.float32(VOX * RES)
cap = 8192
hm = orc.HashMap(cap)
tsdf = np.zeros((cap, RES ** 3), np.float32)
wgt = np.zeros((cap, RES ** 3), np.uint16)
col = np.zeros((cap, RES ** 3, 3), np.uint16)
for k in range(0, a.frames * 2, 2):
    d, c, K, Ts = sc.frames(k, 1, W, H)
    T = Ts[0]
    keys = orc.depth_touch(d[0], K, T, RES, float(VOX), float(trunc), 1000.0,
                           3.0)
    hm.activate(keys)
    idx, _ = hm.find(keys)
    orc.integrate(d[0], c[0], idx, hm.key_buffer(), tsdf, wgt, col, K, K, T,
                  RES, float(VOX), float(trunc), 1000.0, 3.0)
rng, _ = orc.estimate_range(keys, K, T, H, W, 8, RES, float(VOX), 0.1, 3.0)
ref = orc.raycast(hm, tsdf, wgt, col, rng, K, T, H, W, RES, float(VOX), 1000.0,
                  0.1, 3.0, 1.0, TM, 8, attrs=("depth",))
nb = hm.size()
bk = hm.key_buffer()[:nb]
kmin, kmax = bk.min(0) - 2, bk.max(0) + 3
dense = -np.ones(kmax - kmin, int)
dense[tuple((bk - kmin).T)] = np.arange(nb)

Tinv = np.linalg.inv(T)
R = Tinv[:3, :3].astype(np.float32)
o = Tinv[:3, 3].astype(np.float32)
ys, xs = np.mgrid[0:H, 0:W]
xs = xs.ravel().astype(np.float32)
ys = ys.ravel().astype(np.float32)
cam = np.stack([(xs - np.float32(K[0, 2])) / np.float32(K[0, 0]),
                (ys - np.float32(K[1, 2])) / np.float32(K[1, 1]),
                np.ones_like(xs)], 1)
d = (cam @ R.T + o) - o  # the kernel's order: ray point minus origin
r = rng[ys.astype(int) // 8, xs.astype(int) // 8]
N = W * H


def march(D):
    t, tmax = r[:, 0].copy(), r[:, 1].copy()
    active = t < tmax
    tcur = np.ones(N, np.float32)
    n = {k: np.zeros(N, int) for k in
         ("samples", "empty", "crawl", "one", "mid", "trips")}
    preloaded = np.zeros(N, int)
    found = np.zeros(N, bool)
    while active.any():
        idx = np.nonzero(active)[0]
        p = o + t[idx, None] * d[idx]
        b = np.floor(p / bs).astype(int)
        inb = ((b >= kmin) & (b < kmax)).all(1)
        bi = -np.ones(len(idx), int)
        bb = b[inb] - kmin
        bi[inb] = dense[bb[:, 0], bb[:, 1], bb[:, 2]]
        n["samples"][idx] += 1
        emp = bi < 0
        n["empty"][idx[emp]] += 1
        t[idx[emp]] += bs
        preloaded[idx[emp]] = 0
        f = ~emp
        fi = idx[f]
        v = np.clip(((p[f] - b[f] * bs) / VOX).astype(int), 0, RES - 1)
        lin = v[:, 2] * RES * RES + v[:, 1] * RES + v[:, 0]
        ts = tsdf[bi[f], lin]
        w = wgt[bi[f], lin].astype(np.float32)
        need = preloaded[fi] <= 0
        n["trips"][fi[need]] += 1
        hit = (tcur[fi] > 0) & (w >= 1.0) & (ts <= 0)
        tcur[fi] = ts
        found[fi[hit]] = True
        cont = ~hit
        ci, tsc = fi[cont], ts[cont]
        delta = tsc * trunc
        one, crawl = tsc == 1.0, delta < VOX
        n["one"][ci[one]] += 1
        n["crawl"][ci[crawl]] += 1
        n["mid"][ci[~one & ~crawl]] += 1
        rem = np.where(need[cont], D - 1, preloaded[ci] - 1)
        preloaded[ci] = np.where(crawl, rem, 0)
        t[ci] += np.where(crawl, VOX, delta)
        active[idx] &= t[idx] < tmax[idx]
        active[fi[hit]] = False
    return n, found


def line(name, x):
    return "%-16s mean %6.2f  p50 %3d  p90 %3d  p99 %3d  max %3d" % (
        (name, x.mean()) + tuple(np.percentile(x, [50, 90, 99]).astype(int))
        + (x.max(),))


for D in [int(x) for x in a.windows.split(",")]:
    n, found = march(D)
    if D == int(a.windows.split(",")[0]):
        assert abs(found.mean() - (ref["depth"] > 0).mean()) < 1e-3, "march != oracle"
        print("blocks %d, rays that find a surface %.3f (= the oracle's)" %
              (nb, found.mean()))
        for k in ("samples", "empty", "crawl", "one", "mid"):
            print(line(k + " / ray", n[k]))
    tr = n["trips"]
    print("window %d:" % D)
    print("  " + line("loads / ray", tr))
    print("  " + line("loads / wave",
                      tr.reshape(H // 8, 8, W // 8, 8).max((1, 3))))
