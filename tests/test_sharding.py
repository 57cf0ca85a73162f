"""world_size-2 gloo tests (CPU) of the multi-GPU exchange steps."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":  # RCCL: one GPU per rank
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2, backend="gloo"):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret, backend),
             nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _keys_union(rank, world):
    from open3d_amd.sharding import allgather_block_keys
    rng = np.random.RandomState(rank)
    k = rng.randint(-5, 5, (40 + 10 * rank, 3)).astype(np.int32)
    u = allgather_block_keys(torch.from_numpy(k), dist)
    return k, u.numpy()


def test_allgather_block_keys_union():
    out = _run(_keys_union)
    want = np.unique(np.concatenate([o[0] for o in out], 0), axis=0)
    for _, u in out:
        assert np.array_equal(u, want)


def _icp_sums(rank, world):
    """Source-sharded 29-sum + all-reduce equals the single-process sum."""
    from open3d_amd import synthetic as syn
    from open3d_amd.sharding import make_allreduce_sum, shard_range
    p = syn.make_icp_pair(6000, 6000, seed=9)
    idx, d2, cnt = orc.hybrid_search(p["target"], p["source"], 0.08, 1)
    corr = idx[:, 0].astype(np.int64)
    b, e = shard_range(6000, rank, world)
    A = np.zeros(32)
    A[:29] = orc.p2plane_accumulate(p["source"][b:e], p["target"],
                                    p["target_normals"], corr[b:e],
                                    accumulate_double=True)
    A[29] = d2[b:e, 0].astype(np.float64).sum()
    A[30] = cnt[b:e].sum()
    A[31] = e - b
    make_allreduce_sum(dist, torch.device("cpu"))(A)
    full = orc.p2plane_accumulate(p["source"], p["target"],
                                  p["target_normals"], corr,
                                  accumulate_double=True)
    return A, full, float(cnt.sum())


def test_icp_sharded_sums_allreduce():
    out = _run(_icp_sums)
    for A, full, cnt in out:
        assert A[31] == 6000 and A[30] == cnt
        assert np.allclose(A[:29], full, rtol=1e-12, atol=1e-12)
    assert np.array_equal(out[0][0], out[1][0])  # identical on every rank
    st, pose, _, _ = orc.decode_and_solve6x6(out[0][0][:29])
    assert st == 0


def test_shard_range_covers_everything():
    from open3d_amd.sharding import shard_range
    for n in (0, 1, 7, 100, 101):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                b, e = shard_range(n, r, w)
                got += list(range(b, e))
            assert got == list(range(n))


def test_block_owner_host_mirror_matches_library():
    """sharding.block_owner (numpy) == o3dmi_block_owner (the function the
    touch kernels use), and it is a balanced partition."""
    import ctypes as C
    import __graft_entry__ as ge
    ge.build()
    from open3d_amd import _lib, sharding
    L = _lib.lib()
    rng = np.random.default_rng(0)
    keys = np.concatenate([rng.integers(-400, 400, (3000, 3)),
                           [[0, 0, 0], [-1, -1, -1], [(1 << 20) - 1] * 3,
                            [-(1 << 20)] * 3]]).astype(np.int32)
    for world in (1, 2, 3, 8):
        host = sharding.block_owner(keys, world)
        dev = np.array([L.o3dmi_block_owner(
            np.ascontiguousarray(k).ctypes.data_as(C.POINTER(C.c_int)), world)
            for k in keys])
        assert np.array_equal(host, dev)
        counts = np.bincount(host, minlength=world)
        assert counts.min() > 0.8 * len(keys) / world
    bad = np.array([1 << 20, 0, 0], np.int32)
    assert L.o3dmi_block_owner(bad.ctypes.data_as(C.POINTER(C.c_int)), 4) == -1


def _blocks_gather(rank, world):
    from open3d_amd.sharding import allgather_blocks
    rng = np.random.RandomState(10 + rank)
    m = 5 + 3 * rank
    keys = rng.randint(-9, 9, (m, 3)).astype(np.int32)
    tsdf = rng.rand(m, 4, 4, 4, 1).astype(np.float32)
    weight = rng.randint(0, 60000, (m, 4, 4, 4, 1)).astype(np.uint16)
    got = allgather_blocks(torch.from_numpy(keys),
                           [torch.from_numpy(tsdf), torch.from_numpy(weight)],
                           dist)
    return (keys, tsdf, weight), [(k.numpy(), [v.numpy() for v in vs])
                                  for k, vs in got]


def test_allgather_blocks_returns_every_ranks_payload():
    """Ragged block payloads (uint16 rows travel as byte views) arrive intact
    and in rank order on every rank."""
    out = _run(_blocks_gather)
    sent = [o[0] for o in out]
    for _, got in out:
        assert len(got) == 2
        for r in range(2):
            k, (t, w) = got[r]
            assert np.array_equal(k, sent[r][0])
            assert t.dtype == np.float32 and np.array_equal(t, sent[r][1])
            assert w.dtype == np.uint16 and np.array_equal(w, sent[r][2])
