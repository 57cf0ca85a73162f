"""world_size-2 gloo tests (CPU) of the multi-GPU exchange steps."""
import os
import shutil
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as orc


RENDEZVOUS_TIMEOUT_S = 120


def _worker(rank, world, rdzv, fn, ret, backend="gloo"):
    # Rendezvous through a FileStore on a fresh temporary path: no TCP port is
    # picked, so nothing can take it between the pick and the listen (the
    # bind-close-rebind pattern this replaces raced with EADDRINUSE).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.pop("MASTER_PORT", None)
    if backend == "nccl":  # RCCL: one GPU per rank
        torch.cuda.set_device(rank)
    import datetime
    dist.init_process_group(
        backend, init_method="file://" + rdzv, rank=rank, world_size=world,
        timeout=datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S))
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2, backend="gloo"):
    mgr = mp.Manager()
    ret = mgr.dict()
    tmp = tempfile.mkdtemp(prefix="o3dmi_rdzv_")
    try:
        mp.spawn(_worker, args=(world, os.path.join(tmp, "store"), fn, ret,
                                backend), nprocs=world, join=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
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


def _comm_custom_transport(rank, world):
    """The library's o3dmi_comm_t over a caller-provided transport: the three
    exchanges (sum all-reduce, all-gather, all-to-all of byte ranges) reach the
    table with the right pointers / counts / offsets. Host buffers stand in
    for device buffers here (the transport is the only thing that touches
    them), gloo moves the bytes."""
    import ctypes as C
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    dist.barrier()
    from open3d_amd import _lib
    L = _lib.lib()

    def view(ptr, n, dtype):
        return np.ctypeslib.as_array(
            C.cast(ptr, C.POINTER(C.c_uint8)), (int(n),)).view(dtype)

    def allreduce(_u, buf, n, _s):
        a = view(buf, 8 * n, np.float64)
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t)
        a[:] = t.numpy()
        return 0

    def allgather(_u, send, recv, nbytes, _s):
        parts = [torch.empty(int(nbytes), dtype=torch.uint8)
                 for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(
            view(send, nbytes, np.uint8).copy()))
        view(recv, nbytes * world, np.uint8)[:] = torch.cat(parts).numpy()
        return 0

    def alltoallv(_u, send, sb, so, recv, rb, ro, _s):
        reqs, ins = [], {}
        for p in range(world):
            if p == rank:
                continue
            if rb[p]:
                ins[p] = torch.empty(int(rb[p]), dtype=torch.uint8)
                reqs.append(dist.irecv(ins[p], src=p))
            if sb[p]:
                reqs.append(dist.isend(torch.from_numpy(
                    view(send + so[p], sb[p], np.uint8).copy()), dst=p))
        for q in reqs:
            q.wait()
        for p, t in ins.items():
            view(recv + ro[p], rb[p], np.uint8)[:] = t.numpy()
        if sb[rank]:
            view(recv + ro[rank], rb[rank], np.uint8)[:] = \
                view(send + so[rank], sb[rank], np.uint8)
        return 0

    cbs = (_lib.TRANSPORT_ALLREDUCE(allreduce),
           _lib.TRANSPORT_ALLGATHER(allgather),
           _lib.TRANSPORT_ALLTOALLV(alltoallv))
    table = _lib.TransportC(*cbs)
    h = C.c_void_p()
    _lib.check(L.o3dmi_comm_create_custom(C.byref(table), None, rank, world,
                                          C.byref(h)), "create")
    assert L.o3dmi_comm_rank(h) == rank and L.o3dmi_comm_world(h) == world
    sums = np.arange(32, dtype=np.float64) * (rank + 1)
    _lib.check(L.o3dmi_comm_allreduce_sum_f64(h, sums.ctypes.data, 32, None),
               "allreduce")
    mine = np.full(5, rank + 10, np.int64)
    everyone = np.zeros(5 * world, np.int64)
    _lib.check(L.o3dmi_comm_allgather(h, mine.ctypes.data,
                                      everyone.ctypes.data, 40, None),
               "allgather")
    # rank r sends (p + 1) * (r + 1) bytes of value 16 * r + p to peer p,
    # nothing to itself
    sb = [(p + 1) * (rank + 1) if p != rank else 0 for p in range(world)]
    so = np.concatenate([[0], np.cumsum(sb)[:-1]]).astype(np.int64)
    send = np.concatenate([np.full(sb[p], 16 * rank + p, np.uint8)
                           for p in range(world)] + [np.zeros(1, np.uint8)])
    rb = [(rank + 1) * (p + 1) if p != rank else 0 for p in range(world)]
    ro = np.concatenate([[0], np.cumsum(rb)[:-1]]).astype(np.int64)
    recv = np.full(sum(rb) + 1, 255, np.uint8)
    i64 = lambda v: (C.c_int64 * world)(*[int(x) for x in v])
    _lib.check(L.o3dmi_comm_alltoallv(h, send.ctypes.data, i64(sb), i64(so),
                                      recv.ctypes.data, i64(rb), i64(ro),
                                      None), "alltoallv")
    L.o3dmi_comm_destroy(h)
    want = np.concatenate([np.full(rb[p], 16 * p + rank, np.uint8)
                           for p in range(world)] + [np.full(1, 255, np.uint8)])
    return sums, everyone, bool(np.array_equal(recv, want))


def test_library_comm_over_a_custom_transport():
    world = 3
    out = _run(_comm_custom_transport, world=world)
    for sums, everyone, ok in out:
        assert np.array_equal(sums, np.arange(32) * 6.0)   # 1 + 2 + 3
        assert np.array_equal(everyone,
                              np.repeat(np.arange(world) + 10, 5))
        assert ok


def _comm_fallback(rank, world):
    """Comm.for_backend on the "nccl" route when the library's own RCCL
    communicator cannot be made on one rank (mode 0) or on any rank (mode 1).
    By default that is an ERROR on every rank (a scaling run must not degrade
    to the torch transport quietly); with allow_fallback every rank must end
    with the torch.distributed transport, and a communicator that was made on
    the other rank is destroyed either way."""
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    dist.barrier()
    from open3d_amd import sharding
    from open3d_amd.sharding import Comm
    real_rccl, real_backend = Comm.rccl, sharding._backend
    out = []
    for mode in (0, 1):
        for allow in (False, True):
            destroyed = []

            class FakeRccl:
                def destroy(self):
                    destroyed.append(True)

            def rccl(_dist, mode=mode):
                if mode == 1 or rank == 1:
                    raise RuntimeError("no RCCL here")
                return FakeRccl()
            Comm.rccl = staticmethod(rccl)
            sharding._backend = lambda _d: "nccl"
            comm, refused = None, None
            try:
                comm = Comm.for_backend(dist, allow_fallback=allow)
            except RuntimeError as e:
                refused = str(e)
            finally:
                Comm.rccl, sharding._backend = real_rccl, real_backend
            if comm is None:
                out.append((allow, False, refused, 0, len(destroyed), None))
                continue
            is_custom = isinstance(comm, Comm) and comm._keep is not None
            world_seen, transport = comm.world, comm.transport
            ranks = comm.rccl_ranks()
            comm.destroy()
            out.append((allow, is_custom, transport, world_seen,
                        len(destroyed), ranks))
    # and the plain routes are what they were
    plain = Comm.for_backend(dist)   # gloo -> torch transport
    ok = plain._keep is not None and plain.transport == "torch"
    plain.destroy()
    return out, ok


def test_comm_for_backend_refuses_a_silent_fallback_and_falls_back_together():
    out = _run(_comm_fallback, world=2)
    for rank, (cases, ok) in enumerate(out):
        assert ok
        assert len(cases) == 4
        for k, (allow, is_custom, what, world_seen, destroyed,
                ranks) in enumerate(cases):
            mode = k // 2
            if not allow:
                # fatal on EVERY rank, also the one whose own communicator
                # was fine, and the message names the switch
                assert not is_custom and "refusing to fall back" in what, \
                    (rank, mode, what)
            else:
                assert is_custom and what == "torch" and world_seen == 2, \
                    (rank, mode)
                assert ranks == 0          # no RCCL communicator behind it
            # mode 0: rank 0 had made its communicator and gave it up
            assert destroyed == (1 if (mode == 0 and rank == 0) else 0)


def _rccl_availability_agreement(rank, world):
    """Comm.rccl: a rank on which librccl cannot be loaded must not skip the
    collectives the others enter: the ranks agree on availability
    FIRST, and all of them raise."""
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    dist.barrier()
    from open3d_amd import _lib
    from open3d_amd.sharding import Comm
    L = _lib.lib()
    real = L.o3dmi_rccl_available
    L.o3dmi_rccl_available = (lambda: 0) if rank == 1 else (lambda: 1)
    try:
        try:
            Comm.rccl(dist)
            return "no error"
        except RuntimeError as e:
            return str(e)
    finally:
        L.o3dmi_rccl_available = real


def test_comm_rccl_agrees_on_availability_before_any_collective():
    out = _run(_rccl_availability_agreement, world=2)
    assert "librccl not loadable" in out[1]
    assert "not available on every rank" in out[0]
    assert "available here" in out[0] or "(available" in out[0]


def test_rccl_is_resolved_at_run_time_only():
    """The library carries no link dependency on RCCL (dlopen at first use);
    without a communicator the exchange entry points fail cleanly."""
    import ctypes as C
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    from open3d_amd import _lib
    needed = subprocess.run(["readelf", "-d", _lib.SO_PATH],
                            capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()
    L = _lib.lib()
    assert L.o3dmi_rccl_available() in (0, 1)
    assert L.o3dmi_set_comm(None) == 0 and L.o3dmi_set_rccl_comm(None) == 0
    assert L.o3dmi_comm_world(None) == 1 and L.o3dmi_comm_rank(None) == 0
    assert L.o3dmi_comm_allreduce_sum_f64(None, None, 0, None) != 0
