"""SURVEY 8(e), RayCast row: the ray cast of a replicated grid sharded by pixel
rows over the ranks of a communicator (o3dmi_vbg_ray_cast_sharded: a band of
whole 8-row tiles per rank through o3dmi_vbg_raycast_rows, one all-gather per
requested map). Every rank must end with the maps of the single-rank ray cast
BIT FOR BIT -- a pixel's result does not depend on the band it is rendered in.

The ranks are host threads sharing the one GPU (each with its own stream and
its own thread-local communicator over a loopback transport table), the way
the sharded-ICP tests run theirs."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

import _scene as sc

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from open3d_amd import _lib, geometry
    return _lib, geometry


class _Loopback:
    """All-gather between host threads of one process: every rank copies its
    segment into a shared device buffer between two barriers."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=120)
        self.shared = {}
        self.lock = threading.Lock()

    def comm(self, rank):
        from open3d_amd import _lib
        from open3d_amd.core import tensor_from_ptr
        from open3d_amd.sharding import Comm
        world = self.world

        def allgather(_user, send, recv, nbytes, stream_ptr):
            try:
                n = int(nbytes)
                torch.cuda.ExternalStream(stream_ptr).synchronize() \
                    if stream_ptr else torch.cuda.synchronize()
                with self.lock:
                    if n not in self.shared:
                        self.shared[n] = torch.empty(n * world,
                                                     dtype=torch.uint8,
                                                     device="cuda")
                buf = self.shared[n]
                # (the send segment may lie inside recv: in-place gather)
                seg = tensor_from_ptr(send, (n,), _lib.U8, None).clone()
                self.barrier.wait()
                buf[rank * n:(rank + 1) * n].copy_(seg)
                torch.cuda.synchronize()
                self.barrier.wait()
                tensor_from_ptr(recv, (n * world,), _lib.U8, None).copy_(buf)
                torch.cuda.synchronize()
                self.barrier.wait()
                return 0
            except BaseException as e:  # noqa: BLE001
                print("loopback all-gather: %r" % (e,))
                self.barrier.abort()
                return 1

        def allreduce(_user, dev, n, stream_ptr):
            return 1

        def alltoallv(_user, send, sb, so, recv, rb, ro, stream_ptr):
            return 1
        cbs = (_lib.TRANSPORT_ALLREDUCE(allreduce),
               _lib.TRANSPORT_ALLGATHER(allgather),
               _lib.TRANSPORT_ALLTOALLV(alltoallv))
        table = _lib.TransportC(*cbs)
        h = C.c_void_p()
        _lib.check(_lib.lib().o3dmi_comm_create_custom(
            C.byref(table), None, rank, world, C.byref(h)),
            "comm_create_custom")
        return Comm(h, keep=(cbs, table))


def _grid_with_frames(geometry, first, n, w, h):
    from open3d_amd import synthetic
    g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                [torch.float32, torch.uint16, torch.uint16],
                                [1, 1, 3], sc.VOXEL, sc.RES, 8192)
    d, c, K, Ts = synthetic.render_frames(first, n, w, h, device="cuda")
    for i in range(n):
        g.integrate_frame(d[i], c[i], K, K, Ts[i], sc.DEPTH_SCALE,
                          sc.DEPTH_MAX, sc.TRUNC_MULT)
    torch.cuda.synchronize()
    return g, d, K, Ts


# (heights are multiples of the range map's down factor, as the reference's
# {h / 8, w / 8, 2} range map requires; 104 rows = 13 tile rows over 3 ranks,
# 24 rows = 3 tile rows over 5 ranks: two ranks without a band)
@pytest.mark.parametrize("world,w,h", [(3, 320, 240), (3, 160, 104),
                                       (5, 320, 24), (4, 640, 480)])
def test_sharded_ray_cast_equals_the_single_rank_maps_bit_for_bit(world, w, h):
    _lib, geometry = _gpu()
    # the grid comes from whole frames; the ray cast renders the top-left
    # w x h pixels of that view (same intrinsics)
    gw, gh = (640, 480) if w > 320 else (320, 240)
    g, d, K, Ts = _grid_with_frames(geometry, 100, 4, gw, gh)
    T = Ts[2]
    keys = g.compute_unique_block_coordinates(d[2], K, T, sc.DEPTH_SCALE,
                                              sc.DEPTH_MAX, sc.TRUNC_MULT)
    attrs = ("depth", "vertex", "color", "normal")
    want = g.ray_cast(keys, K, T, w, h, attrs, sc.DEPTH_SCALE, 0.1,
                      sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT, 8)
    torch.cuda.synchronize()
    assert float((want["depth"] > 0).float().mean()) > 0.2
    # rows of a band alone: o3dmi_vbg_raycast_rows through the operator with
    # one rank is the plain call
    lb = _Loopback(world)
    out = [None] * world

    def rank_main(rank):
        try:
            comm = lb.comm(rank)
            with torch.cuda.stream(torch.cuda.Stream()):
                comm.install()
                try:
                    out[rank] = g.ray_cast(keys, K, T, w, h, attrs,
                                           sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX,
                                           1.0, sc.TRUNC_MULT, 8, sharded=True)
                    torch.cuda.synchronize()
                finally:
                    comm.uninstall()
            comm.destroy()
        except BaseException as ex:  # noqa: BLE001 - reported below
            out[rank] = ex
            lb.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,))
               for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(180)
    for r in range(world):
        assert not isinstance(out[r], BaseException), out[r]
        assert out[r] is not None, "rank %d did not finish" % r
        for a in attrs + ("range",):
            got = out[r][a].contiguous().view(torch.int32)
            exp = want[a].contiguous().view(torch.int32)
            assert torch.equal(got, exp), (r, a)


def test_a_rank_whose_own_stage_fails_does_not_leave_its_peers_waiting():
    """The call is a collective with a rank-local stage in front (arguments,
    scratch, the band's launch): the ranks agree on that stage's status before
    the first all-gather. Rank 1 asks for a width the kernel seam refuses --
    it gets its own error, the others O3DMI_ERR_PEER, promptly, and the same
    communicators then render the maps."""
    _lib, geometry = _gpu()
    world, w, h = 3, 320, 240
    g, d, K, Ts = _grid_with_frames(geometry, 100, 3, w, h)
    T = Ts[1]
    keys = g.compute_unique_block_coordinates(d[1], K, T, sc.DEPTH_SCALE,
                                              sc.DEPTH_MAX, sc.TRUNC_MULT)
    want = g.ray_cast(keys, K, T, w, h, ("depth",), sc.DEPTH_SCALE, 0.1,
                      sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT, 8)
    torch.cuda.synchronize()
    lb = _Loopback(world)
    status = [None] * world
    again = [None] * world

    def rank_main(rank):
        try:
            comm = lb.comm(rank)
            with torch.cuda.stream(torch.cuda.Stream()):
                comm.install()
                try:
                    try:
                        g.ray_cast(keys, K, T, w + 4 if rank == 1 else w, h,
                                   ("depth",), sc.DEPTH_SCALE, 0.1,
                                   sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT, 8,
                                   sharded=True)
                        status[rank] = 0
                    except _lib.O3DMIError as e:
                        status[rank] = (e.status, str(e))
                    again[rank] = g.ray_cast(keys, K, T, w, h, ("depth",),
                                             sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX,
                                             1.0, sc.TRUNC_MULT, 8,
                                             sharded=True)
                    torch.cuda.synchronize()
                finally:
                    comm.uninstall()
            comm.destroy()
        except BaseException as ex:  # noqa: BLE001 - reported below
            again[rank] = ex
            lb.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,))
               for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not any(t.is_alive() for t in threads), "a rank is still waiting"
    assert status[1] is not None and status[1][0] == 1, status  # INVALID_ARG
    for r in (0, 2):
        assert status[r] is not None and status[r][0] == 10, status  # PEER
        assert "rank 1" in status[r][1], status[r]
    for r in range(world):
        assert not isinstance(again[r], BaseException), again[r]
        assert torch.equal(again[r]["depth"].view(torch.int32),
                           want["depth"].view(torch.int32)), r


def _two_process_rank(rank, world):
    """One rank of the two-process sharded ray cast: own process, own HIP
    context, own replica of the grid (the same frames integrated), the maps
    all-gathered through sharding.Comm.for_backend -- RCCL inside the library
    when every rank has its own GPU, else torch.distributed's gloo staged
    through the host."""
    import torch.distributed as dist
    from open3d_amd import geometry
    from open3d_amd.sharding import Comm
    w, h = 320, 240
    g, d, K, Ts = _grid_with_frames(geometry, 100, 3, w, h)
    T = Ts[1]
    keys = g.compute_unique_block_coordinates(d[1], K, T, sc.DEPTH_SCALE,
                                              sc.DEPTH_MAX, sc.TRUNC_MULT)
    attrs = ("depth", "vertex", "color", "normal")
    args = (keys, K, T, w, h, attrs, sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX, 1.0,
            sc.TRUNC_MULT, 8)
    want = g.ray_cast(*args)
    torch.cuda.synchronize()
    comm = Comm.for_backend(dist)
    comm.install()
    try:
        got = g.ray_cast(*args, sharded=True)
        torch.cuda.synchronize()
    finally:
        Comm.uninstall()
        comm.destroy()
    same = all(torch.equal(got[a].view(torch.int32), want[a].view(torch.int32))
               for a in attrs + ("range",))
    import hashlib
    digest = hashlib.sha256(b"".join(
        got[a].cpu().numpy().tobytes() for a in attrs)).hexdigest()
    return (same, float((want["depth"] > 0).float().mean()), digest,
            dist.get_backend())


@pytest.mark.timeout(600)
def test_sharded_ray_cast_two_processes_over_torch_distributed():
    """The same call with real processes and the transport a multi-GPU run
    uses (`bench.py --gpus N` ranks are processes of torch.distributed.run):
    two ranks, each its own process and grid replica; RCCL when the box has
    two GPUs, gloo with both ranks on this GPU otherwise. Each rank's maps
    equal its own single-rank maps bit for bit, and the ranks' maps each
    other's."""
    from test_sharding import _run
    _gpu()
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    got = _run(_two_process_rank, backend=backend)
    for r in range(2):
        same, covered, _, be = got[r]
        assert be == backend
        assert covered > 0.2
        assert same, r
    assert got[0][2] == got[1][2]


def test_band_rows_are_validated():
    _lib, geometry = _gpu()
    g, d, K, Ts = _grid_with_frames(geometry, 100, 1, 64, 48)
    L = _lib.lib()
    # rows that are not whole tiles are refused by the kernel seam
    hm = g.hashmap()
    rng = torch.zeros((6, 8, 2), dtype=torch.float32, device="cuda")
    dep = torch.zeros((48, 64, 1), dtype=torch.float32, device="cuda")
    st = L.o3dmi_vbg_raycast_rows(
        hm._h, _lib.ptr(g.attribute("tsdf")), _lib.ptr(g.attribute("weight")),
        None, _lib.U16, _lib.ptr(rng), _lib.ptr(dep), *([None] * 9),
        _lib.f64p(np.ascontiguousarray(K, np.float64)),
        _lib.f64p(np.ascontiguousarray(Ts[0], np.float64)), 48, 64, 4, 16,
        sc.RES, C.c_float(sc.VOXEL), C.c_float(sc.DEPTH_SCALE),
        C.c_float(0.1), C.c_float(sc.DEPTH_MAX), C.c_float(1.0),
        C.c_float(sc.TRUNC_MULT), 8, None)
    assert st == 1  # O3DMI_ERR_INVALID_ARG
    # an image that is not a multiple of the range map's down factor has no
    # range cell for its last rows (upstream reads past the map): refused
    with pytest.raises(_lib.O3DMIError):
        g.ray_cast(g.compute_unique_block_coordinates(
            d[0], K, Ts[0], sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT),
            K, Ts[0], 64, 44, ("depth",), sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX,
            1.0, sc.TRUNC_MULT, 8)
