"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no multi-device execution at all, so
this layer is new. Exchanges are tiny and latency-bound:

  * ICP: the source cloud is split across ranks, the target (and its index) is
    replicated; per iteration one all-reduce of 32 float64 (29 Gauss-Newton
    sums, sum d2, match count, shard size) -- every rank then solves the same
    6x6 system, so no broadcast is needed. make_device_allreduce keeps that
    exchange on the device (RCCL on the launch stream, between the final
    reduction kernel and the mailbox post); make_allreduce_sum is the host
    form (numpy buffer), kept for callers without a device collective.
  * Integration, frame-sharded (independent streams, weak scaling): rank r
    owns frames r, r+N, ... into a private grid; the activated block IDs are
    unioned with one padded all-gather.
  * Integration, block-ownership (ONE stream, strong scaling, bit-parity with
    one GPU): every rank sees every frame and runs the cheap block touch, but
    only activates / integrates the blocks it owns
    (VoxelBlockGrid.set_block_ownership(rank, world)); no data-path
    collective at all, the per-voxel work is split N ways and the union of the
    per-rank grids equals the single-GPU grid bit for bit.
  * Frame-sharded grids are combined into one model by
    merge_frame_sharded_grid -> o3dmi_vbg_merge_frame_sharded: every active
    block travels to the rank that OWNS it (the key hash of the block-ownership
    scheme) with one all-to-all per tensor, and the owner folds the partial
    blocks in (the weighted running mean Integrate itself computes, weights
    added). The ranks end with disjoint grids whose union is the model;
    `replicate=True` all-gathers the finished blocks to every rank.

The collectives themselves live in the C++ library (csrc/host/collectives.cpp):
`Comm.rccl(dist)` builds an RCCL communicator inside the library (the unique
id is broadcast through torch.distributed, whatever its backend), and
`Comm.install()` makes it the ICP drivers' all-reduce -- ncclAllReduce on the
launch stream, no Python in the iteration. `Comm.torch(dist)` is the same
interface over torch.distributed calls (gloo in the CPU tests and when several
ranks share one GPU, where RCCL refuses to run).
"""
import ctypes as C
import threading

import numpy as np
import torch


def shard_range(n, rank, world):
    """Contiguous [begin, end) slice of n items for `rank` (balanced)."""
    base, rem = divmod(n, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def make_allreduce_sum(dist, device):
    """Returns f(np.float64 array) that sums it in place over all ranks; this
    is the `allreduce` hook of registration.icp / multi_scale_icp."""
    def _f(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        a[:] = t.cpu().numpy()
    return _f


def make_device_allreduce(dist):
    """The on-device form of the ICP exchange: returns a callable suitable for
    registration.icp / multi_scale_icp's `device_allreduce` argument. The
    driver hands it the device address of the iteration's 32 float64 sums and
    the launch stream; the all-reduce (RCCL under backend "nccl") is enqueued
    on that stream, in place, between the final reduction kernel and the
    kernel that posts the sums to the host mailbox -- no host staging, no
    numpy round trip. Under "gloo" (CPU tests, single-GPU multi-process runs)
    the buffer is staged through the host, which only serves correctness."""
    from . import _lib
    from .core import tensor_from_ptr

    def _f(dev_ptr, n, stream_ptr):
        t = tensor_from_ptr(dev_ptr, (n,), _lib.F64, None)
        cur = torch.cuda.current_stream().cuda_stream
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream_ptr)) \
            if stream_ptr and stream_ptr != cur else None
        if ctx is not None:
            ctx.__enter__()
        try:
            if dist.get_backend() == "gloo":
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
    return _f


def block_owner(keys, world):
    """Owner rank of each {M,3} int32 block key under block-ownership sharding
    (host mirror of the device-side OwnerOf: murmur3 finaliser of the packed
    21-bit-per-coordinate key, upper half modulo `world`)."""
    k = np.asarray(keys, dtype=np.int64)
    bias = 1 << 20
    packed = ((k[:, 0] + bias).astype(np.uint64) << np.uint64(42)) | \
             ((k[:, 1] + bias).astype(np.uint64) << np.uint64(21)) | \
             (k[:, 2] + bias).astype(np.uint64)
    with np.errstate(over="ignore"):
        packed ^= packed >> np.uint64(33)
        packed *= np.uint64(0xff51afd7ed558ccd)
        packed ^= packed >> np.uint64(33)
        packed *= np.uint64(0xc4ceb9fe1a85ec53)
        packed ^= packed >> np.uint64(33)
    return ((packed >> np.uint64(32)) % np.uint64(world)).astype(np.int32)


def allgather_block_keys(keys, dist):
    """Union of {M_r,3} int32 block-key sets over all ranks -> {U,3} int32,
    lexicographically sorted (identical on every rank)."""
    world = dist.get_world_size()
    dev = keys.device
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    m = int(max(int(x.item()) for x in ns))
    pad = torch.zeros((m, 3), dtype=torch.int32, device=dev)
    pad[:keys.shape[0]] = keys
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    allk = torch.cat([b[:int(c.item())] for b, c in zip(bufs, ns)], 0)
    return torch.unique(allk, dim=0)


def allgather_blocks(keys, values, dist):
    """All-gather of every rank's exported blocks (keys {M_r,3} int32 and the
    per-attribute value tensors {M_r,...}); returns a list over ranks of
    (keys, [values]) trimmed to each rank's count. Padded to the largest
    M_r so that one collective per tensor is enough."""
    world = dist.get_world_size()
    dev = keys.device
    # gloo moves host memory: device tensors are staged through the host there
    # (the CPU tests, and single-GPU multi-process runs); RCCL takes them as is
    wire_dev = torch.device("cpu") if dist.get_backend() == "gloo" else dev
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=wire_dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    counts = [int(x.item()) for x in ns]
    m = max(counts)

    def gather(t):
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype,
                          device=wire_dev)
        pad[:t.shape[0]] = t.to(wire_dev)
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
        return [b.to(dev) for b in bufs]

    # no uint16 collectives in gloo / RCCL: 16-bit rows travel as byte views
    def wire(t):
        return t.contiguous().view(torch.uint8) if t.dtype == torch.uint16 \
            else t

    gk = gather(keys)
    gv = [[b.view(v.dtype) for b in gather(wire(v))] for v in values]
    return [(gk[r][:counts[r]], [g[r][:counts[r]] for g in gv])
            for r in range(world)]


def _backend(dist):
    return dist.get_backend()


def _all_ranks(dist, ok):
    """True iff `ok` is true on every rank (one small all-reduce)."""
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


# The Comm installed on this thread (o3dmi_set_comm): kept alive here until it
# is uninstalled -- the library calls the ctypes thunks of a custom transport
# through it, and a Comm that was garbage-collected while installed would leave
# the driver calling freed callbacks.
_tls = threading.local()  # .installed: the Comm o3dmi_set_comm holds (per thread)


class Comm:
    """o3dmi_comm_t: the library-owned collectives of one rank."""

    def __init__(self, handle, keep=None, transport="custom"):
        self.handle = handle
        self._keep = keep  # callbacks / tables that must outlive the handle
        # what carries the collectives: "rccl" (ncclCommInitRank inside the
        # library), "torch" (torch.distributed calls behind a transport
        # table) or "custom"
        self.transport = transport

    def rccl_ranks(self):
        """ncclCommCount of the communicator behind this handle (asked of
        RCCL, not remembered), 0 when another transport carries it."""
        from . import _lib
        return int(_lib.lib().o3dmi_comm_rccl_ranks(self.handle))

    def self_check(self):
        """One all-gather and one all-reduce through THIS communicator on the
        current stream; returns {"allgather_rank_sum", "allreduce_rank_sum",
        "expected", "ok"}: both must equal 0 + 1 + ... + (world - 1)."""
        from . import _lib
        from .core import stream
        L = _lib.lib()
        world, rank = self.world, self.rank
        send = torch.full((1,), float(rank), dtype=torch.float64,
                          device="cuda")
        recv = torch.full((world,), -1.0, dtype=torch.float64, device="cuda")
        _lib.check(L.o3dmi_comm_allgather(
            self.handle, _lib.ptr(send), _lib.ptr(recv), 8, stream()),
            "comm_allgather")
        red = torch.full((1,), float(rank), dtype=torch.float64,
                         device="cuda")
        _lib.check(L.o3dmi_comm_allreduce_sum_f64(
            self.handle, _lib.ptr(red), 1, stream()), "comm_allreduce")
        torch.cuda.synchronize()
        want = world * (world - 1) // 2
        got_g, got_r = float(recv.sum().item()), float(red.item())
        in_order = bool((recv.cpu() == torch.arange(
            world, dtype=torch.float64)).all())
        return {"allgather_rank_sum": got_g, "allreduce_rank_sum": got_r,
                "expected": want,
                "ok": got_g == want and got_r == want and in_order}

    @property
    def rank(self):
        from . import _lib
        return _lib.lib().o3dmi_comm_rank(self.handle)

    @property
    def world(self):
        from . import _lib
        return _lib.lib().o3dmi_comm_world(self.handle)

    @staticmethod
    def rccl(dist):
        """An RCCL communicator created INSIDE the library on the current
        device (ncclCommInitRank); rank 0's unique id travels through
        `dist` (any backend)."""
        from . import _lib
        L = _lib.lib()
        # Every rank joins this agreement BEFORE any of them enters the
        # broadcast below: a rank without a loadable librccl that
        # raised here on its own would leave the others waiting in a
        # collective it never enters.
        if not _all_ranks(dist, bool(L.o3dmi_rccl_available())):
            raise RuntimeError(
                "RCCL is not available on every rank (%s here)" % (
                    "available" if L.o3dmi_rccl_available()
                    else "librccl not loadable"))
        # Rank 0's failure to make an id must not leave the others waiting in
        # the broadcast: it sends the all-zero id, which every rank rejects.
        ident = torch.zeros(128, dtype=torch.uint8)
        if dist.get_rank() == 0:
            buf = (C.c_char * 128)()
            if L.o3dmi_rccl_unique_id(C.cast(buf, C.c_void_p)) == _lib.OK:
                ident = torch.frombuffer(bytearray(buf.raw),
                                         dtype=torch.uint8)
        on_dev = dist.get_backend() != "gloo"
        t = ident.cuda() if on_dev else ident.clone()
        dist.broadcast(t, 0)
        raw = bytes(t.cpu().numpy().tobytes())
        if not any(raw):
            raise RuntimeError("rank 0 could not create an RCCL unique id")
        h = C.c_void_p()
        _lib.check(L.o3dmi_comm_create_rccl(raw, dist.get_rank(),
                                            dist.get_world_size(),
                                            C.byref(h)), "comm_create_rccl")
        return Comm(h, transport="rccl")

    @staticmethod
    def torch(dist):
        """The same three exchanges through torch.distributed calls (a custom
        transport table). Under "gloo" device buffers are staged through the
        host -- for correctness runs (CPU tests, several ranks on one GPU),
        not for speed."""
        from . import _lib
        from .core import tensor_from_ptr
        world, rank = dist.get_world_size(), dist.get_rank()
        host = dist.get_backend() == "gloo"

        def on_stream(stream_ptr):
            cur = torch.cuda.current_stream().cuda_stream
            if stream_ptr and stream_ptr != cur:
                return torch.cuda.stream(torch.cuda.ExternalStream(stream_ptr))
            return None

        def run(stream_ptr, body):
            ctx = on_stream(stream_ptr)
            try:
                if ctx is not None:
                    ctx.__enter__()
                try:
                    body()
                finally:
                    if ctx is not None:
                        ctx.__exit__(None, None, None)
                return 0
            except Exception as e:  # the C side reports a failed transport
                import sys
                print("open3d_amd.sharding transport: %r" % (e,),
                      file=sys.stderr)
                return 1

        def bytes_view(ptr, n):
            return tensor_from_ptr(ptr, (int(n),), _lib.U8, None)

        def allreduce(_user, dev, n, stream_ptr):
            def body():
                t = tensor_from_ptr(dev, (int(n),), _lib.F64, None)
                if host:
                    h = t.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return run(stream_ptr, body)

        def allgather(_user, send, recv, nbytes, stream_ptr):
            def body():
                s = bytes_view(send, nbytes)
                r = bytes_view(recv, nbytes * world)
                if host:
                    parts = [torch.empty(int(nbytes), dtype=torch.uint8)
                             for _ in range(world)]
                    dist.all_gather(parts, s.cpu())
                    r.copy_(torch.cat(parts))
                else:
                    dist.all_gather_into_tensor(r, s)
            return run(stream_ptr, body)

        def alltoallv(_user, send, sb, so, recv, rb, ro, stream_ptr):
            def body():
                outs = [bytes_view(send + so[p], sb[p]) if sb[p] else
                        torch.empty(0, dtype=torch.uint8, device="cuda")
                        for p in range(world)]
                ins = [bytes_view(recv + ro[p], rb[p]) if rb[p] else
                       torch.empty(0, dtype=torch.uint8, device="cuda")
                       for p in range(world)]
                if host:
                    h_out = [t.cpu() for t in outs]
                    h_in = [torch.empty(int(rb[p]), dtype=torch.uint8)
                            for p in range(world)]
                    # gloo has no all_to_all: one all_gather of everybody's
                    # ranges would move world x the bytes, so pairs exchange
                    # with isend / irecv
                    reqs = []
                    for p in range(world):
                        if p == rank:
                            continue
                        if rb[p]:
                            reqs.append(dist.irecv(h_in[p], src=p))
                        if sb[p]:
                            reqs.append(dist.isend(h_out[p], dst=p))
                    for q in reqs:
                        q.wait()
                    if sb[rank]:
                        h_in[rank].copy_(h_out[rank])
                    for p in range(world):
                        if rb[p]:
                            ins[p].copy_(h_in[p])
                else:
                    dist.all_to_all(ins, outs)
            return run(stream_ptr, body)

        cbs = (_lib.TRANSPORT_ALLREDUCE(allreduce),
               _lib.TRANSPORT_ALLGATHER(allgather),
               _lib.TRANSPORT_ALLTOALLV(alltoallv))
        table = _lib.TransportC(*cbs)
        h = C.c_void_p()
        _lib.check(_lib.lib().o3dmi_comm_create_custom(
            C.byref(table), None, rank, world, C.byref(h)),
            "comm_create_custom")
        return Comm(h, keep=(cbs, table), transport="torch")

    @staticmethod
    def for_backend(dist, allow_fallback=False):
        """RCCL inside the library when torch.distributed itself runs on it,
        else the torch.distributed transport (gloo: the functional dry run).
        On the nccl backend a library communicator that cannot be made on
        EVERY rank (librccl not resolvable by dlopen, ncclCommInitRank refused
        ...) is an ERROR on every rank -- a scaling measurement must not
        degrade quietly to staging through torch.distributed -- unless
        `allow_fallback`, in which case every rank takes the torch transport
        together (the ranks agree on it, so that no rank enters a collective
        of a communicator the others do not have) and `Comm.transport` says
        "torch"."""
        if _backend(dist) != "nccl":
            return Comm.torch(dist)
        comm, why = None, None
        try:
            comm = Comm.rccl(dist)
        except Exception as e:  # noqa: BLE001 - reported below
            why = e
        if _all_ranks(dist, comm is not None):
            return comm
        if comm is not None:
            comm.destroy()
        msg = ("open3d_amd.sharding: the library's RCCL communicator is not "
               "available on every rank (%r here)" % (why,))
        if not allow_fallback:
            raise RuntimeError(
                msg + "; refusing to fall back to the torch.distributed "
                "transport silently (pass allow_fallback=True / bench.py "
                "--allow-transport-fallback to accept it)")
        import sys
        print(msg + "; using the torch.distributed transport", file=sys.stderr)
        return Comm.torch(dist)

    def install(self, level_sharding=False):
        """Makes this communicator the all-reduce of the ICP drivers called
        from this host thread (o3dmi_set_comm). level_sharding: every rank
        passes the WHOLE source cloud and the driver shards each pyramid level
        (reference-identical pyramid); otherwise each rank passes its shard."""
        from . import _lib
        _lib.check(_lib.lib().o3dmi_set_comm(self.handle), "set_comm")
        _tls.installed = self
        _lib.check(_lib.lib().o3dmi_set_icp_level_sharding(
            1 if level_sharding else 0), "set_icp_level_sharding")

    @staticmethod
    def uninstall():
        from . import _lib
        _lib.check(_lib.lib().o3dmi_set_comm(None), "set_comm")
        _tls.installed = None
        _lib.check(_lib.lib().o3dmi_set_icp_level_sharding(0),
                   "set_icp_level_sharding")

    def allreduce_sum(self, t):
        """In-place sum of a float64 device tensor over the ranks."""
        from . import _lib
        from .core import stream
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous()
        _lib.check(_lib.lib().o3dmi_comm_allreduce_sum_f64(
            self.handle, _lib.ptr(t), t.numel(), stream()), "comm_allreduce")

    def destroy(self):
        from . import _lib
        if self.handle:
            if getattr(_tls, "installed", None) is self:
                Comm.uninstall()
            _lib.lib().o3dmi_comm_destroy(self.handle)
            self.handle = None


def merge_frame_sharded_grid(grid, comm, replicate=False):
    """Combines the private grids of frame-sharded ranks into ONE model
    (owner-partitioned exchange, see the module docstring). `comm`: a Comm, or
    a torch.distributed module (a Comm is built for its backend). Block set
    and weights equal the single-stream grid's; TSDF / colour equal it up to
    the rounding of the running mean (another association of the same weighted
    sum). With `replicate` every rank ends with the whole model, bit-identical
    across ranks."""
    own = None
    if not isinstance(comm, Comm):
        own = comm = Comm.for_backend(comm)
    try:
        grid.merge_frame_sharded(comm, replicate=replicate)
    finally:
        if own is not None:
            torch.cuda.synchronize()
            own.destroy()
