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
  * Frame-sharded grids are combined into one model, when one is wanted, by
    merge_frame_sharded_grid: each rank exports its active blocks, one padded
    all-gather of keys and value rows, and every rank folds the other ranks'
    blocks into its own grid (VoxelBlockGrid.merge_blocks: the weighted
    running mean Integrate itself computes, weights added).
"""
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


def merge_frame_sharded_grid(grid, dist):
    """Folds every other rank's blocks into `grid` (ascending rank order).
    Afterwards each rank holds the model of the whole stream: identical block
    sets and weights on every rank, TSDF / colour equal to the single-GPU
    stream up to the rounding of the running mean (a different association of
    the same weighted sum; bit-identical for world = 2 on both ranks)."""
    keys, values = grid.export_blocks()
    rank = dist.get_rank()
    for r, (k, v) in enumerate(allgather_blocks(keys, values, dist)):
        if r != rank and k.shape[0]:
            grid.merge_blocks(k.contiguous(), [x.contiguous() for x in v])
