"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no multi-device execution at all, so
this layer is new. Exchanges are tiny and latency-bound:

  * ICP: the source cloud is split across ranks, the target (and its index) is
    replicated; per iteration one all-reduce of 32 float64 (29 Gauss-Newton
    sums, sum d2, match count, shard size) -- every rank then solves the same
    6x6 system, so no broadcast is needed.
  * Integration: frames are sharded across ranks (rank r owns frames r, r+N,
    ...) into private grids; the activated block IDs are unioned with one
    padded all-gather.
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
