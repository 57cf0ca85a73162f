"""One-rank RCCL sanity of the sharding helpers on the GPU box (the 8-GPU run is
the driver's): init_process_group("nccl", world_size=1), then allgather_blocks
(uint16 rows as byte views), allgather_block_keys, merge_frame_sharded_grid and
the ICP all-reduce hook with device tensors. Run from the repository root."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from open3d_amd import geometry, synthetic
from open3d_amd.sharding import allgather_blocks, allgather_block_keys, merge_frame_sharded_grid, make_allreduce_sum
K = synthetic.intrinsics(320, 240)
g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"], [torch.float32, torch.uint16, torch.uint16], [1, 1, 3], 0.008, 16, 4096)
for k in range(3):
    d, c, _, T = synthetic.render_frames(k, 1, 320, 240, device="cuda")
    g.integrate_frame(d[0].contiguous(), c[0].contiguous(), K, K, T[0])
keys, vals = g.export_blocks()
out = allgather_blocks(keys, vals, dist)
assert len(out) == 1 and torch.equal(out[0][0], keys)
for a, b in zip(out[0][1], vals):
    assert a.dtype == b.dtype and np.array_equal(a.cpu().numpy(), b.cpu().numpy())
u = allgather_block_keys(keys, dist)
assert u.shape[0] == keys.shape[0]
merge_frame_sharded_grid(g, dist)
a = np.arange(32, dtype=np.float64); make_allreduce_sum(dist, torch.device("cuda"))(a)
assert np.array_equal(a, np.arange(32))
torch.cuda.synchronize(); dist.destroy_process_group()
print("rccl one-rank ok: %d blocks" % keys.shape[0])
