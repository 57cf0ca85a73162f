"""GPU tests of VoxelBlockGrid::Save / Load and of
HashMap::Reserve's row gather / scatter, through the C ABI."""
import numpy as np
import pytest
import torch

import _scene as sc
from test_vbg_gpu import OracleGrid, _compare_grids, _mk_grid

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open3d_amd import geometry
    return geometry


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rows_by_key(g):
    hm = g.hashmap()
    act = hm.active_buf_indices().cpu().numpy().astype(np.int64)
    keys = hm.key_tensor().cpu().numpy()[act]
    order = np.lexsort(keys.T[::-1])
    out = {"key": keys[order]}
    for name in g.attr_names:
        out[name] = g.attribute(name).cpu().numpy()[act][order]
    return out


@pytest.mark.parametrize("grid_f32", [False, True])
def test_save_load_round_trip(grid_f32, tmp_path):
    geometry = _gpu()
    g = _mk_grid(geometry, grid_f32, block_count=4096)
    og = OracleGrid(grid_f32, 8192)
    for k in range(0, 12, 3):
        d, c, K, Ts = sc.frames(k, 1, 320, 240)
        og.integrate(d[0], c[0], K, Ts[0])
        g.integrate_frame(_dev(d[0]), _dev(c[0]), K, K, Ts[0])
    n = g.hashmap().size()
    path = str(tmp_path / "grid")          # no extension: ".npz" is appended
    g.save(path)
    z = np.load(path + ".npz")
    assert set(z.files) == {"voxel_size", "block_resolution", "HIP:0",
                            "attr_name_tsdf", "attr_name_weight",
                            "attr_name_color", "key", "value_000", "value_001",
                            "value_002"}
    assert z["voxel_size"].dtype == np.float32 and \
        z["voxel_size"][0] == np.float32(sc.VOXEL)
    assert z["block_resolution"].dtype == np.int64 and \
        z["block_resolution"][0] == sc.RES
    assert z["HIP:0"].shape == () and z["HIP:0"].dtype == np.uint8
    assert [int(z["attr_name_" + a][0]) for a in ("tsdf", "weight", "color")] \
        == [0, 1, 2]
    assert z["key"].shape == (n, 3) and z["key"].dtype == np.int32
    wd = np.float32 if grid_f32 else np.uint16
    assert z["value_000"].shape == (n, 16, 16, 16, 1)
    assert z["value_001"].dtype == wd and z["value_002"].shape[-1] == 3
    rows = _rows_by_key(g)
    order = np.lexsort(z["key"].T[::-1])
    assert np.array_equal(z["key"][order], rows["key"])
    assert z["value_000"][order].tobytes() == rows["tsdf"].tobytes()
    assert z["value_001"][order].tobytes() == rows["weight"].tobytes()
    assert z["value_002"][order].tobytes() == rows["color"].tobytes()

    g2 = geometry.VoxelBlockGrid.load(path + ".npz")
    assert g2.attr_names == ["tsdf", "weight", "color"]
    assert g2.block_resolution == sc.RES and \
        np.float32(g2.voxel_size) == np.float32(sc.VOXEL)
    assert g2.hashmap().size() == n and g2.hashmap().capacity() == n
    rows2 = _rows_by_key(g2)
    for k in rows:
        assert rows2[k].tobytes() == rows[k].tobytes(), k
    assert _compare_grids(og, g2)[1]
    # same surface from the loaded grid (order follows buffer indices)
    a, b = g.extract_point_cloud(3.0), g2.extract_point_cloud(3.0)
    key = lambda p: sc.sort_rows(np.concatenate(
        [p["positions"].cpu().numpy(), p["normals"].cpu().numpy(),
         p["colors"].cpu().numpy()], axis=1))
    assert a["positions"].shape[0] > 1000
    assert np.array_equal(key(a), key(b))

    # the loaded grid is full (capacity == size): integrating further frames
    # makes the map Reserve (export rows, rehash, scatter rows back)
    for k in (12, 15):
        d, c, K, Ts = sc.frames(k, 1, 320, 240)
        og.integrate(d[0], c[0], K, Ts[0])
        g2.integrate_frame(_dev(d[0]), _dev(c[0]), K, K, Ts[0])
    assert g2.hashmap().capacity() > n
    assert _compare_grids(og, g2)[1]


def test_load_a_file_written_elsewhere(tmp_path):
    """Interchange: a file with the entries stock Open3D writes (device
    placeholder "CUDA:0", numpy's own zip64 records / deflate)."""
    geometry = _gpu()
    rng = np.random.default_rng(0)
    n, res = 37, 8
    keys = rng.integers(-20, 20, (200, 3), dtype=np.int32)
    keys = np.unique(keys, axis=0)[:n]
    tsdf = rng.standard_normal((n, res, res, res, 1)).astype(np.float32)
    wgt = rng.integers(0, 50, (n, res, res, res, 1)).astype(np.float32)
    for saver, name in ((np.savez, "a.npz"), (np.savez_compressed, "b.npz")):
        p = str(tmp_path / name)
        saver(p, **{"voxel_size": np.array([0.01], np.float32),
                    "block_resolution": np.array([res], np.int64),
                    "CUDA:0": np.zeros((), np.uint8),
                    "attr_name_tsdf": np.array([0], np.int32),
                    "attr_name_weight": np.array([1], np.int32),
                    "key": keys, "value_000": tsdf, "value_001": wgt})
        g = geometry.VoxelBlockGrid.load(p)
        assert g.attr_names == ["tsdf", "weight"]
        assert g.block_resolution == res
        rows = _rows_by_key(g)
        order = np.lexsort(keys.T[::-1])
        assert np.array_equal(rows["key"], keys[order])
        assert rows["tsdf"].tobytes() == tsdf[order].tobytes()
        assert rows["weight"].tobytes() == wgt[order].tobytes()
    with pytest.raises(RuntimeError, match="Attribute names not found"):
        q = str(tmp_path / "bad.npz")
        np.savez(q, key=keys)
        geometry.VoxelBlockGrid.load(q)


def test_empty_grid_round_trip(tmp_path):
    geometry = _gpu()
    g = _mk_grid(geometry, False, block_count=64)
    p = str(tmp_path / "e.npz")
    g.save(p)
    z = np.load(p)
    assert z["key"].shape == (0, 3) and z["value_000"].shape == (0, 16, 16, 16, 1)
    g2 = geometry.VoxelBlockGrid.load(p)
    assert g2.hashmap().size() == 0


def test_grid_to_device_clone_keeps_blocks_and_keeps_integrating():
    """VoxelBlockGrid::To(device) (HashMap::To, core/hashmap/HashMap.cpp:
    230-255): active keys and voxel rows are gathered, copied device to device
    and inserted into a new map of the same capacity. On a one-GPU box the
    target is the grid's own device (a deep copy: the same gather / copy /
    insert path with a plain instead of a peer copy); with two GPUs the clone
    lands on device 1. The clone equals the source per block key, is
    independent of it, and integrates further frames to the same result."""
    import _scene as sc
    from open3d_amd import geometry
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")

    def mk():
        return geometry.VoxelBlockGrid(
            ["tsdf", "weight", "color"],
            [torch.float32, torch.uint16, torch.uint16], [1, 1, 3], sc.VOXEL,
            sc.RES, 2048)

    def blocks(g):
        hm = g.hashmap()
        act = hm.active_buf_indices().cpu().numpy().astype(np.int64)
        keys = hm.key_tensor().cpu().numpy()[act]
        o = np.lexsort(keys.T[::-1])
        act, keys = act[o], keys[o]
        return (keys, g.attribute("tsdf").cpu().numpy()[act],
                g.attribute("weight").cpu().numpy()[act],
                g.attribute("color").cpu().numpy()[act])

    fr = [sc.frames(k, 1, 320, 240) for k in (0, 8, 16, 24)]
    K = fr[0][2]
    dev = [(torch.from_numpy(f[0][0]).cuda(), torch.from_numpy(f[1][0]).cuda(),
            f[3][0]) for f in fr]
    g = mk()
    for d, c, T in dev[:2]:
        g.integrate_frame(d, c, K, K, T)
    target = 1 if torch.cuda.device_count() >= 2 else 0
    clone = g.to(target)
    a = blocks(g)
    with torch.cuda.device(target):
        b = blocks(clone)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert clone.hashmap().capacity() == g.hashmap().capacity()
    # independent storage: the source moves on, the clone does not
    g.integrate_frame(*dev[2][:2], K, K, dev[2][2])
    with torch.cuda.device(target):
        assert np.array_equal(blocks(clone)[2], b[2])
        # and the clone integrates the same frame to the same grid
        if target == 0:
            clone.integrate_frame(*dev[2][:2], K, K, dev[2][2])
        else:
            clone.integrate_frame(dev[2][0].to("cuda:1"),
                                  dev[2][1].to("cuda:1"), K, K, dev[2][2])
        c = blocks(clone)
    for x, y in zip(blocks(g), c):
        assert np.array_equal(x, y)
