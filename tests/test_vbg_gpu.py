"""GPU parity tests of the VoxelBlockGrid path: HIP (through the C ABI) vs the
CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): activated block sets bit-exact; TSDF within
1e-4 (weights / u16 colours are compared exactly -- the arithmetic is the
same float32 sequence, so in practice everything is bit-identical)."""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle as orc
import _scene as sc

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open3d_amd import _lib, geometry
    return _lib, geometry


def _mk_grid(geometry, grid_f32, block_count=4096, with_color=True, res=sc.RES):
    wd = torch.float32 if grid_f32 else torch.uint16
    names = ["tsdf", "weight"] + (["color"] if with_color else [])
    dts = [torch.float32, wd] + ([wd] if with_color else [])
    ch = [1, 1] + ([3] if with_color else [])
    return geometry.VoxelBlockGrid(names, dts, ch, voxel_size=sc.VOXEL,
                                   block_resolution=res,
                                   block_count=block_count)


class OracleGrid:
    """Oracle-side VoxelBlockGrid: hash + numpy value buffers."""

    def __init__(self, grid_f32, capacity, with_color=True, res=sc.RES):
        self.h = orc.HashMap(capacity)
        wd = np.float32 if grid_f32 else np.uint16
        self.res = res
        self.tsdf = np.zeros((capacity, res, res, res), np.float32)
        self.weight = np.zeros((capacity, res, res, res), wd)
        self.color = np.zeros((capacity, res, res, res, 3), wd) \
            if with_color else None

    def integrate(self, depth, color, K, T, keys=None):
        if keys is None:
            keys = orc.depth_touch(depth, K, T, self.res, sc.VOXEL,
                                   sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                                   sc.DEPTH_MAX, 4)
        self.h.activate(keys)
        buf, m = self.h.find(keys)
        assert m.all()
        orc.integrate(depth, color, buf, self.h.key_buffer(), self.tsdf,
                      self.weight, self.color, K, K, T, self.res, sc.VOXEL,
                      sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE, sc.DEPTH_MAX)
        return keys


def _compare_grids(og, g, tsdf_tol=1e-4):
    """Per-key comparison (buffer indices differ between implementations)."""
    hm = g.hashmap()
    n = og.h.size()
    assert hm.size() == n
    okeys = og.h.key_buffer()[:n].copy()
    obuf, _ = og.h.find(okeys)
    gbuf, gm = hm.find(torch.from_numpy(okeys).cuda())
    assert bool(gm.all())
    gbuf = gbuf.cpu().numpy().astype(np.int64)
    # bit-exact activation set: same keys, nothing extra
    gkeys = hm.key_tensor().cpu().numpy()[hm.active_buf_indices().cpu().numpy()]
    assert np.array_equal(sc.sort_rows(gkeys), sc.sort_rows(okeys))
    t = g.attribute("tsdf").cpu().numpy()[gbuf][..., 0]
    w = g.attribute("weight").cpu().numpy()[gbuf][..., 0]
    assert np.array_equal(w, og.weight[obuf])
    err = np.abs(t - og.tsdf[obuf]).max()
    assert err <= tsdf_tol, err
    if og.color is not None:
        c = g.attribute("color").cpu().numpy()[gbuf]
        if og.color.dtype == np.uint16:
            assert np.array_equal(c, og.color[obuf])
        else:
            assert np.abs(c - og.color[obuf]).max() <= 1e-2
    return float(err), bool(np.array_equal(t, og.tsdf[obuf]))


def test_extension_loaded_and_device():
    _lib, _ = _gpu()
    name = C.create_string_buffer(64)
    cus = C.c_int(0)
    mem = C.c_int64(0)
    _lib.check(_lib.lib().o3dmi_device_info(name, 64, C.byref(cus),
                                            C.byref(mem)), "device_info")
    assert name.value.decode().startswith("gfx950"), name.value
    assert cus.value == 256


def test_hash_semantics():
    """cpp/tests/core/HashMap.cpp:138-191 style: duplicates -> one success;
    Find; Erase; GetActiveIndices; Reserve keeps key -> value association."""
    _lib, _ = _gpu()
    from open3d_amd.core import stream
    L = _lib.lib()
    h = C.c_void_p()
    ds = (C.c_int64 * 1)(4)
    _lib.check(L.o3dmi_hash_create(8, 1, ds, stream(), C.byref(h)), "create")
    keys = torch.tensor([[1, 2, 3], [1, 2, 3], [-1, 0, 5], [1, 2, 3],
                         [7, 7, 7]], dtype=torch.int32, device="cuda")
    vals = torch.tensor([10, 11, 12, 13, 14], dtype=torch.int32, device="cuda")
    buf = torch.zeros(5, dtype=torch.int32, device="cuda")
    m = torch.zeros(5, dtype=torch.bool, device="cuda")
    vp = (C.c_void_p * 1)(vals.data_ptr())
    _lib.check(L.o3dmi_hash_insert(h, _lib.ptr(keys), vp, 5, _lib.ptr(buf),
                                   _lib.ptr(m), stream()), "insert")
    n = C.c_int64(0)
    _lib.check(L.o3dmi_hash_size(h, stream(), C.byref(n)), "size")
    assert n.value == 3 and int(m.sum()) == 3
    mk = m.cpu().numpy()
    assert mk[2] and mk[4] and mk[[0, 1, 3]].sum() == 1
    buf2 = torch.zeros(5, dtype=torch.int32, device="cuda")
    m2 = torch.zeros(5, dtype=torch.bool, device="cuda")
    _lib.check(L.o3dmi_hash_find(h, _lib.ptr(keys), 5, None, _lib.ptr(buf2),
                                 _lib.ptr(m2), stream()), "find")
    assert bool(m2.all())
    b2 = buf2.cpu().numpy()
    assert b2[0] == b2[1] == b2[3] and len(set(b2.tolist())) == 3
    # reserve (rehash) keeps values attached to keys
    _lib.check(L.o3dmi_hash_reserve(h, 64, stream()), "reserve")
    assert L.o3dmi_hash_capacity(h) == 64
    _lib.check(L.o3dmi_hash_find(h, _lib.ptr(keys), 5, None, _lib.ptr(buf2),
                                 _lib.ptr(m2), stream()), "find")
    assert bool(m2.all())
    from open3d_amd.core import tensor_from_ptr
    vb = tensor_from_ptr(L.o3dmi_hash_value_buffer(h, 0), (64,), _lib.I32,
                         None).cpu().numpy()
    got = vb[buf2.cpu().numpy()]
    assert got[2] == 12 and got[4] == 14 and got[0] in (10, 11, 13)
    # erase
    ek = keys[2:3].contiguous()
    em = torch.zeros(1, dtype=torch.bool, device="cuda")
    _lib.check(L.o3dmi_hash_erase(h, _lib.ptr(ek), 1, _lib.ptr(em), stream()),
               "erase")
    _lib.check(L.o3dmi_hash_size(h, stream(), C.byref(n)), "size")
    assert bool(em[0]) and n.value == 2
    _lib.check(L.o3dmi_hash_find(h, _lib.ptr(keys), 5, None, _lib.ptr(buf2),
                                 _lib.ptr(m2), stream()), "find")
    assert m2.cpu().numpy().tolist() == [True, True, False, True, True]
    act = torch.zeros(64, dtype=torch.int32, device="cuda")
    _lib.check(L.o3dmi_hash_active_indices(h, _lib.ptr(act), stream(),
                                           C.byref(n)), "active")
    assert n.value == 2
    # out-of-range key is reported, not silently dropped
    bad = torch.tensor([[1 << 21, 0, 0]], dtype=torch.int32, device="cuda")
    _lib.check(L.o3dmi_hash_activate(h, _lib.ptr(bad), 1, None, None, None,
                                     stream()), "activate")
    assert L.o3dmi_hash_size(h, stream(), C.byref(n)) == 4  # KEY_RANGE
    L.o3dmi_hash_destroy(h)


@pytest.mark.timeout(300)
def test_hash_insert_erase_churn_reuses_tombstones():
    """HashMap supports unbounded insert / erase cycling (cpp/tests/core/
    HashMap.cpp:138-191 erases and re-inserts on one map). 200 rounds of
    insert-48 / erase-48 on a 64-block map (128 slots) leave far more
    tombstones behind than there are slots unless inserts reuse them and the
    table is rebuilt when crowded; lookups of absent keys must still end."""
    _lib, _ = _gpu()
    from open3d_amd.core import stream
    L = _lib.lib()
    h = C.c_void_p()
    ds = (C.c_int64 * 1)(4)
    _lib.check(L.o3dmi_hash_create(64, 1, ds, stream(), C.byref(h)), "create")
    rng = np.random.default_rng(5)
    n = C.c_int64(0)
    keep = torch.tensor([[5, 5, 5], [-9, 1, 2]], dtype=torch.int32,
                        device="cuda")
    _lib.check(L.o3dmi_hash_activate(h, _lib.ptr(keep), 2, None, None, None,
                                     stream()), "activate")
    kb = torch.zeros(2, dtype=torch.int32, device="cuda")
    km = torch.zeros(2, dtype=torch.bool, device="cuda")
    _lib.check(L.o3dmi_hash_find(h, _lib.ptr(keep), 2, None, _lib.ptr(kb),
                                 _lib.ptr(km), stream()), "find")
    keep_idx = kb.cpu().numpy().copy()
    for r in range(200):
        k = rng.integers(-1000, 1000, size=(48, 3)).astype(np.int32)
        k = np.unique(k, axis=0)
        kt = torch.from_numpy(k).cuda()
        m = torch.zeros(len(k), dtype=torch.bool, device="cuda")
        b = torch.zeros(len(k), dtype=torch.int32, device="cuda")
        _lib.check(L.o3dmi_hash_activate(h, _lib.ptr(kt), len(k), None,
                                         _lib.ptr(b), _lib.ptr(m), stream()),
                   "activate")
        assert bool(m.all())
        _lib.check(L.o3dmi_hash_size(h, stream(), C.byref(n)), "size")
        assert n.value == len(k) + 2
        # distinct buffer indices, none of them the two long-lived ones
        bi = b.cpu().numpy()
        assert len(set(bi.tolist())) == len(k)
        assert not set(bi.tolist()) & set(keep_idx.tolist())
        absent = torch.from_numpy(
            rng.integers(2000, 3000, size=(64, 3)).astype(np.int32)).cuda()
        am = torch.ones(64, dtype=torch.bool, device="cuda")
        ab = torch.zeros(64, dtype=torch.int32, device="cuda")
        _lib.check(L.o3dmi_hash_find(h, _lib.ptr(absent), 64, None,
                                     _lib.ptr(ab), _lib.ptr(am), stream()),
                   "find")
        assert not bool(am.any())
        em = torch.zeros(len(k), dtype=torch.bool, device="cuda")
        _lib.check(L.o3dmi_hash_erase(h, _lib.ptr(kt), len(k), _lib.ptr(em),
                                      stream()), "erase")
        assert bool(em.all())
        _lib.check(L.o3dmi_hash_find(h, _lib.ptr(kt), len(k), None,
                                     _lib.ptr(b), _lib.ptr(m), stream()),
                   "find")
        assert not bool(m.any())
    _lib.check(L.o3dmi_hash_size(h, stream(), C.byref(n)), "size")
    assert n.value == 2
    # the long-lived keys kept their buffer indices through every rebuild
    _lib.check(L.o3dmi_hash_find(h, _lib.ptr(keep), 2, None, _lib.ptr(kb),
                                 _lib.ptr(km), stream()), "find")
    assert bool(km.all()) and np.array_equal(kb.cpu().numpy(), keep_idx)
    L.o3dmi_hash_destroy(h)


@pytest.mark.parametrize("k", [0, 300, 700])
@pytest.mark.parametrize("f32", [False, True])
def test_depth_touch_block_set_bit_exact(k, f32):
    _lib, geometry = _gpu()
    d, c, K, Ts = sc.frames(k, 1)
    depth = d[0].astype(np.float32) if f32 else d[0]
    want = orc.depth_touch(depth, K, Ts[0], sc.RES, sc.VOXEL,
                           sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, 4)
    g = _mk_grid(geometry, False)
    got = g.compute_unique_block_coordinates(
        torch.from_numpy(depth).cuda(), K, Ts[0], sc.DEPTH_SCALE, sc.DEPTH_MAX,
        sc.TRUNC_MULT).cpu().numpy()
    assert got.shape[0] == want.shape[0] > 100
    assert np.array_equal(sc.sort_rows(got), want)
    # no duplicates
    assert len({tuple(r) for r in got.tolist()}) == got.shape[0]


def test_depth_touch_no_blocks_is_an_error():
    _lib, geometry = _gpu()
    _, _, K, Ts = sc.frames(0, 1)
    g = _mk_grid(geometry, False)
    with pytest.raises(_lib.O3DMIError) as e:
        g.compute_unique_block_coordinates(
            torch.zeros((480, 640), dtype=torch.uint16, device="cuda"), K,
            Ts[0])
    assert "No block is touched" in str(e.value)


def test_unproject_is_the_row_major_scan():
    """o3dmi_unproject compacts in ONE launch and in pixel order: the points
    come out in the row-major scan order of the strided pixels (the oracle's
    order; the reference's atomic-counter order is unspecified), identical
    from run to run, with no sort in the comparison. (Rounds 1-5: an atomic
    counter's arrival order by default, this order behind a switch as three
    launches.)"""
    _lib, geometry = _gpu()
    from open3d_amd.core import stream
    L = _lib.lib()
    d, c, K, Ts = sc.frames(33, 2)
    for f, stride in ((0, 1), (1, 1), (0, 4), (1, 3)):
        cf = (c[f].astype(np.float32) / 255.0).astype(np.float32)
        want_p, want_c = orc.unproject(d[f], cf, K, Ts[f], sc.DEPTH_SCALE,
                                       sc.DEPTH_MAX, stride)
        n = (480 // stride) * (640 // stride)
        runs = []
        for _ in range(2):
            pts = torch.full((n, 3), -7.0, dtype=torch.float32, device="cuda")
            cols = torch.full((n, 3), -7.0, dtype=torch.float32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            _lib.check(L.o3dmi_unproject(
                _lib.ptr(torch.from_numpy(d[f]).cuda()), _lib.U16, 480, 640,
                _lib.ptr(torch.from_numpy(cf).cuda()), _lib.ptr(pts),
                _lib.ptr(cols), _lib.ptr(cnt), _lib.f64p(K), _lib.f64p(Ts[f]),
                C.c_float(sc.DEPTH_SCALE), C.c_float(sc.DEPTH_MAX), stride,
                stream()), "unproject")
            m = int(cnt.item())
            assert m == want_p.shape[0]
            runs.append((pts.cpu().numpy(), cols.cpu().numpy()))
        assert np.array_equal(runs[0][0][:m], want_p)
        assert np.array_equal(runs[0][1][:m], want_c)
        assert np.array_equal(runs[0][0], runs[1][0])
        assert np.array_equal(runs[0][1], runs[1][1])
        assert (runs[0][0][m:] == -7.0).all()       # nothing written past count


def test_unproject_pair_is_two_unprojects():
    """o3dmi_unproject_pair (the model frame's and the camera frame's cloud of
    a tracking step in one launch) gives, for each cloud, o3dmi_unproject's
    points, attributes, count and order to the bit -- float32 depth with an
    attribute image beside uint16 depth without one, two extrinsics, VGA and
    720p, strides 1 / 2 / 3 -- and writes nothing past the counts."""
    _lib, geometry = _gpu()
    from open3d_amd import synthetic as syn
    from open3d_amd.core import stream
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for (W, H) in ((640, 480), (1280, 720)):
        K = syn.intrinsics(W, H)
        d, c, _n, T = syn.render_frames(3, 2, W, H, device="cuda")
        da = (d[0].to(torch.float32)).contiguous()      # depth_scale units
        holes = torch.from_numpy(rng.random((H, W)) < 0.25).cuda()
        da[holes] = 0.0
        da[H // 3:H // 2, :] = 0.0
        attr = torch.from_numpy(
            rng.random((H, W, 3)).astype(np.float32)).cuda()
        db = d[1].contiguous()
        Ta = np.ascontiguousarray(T[0], np.float64)
        Tb = np.ascontiguousarray(np.eye(4), np.float64)
        for stride in (1, 2, 3):
            n = (H // stride) * (W // stride)

            def single(depth, dt, img, Tx):
                pts = torch.full((n, 3), -7.0, dtype=torch.float32,
                                 device="cuda")
                cols = torch.full((n, 3), -7.0, dtype=torch.float32,
                                  device="cuda")
                cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
                _lib.check(L.o3dmi_unproject(
                    _lib.ptr(depth), dt, H, W,
                    _lib.ptr(img) if img is not None else None, _lib.ptr(pts),
                    _lib.ptr(cols) if img is not None else None,
                    _lib.ptr(cnt), _lib.f64p(K), _lib.f64p(Tx),
                    C.c_float(sc.DEPTH_SCALE), C.c_float(sc.DEPTH_MAX),
                    stride, stream()), "unproject")
                return pts.cpu().numpy(), cols.cpu().numpy(), int(cnt.item())

            wa = single(da, _lib.F32, attr, Ta)
            wb = single(db, _lib.U16, None, Tb)
            assert 0 < wa[2] < n and 0 < wb[2] <= n
            for _ in range(2):
                pa = torch.full((n, 3), -7.0, dtype=torch.float32,
                                device="cuda")
                ca = torch.full((n, 3), -7.0, dtype=torch.float32,
                                device="cuda")
                pb = torch.full((n, 3), -7.0, dtype=torch.float32,
                                device="cuda")
                na = torch.zeros(1, dtype=torch.int32, device="cuda")
                nb = torch.zeros(1, dtype=torch.int32, device="cuda")
                _lib.check(L.o3dmi_unproject_pair(
                    _lib.ptr(da), _lib.F32, _lib.ptr(attr), _lib.ptr(pa),
                    _lib.ptr(ca), _lib.ptr(na), _lib.f64p(Ta),
                    _lib.ptr(db), _lib.U16, None, _lib.ptr(pb), None,
                    _lib.ptr(nb), _lib.f64p(Tb), H, W, _lib.f64p(K),
                    C.c_float(sc.DEPTH_SCALE), C.c_float(sc.DEPTH_MAX),
                    stride, stream()), "unproject_pair")
                assert (int(na.item()), int(nb.item())) == (wa[2], wb[2])
                assert np.array_equal(pa.cpu().numpy(), wa[0])
                assert np.array_equal(ca.cpu().numpy(), wa[1])
                assert np.array_equal(pb.cpu().numpy(), wb[0])
    # refusals: aliased outputs, a bad dtype
    pts = torch.zeros((480 * 640, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
    K = syn.intrinsics(640, 480)
    E = np.ascontiguousarray(np.eye(4), np.float64)
    dd = torch.zeros((480, 640), dtype=torch.uint16, device="cuda")
    assert L.o3dmi_unproject_pair(
        _lib.ptr(dd), _lib.U16, None, _lib.ptr(pts), None, _lib.ptr(cnt),
        _lib.f64p(E), _lib.ptr(dd), _lib.U16, None, _lib.ptr(pts), None,
        _lib.ptr(cnt[1:]), _lib.f64p(E), 480, 640, _lib.f64p(K),
        C.c_float(1000.0), C.c_float(3.0), 1, stream()) != 0
    assert L.o3dmi_unproject_pair(
        _lib.ptr(dd), _lib.F64, None, _lib.ptr(pts), None, _lib.ptr(cnt),
        _lib.f64p(E), _lib.ptr(dd), _lib.U16, None, _lib.ptr(pts[1000:]),
        None, _lib.ptr(cnt[1:]), _lib.f64p(E), 480, 640, _lib.f64p(K),
        C.c_float(1000.0), C.c_float(3.0), 1, stream()) != 0


def test_unproject_large_image_many_chunks():
    """A 1280x720 image at stride 1 (450 chunks of 2048 pixels: a workgroup
    reads up to two words per lane of the chunks before it) and at stride 3
    (a size that is no multiple of the chunk), float32 depth, holes of invalid
    depth of every size: row-major order, exact count, nothing written past
    it, twice the same bits."""
    _lib, geometry = _gpu()
    from open3d_amd import synthetic as syn
    from open3d_amd.core import stream
    L = _lib.lib()
    K = syn.intrinsics(1280, 720)
    d, _c, _n, T = syn.render_frames(7, 1, 1280, 720, device="cuda")
    depth = (d[0].to(torch.float32) / 1000.0).contiguous()
    rng = np.random.default_rng(5)
    holes = torch.from_numpy(rng.random((720, 1280)) < 0.3).cuda()
    depth[holes] = 0.0
    depth[100:300, :] = 0.0           # whole chunks without a point
    depth[500:, 640:] = 9.0           # beyond depth_max
    Tn = np.asarray(T[0], np.float64)
    dn = depth.cpu().numpy()
    for stride in (1, 3):
        want_p, _ = orc.unproject(dn, None, K, Tn, 1.0, 3.0, stride)
        n = (720 // stride) * (1280 // stride)
        runs = []
        for _ in range(2):
            pts = torch.full((n, 3), -7.0, dtype=torch.float32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            _lib.check(L.o3dmi_unproject(
                _lib.ptr(depth), _lib.F32, 720, 1280, None, _lib.ptr(pts),
                None, _lib.ptr(cnt), _lib.f64p(K), _lib.f64p(Tn),
                C.c_float(1.0), C.c_float(3.0), stride, stream()),
                "unproject")
            m = int(cnt.item())
            assert m == want_p.shape[0] and 1000 < m < n
            runs.append(pts.cpu().numpy())
        assert np.array_equal(runs[0][:m], want_p)
        assert np.array_equal(runs[0], runs[1])
        assert (runs[0][m:] == -7.0).all()


@pytest.mark.parametrize("input_f32", [False, True])
@pytest.mark.parametrize("grid_f32", [False, True])
def test_integrate_parity_all_dtype_combos(input_f32, grid_f32):
    """The four instantiations of VoxelBlockGridCPU.cpp:212-218, 3 frames."""
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, grid_f32)
    og = OracleGrid(grid_f32, 4096)
    for k in (0, 40, 80):
        d, c, K, Ts = sc.frames(k, 1)
        depth, color = d[0], c[0]
        if input_f32:
            depth, color = sc.as_f32_inputs(depth, color)
        keys = og.integrate(depth, color, K, Ts[0])
        g.integrate(torch.from_numpy(keys).cuda(),
                    torch.from_numpy(depth).cuda(),
                    torch.from_numpy(color).cuda(), K, K, Ts[0],
                    sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
    err, exact = _compare_grids(og, g)
    assert exact, "TSDF not bit-identical (max err %g)" % err


def test_integrate_depth_only_and_res8():
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, False, with_color=False, res=8)
    og = OracleGrid(False, 4096, with_color=False, res=8)
    for k in (10, 20):
        d, c, K, Ts = sc.frames(k, 1)
        keys = og.integrate(d[0], None, K, Ts[0])
        g.integrate(torch.from_numpy(keys).cuda(),
                    torch.from_numpy(d[0]).cuda(), None, K, K, Ts[0],
                    sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
    _compare_grids(og, g)


def test_integrate_scalar_kernel_odd_resolution():
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, True, res=6, block_count=8192)
    og = OracleGrid(True, 8192, res=6)
    d, c, K, Ts = sc.frames(5, 1)
    keys = og.integrate(d[0], c[0], K, Ts[0])
    g.integrate(torch.from_numpy(keys).cuda(), torch.from_numpy(d[0]).cuda(),
                torch.from_numpy(c[0]).cuda(), K, K, Ts[0], sc.DEPTH_SCALE,
                sc.DEPTH_MAX, sc.TRUNC_MULT)
    _compare_grids(og, g)


def test_integrate_different_color_intrinsics_and_size():
    """Colour camera with its own intrinsics / resolution (320x240)."""
    _lib, geometry = _gpu()
    from open3d_amd import synthetic as syn
    d, c, K, Ts = sc.frames(60, 1)
    d2, c2, K2, _ = sc.frames(60, 1, 320, 240)
    g = _mk_grid(geometry, False)
    og = OracleGrid(False, 4096)
    keys = orc.depth_touch(d[0], K, Ts[0], sc.RES, sc.VOXEL,
                           sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, 4)
    og.h.activate(keys)
    buf, _ = og.h.find(keys)
    orc.integrate(d[0], c2[0], buf, og.h.key_buffer(), og.tsdf, og.weight,
                  og.color, K, K2, Ts[0], sc.RES, sc.VOXEL,
                  sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE, sc.DEPTH_MAX)
    g.integrate(torch.from_numpy(keys).cuda(), torch.from_numpy(d[0]).cuda(),
                torch.from_numpy(c2[0]).cuda(), K, K2, Ts[0], sc.DEPTH_SCALE,
                sc.DEPTH_MAX, sc.TRUNC_MULT)
    _compare_grids(og, g)


def test_frame_stream_fast_path_equals_two_step_path():
    """integrate_frame (device-resident counts, fused touch+activate) vs
    compute_unique_block_coordinates + integrate vs the oracle; 6 frames,
    small initial capacity to force a Reserve (rehash) on the way."""
    _lib, geometry = _gpu()
    ga = _mk_grid(geometry, False, block_count=512)
    gb = _mk_grid(geometry, False, block_count=512)
    og = OracleGrid(False, 8192)
    for k in range(100, 160, 10):
        d, c, K, Ts = sc.frames(k, 1)
        dt, ct = torch.from_numpy(d[0]).cuda(), torch.from_numpy(c[0]).cuda()
        og.integrate(d[0], c[0], K, Ts[0])
        ga.integrate_frame(dt, ct, K, K, Ts[0], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                           sc.TRUNC_MULT)
        keys = gb.compute_unique_block_coordinates(dt, K, Ts[0], sc.DEPTH_SCALE,
                                                   sc.DEPTH_MAX, sc.TRUNC_MULT)
        gb.integrate(keys, dt, ct, K, K, Ts[0], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                     sc.TRUNC_MULT)
    assert _compare_grids(og, ga)[1]
    assert _compare_grids(og, gb)[1]
    assert gb.hashmap().capacity() > 512  # grew through Reserve


@pytest.mark.parametrize("group", [1, 2, 3, 4, 6, 8, 11, 16])
@pytest.mark.parametrize("grid_f32", [False, True])
def test_frame_batch_equals_oracle(group, grid_f32):
    """integrate_frames (one native call; `group` frames applied per launch to
    register-resident blocks, next group's touch work in the same launch) vs
    the frame-by-frame oracle, bit-exact, 12 frames, Reserve forced by a small
    capacity, mixed with a two-step-API frame in the middle (capacity
    bookkeeping hand-over)."""
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, grid_f32, block_count=600)
    og = OracleGrid(grid_f32, 16384)
    ks = list(range(200, 320, 10))
    ds, cs, Ts = [], [], []
    for k in ks:
        d, c, K, T = sc.frames(k, 1)
        ds.append(d[0]); cs.append(c[0]); Ts.append(T[0])
    dt = [torch.from_numpy(d).cuda() for d in ds]
    ct = [torch.from_numpy(c).cuda() for c in cs]
    for i in range(len(ks)):
        og.integrate(ds[i], cs[i], K, Ts[i])
    g.integrate_frames(dt[:5], ct[:5], K, K, Ts[:5], sc.DEPTH_SCALE,
                       sc.DEPTH_MAX, sc.TRUNC_MULT, frames_per_launch=group)
    keys = g.compute_unique_block_coordinates(dt[5], K, Ts[5], sc.DEPTH_SCALE,
                                              sc.DEPTH_MAX, sc.TRUNC_MULT)
    g.integrate(keys, dt[5], ct[5], K, K, Ts[5], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                sc.TRUNC_MULT)
    g.integrate_frames(dt[6:], ct[6:], K, K, Ts[6:], sc.DEPTH_SCALE,
                       sc.DEPTH_MAX, sc.TRUNC_MULT, frames_per_launch=group)
    assert _compare_grids(og, g)[1]
    assert g.hashmap().capacity() > 600


def _stream_frames(ks, w=None, h=None):
    ds, cs, Ts, K = [], [], [], None
    for k in ks:
        d, c, K, T = sc.frames(k, 1) if w is None else sc.frames(k, 1, w, h)
        ds.append(d[0]); cs.append(c[0]); Ts.append(T[0])
    dt = [torch.from_numpy(d).cuda() for d in ds]
    ct = [torch.from_numpy(c).cuda() for c in cs]
    return ds, cs, dt, ct, Ts, K


def test_division_forms_are_proven_asynchronously():
    """The short division forms of the integrate role are switched on by an
    on-device proof that runs beside the stream (vbg_stream.hip
    VerifyFastDivision): asking with wait = 1 returns the forms in use from
    then on -- all three verify on gfx950."""
    _lib, geometry = _gpu()
    L = _lib.lib()
    forms = L.o3dmi_vbg_division_forms(C.c_float(sc.VOXEL),
                                       C.c_float(sc.TRUNC_MULT), 1)
    assert forms in (2, 3), forms
    assert L.o3dmi_vbg_division_forms(C.c_float(sc.VOXEL),
                                      C.c_float(sc.TRUNC_MULT), 0) == forms


@pytest.mark.parametrize("group", [4, 12])
def test_run_ahead_on_the_estimate_needs_no_reserve(group):
    """Capacity policy (HashMap.cpp:166-176 without a per-frame sync): a map
    far too small for the frustum bound of a group in flight (9 353 blocks per
    VGA frame) but large enough for what the frames really add is integrated
    with groups issued on the ESTIMATE -- bit-identical to the oracle and
    without a Reserve (the capacity the caller chose stays)."""
    _lib, geometry = _gpu()
    ds, cs, dt, ct, Ts, K = _stream_frames(list(range(100, 340, 10)))
    og = OracleGrid(False, 16384)
    for i in range(len(ds)):
        og.integrate(ds[i], cs[i], K, Ts[i])
    need = og.h.size()
    cap = need + 1500
    assert cap < 9353 * 2  # the strict bound of even one frame pair fails
    g = _mk_grid(geometry, False, block_count=cap)
    g.integrate_frames(dt, ct, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                       sc.TRUNC_MULT, frames_per_launch=group)
    assert _compare_grids(og, g)[1]
    assert g.hashmap().capacity() == cap
    # and once more over the same frames (no block is new any more)
    for i in range(len(ds)):
        og.integrate(ds[i], cs[i], K, Ts[i])
    g.integrate_frames(dt, ct, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                       sc.TRUNC_MULT, frames_per_launch=group)
    assert _compare_grids(og, g)[1]
    assert g.hashmap().capacity() == cap


@pytest.mark.parametrize("group,slack", [(1, 40), (3, 5), (4, 200), (8, 1),
                                         (12, 300), (16, 64)])
def test_group_that_overflows_the_map_is_dropped_and_replayed(group, slack):
    """A group issued on the estimate that runs out of buffer indices is
    dropped on the device as a whole, with every group behind it; the host
    reserves (max(wanted, 2 x capacity), the reference's growth rule) and
    replays from the dropped group's first frame: the grid equals the
    frame-by-frame oracle bit for bit wherever in the stream the map fills
    up."""
    _lib, geometry = _gpu()
    ks = [(i * 53) % 600 for i in range(30)]  # a view that jumps: new blocks
    ds, cs, dt, ct, Ts, K = _stream_frames(ks, 320, 240)
    og = OracleGrid(False, 32768)
    sizes = []
    for i in range(len(ds)):
        og.integrate(ds[i], cs[i], K, Ts[i])
        sizes.append(og.h.size())
    cap = sizes[-1] - 400 + slack  # fills up in the course of the call
    assert sizes[0] < cap < sizes[-1]
    g = _mk_grid(geometry, False, block_count=cap)
    g.integrate_frames(dt, ct, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                       sc.TRUNC_MULT, frames_per_launch=group)
    assert _compare_grids(og, g)[1]
    assert g.hashmap().capacity() >= 2 * cap
    # the map is a normal map afterwards: two-step API on top of it
    d, c, K2, T = sc.frames(700, 1, 320, 240)
    og.integrate(d[0], c[0], K, T[0])
    keys = g.compute_unique_block_coordinates(torch.from_numpy(d[0]).cuda(), K,
                                              T[0], sc.DEPTH_SCALE,
                                              sc.DEPTH_MAX, sc.TRUNC_MULT)
    g.integrate(keys, torch.from_numpy(d[0]).cuda(),
                torch.from_numpy(c[0]).cuda(), K, K, T[0], sc.DEPTH_SCALE,
                sc.DEPTH_MAX, sc.TRUNC_MULT)
    assert _compare_grids(og, g)[1]


def _all_blocks(g):
    """{key: (tsdf, weight, colour) bytes} of every active block."""
    hm = g.hashmap()
    act = hm.active_buf_indices().cpu().numpy().astype(np.int64)
    keys = hm.key_tensor().cpu().numpy()[act]
    order = np.lexsort(keys.T[::-1])
    act, keys = act[order], keys[order]
    return (keys, g.attribute("tsdf").cpu().numpy()[act],
            g.attribute("weight").cpu().numpy()[act],
            g.attribute("color").cpu().numpy()[act])


@pytest.mark.timeout(900)
def test_fused_frame_groups_with_disjoint_frame_bits_stress():
    """The front roles of group g+1 run in the same launch as the integrate
    role of group g and both use the per-slot touch words (frame bits); the
    two groups must not see each other's words (two planes, by group parity).
    Small images and a view that jumps between frames make consecutive groups
    touch overlapping block sets with DIFFERENT frame bits, many fused groups
    (2 to 16 frames) per call, repeated: the grid must equal frame-by-frame
    integration (frames_per_launch = 1) bit for bit every time, and the
    oracle's. The last two repetitions give the map the head-room the run-ahead
    capacity policy wants for 16-frame groups (262 144 blocks), so that those
    groups really run fused."""
    _lib, geometry = _gpu()
    w, h = 160, 120
    ks = [(i * 137) % 1000 for i in range(96)]
    ds, cs, Ts = [], [], []
    for k in ks:
        d, c, K, T = sc.frames(k, 1, w, h)
        ds.append(d[0]); cs.append(c[0]); Ts.append(T[0])
    dt = [torch.from_numpy(d).cuda() for d in ds]
    ct = [torch.from_numpy(c).cuda() for c in cs]
    ref_g = _mk_grid(geometry, False, block_count=16384)
    ref_g.integrate_frames(dt, ct, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                           sc.TRUNC_MULT, frames_per_launch=1)
    want = _all_blocks(ref_g)
    og = OracleGrid(False, 16384)
    for i in range(24):
        og.integrate(ds[i], cs[i], K, Ts[i])
    for rep in range(8):
        g = _mk_grid(geometry, False,
                     block_count=262144 if rep >= 6 else 16384)
        if rep == 0:
            g.integrate_frames(dt[:24], ct[:24], K, K, Ts[:24], sc.DEPTH_SCALE,
                               sc.DEPTH_MAX, sc.TRUNC_MULT,
                               frames_per_launch=4)
            assert _compare_grids(og, g)[1]
            g.integrate_frames(dt[24:], ct[24:], K, K, Ts[24:],
                               sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT,
                               frames_per_launch=4)
        else:
            g.integrate_frames(dt, ct, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                               sc.TRUNC_MULT, frames_per_launch=(2, 3, 16, 5, 12, 16, 13)[rep - 1])
        got = _all_blocks(g)
        for a, b in zip(want, got):
            assert np.array_equal(a, b), rep


def test_frame_batch_depth_only_and_res8():
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, False, with_color=False, res=8, block_count=8192)
    og = OracleGrid(False, 8192, with_color=False, res=8)
    ds, Ts = [], []
    for k in (400, 410, 420):
        d, c, K, T = sc.frames(k, 1)
        ds.append(d[0]); Ts.append(T[0])
        og.integrate(d[0], None, K, T[0])
    g.integrate_frames([torch.from_numpy(d).cuda() for d in ds], None, K, K,
                       Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
    assert _compare_grids(og, g)[1]


def test_pointcloud_touch_parity():
    _lib, geometry = _gpu()
    from open3d_amd.core import stream
    L = _lib.lib()
    rng = np.random.RandomState(3)
    pts = rng.uniform(-2, 2, (5000, 3)).astype(np.float32)
    want = orc.pointcloud_touch(pts, sc.RES, sc.VOXEL, sc.VOXEL * 8)
    h = C.c_void_p()
    _lib.check(L.o3dmi_hash_create(5000 * 8, 0, None, stream(), C.byref(h)),
               "create")
    out = torch.zeros((5000 * 8, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(L.o3dmi_vbg_pointcloud_touch(
        h, _lib.ptr(torch.from_numpy(pts).cuda()), 5000, _lib.ptr(out),
        5000 * 8, _lib.ptr(cnt), sc.RES, C.c_float(sc.VOXEL),
        C.c_float(sc.VOXEL * 8), stream()), "pcd touch")
    n = int(cnt.item())
    assert np.array_equal(sc.sort_rows(out[:n].cpu().numpy()), want)
    L.o3dmi_hash_destroy(h)


def test_unique_block_coordinates_of_a_point_cloud_and_integrate_from_them():
    """VoxelBlockGrid::GetUniqueBlockCoordinates(pcd, trunc)
    (VoxelBlockGrid.cpp:246-267) through the grid: the block set of the
    oracle's PointCloudTouch for clouds of 1, 5000 and 60 000 points (the
    frustum map is re-created when the cloud outgrows it, and reused after a
    depth-image touch), an empty cloud, and an output buffer that is too
    small."""
    _lib, geometry = _gpu()
    from open3d_amd.core import stream
    g = _mk_grid(geometry, False, block_count=8192)
    rng = np.random.RandomState(5)
    d, c, K, Ts = sc.frames(3, 1)
    g.compute_unique_block_coordinates(torch.from_numpy(d[0]).cuda(), K, Ts[0])
    for n, trunc in ((1, 8.0), (5000, 8.0), (60000, 4.0), (300, 8.0)):
        pts = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
        want = orc.pointcloud_touch(pts, sc.RES, sc.VOXEL, sc.VOXEL * trunc)
        got = g.compute_unique_block_coordinates_pcd(
            torch.from_numpy(pts).cuda(), trunc)
        assert got.dtype == torch.int32 and got.shape[1] == 3
        assert np.array_equal(sc.sort_rows(got.cpu().numpy()),
                              sc.sort_rows(want)), n
    empty = g.compute_unique_block_coordinates_pcd(
        torch.zeros((0, 3), dtype=torch.float32, device="cuda"))
    assert empty.shape == (0, 3)
    # too few output rows: refused, not overrun
    pts = torch.from_numpy(rng.uniform(-1, 1, (2000, 3)).astype(np.float32)).cuda()
    out = torch.full((10 + 4, 3), -9, dtype=torch.int32, device="cuda")
    m = C.c_int64(0)
    st = _lib.lib().o3dmi_vbg_get_unique_block_coordinates_pcd(
        g._g, _lib.ptr(pts), 2000, C.c_float(8.0), _lib.ptr(out), 10,
        C.byref(m), stream())
    assert st != 0
    assert (out[10:].cpu().numpy() == -9).all()


@pytest.mark.parametrize("res", [16, 8])
def test_voxel_indices_coordinates_and_flattened_indices(res):
    """GetVoxelIndices / GetVoxelCoordinates /
    GetVoxelCoordinatesAndFlattenedIndices (VoxelBlockGrid.cpp:130-211,
    kernel VoxelBlockGridImpl.h:43-92) on a grid with integrated frames: the
    forms that take the active indices and the forms given buffer indices
    (repeated, unordered), against the oracle on the grid's own key tensor,
    bit for bit; the flattened indices address the voxels they name in the
    value tensor; a buffer index outside the map is refused."""
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, False, block_count=4096, res=res)
    for k in (0, 7):
        d, c, K, Ts = sc.frames(k, 1, 320, 240)
        g.integrate_frame(torch.from_numpy(d[0]).cuda(),
                          torch.from_numpy(c[0]).cuda(), K, K, Ts[0])
    hm = g.hashmap()
    keys = hm.key_tensor().cpu().numpy()
    active = hm.active_buf_indices()
    assert active.shape[0] > 50
    voxel = sc.VOXEL
    rng = np.random.default_rng(2)
    picks = torch.from_numpy(
        active.cpu().numpy()[rng.integers(0, active.shape[0], 37)]).cuda()
    act_sorted = np.sort(active.cpu().numpy())
    for buf in (None, picks, active[:1], active[:0]):
        vi = g.voxel_indices(buf)
        if buf is None:
            # GetActiveIndices' order is unspecified (here as upstream): the
            # blocks are the active ones, in the order the result shows
            b_np = vi[0, ::res ** 3].cpu().numpy().astype(np.int32)
            assert np.array_equal(np.sort(b_np), act_sorted)
        else:
            b_np = buf.cpu().numpy()
        want_vi = orc.voxel_indices(b_np, res)
        assert vi.dtype == torch.int64 and vi.shape == want_vi.shape
        assert np.array_equal(vi.cpu().numpy(), want_vi)
        vc = g.voxel_coordinates(vi)
        assert np.array_equal(vc.cpu().numpy(),
                              orc.voxel_coordinates(want_vi, keys, res))
        coords, flat = g.voxel_coordinates_and_flattened_indices(buf)
        if buf is None:
            b_np = (flat[::res ** 3] // res ** 3).cpu().numpy().astype(np.int32)
            assert np.array_equal(np.sort(b_np), act_sorted)
        wc, wf = orc.voxel_coords_flat(b_np, keys, res, voxel)
        assert coords.dtype == torch.float32 and flat.dtype == torch.int64
        assert np.array_equal(coords.cpu().numpy(), wc)
        assert np.array_equal(flat.cpu().numpy(), wf)
    # the flattened index names the voxel in the value tensor {cap, r, r, r, 1}
    coords, flat = g.voxel_coordinates_and_flattened_indices(picks)
    tsdf = g.attribute("tsdf")
    vi = g.voxel_indices(picks)
    direct = tsdf[vi[0], vi[3], vi[2], vi[1], 0]
    assert torch.equal(tsdf.reshape(-1)[flat], direct)
    bad = vi.clone()
    bad[0, 5] = hm.capacity()
    with pytest.raises(Exception):
        g.voxel_coordinates(bad)


def test_unproject_parity():
    _lib, geometry = _gpu()
    from open3d_amd.core import stream
    L = _lib.lib()
    d, c, K, Ts = sc.frames(33, 1)
    cf = (c[0].astype(np.float32) / 255.0).astype(np.float32)
    for stride in (1, 4):
        want_p, want_c = orc.unproject(d[0], cf, K, Ts[0], sc.DEPTH_SCALE,
                                       sc.DEPTH_MAX, stride)
        n = (480 // stride) * (640 // stride)
        pts = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        cols = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(L.o3dmi_unproject(
            _lib.ptr(torch.from_numpy(d[0]).cuda()), _lib.U16, 480, 640,
            _lib.ptr(torch.from_numpy(cf).cuda()), _lib.ptr(pts),
            _lib.ptr(cols), _lib.ptr(cnt), _lib.f64p(K), _lib.f64p(Ts[0]),
            C.c_float(sc.DEPTH_SCALE), C.c_float(sc.DEPTH_MAX), stride,
            stream()), "unproject")
        m = int(cnt.item())
        assert m == want_p.shape[0]
        got = np.concatenate([pts[:m].cpu().numpy(), cols[:m].cpu().numpy()], 1)
        want = np.concatenate([want_p, want_c], 1)
        assert np.array_equal(sc.sort_rows(got), sc.sort_rows(want))


@pytest.mark.parametrize("grid_f32", [False, True])
def test_raycast_parity(grid_f32):
    """EstimateRange + RayCast vs the oracle after integrating 12 frames;
    every render attribute the reference supports."""
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, grid_f32)
    og = OracleGrid(grid_f32, 4096)
    for k in range(200, 212):
        d, c, K, Ts = sc.frames(k, 1)
        keys = og.integrate(d[0], c[0], K, Ts[0])
        g.integrate(torch.from_numpy(keys).cuda(),
                    torch.from_numpy(d[0]).cuda(),
                    torch.from_numpy(c[0]).cuda(), K, K, Ts[0],
                    sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
    assert _compare_grids(og, g)[1]
    d, c, K, Ts = sc.frames(206, 1)
    T = Ts[0]
    keys = orc.depth_touch(d[0], K, T, sc.RES, sc.VOXEL,
                           sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, 4)
    attrs = ("depth", "vertex", "color", "normal", "index", "mask",
             "interp_ratio", "interp_ratio_dx", "interp_ratio_dy",
             "interp_ratio_dz")
    range_o, needed = orc.estimate_range(keys, K, T, 480, 640, 8, sc.RES,
                                         sc.VOXEL, 0.1, sc.DEPTH_MAX,
                                         frag_buffer_size=65536)
    assert needed < 65536
    want = orc.raycast(og.h, og.tsdf, og.weight, og.color, range_o, K, T, 480,
                       640, sc.RES, sc.VOXEL, sc.DEPTH_SCALE, 0.1,
                       sc.DEPTH_MAX, 3.0, sc.TRUNC_MULT, 8, attrs)
    got = g.ray_cast(torch.from_numpy(keys).cuda(), K, T, 640, 480, attrs,
                     sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX, 3.0, sc.TRUNC_MULT, 8)
    assert np.array_equal(got["range"].cpu().numpy(), range_o)
    hit = want["depth"][..., 0] > 0
    assert hit.mean() > 0.5
    gd = got["depth"].cpu().numpy()
    assert np.array_equal(gd > 0, want["depth"] > 0)
    assert np.abs(gd - want["depth"]).max() <= 1e-3  # depth units (mm)
    for a in ("vertex", "normal", "interp_ratio", "interp_ratio_dx",
              "interp_ratio_dy", "interp_ratio_dz"):
        assert np.abs(got[a].cpu().numpy() - want[a]).max() <= 1e-5, a
    assert np.abs(got["color"].cpu().numpy() - want["color"]).max() <= 1e-5
    assert np.array_equal(got["mask"].cpu().numpy(), want["mask"])
    # voxel indices are buffer-index based: compare through the key of the
    # block they point into.
    res3 = sc.RES ** 3
    gi = got["index"].cpu().numpy()
    wi = want["index"]
    m = want["mask"]
    gk = g.hashmap().key_tensor().cpu().numpy()[(gi // res3)[m]]
    wk = og.h.key_buffer()[(wi // res3)[m]]
    assert np.array_equal(gk, wk)
    assert np.array_equal((gi % res3)[m], (wi % res3)[m])


def _raycast_case(res, width, height, down, weight_threshold, n_frames,
                  grid_f32=False):
    """Integrate n_frames, then EstimateRange + RayCast (depth / vertex /
    normal / colour) on both sides; returns (oracle maps, library maps)."""
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, grid_f32, res=res, block_count=8192)
    og = OracleGrid(grid_f32, 8192, res=res)
    for k in range(300, 300 + 2 * n_frames, 2):
        d, c, K, Ts = sc.frames(k, 1, width, height)
        keys = og.integrate(d[0], c[0], K, Ts[0])
        g.integrate(torch.from_numpy(keys).cuda(),
                    torch.from_numpy(d[0]).cuda(),
                    torch.from_numpy(c[0]).cuda(), K, K, Ts[0],
                    sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
    T = Ts[0]
    attrs = ("depth", "vertex", "normal", "color")
    range_o, needed = orc.estimate_range(keys, K, T, height, width, down, res,
                                         sc.VOXEL, 0.1, sc.DEPTH_MAX,
                                         frag_buffer_size=65536)
    assert needed < 65536
    want = orc.raycast(og.h, og.tsdf, og.weight, og.color, range_o, K, T,
                       height, width, res, sc.VOXEL, sc.DEPTH_SCALE, 0.1,
                       sc.DEPTH_MAX, weight_threshold, sc.TRUNC_MULT, down,
                       attrs)
    got = g.ray_cast(torch.from_numpy(keys).cuda(), K, T, width, height, attrs,
                     sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX, weight_threshold,
                     sc.TRUNC_MULT, down)
    assert np.array_equal(got["range"].cpu().numpy(), range_o)
    return want, {a: got[a].cpu().numpy() for a in attrs}


def _assert_maps(want, got):
    assert np.array_equal(got["depth"] > 0, want["depth"] > 0)
    assert np.abs(got["depth"] - want["depth"]).max() <= 1e-3  # mm
    for a in ("vertex", "normal", "color"):
        assert np.abs(got[a] - want[a]).max() <= 1e-5, a


@pytest.mark.parametrize("res,width,height,down,min_hit", [
    (16, 200, 148, 4, 0.9),  # tiles cut by the right and the lower border
    (8, 320, 240, 4, 0.9),   # the kernel taking the resolution at run time
    (16, 72, 40, 8, 0.5),    # 15 tiles: fewer than two per XCD
])
def test_raycast_partial_tiles_and_block_resolutions(res, width, height, down,
                                                     min_hit):
    """The ray cast's workgroup tile is 32 x 8 pixels and the second phase of
    its march is a wave-wide loop: pixels past the image border keep a lane
    but no ray. Every map against the oracle."""
    want, got = _raycast_case(res, width, height, down, 1.0, 6)
    assert (want["depth"] > 0).mean() > min_hit
    _assert_maps(want, got)


@pytest.mark.parametrize("grid_f32", [False, True])
def test_raycast_long_crawls_through_hidden_surfaces(grid_f32):
    """A weight threshold above every weight in the grid hides all surfaces:
    each ray then walks through its whole truncation band one voxel at a time
    (tsdf near or below zero => stride = voxel size) -- the case the ray
    cast's cooperative march is for (a crawling ray's next samples are taken
    by idle lanes of its wave). No pixel may hit; then, with a threshold that
    only voxels seen by all 6 integrations reach, part of the surfaces are hidden and the
    rays crawl past them to the next one. Every map against the oracle."""
    want, got = _raycast_case(16, 320, 240, 8, 100.0, 6, grid_f32)
    assert not (want["depth"] > 0).any()
    _assert_maps(want, got)
    want, got = _raycast_case(16, 320, 240, 8, 6.0, 6, grid_f32)
    hit = (want["depth"] > 0).mean()
    assert 0.05 < hit < 0.999, hit
    _assert_maps(want, got)


def test_frame_stream_long_run_matches_reference_cpu_bodies():
    """Near-full-size parity: 160 consecutive VGA frames (BASELINE configs[1]
    stream, every frame) through integrate_frames with 4-frame groups vs the
    CPU path frame by frame -- Open3D's own IntegrateCPU/DepthTouchCPU bodies
    (oracle/_ref) when the prebuilt library is present, else the restated
    oracle. Whole grid bit-exact (~5000 blocks, weights up to 160)."""
    _lib, geometry = _gpu()
    import _ref as ref
    impl = ref if ref.available() else orc
    impl.set_threads(8)  # more threads only add contention at this size
    n = 160
    cap = 16384
    g = _mk_grid(geometry, False, block_count=cap)
    og = OracleGrid(False, cap)
    dts, cts, Ts = [], [], []
    for k0 in range(0, n, 16):
        d, c, K, T = sc.frames(k0, 16)
        for i in range(16):
            keys = impl.depth_touch(d[i], K, T[i], sc.RES, sc.VOXEL,
                                    sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                                    sc.DEPTH_MAX, 4)
            og.h.activate(keys)
            buf, _ = og.h.find(keys)
            impl.integrate(d[i], c[i], buf, og.h.key_buffer(), og.tsdf,
                           og.weight, og.color, K, K, T[i], sc.RES, sc.VOXEL,
                           sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                           sc.DEPTH_MAX)
            dts.append(torch.from_numpy(d[i]).cuda())
            cts.append(torch.from_numpy(c[i]).cuda())
            Ts.append(T[i])
    for lo in range(0, n, 50):
        g.integrate_frames(dts[lo:lo + 50], cts[lo:lo + 50], K, K,
                           Ts[lo:lo + 50], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                           sc.TRUNC_MULT, frames_per_launch=4)
    assert _compare_grids(og, g)[1]
    assert og.weight.max() >= 100


def test_frame_stream_edge_cases():
    """Empty batch, a frame that touches nothing (all-zero depth), a frame
    whose depth is entirely beyond depth_max, then a normal frame: the grid
    must equal the oracle's after the normal frame only."""
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, False, block_count=4096)
    og = OracleGrid(False, 4096)
    d, c, K, Ts = sc.frames(77, 1)
    g.integrate_frames([], [], K, K, [], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                       sc.TRUNC_MULT)
    zero = torch.zeros((480, 640), dtype=torch.uint16, device="cuda")
    far = torch.full((480, 640), 60000, dtype=torch.uint16, device="cuda")
    ct = torch.from_numpy(c[0]).cuda()
    dt = torch.from_numpy(d[0]).cuda()
    g.integrate_frames([zero, far, dt, zero], [ct, ct, ct, ct], K, K,
                       [Ts[0]] * 4, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                       sc.TRUNC_MULT)
    g.integrate_frame(far, ct, K, K, Ts[0], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                      sc.TRUNC_MULT)
    og.integrate(d[0], c[0], K, Ts[0])
    assert _compare_grids(og, g)[1]


def test_c5_voxel_size_4mm_grid():
    """BASELINE configs[4] voxel size: 4 mm voxels (6.4 cm blocks, several
    thousand blocks per VGA frame, the map starts smaller than one frame's block
    set and has to Reserve). Whole grid bit-exact against the CPU path; surface extraction
    and a save / load round trip on the result."""
    import os
    import _ref as ref
    _lib, geometry = _gpu()
    voxel = 0.004
    g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                [torch.float32, torch.uint16, torch.uint16],
                                [1, 1, 3], voxel, sc.RES, 2048)
    use_ref = ref.available()
    cap = 65536
    h = orc.HashMap(cap)
    tsdf = np.zeros((cap, 16, 16, 16), np.float32)
    wgt = np.zeros((cap, 16, 16, 16), np.uint16)
    col = np.zeros((cap, 16, 16, 16, 3), np.uint16)
    if use_ref:
        ref.set_threads(min(64, os.cpu_count() or 1))
    orc.set_threads(min(64, os.cpu_count() or 1))
    ds, cs, Ts = [], [], []
    for k in range(0, 12, 2):
        d, c, K, T = sc.frames(k, 1)
        ds.append(d[0]); cs.append(c[0]); Ts.append(T[0])
        keys = (ref if use_ref else orc).depth_touch(
            d[0], K, T[0], sc.RES, voxel, voxel * sc.TRUNC_MULT,
            sc.DEPTH_SCALE, sc.DEPTH_MAX, 4)
        assert keys.shape[0] > 2000
        h.activate(keys)
        buf, m = h.find(keys)
        (ref if use_ref else orc).integrate(
            d[0], c[0], buf, h.key_buffer(), tsdf, wgt, col, K, K, T[0],
            sc.RES, voxel, voxel * sc.TRUNC_MULT, sc.DEPTH_SCALE, sc.DEPTH_MAX)
    g.integrate_frames([torch.from_numpy(x).cuda() for x in ds],
                       [torch.from_numpy(x).cuda() for x in cs], K, K, Ts,
                       sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
    hm = g.hashmap()
    n = h.size()
    assert hm.size() == n and n > 2048 and hm.capacity() >= n  # it reserved
    okeys = h.key_buffer()[:n].copy()
    obuf, _ = h.find(okeys)
    gbuf, gm = hm.find(torch.from_numpy(okeys).cuda())
    assert bool(gm.all())
    gbuf = gbuf.cpu().numpy().astype(np.int64)
    assert np.array_equal(g.attribute("tsdf").cpu().numpy()[gbuf][..., 0],
                          tsdf[obuf])
    assert np.array_equal(g.attribute("weight").cpu().numpy()[gbuf][..., 0],
                          wgt[obuf])
    assert np.array_equal(g.attribute("color").cpu().numpy()[gbuf], col[obuf])
    pcd = g.extract_point_cloud(2.0)
    assert pcd["positions"].shape[0] > 100000


@pytest.mark.parametrize("path", ["frames", "two_step"])
def test_block_ownership_sharding_is_bit_identical(path):
    """Multi-GPU scheme A on one device: `world` grids see the same stream, grid
    r only takes the blocks it owns. Their key sets are the ownership classes of
    the unsharded grid's keys and every block is bit-identical to it."""
    _lib, geometry = _gpu()
    from open3d_amd import sharding
    world = 3
    fr = [sc.frames(k, 1, 320, 240) for k in range(0, 40, 4)]
    K = fr[0][2]
    ds = [torch.from_numpy(f[0][0]).cuda() for f in fr]
    cs = [torch.from_numpy(f[1][0]).cuda() for f in fr]
    Ts = [f[3][0] for f in fr]

    def run(rank, w):
        g = _mk_grid(geometry, False, block_count=4096)
        if w > 1:
            g.set_block_ownership(rank, w)
        if path == "frames":
            g.integrate_frames(ds, cs, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                               sc.TRUNC_MULT)
        else:
            for d, c, T in zip(ds, cs, Ts):
                keys = g.compute_unique_block_coordinates(
                    d, K, T, sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
                g.integrate(keys, d, c, K, K, T, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                            sc.TRUNC_MULT)
        hm = g.hashmap()
        act = hm.active_buf_indices().cpu().numpy().astype(np.int64)
        keys = hm.key_tensor().cpu().numpy()[act]
        order = np.lexsort(keys.T[::-1])
        return (keys[order],
                g.attribute("tsdf").cpu().numpy()[act][order],
                g.attribute("weight").cpu().numpy()[act][order],
                g.attribute("color").cpu().numpy()[act][order])

    full = run(0, 1)
    owner = sharding.block_owner(full[0], world)
    assert np.bincount(owner, minlength=world).min() > 100
    total = 0
    for r in range(world):
        part = run(r, world)
        sel = owner == r
        assert np.array_equal(part[0], full[0][sel])
        for a, b in zip(part[1:], full[1:]):
            assert a.tobytes() == b[sel].tobytes()
        total += part[0].shape[0]
    assert total == full[0].shape[0]


def _blocks_sorted(g, with_color=True):
    hm = g.hashmap()
    act = hm.active_buf_indices().cpu().numpy().astype(np.int64)
    keys = hm.key_tensor().cpu().numpy()[act]
    order = np.lexsort(keys.T[::-1])
    out = [keys[order], g.attribute("tsdf").cpu().numpy()[act][order],
           g.attribute("weight").cpu().numpy()[act][order]]
    if with_color:
        out.append(g.attribute("color").cpu().numpy()[act][order])
    return out


@pytest.mark.parametrize("raw", [False, True])
@pytest.mark.parametrize("world,group,grid_f32,with_color", [
    (1, 4, False, True), (3, 2, False, True), (8, 3, False, True),
    (8, 12, False, True), (2, 16, True, True), (4, 5, False, False)])
def test_sliced_touch_ownership_union_is_the_single_grid(world, group,
                                                         grid_f32, with_color,
                                                         raw, monkeypatch):
    """SURVEY 8(e) scheme A as specified: rank r touches only its band of ray
    tiles, the candidate records of all ranks are gathered (here computed on
    one device: gather_slices), rank r activates the keys it owns and
    integrates them with ONE launch per chunk -- from records prepared on the
    side stream, or (raw) straight from the images. The `world` grids are the
    ownership classes of the single grid the ordinary stream builds, bit for
    bit -- several chunks per call, chunks that end mid-way, depth-only and
    float32 grids."""
    _lib, geometry = _gpu()
    from open3d_amd import sharding
    monkeypatch.setenv("O3DMI_SLICED_RAW", "1" if raw else "0")
    n = 2 * 16 * group + 3 if group <= 3 else 16 * group + 5
    ks = [(i * 7) % 900 for i in range(n)]
    ds, cs, dt, ct, Ts, K = _stream_frames(ks, 320, 240)
    full_g = _mk_grid(geometry, grid_f32, block_count=8192,
                      with_color=with_color)
    full_g.integrate_frames(dt, ct if with_color else None, K, K, Ts,
                            sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT,
                            frames_per_launch=group)
    full = _blocks_sorted(full_g, with_color)
    owner = sharding.block_owner(full[0], world)
    total = 0
    gathered = None
    for r in range(world):
        g = _mk_grid(geometry, grid_f32, block_count=8192,
                     with_color=with_color)
        g.set_block_ownership(r, world)
        batch = g.prepare_frames(dt, ct if with_color else None, K, K, Ts)
        if gathered is None:
            gathered = g.gather_slices(batch, world, sc.DEPTH_SCALE,
                                       sc.DEPTH_MAX, sc.TRUNC_MULT, group)
            assert len(gathered) == -(-n // (16 * group))
        g.integrate_frames_sliced(batch, [t.clone() for t in gathered],
                                  sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT,
                                  frames_per_launch=group)
        part = _blocks_sorted(g, with_color)
        sel = owner == r
        assert np.array_equal(part[0], full[0][sel]), r
        for a, b in zip(part[1:], full[1:]):
            assert a.tobytes() == b[sel].tobytes(), r
        total += part[0].shape[0]
        assert g.sliced_stats()["reapplied"] == 0
    assert total == full[0].shape[0]


def test_sliced_touch_reserves_and_applies_the_chunk_again():
    """A block map too small for a chunk's keys: the chunk is dropped on the
    device, the map reserved and the SAME gathered records applied again; a
    wire segment too small for a slice is an error when the segments are the
    caller's."""
    _lib, geometry = _gpu()
    ks = [(i * 11) % 900 for i in range(70)]
    ds, cs, dt, ct, Ts, K = _stream_frames(ks, 320, 240)
    og = OracleGrid(False, 16384)
    for i in range(len(ds)):
        og.integrate(ds[i], cs[i], K, Ts[i])
    g = _mk_grid(geometry, False, block_count=300)
    batch = g.prepare_frames(dt, ct, K, K, Ts)
    gathered = g.gather_slices(batch, 1, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                               sc.TRUNC_MULT, 4)
    g.integrate_frames_sliced(batch, gathered, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                              sc.TRUNC_MULT, frames_per_launch=4)
    assert _compare_grids(og, g)[1]
    st = g.sliced_stats()
    assert st["reapplied"] >= 1 and g.hashmap().capacity() >= og.h.size()
    # the ordinary stream continues on the same map
    d, c, K2, T = sc.frames(950, 1, 320, 240)
    og.integrate(d[0], c[0], K, T[0])
    g.integrate_frame(torch.from_numpy(d[0]).cuda(),
                      torch.from_numpy(c[0]).cuda(), K, K, T[0],
                      sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)
    assert _compare_grids(og, g)[1]
    g2 = _mk_grid(geometry, False, block_count=8192)
    g2.set_slice_capacity(16, 64)
    b2 = g2.prepare_frames(dt, ct, K, K, Ts)
    small = g2.gather_slices(b2, 1, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                             sc.TRUNC_MULT, 4)
    with pytest.raises(_lib.O3DMIError):
        g2.integrate_frames_sliced(b2, small, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                                   sc.TRUNC_MULT, frames_per_launch=4)


def test_a_peers_abort_flag_ends_the_call_and_leaves_the_map_usable():
    """o3dmi_vbg_integrate_frames on the sliced path is collective: a rank
    that has to leave it with an error delivers its next all-gather with an
    empty segment flagged kSliceFlagAbort (sliced_path.h), and every rank
    returns O3DMI_ERR_PEER at that chunk instead of waiting in a collective
    nobody else will enter. Receiver side, on one device: the second chunk's
    gathered segments carry the flag in rank 1's header -> the call ends with
    status 10 after the first chunk, the map is consistent (size / export
    work), and integrating the remaining frames afterwards gives the grid of
    an undisturbed run."""
    _lib, geometry = _gpu()
    group, world = 2, 2
    n = 16 * group * 2 + 5  # three chunks
    ks = [(i * 13) % 900 for i in range(n)]
    ds, cs, dt, ct, Ts, K = _stream_frames(ks, 320, 240)

    def grid():
        g = _mk_grid(geometry, False, block_count=8192)
        g.set_block_ownership(0, world)
        return g
    clean = grid()
    batch = clean.prepare_frames(dt, ct, K, K, Ts)
    gathered = clean.gather_slices(batch, world, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                                   sc.TRUNC_MULT, group)
    assert len(gathered) == 3
    clean.integrate_frames_sliced(batch, [t.clone() for t in gathered],
                                  sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT,
                                  frames_per_launch=group)
    want = _blocks_sorted(clean, True)

    g = grid()
    b = g.prepare_frames(dt, ct, K, K, Ts)
    seg = g.slice_segment_bytes()
    bad = [t.clone() for t in gathered]
    hdr = bad[1][seg:seg + 8].view(torch.int32)  # rank 1: {count, flags}
    hdr[0] = 0
    hdr[1] = 8  # kSliceFlagAbort
    with pytest.raises(_lib.O3DMIError) as ei:
        g.integrate_frames_sliced(b, bad, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                                  sc.TRUNC_MULT, frames_per_launch=group)
    assert ei.value.status == 10  # O3DMI_ERR_PEER
    # the map answers (no "overflow not recovered"), holds the first chunk
    first = g.hashmap().size()
    assert 0 < first <= want[0].shape[0]
    # the rest of the stream, undisturbed, on the same grid
    cf = 16 * group
    rest = g.prepare_frames(dt[cf:], ct[cf:], K, K, Ts[cf:])
    g.integrate_frames_sliced(rest, [t.clone() for t in gathered[1:]],
                              sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT,
                              frames_per_launch=group)
    got = _blocks_sorted(g, True)
    for a, c in zip(got, want):
        assert a.tobytes() == c.tobytes()


def _sorted_blocks(g):
    """Exported blocks in lexicographic key order (host)."""
    keys, vals = g.export_blocks()
    keys = keys.cpu().numpy()
    order = np.lexsort(keys.T[::-1])
    return [keys[order]] + [v.cpu().numpy()[order] for v in vals]


@pytest.mark.parametrize("f32_grid", [False, True])
def test_merge_of_one_frame_grids_is_integration(f32_grid):
    """o3dmi_vbg_merge_blocks computes Integrate's own running mean: folding
    one-frame grids in, frame by frame, is bit-identical to integrating the
    frames into one grid (block set, TSDF, weight, colour). Also: export_blocks
    is the order / content of the active rows, and merging nothing or an empty
    grid changes nothing."""
    _lib, geometry = _gpu()
    fr = [sc.frames(k, 1, 320, 240) for k in range(0, 24, 4)]
    K = fr[0][2]

    def integrate(g, f):
        g.integrate_frame(torch.from_numpy(f[0][0]).cuda(),
                          torch.from_numpy(f[1][0]).cuda(), K, K, f[3][0],
                          sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT)

    full = _mk_grid(geometry, f32_grid, block_count=4096)
    for f in fr:
        integrate(full, f)
    merged = _mk_grid(geometry, f32_grid, block_count=64)  # grows by Reserve
    for f in fr:
        one = _mk_grid(geometry, f32_grid, block_count=4096)
        integrate(one, f)
        keys, vals = one.export_blocks()
        hm = one.hashmap()
        act = np.sort(hm.active_buf_indices().cpu().numpy().astype(np.int64))
        assert np.array_equal(keys.cpu().numpy(),
                              hm.key_tensor().cpu().numpy()[act])
        for name, v in zip(one.attr_names, vals):
            assert np.array_equal(v.cpu().numpy(),
                                  one.attribute(name).cpu().numpy()[act])
        merged.merge_blocks(keys, vals)
    want, got = _sorted_blocks(full), _sorted_blocks(merged)
    assert got[0].shape[0] > 300
    for a, b in zip(want, got):
        assert a.shape == b.shape and a.tobytes() == b.tobytes()
    # nothing / an empty grid folded in: unchanged
    empty = _mk_grid(geometry, f32_grid, block_count=16)
    k0, v0 = empty.export_blocks()
    assert k0.shape[0] == 0
    merged.merge_blocks(k0, v0)
    for a, b in zip(want, _sorted_blocks(merged)):
        assert a.tobytes() == b.tobytes()
    with pytest.raises(ValueError, match="attribute layout"):
        merged.merge_blocks(keys, vals[:-1] + [vals[-1][:1]])


@pytest.mark.parametrize("f32_grid", [False, True])
def test_frame_sharded_grids_merge_to_the_single_stream_model(f32_grid):
    """SURVEY 8(e) scheme B on one device: two ranks integrate the even / odd
    frames into private grids, then each folds the other's blocks in. Block set
    and weights equal the single-stream grid exactly; TSDF is the same weighted
    sum in another association (<= 1e-5, inside the 1e-4 bar); colour within
    the truncation slack of the uint16 stores. The two ranks end bit-identical
    (the two-term mean is commutative)."""
    _lib, geometry = _gpu()
    fr = [sc.frames(k, 1, 320, 240) for k in range(0, 36, 3)]
    K = fr[0][2]

    def run(sel):
        g = _mk_grid(geometry, f32_grid, block_count=4096)
        g.integrate_frames([torch.from_numpy(f[0][0]).cuda() for f in sel],
                           [torch.from_numpy(f[1][0]).cuda() for f in sel],
                           K, K, [f[3][0] for f in sel], sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, sc.TRUNC_MULT)
        return g

    full = run(fr)
    parts = [run(fr[0::2]), run(fr[1::2])]
    exported = [g.export_blocks() for g in parts]
    parts[0].merge_blocks(*exported[1])
    parts[1].merge_blocks(*exported[0])
    want = _sorted_blocks(full)
    got = [_sorted_blocks(g) for g in parts]
    for a, b in zip(got[0], got[1]):
        assert a.tobytes() == b.tobytes()
    names = full.attr_names
    for name, a, b in zip(["key"] + names, want, got[0]):
        if name in ("key", "weight"):
            assert np.array_equal(a, b), name
        elif name == "tsdf":
            assert np.abs(a - b).max() <= 1e-5
        else:
            d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
            assert d <= (1e-2 if f32_grid else len(fr)), d
    assert want[names.index("weight") + 1].max() >= len(fr) // 2


def _frame_sharded_rank(rank, world):
    """One rank of the multi-process frame-sharded run (gloo between the
    processes, the device for everything else)."""
    import torch.distributed as dist
    from open3d_amd.sharding import merge_frame_sharded_grid
    _lib, geometry = _gpu()
    fr = [sc.frames(k, 1, 320, 240) for k in range(0, 24, 3)][rank::world]
    K = fr[0][2]
    g = _mk_grid(geometry, False, block_count=4096)
    g.integrate_frames([torch.from_numpy(f[0][0]).cuda() for f in fr],
                       [torch.from_numpy(f[1][0]).cuda() for f in fr],
                       K, K, [f[3][0] for f in fr], sc.DEPTH_SCALE,
                       sc.DEPTH_MAX, sc.TRUNC_MULT)
    merge_frame_sharded_grid(g, dist, replicate=True)
    torch.cuda.synchronize()
    return _sorted_blocks(g)


def test_frame_sharded_merge_across_two_processes():
    """sharding.merge_frame_sharded_grid end to end: two processes (one rank
    each, sharing this GPU; gloo as the transport) integrate the even / odd
    frames, run the library's owner-partitioned exchange
    (o3dmi_vbg_merge_frame_sharded) and replicate the finished blocks
    (o3dmi_vbg_allgather_owned_blocks). Both ranks end with the single-stream
    block set and weights, TSDF within 1e-5, bit-identical to each other."""
    from test_sharding import _run
    _lib, geometry = _gpu()
    fr = [sc.frames(k, 1, 320, 240) for k in range(0, 24, 3)]
    K = fr[0][2]
    full = _mk_grid(geometry, False, block_count=4096)
    full.integrate_frames([torch.from_numpy(f[0][0]).cuda() for f in fr],
                          [torch.from_numpy(f[1][0]).cuda() for f in fr],
                          K, K, [f[3][0] for f in fr], sc.DEPTH_SCALE,
                          sc.DEPTH_MAX, sc.TRUNC_MULT)
    want = _sorted_blocks(full)
    got = _run(_frame_sharded_rank)
    for a, b in zip(got[0], got[1]):
        assert a.tobytes() == b.tobytes()
    keys, tsdf, weight, color = got[0]
    assert np.array_equal(keys, want[0])
    assert np.array_equal(weight, want[2])
    assert np.abs(tsdf - want[1]).max() <= 1e-5
    assert np.abs(color.astype(np.int64) - want[3].astype(np.int64)).max() \
        <= len(fr)


def _sliced_rank(rank, world):
    """One rank of the multi-process block-ownership run with the sliced touch:
    integrate_frames on a grid with an ownership and the library's
    communicator installed (torch.distributed / gloo as its transport)."""
    import torch.distributed as dist
    from open3d_amd.sharding import Comm
    _lib, geometry = _gpu()
    ks = [(i * 7) % 900 for i in range(70)]
    ds, cs, dt, ct, Ts, K = _stream_frames(ks, 320, 240)
    g = _mk_grid(geometry, False, block_count=8192)
    g.set_block_ownership(rank, world)
    comm = Comm.torch(dist)
    comm.install()
    try:
        # two calls: the second starts while the side stream still holds the
        # first one's last chunk
        g.integrate_frames(dt[:40], ct[:40], K, K, Ts[:40], sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, sc.TRUNC_MULT, frames_per_launch=2)
        g.integrate_frames(dt[40:], ct[40:], K, K, Ts[40:], sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, sc.TRUNC_MULT, frames_per_launch=2)
        torch.cuda.synchronize()
        st = g.sliced_stats()
    finally:
        Comm.uninstall()
        comm.destroy()
    return _blocks_sorted(g), st


def test_sliced_touch_across_two_processes():
    """The sliced path end to end with a REAL exchange: two processes (one
    rank each, sharing this GPU), the library's communicator over
    torch.distributed (gloo), `integrate_frames` dispatching to the sliced
    path by itself. The two grids are the two ownership classes of the single
    grid, bit for bit, and every chunk went through the all-gather."""
    from test_sharding import _run
    from open3d_amd import sharding
    _lib, geometry = _gpu()
    ks = [(i * 7) % 900 for i in range(70)]
    ds, cs, dt, ct, Ts, K = _stream_frames(ks, 320, 240)
    full_g = _mk_grid(geometry, False, block_count=8192)
    full_g.integrate_frames(dt, ct, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                            sc.TRUNC_MULT, frames_per_launch=2)
    full = _blocks_sorted(full_g)
    owner = sharding.block_owner(full[0], 2)
    got = _run(_sliced_rank)
    for r in range(2):
        part, st = got[r]
        assert st["chunks"] == 2 + 1  # 40 frames = 2 chunks of 32, 30 = 1
        sel = owner == r
        assert np.array_equal(part[0], full[0][sel])
        for a, b in zip(part[1:], full[1:]):
            assert a.tobytes() == b[sel].tobytes()


def test_sliced_touch_all_gather_over_rccl_single_rank():
    """ncclAllGather on the side stream of the sliced path, on the hardware at
    hand: a one-rank RCCL communicator (more ranks need more GPUs than the
    test box has) -- the grid equals the ordinary stream's."""
    _lib, geometry = _gpu()
    from open3d_amd.sharding import Comm
    L = _lib.lib()
    if not L.o3dmi_rccl_available():
        pytest.skip("no librccl.so in this process")
    ident = (C.c_char * 128)()
    _lib.check(L.o3dmi_rccl_unique_id(C.cast(ident, C.c_void_p)), "unique_id")
    h = C.c_void_p()
    _lib.check(L.o3dmi_comm_create_rccl(ident.raw, 0, 1, C.byref(h)),
               "comm_create_rccl")
    comm = Comm(h)
    ks = [(i * 7) % 900 for i in range(40)]
    ds, cs, dt, ct, Ts, K = _stream_frames(ks, 320, 240)
    want_g = _mk_grid(geometry, False, block_count=8192)
    want_g.integrate_frames(dt, ct, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                            sc.TRUNC_MULT, frames_per_launch=2)
    g = _mk_grid(geometry, False, block_count=8192)
    batch = g.prepare_frames(dt, ct, K, K, Ts)
    comm.install()
    try:
        g.integrate_frames_sliced(batch, None, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                                  sc.TRUNC_MULT, frames_per_launch=2)
        torch.cuda.synchronize()
    finally:
        Comm.uninstall()
        comm.destroy()
    for a, b in zip(_blocks_sorted(g), _blocks_sorted(want_g)):
        assert a.tobytes() == b.tobytes()


def test_last_frame_block_coordinates_equal_a_second_block_touch():
    """o3dmi_vbg_last_frame_block_coordinates (extension): after
    integrate_frame the device-side block list is the set
    GetUniqueBlockCoordinates returns for the same frame, and a ray cast over
    it with the count left on the device produces the very same maps."""
    _lib, geometry = _gpu()
    g = _mk_grid(geometry, False, block_count=16384)
    for k in (300, 305, 310):
        d, c, K, T = sc.frames(k, 1)
        dt, ct = torch.from_numpy(d[0]).cuda(), torch.from_numpy(c[0]).cuda()
        g.integrate_frame(dt, ct, K, K, T[0], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                          sc.TRUNC_MULT)
        coords, cnt = g.last_frame_block_coordinates(
            (d.shape[1] // 4) * (d.shape[2] // 4) * 4)
        want = g.compute_unique_block_coordinates(dt, K, T[0], sc.DEPTH_SCALE,
                                                  sc.DEPTH_MAX, sc.TRUNC_MULT)
        n = int(cnt.item())
        assert n == want.shape[0] and n > 100
        assert np.array_equal(sc.sort_rows(coords[:n].cpu().numpy()),
                              sc.sort_rows(want.cpu().numpy()))
        h, w = d.shape[1], d.shape[2]
        a = g.ray_cast(coords, K, T[0], w, h, ("depth", "normal", "color"),
                       sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT,
                       block_count_dev=cnt)
        b = g.ray_cast(want, K, T[0], w, h, ("depth", "normal", "color"),
                       sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT)
        for name in ("depth", "normal", "color"):
            assert torch.equal(a[name], b[name]), name
        assert float((a["depth"] > 0).float().mean()) > 0.5
        # Round 6: no block coordinates and no range map at all -- the grid's
        # own list and its own range map, which the ray cast that consumes it
        # leaves clean (second and third frame: no clearing launch); a call
        # with other depth limits in between must refill it, and a call with
        # a caller's map must not disturb it
        for lim in ((0.1, sc.DEPTH_MAX), (0.1, sc.DEPTH_MAX), (0.3, 2.0),
                    (0.1, sc.DEPTH_MAX)):
            own = g.ray_cast_last_frame(K, T[0], w, h, ("depth", "normal"),
                                        sc.DEPTH_SCALE, lim[0], lim[1], 1.0,
                                        sc.TRUNC_MULT)
            ref = g.ray_cast(want, K, T[0], w, h, ("depth", "normal"),
                             sc.DEPTH_SCALE, lim[0], lim[1], 1.0,
                             sc.TRUNC_MULT)
            for name in ("depth", "normal"):
                assert torch.equal(own[name], ref[name]), (name, lim)
    # a size whose range map cannot be self-cleaning (down factor 4), and an
    # image size change: both render what the explicit call renders
    own = g.ray_cast_last_frame(K, T[0], w, h, ("depth",), sc.DEPTH_SCALE,
                                0.1, sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT,
                                range_map_down_factor=4)
    ref = g.ray_cast(want, K, T[0], w, h, ("depth",), sc.DEPTH_SCALE, 0.1,
                     sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT,
                     range_map_down_factor=4)
    assert torch.equal(own["depth"], ref["depth"])
    own = g.ray_cast_last_frame(K, T[0], w, h, ("depth",), sc.DEPTH_SCALE,
                                0.1, sc.DEPTH_MAX, 1.0, sc.TRUNC_MULT)
    assert torch.equal(own["depth"], a["depth"])
    # without a frame-stream integration there is no list to take
    g2 = _mk_grid(geometry, False, block_count=1024)
    with pytest.raises(_lib.O3DMIError):
        g2.ray_cast_last_frame(K, T[0], w, h, ("depth",))


def test_preload_loads_the_kernels_and_may_be_repeated():
    """o3dmi_preload (extension): loads the code objects of the tracking /
    integration path; idempotent."""
    _lib, _ = _gpu()
    L = _lib.lib()
    assert L.o3dmi_preload() == 0
    assert L.o3dmi_preload() == 0


def test_sort_indices_is_a_counting_sort():
    """o3dmi_sort_indices (scan.hip): ascending order of buffer indices --
    distinct ones, duplicates, a range of more than one scan tile."""
    _lib, _ = _gpu()
    from open3d_amd.core import stream
    L = _lib.lib()
    rng = np.random.default_rng(5)
    cases = [rng.permutation(300000)[:120000].astype(np.int32),
             rng.integers(0, 50, 5000).astype(np.int32),
             np.array([7, 3], np.int32), np.array([1 << 20, 0, 5, 5], np.int32),
             rng.permutation(2_000_000)[:600000].astype(np.int32)]
    for a in cases:
        t = torch.from_numpy(a.copy()).cuda()
        _lib.check(L.o3dmi_sort_indices(_lib.ptr(t), a.size, stream()), "sort")
        assert np.array_equal(t.cpu().numpy(), np.sort(a))
    bad = torch.tensor([3, -1, 2], dtype=torch.int32, device="cuda")
    assert L.o3dmi_sort_indices(_lib.ptr(bad), 3, stream()) != 0
