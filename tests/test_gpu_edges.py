"""Edges the earlier suites stopped short of:
(a) the run merge is an N-way max (/root/reference/src/3-capture_decode.py:48 MAX_NB_RUNS, :95-96 np.max over the runs' codes): 4 and 8 runs
    (SLGC_MAX_RUNS = 8) through the fused kernel, the two-kernel path, the host path and codes_to_pixels against the oracle; 9 runs refused everywhere;
(b) a stack of 4 GB and more (12 000 x 8 000 x 46 generated on the device): 32-bit buffer offsets no longer reach, slgc_decode_dev takes the
    wide-offset kernel -- bands of it against the oracle;
(c) frame counts outside 14..65 (N = 66: code length 16 does not fit the int16 maps' 15 bits; N = 13: the reference's own frame indices
    run off the stack, decode_codes.py:109-111) refused by every entry point that takes a frame count."""
import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

if conftest.has_gpu():
    import oracle_c as oc
    import oracle_np as onp
    import bench
    from scanner import _native

ERR = (ValueError, RuntimeError)          # SlgcError is a RuntimeError; SLGC_EINVAL surfaces as ValueError


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("N,W,H", [(44, 256, 64), (26, 132, 40)])          # a specialised frame count and the generic kernel
@pytest.mark.parametrize("n_runs", [4, 8])
def test_many_runs_are_an_n_way_max(ctx, N, W, H, n_runs):
    px = W * H
    pw, ph = 160, 120
    calib = bench.calibration(W, H, pw, ph)
    ctx.set_calibration(*calib)
    # runs that differ: independent noise, and every run but one loses a block of pixels (black frames) -- the merged code survives through the others
    runs = np.stack([onp.synth_scene_int(N, H, W, seed=50 + r, noise=2 + r)[0] for r in range(n_runs)])
    for r in range(n_runs):
        if r != 1:
            runs[r, :, :, 8 * r:8 * r + 8] = 0
    ref_h, ref_v, ref_xyz = oc.scan_dense(runs, (pw, ph), *calib)
    single_h, _ = oc.decode(runs[0])
    assert not np.array_equal(single_h, ref_h)                           # the merge matters on this input
    stack = ctx.alloc(runs.nbytes).upload(runs)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    for mode, want_path in ((_native.TRI_ALGEBRAIC, "fused"), (_native.TRI_ALGEBRAIC | _native.TRI_SPLIT, "split")):
        maps.zero()
        xyz.zero()
        ctx.scan_dev(stack.ptr, n_runs, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
        ctx.synchronize()
        assert ctx.last_scan_path()["path"] == want_path
        gh, gv = maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2)
        assert np.array_equal(gh, ref_h) and np.array_equal(gv, ref_v), (mode, n_runs)
        gx = xyz.download((H, W, 3), np.float32)
        ok = (ref_h != -1) & (ref_v != -1)
        fin = np.isfinite(np.moveaxis(ref_xyz, 0, -1)).all(axis=2) & ok
        np.testing.assert_allclose(gx[fin], np.moveaxis(ref_xyz, 0, -1)[fin], rtol=1e-4, atol=0)       # BASELINE.json: XYZ within 1e-4 relative
        assert np.isnan(gx[~ok]).all()
    # host path: decode() on a list of runs (uint8 and the reference's float64), and the merge alone through codes_to_pixels
    for dt in (np.uint8, np.float64):
        hp, vp = ctx.decode([r.astype(dt) for r in runs])
        assert hp.dtype == np.int64 and np.array_equal(hp, ref_h) and np.array_equal(vp, ref_v)
    codes = [ctx.codes(r) for r in runs]
    hp, vp = ctx.codes_to_pixels(np.stack([c[0] for c in codes]), np.stack([c[1] for c in codes]))
    assert np.array_equal(hp, ref_h) and np.array_equal(vp, ref_v)
    for b in (stack, maps, xyz):
        b.free()


def test_many_runs_against_the_reference_generated_goldens(ctx):
    """tests/golden/multirun.npz (the reference's own code run on 4 / 8 runs): device-resident decode, host decode and the code-plane merge."""
    cases = conftest.load_cases("multirun.npz")
    for name, c in cases.items():
        runs = c["stacks"]
        R, N, H, W = runs.shape
        px = H * W
        hp, vp = ctx.decode(list(runs))
        assert np.array_equal(hp, c["merged_h_pixels"]) and np.array_equal(vp, c["merged_v_pixels"]), name
        hp, vp = ctx.decode([r.astype(np.float64) for r in runs])
        assert np.array_equal(hp, c["merged_h_pixels"]) and np.array_equal(vp, c["merged_v_pixels"]), name
        codes = [ctx.codes(r) for r in runs]
        hp, vp = ctx.codes_to_pixels(np.stack([x[0] for x in codes]), np.stack([x[1] for x in codes]))
        assert np.array_equal(hp, c["merged_h_pixels"]) and np.array_equal(vp, c["merged_v_pixels"]), name
        stack, maps = ctx.alloc(runs.nbytes).upload(runs), ctx.alloc(px * 4 + 64)
        ctx.decode_dev(stack.ptr, R, N * px, px, N, H, W, maps.at(0), maps.at(px * 2))
        ctx.synchronize()
        assert np.array_equal(maps.download((H, W), np.int16), c["merged_h_pixels"]) and np.array_equal(maps.download((H, W), np.int16, px * 2), c["merged_v_pixels"]), name
        stack.free()
        maps.free()


def test_nine_runs_are_refused_everywhere(ctx):
    N, W, H = 44, 64, 32
    px = W * H
    ctx.set_calibration(*bench.calibration(W, H, 160, 120))
    runs = np.stack([onp.synth_scene_int(N, H, W, seed=3)[0]] * 9)
    stack = ctx.alloc(runs.nbytes).upload(runs)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    lists = ctx.alloc_cloud_lists(px, colors=False)
    with pytest.raises(ERR, match="n_runs"):
        ctx.decode_dev(stack.ptr, 9, N * px, px, N, H, W, maps.at(0), maps.at(px * 2))
    for mode in (_native.TRI_ALGEBRAIC, _native.TRI_ALGEBRAIC | _native.TRI_SPLIT, _native.TRI_EXACT):
        with pytest.raises(ERR, match="n_runs"):
            ctx.scan_dev(stack.ptr, 9, N * px, px, N, H, W, 0, (160, 120), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
    with pytest.raises(ERR, match="n_runs"):
        ctx.cloud_dev(stack.ptr, 9, N * px, px, N, H, W, (160, 120), None, lists)
    with pytest.raises(ERR, match="n_runs"):
        ctx.decode(list(runs))
    with pytest.raises(ERR, match="n_runs"):
        ctx.pipeline(list(runs), (160, 120))
    # the merge of int8 code planes itself takes any number of runs (np.max over them, src/3-capture_decode.py:95-96): only the stack paths hold 8
    hc, vc = ctx.codes(runs[0])
    hp, vp = ctx.codes_to_pixels(np.stack([hc] * 9), np.stack([vc] * 9))
    assert np.array_equal(hp, oc.decode(runs[0])[0]) and np.array_equal(vp, oc.decode(runs[0])[1])
    with pytest.raises(ERR):
        ctx.decode_dev(stack.ptr, 0, N * px, px, N, H, W, maps.at(0), maps.at(px * 2))       # ... and none at all
    # 8 of the same run = the run itself
    ctx.decode_dev(stack.ptr, 8, N * px, px, N, H, W, maps.at(0), maps.at(px * 2))
    ctx.synchronize()
    assert np.array_equal(maps.download((H, W), np.int16), oc.decode(runs[0])[0])


@pytest.mark.parametrize("N", [66, 13, 0, -4, 1000])
def test_frame_counts_outside_14_to_65_are_refused_by_every_entry_point(ctx, N):
    W, H = 32, 16
    px = W * H
    ctx.set_calibration(*bench.calibration(W, H, 160, 120))
    n_alloc = max(N, 14)
    host = np.zeros((n_alloc, H, W), np.uint8)[:max(N, 0)]
    dev = ctx.alloc(max(n_alloc, 70) * px * 3)
    maps, xyz = ctx.alloc(px * 4 + 64), ctx.alloc(px * 12 + 64)
    lists = ctx.alloc_cloud_lists(px, colors=False)
    L = _native.lib()
    z = np.zeros((H, W))
    hc = np.zeros((16, H, W), np.int8)
    calls = {
        "direct_indirect": lambda: L.slgc_direct_indirect(ctx._h, host.ctypes.data, 0, N, H, W, z.ctypes.data, z.ctypes.data),
        "is_lit": lambda: L.slgc_is_lit(ctx._h, host.ctypes.data, 0, N, H, W, z.ctypes.data, z.ctypes.data, 1.0, 10.0, hc.ctypes.data, hc.ctypes.data),
        "codes": lambda: L.slgc_codes(ctx._h, host.ctypes.data, 0, N, H, W, 1.0, 10.0, hc.ctypes.data, hc.ctypes.data),
        "decode": lambda: L.slgc_decode(ctx._h, (_native.C.c_void_p * 1)(host.ctypes.data), 0, 1, N, H, W, 1.0, 10.0, z.ctypes.data, z.ctypes.data),
        "pipeline_count": lambda: L.slgc_pipeline_count(ctx._h, (_native.C.c_void_p * 1)(host.ctypes.data), 0, 1, N, H, W, 1.0, 10.0, 160, 120, None, 0, 0, float("nan"),
                                                        _native.C.byref(_native.C.c_int64())),
        "decode_dev": lambda: L.slgc_decode_dev(ctx._h, dev.ptr, 1, 0, px, N, H, W, 1.0, 10.0, maps.at(0), maps.at(px * 2), 0),
        "scan_dev": lambda: L.slgc_scan_dev(ctx._h, dev.ptr, 1, 0, px, N, H, W, 0, 160, 120, 1.0, 10.0, 1, maps.at(0), maps.at(px * 2), xyz.ptr, None),
        "scan_dev split": lambda: L.slgc_scan_dev(ctx._h, dev.ptr, 1, 0, px, N, H, W, 0, 160, 120, 1.0, 10.0, 1 | 4, maps.at(0), maps.at(px * 2), xyz.ptr, None),
        "scan_bgr_dev": lambda: L.slgc_scan_bgr_dev(ctx._h, dev.ptr, 1, 0, 3 * px, N, H, W, 0, 160, 120, 15, 1.0, 10.0, 1, maps.at(0), maps.at(px * 2), xyz.ptr, None),
        "decode_bgr_dev": lambda: L.slgc_decode_bgr_dev(ctx._h, dev.ptr, 1, 0, 3 * px, N, H, W, 15, 1.0, 10.0, maps.at(0), maps.at(px * 2)),
        "scan_batch_dev": lambda: L.slgc_scan_batch_dev(ctx._h, dev.ptr, 2, 70 * px, px, N, H, W, 0, 160, 120, 1.0, 10.0, 1, maps.at(0), maps.at(px * 2), xyz.ptr),
        "cloud_dev": lambda: L.slgc_cloud_dev(ctx._h, dev.ptr, 1, 0, px, N, H, W, 160, 120, 1.0, 10.0, None, None, None, lists.cam.ptr, lists.proj.ptr, lists.pts.ptr, None,
                                              lists.count.ptr),
    }
    for name, call in calls.items():
        rc = call()
        assert rc == -1, f"{name} with N = {N}: status {rc}, expected SLGC_EINVAL (-1)"
        assert L.slgc_last_error(ctx._h), name
    # ... and the context still works
    st, _, _ = onp.synth_scene_int(26, H, W, seed=2)
    hp, vp = ctx.decode(st)
    assert np.array_equal(hp, oc.decode(st)[0])


def test_stack_of_4gb_and_more_takes_the_wide_offset_kernel(ctx):
    """12 000 x 8 000 x 46 = 4.42 GB: past what a 32-bit buffer offset reaches, so slgc_decode_dev leaves the packed kernel for the lane-mask
    kernel with 64-bit addressing (decode.hip: fits32 == false) -- at scale, on a device-generated capture; three bands of the maps against the
    oracle's decode of the same bytes (the decode is per pixel: a band of the stack decodes to the band of the maps)."""
    W, H, N = 12000, 8000, 46
    px = W * H
    assert N * px >= 1 << 32
    stack = ctx.alloc(N * px)
    ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=9, noise=3, shadow=True)
    maps = ctx.alloc(px * 4).zero()
    ctx.decode_dev(stack.ptr, 1, N * px, px, N, H, W, maps.at(0), maps.at(px * 2))
    ctx.synchronize()
    assert ctx.last_scan_path()["fallback_kernels"]["decode"]              # the wide-offset (lane-mask) kernel ran, not the packed one
    total_valid = 0
    for y0, rows in ((0, 6), (3997, 7), (H - 5, 5)):
        band = np.stack([stack.download((rows, W), np.uint8, byte_offset=f * px + y0 * W) for f in range(N)])
        ref_h, ref_v = oc.decode(band)
        gh = maps.download((rows, W), np.int16, byte_offset=y0 * W * 2)
        gv = maps.download((rows, W), np.int16, byte_offset=px * 2 + y0 * W * 2)
        assert np.array_equal(gh, ref_h) and np.array_equal(gv, ref_v), y0
        total_valid += int(((ref_h != -1) & (ref_v != -1)).sum())
    assert total_valid > 50_000
    stack.free()
    maps.free()


@pytest.mark.parametrize("n_cam,n_proj", [(0, 0), (4, 5), (5, 8), (8, 12), (12, 4), (14, 14)])
def test_every_distortion_vector_length_the_abi_accepts(ctx, n_cam, n_proj):
    """slgc_set_calibration takes k1,k2,p1,p2[,k3[,k4,k5,k6[,s1..s4[,tauX,tauY]]]] like cv2.undistortPoints (the reference's files hold 5, as 5x1 and
    1x5: triangulate.py:84-85): rational (k4..k6) and thin-prism (s1..s4) terms against both oracles' restatement, through slgc_undistort_points,
    slgc_triangulate and the dense scan; zero tilt accepted, non-zero tilt refused."""
    rng = np.random.default_rng(100 * n_cam + n_proj)
    full = np.array([-0.12, 0.09, 0.002, -0.003, -0.03, 0.05, -0.04, 0.02, 0.004, -0.002, 0.003, 0.001, 0.0, 0.0])
    cd, pd = full[:n_cam] * rng.uniform(0.5, 1.5, n_cam), full[:n_proj] * rng.uniform(0.5, 1.5, n_proj)
    W, H, pw, ph = 320, 200, 256, 160
    K = np.array([[300.0, 0, W / 2], [0, 300.0, H / 2], [0, 0, 1]])
    pk = np.array([[280.0, 0, pw / 2], [0, 280.0, ph / 2], [0, 0, 1]])
    th = np.deg2rad(-18.0)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T = np.array([[0.22], [0.01], [0.03]])
    ctx.set_calibration(K, cd, pk, pd, R, T)
    M = 5000
    cam = np.stack([rng.integers(0, W, M), rng.integers(0, H, M)], 1).astype(np.float32)
    proj = np.stack([rng.integers(0, pw, M), rng.integers(0, ph, M)], 1).astype(np.float32)
    for which, pts, Kx, dx, Rx in ((False, cam, K, cd, R), (True, proj, pk, pd, None)):
        got = ctx.undistort_points(pts, projector=which)
        ref_np = onp.undistort_points(pts, Kx, dx, R=Rx).reshape(-1, 2)
        ref_c = oc.undistort(pts, Kx, dx, R=Rx)
        assert np.array_equal(ref_np, ref_c)                              # the two oracles agree bit for bit (float32 out)
        np.testing.assert_allclose(got, ref_np, rtol=2e-6, atol=1e-7)
    want = oc.triangulate(cam, proj, K, cd, pk, pd, R, T)
    got = ctx.triangulate(cam, proj)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-4, atol=0)
    # dense scan (ray tables built from the same calibration)
    N = 26
    st, _, _ = onp.synth_scene_int(N, H, W, seed=4)
    rh, rv, rx = oc.scan_dense(st, (pw, ph), K, cd, pk, pd, R, T)
    px = W * H
    stack, maps, xyz = ctx.alloc(st.nbytes).upload(st), ctx.alloc(px * 4), ctx.alloc(px * 12)
    ctx.scan_dev(stack.ptr, 1, st.nbytes, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2))
    ctx.synchronize()
    gx = xyz.download((H, W, 3), np.float32)
    ok = (rh != -1) & (rv != -1) & np.isfinite(np.moveaxis(rx, 0, -1)).all(axis=2)
    assert np.array_equal(maps.download((H, W), np.int16), rh) and ok.sum() > 1000
    np.testing.assert_allclose(gx[ok], np.moveaxis(rx, 0, -1)[ok], rtol=1e-4, atol=0)
    for b in (stack, maps, xyz):
        b.free()
    if n_cam == 14:                                                       # tilted-sensor terms are not supported: refused, not ignored
        bad = cd.copy()
        bad[12] = 0.01
        with pytest.raises(ERR, match="tilt"):
            ctx.set_calibration(K, bad, pk, pd, R, T)
        with pytest.raises(ERR):
            ctx.set_calibration(K, np.zeros(15), pk, pd, R, T)
