"""Exactly what bench.py times, at BASELINE.json's full sizes, EVERY pixel against the CPU oracle (-m gpu).

configs[2] (4096x3000x44, 1920x1200 projector) and configs[1] (1920x1080x44, 1920x1080 projector) with bench.py's own calibration
and bench.py's own device-generated stack (seed 1, noise 3, shadow) go through the call bench.py makes in its timed region --
``slgc_scan_dev`` with no count, caller-provided int16 maps, algebraic mode = ONE fused kernel -- and through the two-kernel
pipeline and a two-run merge.  The oracle (oracle/slgc_oracle.c, all host cores) decodes and triangulates the downloaded stack:
maps bit-exact, finite mask identical, XYZ within 1e-4 elementwise (reference: decode_codes.py:90-229, src/3-capture_decode.py:95-100,
triangulate.py:56-61,84-95).  configs[4] (16 independent 1920x1080x44 scans, replicas only) runs through bench.py's own lanes.
"""
import os

import numpy as np
import pytest

import bench
import oracle_c as oc
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]

XYZ_RTOL = 1e-4   # BASELINE.json: triangulated XYZ within 1e-4 relative


@pytest.fixture(scope="module")
def ctx():
    from scanner import _native
    c = _native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", autouse=True)
def oracle_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    oc.set_threads(max(1, min(64, n)))
    yield
    oc.set_threads(1)


def flat_mask_np(h, v, proj_size, K, cd, pk, pd, R, T, amp=60.0):
    """NumPy twin of tri_is_flat (csrc/tri_math.h, amp = kGuardAmp) on the oracle's float32 rays: which decodable pixels take the
    guarded path."""
    H, W = h.shape
    ok = (h != -1) & (v != -1)
    yy, xx = np.nonzero(ok)
    cam = oc.undistort(np.stack([xx, yy], 1).astype(np.float32), K, cd, R).astype(np.float64)
    pu = np.minimum(proj_size[0] - 1, h[ok])
    pv = np.minimum(proj_size[1] - 1, v[ok])
    prj = oc.undistort(np.stack([pu, pv], 1).astype(np.float32), pk, pd).astype(np.float64)
    t = np.asarray(T, dtype=np.float64).reshape(3)
    tl2 = float(t @ t)
    A = -(t[0] * cam[:, 0] + t[1] * cam[:, 1] + t[2])
    B = t[0] * prj[:, 0] + t[1] * prj[:, 1] + t[2]
    ta = tl2 * (cam[:, 0] ** 2 + cam[:, 1] ** 2 + 1.0)
    tb = tl2 * (prj[:, 0] ** 2 + prj[:, 1] ** 2 + 1.0)
    ra, rb = ta - A * A, tb - B * B
    D = np.sqrt(np.maximum(ra, 0)) * B + A * np.sqrt(np.maximum(rb, 0))
    flat = (rb < tb / amp) | (D * D * np.minimum(ra * tb, rb * ta) < (2.0 / amp) ** 2 * (ta * tb) ** 2)
    out = np.zeros((H, W), bool)
    out[yy, xx] = flat
    return out


def compare_scan(got_h, got_v, got_xyz, ref_h, ref_v, ref_xyz, what):
    assert np.array_equal(got_h, ref_h), f"{what}: h map differs in {(got_h != ref_h).sum()} pixels"
    assert np.array_equal(got_v, ref_v), f"{what}: v map differs in {(got_v != ref_v).sum()} pixels"
    ok = (ref_h != -1) & (ref_v != -1)
    fin = np.isfinite(got_xyz).all(axis=2)
    assert np.isnan(got_xyz[~ok]).all(), f"{what}: undecodable pixels must be NaN"
    ref = np.moveaxis(ref_xyz, 0, -1)
    rfin = np.isfinite(ref).all(axis=2)
    assert np.array_equal(fin & ok, rfin & ok), f"{what}: finite mask differs in {((fin & ok) != (rfin & ok)).sum()} pixels"
    m = ok & rfin
    err = np.abs(got_xyz[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), 1e-300)
    worst = float(err.max()) if err.size else 0.0
    assert worst <= XYZ_RTOL, f"{what}: worst elementwise relative XYZ error {worst:.3e} > {XYZ_RTOL}"
    return int(ok.sum()), worst


@pytest.mark.parametrize("workload,expect_valid", [("c3_4096x3000x44", 9_924_736), ("c2_1920x1080x44", None), ("c3_4096x3000x46", None), ("c2_1920x1080x46", None)])
def test_bench_configuration_every_pixel(ctx, workload, expect_valid):
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS[workload]
    calib = bench.calibration(W, H, pw, ph)
    ctx.set_calibration(*calib)
    px = W * H
    stack = ctx.alloc(N * px)
    ctx.synth_scene_dev(stack.ptr, px, N, H, W, row0=0, rows=H, seed=1, noise=3, shadow=True)     # bench.py's stacks[0]
    maps, xyz, cnt = ctx.alloc(px * 4), ctx.alloc(px * 12), ctx.alloc(16)
    ctx.synchronize()
    st = stack.download((N, H, W), np.uint8)
    ref_h, ref_v, ref_xyz = oc.scan_dense(st, (pw, ph), *calib)

    def run(mode):
        maps.zero()
        xyz.zero()
        ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
        ctx.synchronize()
        return (maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), xyz.download((H, W, 3), np.float32))

    # (1) the timed region of bench.py: fused kernel, algebraic (guarded) form, no count -- with the camera rays from the per-pixel
    # table (cam_nodes 0), interpolated from the every-4th-column table (2) and as bench.py runs it (1: node table above 12 MB of rays)
    worst_fused = worst_split = 0.0
    for nodes in (0, 2, 1):
        ctx.tune("cam_nodes", nodes)
        valid, w = compare_scan(*run(_native.TRI_ALGEBRAIC), ref_h, ref_v, ref_xyz, f"{workload} fused cam_nodes={nodes}")
        worst_fused = max(worst_fused, w)
        if expect_valid is not None:
            assert valid == expect_valid
        # (2) bench.py's "split_pipeline": decode kernel + dense triangulation kernel
        _, w = compare_scan(*run(_native.TRI_ALGEBRAIC | _native.TRI_SPLIT), ref_h, ref_v, ref_xyz, f"{workload} split cam_nodes={nodes}")
        worst_split = max(worst_split, w)
        in_use, node_err = ctx.ray_table_info()
        assert in_use == (nodes == 2 or (nodes == 1 and W * H * 8 > 12 << 20)) and 0.0 <= node_err <= 2.4e-7
    # (3) the exact (acos / sin) dense kernel on the same maps
    _, worst_exact = compare_scan(*run(_native.TRI_EXACT), ref_h, ref_v, ref_xyz, workload + " exact")
    assert worst_exact < 1e-6
    # guard path: how many pixels the fused / dense kernels redo on the reference's float32 intermediates
    cnt.zero()
    ctx.guard_count_dev(maps.at(0), maps.at(px * 2), H, W, 0, (pw, ph), cnt.ptr)
    ctx.synchronize()
    n_ok, n_flat = (int(x) for x in cnt.download((2,), np.uint64))
    assert n_ok == valid
    n_ref = int(flat_mask_np(ref_h, ref_v, (pw, ph), *calib).sum())
    assert abs(n_flat - n_ref) <= 2 + n_ref // 500          # same test on the same rays; float32 rounding moves pixels sitting on the threshold
    print(f"\n{workload}: {valid} / {px} decodable, {n_flat} on the guarded path ({100.0 * n_flat / max(valid, 1):.3f} %), worst rel. XYZ error "
          f"fused {worst_fused:.2e} split {worst_split:.2e} exact {worst_exact:.2e}")
    for b in (stack, maps, xyz, cnt):
        b.free()


@pytest.mark.parametrize("cam", ["cam_1080", "cam_1440"])
def test_node_table_on_the_reference_lenses(ctx, cam):
    """The two camera models the reference ships (calibration_results/cam_1080, cam_1440 -- the second one strongly distorted, k1 = -0.51), at
    their native sizes, with the scan kernels forced onto the every-4th-column node table: every pixel against the oracle like the bench
    configurations, and the table's measured error inside its acceptance limit (or the table dropped, which the test reports)."""
    from scanner import _native
    from scanner import reference_calibration as rc
    K, cd, (W, H) = {"cam_1080": (rc.CAM_MTX, rc.CAM_DIST, (1920, 1080)), "cam_1440": (rc.CAM1440_MTX, rc.CAM1440_DIST, (2560, 1440))}[cam]
    pw, ph, N = 1920, 1080, 44
    _, _, pk, pd, R, T = bench.calibration(1920, 1080, pw, ph)
    calib = (K, cd, pk, pd, R, T)
    ctx.set_calibration(*calib)
    px = W * H
    stack = ctx.alloc(N * px)
    ctx.synth_scene_dev(stack.ptr, px, N, H, W, row0=0, rows=H, seed=2, noise=3, shadow=True)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    ctx.synchronize()
    ref_h, ref_v, ref_xyz = oc.scan_dense(stack.download((N, H, W), np.uint8), (pw, ph), *calib)
    worst = {}
    for nodes in (2, 0):
        ctx.tune("cam_nodes", nodes)
        for name, mode in (("fused", _native.TRI_ALGEBRAIC), ("split", _native.TRI_ALGEBRAIC | _native.TRI_SPLIT)):
            maps.zero()
            xyz.zero()
            ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
            ctx.synchronize()
            got = (maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), xyz.download((H, W, 3), np.float32))
            _, worst[(nodes, name)] = compare_scan(*got, ref_h, ref_v, ref_xyz, f"{cam} {name} cam_nodes={nodes}")
        if nodes == 2:
            in_use, err = ctx.ray_table_info()
            assert (in_use and 0.0 <= err <= 2.4e-7) or (not in_use and err > 2.4e-7)
            print(f"\n{cam} {W}x{H}: node table {'in use' if in_use else 'dropped'}, error measure {err:.3g} (limit 2.4e-7)")
    ctx.tune("cam_nodes", 1)
    print({k: f"{v:.2e}" for k, v in worst.items()})
    for b in (stack, maps, xyz):
        b.free()


@pytest.mark.parametrize("workload", ["c3_4096x3000x44", "c3_4096x3000x46"])
def test_bench_configuration_two_runs_every_pixel(ctx, workload):
    """src/3-capture_decode.py:95-96 (MAX_NB_RUNS = 2) at 4096x3000x44 / x46: two captures max-merged per code bit inside the fused kernel."""
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS[workload]
    calib = bench.calibration(W, H, pw, ph)
    ctx.set_calibration(*calib)
    px = W * H
    stack = ctx.alloc(2 * N * px)
    ctx.synth_scene_dev(stack.at(0), px, N, H, W, seed=1, noise=3, shadow=True)
    ctx.synth_scene_dev(stack.at(N * px), px, N, H, W, seed=2, noise=9, shadow=False)    # noisier second capture, no shadow: the merge matters
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    ctx.scan_dev(stack.ptr, 2, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=_native.TRI_ALGEBRAIC)
    ctx.synchronize()
    st = stack.download((2, N, H, W), np.uint8)
    ref_h, ref_v, ref_xyz = oc.scan_dense(st, (pw, ph), *calib)
    single_h, _ = oc.decode(st[0])
    assert (ref_h != single_h).sum() > 1000                                                # the second run really changes the result
    valid, worst = compare_scan(maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), xyz.download((H, W, 3), np.float32),
                                ref_h, ref_v, ref_xyz, "two runs fused")
    print(f"\ntwo runs: {valid} / {px} decodable, worst rel. XYZ error {worst:.2e}")
    for b in (stack, maps, xyz):
        b.free()


@pytest.mark.parametrize("scene", ["physical", "s-scene"])
def test_throughput_mode_scans_match_oracle(ctx, scene):
    """BASELINE.json configs[4]: bench.py's own lanes (contexts / streams / rotated stacks), one step of 16 scans; the results of two
    of the scans (one per lane) are compared with the oracle, every pixel."""
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS["c2_1920x1080x44"]
    calib = bench.calibration(W, H, pw, ph, rig=bench.SCENES[scene]["rig"])
    px = W * H
    lanes = bench.throughput_lanes(_native, 0, 16, 2, scene)
    try:
        plan = bench.throughput_step(lanes, 16, 0, _native.TRI_ALGEBRAIC)                 # [(lane index, stack index)] in issue order
        for c, _, _, _ in lanes:
            c.synchronize()
        assert len(plan) == 16
        checked = set()
        for li, si in reversed(plan):                     # the LAST scan issued on each lane is what its output buffers hold
            if li in checked:
                continue
            checked.add(li)
            c, stacks, maps, xyz = lanes[li]
            st = stacks[si].download((N, H, W), np.uint8)
            ref_h, ref_v, ref_xyz = oc.scan_dense(st, (pw, ph), *calib)
            compare_scan(maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), xyz.download((H, W, 3), np.float32),
                         ref_h, ref_v, ref_xyz, f"throughput lane {li} stack {si}")
        assert len(checked) == 2
    finally:
        for c, _, _, _ in lanes:
            c.close()


@pytest.mark.parametrize("shape", [(1920, 1080, 44, 5), (1024, 96, 26, 3), (130, 33, 26, 3)])
def test_batched_scans_equal_single_scans(ctx, shape):
    """slgc_scan_batch_dev (BASELINE configs[4] in one launch): every scan of the batch bit-identical to slgc_scan_dev on the same stack, for a
    shape that batches (a whole number of 512-pixel workgroups per scan), one with a single workgroup row ... and one that falls back."""
    from scanner import _native
    W, H, N, B = shape
    pw, ph = 1920, 1080
    ctx.set_calibration(*bench.calibration(1920, 1080, pw, ph))
    px = W * H
    stacks = ctx.alloc(B * N * px)
    for s in range(B):
        ctx.synth_scene_dev(stacks.at(s * N * px), px, N, H + 13 * B, W, row0=13 * s, rows=H, seed=40 + s, noise=3, shadow=True)   # a different window of one tall scene per scan
    bh, bv, bx = ctx.alloc(B * px * 2), ctx.alloc(B * px * 2), ctx.alloc(B * px * 12)
    ctx.scan_batch_dev(stacks.ptr, B, N * px, px, N, H, W, 0, (pw, ph), bx.ptr, bh.ptr, bv.ptr)
    ctx.synchronize()
    got_h, got_v, got_x = bh.download((B, H, W), np.int16), bv.download((B, H, W), np.int16), bx.download((B, H, W, 3), np.float32)
    maps, xyz = ctx.alloc(px * 4 + 64), ctx.alloc(px * 12)
    voff = (px * 2 + 31) // 32 * 32
    for s in range(B):
        ctx.scan_dev(stacks.at(s * N * px), 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(voff), mode=_native.TRI_ALGEBRAIC)
        ctx.synchronize()
        assert np.array_equal(got_h[s], maps.download((H, W), np.int16)) and np.array_equal(got_v[s], maps.download((H, W), np.int16, voff)), s
        one = xyz.download((H, W, 3), np.float32)
        if px % 512 == 0:                                                    # one launch: the same kernel on the same bytes
            assert np.array_equal(got_x[s].view(np.uint32), one.view(np.uint32)), s
        else:                                                                # scan after scan at other buffer alignments: the ragged-shape kernels
            assert np.array_equal(np.isnan(got_x[s]), np.isnan(one)), s
            np.testing.assert_allclose(got_x[s], one, rtol=1e-5, atol=0, equal_nan=True)
    assert len({got_h[s].tobytes() for s in range(B)}) == B                  # the scans differ from each other
    if (W, H) == (1920, 1080):                                               # and one of them against the oracle
        st = stacks.download((N, H, W), np.uint8, 2 * N * px)
        compare_scan(got_h[2], got_v[2], got_x[2], *oc.scan_dense(st, (pw, ph), *bench.calibration(1920, 1080, pw, ph)), "batched scan 2")
    for b in (stacks, bh, bv, bx, maps, xyz):
        b.free()


@pytest.mark.parametrize("workload", ["c3_4096x3000x44", None])
def test_device_resident_reference_product(ctx, workload):
    """slgc_cloud_lists_dev: the x-major lists of get_cam_proj_pts (triangulate.py:52-71), the colour gather from a device-resident
    white image and the float64 (3,M) point array (:95), built in HBM from the fused scan's maps + dense XYZ -- vs the oracle."""
    from scanner import _native
    rng = np.random.default_rng(5)
    if workload:
        W, H, pw, ph, N = bench.WORKLOADS[workload]
        calib = bench.calibration(W, H, pw, ph)
    else:                                                    # ragged: W not a multiple of the 64-column tile, H not of the 32-row chunk
        W, H, pw, ph, N = 332, 77, 300, 200, 44
        K = np.array([[300.0, 0, W / 2], [0, 300.0, H / 2], [0, 0, 1]])
        _, cd, pk, pd, R, T = bench.calibration(1920, 1080, pw, ph)
        calib = (K, cd, pk, pd, R, T)
    ctx.set_calibration(*calib)
    px = W * H
    stack = ctx.alloc(N * px)
    ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=3, noise=3, shadow=True)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    white_h = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    white = ctx.alloc(px * 3).upload(white_h)
    lists = ctx.alloc_cloud_lists(px, colors=True)
    ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=_native.TRI_ALGEBRAIC)
    ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), xyz.ptr, white.ptr, W, H, (pw, ph), lists)
    cam, proj, pts, col = lists.download()
    h = maps.download((H, W), np.int16).astype(np.int64)
    v = maps.download((H, W), np.int16, px * 2).astype(np.int64)
    dense = xyz.download((H, W, 3), np.float32)
    rcam, rproj, rcol = oc.cam_proj_pts(h, v, (W, H), (pw, ph), white_h, order="x")
    assert len(rcam) > 0.5 * px
    assert np.array_equal(cam, rcam) and np.array_equal(proj, rproj) and np.array_equal(col, rcol)
    xs, ys = rcam[:, 0].astype(np.int64), rcam[:, 1].astype(np.int64)
    assert pts.shape == (3, len(rcam)) and np.array_equal(pts, dense[ys, xs].astype(np.float64).T)
    # lists without points / colours
    bare = ctx.alloc_cloud_lists(px, colors=False, points=False)
    ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), None, None, W, H, (pw, ph), bare)
    c2, p2, none_pts, none_col = bare.download()
    assert none_pts is None and none_col is None and np.array_equal(c2, rcam) and np.array_equal(p2, rproj)
    for b in (stack, maps, xyz, white):
        b.free()
    lists.free()
    bare.free()


@pytest.mark.parametrize("workload", ["c1_1280x720x42", "c2_1920x1080x44", "c3_4096x3000x44"])
def test_executed_path_is_observable(ctx, workload):
    """slgc_last_scan_path: the call bench.py times runs the fused kernel specialised for the workload's frame count at the three BASELINE
    sizes; a count request, a split request, the exact mode and a misaligned XYZ buffer take the two-kernel path -- and say so."""
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS[workload]
    ctx.set_calibration(*bench.calibration(W, H, pw, ph))
    ctx.tune("cam_nodes", 1)
    px = W * H
    stack = ctx.alloc(N * px)
    ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=1, noise=3, shadow=True)
    maps, xyz, cnt = ctx.alloc(px * 4), ctx.alloc(px * 12 + 16), ctx.alloc(16).zero()

    def scan(mode, count=None, xyz_ptr=None):
        ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz_ptr or xyz.ptr, count, maps.at(0), maps.at(px * 2), mode=mode)
        return ctx.last_scan_path()

    big = W * H * 8 > 12 << 20
    clean = {"decode": False, "triangulation": False}
    assert scan(_native.TRI_ALGEBRAIC) == {"path": "fused", "ns_frames": N, "node_table": big, "guard": True, "fallback_kernels": clean}
    assert scan(_native.TRI_ALGEBRAIC | _native.TRI_SPLIT) == {"path": "split", "ns_frames": N, "node_table": big, "guard": True, "fallback_kernels": clean}
    assert scan(_native.TRI_ALGEBRAIC, count=cnt.ptr)["path"] == "split"
    assert scan(_native.TRI_EXACT) == {"path": "split", "ns_frames": N, "node_table": False, "guard": False, "fallback_kernels": clean}
    ragged = scan(_native.TRI_ALGEBRAIC, xyz_ptr=xyz.ptr + 4)                                   # XYZ not 16-byte aligned: per-pixel triangulation kernel
    assert ragged["path"] == "split-ragged" and ragged["fallback_kernels"] == {"decode": False, "triangulation": True}
    # a triangulation on its own reports only its own fallback -- not the ragged state an earlier, unrelated decode left behind
    ctx.decode_dev(stack.ptr + 1, 1, N * px, px, N, H - 1, W, maps.at(0), maps.at(px * 2))     # misaligned stack: the byte-wide decode kernel
    assert ctx.last_scan_path()["fallback_kernels"]["decode"]
    scan(_native.TRI_ALGEBRAIC)                                                                 # ... an unrelated fused scan in between
    ctx.triangulate_maps_dev(maps.at(0), maps.at(px * 2), H, W, 0, (pw, ph), xyz.ptr)
    alone = ctx.last_scan_path()
    assert alone["path"] == "split" and alone["fallback_kernels"] == clean, alone
    ctx.decode_dev(stack.ptr + 1, 1, N * px, px, N, H - 1, W, maps.at(0), maps.at(px * 2))
    ctx.triangulate_maps_dev(maps.at(0), maps.at(px * 2), H - 1, W, 0, (pw, ph), xyz.ptr)       # directly after a decode: completes that two-kernel scan
    assert ctx.last_scan_path()["path"] == "split-ragged"
    # ... and a SECOND triangulation after it, with no fused scan in between, is on its own again: the decode's bit is gone
    ctx.triangulate_maps_dev(maps.at(0), maps.at(px * 2), H - 1, W, 0, (pw, ph), xyz.ptr)
    again = ctx.last_scan_path()
    assert again["path"] == "split" and again["fallback_kernels"] == clean, again
    # a triangulation through the per-pixel fallback, then a decode alone: the decode reports only its own (clean) bit
    ctx.triangulate_maps_dev(maps.at(0), maps.at(px * 2), H - 1, W, 0, (pw, ph), xyz.ptr + 4)
    assert ctx.last_scan_path()["fallback_kernels"] == {"decode": False, "triangulation": True}
    ctx.decode_dev(stack.ptr, 1, N * px, px, N, H, W, maps.at(0), maps.at(px * 2))
    assert ctx.last_scan_path()["fallback_kernels"] == clean
    ctx.tune("park", 0)
    assert scan(_native.TRI_ALGEBRAIC) == {"path": "fused", "ns_frames": 0, "node_table": big, "guard": True, "fallback_kernels": clean}
    ctx.tune("park", 1)
    ctx.synchronize()
    for b in (stack, maps, xyz, cnt):
        b.free()


@pytest.mark.parametrize("workload", ["c3_4096x3000x44", "c2_1920x1080x44", "c1_1280x720x42"])
def test_scan_without_map_buffers_is_xyz_only(ctx, workload):
    """slgc_scan_dev / slgc_scan_batch_dev with d_h = d_v = NULL: the same XYZ bit for bit as with map buffers, on the fused kernel (which then
    stores no maps: poisoned guard words either side of the XYZ stay put), on the two-kernel path (maps in scratch) and batched."""
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS[workload]
    ctx.set_calibration(*bench.calibration(W, H, pw, ph))
    px = W * H
    stack = ctx.alloc(N * px)
    ctx.synth_physical_dev(stack.ptr, px, N, H, W, (pw, ph), seed=5, noise=2)
    maps, ref, got = ctx.alloc(px * 4), ctx.alloc(px * 12), ctx.alloc(px * 12 + 512)
    for mode in (_native.TRI_ALGEBRAIC, _native.TRI_ALGEBRAIC | _native.TRI_SPLIT, _native.TRI_EXACT):
        ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), ref.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
        with_maps = ctx.last_scan_path()["path"]
        ctx.dev_memset(got.ptr, 0x5a, px * 12 + 512)
        ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), got.ptr + 256, None, None, None, mode=mode)
        assert ctx.last_scan_path()["path"] == with_maps == ("fused" if mode == _native.TRI_ALGEBRAIC else "split")
        ctx.synchronize()
        a, b = ref.download((px, 3), np.float32), got.download((px, 3), np.float32, 256)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), mode
        assert np.isfinite(a[:, 2]).sum() > px // 50
        assert (got.download((256,), np.uint8) == 0x5a).all() and (got.download((256,), np.uint8, 256 + px * 12) == 0x5a).all()
    if px % 512 == 0:
        B = 3
        stacks, bx, bx2, bm = ctx.alloc(B * N * px), ctx.alloc(B * px * 12), ctx.alloc(B * px * 12), ctx.alloc(B * px * 4)
        for s in range(B):
            ctx.synth_physical_dev(stacks.at(s * N * px), px, N, H, W, (pw, ph), seed=9 + s, noise=2)
        ctx.scan_batch_dev(stacks.ptr, B, N * px, px, N, H, W, 0, (pw, ph), bx.ptr, bm.at(0), bm.at(B * px * 2))
        ctx.scan_batch_dev(stacks.ptr, B, N * px, px, N, H, W, 0, (pw, ph), bx2.ptr)
        assert ctx.last_scan_path()["path"] == "batch-fused"
        ctx.synchronize()
        assert np.array_equal(bx.download((B * px * 3,), np.uint32), bx2.download((B * px * 3,), np.uint32))
        for b_ in (stacks, bx, bx2, bm):
            b_.free()
    for b_ in (stack, maps, ref, got):
        b_.free()


@pytest.mark.parametrize("workload", ["c3_4096x3000x44", "c2_1920x1080x44", "ragged", "ragged4"])
def test_cloud_dev_lists_without_dense_xyz(ctx, workload):
    """slgc_cloud_dev: decode kernel, then the x-major list build with the triangulation INSIDE it (no dense XYZ round trip) -- the lists of
    get_cam_proj_pts (triangulate.py:52-71) bit-exact against the oracle, the float64 (3,M) points (:84-95) bit-identical with the fused scan
    kernel's float32 XYZ where that kernel runs (4096x3000: camera rays from the node table; 1920x1080: per-pixel table) and within 1e-4 of the
    oracle's triangulation everywhere, on shapes that are not a multiple of the 64-column tile / 32-row chunk / 4-pixel group too."""
    from scanner import _native
    rng = np.random.default_rng(7)
    if workload.startswith("ragged"):
        W, H, pw, ph, N = (332, 77, 300, 200, 44) if workload == "ragged4" else (331, 45, 300, 200, 26)
        K = np.array([[300.0, 0, W / 2], [0, 300.0, H / 2], [0, 0, 1]])
        _, cd, pk, pd, R, T = bench.calibration(1920, 1080, pw, ph)
        calib = (K, cd, pk, pd, R, T)
    else:
        W, H, pw, ph, N = bench.WORKLOADS[workload]
        calib = bench.calibration(W, H, pw, ph)
    ctx.set_calibration(*calib)
    ctx.tune("cam_nodes", 1)
    px = W * H
    stack = ctx.alloc(N * px)
    ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=3, noise=3, shadow=True)
    maps, maps2, xyz = ctx.alloc(px * 4 + 64), ctx.alloc(px * 4 + 64), ctx.alloc(px * 12)
    voff = (px * 2 + 31) // 32 * 32
    white_h = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    white = ctx.alloc(px * 3).upload(white_h)
    lists = ctx.alloc_cloud_lists(px, colors=True)
    ctx.cloud_dev(stack.ptr, 1, N * px, px, N, H, W, (pw, ph), white.ptr, lists, d_h=maps.at(0), d_v=maps.at(voff))
    path = ctx.last_scan_path()
    assert path["path"] == "cloud" and path["node_table"] == (W * H * 8 > 12 << 20 and W % 4 == 0)
    assert ctx.last_list_kernel() == ("whole-lines" if W * H >= 4 << 20 else "tile-runs")           # the library's own choice: whole lines where they pay
    cam, proj, pts, col = lists.download()
    h = maps.download((H, W), np.int16).astype(np.int64)
    v = maps.download((H, W), np.int16, voff).astype(np.int64)
    st = stack.download((N, H, W), np.uint8)
    ref_h, ref_v, ref_xyz = oc.scan_dense(st, (pw, ph), *calib)
    assert np.array_equal(h, ref_h) and np.array_equal(v, ref_v)
    rcam, rproj, rcol = oc.cam_proj_pts(h, v, (W, H), (pw, ph), white_h, order="x")
    assert len(rcam) > 0.5 * px
    assert np.array_equal(cam, rcam) and np.array_equal(proj, rproj) and np.array_equal(col, rcol)
    xs, ys = rcam[:, 0].astype(np.int64), rcam[:, 1].astype(np.int64)
    want = ref_xyz[:, ys, xs]                                              # (3, M) float64, the oracle's triangulate.py:84-95
    assert pts.shape == want.shape
    fin = np.isfinite(want).all(axis=0)
    assert np.array_equal(np.isfinite(pts).all(axis=0), fin)
    err = np.abs(pts[:, fin] - want[:, fin]) / np.maximum(np.abs(want[:, fin]), 1e-300)
    assert float(err.max()) <= XYZ_RTOL
    # the same points as the fused scan kernel's dense XYZ, bit for bit (both run tri_math.h on the same rays)
    ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps2.at(0), maps2.at(voff), mode=_native.TRI_ALGEBRAIC)
    if ctx.last_scan_path()["path"] == "fused":
        dense = xyz.download((H, W, 3), np.float32)
        assert np.array_equal(pts, dense[ys, xs].astype(np.float64).T)
    else:
        assert workload == "ragged"                                        # 331 x 45 pixels are not a whole number of 4-pixel groups
    # lists only (no points, no colours), maps in the library's workspace
    bare = ctx.alloc_cloud_lists(px, colors=False, points=False)
    ctx.cloud_dev(stack.ptr, 1, N * px, px, N, H, W, (pw, ph), None, bare)
    c2, p2, none_pts, none_col = bare.download()
    assert none_pts is None and none_col is None and np.array_equal(c2, rcam) and np.array_equal(p2, rproj)
    # points + colours only (what src/4-triangulate.py:67-68 keeps): the same arrays, the two correspondence lists never written
    prod = ctx.alloc_cloud_lists(px, colors=True, lists=False)
    ctx.cloud_dev(stack.ptr, 1, N * px, px, N, H, W, (pw, ph), white.ptr, prod)
    n3, n4, pts3, col3 = prod.download()
    assert n3 is None and n4 is None and np.array_equal(pts3, pts) and np.array_equal(col3, col)
    prod.free()
    # slgc_cloud32_dev, the explicitly non-reference float32 product: exactly the float64 arrays rounded to float32, with and without the lists
    for with_lists in (True, False):
        p32 = ctx.alloc_cloud_lists(px, colors=True, lists=with_lists, f32=True)
        ctx.cloud_dev(stack.ptr, 1, N * px, px, N, H, W, (pw, ph), white.ptr, p32)
        c5, p5, pts5, col5 = p32.download()
        assert pts5.dtype == np.float32 and col5.dtype == np.float32 and pts5.shape == pts.shape and col5.shape == col.shape
        assert np.array_equal(pts5, pts.astype(np.float32), equal_nan=True) and np.array_equal(col5, col.astype(np.float32))
        assert (c5 is None and p5 is None) if not with_lists else (np.array_equal(c5, rcam) and np.array_equal(p5, rproj))
        p32.free()
    print(f"\n{workload}: {len(rcam)} points, worst rel. XYZ error vs the oracle {float(err.max()):.2e}, {path}")
    for b in (stack, maps, maps2, xyz, white):
        b.free()
    lists.free()
    bare.free()


LINE_CASES = [
    # (W, H, how the validity mask is made)
    (256, 200, "density 0.0"), (256, 200, "density 0.01"), (256, 200, "density 0.1"), (256, 200, "density 0.5"), (256, 200, "density 0.97"),
    (256, 200, "density 1.0"), (128, 3000, "density 0.8"), (332, 77, "density 0.8"), (64, 33, "density 0.9"), (4, 1, "density 1.0"),
    (256, 200, "rows 31 32"), (256, 200, "every 40th row"), (256, 200, "one column"), (256, 200, "checkerboard"), (256, 200, "bands"),
    (1920, 1080, "density 0.85"), (4096, 3000, "density 0.3"), (4096, 3000, "bands"),       # thousands of tiles; the sizes where the library picks whole lines itself
]


@pytest.mark.parametrize("W,H,how", LINE_CASES)
def test_list_build_in_whole_lines_equals_tile_runs(ctx, W, H, how):
    """The x-major list build of slgc_cloud_dev exists twice: k_xmajor_scatter (a tile writes its own run of every column) and k_xmajor_lines
    (a tile writes whole 16-record groups, completing the group that straddles the seam from the rows below; sparse columns fall back to
    partial writes by a per-column rule).  Same four arrays bit for bit on hand-made maps of every density -- empty, a few pixels per
    column (every record an orphan of a group begun tiles earlier), half, nearly all, all -- and on structured masks that put a column's
    records exactly at tile seams; the lists against the oracle too, and nothing written past the M-th record."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((W, H, how)).encode()))
    pw, ph = 300, 200
    K = np.array([[300.0, 0, W / 2], [0, 300.0, H / 2], [0, 0, 1]])
    _, cd, pk, pd, R, T = bench.calibration(1920, 1080, pw, ph)
    ctx.set_calibration(K, cd, pk, pd, R, T)
    ctx.tune("cam_nodes", 2 if W % 4 == 0 else 1)                              # node table wherever it is accurate: both kernel variants get exercised
    yy, xx = np.mgrid[0:H, 0:W]
    if how.startswith("density"):
        mask = rng.random((H, W)) < float(how.split()[1])
    elif how == "rows 31 32":
        mask = (yy == 31) | (yy == 32)
    elif how == "every 40th row":
        mask = yy % 40 == 7
    elif how == "one column":
        mask = xx == 67
    elif how == "checkerboard":
        mask = (xx + yy) % 2 == 0
    else:                                                                       # dense bands of 48 rows between empty bands of 30
        mask = (yy % 78) < 48
    h = np.where(mask, rng.integers(0, pw + 20, (H, W)), -1).astype(np.int16)   # a few codes past the projector: clamped (triangulate.py:60-61)
    v = np.where(mask, rng.integers(0, ph + 20, (H, W)), -1).astype(np.int16)
    half = rng.random((H, W)) < 0.02                                            # and pixels where only one map decoded
    h[half & (xx % 2 == 0)] = -1
    v[half & (xx % 2 == 1)] = -1
    px = W * H
    white_h = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    dh, dv, white = ctx.alloc(max(16, px * 2)).upload(h), ctx.alloc(max(16, px * 2)).upload(v), ctx.alloc(max(16, px * 3)).upload(white_h)
    ctx.build_ray_tables_dev(H, W, 0, (pw, ph))
    got = {}
    for lines in (0, 1):
        ctx.tune("lists_lines", 2 * lines)                                      # 2 = wherever the shape allows (1 = only images of >= 2048 tiles)
        L = ctx.alloc_cloud_lists(px, colors=True)
        for buf, nbytes in ((L.cam, px * 8), (L.proj, px * 8), (L.pts, px * 24), (L.colors, px * 24)):
            ctx.dev_memset(buf.ptr, 0xA5, max(16, nbytes))
        ctx.cloud_lists_dev(dh.ptr, dv.ptr, None, white.ptr, W, H, (pw, ph), L)
        assert ctx.last_list_kernel() == ("whole-lines" if lines and W % 4 == 0 and px >= 4 else "tile-runs")
        M = L.total()
        raw = {"cam": L.cam.download((px, 2), np.float32), "proj": L.proj.download((px, 2), np.float32), "col": L.colors.download((px, 3), np.float64)}
        for name, a in raw.items():
            assert (a[M:].view(np.uint8) == 0xA5).all(), (lines, name)
        planes = L.pts.download((3 * px,), np.float64)
        assert (planes[3 * M:].view(np.uint8) == 0xA5).all(), lines
        got[lines] = L.download()
        L.free()
    ctx.tune("lists_lines", 1)
    rcam, rproj, rcol = oc.cam_proj_pts(h.astype(np.int64), v.astype(np.int64), (W, H), (pw, ph), white_h, order="x")
    assert len(rcam) == int(((h != -1) & (v != -1)).sum())
    for a, b, name in zip(got[1], got[0], ("cam", "proj", "pts", "colours")):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    assert np.array_equal(got[1][0], rcam) and np.array_equal(got[1][1], rproj) and np.array_equal(got[1][3], rcol)
    if len(rcam):
        assert np.isfinite(got[1][2]).any() or how == "density 0.0"
    for b in (dh, dv, white):
        b.free()


def test_movement_yardstick_runs_and_refuses_bad_shapes(ctx):
    """slgc_move_only_dev (the kernel bench.py times beside the graded one: it only moves a scan's bytes) -- it runs on the three frame counts,
    touches exactly the output buffers it is given, and refuses the shapes it is not built for."""
    W, H, N = 1920, 1080, 44
    px = W * H
    stack, maps, xyz = ctx.alloc(46 * px), ctx.alloc(px * 4 + 64), ctx.alloc(px * 12 + 64)
    ctx.dev_memset(stack.ptr, 7, 46 * px)
    for n in (42, 44, 46):
        ctx.dev_memset(maps.ptr, 0xA5, px * 4 + 64)
        ctx.dev_memset(xyz.ptr, 0xA5, px * 12 + 64)
        ctx.move_only_dev(stack.ptr, px, n, px, d_xyz=xyz.ptr)
        ctx.synchronize()
        assert (maps.download((px * 4 + 64,), np.uint8) == 0xA5).all()                       # no map buffers given: none written
        got = xyz.download((px * 12 + 64,), np.uint8)
        assert (got[:px * 12] != 0xA5).any() and (got[px * 12:] == 0xA5).all()
        ctx.move_only_dev(stack.ptr, px, n, px, d_h=maps.at(0), d_v=maps.at(px * 2), d_xyz=xyz.ptr)
        ctx.synchronize()
        got = maps.download((px * 4 + 64,), np.uint8)
        assert (got[:px * 4] != 0xA5).any() and (got[px * 4:] == 0xA5).all()
    for bad in (dict(N=43), dict(npix=px - 4), dict(d_h=maps.at(0))):
        kw = dict(N=N, npix=px, d_h=None, d_v=None, d_xyz=xyz.ptr)
        kw.update(bad)
        with pytest.raises(Exception):
            ctx.move_only_dev(stack.ptr, px, kw["N"], kw["npix"], d_h=kw["d_h"], d_v=kw["d_v"], d_xyz=kw["d_xyz"])
    for b in (stack, maps, xyz):
        b.free()


@pytest.mark.parametrize("workload", ["c2_1920x1080x44", "b8_4096x375x44", "c3_4096x3000x44"])
def test_issue_priorities_change_no_bit(ctx, workload):
    """slgc_tune "prio" (s_setprio per phase of the fused kernel; -1 = the library's choice by launch shape) is a scheduling knob: maps and XYZ
    are bit-identical under every setting, and against the oracle under the default."""
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS[workload]
    calib = bench.calibration(W, H, pw, ph)
    ctx.set_calibration(*calib)
    px = W * H
    stack = ctx.alloc(N * px)
    ctx.synth_scene_dev(stack.ptr, px, N, H, W, row0=0, rows=H, seed=5, noise=3, shadow=True)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    got = {}
    try:
        for prio in (-1, 0, 210, 321, 13):
            ctx.tune("prio", prio)
            maps.zero()
            xyz.zero()
            ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=_native.TRI_ALGEBRAIC)
            ctx.synchronize()
            assert ctx.last_scan_path()["path"] == "fused"
            got[prio] = (maps.download((2, H, W), np.int16), xyz.download((H, W, 3), np.float32))
    finally:
        ctx.tune("prio", -1)
    for prio, (m, x) in got.items():
        assert np.array_equal(m, got[-1][0]) and np.array_equal(x.view(np.uint32), got[-1][1].view(np.uint32)), prio
    st = stack.download((N, H, W), np.uint8)
    ref_h, ref_v, ref_xyz = oc.scan_dense(st, (pw, ph), *calib)
    compare_scan(got[-1][0][0], got[-1][0][1], got[-1][1], ref_h, ref_v, ref_xyz, f"{workload} default priorities")
    for b in (stack, maps, xyz):
        b.free()


@pytest.mark.parametrize("N,proj", [(50, (3840, 2160)), (54, (7680, 4320))])
def test_4k_and_8k_projector_frame_counts_take_the_specialised_kernels(ctx, N, proj):
    """The reference's generator emits 4 * ceil(log2(max(w, h))) + 2 frames (generate_codes.py:22-25,53): 50 for a 4K projector (L = 12 code
    bits), 54 for 8K (L = 13).  Both have their own compiled specialisation of the decode / fused / BGR kernels (threshold frames parked in
    LDS): every pixel of a 1920x1080 capture against the oracle -- fused, split, two runs -- and the executed path says which kernel ran."""
    from scanner import _native
    W, H = 1920, 1080
    pw, ph = proj
    calib = bench.calibration(W, H, pw, ph)
    ctx.set_calibration(*calib)
    px = W * H
    stack = ctx.alloc(2 * N * px)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    L = (N - 2) // 4
    for scene in ("s-scene", "s-uniform"):
        if scene == "s-scene":                                                           # dense, smooth codes (the low bits) ...
            ctx.synth_scene_dev(stack.at(0), px, N, H, W, seed=1, noise=3, shadow=True)
            ctx.synth_scene_dev(stack.at(N * px), px, N, H, W, seed=2, noise=9, shadow=False)
        else:                                                                            # ... and every byte random: arbitrary codes, every bit of the L in use
            ctx.synth_uniform_dev(stack.at(0), px, N, H, W, seed=3)
            ctx.synth_uniform_dev(stack.at(N * px), px, N, H, W, seed=4)
        ctx.synchronize()
        st = stack.download((2, N, H, W), np.uint8)
        for n_runs in (1, 2):
            ref_h, ref_v, ref_xyz = oc.scan_dense(st[0] if n_runs == 1 else st, (pw, ph), *calib)
            if scene == "s-uniform":
                assert int(ref_h.max()) >= (1 << (L - 1)) and int(ref_v.max()) >= (1 << (L - 1))       # the top code bit is in use
            for mode in (_native.TRI_ALGEBRAIC, _native.TRI_ALGEBRAIC | _native.TRI_SPLIT):
                maps.zero()
                xyz.zero()
                ctx.scan_dev(stack.ptr, n_runs, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
                ctx.synchronize()
                path = ctx.last_scan_path()
                assert path["ns_frames"] == N and path["path"] == ("fused" if mode == _native.TRI_ALGEBRAIC else "split"), path
                valid, worst = compare_scan(maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), xyz.download((H, W, 3), np.float32),
                                            ref_h, ref_v, ref_xyz, f"N={N} {scene} runs={n_runs} mode={mode}")
                assert valid > (0.5 if scene == "s-scene" else 0.05) * px
    for b in (stack, maps, xyz):
        b.free()
