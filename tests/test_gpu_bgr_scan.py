"""slgc_scan_bgr_dev (-m gpu): the scan straight from the camera's BGR frames.

/root/reference/src/3-capture_decode.py:66-70 converts every camera frame with cv2.cvtColor(BGR2GRAY) and stores it into the grey stack that
:75 hands to get_codes.  The fused kernel can form that luma inside its frame loads (3 N + 12 bytes per pixel instead of 3 N + N + N + 12).
Checked here: maps and XYZ bit-identical with the chain slgc_to_gray_dev -> slgc_scan_dev on the same BGR capture (both coefficient sets,
one and two runs, every shape that takes the one-kernel path and shapes that must fall back), and against the CPU oracle chain
(oracle_np.bgr_to_gray -> oracle_c.scan_dense) every pixel; at BASELINE configs[2] and [1] sizes too."""
import os

import numpy as np
import pytest

import bench
import oracle_c as oc
import oracle_np as onp
from conftest import ext, has_gpu
from test_gpu_fullsize import compare_scan

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]


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


def make_capture(ctx, scene, N, H, W, psize, n_runs=1, pad=0, row0=0, rows=None, Himg=None):
    """-> (bgr device buffer [n_runs][N][rows][W][3] with `pad` bytes between planes, grey device buffer the generator made)"""
    rows = H if rows is None else rows
    px = rows * W
    gray = ctx.alloc(max(16, n_runs * N * px))
    plane = 3 * px + pad
    bgr = ctx.alloc(max(16, n_runs * N * plane)).zero()
    for r in range(n_runs):
        bench.synth_into(ctx, scene, gray.at(r * N * px) if r else gray.ptr, px, N, Himg or H, W, psize, 5 + 3 * r, row0=row0, rows=rows)
        ctx.synth_bgr_dev(gray.at(r * N * px) if r else gray.ptr, px, N, Himg or H, W, bgr.at(r * N * plane) if r else bgr.ptr, plane, row0=row0, rows=rows)
    ctx.synchronize()
    return bgr, gray, plane


def test_bgr_capture_generator_equals_numpy_twin(ctx):
    N, H, W = 14, 37, 64
    ctx.set_calibration(*bench.calibration(W, H, 64, 48))
    bgr, gray, plane = make_capture(ctx, "s-scene", N, H, W, (64, 48), pad=8, row0=5, rows=20, Himg=H)
    g = gray.download((N, 20, W), np.uint8)
    got = bgr.download((N, plane), np.uint8)
    ref = onp.gray_to_bgr_capture(g, row0=5)
    assert np.array_equal(got[:, :3 * 20 * W].reshape(N, 20, W, 3), ref) and not got[:, 3 * 20 * W:].any()
    assert (ref[..., 0] != ref[..., 1]).mean() > 0.5 and (ref[..., 2] != ref[..., 1]).mean() > 0.5
    bgr.free()
    gray.free()


CASES = [
    # W, H, N, runs, pad, coeff_bits, scene, expect one kernel
    (256, 64, 44, 1, 0, 15, "s-scene", True),
    (256, 64, 44, 2, 0, 15, "s-scene", True),
    (260, 33, 42, 1, 12, 15, "s-uniform", True),
    (512, 96, 46, 1, 0, 14, "s-scene", True),
    (512, 96, 46, 2, 4, 14, "s-uniform", True),
    (512, 96, 50, 1, 0, 15, "s-uniform", True),     # a 4K projector's frame count (L = 12)
    (256, 64, 54, 2, 0, 15, "s-uniform", True),     # 8K (L = 13)
    (130, 31, 44, 1, 0, 15, "s-scene", False),      # 130 * 31 is not a multiple of 4 pixels
    (256, 64, 26, 1, 0, 15, "s-scene", False),      # no specialised kernel for 26 frames
    (256, 64, 44, 1, 2, 15, "s-scene", False),      # planes not 4-byte aligned
]


@pytest.mark.parametrize("W,H,N,runs,pad,bits,scene,fused", CASES)
def test_bgr_scan_equals_to_gray_then_scan_and_the_oracle(ctx, W, H, N, runs, pad, bits, scene, fused):
    from scanner import _native
    psize = (300, 200)
    calib = bench.calibration(W, H, *psize)
    calib = (calib[0].copy(),) + tuple(calib[1:])
    calib[0][0, 2], calib[0][1, 2], calib[0][0, 0], calib[0][1, 1] = W / 2, H / 2, 1.2 * W, 1.2 * W
    ctx.set_calibration(*calib)
    if scene == "s-uniform" and W % 4:
        pytest.skip("the uniform generator writes dwords")
    px = W * H
    bgr, gray, plane = make_capture(ctx, scene, N, H, W, psize, n_runs=runs, pad=pad)
    maps, xyz = ctx.alloc(px * 4 + 64).zero(), ctx.alloc(px * 12).zero()
    ctx.scan_bgr_dev(bgr.ptr, runs, N * plane, plane, N, H, W, 0, psize, xyz.ptr, None, maps.at(0), maps.at(px * 2), coeff_bits=bits)
    ctx.synchronize()
    path = ctx.last_scan_path()["path"]
    assert (path == "fused-bgr") == fused, path
    got = (maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), xyz.download((H, W, 3), np.float32))
    # the two-step chain on the device: grey stack through slgc_to_gray_dev, then the ordinary scan
    g2 = ctx.alloc(runs * N * px)
    lib = _native.lib()
    for r in range(runs):
        for f in range(N):
            ctx._ck(lib.slgc_to_gray_dev(ctx._h, bgr.at((r * N + f) * plane) if (r or f) else bgr.ptr, px, bits, g2.at((r * N + f) * px) if (r or f) else g2.ptr))
    m2, x2 = ctx.alloc(px * 4 + 64).zero(), ctx.alloc(px * 12).zero()
    ctx.scan_dev(g2.ptr, runs, N * px, px, N, H, W, 0, psize, x2.ptr, None, m2.at(0), m2.at(px * 2))
    ctx.synchronize()
    chain = (m2.download((H, W), np.int16), m2.download((H, W), np.int16, px * 2), x2.download((H, W, 3), np.float32))
    assert np.array_equal(got[0], chain[0]) and np.array_equal(got[1], chain[1])
    assert np.array_equal(got[2].view(np.uint32), chain[2].view(np.uint32)), "XYZ must be bit-identical with to_gray + scan"
    # the oracle chain on the CPU
    raw = bgr.download((runs, N, plane), np.uint8)[:, :, :3 * px].reshape(runs, N, H, W, 3)
    gref = onp.bgr_to_gray(raw, coeff_bits=bits)
    assert np.array_equal(g2.download((runs, N, H, W), np.uint8), gref)
    ref_h, ref_v, ref_xyz = oc.scan_dense(gref if runs > 1 else gref[0], psize, *calib)
    valid, worst = compare_scan(*got, ref_h, ref_v, ref_xyz, f"bgr scan {W}x{H}x{N} runs={runs}")
    assert valid > 0
    # the maps alone (slgc_decode_bgr_dev): the decode kernel reading the BGR frames, or the same two-step chain
    m3 = ctx.alloc(px * 4 + 64).zero()
    ctx.decode_bgr_dev(bgr.ptr, runs, N * plane, plane, N, H, W, m3.at(0), m3.at(px * 2), coeff_bits=bits)
    ctx.synchronize()
    assert np.array_equal(m3.download((H, W), np.int16), ref_h) and np.array_equal(m3.download((H, W), np.int16, px * 2), ref_v)
    if fused:
        assert ctx.last_scan_path()["ns_frames"] == N                      # the specialised decode kernel, reading BGR
    for b in (bgr, gray, maps, xyz, g2, m2, x2, m3):
        b.free()


@pytest.mark.parametrize("workload", ["c3_4096x3000x44", "c2_1920x1080x44", ext("c3_4096x3000x46"), "c2_1920x1080x46"])
def test_bgr_scan_at_bench_sizes_every_pixel(ctx, workload):
    W, H, pw, ph, N = bench.WORKLOADS[workload]
    scene = "physical" if N != 46 else "s-scene"          # 46 frames: the S-scene's 11-bit codes run past the projector's edge (clamp)
    calib = bench.calibration(W, H, pw, ph, rig=bench.SCENES[scene]["rig"])
    ctx.set_calibration(*calib)
    px = W * H
    bgr, gray, plane = make_capture(ctx, scene, N, H, W, (pw, ph))
    gray.free()
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    ctx.scan_bgr_dev(bgr.ptr, 1, N * plane, plane, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2))
    ctx.synchronize()
    assert ctx.last_scan_path()["path"] == "fused-bgr"
    raw = bgr.download((N, H, W, 3), np.uint8)
    gref = onp.bgr_to_gray(raw)
    del raw
    ref_h, ref_v, ref_xyz = oc.scan_dense(gref, (pw, ph), *calib)
    valid, worst = compare_scan(maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), xyz.download((H, W, 3), np.float32), ref_h, ref_v, ref_xyz,
                                f"{workload} bgr scan")
    assert valid > (0.8 if scene == "physical" else 0.5) * px
    print(f"\n{workload} from BGR frames: {valid} / {px} decodable, worst rel. XYZ error {worst:.2e}")
    for b in (bgr, maps, xyz):
        b.free()


def test_new_entry_points_refuse_bad_arguments(ctx):
    """Error behaviour of the round-4 entry points: int status + message across the C-ABI (the binding raises ValueError for SLGC_EINVAL, SlgcError
    otherwise), the context stays usable."""
    from scanner import _native
    W, H, N = 64, 16, 44
    ctx.set_calibration(*bench.calibration(W, H, 64, 48))
    px = W * H
    bgr, xyz, maps = ctx.alloc(3 * N * px), ctx.alloc(px * 12), ctx.alloc(px * 4)
    with pytest.raises((ValueError, _native.SlgcError), match="coeff_bits"):
        ctx.scan_bgr_dev(bgr.ptr, 1, 3 * N * px, 3 * px, N, H, W, 0, (64, 48), xyz.ptr, coeff_bits=13)
    with pytest.raises((ValueError, _native.SlgcError), match="plane_stride"):
        ctx.scan_bgr_dev(bgr.ptr, 1, 3 * N * px, px, N, H, W, 0, (64, 48), xyz.ptr)                     # a grey plane stride for BGR planes
    with pytest.raises((ValueError, _native.SlgcError), match="null"):
        ctx.scan_bgr_dev(None, 1, 3 * N * px, 3 * px, N, H, W, 0, (64, 48), xyz.ptr)
    with pytest.raises((ValueError, _native.SlgcError), match="unsupported N"):
        ctx.scan_bgr_dev(bgr.ptr, 1, 3 * 13 * px, 3 * px, 13, H, W, 0, (64, 48), xyz.ptr)
    with pytest.raises((ValueError, _native.SlgcError), match="integer eps"):
        ctx.scan_bgr_dev(bgr.ptr, 1, 3 * N * px, 3 * px, N, H, W, 0, (64, 48), xyz.ptr, eps=0.5)
    with pytest.raises((ValueError, _native.SlgcError), match="bad synth"):
        ctx.synth_uniform_dev(bgr.ptr, px, N, H, 63, seed=1)                                             # W not a multiple of 4
    with pytest.raises((ValueError, _native.SlgcError), match="bad synth"):
        ctx.synth_physical_dev(bgr.ptr, px, N, H, W, (64, 48), gains=(10, 999), r2_max=1.0)
    with pytest.raises((ValueError, _native.SlgcError), match="bad synth"):
        ctx.synth_bgr_dev(bgr.ptr, px, N, H, W, xyz.ptr, px)                                              # BGR plane stride too small
    p32 = ctx.alloc_cloud_lists(px, colors=False, points=False, f32=True)
    with pytest.raises((ValueError, _native.SlgcError), match="d_pts32 is required"):
        ctx.cloud_dev(bgr.ptr, 1, N * px, px, N, H, W, (64, 48), None, p32)
    p32.free()
    lists32 = ctx.alloc_cloud_lists(px, f32=True)
    with pytest.raises(ValueError, match="float64"):
        ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), None, None, W, H, (64, 48), lists32)
    lists32.free()
    # an empty band is not an error
    ctx.scan_bgr_dev(bgr.ptr, 1, 0, 0, N, 0, W, 0, (64, 48), xyz.ptr)
    ctx.synchronize()
    # ... and the context still works
    bench.synth_into(ctx, "s-scene", maps.ptr if False else bgr.ptr, px, N, H, W, (64, 48), 1)
    ctx.scan_dev(bgr.ptr, 1, N * px, px, N, H, W, 0, (64, 48), xyz.ptr)
    ctx.synchronize()
    assert np.isfinite(xyz.download((H, W, 3), np.float32)).any()
    for b in (bgr, xyz, maps):
        b.free()


def test_bgr_scan_fuzz_vs_chain_and_oracle(ctx):
    """Seeded random cases (SLGC_FUZZ_SCALE multiplies them): frame counts with and without a specialised kernel, one or two runs, plane padding that keeps or
    breaks the 4-byte alignment, both coefficient sets, random calibrations, random bytes or the S-scene: whichever path slgc_scan_bgr_dev takes, maps and XYZ are
    bit-identical with slgc_to_gray_dev + slgc_scan_dev and match the oracle chain."""
    from scanner import _native
    rng = np.random.default_rng(909)
    lib = _native.lib()
    took = {"fused-bgr": 0, "other": 0}
    for case in range(30 * int(os.environ.get("SLGC_FUZZ_SCALE", "1"))):
        N = int(rng.choice([42, 44, 46, 44, 46, 26, 30]))
        W = int(rng.integers(2, 60)) * 4 if case % 5 else int(rng.integers(9, 200))
        H = int(rng.integers(1, 40))
        runs = int(rng.integers(1, 3))
        pad = int(rng.choice([0, 0, 4, 16, 6]))
        bits = int(rng.choice([15, 15, 14]))
        psize = (int(rng.integers(8, 2100)), int(rng.integers(8, 1300)))
        K = np.array([[float(rng.uniform(0.9, 2.2)) * max(W, H), 0, W / 2 + float(rng.uniform(-3, 3))], [0, float(rng.uniform(0.9, 2.2)) * max(W, H), H / 2 + float(rng.uniform(-3, 3))], [0, 0, 1]])
        _, cd, pk, pd, R, T = bench.calibration(1920, 1080, *psize)
        calib = (K, cd * float(rng.uniform(0, 1.3)), pk, pd * float(rng.uniform(0, 1.2)), R, T * float(rng.uniform(0.7, 1.4)))
        ctx.set_calibration(*calib)
        px = W * H
        if case % 3 == 0:
            gray_h = rng.integers(0, 256, (runs, N, H, W), dtype=np.uint8)
        else:
            gray_h = np.stack([onp.synth_scene_int(N, H, W, seed=int(rng.integers(1 << 30)), noise=int(rng.integers(0, 9)))[0] for _ in range(runs)])
        bgr_h = onp.gray_to_bgr_capture(gray_h.reshape(runs * N, H, W)).reshape(runs, N, px * 3)
        plane = 3 * px + pad
        buf = np.zeros((runs, N, plane), np.uint8)
        buf[:, :, :3 * px] = bgr_h
        d_bgr = ctx.alloc(max(16, buf.nbytes)).upload(buf)
        maps, xyz = ctx.alloc(px * 4 + 64).zero(), ctx.alloc(px * 12 + 16).zero()
        voff = (px * 2 + 31) // 32 * 32
        ctx.scan_bgr_dev(d_bgr.ptr, runs, N * plane, plane, N, H, W, 0, psize, xyz.ptr, None, maps.at(0), maps.at(voff), coeff_bits=bits)
        ctx.synchronize()
        path = ctx.last_scan_path()["path"]
        took["fused-bgr" if path == "fused-bgr" else "other"] += 1
        assert (path == "fused-bgr") == (N in (42, 44, 46) and px % 4 == 0 and pad % 4 == 0 and px >= 4), (case, path, N, W, H, pad)
        got = (maps.download((H, W), np.int16), maps.download((H, W), np.int16, voff), xyz.download((H, W, 3), np.float32))
        gref = onp.bgr_to_gray(buf[:, :, :3 * px].reshape(runs, N, H, W, 3), coeff_bits=bits)
        g2 = ctx.alloc(max(16, gref.nbytes)).upload(gref)
        m2, x2 = ctx.alloc(px * 4 + 64).zero(), ctx.alloc(px * 12 + 16).zero()
        ctx.scan_dev(g2.ptr, runs, N * px, px, N, H, W, 0, psize, x2.ptr, None, m2.at(0), m2.at(voff))
        ctx.synchronize()
        tag = f"case {case}: {W}x{H}x{N} runs={runs} pad={pad} bits={bits} path={path}"
        assert np.array_equal(got[0], m2.download((H, W), np.int16)) and np.array_equal(got[1], m2.download((H, W), np.int16, voff)), tag
        assert np.array_equal(got[2].view(np.uint32), x2.download((H, W, 3), np.float32).view(np.uint32)), tag
        ref_h, ref_v, ref_xyz = oc.scan_dense(gref if runs > 1 else gref[0], psize, *calib)
        compare_scan(*got, ref_h, ref_v, ref_xyz, tag)
        for b in (d_bgr, maps, xyz, g2, m2, x2):
            b.free()
    assert took["fused-bgr"] > 5 and took["other"] > 5, took
