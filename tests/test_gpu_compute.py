"""Triangulation.compute() as ONE device-resident call (slgc_compute_count / _fetch) against the three-call composition it replaces
(get_cam_proj_pts -> triangulate -> filter_3d_pts, /root/reference/scanner/triangulation/triangulate.py:39-122 as src/4-triangulate.py:62-71
calls them): bit-identical on the reference-generated goldens, on ragged / empty / out-of-int16 maps, and at 4096x3000 (BASELINE configs[2])."""
import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

if conftest.has_gpu():
    from scanner import _native
    from scanner.triangulation.triangulate import Triangulation


@pytest.fixture(scope="module")
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


def _tri(c, ctx, h=None, v=None):
    return Triangulation(c["h"] if h is None else h, c["v"] if v is None else v, tuple(int(x) for x in c["cam_size"]), c["cam_mtx"],
                         c["cam_dist"], tuple(int(x) for x in c["proj_size"]), tuple(int(x) for x in c["proj_calib_size"]),
                         c["proj_mtx"].copy(), c["proj_dist"], c["R"], c["T"], None, ctx=ctx)


def _same(a, b):
    for x, y in zip(a, b):
        if x is None or y is None:
            assert x is None and y is None
        else:
            assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y, equal_nan=True)


def test_compute_goldens_bit_identical_to_three_calls(ctx, tri_cases):
    for name, c in tri_cases.items():
        t = _tri(c, ctx)
        for white in (c["white"], None):
            for thr in (None, 0.5, 0.05, 1e9):
                for exact in (True, False):
                    for order in ("x", "row"):
                        _same(t.compute(white, threshold=thr, exact=exact, order=order),
                              t.compute_by_calls(white, threshold=thr, exact=exact, order=order))
        # against the reference's own outputs: unfiltered points within the tolerance BASELINE.json states (1e-4 relative; the exact kernel
        # achieves 1e-9), colours bit for bit; the filtered selection equals the reference's wherever no coordinate sits within 1e-9 of the box
        pts, col = t.compute(c["white"])
        np.testing.assert_allclose(pts, c["pts"], rtol=1e-9, atol=1e-12)
        assert np.array_equal(col, c["colors"])
        fp, fc = t.compute(c["white"], threshold=0.5)
        if not np.any(np.abs(np.abs(c["pts"]) - 0.5) < 1e-9):
            assert fp.shape == c["filt_pts"].shape and np.array_equal(fc, c["filt_colors"])
            np.testing.assert_allclose(fp, c["filt_pts"], rtol=1e-9, atol=1e-12)
        assert ctx.last_input_path() == 1                      # the maps crossed the link as int16


def test_compute_maps_that_do_not_fit_int16_and_odd_values(ctx, tri_cases):
    c = next(iter(tri_cases.values()))
    h, v = c["h"].astype(np.int64).copy(), c["v"].astype(np.int64).copy()
    # values other than -1 below zero are ordinary values to the reference (only == -1 is skipped, triangulate.py:56): int16 holds them
    h[0, :5] = [-2, -32768, 32767, 0, -1]
    t = _tri(c, ctx, h, v)
    _same(t.compute(c["white"], threshold=0.5), t.compute_by_calls(c["white"], threshold=0.5))
    assert ctx.last_input_path() == 1
    # one value past int16: the maps are shipped as int64, same result (clamped to the projector like any large value, :60-61)
    h[1, 1] = 40000
    v[2, 2] = (1 << 40) + 3
    t = _tri(c, ctx, h, v)
    _same(t.compute(c["white"], threshold=None), t.compute_by_calls(c["white"], threshold=None))
    assert ctx.last_input_path() == 2


@pytest.mark.parametrize("W,H", [(1, 1), (3, 2), (65, 33), (517, 301)])
def test_compute_ragged_and_empty(ctx, tri_cases, W, H):
    c = next(iter(tri_cases.values()))
    rng = np.random.default_rng(W * 77 + H)
    h = rng.integers(-1, 1500, (H, W)).astype(np.int64)
    v = rng.integers(-1, 900, (H, W)).astype(np.int64)
    h[rng.random((H, W)) < 0.3] = -1
    white = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    cc = dict(c)
    cc["cam_size"] = np.array([W, H])
    for hh in (h, np.full_like(h, -1)):                      # the second: no decodable pixel at all
        t = _tri(cc, ctx, hh, v)
        for thr in (None, 0.5):
            _same(t.compute(white, threshold=thr), t.compute_by_calls(white, threshold=thr))
            _same(t.compute(None, threshold=thr), t.compute_by_calls(None, threshold=thr))


def test_compute_fuzz_vs_three_calls(ctx, tri_cases):
    """Seeded fuzz (SLGC_FUZZ_SCALE multiplies the case count): random sizes, maps with -1 / other negatives / values past the projector and past int16,
    random thresholds and orders -- compute() must stay bit-identical to the three-call composition."""
    import os
    c = next(iter(tri_cases.values()))
    rng = np.random.default_rng(2026)
    for k in range(60 * int(os.environ.get("SLGC_FUZZ_SCALE", "1"))):
        W, H = int(rng.integers(1, 160)), int(rng.integers(1, 120))
        hi = int(rng.choice([40, 1500, 33000, 70000]))
        h = rng.integers(-3, hi, (H, W)).astype(np.int64)
        v = rng.integers(-3, hi, (H, W)).astype(np.int64)
        h[rng.random((H, W)) < rng.random()] = -1
        white = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if rng.random() < 0.7 else None
        cc = dict(c)
        cc["cam_size"] = np.array([W, H])
        t = _tri(cc, ctx, h, v)
        thr = None if rng.random() < 0.3 else float(rng.choice([0.05, 0.5, 3.0, 1e6]))
        order = "x" if rng.random() < 0.7 else "row"
        exact = bool(rng.random() < 0.5)
        _same(t.compute(white, threshold=thr, exact=exact, order=order), t.compute_by_calls(white, threshold=thr, exact=exact, order=order))
        assert ctx.last_input_path() == (2 if (h.max() > 32767 or v.max() > 32767) else 1), (k, W, H, hi)


def test_compute_needs_calibration():
    c2 = _native.Context(0)
    try:
        with pytest.raises(_native.SlgcError):
            c2.compute(np.zeros((4, 4), np.int64), np.zeros((4, 4), np.int64), (4, 4), (8, 8))
    finally:
        c2.close()


def test_compute_full_size_c3(ctx):
    """BASELINE configs[2] (4096x3000 camera, 1920x1200 projector, 44 frames): decode the S-scene capture, then compute() against the three
    calls on all 12.3 M pixels -- bit for bit, and the fused call must be the cheaper one."""
    import time
    import bench
    W, H, pw, ph, N = bench.WORKLOADS["c3_4096x3000x44"]
    px = W * H
    d = ctx.alloc(N * px)
    ctx.synth_scene_dev(d.ptr, px, N, H, W, seed=1, noise=3, shadow=True)
    maps = ctx.alloc(px * 4)
    ctx.decode_dev(d.ptr, 1, N * px, px, N, H, W, maps.at(0), maps.at(px * 2))
    ctx.synchronize()
    h = maps.download((H, W), np.int16).astype(np.int64)
    v = maps.download((H, W), np.int16, byte_offset=px * 2).astype(np.int64)
    white1 = d.download((H, W), np.uint8, byte_offset=px)          # frame 1 = the white frame
    white = np.repeat(white1[:, :, None], 3, axis=2)
    d.free()
    maps.free()
    K, cd, pk, pd, R, T = bench.calibration(W, H, pw, ph)
    t = Triangulation(h, v, (W, H), K, cd, (pw, ph), (pw, ph), pk.copy(), pd, R, T, None, ctx=ctx)
    best = {"fused": 1e9, "calls": 1e9}
    for _ in range(3):
        t0 = time.perf_counter()
        a = t.compute(white, threshold=0.5)
        t1 = time.perf_counter()
        b = t.compute_by_calls(white, threshold=0.5)
        t2 = time.perf_counter()
        best["fused"] = min(best["fused"], t1 - t0)
        best["calls"] = min(best["calls"], t2 - t1)
        _same(a, b)
        del b
    assert a[0].shape[1] > 1_000_000
    print(f"compute(): fused {best['fused'] * 1e3:.1f} ms, three calls {best['calls'] * 1e3:.1f} ms")
    assert best["fused"] < 0.6 * best["calls"]
