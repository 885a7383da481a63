"""Physically consistent synthetic capture (-m gpu): generator vs its NumPy twin, and recovered XYZ vs the TRUE surface.

The S-scene of SURVEY.md 8(d) encodes arbitrary (h, v) maps: they violate epipolar geometry, so triangulation parity there is parity with
a formula, never with a shape.  ``slgc_synth_physical_dev`` casts every camera pixel's ray into a plane + sphere scene, carries the hit
point through the stereo pose and the projector's forward lens model, and encodes the projector pixel that really lights it -- what
/root/reference/src/4-triangulate.py:50-64 assumes of its inputs.  The accuracy assertion below does NOT go through the oracle: the
truth is the generator's own surface point, and the bound is the geometry of half a projector pixel (law of sines differentiated:
d len / len = sin(alpha) / (sin(beta) sin(gamma)) * d beta, /root/reference/scanner/triangulation/triangulate.py:91-94)."""
import numpy as np
import pytest

import bench
import oracle_np as onp
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]


@pytest.fixture(scope="module")
def ctx():
    from scanner import _native
    c = _native.Context(0)
    yield c
    c.close()


def matched_rig(W, H, pw, ph):
    """A rig whose camera and projector see the same part of the scene (most camera pixels are lit): mild lens distortion on both, the
    projector 0.2 m to the camera's left, toed in by 24 degrees."""
    K = np.array([[2.2 * W, 0, W / 2 + 3.0], [0, 2.2 * W, H / 2 - 2.0], [0, 0, 1]])
    cd = np.array([-0.08, 0.05, 0.0007, -0.0004, 0.01])
    pk = np.array([[1.9 * pw, 0, pw / 2 - 5.0], [0, 1.9 * pw, ph / 2 + 4.0], [0, 0, 1]])
    pd = np.array([0.04, -0.06, -0.0005, 0.0008, 0.02])
    th = np.deg2rad(-24.0)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T = np.array([[0.20], [0.005], [0.045]])
    return K, cd, pk, pd, R, T


def generate(ctx, N, H, W, psize, seed=1, noise=3, row0=0, rows=None):
    rows = H if rows is None else rows
    px = rows * W
    stack, hv, truth = ctx.alloc(max(16, N * px)), ctx.alloc(max(16, px * 4)), ctx.alloc(max(16, px * 12))
    ctx.synth_physical_dev(stack.ptr, px, N, H, W, psize, row0=row0, rows=rows, seed=seed, noise=noise, d_h_true=hv.at(0), d_v_true=hv.at(px * 2),
                           d_truth_xyz=truth.ptr)
    ctx.synchronize()
    return stack, hv, truth


@pytest.mark.parametrize("case", ["bench_c1_crop", "matched", "matched_band", "ragged"])
def test_generator_equals_numpy_twin(ctx, case):
    """Stack bytes, encoded codes and true points bit-identical with oracle_np.synth_physical (float64 + - * / sqrt in one order)."""
    if case == "bench_c1_crop":
        W, H, pw, ph, N = 1280, 720, 1280, 800, 42
        calib, row0, rows = bench.calibration(W, H, pw, ph), 420, 96
    elif case == "matched":
        W, H, pw, ph, N = 320, 200, 256, 192, 34
        calib, row0, rows = matched_rig(W, H, pw, ph), 0, H
    elif case == "matched_band":
        W, H, pw, ph, N = 320, 200, 256, 192, 46
        calib, row0, rows = matched_rig(W, H, pw, ph), 77, 41
    else:
        W, H, pw, ph, N = 203, 57, 200, 150, 38
        calib, row0, rows = matched_rig(W, H, pw, ph), 3, 50
    ctx.set_calibration(*calib)
    stack, hv, truth = generate(ctx, N, H, W, (pw, ph), seed=5, noise=4, row0=row0, rows=rows)
    px = rows * W
    st, h, v, tr = onp.synth_physical(N, H, W, (pw, ph), calib, seed=5, noise=4, row0=row0, rows=rows)
    assert np.array_equal(hv.download((rows, W), np.int16), h) and np.array_equal(hv.download((rows, W), np.int16, px * 2), v)
    assert np.array_equal(stack.download((N, rows, W), np.uint8), st)
    got = truth.download((rows, W, 3), np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(tr)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(tr.astype(np.float32)))
    lit = h != -1
    assert 0.05 < lit.mean() and (case == "bench_c1_crop" or lit.mean() > 0.5)
    for b in (stack, hv, truth):
        b.free()


def quantisation_bound(truth, T, pk, slack=1.3):
    """Per-pixel bound on |recovered - true| from half a projector pixel of code quantisation, pure geometry of the TRUE point P (frame:
    relative to the camera centre, projector axes; the projector centre sits at -T): the recovered point lies on the camera ray at the
    range the law of sines gives, d len = len * sin(alpha) / (sin(beta) sin(gamma)) * d beta, and half a pixel in both axes subtends at
    most 0.5 * sqrt(1/fx^2 + 1/fy^2) in normalised coordinates (slack: the lens model stretches a pixel by up to ~1.22 inside the lit region)."""
    t = np.asarray(T, dtype=np.float64).reshape(3)
    tl = np.linalg.norm(t)
    P = truth.astype(np.float64)
    ln = np.linalg.norm(P, axis=-1)
    Q = P + t                                                 # from the projector centre to the point
    qn = np.linalg.norm(Q, axis=-1)
    cos_a = -(P @ t) / (ln * tl)
    cos_b = (Q @ t) / (qn * tl)
    sin_a, sin_b = np.sqrt(1 - cos_a ** 2), np.sqrt(1 - cos_b ** 2)
    sin_g = sin_a * cos_b + cos_a * sin_b
    dbeta = slack * 0.5 * np.sqrt(1.0 / pk[0, 0] ** 2 + 1.0 / pk[1, 1] ** 2)
    return ln * sin_a / (sin_b * sin_g) * dbeta


@pytest.mark.parametrize("workload", ["c1_1280x720x42", "c2_1920x1080x44", "c3_4096x3000x44", "matched_1024x768x46"])
def test_recovered_surface_within_code_quantisation_of_the_truth(ctx, workload):
    from scanner import _native
    if workload.startswith("matched"):
        W, H, pw, ph, N = 1024, 768, 1024, 768, 46
        calib = matched_rig(W, H, pw, ph)
    else:
        W, H, pw, ph, N = bench.WORKLOADS[workload]
        calib = bench.calibration(W, H, pw, ph)
    ctx.set_calibration(*calib)
    px = W * H
    stack, hv, truth = generate(ctx, N, H, W, (pw, ph))
    maps, xyz, cnt = ctx.alloc(px * 4), ctx.alloc(px * 12), ctx.alloc(16).zero()
    ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=_native.TRI_ALGEBRAIC)
    assert ctx.last_scan_path()["path"] == "fused"
    ctx.guard_count_dev(maps.at(0), maps.at(px * 2), H, W, 0, (pw, ph), cnt.ptr)
    ctx.synchronize()
    n_ok, n_flat = (int(x) for x in cnt.download((2,), np.uint64))
    h_true, v_true = hv.download((H, W), np.int16), hv.download((H, W), np.int16, px * 2)
    h, v = maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2)
    got, tru = xyz.download((H, W, 3), np.float32), truth.download((H, W, 3), np.float32)
    lit = h_true != -1
    ok = (h != -1) & (v != -1)
    assert not (ok & ~lit).any()                                          # nothing decodes where the projector does not reach
    assert ok.sum() >= 0.93 * lit.sum() and n_ok == int(ok.sum())         # (N = 44: the reference's fractional pattern_len drops some lit pixels)
    assert np.array_equal(h[ok], h_true[ok]) and np.array_equal(v[ok], v_true[ok])       # the decoder returns the code that was projected
    err = np.linalg.norm(got[ok].astype(np.float64) - tru[ok], axis=1)
    bound = quantisation_bound(tru[ok], calib[5], calib[2])
    ratio = err / bound
    rng = np.linalg.norm(tru[ok].astype(np.float64), axis=1)
    print(f"\n{workload}: lit {lit.mean():.3f} of the image, decoded {ok.sum()} of {lit.sum()} lit pixels, on the guarded path {n_flat} "
          f"({100.0 * n_flat / max(1, n_ok):.4f} %); |recovered - true| median {np.median(err) * 1e3:.3f} mm, max {err.max() * 1e3:.3f} mm at ranges "
          f"{rng.min():.3f}..{rng.max():.3f} m; worst error / quantisation bound {ratio.max():.3f}")
    assert ratio.max() <= 1.0, f"recovered surface off by {ratio.max():.2f}x the code-quantisation bound"
    assert np.median(err) < 0.4e-3 and n_flat <= 0.001 * n_ok
    for b in (stack, hv, truth, maps, xyz, cnt):
        b.free()


def test_random_rigs_recover_the_true_surface(ctx):
    """Property over 12 random stereo rigs (focal lengths, principal points, lens coefficients, toe-in, baseline, image and projector sizes,
    frame counts, ragged shapes included): whatever path slgc_scan_dev takes, the maps are the projected codes and the surface lies within the
    quantisation bound of the truth."""
    from scanner import _native
    rng = np.random.default_rng(2026)
    worst = 0.0
    for trial in range(12):
        W, H = int(rng.integers(60, 180)) * 4 + (0 if trial % 3 else int(rng.integers(0, 4))), int(rng.integers(90, 400))
        pw, ph = int(rng.integers(200, 600)), int(rng.integers(150, 450))
        L = int(np.ceil(np.log2(max(pw, ph))))
        N = 4 * L + 2 + 2 * int(rng.integers(0, 2))                                   # enough bits for the whole raster, sometimes two spare frames
        K = np.array([[rng.uniform(1.9, 2.6) * W, 0, W / 2 + rng.uniform(-6, 6)], [0, rng.uniform(1.9, 2.6) * W, H / 2 + rng.uniform(-6, 6)], [0, 0, 1]])
        pk = np.array([[rng.uniform(1.6, 2.2) * pw, 0, pw / 2 + rng.uniform(-8, 8)], [0, rng.uniform(1.6, 2.2) * pw, ph / 2 + rng.uniform(-8, 8)], [0, 0, 1]])
        sc = rng.uniform(0, 1.5)
        cd = sc * np.array([-0.08, 0.05, 0.0007, -0.0004, 0.01])
        pd = rng.uniform(0, 1.5) * np.array([0.04, -0.06, -0.0005, 0.0008, 0.02])
        th = np.deg2rad(rng.uniform(-28, -18))
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        T = np.array([[rng.uniform(0.16, 0.24)], [rng.uniform(-0.01, 0.01)], [rng.uniform(0.02, 0.06)]])
        calib = (K, cd, pk, pd, R, T)
        ctx.set_calibration(*calib)
        px = W * H
        stack, hv, truth = generate(ctx, N, H, W, (pw, ph), seed=100 + trial, noise=int(rng.integers(0, 5)))
        maps, xyz = ctx.alloc(px * 4 + 64), ctx.alloc(px * 12)
        voff = (px * 2 + 31) // 32 * 32
        ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(voff), mode=_native.TRI_ALGEBRAIC)
        ctx.synchronize()
        h_true, v_true = hv.download((H, W), np.int16), hv.download((H, W), np.int16, px * 2)
        h, v = maps.download((H, W), np.int16), maps.download((H, W), np.int16, voff)
        got, tru = xyz.download((H, W, 3), np.float32), truth.download((H, W, 3), np.float32)
        lit, ok = h_true != -1, (h != -1) & (v != -1)
        assert lit.mean() > 0.1, (trial, lit.mean())
        assert not (ok & ~lit).any() and ok.sum() >= 0.85 * lit.sum(), (trial, ok.sum(), lit.sum())       # (N = 4 L + 4: the reference's fractional pattern_len drops some lit pixels)
        assert np.array_equal(h[ok], h_true[ok]) and np.array_equal(v[ok], v_true[ok]), trial
        err = np.linalg.norm(got[ok].astype(np.float64) - tru[ok], axis=1)
        ratio = float((err / quantisation_bound(tru[ok], T, pk)).max())
        assert ratio <= 1.0, (trial, ratio, ctx.last_scan_path())
        worst = max(worst, ratio)
        for b in (stack, hv, truth, maps, xyz):
            b.free()
    print(f"\n12 random rigs: worst error / quantisation bound {worst:.3f}")
