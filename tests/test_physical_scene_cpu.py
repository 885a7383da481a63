"""The physically consistent synthetic capture on the CPU (no GPU): the NumPy twin of csrc/synth.hip's generator is consistent with the
reference's own method as the oracle restates it -- decoding its frames gives back the projector pixel that was encoded, nothing decodes where
the projector does not reach, and the triangulated point lies within the half-pixel code-quantisation bound of the TRUE surface point
(pure geometry of the truth, /root/reference/scanner/triangulation/triangulate.py:91-94 differentiated; tests/test_gpu_physical.py holds the
device kernels to the same bound)."""
import numpy as np
import pytest

import oracle_c as oc
import oracle_np as onp


def rig(W, H, pw, ph, toe_deg=-24.0, base=0.20):
    K = np.array([[2.2 * W, 0, W / 2 + 3.0], [0, 2.2 * W, H / 2 - 2.0], [0, 0, 1]])
    cd = np.array([-0.08, 0.05, 0.0007, -0.0004, 0.01])
    pk = np.array([[1.9 * pw, 0, pw / 2 - 5.0], [0, 1.9 * pw, ph / 2 + 4.0], [0, 0, 1]])
    pd = np.array([0.04, -0.06, -0.0005, 0.0008, 0.02])
    th = np.deg2rad(toe_deg)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T = np.array([[base], [0.005], [0.045]])
    return K, cd, pk, pd, R, T


def bound(truth, T, pk, slack=1.3):
    t = np.asarray(T, dtype=np.float64).reshape(3)
    tl = np.linalg.norm(t)
    ln = np.linalg.norm(truth, axis=-1)
    Q = truth + t
    cos_a = -(truth @ t) / (ln * tl)
    cos_b = (Q @ t) / (np.linalg.norm(Q, axis=-1) * tl)
    sin_a, sin_b = np.sqrt(1 - cos_a ** 2), np.sqrt(1 - cos_b ** 2)
    return ln * sin_a / (sin_b * (sin_a * cos_b + cos_a * sin_b)) * slack * 0.5 * np.sqrt(1.0 / pk[0, 0] ** 2 + 1.0 / pk[1, 1] ** 2)


@pytest.mark.parametrize("shape", [(200, 150, 192, 144, 34), (256, 192, 256, 192, 46), (161, 97, 128, 96, 30)])
def test_twin_round_trips_through_the_oracle(shape):
    W, H, pw, ph, N = shape
    calib = rig(W, H, pw, ph)
    stack, h, v, truth = onp.synth_physical(N, H, W, (pw, ph), calib, seed=3, noise=3)
    lit = h != -1
    assert 0.5 < lit.mean() < 1.0 and np.array_equal(lit, v != -1) and np.array_equal(np.isfinite(truth[..., 0]), lit)
    assert stack.shape == (N, H, W) and stack.dtype == np.uint8
    assert (stack[:, ~lit] <= 15 + 3).all()                                   # unlit pixels sit at ambient level in every frame (+ noise)
    fh, fv, fx = oc.scan_dense(stack, (pw, ph), *calib)
    ok = (fh != -1) & (fv != -1)
    assert not (ok & ~lit).any() and ok.sum() >= 0.98 * lit.sum()
    assert np.array_equal(fh[ok], h[ok]) and np.array_equal(fv[ok], v[ok])
    rec = np.moveaxis(fx, 0, -1)[ok]
    err = np.linalg.norm(rec - truth[ok], axis=1)
    ratio = err / bound(truth[ok], calib[5], calib[2])
    assert ratio.max() <= 1.0 and np.median(ratio) < 0.5
    # a row band of the same image is the same image (the multi-GPU ranks generate only their band)
    band = onp.synth_physical(N, H, W, (pw, ph), calib, seed=3, noise=3, row0=H // 3, rows=H // 4)
    assert np.array_equal(band[0], stack[:, H // 3:H // 3 + H // 4]) and np.array_equal(band[1], h[H // 3:H // 3 + H // 4])


def test_codes_beyond_the_pattern_length_stay_dark():
    """44 frames carry 10 bits per axis: a 1920-pixel projector can only address columns below 1024 (BASELINE.json configs[1], [2])."""
    W, H, pw, ph = 96, 64, 1920, 1080
    calib = rig(W, H, pw, ph)
    h10, v10, _ = onp.synth_physical_codes(H, W, (pw, ph), *calib, code_bits=10)
    h11, v11, _ = onp.synth_physical_codes(H, W, (pw, ph), *calib, code_bits=11)
    assert h11.max() > 1023 and h10.max() <= 1023 and (h10 != -1).sum() < (h11 != -1).sum()
    same = h10 != -1
    assert np.array_equal(h10[same], h11[same]) and np.array_equal(v10[same], v11[same])
