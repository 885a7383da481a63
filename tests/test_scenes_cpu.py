"""bench.py's synthetic captures, host side (no GPU): the S-uniform twin and the covering rig (benchlib/common.py)."""
import numpy as np

import bench
import oracle_c as oc
import oracle_np as onp


def test_uniform_twin_is_a_counter_hash_of_the_whole_image():
    N, H, W = 14, 40, 64
    whole = onp.synth_uniform(N, H, W, seed=3)
    assert whole.shape == (N, H, W) and whole.dtype == np.uint8
    assert np.array_equal(onp.synth_uniform(N, H, W, seed=3, row0=11, rows=9), whole[:, 11:20])
    assert not np.array_equal(onp.synth_uniform(N, H, W, seed=4), whole)
    big = onp.synth_uniform(44, 64, 256, seed=0)
    h, v = oc.decode(big)
    frac = ((h != -1) & (v != -1)).mean()
    assert 0.12 < frac < 0.32, frac                    # ~23 % of the pixels of a uniform stack decode


def test_covering_rig_lights_the_camera_image():
    """The headline capture of bench.py: with the covering rig > 90 % of the camera pixels see the projected pattern (SURVEY.md 8(d)'s own
    calibration reaches 9-28 %), and the clean capture decodes to the projected codes."""
    for name in ("c3_4096x3000x44", "c2_1920x1080x44", "c1_1280x720x42"):
        W, H, pw, ph, N = bench.WORKLOADS[name]
        s = 8
        calib = list(bench.calibration(W, H, pw, ph, rig="covering"))
        calib[0] = calib[0].copy()
        calib[0][:2] /= s
        cfg = bench.SCENES["physical"]
        st, h, v, _ = onp.synth_physical(N, H // s, W // s, (pw, ph), calib, seed=1, noise=cfg["noise"], gains=cfg["gains"], r2_max=cfg["r2_max"])
        lit = h != -1
        assert lit.mean() > 0.9, (name, lit.mean())
        assert h.max() < 1024 and v.max() < min(1024, ph)
        gh, gv = oc.decode(st)
        ok = (gh != -1) & (gv != -1)
        assert ok.mean() > 0.8 and not (ok & ~lit).any()
        assert np.array_equal(gh[ok], h[ok]) and np.array_equal(gv[ok], v[ok])
    assert bench.SCENES["physical"]["rig"] == "covering" and bench.SCENES["s-scene"]["rig"] == "survey"
    K0 = bench.calibration(4096, 3000, 1920, 1200)
    K1 = bench.calibration(4096, 3000, 1920, 1200, rig="survey")
    assert all(np.array_equal(a, b) for a, b in zip(K0, K1))          # the default stays SURVEY.md 8(d)'s calibration
