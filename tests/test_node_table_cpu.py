"""CPU check of the arithmetic behind the camera node table (csrc/tri_math.h: cam_rays_from_nodes): the cubic through four exact float32 nodes,
formed as e0 + (Lm*dm + L1*d1 + L2*d2) in float32, against the exactly evaluated float32 ray of every pixel of sampled rows -- for the three
calibrations bench.py uses.  (The GPU measures the same thing over every pixel when it builds the table: slgc_ray_table_info.)"""
import os
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT

sys.path.insert(0, ROOT)
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import oracle_np as onp  # noqa: E402

WEIGHTS = {1: (-0.0546875, 0.2734375, -0.0390625), 2: (-0.0625, 0.5625, -0.0625), 3: (-0.0390625, 0.8203125, -0.0546875)}   # Lagrange, t = j / 4


def test_lagrange_weights_are_the_cubic_through_nodes_minus1_0_1_2():
    for j, (wm, w1, w2) in WEIGHTS.items():
        t = j / 4.0
        assert wm == -t * (t - 1) * (t - 2) / 6 and w1 == -(t + 1) * t * (t - 2) / 2 and w2 == (t + 1) * t * (t - 1) / 6
        assert float(np.float32(wm)) == wm and float(np.float32(w1)) == w1 and float(np.float32(w2)) == w2      # exact in float32


@pytest.mark.parametrize("workload", ["c3_4096x3000x44", "c2_1920x1080x44", "c1_1280x720x42"])
def test_interpolated_rays_stay_within_two_ulp(workload):
    W, H, pw, ph, _ = bench.WORKLOADS[workload]
    K, cd, _, _, R, _ = bench.calibration(W, H, pw, ph)
    rows = np.arange(0, H, max(1, H // 24))
    xn = np.arange(-4, W + 8, 4, dtype=np.float32)                        # nodes -1 .. W / 4 + 1 of a row
    worst = 0.0
    for y in rows:
        nodes = onp.undistort_points(np.stack([xn, np.full_like(xn, y)], 1), K, cd, R).reshape(-1, 2)           # exact float32 rays at the nodes
        exact = onp.undistort_points(np.stack([np.arange(W, dtype=np.float32), np.full(W, y, np.float32)], 1), K, cd, R).reshape(-1, 2)
        g = np.arange(W // 4)
        for comp in (0, 1):
            e0 = nodes[g + 1, comp]
            dm, d1, d2 = nodes[g, comp] - e0, nodes[g + 2, comp] - e0, nodes[g + 3, comp] - e0
            assert np.array_equal(e0, exact[4 * g, comp])                  # pixel 4 g is the node itself
            for j, (wm, w1, w2) in WEIGHTS.items():
                s = np.float32(wm) * dm
                s = (np.float32(w1) * d1 + s).astype(np.float32)           # mul + add where the kernel has an fma: slightly pessimistic
                s = (np.float32(w2) * d2 + s).astype(np.float32)
                r = (e0 + s).astype(np.float32)
                t = exact[4 * g + j, comp]
                err = np.abs(r.astype(np.float64) - t.astype(np.float64)) / np.maximum(1.0, np.abs(t))
                worst = max(worst, float(err.max()))
                big = np.abs(t) >= 1e-3                                    # components the kernels use (smaller ones re-read the exact table)
                if big.any():
                    rel = np.abs(r[big].astype(np.float64) - t[big]) / np.abs(t[big])
                    assert rel.max() <= 4e-6
    assert worst <= 2.4e-7, worst                                          # the acceptance limit of ensure_luts (2 ulp of a float32 in [1, 2))
