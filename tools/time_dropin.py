#!/usr/bin/env python3
"""The reference's two scripts as they would run unchanged on this package (way A), call by call, PCIe included:
src/3-capture_decode.py:75-100 (get_codes per run, merge, gray_to_decimal loops -> codes_to_pixels here) and src/4-triangulate.py:50-71
(Triangulate, get_cam_proj_pts, triangulate, filter_3d_pts).  python tools/time_dropin.py [--workload c3_4096x3000x44]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scanner import _native  # noqa: E402
from scanner.grayCode.decode_codes import codes_to_pixels, get_codes  # noqa: E402
from scanner.triangulation import Triangulate  # noqa: E402
from scanner.triangulation.triangulate import Triangulation  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3_4096x3000x44")
ap.add_argument("--float64", action="store_true", help="hand get_codes the float64 stack the reference script builds (8x the host memory)")
args = ap.parse_args()
W, H, pw, ph, N = bench.WORKLOADS[args.workload]
px = W * H
ctx = _native.default_context()
d = ctx.alloc(N * px)
ctx.synth_scene_dev(d.ptr, px, N, H, W, seed=1, noise=3, shadow=True)
st = d.download((N, H, W), np.uint8)
d.free()
if args.float64:
    st = st.astype(np.float64)
white = np.repeat(st[1][:, :, None], 3, axis=2).astype(np.uint8)
K, cd, pk, pd, R, T = bench.calibration(W, H, pw, ph)


def clock(label, fn, times):
    t0 = time.perf_counter()
    out = fn()
    times.append((label, (time.perf_counter() - t0) * 1e3))
    return out


for rep in range(8):
    times = []
    hc, vc = clock("get_codes(images)                      3:75", lambda: get_codes(st), times)
    hp, vp = clock("merge + gray_to_decimal loops          3:95-100", lambda: codes_to_pixels(hc, vc), times)
    pm = pk.copy()
    tri = clock("Triangulate(...)                        4:50", lambda: Triangulate(hp, vp, (W, H), K, cd, (pw, ph), (pw, ph), pm, pd, R, T, None), times)
    cam, proj, col = clock("get_cam_proj_pts(img_white)             4:62", lambda: tri.get_cam_proj_pts(white), times)
    pts = clock("triangulate(cam_pts, proj_pts)          4:63", lambda: tri.triangulate(cam, proj), times)
    fp, fc = clock("filter_3d_pts(pts, colors, 0.5)         4:71", lambda: tri.filter_3d_pts(pts, col, threshold=0.5), times)
    pm2 = pk.copy()
    t0 = time.perf_counter()
    fused = Triangulation(hp, vp, (W, H), K, cd, (pw, ph), (pw, ph), pm2, pd, R, T, None).compute(white, threshold=0.5)
    t_fused = (time.perf_counter() - t0) * 1e3
    assert np.array_equal(fused[0], fp) and np.array_equal(fused[1], fc)
    del hc, vc, hp, vp, cam, proj, col, pts, fused
    print(f"pass {rep + 1}: {sum(t for _, t in times):.1f} ms" + ("   (one scan in a fresh process, as the reference's scripts run)" if rep == 0 else ""))
total = sum(t for _, t in times)
print(f"{args.workload}, {'float64' if args.float64 else 'uint8'} stack, last pass ({fp.shape[1]} points kept):")
for label, t in times:
    print(f"  {label:58s} {t:8.1f} ms")
three = sum(t for label, t in times if "4:6" in label or "4:7" in label)
print(f"  {'Triangulation.compute(img_white, threshold=0.5): the three calls of 4:62-71 as ONE device-resident call':58s} {t_fused:8.1f} ms = {t_fused / three:.2f} x the three calls ({three:.1f} ms), bit-identical")
print(f"  {'both scripts, call by call':58s} {total:8.1f} ms = {px / 1e6 / (total * 1e-3):.0f} Mpixels/s   (reference: ~0.05 Mpixels/s)")
