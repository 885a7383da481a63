#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer API (NumPy in, NumPy out) at BASELINE sizes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd")); sys.path.insert(0, ROOT)
from scanner import _native
from scanner.pipeline import scan_to_cloud
import bench
for (W, H, PW, PH, N) in ((1920, 1080, 1920, 1080, 44), (4096, 3000, 1920, 1200, 44)):
    ctx = _native.Context(0)
    d = ctx.alloc(N * W * H); ctx.synth_scene_dev(d.ptr, W * H, N, H, W); ctx.synchronize()
    st = d.download((N, H, W), np.uint8); d.free()
    mp = W * H / 1e6
    for name, fn in (("decode(uint8)", lambda: ctx.decode(st)),
                     ("get_codes(uint8)", lambda: ctx.codes(st)),):
        for _ in range(5):
            fn()                                          # steady state of a loop: result buffers recycled (_native._ResultPool)
        t = time.perf_counter(); fn(); dt = time.perf_counter() - t
        print(f"{W}x{H}x{N} {name:18s} {dt*1e3:8.1f} ms  {mp/dt:8.1f} Mpix/s", flush=True)
    if W <= 1920:
        f64 = st.astype(np.float64)
        for _ in range(5):
            ctx.decode(f64)
        t = time.perf_counter(); ctx.decode(f64); dt = time.perf_counter() - t
        print(f"{W}x{H}x{N} decode(float64)     {dt*1e3:8.1f} ms  {mp/dt:8.1f} Mpix/s", flush=True)
        del f64
    K, cd, pk, pd, R, T = bench.calibration(W, H, PW, PH)
    white = np.repeat(st[1][:, :, None], 3, axis=2)
    pm = pk.copy()
    for thr in (None,):
        for _ in range(5):
            scan_to_cloud(st, K, cd, (PW, PH), (PW, PH), pm, pd, R, T, img_white=white, ctx=ctx)
        t = time.perf_counter(); out = scan_to_cloud(st, K, cd, (PW, PH), (PW, PH), pm, pd, R, T, img_white=white, ctx=ctx); dt = time.perf_counter() - t
        print(f"{W}x{H}x{N} scan_to_cloud      {dt*1e3:8.1f} ms  {mp/dt:8.1f} Mpix/s  ({out['pts'].shape[1]} points, x-major, float64)", flush=True)
    ctx.close()
