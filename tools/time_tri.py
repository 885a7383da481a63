#!/usr/bin/env python3
"""Time the dense triangulation kernel alone (HIP events) on decoded maps of the synthetic C3 scene."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd")); sys.path.insert(0, ROOT)
from scanner import _native
import bench
W, H, PW, PH, N = 4096, 3000, 1920, 1200, 44
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, PW, PH))
st = ctx.alloc(N * W * H); ctx.synth_scene_dev(st.ptr, W * H, N, H, W)
maps = ctx.alloc(W * H * 4); xyz = ctx.alloc(W * H * 12)
ctx.decode_dev(st.ptr, 1, N * W * H, W * H, N, H, W, maps.at(0), maps.at(W * H * 2)); ctx.synchronize()
for mode in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,0").split(",")]:
    for i in range(3):
        ctx.triangulate_maps_dev(maps.at(0), maps.at(W * H * 2), H, W, 0, (PW, PH), xyz.ptr, None, mode=mode)
    ctx.event_record(0)
    for i in range(20):
        ctx.triangulate_maps_dev(maps.at(0), maps.at(W * H * 2), H, W, 0, (PW, PH), xyz.ptr, None, mode=mode)
    ctx.event_record(1)
    print(f"mode {mode}: {ctx.event_elapsed_ms(0, 1) / 20 * 1e3:.1f} us", flush=True)
