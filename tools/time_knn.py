#!/usr/bin/env python3
"""Time the k-NN mean-distance kernel (outlier-removal arithmetic) on the cloud of the synthetic 4096x3000 scan."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd")); sys.path.insert(0, ROOT)
from scanner import _native
from scanner import pointcloud as pc
import bench
W, H, PW, PH, N = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,3000,1920,1200,44").split(","))
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, PW, PH))
st = ctx.alloc(N * W * H); ctx.synth_scene_dev(st.ptr, W * H, N, H, W)
xyz = ctx.alloc(W * H * 12)
ctx.scan_dev(st.ptr, 1, N * W * H, W * H, N, H, W, 0, (PW, PH), xyz.ptr, None, mode=1); ctx.synchronize()
cloud = xyz.download((H * W, 3), np.float32)
cloud = cloud[np.isfinite(cloud[:, 0])]
st.free(); xyz.free()
print("points:", len(cloud), "bbox", cloud.min(0), cloud.max(0), flush=True)
for rep in range(2):
    t = time.perf_counter(); avg = ctx.knn_mean_distance(cloud, 20); dt = time.perf_counter() - t
    print(f"knn_mean_distance k=20: {dt:.3f} s  ({len(cloud) / dt / 1e6:.1f} Mpts/s incl. PCIe)  mean {avg.mean():.3e}", flush=True)
t = time.perf_counter(); inl, ind = pc.remove_statistical_outlier(cloud, ctx=ctx); dt = time.perf_counter() - t
print(f"remove_statistical_outlier: {dt:.3f} s, kept {len(ind)} of {len(cloud)}")
