import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd")); sys.path.insert(0, ROOT)
import bench
from scanner import _native
case = sys.argv[1]
ctx = _native.Context(0)
W, H, pw, ph, N = bench.WORKLOADS["c3_4096x3000x44"]
calib = bench.calibration(W, H, pw, ph)
px = W * H
if "prior" in case:                       # what the test module did before: a ragged product on another calibration, then c3 scans
    K = np.array([[300.0, 0, 166], [0, 300.0, 38.5], [0, 0, 1]])
    _, cd, pk, pd, R, T = bench.calibration(1920, 1080, 300, 200)
    ctx.set_calibration(K, cd, pk, pd, R, T)
    st = ctx.alloc(44 * 332 * 77); ctx.synth_scene_dev(st.ptr, 332 * 77, 44, 77, 332, seed=3)
    m, x = ctx.alloc(332 * 77 * 4), ctx.alloc(332 * 77 * 12)
    ctx.scan_dev(st.ptr, 1, 44 * 332 * 77, 332 * 77, 44, 77, 332, 0, (300, 200), x.ptr, None, m.at(0), m.at(332 * 77 * 2))
    l = ctx.alloc_cloud_lists(332 * 77, colors=False)
    ctx.cloud_lists_dev(m.at(0), m.at(332 * 77 * 2), x.ptr, None, 332, 77, (300, 200), l)
    print("prior total", l.total(), flush=True)
ctx.set_calibration(*calib)
if "nodes0" in case:
    ctx.tune("cam_nodes", 0)
stack = ctx.alloc(N * px)
ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=3, noise=3, shadow=True)
maps = ctx.alloc(px * 4 + 64)
white = ctx.alloc(px * 3).upload(np.random.default_rng(7).integers(0, 256, (H, W, 3), dtype=np.uint8))
lists = ctx.alloc_cloud_lists(px, colors="nocol" not in case, points="nopts" not in case)
ctx.synchronize(); print("setup ok", flush=True)
if "decode_only" in case:
    ctx.decode_dev(stack.ptr, 1, N * px, px, N, H, W, maps.at(0), maps.at(px * 2)); ctx.synchronize(); print("decode ok", flush=True)
    ctx.build_ray_tables_dev(H, W, 0, (pw, ph)); ctx.synchronize(); print("tables ok", ctx.ray_table_info(), flush=True)
    ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), None, white.ptr, W, H, (pw, ph), lists); print("lists total", lists.total(), flush=True)
else:
    ctx.cloud_dev(stack.ptr, 1, N * px, px, N, H, W, (pw, ph), white.ptr, lists, d_h=maps.at(0), d_v=maps.at(px * 2))
    print("total", lists.total(), ctx.last_scan_path(), flush=True)
