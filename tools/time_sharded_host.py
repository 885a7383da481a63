#!/usr/bin/env python3
"""Host-side issue time vs wall time of the pipelined sharded scan at nranks = 1 (is the Python / RCCL enqueue path the bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd")); sys.path.insert(0, ROOT)
from scanner import _native, sharded
import bench
W, H, PW, PH, N = 4096, 3000, 1920, 1200, 44
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, PW, PH))
ctx.comm_init(0, 1, _native.Context.comm_unique_id())
stacks = []
for b in range(2):
    s = ctx.alloc(N * W * H); ctx.synth_scene_dev(s.ptr, W * H, N, H, W, seed=1 + b); stacks.append(s)
sc = sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), sharded.ShardPlan(H, W, 1), (PW, PH), N, mode=1)
for mode in ("submit", "scan_sharded_dev"):
    h, v, x = sc._sets[0][0], sc._sets[0][1], sc.xyz_full
    def step(i):
        if mode == "submit":
            sc.submit(stacks[i % 2].ptr, W * H)
        else:
            ctx.scan_sharded_dev(stacks[i % 2].ptr, 1, N * W * H, W * H, N, H, W, (PW, PH), h.ptr, v.ptr, x.ptr, mode=1)
    for i in range(10): step(i)
    sc.flush(); ctx.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K): step(i)
    t1 = time.perf_counter()
    sc.flush(); ctx.synchronize()
    t2 = time.perf_counter()
    print(f"{mode:18s} host issue {1e6 * (t1 - t0) / K:7.1f} us/scan   wall {1e6 * (t2 - t0) / K:7.1f} us/scan", flush=True)
ctx.comm_destroy(); ctx.close()
