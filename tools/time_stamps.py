#!/usr/bin/env python3
"""Where a single launch of the fused kernel spends its time, from inside: the stamp build (make -C 3dscanner-graycode_amd variant NAME=stamps EXTRA=-DSLGC_STAMPS)
writes five s_memrealtime stamps per wave -- started / thresholds done / every frame consumed / tail done, stores issued / stores acknowledged -- and this
prints, relative to the first wave's start, when the k-th percentile of the waves reaches each of them.
   SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_stamps.so python tools/time_stamps.py [workload ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scanner import _native  # noqa: E402

lib = _native.lib()
lib.slgc_diag_stamps.restype = C.c_int
lib.slgc_diag_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
for wl in (sys.argv[1:] or ["c1_1280x720x42", "c2_1920x1080x44", "c3_4096x3000x44"]):
    W, H, pw, ph, N = bench.WORKLOADS[wl]
    px = W * H
    ctx = _native.Context(0)
    scene = "physical"
    ctx.set_calibration(*bench.calibration(W, H, pw, ph, rig=bench.SCENES[scene]["rig"]))
    stacks = []
    for b in range(max(2, -(-300_000_000 // (N * px)))):
        s = ctx.alloc(N * px)
        bench.synth_into(ctx, scene, s.ptr, px, N, H, W, (pw, ph), 1 + b)
        stacks.append(s)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    for i in range(40):
        ctx.scan_dev(stacks[i % len(stacks)].ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2))
    rows = []
    for rep in range(9):
        ctx.scan_dev(stacks[rep % len(stacks)].ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2))
        n = C.c_size_t()
        nw = (px // 4 + 127) // 128 * 2
        buf = np.zeros((nw, 5), np.uint64)
        rc = lib.slgc_diag_stamps(ctx._h, buf.ctypes.data_as(C.c_void_p), nw, C.byref(n))
        assert rc == 0 and n.value == nw, (rc, n.value, nw)
        live = buf[:, 0] != 0
        t = (buf[live].astype(np.int64) - int(buf[live, 0].min())) * 0.01          # microseconds (100 MHz)
        rows.append([np.percentile(t[:, k], q) for k in range(5) for q in (0, 50, 99, 100)])
    r = np.median(np.array(rows), axis=0).reshape(5, 4)
    names = ["wave started", "thresholds done", "every frame consumed", "tail done, stores issued", "stores acknowledged"]
    print(f"{wl}: {int(live.sum())} waves; microseconds after the first wave started (min / median / p99 / max over the waves; median of 9 launches)")
    for k in range(5):
        print(f"   {names[k]:26s} {r[k, 0]:7.2f} {r[k, 1]:7.2f} {r[k, 2]:7.2f} {r[k, 3]:7.2f}")
    ctx.close()
