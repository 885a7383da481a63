#!/usr/bin/env python3
"""Decode kernel and fused scan on the planar stack [N][H][W] against the tile-interleaved one [tile][N][2^k] (slgc_tune "stack_tile_log2"),
interleaved in one process on rotated stacks, per-launch HIP-event times (median).  VERDICT r5 item 1: is the layout worth adopting?
  python tools/time_tiled.py [--workload c3_4096x3000x44] [--k 10 12 13 16] [--scene s-scene]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scanner import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3_4096x3000x44")
ap.add_argument("--k", type=int, nargs="+", default=[10, 12, 13, 16])
ap.add_argument("--scene", default="s-scene", choices=sorted(bench.SCENES))
ap.add_argument("--launches", type=int, default=60)
args = ap.parse_args()
W, H, pw, ph, N = bench.WORKLOADS[args.workload]
px = W * H
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, pw, ph))
nst = max(2, -(-(300 << 20) // (N * px)))                       # rotate over > 256 MB: nothing served from the Infinity Cache
planar = [ctx.alloc(N * px) for _ in range(nst)]
for i, b in enumerate(planar):
    bench.synth_into(ctx, args.scene, b.ptr, px, N, H, W, (pw, ph), 1 + i, row0=0, rows=H)
maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)


def timed(fn):
    for i in range(6):
        fn(i)
    ctx.synchronize()
    ctx.prof_begin(args.launches + 8)
    for i in range(args.launches):
        fn(i)
    ctx.prof_end()
    return float(np.median(ctx.prof_samples())) * 1e3


rows = []
layouts = [(0, planar, N * px)]
for k in args.k:
    tb = ctx.tiled_stack_bytes(N, px, k)
    tl = [ctx.alloc(tb) for _ in range(nst)]
    for src, dst in zip(planar, tl):
        ctx.tile_stack_dev(src.ptr, px, N, px, k, dst.ptr)
    layouts.append((k, tl, tb))
ctx.synchronize()
ref = None
for rep in range(3):
    for k, stacks, stride in layouts:
        ctx.tune("stack_tile_log2", k)
        ps = (1 << k) if k else px
        d = timed(lambda i: ctx.decode_dev(stacks[i % nst].ptr, 1, stride, ps, N, H, W, maps.at(0), maps.at(px * 2)))
        f = timed(lambda i: ctx.scan_dev(stacks[i % nst].ptr, 1, stride, ps, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2)))
        ctx.synchronize()
        dig = int(maps.download((px * 2,), np.int16).astype(np.int64).sum()) ^ int(xyz.download((px * 3,), np.uint32)[::97].astype(np.int64).sum())
        ref = dig if ref is None else ref
        rows.append((rep, k, d, f, dig == ref))
ctx.tune("stack_tile_log2", 0)
print(f"{args.workload}, {args.scene}: median launch of {args.launches}, three interleaved passes; frac = (N + 4 | N + 12) B/px over 8 TB/s")
for k in [0] + args.k:
    ds = [r[2] for r in rows if r[1] == k]
    fs = [r[3] for r in rows if r[1] == k]
    ok = all(r[4] for r in rows if r[1] == k)
    d, f = float(np.median(ds)), float(np.median(fs))
    name = "planar [N][H][W]" if k == 0 else f"tiled, plane piece {1 << k} B (k = {k})"
    print(f"  {name:38s} decode {d:7.2f} us  frac {(N + 4) * px / d / 8e6:.4f}   fused {f:7.2f} us  frac {(N + 12) * px / f / 8e6:.4f}   "
          f"passes decode {' '.join(f'{x:.1f}' for x in ds)} | fused {' '.join(f'{x:.1f}' for x in fs)}   results {'identical' if ok else 'DIFFER'}")
