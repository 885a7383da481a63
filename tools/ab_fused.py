#!/usr/bin/env python3
"""Interleaved same-process A/B of the scan kernels over slgc_tune knobs (cdna_hip_programming.md rule 24: N variants x M rounds in ONE
process, report the distribution).  Every configuration's maps + XYZ are hashed: all must be identical.

  python tools/ab_fused.py --knobs "fuse_tail=0,1;proj_tile=0,1" [--rounds 6] [--iters 40] [--pipeline fused|split|decode] [--workload c3_4096x3000x44]
  SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_diag.so python tools/ab_fused.py --knobs "fuse_abl=0,5,6,7,8"      (diagnostic build: ablations)
"""
import argparse
import hashlib
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scanner import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--knobs", default="fuse_tail=0,1")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--pipeline", default="fused", choices=["fused", "split", "decode"])
ap.add_argument("--workload", default="c3_4096x3000x44")
ap.add_argument("--scene", default="physical", choices=sorted(bench.SCENES), help="synthetic capture (bench.py --scene)")
ap.add_argument("--preheat", type=float, default=0.2, help="seconds of untimed launches first (0 under a counter profiler)")
args = ap.parse_args()

W, H, pw, ph, N = bench.WORKLOADS[args.workload]
px = W * H
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, pw, ph, rig=bench.SCENES[args.scene]["rig"]))
stacks = []
for b in range(max(2, -(-300_000_000 // (N * px)))):
    s = ctx.alloc(N * px)
    bench.synth_into(ctx, args.scene, s.ptr, px, N, H, W, (pw, ph), 1 + b)
    stacks.append(s)
maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
names, values = [], []
for part in args.knobs.split(";"):
    k, v = part.split("=")
    names.append(k.strip())
    values.append([int(x) for x in v.split(",")])
configs = list(itertools.product(*values))
mode = _native.TRI_ALGEBRAIC | (_native.TRI_SPLIT if args.pipeline == "split" else 0)


def launch(i):
    s = stacks[i % len(stacks)]
    if args.pipeline == "decode":
        ctx.decode_dev(s.ptr, 1, N * px, px, N, H, W, maps.at(0), maps.at(px * 2), variant=VARIANT[0])
    else:
        ctx.scan_dev(s.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)


VARIANT = [0]          # pseudo-knob "variant" (decode pipeline only): slgc_decode_dev's kernel variant code (0 = the library's choice), e.g. 1108128 = 8 pixels per lane


def apply(cfg):
    for k, v in zip(names, cfg):
        if k == "variant":
            VARIANT[0] = v
        else:
            ctx.tune(k, v)


t_end = __import__("time").perf_counter() + args.preheat          # pre-heat
i = 0
while __import__("time").perf_counter() < t_end:
    launch(i)
    i += 1
ctx.synchronize()
samples = {c: [] for c in configs}
steps = {c: [] for c in configs}
digests = {}
for r in range(args.rounds):
    for cfg in configs:
        apply(cfg)
        for i in range(3):
            launch(i)
        ctx.synchronize()
        ctx.prof_begin(args.iters + 1, 1)
        ctx.event_record(0)
        for i in range(args.iters):
            launch(i)
        ctx.event_record(1)
        ms, n = ctx.prof_end()
        samples[cfg].extend(ctx.prof_samples().tolist())
        steps[cfg].append(ctx.event_elapsed_ms(0, 1) / args.iters)
        if r == 0:
            launch(0)
            ctx.synchronize()
            hsh = hashlib.blake2b(digest_size=8)
            hsh.update(maps.download((px * 2,), np.int16).data)
            if args.pipeline != "decode":
                hsh.update(np.nan_to_num(xyz.download((px * 3,), np.float32), nan=-1.0).data)
            digests[cfg] = hsh.hexdigest()
print(f"{args.workload} scene={args.scene} pipeline={args.pipeline} rounds={args.rounds} iters={args.iters}  (first kernel of the step: HIP events bound to its dispatch; step: whole step)")
per_px = (N + 4) if args.pipeline in ("split", "decode") else (N + 12)
for cfg in configs:
    s = np.sort(np.array(samples[cfg])) * 1e3
    st = np.sort(np.array(steps[cfg])) * 1e3
    med = float(np.median(s))
    print("  " + " ".join(f"{k}={v}" for k, v in zip(names, cfg)) + f":  kernel median {med:7.2f} us  min {s[0]:7.2f}  p95 {s[int(0.95 * (len(s) - 1))]:7.2f}"
          f"  | step median {float(np.median(st)):7.2f} us min {st[0]:7.2f} | frac {per_px * px / (med * 1e-6) / 8e12:.3f} | digest {digests[cfg]}")
print("ray tables: node table in use %s, max relative difference of the interpolated rays %.3g" % ctx.ray_table_info())
if len(set(digests.values())) != 1 and "cam_nodes" not in names:      # cam_nodes is the one knob that is not bit-neutral (include/slgc.h)
    print("RESULTS DIFFER BETWEEN CONFIGURATIONS")
    sys.exit(1)
ctx.close()
