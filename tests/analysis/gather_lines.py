#!/usr/bin/env python3
"""How many DISTINCT 128-byte lines of the projector table (4 bytes per projector pixel, row-major) the 256 pixels of one wave of the fused scan
kernel touch, per synthetic capture -- the bound on what sorting / coalescing a wave's gathers by line could save.  CPU only:
the oracle decodes the capture, the statistic is taken on its maps.
  python tests/analysis/gather_lines.py [--size 1920x1080] [--proj 1920x1200] [--frames 44]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_c as oc  # noqa: E402
import oracle_np as onp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--proj", default="1920x1200")
ap.add_argument("--frames", type=int, default=44)
args = ap.parse_args()
W, H = (int(x) for x in args.size.split("x"))
pw, ph = (int(x) for x in args.proj.split("x"))
N = args.frames
oc.set_threads(8)
for name, st in (("s-uniform", onp.synth_uniform(N, H, W, seed=7)), ("s-scene", onp.synth_scene_int(N, H, W, seed=1)[0])):
    h, v = oc.decode(st)
    ok = ((h != -1) & (v != -1)).reshape(-1)
    line = ((np.minimum(v, ph - 1) * pw + np.minimum(h, pw - 1)) * 4 // 128).reshape(-1)
    n = ok.size // 256 * 256
    okw, lw = ok[:n].reshape(-1, 256), line[:n].reshape(-1, 256)
    lw = np.where(okw, lw, -1)
    srt = np.sort(lw, axis=1)
    distinct = ((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)).sum(axis=1) + (srt[:, 0] >= 0)
    valid = okw.sum(axis=1)
    print(f"{name:10s} {W}x{H}x{N}: {ok.mean() * 100:5.1f} % of the pixels valid; per wave of 256 pixels: {valid.mean():6.1f} gathers, {distinct.mean():6.1f} distinct lines "
          f"({distinct.sum() / max(1, valid.sum()):.3f} lines per gather); all waves: {int(distinct.sum())} line fetches x 128 B = {distinct.sum() * 128 / 1e6:.1f} MB "
          f"for {valid.sum() * 4 / 1e6:.1f} MB of entries")
