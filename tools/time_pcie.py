#!/usr/bin/env python3
"""Where the PCIe-inclusive time of the host-buffer API goes: raw transfers of the sizes `decode()` moves at 4096x3000x44, into fresh and into
touched NumPy arrays, against the call itself."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd")); sys.path.insert(0, ROOT)
from scanner import _native
import bench
W, H, PW, PH, N = bench.WORKLOADS["c3_4096x3000x44"]
px = W * H
ctx = _native.Context(0)
d = ctx.alloc(N * px); ctx.synth_scene_dev(d.ptr, px, N, H, W); ctx.synchronize()
st = d.download((N, H, W), np.uint8)


def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


print(f"H2D {st.nbytes / 1e6:.0f} MB pageable uint8 stack: {t(lambda: d.upload(st)):.1f} ms")
m16 = ctx.alloc(px * 4); m64 = ctx.alloc(px * 16)
print(f"D2H {px * 4 / 1e6:.0f} MB into a fresh array (int16 maps): {t(lambda: m16.download((2, H, W), np.int16)):.1f} ms")
print(f"D2H {px * 16 / 1e6:.0f} MB into a fresh array (int64 maps): {t(lambda: m64.download((2, H, W), np.int64)):.1f} ms")
a16 = m16.download((2, H, W), np.int16)
print(f"host widen int16 -> int64 (NumPy astype, one thread): {t(lambda: a16.astype(np.int64)):.1f} ms")
print(f"np.empty + touch of {px * 16 / 1e6:.0f} MB: {t(lambda: np.empty((2, H, W), np.int64).fill(0)):.1f} ms")
print(f"decode(uint8): {t(lambda: ctx.decode(st)):.1f} ms")
print(f"codes(uint8): {t(lambda: ctx.codes(st)):.1f} ms")
