#!/usr/bin/env python3
"""Does the decode kernel's time depend on where its buffers sit?  One process, fresh allocations per trial."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
from scanner import _native
W, H, N = 4096, 3000, 44
plane = W * H
ctx = _native.Context(0)

def timed(stacks, out, off_h, off_v, pstride, iters=40):
    for i in range(4):
        ctx.decode_dev(stacks[i % len(stacks)].ptr, 1, N * pstride, pstride, N, H, W, out.at(off_h), out.at(off_v))
    ctx.synchronize()
    ctx.prof_begin(iters + 1)
    for i in range(iters):
        ctx.decode_dev(stacks[i % len(stacks)].ptr, 1, N * pstride, pstride, N, H, W, out.at(off_h), out.at(off_v))
    ms, n = ctx.prof_end()
    return ms / n * 1e3

for trial in range(4):
    pad = [0, 0, 256, 256][trial]
    pstride = plane + pad
    stacks = []
    for b in range(2):
        s = ctx.alloc(N * pstride + 4096)
        ctx.synth_scene_dev(s.ptr, pstride, N, H, W, seed=1 + b)
        stacks.append(s)
    out = ctx.alloc(plane * 4 + (8 << 20))
    res = []
    for off in (0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096):
        t = timed(stacks, out, off, off + plane * 2 + (0 if off == 0 else 256), pstride)
        res.append(f"{off:>8d}:{t:6.1f}")
    print(f"trial {trial} pad {pad} stack VAs {[hex(s.ptr) for s in stacks]} out {hex(out.ptr)} | " + "  ".join(res), flush=True)
    for s in stacks:
        s.free()
    out.free()
    junk = ctx.alloc((trial + 1) * 37 * (1 << 20))      # shift the next trial's addresses
ctx.close()
