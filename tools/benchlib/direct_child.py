#!/usr/bin/env python3
"""bench.py, N > 1: the direct exchange (csrc/direct.hip) timed in CHILD processes, one per rank.

The direct exchange has never run between two GPUs (the builder's boxes have one): its first run on a node is inside the driver's scaling
bench.  A hang there is bounded by GPU-side deadlines -- but a fault (a peer mapping that does not behave, a memory access fault, which
the HSA runtime answers with abort()) would take the rank down, and with it the one JSON line the run owes.  So every rank of bench.py
starts this script as a child after its own (RCCL) region is measured: the children build their own contexts on the rank's GPU, generate
the same synthetic captures, run the pipelined sharded scan over the direct exchange, compare the reassembled maps with the digest the
parent's main strategy produced, and print one JSON object.  A child that dies or hangs costs its parent an "error" entry, nothing else.

  python3 tools/benchlib/direct_child.py --rank R --nranks G --device D --key K --cam WxH --proj WxH --frames N --scene S --wire hv24
          --buffers B --plane-pad P --steps K --last-stack I --main-digest 0x... [--mode M]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "3dscanner-graycode_amd"), os.path.join(ROOT, "tools"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--nranks", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--key", required=True)
    ap.add_argument("--cam", required=True)
    ap.add_argument("--proj", required=True)
    ap.add_argument("--frames", type=int, required=True)
    ap.add_argument("--scene", default="physical")
    ap.add_argument("--wire", default="hv24")
    ap.add_argument("--buffers", type=int, default=2)
    ap.add_argument("--plane-pad", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--last-stack", type=int, default=0)
    ap.add_argument("--main-digest", default="")
    ap.add_argument("--mode", type=int, default=None)
    a = ap.parse_args()
    os.environ.setdefault("SLGC_DIRECT_TIMEOUT_S", "5")          # an extra after the counted region: a healthy exchange takes well under a millisecond
    from benchlib.common import SCENES, calibration, digest64, synth_into
    from scanner import _native, sharded

    cam_w, cam_h = (int(x) for x in a.cam.split("x"))
    proj_w, proj_h = (int(x) for x in a.proj.split("x"))
    N, G, rank = a.frames, a.nranks, a.rank
    mode = _native.TRI_ALGEBRAIC if a.mode is None else a.mode
    ctx = _native.Context(a.device)
    ctx.set_calibration(*calibration(cam_w, cam_h, proj_w, proj_h, rig=SCENES[a.scene]["rig"]))
    plan = sharded.ShardPlan(cam_h, cam_w, G)
    row0, rows = plan.band(rank)
    plane = rows * cam_w + a.plane_pad
    stacks = []
    for b in range(max(1, a.buffers)):                           # the parent's captures: same generator, same seeds, same band
        s = ctx.alloc(max(16, N * plane))
        if rows:
            synth_into(ctx, a.scene, s.ptr, plane, N, cam_h, cam_w, (proj_w, proj_h), 1 + b, row0=row0, rows=rows)
        stacks.append(s)
    exch = sharded.DirectExchange(ctx, rank, G, a.key)
    sc = sharded.ShardedScanner(ctx, exch, plan, (proj_w, proj_h), N, mode=mode, exchange_kind="maps", wire=a.wire)
    K = max(1, a.steps)
    for i in range(3):
        sc.submit(stacks[i % len(stacks)].ptr, plane)
    sc.flush()
    ctx.synchronize()
    exch.barrier()
    t0 = time.perf_counter()
    for i in range(K):
        sc.submit(stacks[i % len(stacks)].ptr, plane)
    sc.flush()
    ctx.synchronize()
    exch.barrier()
    el = max(exch.allgather_i64(int((time.perf_counter() - t0) * 1e9))) * 1e-9
    sc.submit(stacks[a.last_stack % len(stacks)].ptr, plane)    # the stack the parent's main strategy finished on
    sc.flush()
    h, v, _ = sc.fetch_dense()
    mine = f"{digest64(h, v):016x}"
    want = a.main_digest.lower().replace("0x", "")
    same = exch.allgather_i64(1 if (not want or mine == want) else 0)
    out = {"value": round(cam_w * cam_h / 1e6 * K / el, 1), "unit": "Mpixels/s", "steps": K, "exchange_impl": "direct",
           "bytes_per_pixel_on_the_links": 3 if a.wire == "hv24" else 4, "maps_equal_main_strategy_on_every_rank": bool(all(same)),
           "compared_with_main_digest": bool(want), "digest": mine, "in": "child processes, one per rank (tools/benchlib/direct_child.py)"}
    exch.barrier()
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
