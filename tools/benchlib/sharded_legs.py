"""bench.py: report and self-verification of the row-sharded scan (N > 1 or --force-sharded)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

from .common import digest64, synth_into


def sharded_report(ctx, scanner, args, G, rank, stacks, plane, N, rows, cam_w, cam_h, row0, proj_size, mode, elapsed):
    """compute-only vs with-exchange rates of the sharded scan and the bytes each rank puts on / takes off the links."""
    K = max(5, args.steps // 2)
    t_compute = None
    if args.exchange in ("maps", "xyz"):
        for rep in range(2):                                   # first pass warms up
            ctx.synchronize()
            ctx.comm_barrier()
            t0 = time.perf_counter()
            for i in range(K):
                scanner.compute_only(stacks[i % len(stacks)].ptr, plane)
            ctx.synchronize()
            t_compute = time.perf_counter() - t0
        t_compute = ctx.comm_allreduce_max(t_compute)
    # RCCL's own account of the job (not WORLD_SIZE): ncclCommCount, and the PCI bus id of every rank's GPU gathered over the communicator.  A
    # library that cannot answer (an RCCL without ncclCommCount, a failing hipDeviceGetPCIBusId) costs these keys, not the line.
    test_mode = os.environ.get("SLGC_RANKS_AS_HOSTS") == "1"      # several ranks share the one GPU of a test box on purpose
    comm, devices, distinct, info_error = {"nranks": None, "rank": None}, None, None, None
    try:
        comm = ctx.comm_info()
        devices, distinct = ctx.comm_rank_devices()
    except Exception as e:  # noqa: BLE001
        info_error = f"{type(e).__name__}: {e}"[:200]
    if comm["nranks"] is not None and comm["nranks"] != G:
        raise RuntimeError(f"RCCL reports {comm['nranks']} ranks, the launcher promised {G}")
    if distinct is not None and distinct != comm["nranks"] and not test_mode:
        raise RuntimeError(f"{comm['nranks']} RCCL ranks sit on {distinct} distinct GPUs ({devices}): one rank per GPU is the plan "
                           "(SLGC_RANKS_AS_HOSTS=1 is the one-GPU test mode)")
    px = cam_w * cam_h
    per_px = {"maps": 3 if scanner.wire == "hv24" else 4, "xyz": 16, "records": 16}[args.exchange]
    info = {"rccl_nranks": comm["nranks"] if comm["nranks"] is not None else G, "rccl_nranks_source": "ncclCommCount" if comm["nranks"] is not None else "WORLD_SIZE (unverified)",
            "rccl_rank": comm["rank"], "rank_devices": devices, "distinct_devices": distinct, "rccl_info_error": info_error,
            "ranks_share_gpus_test_mode": bool(test_mode and distinct is not None and distinct != comm["nranks"]), "exchange": args.exchange, "exchange_impl": args.exchange_impl, "wire": scanner.wire if args.exchange == "maps" else None,
            "overlap": not args.no_overlap and args.exchange != "records",
            "exchange_bytes_per_rank": {"sent": int(rows * cam_w * per_px), "received": int((px - rows * cam_w) * per_px)},
            "with_exchange_value": round(px / 1e6 * args.steps / elapsed, 1), "unit": "Mpixels/s"}
    if t_compute:
        info["compute_only_value"] = round(px / 1e6 * K / t_compute, 1)
        info["compute_only_note"] = "the same kernels per rank with the exchange left out (what the links would have to keep up with)"
    return info


def verify_sharded(ctx, scanner, G, rank, N, cam_w, cam_h, proj_size, seed, plane_pad, scene="s-scene"):
    """After the timed region: (1) every rank hashes the reassembled int16 maps and a strided XYZ sample it holds -> all-gather ->
    must be equal on all ranks; (2) every rank scans the SAME full image (same seed) alone on its own GPU with the fused kernel ->
    maps must be bit-identical, the XYZ sample equal to float32 resolution."""
    from scanner import _native
    px = cam_w * cam_h
    h, v, xyz = scanner.fetch_dense()
    sample = xyz.reshape(-1, 3)[::97]
    mine = digest64(h, v, np.nan_to_num(sample, nan=-1.0))
    ranks_equal = True
    allh = ctx.comm_allgather_i64(mine)                      # nranks = 1 (--force-sharded) included: the same calls as on a node
    ranks_equal = len(allh) == G and all(x == allh[0] for x in allh)
    full = ctx.alloc(N * px)
    synth_into(ctx, scene, full.ptr, px, N, cam_h, cam_w, proj_size, seed, row0=0, rows=cam_h)
    m1, x1 = ctx.alloc(px * 4), ctx.alloc(px * 12)
    ctx.scan_dev(full.ptr, 1, N * px, px, N, cam_h, cam_w, 0, proj_size, x1.ptr, None, m1.at(0), m1.at(px * 2), mode=_native.TRI_ALGEBRAIC)
    ctx.synchronize()
    h1, v1 = m1.download((cam_h, cam_w), np.int16), m1.download((cam_h, cam_w), np.int16, px * 2)
    s1 = x1.download((px, 3), np.float32)[::97]
    maps_equal = bool(np.array_equal(h, h1) and np.array_equal(v, v1))
    fin = np.isfinite(s1).all(axis=1)
    # bit for bit: the ray-table choice is taken for the whole image (slgc_tune "image_rows"), whatever band a rank scans
    xyz_equal = bool(np.array_equal(np.isfinite(sample).all(axis=1), fin) and np.array_equal(sample[fin].view(np.uint32), s1[fin].view(np.uint32)))
    for b in (full, m1, x1):
        b.free()
    ok_local = maps_equal and xyz_equal
    ok_all = ok_local
    oks = ctx.comm_allgather_i64(1 if ok_local else 0)
    ok_all = len(oks) == G and all(oks)
    return {"ok": bool(ranks_equal and ok_all), "ranks_hold_identical_results": bool(ranks_equal), "maps_equal_single_gpu_scan": maps_equal,
            "xyz_sample_equal_single_gpu_scan": xyz_equal, "valid_pixels": int(((h != -1) & (v != -1)).sum()), "digest": f"{mine:016x}",
            "note": "every rank compared its reassembled maps and a 1/97 XYZ sample, both bit for bit, with a single-GPU fused scan of the same "
                    "stack, and its digest with every other rank's"}


def direct_children(args, rank, G, device, key, cam, proj, N, wire, n_buffers, last_stack, main_digest, mode, steps):
    """Every rank starts tools/benchlib/direct_child.py on its own GPU and waits for it (bounded); -> the child's report, or {"error": ...}.
    Collective by construction: all ranks call this at the same point, the children meet each other through the direct exchange's segment."""
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "direct_child.py"), "--rank", str(rank), "--nranks", str(G),
           "--device", str(device), "--key", key, "--cam", f"{cam[0]}x{cam[1]}", "--proj", f"{proj[0]}x{proj[1]}", "--frames", str(N), "--scene", args.scene,
           "--wire", wire, "--buffers", str(n_buffers), "--plane-pad", str(args.plane_pad), "--steps", str(steps), "--last-stack", str(last_stack),
           "--main-digest", f"{main_digest:016x}", "--mode", str(int(mode))]
    limit = float(os.environ.get("SLGC_BENCH_DIRECT_CHILD_TIMEOUT_S", "60"))
    # the children's own host-side deadline ends before this parent's limit does: a child that cannot assemble its job reports why (and rank 0
    # unlinks the segment) instead of being killed mid-barrier with the name left behind
    env = dict(os.environ, SLGC_DIRECT_HOST_TIMEOUT_S=os.environ.get("SLGC_DIRECT_HOST_TIMEOUT_S", f"{max(5.0, 0.6 * limit):.0f}"))
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, text=True, env=env)
    except subprocess.TimeoutExpired:
        return {"error": f"child of rank {rank} did not finish within {limit:.0f} s"}
    except OSError as e:
        return {"error": f"child of rank {rank} could not start: {e}"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        tail = " | ".join(ln for ln in r.stderr.strip().splitlines()[-3:])
        return {"error": f"child of rank {rank}: exit code {r.returncode}; {tail[-300:]}"}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": f"child of rank {rank}: unreadable report ({e})"}
