#!/usr/bin/env python3
"""bench.py -- Mpixels/s of Gray-code decode + triangulation on MI355X (BASELINE.json metric).

One "step" = one scan: the 4096x3000 camera, 44-frame uint8 stack (BASELINE.json configs[2]; resident in HBM
before the timed region, generated on the device) goes through the decode kernel and the triangulation kernel
and leaves a dense float32 XYZ map + int16 projector maps in HBM.  No torch: HIP through the ctypes C-ABI.

  python bench.py --gpus 1 --steps K --warmup W            single GPU
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
      one rank per GPU: the same scan row-sharded across the N GPUs (configs[3]); each step ends with the RCCL
      all-gatherv that reassembles the compacted point cloud (float32 XYZ + uint32 pixel key) on every rank.

The timed region runs the library's default pipeline: ONE kernel per scan (the decode kernel with the triangulation tail,
`--pipeline fused`).  At N=1 the two-kernel pipeline (`--pipeline split`: the decode kernel as its own launch, then the dense
triangulation kernel) is timed right after over the same K steps and reported in the extra object "split_pipeline", so the
decode kernel's own roofline fraction (the north star's 60 % target) is measured in the same run.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      decode kernel: algorithmic bytes (N+4 per pixel) / mean launch duration from HIP events recorded on the
                launch stream inside the timed region, against the 8 TB/s HBM3E peak
  cpu_baseline  the reference-equivalent NumPy/Python port (oracle/oracle_np.py, kind "port") timed on this host on a
                bounded crop of the same workload (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))

WORKLOADS = {
    # name: (cam_w, cam_h, proj_w, proj_h, N)
    "c3_4096x3000x44": (4096, 3000, 1920, 1200, 44),
    "c2_1920x1080x44": (1920, 1080, 1920, 1080, 44),
    "c3_4096x3000x46": (4096, 3000, 1920, 1200, 46),
    "c1_1280x720x42": (1280, 720, 1280, 800, 42),
    "c2_1920x1080x46": (1920, 1080, 1920, 1080, 46),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def calibration(cam_w, cam_h, proj_w, proj_h):
    """repo intrinsics; fx=fy=3000 for the 4096x3000 camera; synthetic extrinsics."""
    from scanner import reference_calibration as rc
    K = rc.CAM_MTX.copy()
    if cam_w > 1920:
        K[0, 0] = K[1, 1] = 3000.0
        K[0, 2], K[1, 2] = cam_w / 2.0, cam_h / 2.0
    pk = rc.PROJ_MTX.copy()
    pk[0, :] *= proj_w / 1920.0          # Triangulate.__init__ scaling, triangulate.py:28-33
    pk[1, :] *= proj_h / 1080.0
    th = np.deg2rad(-20.0)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T = np.array([[0.25], [0.02], [0.04]])
    return K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T


def rendezvous_uid(rank, world):
    """Share the RCCL unique id between the ranks torch.distributed.run started (same parent pid)."""
    from scanner import _native, sharded
    return sharded.share_unique_id(rank, _native.Context.comm_unique_id)


def cpu_baseline(N, crop_w, crop_h, calib, proj_size):
    """Reference-equivalent CPU path on a crop of the same synthetic scene: vectorised NumPy get_codes, then the
    per-pixel Python loops of src/3-capture_decode.py:99-100 and triangulate.py:52-64, then NumPy triangulation."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_np as onp
    st, _, _ = onp.synth_scene_int(N, crop_h, crop_w, seed=1)
    white = np.repeat(st[1][:, :, None], 3, axis=2)
    K, cd, pk, pd, R, T = calib
    t0 = time.perf_counter()
    hc, vc = onp.get_codes(st.astype(np.float64))                      # float64 stack like the reference driver
    hp, vp = onp.codes_to_pixels_loops(hc, vc)
    cam, proj, _ = onp.cam_proj_pts_loops(hp, vp, (crop_w, crop_h), proj_size, white)
    pts = onp.triangulate(cam, proj, K, cd, pk, pd, R, T)
    dt = time.perf_counter() - t0
    mpix = crop_w * crop_h / 1e6
    import oracle_c as oc
    t1 = time.perf_counter()
    oc.scan_dense(st, proj_size, K, cd, pk, pd, R, T)
    dt_c = time.perf_counter() - t1
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    dt_mt, used = None, 1
    for n in sorted({min(cores, c) for c in (8, 16, 32, 64, 128, cores)}):     # a container's CPU quota can be far below its visible cores
        oc.set_threads(n)
        oc.scan_dense(st, proj_size, K, cd, pk, pd, R, T)              # thread pool start-up
        t2 = time.perf_counter()
        oc.scan_dense(st, proj_size, K, cd, pk, pd, R, T)
        d = time.perf_counter() - t2
        if dt_mt is None or d < dt_mt:
            dt_mt, used = d, n
    oc.set_threads(1)
    return {"value": round(mpix / dt, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"{crop_w}x{crop_h}x{N} crop of the same synthetic scene, decode+triangulate, {dt:.1f} s, "
                      f"{pts.shape[1]} points; NumPy/Python port of the reference path (oracle/oracle_np.py), 1 thread; "
                      f"host has {os.cpu_count()} cores",
            "c_oracle_value": round(mpix / dt_c, 3), "c_oracle_note": "plain-C scalar oracle, 1 thread, same crop",
            "c_oracle_all_cores_value": round(mpix / dt_mt, 3), "c_oracle_all_cores": used, "host_cores_visible": cores,
            "c_oracle_all_cores_note": "the same C oracle with its per-pixel loops on host threads (OpenMP), best of 8/16/32/64/128/all visible cores, same crop"}


def throughput_mode(ctx, _native, G, steps, mode, device, n_streams=2):
    """BASELINE.json configs[4]: 16 independent 1920x1080x44 scans per step, spread over the G GPUs, no collective
    (replicas only -- SURVEY.md 8(e)).  Each GPU streams its scans back to back over `n_streams` contexts (HIP streams) so the
    tail of one scan's kernel overlaps the head of the next.  Returns (seconds, scans per step over all ranks, Mpixels per scan)."""
    cw, ch, pw, ph, n = WORKLOADS["c2_1920x1080x44"]
    per_rank = max(1, 16 // G)
    px = cw * ch
    lanes = []
    for sidx in range(max(1, n_streams)):
        c = _native.Context(device)
        c.set_calibration(*calibration(cw, ch, pw, ph))
        stacks = []
        for b in range(max(2, -(-max(per_rank, 4) // max(1, n_streams)))):   # >= 4 distinct stacks in total (364 MB > Infinity Cache)
            st = c.alloc(n * px)
            c.synth_scene_dev(st.ptr, px, n, ch, cw, seed=11 + 7 * sidx + b)
            stacks.append(st)
        lanes.append((c, stacks, c.alloc(px * 4), c.alloc(px * 12)))

    def one_step(i):
        for j in range(per_rank):
            c, stacks, maps, xyz = lanes[j % len(lanes)]
            st = stacks[(i * per_rank + j) // len(lanes) % len(stacks)]
            c.scan_dev(st.ptr, 1, n * px, px, n, ch, cw, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)

    def sync_all():
        for c, _, _, _ in lanes:
            c.synchronize()

    for i in range(3):
        one_step(i)
    sync_all()
    if G > 1:
        ctx.comm_barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(i)
    sync_all()
    if G > 1:
        ctx.comm_barrier()
    el = time.perf_counter() - t0
    if G > 1:
        el = ctx.comm_allreduce_max(el)
    for c, _, _, _ in lanes:
        c.close()
    return el, per_rank * G, px / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3_4096x3000x44", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="algebraic", choices=["algebraic", "exact"])
    ap.add_argument("--pipeline", default="fused", choices=["split", "fused"],
                    help="fused: decode kernel with the triangulation tail, one launch per scan (default, the library's own choice); "
                         "split: decode kernel + triangulation kernel")
    ap.add_argument("--tri", default="lut", choices=["lut", "direct"], help="ray tables (default) or per-pixel undistortPoints")
    ap.add_argument("--variant", type=int, default=0, help="decode kernel variant (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="bracket every n-th kernel launch of the timed region with a HIP event pair (1 = all; each pair costs ~1.5 %% of a step)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams (contexts) per GPU in the throughput-mode measurement")
    ap.add_argument("--no-throughput-mode", action="store_true", help="skip the configs[4] (16 independent scans) extra measurement")
    ap.add_argument("--exchange", default="maps", choices=["maps", "records"],
                    help="multi-GPU reassembly: all-gather the int16 map bands and triangulate everywhere (default), or all-gatherv "
                         "compacted 16-byte XYZ+key records")
    ap.add_argument("--wire", default="auto", choices=["auto", "int16", "hv24"],
                    help="sharded 'maps' exchange: send the int16 maps (4 B/pixel) or the packed 3 B/pixel wire format (auto: hv24 for G > 1)")
    ap.add_argument("--no-overlap", action="store_true", help="sharded 'maps' mode: do not pipeline the exchange with the neighbouring scans")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded path (compaction + RCCL exchange) even on 1 GPU")
    ap.add_argument("--plane-pad", type=int, default=0,
                    help="extra bytes between frame planes in HBM (multiple of 16; 0 = contiguous [N,H,W] like the reference)")
    ap.add_argument("--buffers", type=int, default=0,
                    help="distinct input stacks rotated between steps (0 = as many as needed to exceed the 256 MB Infinity Cache, >= 2)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    from scanner import _native
    cam_w, cam_h, proj_w, proj_h, N = WORKLOADS[args.workload]
    G = args.gpus
    if cam_h % G:
        sys.exit(f"image height {cam_h} not divisible by {G} GPUs")
    rows = cam_h // G
    row0 = rank * rows
    mode = (_native.TRI_ALGEBRAIC if args.mode == "algebraic" else _native.TRI_EXACT) | (2 if args.tri == "direct" else 0)
    mode_split, mode_fused = mode | _native.TRI_SPLIT, mode & ~_native.TRI_SPLIT
    mode = mode_split if args.pipeline == "split" else mode_fused

    n_dev = max(1, _native.device_count())
    device = local_rank % n_dev                 # a launcher that narrows device visibility per rank leaves only device 0 visible
    ctx = _native.Context(device)
    calib = calibration(cam_w, cam_h, proj_w, proj_h)
    ctx.set_calibration(*calib)
    uid_path = None
    use_comm = G > 1 or args.force_sharded
    if G == 1 and args.force_sharded:
        ctx.comm_init(0, 1, _native.Context.comm_unique_id())
    if G > 1:
        uid, uid_path = rendezvous_uid(rank, G)
        ctx.comm_init(rank, G, uid)
        ctx.comm_barrier()
        if rank == 0:
            try:
                os.remove(uid_path)
            except OSError:
                pass

    if use_comm:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL prints its version banner through C stdio: flush it now so rank 0's JSON stays the last line
    band_px = rows * cam_w
    plane = band_px + args.plane_pad        # each rank holds only its row band of every frame; pad 0 = the reference's contiguous [N,H,W]
    stacks = []
    if args.buffers <= 0:                                   # enough distinct stacks to exceed the 256 MB Infinity Cache
        args.buffers = max(2, -(-300_000_000 // (N * plane)))
    for b in range(max(1, args.buffers)):
        s = ctx.alloc(N * plane)
        ctx.synth_scene_dev(s.ptr, plane, N, cam_h, cam_w, row0=row0, rows=rows, seed=1 + b, noise=3, shadow=True)
        stacks.append(s)
    maps = ctx.alloc(band_px * 4)
    xyz = ctx.alloc(band_px * 12)
    count = ctx.alloc(8).zero()
    sharded_scanner = None
    if use_comm:
        from scanner import sharded
        sharded_scanner = sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), sharded.ShardPlan(cam_h, cam_w, G),
                                                 (proj_w, proj_h), N, mode=mode, exchange_kind=args.exchange, wire=args.wire)
    ctx.synchronize()

    def step(i, counted=False, mode=mode):
        s = stacks[i % len(stacks)]
        if use_comm and args.exchange == "maps" and not args.no_overlap:
            return sharded_scanner.submit(s.ptr, plane)      # pipelined: exchange of this scan overlaps the neighbours' kernels
        if use_comm:
            return sharded_scanner.scan(s.ptr, plane)        # band scan + compaction + counts + RCCL all-gatherv
        ctx.scan_dev(s.ptr, 1, N * plane, plane, N, rows, cam_w, row0, (proj_w, proj_h), xyz.ptr, count.ptr if counted else None,
                     maps.at(0), maps.at(band_px * 2), mode=mode)
        return None

    def timed(K, W_, **kw):
        for i in range(W_):
            step(i, **kw)
        if sharded_scanner is not None:
            sharded_scanner.flush()
        ctx.synchronize()
        if G > 1:
            ctx.comm_barrier()
        ctx.prof_begin(K + 8, args.event_stride)     # HIP-event pair around every event_stride-th kernel launch of the region
        t0 = time.perf_counter()
        tot = None
        for i in range(K):
            tot = step(i, **kw)
        if sharded_scanner is not None:
            sharded_scanner.flush()                          # the K-th scan's exchange + triangulation are inside the timed region
        ctx.synchronize()
        if G > 1:
            ctx.comm_barrier()
        el = time.perf_counter() - t0
        kms, kn = ctx.prof_end()
        if G > 1:
            el = ctx.comm_allreduce_max(el)
            kms = ctx.comm_allreduce_max(kms)
        return el, kms, kn, tot

    elapsed, dec_ms, dec_n, total_pts = timed(args.steps, args.warmup)
    other = None
    if G == 1 and not use_comm and args.mode == "algebraic" and args.tri == "lut":
        om = mode_fused if args.pipeline == "split" else mode_split
        other = timed(args.steps, max(2, args.warmup // 2), mode=om)

    dec_alone = None
    if G == 1 and not use_comm:
        # the decode kernel by itself, back to back over the rotated stacks (no other kernel's write-back in its way)
        for i in range(3):
            ctx.decode_dev(stacks[i % len(stacks)].ptr, 1, N * plane, plane, N, rows, cam_w, maps.at(0), maps.at(band_px * 2), variant=args.variant)
        ctx.synchronize()
        ctx.prof_begin(args.steps + 8)
        for i in range(args.steps):
            ctx.decode_dev(stacks[i % len(stacks)].ptr, 1, N * plane, plane, N, rows, cam_w, maps.at(0), maps.at(band_px * 2), variant=args.variant)
        dec_alone = ctx.prof_end()
    thr = None
    if not args.no_throughput_mode and args.mode == "algebraic" and args.tri == "lut":
        thr = throughput_mode(ctx, _native, G, max(5, args.steps // 4), mode_fused, device, args.streams)
    if not use_comm:
        count.zero()
        step(0, counted=True)                                   # untimed: valid-pixel count of one scan, for the report
        ctx.synchronize()
        valid = int(count.download((1,), np.uint64)[0])
    elif args.exchange == "records":
        valid = float(total_pts)
    else:                                                       # maps exchange: count the reassembled dense cloud once, untimed
        count.zero()
        ctx.triangulate_maps_dev(sharded_scanner.h_full.ptr, sharded_scanner.v_full.ptr, cam_h, cam_w, 0, (proj_w, proj_h),
                                 sharded_scanner.xyz_full.ptr, count.ptr, mode=mode & 3)
        ctx.synchronize()
        valid = int(count.download((1,), np.uint64)[0])
    if rank == 0:
        mpix_per_step = cam_w * cam_h / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        value = mpix_per_step * args.steps / elapsed
        def kernel_roofline(pipeline, kms, kn):
            """decode kernel: N bytes in + 2x int16 out per pixel; fused kernel: + 12 B float32 XYZ out."""
            per_px = (N + 4) if pipeline == "split" else (N + 4 + 12)
            avg_ms = kms / max(1, kn)
            ach = per_px * band_px / (avg_ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": None, "kernel": "k_decode_pk<4,128,nt>" + ("" if pipeline == "split" else " + triangulation tail (fused)"),
                    "avg_launch_ms": round(avg_ms, 5), "launches_timed": kn, "algorithmic_bytes_per_px": per_px,
                    "algorithmic_bytes_per_launch": per_px * band_px}

        out = {
            "metric": "Mpixels/s decode+triangulate", "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": G,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{cam_w}x{cam_h} cam, {proj_w}x{proj_h} proj, {N} uint8 frames (BASELINE.json configs[2]"
                                   + ("" if not use_comm else f", row-sharded over {G} GPUs + RCCL all-gatherv = configs[3]") + ")",
                       **({} if not use_comm else {"exchange": ((("map bands packed to 3 B/pixel, all-gathered, unpacked" if sharded_scanner.wire == "hv24"
                                                                  else "int16 map bands all-gathered") + ", every rank triangulates the full maps")
                                                                + ("" if args.no_overlap else "; exchange of scan i overlaps triangulation of i-1 and decode of i+1")
                                                                if args.exchange == "maps" else
                                                                "compacted 16-byte XYZ+key records all-gathered")}),
                       "pipeline": ("decode kernel per band + full-image triangulation kernel per rank" if (use_comm and args.exchange == "maps")
                                    else args.pipeline + (" (decode kernel + triangulation kernel)" if args.pipeline == "split" else " (one kernel)")),
                       "rows_per_gpu": rows, "triangulation": args.mode + "/" + args.tri, "input_buffers_rotated": len(stacks), "plane_pad_bytes": args.plane_pad,
                       "outputs": "int16 h/v maps + dense float32 XYZ in HBM"
                                  + ("" if not use_comm else "; whole cloud reassembled on every rank")},
            # the kernel bracketed by the event pairs: the fused scan kernel, or (split / sharded "maps" strategy) the decode kernel
            "roofline": kernel_roofline("split" if (use_comm and args.exchange == "maps") else args.pipeline, dec_ms, dec_n),
            "valid_pixels_per_scan": valid,
            "device": ctx.device_name(),
        }
        def add_traffic(roof, pipeline):
            """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), same workload only."""
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            try:
                t = json.load(open(tpath)).get(f"{args.workload}/g{G}/{pipeline}")
                if t:
                    roof["traffic"] = t["hbm_bytes_per_launch"]
                    roof["traffic_source"] = t.get("source")
            except Exception:
                pass

        add_traffic(out["roofline"], args.pipeline)
        if other is not None:
            o_el, o_kms, o_kn, _ = other
            o_name = "fused" if args.pipeline == "split" else "split"
            out[o_name + "_pipeline"] = {"value": round(mpix_per_step * args.steps / o_el, 1), "unit": "Mpixels/s",
                                         "ms_per_step": round(o_el / args.steps * 1e3, 4), "steps": args.steps,
                                         "roofline": kernel_roofline(o_name, o_kms, o_kn),
                                         "note": "same scan, same run, timed right after the main region"}
            add_traffic(out[o_name + "_pipeline"]["roofline"], o_name)
        if dec_alone is not None:
            out["decode_kernel_alone"] = {"roofline": kernel_roofline("split", dec_alone[0], dec_alone[1]),
                                          "note": "decode kernel launched back to back on the rotated stacks, same run (the north star's "
                                                  ">= 60 % of HBM roofline on the decode kernel at 4096x3000x44)"}
            add_traffic(out["decode_kernel_alone"]["roofline"], "split")
        if thr is not None:
            t_el, t_scans, t_mpix = thr
            t_steps = max(5, args.steps // 4)
            out["throughput_mode"] = {"value": round(t_scans * t_mpix * t_steps / t_el, 1), "unit": "Mpixels/s",
                                      "config": f"{t_scans} independent 1920x1080x44 scans per step ({t_scans // G} per GPU), no collective "
                                                "(BASELINE.json configs[4], replicas only)",
                                      "scans_per_s": round(t_scans * t_steps / t_el, 1), "steps": t_steps, "streams_per_gpu": args.streams,
                                      "scaling": "weak"}
        if G == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, 2048, 1024, calib, (proj_w, proj_h))
        if use_comm:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if G > 1:
        ctx.comm_barrier()
    ctx.close()


if __name__ == "__main__":
    main()
