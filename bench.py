#!/usr/bin/env python3
"""bench.py -- Mpixels/s of Gray-code decode + triangulation on MI355X (BASELINE.json metric).

One "step" = one scan: the 4096x3000 camera, 44-frame uint8 stack (BASELINE.json configs[2]; resident in HBM
before the timed region, generated on the device) goes through the decode kernel and the triangulation kernel
and leaves a dense float32 XYZ map + int16 projector maps in HBM.  No torch: HIP through the ctypes C-ABI.

  python bench.py --gpus 1 --steps K --warmup W            single GPU
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
      one rank per GPU: the same scan row-sharded across the N GPUs (configs[3]); each step ends with the RCCL
      all-gatherv that reassembles the compacted point cloud (float32 XYZ + uint32 pixel key) on every rank.

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
    from scanner import _native
    path = f"/tmp/slgc_uid_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    if rank == 0:
        uid = _native.Context.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid, path
    t0 = time.time()
    while time.time() - t0 < 300:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == _native.UNIQUE_ID_BYTES:
                return uid, path
        except FileNotFoundError:
            pass
        time.sleep(0.05)
    raise RuntimeError("timed out waiting for the RCCL unique id from rank 0")


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
    return {"value": round(mpix / dt, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"{crop_w}x{crop_h}x{N} crop of the same synthetic scene, decode+triangulate, {dt:.1f} s, "
                      f"{pts.shape[1]} points; NumPy/Python port of the reference path (oracle/oracle_np.py), 1 thread; "
                      f"host has {os.cpu_count()} cores",
            "c_oracle_value": round(mpix / dt_c, 3), "c_oracle_note": "plain-C scalar oracle, 1 thread, same crop"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3_4096x3000x44", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="algebraic", choices=["algebraic", "exact"])
    ap.add_argument("--tri", default="lut", choices=["lut", "direct"], help="ray tables (default) or per-pixel undistortPoints")
    ap.add_argument("--variant", type=int, default=0, help="decode kernel variant (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--buffers", type=int, default=2, help="distinct input stacks rotated between steps")
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

    ctx = _native.Context(local_rank)
    calib = calibration(cam_w, cam_h, proj_w, proj_h)
    ctx.set_calibration(*calib)
    uid_path = None
    if G > 1:
        uid, uid_path = rendezvous_uid(rank, G)
        ctx.comm_init(rank, G, uid)
        ctx.comm_barrier()
        if rank == 0:
            try:
                os.remove(uid_path)
            except OSError:
                pass

    band_px = rows * cam_w
    plane = band_px                         # each rank holds only its row band of every frame
    stacks = []
    for b in range(max(1, args.buffers)):
        s = ctx.alloc(N * plane)
        ctx.synth_scene_dev(s.ptr, plane, N, cam_h, cam_w, row0=row0, rows=rows, seed=1 + b, noise=3, shadow=True)
        stacks.append(s)
    maps = ctx.alloc(band_px * 4)
    xyz = ctx.alloc(band_px * 12)
    count = ctx.alloc(8).zero()
    pts = keys = recv_pts = recv_keys = None
    if G > 1:
        pts, keys = ctx.alloc(band_px * 12), ctx.alloc(band_px * 4)
        recv_pts, recv_keys = ctx.alloc(cam_w * cam_h * 12), ctx.alloc(cam_w * cam_h * 4)
    ctx.synchronize()

    def step(i, counted=False):
        s = stacks[i % len(stacks)]
        ctx.scan_dev(s.ptr, 1, N * plane, plane, N, rows, cam_w, row0, (proj_w, proj_h), xyz.ptr, count.ptr if counted else None,
                     maps.at(0), maps.at(band_px * 2), mode=mode)
        if G > 1:
            ctx.compact_dev(xyz.ptr, rows, cam_w, row0, pts.ptr, keys.ptr, count.ptr)
            m = int(count.download((1,), np.uint64)[0])                 # host needs M_r for the displacements
            allm = ctx.comm_allgather_i64(m)
            displ = np.concatenate([[0], np.cumsum(allm)[:-1]])
            ctx.comm_allgatherv(pts.ptr, recv_pts.ptr, [12 * x for x in allm], [12 * int(d) for d in displ])
            ctx.comm_allgatherv(keys.ptr, recv_keys.ptr, [4 * x for x in allm], [4 * int(d) for d in displ])
            return sum(allm)
        return None

    for i in range(args.warmup):
        step(i)
    ctx.synchronize()
    if G > 1:
        ctx.comm_barrier()
    count.zero()
    ctx.synchronize()
    ctx.prof_begin(args.steps + 8)
    t0 = time.perf_counter()
    total_pts = None
    for i in range(args.steps):
        total_pts = step(i)
    ctx.synchronize()
    if G > 1:
        ctx.comm_barrier()
    elapsed = time.perf_counter() - t0
    dec_ms, dec_n = ctx.prof_end()
    if G > 1:
        elapsed = ctx.comm_allreduce_max(elapsed)
        dec_ms = ctx.comm_allreduce_max(dec_ms)

    if G == 1:
        count.zero()
        step(0, counted=True)                                   # untimed: valid-pixel count of one scan, for the report
        ctx.synchronize()
        valid = int(count.download((1,), np.uint64)[0])
    else:
        valid = float(total_pts)
    if rank == 0:
        mpix_per_step = cam_w * cam_h / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        value = mpix_per_step * args.steps / elapsed
        dec_avg_ms = dec_ms / max(1, dec_n)
        algo_bytes = (N + 4) * band_px                      # N bytes in + 2x int16 out per pixel
        achieved = algo_bytes / (dec_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "Mpixels/s decode+triangulate", "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": G,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{cam_w}x{cam_h} cam, {proj_w}x{proj_h} proj, {N} uint8 frames (BASELINE.json configs[2]"
                                   + ("" if G == 1 else f", row-sharded over {G} GPUs + RCCL all-gatherv = configs[3]") + ")",
                       "rows_per_gpu": rows, "triangulation": args.mode + "/" + args.tri, "input_buffers_rotated": len(stacks),
                       "outputs": "int16 h/v maps + dense float32 XYZ in HBM"
                                  + ("" if G == 1 else "; compacted XYZ+key all-gathered to every rank")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "k_decode_fast", "avg_launch_ms": round(dec_avg_ms, 5), "launches_timed": dec_n,
                         "algorithmic_bytes_per_launch": algo_bytes},
            "valid_pixels_per_scan": valid,
            "device": ctx.device_name(),
        }
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath)).get(f"{args.workload}/g{G}")
                if t:
                    out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = t.get("source")
            except Exception:
                pass
        if G == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, 1536, 768, calib, (proj_w, proj_h))
        print(json.dumps(out), flush=True)
    if G > 1:
        ctx.comm_barrier()
    ctx.close()


if __name__ == "__main__":
    main()
