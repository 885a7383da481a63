#!/usr/bin/env python3
"""bench.py -- Mpixels/s of Gray-code decode + triangulation on MI355X (BASELINE.json metric).

One "step" = one scan: the 4096x3000 camera, 44-frame uint8 stack (BASELINE.json configs[2]; resident in HBM
before the timed region, generated on the device) goes through ONE fused kernel (decode with the triangulation tail) and
leaves int16 projector maps + a dense float32 XYZ map in HBM.  No torch anywhere: HIP through the ctypes C-ABI.

  python bench.py [--gpus 1] --steps K --warmup W            single GPU
  python bench.py --gpus N ...                               N > 1: this process starts N fresh rank processes itself (one per
                                                             GPU, before anything touches a GPU) and relays rank 0's JSON line
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
                                                             the driver's launcher: RANK / LOCAL_RANK / WORLD_SIZE from the env
      N > 1 = the same scan row-sharded across the N GPUs (configs[3]): every step ends with the RCCL exchange that reassembles
      the whole cloud on every rank; after the timed region every rank hashes what it holds and the hashes are compared across
      ranks and with a single-GPU scan of the same stack (the run verifies itself: "verify" in the JSON).

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline          the dominant kernel of the timed region: algorithmic bytes (N+12 B/pixel fused, N+4 decode) /
                    launch duration from HIP events bound to the kernel's own dispatch inside the timed region, vs 8 TB/s
  cpu_baseline      the reference-cost NumPy/Python port (oracle/oracle_np.py, kind "port") on BASELINE configs[0] at full size,
                    1 thread, plus the plain-C oracle on 1 and on all host cores (N = 1 only)
  split_pipeline / xyz_only / decode_kernel_alone / throughput_mode / reference_product    N = 1 extras, same run
"""
import argparse
import hashlib
import json
import os
import subprocess
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
    # small test workloads (tests/test_gpu_rccl_multi.py): an odd height (ragged bands at any G) and an even one
    "t_516x1031x44": (516, 1031, 300, 200, 44),
    "t_512x1024x44": (512, 1024, 300, 200, 44),
    # one rank's band of the headline image at 2 / 4 / 8 ranks (timing the band kernels on one GPU: tools/ab_fused.py --workload ...)
    "b2_4096x1500x44": (4096, 1500, 1920, 1200, 44),
    "b4_4096x750x44": (4096, 750, 1920, 1200, 44),
    "b8_4096x375x44": (4096, 375, 1920, 1200, 44),
}
BASELINE_CONFIG = {"c1_1280x720x42": 0, "c2_1920x1080x44": 1, "c3_4096x3000x44": 2}      # --workload -> index into BASELINE.json "configs"


def workload_label(name, cam_w, cam_h, proj_w, proj_h, N, sharded_over=0):
    idx = BASELINE_CONFIG.get(name)
    if idx is None:
        tag = "not a BASELINE.json config: a variant / test / band workload"
    elif sharded_over and idx == 2:
        tag = f"BASELINE.json configs[2] row-sharded over {sharded_over} GPU(s) + RCCL exchange = configs[3]"
    elif sharded_over:
        tag = f"BASELINE.json configs[{idx}] row-sharded over {sharded_over} GPU(s) + RCCL exchange"
    else:
        tag = f"BASELINE.json configs[{idx}]"
    return f"{cam_w}x{cam_h} cam, {proj_w}x{proj_h} proj, {N} uint8 frames ({tag})"


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PREHEAT_S = 0.15       # untimed back-to-back scans before the counted warm-up: the clocks of a fresh box ramp for ~100 ms


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


# ------------------------------------------------------------------------------------------------ N > 1 launcher
def spawn_ranks(n, argv):
    """Start n rank processes of this script (fresh interpreters: nothing in THIS process has touched a GPU, and no process that
    has is ever exec'ed over), one per GPU, with the env torch.distributed.run would give them; relay their output; exit code = worst."""
    key = f"self_{os.getpid()}_{int(time.time() * 1e6)}"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), SLGC_UID_KEY=key, MASTER_ADDR="127.0.0.1",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if os.environ.get("SLGC_RANKS_AS_HOSTS") == "1":
            # TEST MODE for boxes with fewer GPUs than ranks: RCCL refuses two ranks on one device of one host ("Duplicate GPU detected"), so every
            # rank claims a host of its own (NCCL_HOSTID) and the ranks talk over the loopback socket transport.  Every RCCL call of the sharded
            # path then runs with nranks > 1 for real -- in-place ncclAllGather, grouped ncclBroadcast, the communication stream and its events --
            # at the speed of a TCP socket: a correctness mode, never a measurement.
            env.update(NCCL_HOSTID=f"slgc-rank-{r}-{key}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    deadline = time.time() + float(os.environ.get("SLGC_BENCH_TIMEOUT_S", "900")) + 15.0       # rank 0's own watchdog (same limit) prints its line first
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        failed = any(rc not in (None, 0) for rc in rcs)
        if failed or time.time() > deadline:             # one rank died (or the run hangs): the others would wait in a collective for ever
            t_kill = time.time() + (5.0 if failed else 0.0)
            while time.time() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()                            # exact pids this process started
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    sys.exit(max((abs(rc) for rc in rcs), default=0) and 1)


# ------------------------------------------------------------------------------------------------ CPU baseline (N = 1)
def cpu_baseline():
    """the reference-cost CPU path on BASELINE configs[0] (1280x720 camera, 1280x800 projector, 42 frames) at FULL
    size: get_codes with the reference's cost shape (fancy-index copies, np.repeat, ten np.where scatters), the per-pixel Python
    loops of src/3-capture_decode.py:99-100 and triangulate.py:52-64, then the NumPy law of sines -- one thread, like the reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    import oracle_np as onp
    cw, ch, pw, ph, n = WORKLOADS["c1_1280x720x42"]
    K, cd, pk, pd, R, T = calibration(cw, ch, pw, ph)
    st, _, _ = onp.synth_scene_int(n, ch, cw, seed=1)
    white = np.repeat(st[1][:, :, None], 3, axis=2)
    t0 = time.perf_counter()
    hc, vc = onp.get_codes_loops(st.astype(np.float64))                # float64 stack like the reference driver (src/3:68-70)
    t_codes = time.perf_counter() - t0
    hp, vp = onp.codes_to_pixels_loops(hc, vc)
    t_pix = time.perf_counter() - t0 - t_codes
    cam, proj, _ = onp.cam_proj_pts_loops(hp, vp, (cw, ch), (pw, ph), white)
    pts = onp.triangulate(cam, proj, K, cd, pk, pd, R, T)
    dt = time.perf_counter() - t0
    mpix = cw * ch / 1e6
    # strong baseline: the plain-C oracle on the headline workload's own size class (a 2048x1024 crop of the 44-frame scene)
    st2, _, _ = onp.synth_scene_int(44, 1024, 2048, seed=1)
    cal3 = calibration(4096, 3000, 1920, 1200)
    t1 = time.perf_counter()
    oc.scan_dense(st2, (1920, 1200), *cal3)
    dt_c = time.perf_counter() - t1
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    dt_mt, used = None, 1
    for nthr in sorted({min(cores, c) for c in (8, 16, 32, 64, 128, cores)}):     # a container's CPU quota can be far below its visible cores
        oc.set_threads(nthr)
        oc.scan_dense(st2, (1920, 1200), *cal3)                        # thread pool start-up
        t2 = time.perf_counter()
        oc.scan_dense(st2, (1920, 1200), *cal3)
        d = time.perf_counter() - t2
        if dt_mt is None or d < dt_mt:
            dt_mt, used = d, nthr
    oc.set_threads(1)
    mpix2 = 2048 * 1024 / 1e6
    return {"value": round(mpix / dt, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"BASELINE configs[0] at full size: {cw}x{ch} camera, {pw}x{ph} projector, {n} frames, synthetic scene, decode + "
                      f"triangulate, {dt:.1f} s ({t_codes:.1f} s get_codes, {t_pix:.1f} s gray_to_decimal loops), {pts.shape[1]} points; "
                      "NumPy/Python port with the reference's cost shape (oracle/oracle_np.py *_loops), 1 thread like the reference",
            "reference_measured": {"value": 0.046, "unit": "Mpixels/s", "note": "the reference itself, end to end at 1920x1080x44 in the "
                                   "survey container (BASELINE.md section 2); it cannot travel to the GPU box"},
            "c_oracle_value": round(mpix2 / dt_c, 3), "c_oracle_note": "plain-C scalar oracle (oracle/slgc_oracle.c), 1 thread, 2048x1024x44 crop of the headline scene",
            "c_oracle_all_cores_value": round(mpix2 / dt_mt, 3), "c_oracle_all_cores": used, "host_cores_visible": cores,
            "c_oracle_all_cores_note": "the same C oracle, per-pixel loops on host threads (OpenMP), best of 8/16/32/64/128/all visible cores, same crop"}


# ------------------------------------------------------------------------------------------------ configs[4]
def synth_into(c, scene, d_ptr, plane, n, H, W, proj_size, seed, row0=0, rows=None):
    """One synthetic capture into HBM: the physically consistent scene (needs the context's calibration) or SURVEY 8(d)'s S-scene."""
    if scene == "physical":
        c.synth_physical_dev(d_ptr, plane, n, H, W, proj_size, row0=row0, rows=rows, seed=seed, noise=3)
    else:
        c.synth_scene_dev(d_ptr, plane, n, H, W, row0=row0, rows=rows, seed=seed, noise=3, shadow=True)


def throughput_lanes(_native, device, per_rank, n_streams=2, scene="s-scene"):
    """BASELINE.json configs[4] on one GPU: `n_streams` contexts (HIP streams), each with its own rotated 1920x1080x44 stacks
    (>= 4 distinct stacks in total: 364 MB > Infinity Cache) and one set of output buffers.  -> [(ctx, stacks, maps, xyz)]"""
    cw, ch, pw, ph, n = WORKLOADS["c2_1920x1080x44"]
    px = cw * ch
    lanes = []
    for sidx in range(max(1, n_streams)):
        c = _native.Context(device)
        c.set_calibration(*calibration(cw, ch, pw, ph))
        stacks = []
        for b in range(max(2, -(-max(per_rank, 4) // max(1, n_streams)))):
            st = c.alloc(n * px)
            synth_into(c, scene, st.ptr, px, n, ch, cw, (pw, ph), 11 + 7 * sidx + b)
            stacks.append(st)
        lanes.append((c, stacks, c.alloc(px * 4), c.alloc(px * 12)))
    return lanes


def throughput_step(lanes, per_rank, i, mode):
    """Issue one step = `per_rank` independent scans, round-robin over the lanes.  -> [(lane index, stack index)] in issue order."""
    cw, ch, pw, ph, n = WORKLOADS["c2_1920x1080x44"]
    px = cw * ch
    plan = []
    for j in range(per_rank):
        li = j % len(lanes)
        c, stacks, maps, xyz = lanes[li]
        si = (i * per_rank + j) // len(lanes) % len(stacks)
        c.scan_dev(stacks[si].ptr, 1, n * px, px, n, ch, cw, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
        plan.append((li, si))
    return plan


def throughput_batched(ctx, _native, G, steps, mode, device, collective=False, scene="s-scene"):
    """configs[4] through slgc_scan_batch_dev: the GPU's share of the 16 scans in ONE launch per step (3 rotated sets of stacks: > Infinity
    Cache), no collective.  Returns (seconds, scans per step over all ranks, Mpixels per scan)."""
    cw, ch, pw, ph, n = WORKLOADS["c2_1920x1080x44"]
    px = cw * ch
    per_rank = max(1, 16 // G)
    c = _native.Context(device)
    c.set_calibration(*calibration(cw, ch, pw, ph))
    sets = []
    for b in range(3):
        st = c.alloc(per_rank * n * px)
        for s in range(per_rank):
            synth_into(c, scene, st.at(s * n * px), px, n, ch, cw, (pw, ph), 11 + 16 * b + s)
        sets.append(st)
    maps_h, maps_v, xyz = c.alloc(per_rank * px * 2), c.alloc(per_rank * px * 2), c.alloc(per_rank * px * 12)

    def step(i):
        c.scan_batch_dev(sets[i % 3].ptr, per_rank, n * px, px, n, ch, cw, 0, (pw, ph), xyz.ptr, maps_h.ptr, maps_v.ptr, mode=mode)

    for i in range(3):
        step(i)
    c.synchronize()
    if G > 1 or collective:                             # --force-sharded at one rank walks through every collective call too
        ctx.comm_barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    c.synchronize()
    if G > 1 or collective:                             # --force-sharded at one rank walks through every collective call too
        ctx.comm_barrier()
    el = time.perf_counter() - t0
    if G > 1 or collective:                             # --force-sharded at one rank walks through every collective call too
        el = ctx.comm_allreduce_max(el)
    c.close()
    return el, per_rank * G, cw * ch / 1e6


def throughput_mode(ctx, _native, G, steps, mode, device, n_streams=2, collective=False, scene="s-scene"):
    """16 independent 1920x1080x44 scans per step spread over the G GPUs, no collective (replicas only -- SURVEY.md 8(e)).  Each GPU
    streams its scans back to back over `n_streams` HIP streams so the tail of one scan's kernel overlaps the head of the next.
    Returns (seconds, scans per step over all ranks, Mpixels per scan)."""
    cw, ch = WORKLOADS["c2_1920x1080x44"][:2]
    per_rank = max(1, 16 // G)
    lanes = throughput_lanes(_native, device, per_rank, n_streams, scene)

    def sync_all():
        for c, _, _, _ in lanes:
            c.synchronize()

    for i in range(3):
        throughput_step(lanes, per_rank, i, mode)
    sync_all()
    if G > 1 or collective:                             # --force-sharded at one rank walks through every collective call too
        ctx.comm_barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        throughput_step(lanes, per_rank, i, mode)
    sync_all()
    if G > 1 or collective:                             # --force-sharded at one rank walks through every collective call too
        ctx.comm_barrier()
    el = time.perf_counter() - t0
    if G > 1 or collective:                             # --force-sharded at one rank walks through every collective call too
        el = ctx.comm_allreduce_max(el)
    for c, _, _, _ in lanes:
        c.close()
    return el, per_rank * G, cw * ch / 1e6


# ------------------------------------------------------------------------------------------------ helpers
def launch_stats(samples_ms):
    s = np.sort(np.asarray(samples_ms, dtype=np.float64))
    if s.size == 0:
        return {}
    q = lambda f: float(s[min(s.size - 1, int(round(f * (s.size - 1))))])     # noqa: E731
    return {"min_launch_ms": round(float(s[0]), 5), "median_launch_ms": round(q(0.5), 5), "p95_launch_ms": round(q(0.95), 5),
            "max_launch_ms": round(float(s[-1]), 5)}


def digest64(*arrays):
    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        h.update(np.ascontiguousarray(a).view(np.uint8).reshape(-1).data)
    return int.from_bytes(h.digest(), "little") >> 1          # 63 bits: travels through the int64 all-gather unchanged


SCAN_SOURCES = ("api.hip", "decode.hip", "slgc_internal.h", "tri_math.h", "triangulate.hip")   # what the scan kernels and their launch defaults compile from


def csrc_fingerprint():
    """Hash of the scan kernels' sources: profiles/traffic.json carries the fingerprint it was measured on, a mismatch = stale counters."""
    h = hashlib.blake2b(digest_size=8)
    d = os.path.join(ROOT, "3dscanner-graycode_amd", "csrc")
    for name in SCAN_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


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
    ap.add_argument("--event-stride", type=int, default=0,
                    help="bracket every n-th kernel launch of the timed region with a HIP event pair (0 = auto: 1 below 16 steps, 2 below 64, "
                         "else 4; a pair costs ~1.5 %% of a step)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams (contexts) per GPU in the throughput-mode measurement")
    ap.add_argument("--no-throughput-mode", action="store_true", help="skip the configs[4] (16 independent scans) extra measurement")
    ap.add_argument("--no-extras", action="store_true", help="only the headline timed region (no split / alone / throughput / reference-product / CPU legs)")
    ap.add_argument("--exchange", default="maps", choices=["maps", "records", "xyz"],
                    help="multi-GPU reassembly: all-gather the int16 map bands and triangulate everywhere (default); all-gather maps + "
                         "float32 XYZ bands produced by the fused kernel on each band (xyz); or all-gatherv compacted 16-byte XYZ+key records")
    ap.add_argument("--wire", default="auto", choices=["auto", "int16", "hv24"],
                    help="sharded 'maps' exchange: the int16 maps as they are (4 B/pixel) or packed to 3 B/pixel (codes of <= 11 bits); auto (default) = "
                         "packed when there is more than one rank and the codes fit -- a sharded scan is bound by its exchange (SURVEY.md 8(e)), a quarter "
                         "fewer bytes on the links for two small streaming kernels; the other wire is timed after the counted region (sharded_alternatives)")
    ap.add_argument("--no-overlap", action="store_true", help="sharded modes: do not pipeline the exchange with the neighbouring scans")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded path (RCCL exchange at nranks = 1) even on 1 GPU")
    ap.add_argument("--no-verify", action="store_true", help="sharded modes: skip the post-run cross-rank / single-GPU verification")
    ap.add_argument("--plane-pad", type=int, default=0,
                    help="extra bytes between frame planes in HBM (multiple of 16; 0 = contiguous [N,H,W] like the reference)")
    ap.add_argument("--preheat", type=float, default=PREHEAT_S,
                    help="seconds of untimed back-to-back scans before the counted warm-up (clock ramp of a fresh box); 0 under a counter profiler")
    ap.add_argument("--scene", default="physical", choices=["s-scene", "physical"],
                    help="synthetic capture of the timed region: the physically consistent plane + sphere scene (slgc_synth_physical_dev: one surface "
                         "seen by camera and projector through the benchmark calibration, 9-28 %% of the camera pixels lit; default) or SURVEY.md 8(d)'s "
                         "S-scene (arbitrary smooth code maps, ~80 %% decodable but epipolar-inconsistent).  The kernels do the same work per pixel "
                         "whether it decodes or not; the other scene is timed as an extra leg of the same run, and the decode-only legs always use the S-scene")
    ap.add_argument("--image-rows", type=int, default=0,
                    help="single-GPU band workloads: height of the whole image the band belongs to (slgc_tune image_rows; 0 = the band is the image)")
    ap.add_argument("--sustained", type=float, default=1.0,
                    help="seconds of back-to-back headline scans in the extra 'sustained' leg (0 = skip); long enough for SMI samplers to see the GPU busy")
    ap.add_argument("--buffers", type=int, default=0,
                    help="distinct input stacks rotated between steps (0 = as many as needed to exceed the 256 MB Infinity Cache, >= 2)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])              # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    try:
        run_rank(args, rank, local_rank, world)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        if rank == 0:       # the driver still gets a line: what failed, no number
            print(json.dumps({"metric": "Mpixels/s decode+triangulate", "value": None, "unit": "Mpixels/s", "n_gpus": args.gpus,
                              "steps": args.steps, "warmup": args.warmup, "error": f"{type(e).__name__}: {e}"}), flush=True)
        sys.exit(1)


STAGE = ["start"]          # where the run is (printed by the watchdogs: a hang names its stage)


def run_rank(args, rank, local_rank, world):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's host driver only supports dmabuf IPC: RCCL's P2P set-up fails without it
    if os.environ.get("SLGC_RANKS_AS_HOSTS") == "1" and world > 1 and "NCCL_HOSTID" not in os.environ:
        # the same TEST MODE under an external launcher (torch.distributed.run gives every rank the same environment): see spawn_ranks
        os.environ.update(NCCL_HOSTID=f"slgc-rank-{rank}-{os.environ.get('MASTER_PORT', '0')}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket")
    from scanner import _native
    cam_w, cam_h, proj_w, proj_h, N = WORKLOADS[args.workload]
    G = args.gpus
    use_comm = G > 1 or args.force_sharded
    mode = (_native.TRI_ALGEBRAIC if args.mode == "algebraic" else _native.TRI_EXACT) | (2 if args.tri == "direct" else 0)
    mode_split, mode_fused = mode | _native.TRI_SPLIT, mode & ~_native.TRI_SPLIT
    mode = mode_split if args.pipeline == "split" else mode_fused
    if args.event_stride <= 0:
        args.event_stride = 1 if args.steps < 16 else 2 if args.steps < 64 else 4

    if G > 1:
        # a rank that waits for ever in a collective (a peer died, the fabric is unhappy) must not leave the driver without a line
        import threading

        def give_up():
            if rank == 0:
                print(json.dumps({"metric": "Mpixels/s decode+triangulate", "value": None, "unit": "Mpixels/s", "n_gpus": G, "steps": args.steps,
                                  "warmup": args.warmup, "error": f"timed out after SLGC_BENCH_TIMEOUT_S in the multi-rank run (stage: {STAGE[0]})"}), flush=True)
            os._exit(4)

        killer = threading.Timer(float(os.environ.get("SLGC_BENCH_TIMEOUT_S", "900")), give_up)
        killer.daemon = True
        killer.start()

    STAGE[0] = "context"
    n_dev = max(1, _native.device_count())
    device = local_rank % n_dev                 # a launcher that narrows device visibility per rank leaves only device 0 visible
    ctx = _native.Context(device)
    calib = calibration(cam_w, cam_h, proj_w, proj_h)
    ctx.set_calibration(*calib)
    if G == 1 and args.force_sharded:
        ctx.comm_init(0, 1, _native.Context.comm_unique_id())
    if G > 1:
        from scanner import sharded
        STAGE[0] = "rccl unique id"
        uid, uid_path = sharded.share_unique_id(rank, _native.Context.comm_unique_id, key=os.environ.get("SLGC_UID_KEY"))
        STAGE[0] = "ncclCommInitRank"
        ctx.comm_init(rank, G, uid)
        STAGE[0] = "first barrier"
        ctx.comm_barrier()
        STAGE[0] = "setup"
        if rank == 0:
            try:
                os.remove(uid_path)
            except OSError:
                pass
    if use_comm:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL prints its version banner through C stdio: flush it now so rank 0's JSON stays the last line

    from scanner import sharded
    plan = sharded.ShardPlan(cam_h, cam_w, G)
    row0, rows = plan.band(rank)
    band_px = rows * cam_w
    plane = band_px + args.plane_pad        # each rank holds only its row band of every frame; pad 0 = the reference's contiguous [N,H,W]
    if args.buffers <= 0:                                   # enough distinct stacks to exceed the 256 MB Infinity Cache
        args.buffers = max(2, -(-300_000_000 // max(1, N * plane)))
    def make_stacks(scene):
        out_ = []
        for b in range(max(1, args.buffers)):
            s = ctx.alloc(max(16, N * plane))
            if rows and scene == "physical":
                ctx.synth_physical_dev(s.ptr, plane, N, cam_h, cam_w, (proj_w, proj_h), row0=row0, rows=rows, seed=1 + b, noise=3)
            elif rows:
                ctx.synth_scene_dev(s.ptr, plane, N, cam_h, cam_w, row0=row0, rows=rows, seed=1 + b, noise=3, shadow=True)
            out_.append(s)
        return out_

    stacks = make_stacks(args.scene)
    maps = ctx.alloc(max(16, band_px * 4))
    xyz = ctx.alloc(max(16, band_px * 12))
    count = ctx.alloc(16).zero()
    sharded_scanner = None
    if use_comm:                            # (sets slgc_tune "image_rows": the ray-table choice below is the whole image's)
        sharded_scanner = sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), plan, (proj_w, proj_h), N, mode=mode,
                                                 exchange_kind=args.exchange, wire=args.wire)
    elif args.image_rows > 0:               # band workloads (b2 / b4 / b8): time the band kernels as a rank of the sharded scan would run them
        ctx.tune("image_rows", args.image_rows)
    # per-calibration work, hoisted out of the scans and timed on its own: both undistortPoints calls evaluated into the two ray tables
    ctx.synchronize()
    if use_comm:
        ctx.tune("image_rows", cam_h)       # what the scanner sets around its own calls: the tables built here are the ones it will use
    ctx.event_record(0)
    ctx.build_ray_tables_dev(cam_h if use_comm and args.exchange == "maps" else rows, cam_w, 0 if use_comm and args.exchange == "maps" else row0,
                             (proj_w, proj_h))
    ctx.event_record(1)
    luts_us = ctx.event_elapsed_ms(0, 1) * 1e3
    if use_comm:
        ctx.tune("image_rows", 0)
    ctx.synchronize()
    pipelined = use_comm and args.exchange in ("maps", "xyz") and not args.no_overlap

    def step(i, counted=False, mode=mode, src=None, no_maps=False):
        src = stacks if src is None else src
        s = src[i % len(src)]
        if pipelined:
            return sharded_scanner.submit(s.ptr, plane)      # exchange of this scan overlaps the neighbours' kernels
        if use_comm:
            return sharded_scanner.scan(s.ptr, plane)
        ctx.scan_dev(s.ptr, 1, N * plane, plane, N, rows, cam_w, row0, (proj_w, proj_h), xyz.ptr, count.ptr if counted else None,
                     None if no_maps else maps.at(0), None if no_maps else maps.at(band_px * 2), mode=mode)
        return None

    def drain():
        if sharded_scanner is not None:
            sharded_scanner.flush()
        ctx.synchronize()

    def timed(K, W_, stride=None, preheat=True, **kw):
        if preheat and args.preheat > 0:                     # untimed: bring the clocks up before the counted warm-up
            t_end = time.perf_counter() + args.preheat
            i = 0
            more = True
            while more:
                for _ in range(16):
                    step(i, **kw)
                    i += 1
                drain()
                more = time.perf_counter() < t_end
                if use_comm:                                 # every rank must run the SAME number of scans (each one is a collective): the
                    more = ctx.comm_allreduce_max(1.0 if more else 0.0) > 0.5       # ranks agree on going on -- a clock per rank would not
        for i in range(W_):
            step(i, **kw)
        drain()
        if use_comm:
            ctx.comm_barrier()
        ctx.prof_begin(K + 8, stride or args.event_stride)   # HIP-event pair bound to every stride-th kernel dispatch of the region
        t0 = time.perf_counter()
        tot = None
        for i in range(K):
            tot = step(i, **kw)
        drain()                                              # the K-th scan's exchange + triangulation are inside the timed region
        if use_comm:
            ctx.comm_barrier()
        el = time.perf_counter() - t0
        kms, kn = ctx.prof_end()
        samples = ctx.prof_samples()
        if use_comm:
            el = ctx.comm_allreduce_max(el)
            kms = ctx.comm_allreduce_max(kms)
        return el, kms, kn, tot, samples

    STAGE[0] = "timed region (incl. pre-heat and warm-up)"
    elapsed, dec_ms, dec_n, total_pts, dec_samples = timed(args.steps, args.warmup)
    STAGE[0] = "extras"
    executed = ctx.last_scan_path()                          # what the library actually launched in the timed region (not what this script asked for)
    last_stack = (args.steps - 1) % len(stacks)              # what the output buffers hold now
    single = G == 1 and not use_comm
    extras = single and not args.no_extras and args.mode == "algebraic" and args.tri == "lut"

    def scene_stats(src):
        """valid / guard-flagged pixels of one scan of src[0] (untimed)"""
        count.zero()
        ctx.scan_dev(src[0].ptr, 1, N * plane, plane, N, rows, cam_w, row0, (proj_w, proj_h), xyz.ptr, None, maps.at(0), maps.at(band_px * 2), mode=mode_fused)
        ctx.guard_count_dev(maps.at(0), maps.at(band_px * 2), rows, cam_w, row0, (proj_w, proj_h), count.ptr)
        ctx.synchronize()
        return tuple(int(x) for x in count.download((2,), np.uint64))

    other_scene = None
    if extras:
        # the same kernel on the other synthetic capture: S-scene = arbitrary smooth code maps (epipolar-inconsistent: part of its pixels
        # takes the guarded float64 path), physical = one surface seen by camera and projector (few lit pixels with these calibrations)
        o_name = "physical" if args.scene == "s-scene" else "s-scene"
        o_stacks = make_stacks(o_name)
        s_scene_stacks = o_stacks if o_name == "s-scene" else stacks
        o_el, o_kms, o_kn, _, o_samples = timed(args.steps, max(2, args.warmup // 2), preheat=False, mode=mode_fused, src=o_stacks)
        o_exec = ctx.last_scan_path()
        o_valid, o_flag = scene_stats(o_stacks)
        acc = None
        if o_name == "physical" or args.scene == "physical":
            acc = physical_accuracy(ctx, N, cam_h, cam_w, row0, rows, plane, (proj_w, proj_h), calib, maps, xyz, band_px, mode_fused)
        other_scene = (o_name, o_el, o_kms, o_kn, o_samples, o_exec, o_valid, o_flag, acc)

    sustained = None
    if single and not args.no_extras and args.sustained > 0:
        sustained = sustained_leg(ctx, step, drain, args.sustained, cam_w * rows / 1e6)

    other = None
    if extras:
        om = mode_fused if args.pipeline == "split" else mode_split
        other = timed(args.steps, max(2, args.warmup // 2), preheat=False, mode=om)
        other_executed = ctx.last_scan_path()

    xyz_only = None
    if extras:
        # the same scan for a caller that wants the cloud only (no map buffers passed): the fused kernel then moves exactly SURVEY 8(d)'s N + 12 B/pixel
        xyz_only = timed(args.steps, max(2, args.warmup // 2), preheat=False, mode=mode_fused, no_maps=True)
        xyz_only_executed = ctx.last_scan_path()

    dec_alone = None
    if extras:
        # the decode kernel by itself, back to back over the rotated stacks (no other kernel's write-back in its way)
        for i in range(3):
            ctx.decode_dev(s_scene_stacks[i % len(stacks)].ptr, 1, N * plane, plane, N, rows, cam_w, maps.at(0), maps.at(band_px * 2), variant=args.variant)
        ctx.synchronize()
        ctx.prof_begin(args.steps + 8)
        for i in range(args.steps):
            ctx.decode_dev(s_scene_stacks[i % len(stacks)].ptr, 1, N * plane, plane, N, rows, cam_w, maps.at(0), maps.at(band_px * 2), variant=args.variant)
        dec_alone = ctx.prof_end() + (ctx.prof_samples(),)
        dec_alone_exec = ctx.last_scan_path()

    movement = None
    if extras and N in (42, 44, 46) and band_px % 256 == 0 and plane % 4 == 0:
        # the yardstick: a kernel that ONLY moves the bytes of this scan (slgc_move_only_dev) -- N planes in; maps + 12 B/px, 12 B/px alone, or the
        # maps alone out -- launched back to back over the same rotated stacks, timed with HIP events around the batch
        def move(K, **out_ptrs):
            for i in range(3):
                ctx.move_only_dev(stacks[i % len(stacks)].ptr, plane, N, band_px, **out_ptrs)
            ctx.synchronize()
            ctx.event_record(2)
            for i in range(K):
                ctx.move_only_dev(stacks[i % len(stacks)].ptr, plane, N, band_px, **out_ptrs)
            ctx.event_record(3)
            ctx.synchronize()
            return ctx.event_elapsed_ms(2, 3) / K
        K_mv = max(10, args.steps)
        movement = {"fused_with_maps_ms": move(K_mv, d_h=maps.at(0), d_v=maps.at(band_px * 2), d_xyz=xyz.ptr),
                    "fused_xyz_only_ms": move(K_mv, d_xyz=xyz.ptr),
                    "decode_ms": move(K_mv, d_h=maps.at(0), d_v=maps.at(band_px * 2))}

    ref_product = None
    if extras and row0 == 0:
        ref_product = reference_product(ctx, _native, s_scene_stacks, N, plane, rows, cam_w, row0, (proj_w, proj_h), maps, xyz, band_px, args.steps, mode_fused)
        ref_product["scene"] = "s-scene"

    thr = thr_batched = None
    if not args.no_throughput_mode and not args.no_extras and args.mode == "algebraic" and args.tri == "lut":
        thr = throughput_mode(ctx, _native, G, max(5, args.steps // 4), mode_fused, device, args.streams, collective=use_comm, scene=args.scene)
        thr_batched = throughput_batched(ctx, _native, G, max(5, args.steps // 4), mode_fused, device, collective=use_comm, scene=args.scene)

    # ---- what one scan holds: valid pixels, pixels on the guarded triangulation path (untimed)
    count.zero()
    if not use_comm:
        ctx.scan_dev(stacks[0].ptr, 1, N * plane, plane, N, rows, cam_w, row0, (proj_w, proj_h), xyz.ptr, None, maps.at(0), maps.at(band_px * 2), mode=mode)
        ctx.guard_count_dev(maps.at(0), maps.at(band_px * 2), rows, cam_w, row0, (proj_w, proj_h), count.ptr)
    elif args.exchange == "records":
        pass
    else:
        ctx.guard_count_dev(sharded_scanner.h_full.ptr, sharded_scanner.v_full.ptr, cam_h, cam_w, 0, (proj_w, proj_h), count.ptr)
    ctx.synchronize()
    valid, flagged = (int(x) for x in count.download((2,), np.uint64))
    if use_comm and args.exchange == "records":
        valid, flagged = int(total_pts), None

    verify = None
    shard_info = None
    if use_comm:
        STAGE[0] = "sharded report (compute-only timing)"
        shard_info = sharded_report(ctx, sharded_scanner, args, G, rank, stacks, plane, N, rows, cam_w, cam_h, row0, (proj_w, proj_h), mode, elapsed)
        if not args.no_verify and args.exchange in ("maps", "xyz"):
            STAGE[0] = "verification"
            verify = verify_sharded(ctx, sharded_scanner, G, rank, N, cam_w, cam_h, (proj_w, proj_h), 1 + last_stack, args.plane_pad, args.scene)

    out = None
    if rank == 0:
        mpix_per_step = cam_w * cam_h / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        value = mpix_per_step * args.steps / elapsed
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            traffic_db = json.load(open(tpath))
        except Exception:  # noqa: BLE001
            traffic_db = {}
        fp = csrc_fingerprint()

        def kernel_name(ex, pipeline):
            spec = f"NS={ex['ns_frames']} (threshold frames parked in LDS)" if ex["ns_frames"] else "generic frame count"
            if pipeline == "split":
                return f"k_decode_pk<4,128,nt> {spec}"
            return f"k_decode_pk<4,128,nt,FUSE=2> {spec} + triangulation tail (camera rays: {'node table' if ex['node_table'] else 'per-pixel table'})"

        def kernel_roofline(pipeline, kms, kn, samples, ex=None, scene=None):
            """SURVEY.md 8(d) byte definitions: decode kernel N + 4 B/pixel (N uint8 reads, 2 int16 writes); fused decode -> XYZ
            N + 12 B/pixel.  The fused kernel also writes the 4 B/pixel maps (a product): frac_incl_maps counts them too."""
            per_px = (N + 4) if pipeline == "split" else (N + 12)
            avg_ms = kms / max(1, kn)
            ach = per_px * band_px / (avg_ms * 1e-3) / 1e9
            r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                 "traffic": None, "kernel": kernel_name(ex or executed, pipeline),
                 "avg_launch_ms": round(avg_ms, 5), "launches_timed": kn, **launch_stats(samples),
                 "algorithmic_bytes_per_px": per_px, "algorithmic_bytes_per_launch": per_px * band_px}
            if pipeline != "split":
                r["frac_incl_maps"] = round((N + 16) * band_px / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                r["frac_incl_maps_note"] = "N + 16 B/pixel: the 4 B/pixel int16 maps the fused kernel also writes counted as algorithmic"
            t = traffic_db.get(f"{args.workload}/g{G}/{pipeline}")
            if t and t.get("csrc_fingerprint") == fp:
                # the counters were collected per scene for the fused kernel (tools/pmc.sh): fewer lit pixels = fewer projector-table lines gathered
                sc = scene or args.scene
                r["traffic"] = t.get("s_scene_hbm_bytes_per_launch") if (sc == "s-scene" and pipeline != "split" and "s_scene_hbm_bytes_per_launch" in t) else t["hbm_bytes_per_launch"]
                r["traffic_source"] = t.get("source")
                r["traffic_scene"] = "s-scene" if (pipeline == "split" or sc == "s-scene") else "physical"
            elif t:
                r["traffic_note"] = ("profiles/traffic.json was measured on other kernel sources (fingerprint mismatch): stale, not reported; "
                                     "re-run tools/pmc.sh")
            return r

        main_pipeline = "split" if (use_comm and args.exchange == "maps") else args.pipeline
        if not use_comm:                                     # the label follows the launch, not the request
            main_pipeline = "fused" if executed["path"] in ("fused", "batch-fused") else "split"
        out = {
            "metric": "Mpixels/s decode+triangulate", "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": G,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_label(args.workload, cam_w, cam_h, proj_w, proj_h, N, G if use_comm else 0),
                       "pipeline": ({"maps": "decode kernel per band, map bands all-gathered, full-image triangulation kernel on every rank",
                                     "xyz": "fused kernel per band, map + XYZ bands all-gathered in place",
                                     "records": "fused kernel per band, compaction, 16-byte XYZ+key records all-gatherv'ed"}[args.exchange] if use_comm
                                    else {"fused": "fused (one kernel)", "batch-fused": "fused (one kernel, batched)", "split": "split (decode kernel + triangulation kernel)",
                                          "split-ragged": "split (decode kernel + triangulation kernel, with byte-wide / per-pixel fallback kernels)"}.get(executed["path"], executed["path"])),
                       "executed": {**executed, "source": "slgc_last_scan_path after the timed region", "requested_pipeline": args.pipeline},
                       "scene": {"s-scene": "S-scene (SURVEY.md 8(d): smooth synthetic code maps, shadow rectangle, noise 3)",
                                 "physical": "physical (plane + sphere seen by camera and projector through the calibration, noise 3)"}[args.scene],
                       "rows_per_gpu": rows, "triangulation": args.mode + "/" + args.tri, "input_buffers_rotated": len(stacks), "plane_pad_bytes": args.plane_pad,
                       "outputs": "int16 h/v maps + dense float32 XYZ in HBM" + ("" if not use_comm else "; whole cloud reassembled on every rank"),
                       "preheat_s": args.preheat, "event_stride": args.event_stride,
                       "luts_hoisted_us": round(luts_us, 1),
                       "luts_hoisted_note": "per-calibration ray tables (both cv2.undistortPoints calls on integer pixel coordinates) built once "
                                            "before the timed region, not per scan",
                       "camera_rays": (lambda use, err: {"node_table_in_use": use, "node_table_error_vs_limit_2.4e-7": err,
                                                         "note": "per-pixel table 8 B/pixel, or (bands above 64 MB of rays) the every-4th-column "
                                                                 "table 2 B/pixel + a cubic through 4 nodes per 4-pixel group; flat triangles and "
                                                                 "zero-crossing rays always read the exact per-pixel table"})(*ctx.ray_table_info()),
                       "guard_flagged_pixels": flagged,
                       "guard_note": "decodable pixels of one scan that triangulation redoes on the reference's float32 intermediates (flat triangles)"},
            "roofline": kernel_roofline(main_pipeline, dec_ms, dec_n, dec_samples),
            "valid_pixels_per_scan": valid,
            "device": ctx.device_name(),
        }
        if shard_info:
            out["sharded"] = shard_info
        if verify is not None:
            out["verify"] = verify
        if other is not None:
            o_el, o_kms, o_kn, _, o_samples = other
            o_name = "fused" if args.pipeline == "split" else "split"
            out[o_name + "_pipeline"] = {"value": round(mpix_per_step * args.steps / o_el, 1), "unit": "Mpixels/s",
                                         "ms_per_step": round(o_el / args.steps * 1e3, 4), "steps": args.steps,
                                         "executed": other_executed,
                                         "roofline": kernel_roofline("fused" if other_executed["path"] == "fused" else "split", o_kms, o_kn, o_samples, other_executed),
                                         "note": "same scan, same run, timed right after the main region"}
        if movement is not None:
            def beside(roof, key):
                mv = movement[key]
                roof["movement_only"] = {"avg_launch_ms": round(mv, 5), "kernel_over_movement": round(roof["avg_launch_ms"] / mv, 3),
                                         "frac": round(roof["algorithmic_bytes_per_launch"] / (mv * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "slgc_move_only_dev in the same run: a kernel that only moves this scan's bytes (N planes read 4 B per lane "
                                                 "and plane; the same outputs in the same store shapes); frac = what IT reaches on the same algorithmic bytes"}
            if main_pipeline == "fused":
                beside(out["roofline"], "fused_with_maps_ms")
        if xyz_only is not None:
            x_el, x_kms, x_kn, _, x_samples = xyz_only
            xr = kernel_roofline("fused" if xyz_only_executed["path"] == "fused" else "split", x_kms, x_kn, x_samples, xyz_only_executed)
            xr.pop("frac_incl_maps", None), xr.pop("frac_incl_maps_note", None)
            xr["traffic"] = None                              # the committed counters are of the kernel that also stores the maps
            xr.pop("traffic_source", None), xr.pop("traffic_scene", None)
            if movement is not None and xyz_only_executed["path"] == "fused":
                beside(xr, "fused_xyz_only_ms")
            out["xyz_only"] = {"value": round(mpix_per_step * args.steps / x_el, 1), "unit": "Mpixels/s", "ms_per_step": round(x_el / args.steps * 1e3, 4),
                               "steps": args.steps, "executed": xyz_only_executed, "roofline": xr,
                               "note": "the headline scan with d_h = d_v = NULL (cloud wanted, maps not): same kernel, the two int16 map stores "
                                       "skipped; XYZ bit-identical (tests/test_gpu_fullsize.py).  NOT the headline: the reference's decode script "
                                       "keeps the maps, so `value` is measured with them stored"}
        if dec_alone is not None:
            dr = kernel_roofline("split", *dec_alone, ex=dec_alone_exec)
            if movement is not None:
                beside(dr, "decode_ms")
            out["decode_kernel_alone"] = {"roofline": dr, "scene": "s-scene",
                                          "note": "decode kernel launched back to back on rotated S-scene stacks (~80 % of the pixels decodable: the decode "
                                                  "kernel's heavier input), same run (the north star's >= 60 % of HBM roofline on the decode kernel at 4096x3000x44)"}
        if ref_product is not None:
            out["reference_product"] = ref_product
        if sustained is not None:
            out["sustained"] = sustained
        if other_scene is not None:
            o_name, o_el, o_kms, o_kn, o_samples, o_exec, o_valid, o_flag, acc = other_scene
            out["other_scene"] = {"scene": o_name, "value": round(mpix_per_step * args.steps / o_el, 1), "unit": "Mpixels/s",
                                  "ms_per_step": round(o_el / args.steps * 1e3, 4), "executed": o_exec,
                                  "roofline": kernel_roofline("fused" if o_exec["path"] == "fused" else "split", o_kms, o_kn, o_samples, o_exec, scene=o_name),
                                  "valid_pixels_per_scan": o_valid, "guard_flagged_pixels": o_flag,
                                  "note": "the headline step on the other synthetic capture, same run (bench.py --scene picks which one is the headline)"}
            if acc is not None:
                out["physical_scene_accuracy"] = acc
        if thr is not None:
            t_el, t_scans, t_mpix = thr
            t_steps = max(5, args.steps // 4)
            out["throughput_mode"] = {"value": round(t_scans * t_mpix * t_steps / t_el, 1), "unit": "Mpixels/s",
                                      "config": f"{t_scans} independent 1920x1080x44 scans per step ({t_scans // G} per GPU), no collective "
                                                "(BASELINE.json configs[4], replicas only)",
                                      "scans_per_s": round(t_scans * t_steps / t_el, 1), "steps": t_steps, "streams_per_gpu": args.streams,
                                      "scene": args.scene, "scaling": "weak"}
            if thr_batched is not None:
                b_el, b_scans, b_mpix = thr_batched
                out["throughput_mode"]["batched"] = {"value": round(b_scans * b_mpix * t_steps / b_el, 1), "unit": "Mpixels/s",
                                                     "scans_per_s": round(b_scans * t_steps / b_el, 1),
                                                     "note": "the same scans through slgc_scan_batch_dev: each GPU's share in one launch per step"}
        if single and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline()

    import threading
    emit_lock, emitted = threading.Lock(), []

    def emit(extra=None):
        with emit_lock:                                      # exactly one JSON line, whichever thread gets here first (main or the watchdog)
            if emitted:
                return
            emitted.append(True)
            if rank == 0:
                import ctypes
                ctypes.CDLL(None).fflush(None)
                line = dict(out)
                if extra:
                    line.update(extra)
                print(json.dumps(line), flush=True)

    # Extras of the multi-rank run, AFTER everything above is measured and assembled: a watchdog prints the line as it stands and ends
    # the process if they do not come back (a hang in a collective that has never run on more than one GPU must not cost the run).
    def bail():
        emit({"sharded_alternatives": {"error": f"timed out: the line was printed without them (stage: {STAGE[0]})"}})
        os._exit(3 if (verify is not None and not verify.get("ok", False)) else 5)      # 5 = the extras hung (the headline above is complete)

    watchdog = threading.Timer(float(os.environ.get("SLGC_BENCH_ALT_TIMEOUT_S", "120")), bail)
    watchdog.daemon = True
    watchdog.start()
    alternatives = None
    STAGE[0] = "sharded alternatives"
    if use_comm and args.exchange == "maps" and not args.no_extras and pipelined:
        # The first run on real xGMI is rare: time the other exchange forms too (same stacks, same pipelining, each verified against the maps
        # the main strategy left) -- extras after the counted region, a failure here is reported and changes nothing above.
        sharded_scanner.submit(stacks[last_stack].ptr, plane)
        sharded_scanner.flush()
        h_main, v_main, _ = sharded_scanner.fetch_dense()
        main_digest = digest64(h_main, v_main)
        alternatives = {}
        other_wire = "int16" if sharded_scanner.wire == "hv24" else "hv24"           # whichever wire the main strategy did not use
        for label, kind, wire in (("maps_" + other_wire, "maps", other_wire), ("xyz", "xyz", "int16")):
            if wire == "hv24" and int((N - 2) / 4) > _native.WIRE_MAX_CODE_BITS:
                continue
            try:
                alt = sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), plan, (proj_w, proj_h), N, mode=mode, exchange_kind=kind, wire=wire)
                K = max(5, args.steps // 2)
                for i in range(3):
                    alt.submit(stacks[i % len(stacks)].ptr, plane)
                alt.flush()
                ctx.synchronize()
                ctx.comm_barrier()
                t0 = time.perf_counter()
                for i in range(K):
                    alt.submit(stacks[i % len(stacks)].ptr, plane)
                alt.flush()
                ctx.synchronize()
                ctx.comm_barrier()
                el_alt = ctx.comm_allreduce_max(time.perf_counter() - t0)
                alt.submit(stacks[last_stack].ptr, plane)                 # the stack the main strategy finished on
                alt.flush()
                ha, va, _ = alt.fetch_dense()
                same = ctx.comm_allgather_i64(1 if digest64(ha, va) == main_digest else 0)
                alternatives[label] = {"value": round(cam_w * cam_h / 1e6 * K / el_alt, 1), "unit": "Mpixels/s", "steps": K,
                                       "bytes_per_pixel_on_the_links": {"maps_hv24": 3, "maps_int16": 4, "xyz": 16}[label],
                                       "maps_equal_main_strategy_on_every_rank": bool(all(same))}
                del alt
            except Exception as e:  # noqa: BLE001
                alternatives[label] = {"error": f"{type(e).__name__}: {e}"}

    STAGE[0] = "final barrier"
    watchdog.cancel()
    emit({"sharded_alternatives": alternatives} if (rank == 0 and alternatives) else None)
    if use_comm:
        ctx.comm_barrier()
    ctx.close()
    if verify is not None and not verify.get("ok", False):
        sys.exit(3)
    if single and args.pipeline == "fused" and args.mode == "algebraic" and args.tri == "lut" and executed["path"] != "fused":
        print(f"bench.py: the headline asked for the fused kernel but the library launched '{executed['path']}'", file=sys.stderr)
        sys.exit(6)


def physical_accuracy(ctx, N, H, W, row0, rows, plane, proj_size, calib, maps, xyz, band_px, mode_fused):
    """Recovered XYZ of one fused scan of the physical scene against the generator's TRUE surface points (not against the oracle): the error
    is the method's -- half a projector pixel of code quantisation seen through the triangulation geometry."""
    st, truth = ctx.alloc(max(16, N * plane)), ctx.alloc(max(16, band_px * 12))
    ctx.synth_physical_dev(st.ptr, plane, N, H, W, proj_size, row0=row0, rows=rows, seed=1, noise=3, d_truth_xyz=truth.ptr)
    ctx.scan_dev(st.ptr, 1, N * plane, plane, N, rows, W, row0, proj_size, xyz.ptr, None, maps.at(0), maps.at(band_px * 2), mode=mode_fused)
    ctx.synchronize()
    got = xyz.download((band_px, 3), np.float32)[::5]
    tru = truth.download((band_px, 3), np.float32)[::5]
    ok = np.isfinite(got).all(axis=1) & np.isfinite(tru).all(axis=1)
    err = np.linalg.norm(got[ok].astype(np.float64) - tru[ok], axis=1)
    rng = np.linalg.norm(tru[ok].astype(np.float64), axis=1)
    st.free()
    truth.free()
    if not err.size:
        return {"error": "no lit pixel decoded"}
    return {"pixels_compared": int(ok.sum()), "sampling": "every 5th pixel", "median_error_mm": round(float(np.median(err)) * 1e3, 4),
            "max_error_mm": round(float(err.max()) * 1e3, 4), "max_relative_error": float(f"{float((err / rng).max()):.3e}"),
            "range_m": [round(float(rng.min()), 3), round(float(rng.max()), 3)],
            "note": "|recovered - true surface point| of a fused scan of the physical scene; the truth comes from the generator's ray casting, "
                    "not from the CPU oracle; tests/test_gpu_physical.py bounds it per pixel by the code-quantisation geometry"}


class GpuSampler:
    """Shader clock and busy percentage of the GPU from sysfs (amdgpu: pp_dpm_sclk marks the active level with '*', gpu_busy_percent),
    sampled from a thread while a leg runs.  Whatever is not readable on this box stays None."""

    def __init__(self, pci=None, period=0.05):
        import glob
        import threading
        self.period, self.clk, self.busy = period, [], []
        cands = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
        mine = [d for d in cands if pci and os.path.basename(os.path.realpath(d)).lower() == pci]      # the HIP device's own node, by PCI address
        direct = os.path.join("/sys/bus/pci/devices", pci or "-")
        self.dev = mine[0] if mine else direct if os.path.exists(os.path.join(direct, "pp_dpm_sclk")) else (cands[0] if len(cands) == 1 else None)
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            for ln in open(os.path.join(self.dev, "pp_dpm_sclk")):
                if "*" in ln:
                    self.clk.append(float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
        except Exception:  # noqa: BLE001
            pass
        try:
            self.busy.append(float(open(os.path.join(self.dev, "gpu_busy_percent")).read()))
        except Exception:  # noqa: BLE001
            pass

    def _run(self):
        while not self._stop.wait(self.period):
            self._read()

    def __enter__(self):
        if self.dev:
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.dev:
            self._t.join(timeout=1.0)

    def report(self):
        return {"sclk_mhz_mean": round(float(np.mean(self.clk)), 1) if self.clk else None, "sclk_mhz_min": min(self.clk) if self.clk else None,
                "gpu_busy_percent_mean": round(float(np.mean(self.busy)), 1) if self.busy else None, "samples": max(len(self.clk), len(self.busy)),
                "source": (self.dev + "/{pp_dpm_sclk,gpu_busy_percent}") if self.dev else "no readable amdgpu sysfs node"}


def sustained_leg(ctx, step, drain, seconds, mpix_per_step):
    """The headline step launched back to back for `seconds` (>= 1 s: long enough for an SMI sampler -- the driver's or the one here -- to see
    the GPU busy), one host synchronisation every 64 scans."""
    n = 0
    try:
        pci = ctx.device_pci_bus_id()
    except Exception:  # noqa: BLE001
        pci = None
    with GpuSampler(pci) as smp:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(64):
                step(n)
                n += 1
            drain()
        el = time.perf_counter() - t0
    return {"value": round(mpix_per_step * n / el, 1), "unit": "Mpixels/s", "seconds": round(el, 3), "scans": n, "ms_per_scan": round(el / n * 1e3, 4),
            "gpu": smp.report(), "note": "same step as the headline, back to back for >= 1 s; clock / busy sampled from sysfs every 50 ms while it ran"}


def reference_product(ctx, _native, stacks, N, plane, rows, W, row0, proj_size, maps, xyz, band_px, steps, mode_fused):
    """The reference-shaped product, device resident: x-major cam_pts / proj_pts (float32 [M,2]), Pts float64 (3,M) and colors float64
    [M,3] gathered from a device-resident white image (triangulate.py:52-71, 84-95).  Timed end to end per scan, two ways:
    slgc_cloud_dev (decode kernel + list build that triangulates in-kernel: no dense XYZ) and, for comparison, round 2's
    slgc_scan_dev + slgc_cloud_lists_dev (fused scan writes dense XYZ, the list build reads it back)."""
    white = ctx.alloc(band_px * 3)
    ctx.dev_memset(white.ptr, 0x80, band_px * 3)
    lists = ctx.alloc_cloud_lists(band_px, colors=True)
    K = max(5, steps // 4)

    def via_cloud(i):
        s = stacks[i % len(stacks)]
        ctx.cloud_dev(s.ptr, 1, N * plane, plane, N, rows, W, proj_size, white.ptr, lists, d_h=maps.at(0), d_v=maps.at(band_px * 2))

    def via_dense(i):
        s = stacks[i % len(stacks)]
        ctx.scan_dev(s.ptr, 1, N * plane, plane, N, rows, W, row0, proj_size, xyz.ptr, None, maps.at(0), maps.at(band_px * 2), mode=mode_fused)
        ctx.cloud_lists_dev(maps.at(0), maps.at(band_px * 2), xyz.ptr, white.ptr, W, rows, proj_size, lists)

    def run(one):
        for i in range(3):
            one(i)
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            one(i)
        ctx.synchronize()
        return time.perf_counter() - t0

    prod = ctx.alloc_cloud_lists(band_px, colors=True, lists=False)

    def via_cloud_points(i):
        s = stacks[i % len(stacks)]
        ctx.cloud_dev(s.ptr, 1, N * plane, plane, N, rows, W, proj_size, white.ptr, prod, d_h=maps.at(0), d_v=maps.at(band_px * 2))

    el_dense = run(via_dense)
    el_points = run(via_cloud_points)
    el = run(via_cloud)
    executed = ctx.last_scan_path()
    list_kernel = ctx.last_list_kernel()
    M = lists.total()
    # the list stage alone: decode once, then K list builds back to back (maps stay in place)
    ctx.decode_dev(stacks[0].ptr, 1, N * plane, plane, N, rows, W, maps.at(0), maps.at(band_px * 2))
    ctx.event_record(2)
    for _ in range(K):
        ctx.cloud_lists_dev(maps.at(0), maps.at(band_px * 2), None, white.ptr, W, rows, proj_size, lists)      # d_xyz = None: triangulate in-kernel
    ctx.event_record(3)
    stage_ms = ctx.event_elapsed_ms(2, 3) / K
    # ALGORITHMIC bytes of the list stage: maps read twice (count + scatter) 8, camera rays 2 (node table) or 8 (per-pixel table), white 3 per
    # pixel in; 8 + 8 + 24 + 24 per valid pixel out.  (The whole-lines scatter reads every tile's maps / white bytes / nodes a second time as
    # the halo of the tile above: not counted here.)
    ray_b = 2 if executed["node_table"] else 8
    stage_bytes = band_px * (8 + ray_b + 3) + M * 64
    out = {"value": round(band_px / 1e6 * K / el, 1), "unit": "Mpixels/s", "ms_per_scan": round(el / K * 1e3, 4), "steps": K, "points": int(M),
           "executed": {**executed, "list_kernel": list_kernel},
           "list_stage_ms": round(stage_ms, 4), "list_stage_bytes": int(stage_bytes),
           "list_stage_roofline": {"bound": "hbm", "achieved": round(stage_bytes / (stage_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(stage_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
           "points_and_colours_only": {"value": round(band_px / 1e6 * K / el_points, 1), "ms_per_scan": round(el_points / K * 1e3, 4),
                                       "note": "slgc_cloud_dev without the two correspondence lists (intermediates of src/4-triangulate.py:62-64; the script keeps "
                                               "pts_3d and colors, :67-68): 48 instead of 64 bytes written per point"},
           "via_dense_xyz": {"value": round(band_px / 1e6 * K / el_dense, 1), "ms_per_scan": round(el_dense / K * 1e3, 4),
                             "note": "round 2's route: fused scan (writes 12 B/pixel of dense XYZ) + slgc_cloud_lists_dev (reads it back)"},
           "note": "slgc_cloud_dev: decode kernel + x-major list build (count, column prefix, LDS-transposed scatter that triangulates each valid "
                   "pixel in-kernel, folds in the colour gather and the float64 (3,M) points and writes whole aligned 128-byte lines); everything stays in HBM, no dense XYZ; "
                   "list_stage_ms = the list build alone, mean of %d back-to-back builds" % K}
    white.free()
    lists.free()
    prod.free()
    return out


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
    px = cam_w * cam_h
    per_px = {"maps": 3 if scanner.wire == "hv24" else 4, "xyz": 16, "records": 16}[args.exchange]
    info = {"rccl_nranks": G, "exchange": args.exchange, "wire": scanner.wire if args.exchange == "maps" else None,
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
    if scene == "physical":
        ctx.synth_physical_dev(full.ptr, px, N, cam_h, cam_w, proj_size, row0=0, rows=cam_h, seed=seed, noise=3)
    else:
        ctx.synth_scene_dev(full.ptr, px, N, cam_h, cam_w, row0=0, rows=cam_h, seed=seed, noise=3, shadow=True)
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


if __name__ == "__main__":
    main()
