#!/usr/bin/env python3
"""bench.py -- Mpixels/s of Gray-code decode + triangulation on MI355X (BASELINE.json metric).

One "step" = one scan: the 4096x3000 camera, 44-frame uint8 stack (BASELINE.json configs[2]; resident in HBM
before the timed region, generated on the device) goes through ONE fused kernel (decode with the triangulation tail) and
leaves int16 projector maps + a dense float32 XYZ map in HBM.  No torch anywhere: HIP through the ctypes C-ABI.
The capture of the timed region is the physically consistent scene on the covering rig (tools/benchlib/common.py: 96.5 % of the pixels decode);
this file is the entry point only -- the run itself lives in tools/benchlib/ (run.py, legs.py, sharded_legs.py, cpu.py, pmc.py).

  python bench.py [--gpus 1] --steps K --warmup W            single GPU
  python bench.py --gpus N ...                               N > 1: this process starts N fresh rank processes itself (one per
                                                             GPU, before anything touches a GPU) and relays rank 0's JSON line
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
                                                             the driver's launcher: RANK / LOCAL_RANK / WORLD_SIZE from the env
      N > 1 = the same scan row-sharded across the N GPUs (configs[3]): every step ends with the RCCL exchange that reassembles
      the whole cloud on every rank; after the timed region every rank hashes what it holds and the hashes are compared across
      ranks and with a single-GPU scan of the same stack (the run verifies itself: "verify" in the JSON).

Rank 0 prints ONE compact JSON line LAST on stdout (< 4 KB, fixed key set: tools/benchlib/line.py) -- the bench contract's keys with
  roofline          the dominant kernel of the timed region: algorithmic bytes (N+12 B/pixel fused, N+4 decode) /
                    launch duration from HIP events bound to the kernel's own dispatch inside the timed region, vs 8 TB/s; frac from the
                    MEDIAN launch, frac_mean and the count of launches above 2 x median beside it; traffic = HBM bytes per launch from
                    the run's OWN counters (two rocprofv3 --pmc children started before this process touches the GPU: --pmc auto),
                    else the committed constant of profiles/traffic.json (traffic_source says which)
  cpu_baseline      the reference-cost NumPy/Python port (oracle/oracle_np.py, kind "port") on a bounded sample of BASELINE configs[0],
                    1 thread, plus the plain-C oracle on 1 thread and on the best thread count (N = 1 only; timed before this process
                    touches the GPU, beside the counter children)
  s_scene_frac / decode_kernel_frac / throughput_mode_value    the fused kernel on SURVEY.md 8(d)'s S-scene, the decode kernel alone on it
                    (the north star's >= 60 % target), configs[4] through the batched launch
and writes the FULL report -- every leg, every roofline object, notes, where the run's seconds went -- to gpurun_out/bench_extras.json
(--extras-file).  --extras full adds the other scenes, two runs, split pipeline, XYZ only, ingest, small images, reference product,
sustained and the full-size CPU baselines to that file; the printed line keeps its shape.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from benchlib.common import PREHEAT_S, SCENES, WORKLOADS, calibration, csrc_fingerprint, synth_into  # noqa: E402,F401  (tests and tools use bench.WORKLOADS / bench.calibration)
from benchlib.legs import throughput_lanes, throughput_step  # noqa: E402,F401
from benchlib.run import run_rank  # noqa: E402


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
            env.setdefault("GPU_MAX_HW_QUEUES", "2")      # up to 8 processes share one device here: fewer hardware queues each (8 ranks: 7.8 -> 6.2 s per run, NOTES.md)
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
    ap.add_argument("--extras", default="lite", choices=["none", "lite", "full"],
                    help="lite (default): the timed region + what the printed line needs -- the decode kernel alone, the S-scene leg, the movement-only "
                         "yardstick, throughput mode, a bounded CPU baseline; full: every leg (all scenes, two runs, split pipeline, XYZ only, ingest, small "
                         "images, reference product, sustained, full-size CPU baselines) into the side file; none: the timed region only")
    ap.add_argument("--no-extras", action="store_true", help="the same as --extras none")
    ap.add_argument("--extras-file", default=None, help="where the full report goes (default gpurun_out/bench_extras.json under the repo root)")
    ap.add_argument("--exchange", default="maps", choices=["maps", "records", "xyz"],
                    help="multi-GPU reassembly: all-gather the int16 map bands and triangulate everywhere (default); all-gather maps + "
                         "float32 XYZ bands produced by the fused kernel on each band (xyz); or all-gatherv compacted 16-byte XYZ+key records")
    ap.add_argument("--exchange-impl", default="rccl", choices=["rccl", "direct"],
                    help="how the bands travel: rccl (default) = ncclAllGather / grouped ncclBroadcast; direct = every rank pushes its band straight into "
                         "every peer's buffer over IPC mappings, all xGMI links at once (csrc/direct.hip; EXPERIMENTAL until timed on a node -- the one not "
                         "chosen is timed after the counted region: sharded_alternatives)")
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
    ap.add_argument("--scene", default="physical", choices=sorted(SCENES),
                    help="synthetic capture of the timed region (benchlib/common.py SCENES; every one is also timed as a leg of the default run): physical = "
                         "plane + sphere seen by camera and projector through the covering rig, > 90 %% of the camera pixels lit by one consistent surface "
                         "(default); noisy-physical = the same, dim and noisy (wrong codes appear); physical-survey = the same surface through SURVEY.md "
                         "8(d)'s calibration (9-28 %% lit); s-scene / s-uniform = SURVEY.md 8(d)'s synthetic inputs on SURVEY.md 8(d)'s calibration")
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic from this run's own counters: before touching the GPU, start two rocprofv3 --pmc children (FETCH_SIZE, WRITE_SIZE) of "
                         "a 3-step headline run -- they run while this process times the CPU baseline -- and fold their bytes per launch into the line "
                         "(auto: single GPU, extras lite / full, rocprofv3 on PATH; otherwise / on failure the committed constant of profiles/traffic.json "
                         "is used, and the line says which)")
    ap.add_argument("--pmc-timeout", type=float, default=40.0, help="seconds one counter child may take")
    ap.add_argument("--no-small-images", action="store_true", help="skip the BASELINE configs[1] / configs[0] fused-kernel legs of the default run")
    ap.add_argument("--image-rows", type=int, default=0,
                    help="single-GPU band workloads: height of the whole image the band belongs to (slgc_tune image_rows; 0 = the band is the image)")
    ap.add_argument("--sustained", type=float, default=5.5,
                    help="seconds of back-to-back headline scans in the extra 'sustained' leg (0 = skip); longer than the 5 s cadence of the driver's SMI "
                         "sampler, so that something outside this script sees the GPU busy (round 4's 1 s leg never was)")
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
    if args.no_extras:
        args.extras = "none"
    args.pmc_live, args.pmc_note, args.cpu_baseline_result, args.pre_gpu_seconds = None, "not requested", None, {}
    single = world == 1 and args.gpus == 1 and not args.force_sharded
    # Before this process touches the GPU: the counter children (GPU, one after the other), THEN the CPU baseline (host cores) -- neither is inside
    # any timed region.  One after the other: each counter child is a Python process + the profiler + synthetic data generation on host cores,
    # and the C-oracle thread sweep of the CPU baseline uses every core -- side by side (round 5) they biased the CPU figures low.
    pmc_thread = None
    if args.pmc != "off" and single and (args.pmc == "on" or args.extras != "none"):
        import threading
        from benchlib import pmc
        cw, ch, _, _, _ = WORKLOADS[args.workload]

        def counters():
            t0 = time.perf_counter()
            live, note = pmc.collect(os.path.abspath(__file__), args.workload, args.scene, args.pipeline, grid_size=-(-(cw * ch // 4) // 128) * 128,
                                     timeout_s=args.pmc_timeout)
            args.pmc_note = note or "collected"
            if live:
                args.pmc_live = {f"{args.pipeline}/{args.scene}": live}
                if args.scene != "s-scene" and args.extras != "none":      # SURVEY 8(d)'s own capture rides inside the roofline object: its traffic too
                    live_s, _ = pmc.collect(os.path.abspath(__file__), args.workload, "s-scene", args.pipeline, grid_size=-(-(cw * ch // 4) // 128) * 128,
                                            timeout_s=args.pmc_timeout)
                    if live_s:
                        args.pmc_live[f"{args.pipeline}/s-scene"] = live_s
            args.pre_gpu_seconds["counter children"] = round(time.perf_counter() - t0, 3)

        pmc_thread = threading.Thread(target=counters, daemon=True)
        pmc_thread.start()
        pmc_thread.join()
        pmc_thread = None
    if single and args.extras != "none" and not args.no_cpu_baseline and os.environ.get("SLGC_BENCH_PMC_CHILD") != "1":
        from benchlib.cpu import cpu_baseline
        t0 = time.perf_counter()
        try:
            args.cpu_baseline_result = cpu_baseline(level=args.extras)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: CPU baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
        args.pre_gpu_seconds["cpu baseline"] = round(time.perf_counter() - t0, 3)
    if pmc_thread is not None:
        pmc_thread.join()
    try:
        run_rank(args, rank, local_rank, world)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        if rank == 0:       # the driver still gets a line: what failed, no number
            print(json.dumps({"metric": "Mpixels/s decode+triangulate", "value": None, "unit": "Mpixels/s", "n_gpus": args.gpus,
                              "steps": args.steps, "warmup": args.warmup, "error": f"{type(e).__name__}: {e}"}), flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()

