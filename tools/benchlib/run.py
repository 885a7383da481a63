"""bench.py: what one rank does (set-up, timed region, extras, the JSON line)."""
import gc
import json
import os
import sys
import time

import numpy as np

from .common import (BASELINE_CONFIG, HBM_PEAK_GBS, PREHEAT_S, ROOT, SCENES, STAGE, WORKLOADS, calibration, csrc_fingerprint, digest64, launch_stats,
                     synth_into, workload_label)
from .cpu import cpu_baseline
from .line import dump_line
from .legs import ingest_leg, physical_accuracy, reference_product, small_image_legs, sustained_leg, throughput_batched, throughput_mode
from .sharded_legs import direct_children, sharded_report, verify_sharded


def run_rank(args, rank, local_rank, world):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's host driver only supports dmabuf IPC: RCCL's P2P set-up fails without it
    if os.environ.get("SLGC_RANKS_AS_HOSTS") == "1" and world > 1 and "NCCL_HOSTID" not in os.environ:
        # the same TEST MODE under an external launcher (torch.distributed.run gives every rank the same environment): see spawn_ranks
        os.environ.update(NCCL_HOSTID=f"slgc-rank-{rank}-{os.environ.get('MASTER_PORT', '0')}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket")
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    from scanner import _native
    if os.environ.get("SLGC_BENCH_FAULTHANDLER_S"):            # stall hunting (tools/jobs/rccl_hang_hunt.py): every rank's Python stacks after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SLGC_BENCH_FAULTHANDLER_S"]), exit=False, file=sys.stderr)
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
    rig_now = [SCENES[args.scene]["rig"]]             # the calibration the context holds: every synthetic capture belongs to a rig (common.SCENES)
    ctx.set_calibration(*calibration(cam_w, cam_h, proj_w, proj_h, rig=rig_now[0]))

    def use_rig(rig):
        """Switch the context to another rig's calibration (the next scan rebuilds the ray tables: legs run their warm-up after this)."""
        if rig != rig_now[0]:
            ctx.synchronize()
            ctx.set_calibration(*calibration(cam_w, cam_h, proj_w, proj_h, rig=rig))
            rig_now[0] = rig
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
    def make_stacks(scene, n_runs=1):
        """args.buffers rotated captures of `scene` (n_runs > 1: that many captures of the same scene back to back in one buffer, N * plane apart,
        each with its own noise -- the reference's MAX_NB_RUNS repeats, src/3-capture_decode.py:48)."""
        use_rig(SCENES[scene]["rig"])
        out_ = []
        for b in range(max(1, args.buffers)):
            s = ctx.alloc(max(16, n_runs * N * plane))
            for r in range(n_runs):
                if rows:
                    synth_into(ctx, scene, s.at(r * N * plane) if r else s.ptr, plane, N, cam_h, cam_w, (proj_w, proj_h), 1 + b + 100 * r, row0=row0, rows=rows)
            out_.append(s)
        return out_

    t_run0 = time.perf_counter()
    leg_s = {}

    class leg:                                  # wall seconds of every leg of the run (side file: where the run's time goes)
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            STAGE[0] = self.name
            self.t0 = time.perf_counter()

        def __exit__(self, *exc):
            leg_s[self.name] = round(leg_s.get(self.name, 0.0) + time.perf_counter() - self.t0, 3)
            STAGE[0] = "extras"

    single = G == 1 and not use_comm
    extras = single and args.extras != "none" and args.mode == "algebraic" and args.tri == "lut"       # "lite" legs: what the printed line needs
    extras_full = extras and args.extras == "full"                                                     # everything else (side file only)
    # EVERY input stack of every leg is allocated and filled here, before the first timed window: a hipMalloc / hipFree of a 0.5 GB stack
    # between two legs left one launch of 10 ms in the next leg's samples (round 4: first touch of fresh page tables inside the event pairs)
    with leg("synthetic captures"):
        stacks = make_stacks(args.scene)
        scene_stacks = {args.scene: stacks}
        if extras:
            for name in (("s-scene", "s-uniform", "noisy-physical", "physical") if extras_full else ("s-scene",)):
                if name in scene_stacks or (SCENES[name]["kind"] == "uniform" and (cam_w % 4 or plane % 4)):
                    continue
                scene_stacks[name] = make_stacks(name)
        pairs = make_stacks(args.scene, n_runs=2) if extras_full else None
        use_rig(SCENES[args.scene]["rig"])
    maps = ctx.alloc(max(16, band_px * 4))
    xyz = ctx.alloc(max(16, band_px * 12))
    count = ctx.alloc(16).zero()
    sharded_scanner = None
    if use_comm:                            # (sets slgc_tune "image_rows": the ray-table choice below is the whole image's)
        direct_key = "bench_" + (os.environ.get("SLGC_UID_KEY") or f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
        if args.exchange_impl == "direct":
            STAGE[0] = "direct exchange set-up (shared segment, IPC handles)"
            exch = sharded.DirectExchange(ctx, rank, G, direct_key)
        else:
            exch = sharded.RcclExchange(ctx)
        sharded_scanner = sharded.ShardedScanner(ctx, exch, plan, (proj_w, proj_h), N, mode=mode, exchange_kind=args.exchange, wire=args.wire)
        STAGE[0] = "setup"
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

    def step(i, counted=False, mode=mode, src=None, no_maps=False, n_runs=1):
        src = stacks if src is None else src
        s = src[i % len(src)]
        if pipelined:
            return sharded_scanner.submit(s.ptr, plane)      # exchange of this scan overlaps the neighbours' kernels
        if use_comm:
            return sharded_scanner.scan(s.ptr, plane)
        ctx.scan_dev(s.ptr, n_runs, N * plane, plane, N, rows, cam_w, row0, (proj_w, proj_h), xyz.ptr, count.ptr if counted else None,
                     None if no_maps else maps.at(0), None if no_maps else maps.at(band_px * 2), mode=mode)
        return None

    def drain():
        if sharded_scanner is not None:
            sharded_scanner.flush()
        ctx.synchronize()

    def timed(K, W_, stride=None, preheat=True, scene=None, **kw):
        use_rig(SCENES[scene or args.scene]["rig"])
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
        # untimed rehearsal of the event-bound launch path (the timed region's launches carry HIP-event pairs, the warm-up's do not): on a box's
        # first run the runtime's pages behind it were touched for the first time INSIDE the region -- 0.15 ms of host stall in 2.4 ms (round 5:
        # `value` 6 % under the kernel's rate in the first of five runs on a fresh box, never in the other four)
        ctx.prof_begin(8, 1)
        for i in range(2):
            step(i, **kw)
        drain()
        ctx.prof_end()
        ctx.prof_samples()
        for i in range(W_):
            step(i, **kw)
        drain()
        if use_comm:
            ctx.comm_barrier()
        ctx.prof_begin(K + 8, stride or args.event_stride)   # HIP-event pair bound to every stride-th kernel dispatch of the region
        gc_was = gc.isenabled()
        gc.disable()                                         # (a collection inside a 2.4 ms region is a visible share of it)
        t0 = time.perf_counter()
        tot = None
        for i in range(K):
            tot = step(i, **kw)
        drain()                                              # the K-th scan's exchange + triangulation are inside the timed region
        if use_comm:
            ctx.comm_barrier()
        el = time.perf_counter() - t0
        if gc_was:
            gc.enable()
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

    def scene_stats(src):
        """valid / guard-flagged pixels of one scan of src[0] (untimed)"""
        count.zero()
        ctx.scan_dev(src[0].ptr, 1, N * plane, plane, N, rows, cam_w, row0, (proj_w, proj_h), xyz.ptr, None, maps.at(0), maps.at(band_px * 2), mode=mode_fused)
        ctx.guard_count_dev(maps.at(0), maps.at(band_px * 2), rows, cam_w, row0, (proj_w, proj_h), count.ptr)
        ctx.synchronize()
        return tuple(int(x) for x in count.download((2,), np.uint64))

    def time_decode(src):
        """the decode kernel by itself, back to back over the rotated stacks (no other kernel's write-back in its way) -> (ms, launches, samples), path"""
        for i in range(max(3, len(src))):
            ctx.decode_dev(src[i % len(src)].ptr, 1, N * plane, plane, N, rows, cam_w, maps.at(0), maps.at(band_px * 2), variant=args.variant)
        ctx.synchronize()
        ctx.prof_begin(args.steps + 8)
        for i in range(args.steps):
            ctx.decode_dev(src[i % len(src)].ptr, 1, N * plane, plane, N, rows, cam_w, maps.at(0), maps.at(band_px * 2), variant=args.variant)
        return ctx.prof_end() + (ctx.prof_samples(),), ctx.last_scan_path()

    # ---- the same kernels on every synthetic capture of common.SCENES (SURVEY.md 8(d) names S-uniform as the worst case and the S-scene as the
    # realistic one; the physical ones are what a scanner sees): fused scan + decode kernel timed, valid / guard-flagged pixels counted
    scene_legs = {}
    s_scene_stacks = scene_stacks.get("s-scene")
    W_leg = max(2, args.warmup // 2, len(stacks))            # the warm-up of a leg touches every one of its rotated stacks
    if extras:
        for name, st in scene_stacks.items():
            if name == args.scene:
                continue
            with leg(f"scene leg {name}"):
                o_el, o_kms, o_kn, _, o_samples = timed(args.steps, W_leg, preheat=False, scene=name, mode=mode_fused, src=st)
                o_exec = ctx.last_scan_path()
                o_valid, o_flag = scene_stats(st)
                o_dec, o_dec_exec = time_decode(st)
                scene_legs[name] = [o_el, o_kms, o_kn, o_samples, o_exec, o_valid, o_flag, None, o_dec, o_dec_exec]
    # The same step with EVERY launch bracketed by its event pair.  A bracketed launch is fenced off from its neighbours (the step grows by the
    # ~5 us the pairs cost), an unbracketed one starts in the drain of the launch before it: at 4096x3000 the two agree, a 25 us kernel
    # (1920x1080) measures 23.2 us fenced and 24.9 us back to back on the same box (tools/jobs/r5_c2_conditions.sh).  roofline.frac stays the
    # timed region's; this is the kernel by itself.
    isolated = None
    if extras:
        with leg("isolated launches"):
            use_rig(SCENES[args.scene]["rig"])
            i_el, i_kms, i_kn, _, i_samples = timed(min(40, max(10, args.steps)), W_leg, stride=1, preheat=False)
            isolated = (i_kms, i_kn, i_samples, ctx.last_scan_path())
    head_dec = None
    if extras:
        with leg("decode kernel alone"):
            use_rig(SCENES[args.scene]["rig"])
            head_dec = time_decode(stacks)
    # (accuracy against the generator's truth allocates: after every timed window of the scene legs)
    head_acc = None
    if extras_full:
        with leg("accuracy vs true surface"):
            for name in scene_legs:
                if SCENES[name]["kind"] == "physical":
                    use_rig(SCENES[name]["rig"])
                    scene_legs[name][7] = physical_accuracy(ctx, N, cam_h, cam_w, row0, rows, plane, (proj_w, proj_h), name, maps, xyz, band_px, mode_fused)
            if SCENES[args.scene]["kind"] == "physical":
                use_rig(SCENES[args.scene]["rig"])
                head_acc = physical_accuracy(ctx, N, cam_h, cam_w, row0, rows, plane, (proj_w, proj_h), args.scene, maps, xyz, band_px, mode_fused)

    # ---- two captures per scan, max-merged per code bit inside the kernel (the reference always merges MAX_NB_RUNS = 2: src/3-capture_decode.py:48,95-96)
    two_runs = None
    if extras_full:
        with leg("two runs"):
            two_runs = timed(args.steps, W_leg, preheat=False, mode=mode_fused, src=pairs, n_runs=2) + (ctx.last_scan_path(),)

    sustained = None
    use_rig(SCENES[args.scene]["rig"])
    if single and args.extras != "none" and args.sustained > 0:
        with leg("sustained"):
            sustained = sustained_leg(ctx, step, drain, args.sustained, cam_w * rows / 1e6)

    other = None
    if extras_full:
        with leg("other pipeline"):
            om = mode_fused if args.pipeline == "split" else mode_split
            other = timed(args.steps, W_leg, preheat=False, mode=om)
            other_executed = ctx.last_scan_path()

    xyz_only = None
    if extras_full:
        # the same scan for a caller that wants the cloud only (no map buffers passed): the fused kernel then moves exactly SURVEY 8(d)'s N + 12 B/pixel
        with leg("xyz only"):
            xyz_only = timed(args.steps, W_leg, preheat=False, mode=mode_fused, no_maps=True)
            xyz_only_executed = ctx.last_scan_path()

    dec_alone = None
    if extras and s_scene_stacks is not None:
        if args.scene == "s-scene":
            dec_alone, dec_alone_exec = head_dec
        else:
            dec_alone, dec_alone_exec = scene_legs["s-scene"][8], scene_legs["s-scene"][9]     # (timed in the S-scene leg above)

    movement = None
    if extras and N in (42, 44, 46) and band_px % 256 == 0 and plane % 4 == 0:
        t_mv0 = time.perf_counter()
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
        leg_s["movement-only yardstick"] = round(time.perf_counter() - t_mv0, 3)

    ref_product = None
    if extras_full and row0 == 0 and s_scene_stacks is not None:
        with leg("reference product"):
            use_rig(SCENES["s-scene"]["rig"])
            ref_product = reference_product(ctx, _native, s_scene_stacks, N, plane, rows, cam_w, row0, (proj_w, proj_h), maps, xyz, band_px, args.steps,
                                            mode_fused)
            ref_product["scene"] = "s-scene"

    thr = thr_batched = None
    if not args.no_throughput_mode and args.extras != "none" and args.mode == "algebraic" and args.tri == "lut":
        with leg("throughput mode"):
            thr = throughput_mode(ctx, _native, G, max(5, args.steps // 4), mode_fused, device, args.streams, collective=use_comm, scene=args.scene)
            thr_batched = throughput_batched(ctx, _native, G, max(5, args.steps // 4), mode_fused, device, collective=use_comm, scene=args.scene)

    ingest = None
    if extras_full and row0 == 0 and N in (42, 44, 46) and band_px % 4 == 0:
        with leg("ingest"):
            use_rig(SCENES[args.scene]["rig"])
            ingest = ingest_leg(ctx, args.scene, N, rows, cam_w, (proj_w, proj_h), max(5, args.steps // 2), mode_fused, maps, xyz)

    small = None
    if extras_full and args.workload == "c3_4096x3000x44" and not args.no_small_images:
        with leg("small images"):
            small = small_image_legs(_native, device, args.steps, mode_fused)

    # ---- what one scan holds: valid pixels, pixels on the guarded triangulation path (untimed)
    use_rig(SCENES[args.scene]["rig"])
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
            return f"k_decode_pk<4,128,nt,FUSE> {spec} + triangulation tail (camera rays: {'node table' if ex['node_table'] else 'per-pixel table'})"

        def kernel_roofline(pipeline, kms, kn, samples, ex=None, scene=None, n_runs=1, step_ms=None):
            """SURVEY.md 8(d) byte definitions: decode kernel N + 4 B/pixel (N uint8 reads, 2 int16 writes); fused decode -> XYZ
            N + 12 B/pixel (n_runs captures per scan: n_runs * N frame bytes).  The fused kernel also writes the 4 B/pixel maps (a product):
            frac_incl_maps counts them too."""
            per_px = (n_runs * N + 4) if pipeline == "split" else (n_runs * N + 12)
            avg_ms = kms / max(1, kn)
            st = launch_stats(samples)
            med_ms = st.get("median_launch_ms") or avg_ms
            # step_ms = the region's elapsed time / its launches, for regions of ONE kernel per step on one stream: the kernels run one after the
            # other, so their mean duration cannot exceed it.  Where the bracketed launches' median does (a 25 us kernel: an event pair perturbs the
            # launch it brackets -- 25.3 us sampled inside steps of 24.3 us), the sample is not the region's kernel time; the step time is its bound.
            sampled_ms, from_step = med_ms, False
            if step_ms and kn and med_ms > step_ms:
                med_ms, from_step = step_ms, True
            # frac / achieved come from the MEDIAN launch of the timed region (one hiccup of the box -- round 4 saw a 10 ms launch among twenty of
            # 0.1 ms -- must not decide the number); the mean is beside it (frac_mean, avg_launch_ms) with the count of launches above 2 x median
            ach = per_px * band_px / (med_ms * 1e-3) / 1e9
            ach_mean = per_px * band_px / (avg_ms * 1e-3) / 1e9
            r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                 "frac_mean": round(ach_mean / HBM_PEAK_GBS, 4), "achieved_mean": round(ach_mean, 1),
                 "frac_from": ("step time: the bracketed launches (median %.5f ms) ran slower than the region's average step (%.5f ms), which bounds the kernel's "
                               "mean duration from above" % (sampled_ms, step_ms)) if from_step else "median launch",
                 "traffic": None, "kernel": kernel_name(ex or executed, pipeline),
                 "avg_launch_ms": round(avg_ms, 5), "launches_timed": kn, **st, "launch_ms_used": round(med_ms, 5),
                 "algorithmic_bytes_per_px": per_px, "algorithmic_bytes_per_launch": per_px * band_px}
            if pipeline != "split":
                r["frac_incl_maps"] = round((n_runs * N + 16) * band_px / (med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                r["frac_incl_maps_note"] = "N + 16 B/pixel: the 4 B/pixel int16 maps the fused kernel also writes counted as algorithmic"
            sc = scene or args.scene
            live = (getattr(args, "pmc_live", None) or {}).get(f"{pipeline}/{sc}") if n_runs == 1 else None
            t = traffic_db.get(f"{args.workload}/g{G}/{pipeline}/{sc}") if n_runs == 1 else None
            if live:
                # counters collected by THIS run: bench.py started rocprofv3 --pmc children (FETCH_SIZE, WRITE_SIZE: separate passes, no tracing)
                # before touching the GPU (benchlib/pmc.py)
                r["traffic"] = live["hbm_bytes_per_launch"]
                r["traffic_source"] = live["source"]
                r["traffic_scene"] = sc
            elif t and t.get("csrc_fingerprint") == fp:
                r["traffic"] = t["hbm_bytes_per_launch"]
                r["traffic_source"] = "committed constant (profiles/traffic.json, same kernel sources by fingerprint): " + str(t.get("source"))
                r["traffic_scene"] = sc
            elif t:
                r["traffic_note"] = ("profiles/traffic.json was measured on other kernel sources (fingerprint mismatch): stale, not reported; "
                                     "re-run tools/pmc.sh")
            if r["traffic"]:
                r["traffic_over_algorithmic"] = round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)
            return r

        main_pipeline = "split" if (use_comm and args.exchange == "maps") else args.pipeline
        if not use_comm:                                     # the label follows the launch, not the request
            main_pipeline = "fused" if executed["path"] in ("fused", "batch-fused") else "split"
        out = {
            "metric": "Mpixels/s decode+triangulate", "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": G,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_label(args.workload, cam_w, cam_h, proj_w, proj_h, N, G if use_comm else 0, "direct (IPC push)" if args.exchange_impl == "direct" else "RCCL"),
                       "pipeline": ({"maps": "decode kernel per band, map bands all-gathered, full-image triangulation kernel on every rank",
                                     "xyz": "fused kernel per band, map + XYZ bands all-gathered in place",
                                     "records": "fused kernel per band, compaction, 16-byte XYZ+key records all-gatherv'ed"}[args.exchange] if use_comm
                                    else {"fused": "fused (one kernel)", "batch-fused": "fused (one kernel, batched)", "split": "split (decode kernel + triangulation kernel)",
                                          "split-ragged": "split (decode kernel + triangulation kernel, with byte-wide / per-pixel fallback kernels)"}.get(executed["path"], executed["path"])),
                       "executed": {**executed, "source": "slgc_last_scan_path after the timed region", "requested_pipeline": args.pipeline},
                       "extras": args.extras,
                       "scene": SCENES[args.scene]["label"], "scene_name": args.scene, "rig": SCENES[args.scene]["rig"],
                       "rig_note": "survey = SURVEY.md 8(d)'s calibration; covering = the same camera and stereo pose with a projector whose addressable "
                                   "2^L x 2^L raster covers the camera's field of view (benchlib/common.py: calibration)",
                       "valid_pixels_per_scan": valid, "valid_fraction": round(valid / max(1, band_px), 4) if valid is not None else None,
                       "rows_per_gpu": rows, "triangulation": args.mode + "/" + args.tri, "input_buffers_rotated": len(stacks), "plane_pad_bytes": args.plane_pad,
                       "outputs": "int16 h/v maps + dense float32 XYZ in HBM" + ("" if not use_comm else "; whole cloud reassembled on every rank"),
                       "preheat_s": args.preheat, "event_stride": args.event_stride,
                       "luts_hoisted_us": round(luts_us, 1),
                       "luts_hoisted_note": "per-calibration ray tables (both cv2.undistortPoints calls on integer pixel coordinates) built once "
                                            "before the timed region, not per scan",
                       "camera_rays": (lambda use, err: {"node_table_in_use": use, "node_table_error_vs_limit_2.4e-7": err,
                                                         "note": "per-pixel table 8 B/pixel, or (images above 12 MB of rays) the every-4th-column "
                                                                 "table 2 B/pixel + a cubic through 4 nodes per 4-pixel group; flat triangles and "
                                                                 "zero-crossing rays always read the exact per-pixel table"})(*ctx.ray_table_info()),
                       "guard_flagged_pixels": flagged,
                       "guard_note": "decodable pixels of one scan that triangulation redoes on the reference's float32 intermediates (flat triangles)"},
            "roofline": kernel_roofline(main_pipeline, dec_ms, dec_n, dec_samples, step_ms=(ms_per_step if (single and main_pipeline == "fused") else None)),
            "valid_pixels_per_scan": valid,
            "device": ctx.device_name(),
            "pmc": {"requested": getattr(args, "pmc", "off"), "result": getattr(args, "pmc_note", None)},
        }
        if isolated is not None and isolated[3]["path"] in ("fused", "split"):
            ir = kernel_roofline("fused" if isolated[3]["path"] == "fused" else "split", isolated[0], isolated[1], isolated[2], isolated[3])
            out["roofline"]["isolated_frac"] = ir["frac"]
            out["roofline"]["isolated"] = {"frac": ir["frac"], "frac_mean": ir["frac_mean"], "median_launch_ms": ir.get("median_launch_ms"), "avg_launch_ms": ir["avg_launch_ms"],
                                           "launches_timed": ir["launches_timed"], "outliers": ir.get("outliers"),
                                           "note": "the same step right after the timed region with every launch bracketed by its HIP-event pair (fenced off from its "
                                                   "neighbours); `frac` above is the timed region's own launches, every event_stride-th bracketed, the rest back to back"}
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
            xr["bound_note"] = ("N bytes read + 12 written per pixel: a read-dominated stream (bound \"hbm\"); the store-bound stage of this library is "
                                "the x-major list build -- reference_product.list_stage_roofline carries bound \"hbm-write\" and the write ceiling measured in-run")
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
            dr = kernel_roofline("split", *dec_alone, ex=dec_alone_exec, scene="s-scene")
            if movement is not None:
                beside(dr, "decode_ms")
            out["decode_kernel_alone"] = {"roofline": dr, "scene": "s-scene",
                                          "note": "decode kernel launched back to back on rotated S-scene stacks (~80 % of the pixels decodable: the decode "
                                                  "kernel's heavier input), same run (the north star's >= 60 % of HBM roofline on the decode kernel at 4096x3000x44)"}
        if ingest is not None:
            out["ingest"] = ingest
        if small is not None:
            out["small_images"] = small
        if ref_product is not None:
            out["reference_product"] = ref_product
        if sustained is not None:
            out["sustained"] = sustained
        if extras:
            # one table over every synthetic capture: fused scan + decode kernel, valid / guard-flagged pixels
            table = {args.scene: {"scene": SCENES[args.scene]["label"], "rig": SCENES[args.scene]["rig"], "headline": True, "value": round(value, 1), "unit": "Mpixels/s",
                                  "ms_per_step": round(ms_per_step, 4), "frac": out["roofline"]["frac"], "frac_mean": out["roofline"]["frac_mean"],
                                  "avg_launch_ms": out["roofline"]["avg_launch_ms"], "median_launch_ms": out["roofline"].get("median_launch_ms"),
                                  "outliers": out["roofline"].get("outliers"),
                                  "valid_pixels_per_scan": valid, "guard_flagged_pixels": flagged, "executed": executed,
                                  "decode_kernel": (lambda d: {"frac": d["frac"], "avg_launch_ms": d["avg_launch_ms"]})(kernel_roofline("split", *head_dec[0], ex=head_dec[1]))}}
            for name, (o_el, o_kms, o_kn, o_samples, o_exec, o_valid, o_flag, acc, o_dec, o_dec_exec) in scene_legs.items():
                fr = kernel_roofline("fused" if o_exec["path"] == "fused" else "split", o_kms, o_kn, o_samples, o_exec, scene=name,
                                     step_ms=(o_el / args.steps * 1e3 if o_exec["path"] == "fused" else None))
                dr = kernel_roofline("split", *o_dec, ex=o_dec_exec, scene=name)
                table[name] = {"scene": SCENES[name]["label"], "rig": SCENES[name]["rig"], "value": round(mpix_per_step * args.steps / o_el, 1), "unit": "Mpixels/s",
                               "ms_per_step": round(o_el / args.steps * 1e3, 4), "frac": fr["frac"], "frac_mean": fr["frac_mean"], "avg_launch_ms": fr["avg_launch_ms"],
                               "median_launch_ms": fr.get("median_launch_ms"), "outliers": fr.get("outliers"),
                               "fused_time_over_s_scene": None, "valid_pixels_per_scan": o_valid, "guard_flagged_pixels": o_flag, "executed": o_exec,
                               "roofline": fr, "decode_kernel": {"frac": dr["frac"], "avg_launch_ms": dr["avg_launch_ms"], "roofline": dr}}
                if acc is not None:
                    table[name]["accuracy_vs_true_surface"] = acc
            ref_ms = table.get("s-scene", {}).get("median_launch_ms")
            for name, row in table.items():
                row["fused_time_over_s_scene"] = round(row["median_launch_ms"] / ref_ms, 3) if ref_ms and row.get("median_launch_ms") else None
            out["scenes"] = table
            # SURVEY.md 8(d) defines the synthetic capture as the S-scene: its fraction rides INSIDE the roofline object next to the headline capture's
            #  -- the same kernel, same run, same byte definition
            srow = table.get("s-scene")
            if srow and out.get("roofline"):
                sr = srow.get("roofline") or out["roofline"]
                out["roofline"]["frac_s_scene"] = srow["frac"]
                out["roofline"]["traffic_over_algorithmic_s_scene"] = sr.get("traffic_over_algorithmic")
                out["roofline"]["headline_scene"] = args.scene
            if "s-scene" in scene_legs:                      # (kept under its old key as well)
                row = table["s-scene"]
                out["other_scene"] = {"scene": "s-scene", "value": row["value"], "unit": "Mpixels/s", "ms_per_step": row["ms_per_step"], "executed": row["executed"],
                                      "roofline": row["roofline"], "valid_pixels_per_scan": row["valid_pixels_per_scan"],
                                      "guard_flagged_pixels": row["guard_flagged_pixels"],
                                      "note": "the headline step on SURVEY.md 8(d)'s S-scene with SURVEY.md 8(d)'s calibration, same run (the like-for-like figure of earlier rounds)"}
            if head_acc is not None:
                out["physical_scene_accuracy"] = head_acc
        if two_runs is not None:
            t_el, t_kms, t_kn, _, t_samples, t_exec = two_runs
            out["two_runs"] = {"value": round(mpix_per_step * args.steps / t_el, 1), "unit": "Mpixels/s", "ms_per_step": round(t_el / args.steps * 1e3, 4), "executed": t_exec,
                               "roofline": kernel_roofline("fused" if t_exec["path"] == "fused" else "split", t_kms, t_kn, t_samples, t_exec, n_runs=2),
                               "note": "n_runs = 2: two captures of the headline scene (independent noise) per scan, max-merged per code bit inside the kernel "
                                       "like the reference's MAX_NB_RUNS = 2 (src/3-capture_decode.py:48,95-96); 2 N + 12 algorithmic bytes per pixel"}
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
        if single and not args.no_cpu_baseline and args.extras != "none":
            # (bench.py runs the CPU leg BEFORE this process touches the GPU, beside the counter children: args.cpu_baseline_result)
            with leg("cpu baseline (in line)"):
                out["cpu_baseline"] = getattr(args, "cpu_baseline_result", None) or cpu_baseline(level=args.extras)
        if extras and head_dec is not None:
            out["decode_kernel_headline"] = {"roofline": kernel_roofline("split", *head_dec[0], ex=head_dec[1]), "scene": args.scene,
                                             "note": "decode kernel launched back to back on the headline scene's rotated stacks, same run"}
        out["leg_seconds"] = dict(leg_s, **(getattr(args, "pre_gpu_seconds", None) or {}), rank_total=round(time.perf_counter() - t_run0, 3))

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
                report = dict(out)
                if extra:
                    report.update(extra)
                # the full report goes to a side file; the LAST stdout line is the compact object the driver parses (benchlib/line.py)
                path = getattr(args, "extras_file", None) or os.path.join(ROOT, "gpurun_out", "bench_extras.json")
                try:
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    with open(path, "w") as f:
                        json.dump(report, f, indent=1)
                    shown = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
                except OSError as e:
                    shown = None
                    print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
                print(dump_line(report, shown), flush=True)

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
    if use_comm and args.exchange == "maps" and args.extras != "none" and pipelined:
        # The first run on real xGMI is rare: time the other exchange forms too (same stacks, same pipelining, each verified against the maps
        # the main strategy left) -- extras after the counted region, a failure here is reported and changes nothing above.
        sharded_scanner.submit(stacks[last_stack].ptr, plane)
        sharded_scanner.flush()
        h_main, v_main, _ = sharded_scanner.fetch_dense()
        main_digest = digest64(h_main, v_main)
        alternatives = {}
        other_wire = "int16" if sharded_scanner.wire == "hv24" else "hv24"           # whichever wire the main strategy did not use
        other_impl = "rccl" if args.exchange_impl == "direct" else "direct"                # ... and the exchange implementation it did not use
        for label, kind, wire, impl in (("maps_" + other_wire, "maps", other_wire, args.exchange_impl), ("xyz", "xyz", "int16", args.exchange_impl),
                                        (f"maps_{sharded_scanner.wire}_{other_impl}", "maps", sharded_scanner.wire, other_impl)):
            if wire == "hv24" and int((N - 2) / 4) > _native.WIRE_MAX_CODE_BITS:
                continue
            if impl == "direct" and args.exchange_impl != "direct" and os.environ.get("SLGC_BENCH_NO_DIRECT_ALT") == "1":
                continue
            if impl != args.exchange_impl and rank == 0:
                # the exchange implementation that has never run on this kind of machine goes last, after what is known so far is on disk: if it
                # takes the process down the side file still has the rest (the printed line needs this process alive: exactly ONE line is printed)
                try:
                    side = getattr(args, "extras_file", None) or os.path.join(ROOT, "gpurun_out", "bench_extras.json")
                    os.makedirs(os.path.dirname(side), exist_ok=True)
                    with open(side, "w") as f:
                        json.dump(dict(out, sharded_alternatives=dict(alternatives), provisional=f"written before the {label} alternative ran"), f, indent=1)
                except OSError:
                    pass
            if impl == "direct" and args.exchange_impl != "direct" and os.environ.get("SLGC_BENCH_DIRECT_IN_PROCESS") != "1":
                # ... and it runs in CHILD processes, one per rank (direct_child.py): a fault in a path that has never crossed a link takes a
                # child down, not the rank that still owes the line
                STAGE[0] = f"sharded alternatives: {label} (child processes)"
                alternatives[label] = direct_children(args, rank, G, device, direct_key + "_child", (cam_w, cam_h), (proj_w, proj_h), N, sharded_scanner.wire,
                                                      len(stacks), last_stack, main_digest, mode, max(5, args.steps // 2))
                continue
            try:
                STAGE[0] = f"sharded alternatives: {label}"
                if impl == "direct":
                    os.environ.setdefault("SLGC_DIRECT_TIMEOUT_S", "5")     # an extra after the counted region: a healthy exchange takes well under a millisecond
                    alt_exch = exch if args.exchange_impl == "direct" else sharded.DirectExchange(ctx, rank, G, direct_key + "_alt")
                else:
                    alt_exch = sharded.RcclExchange(ctx)
                alt = sharded.ShardedScanner(ctx, alt_exch, plan, (proj_w, proj_h), N, mode=mode, exchange_kind=kind, wire=wire)
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
                alternatives[label] = {"value": round(cam_w * cam_h / 1e6 * K / el_alt, 1), "unit": "Mpixels/s", "steps": K, "exchange_impl": impl,
                                       "bytes_per_pixel_on_the_links": 16 if kind == "xyz" else 3 if wire == "hv24" else 4,
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


