"""bench.py: extra legs of the single-GPU run (throughput mode, sustained, reference-shaped product, accuracy against the generator's truth)."""
import os
import time

import numpy as np

from .common import HBM_PEAK_GBS, SCENES, WORKLOADS, calibration, synth_into


# ------------------------------------------------------------------------------------------------ configs[4]
def throughput_lanes(_native, device, per_rank, n_streams=2, scene="s-scene"):
    """BASELINE.json configs[4] on one GPU: `n_streams` contexts (HIP streams), each with its own rotated 1920x1080x44 stacks
    (>= 4 distinct stacks in total: 364 MB > Infinity Cache) and one set of output buffers.  -> [(ctx, stacks, maps, xyz)]"""
    cw, ch, pw, ph, n = WORKLOADS["c2_1920x1080x44"]
    px = cw * ch
    lanes = []
    for sidx in range(max(1, n_streams)):
        c = _native.Context(device)
        c.set_calibration(*calibration(cw, ch, pw, ph, rig=SCENES[scene]["rig"]))
        if n_streams > 1:
            # several contexts share the GPU here: the per-phase issue priorities the library gives a launch that fills the chip in one round
            # ("tail last") assume the launch is alone -- with a second stream's kernel beside it they cost 12 % (90 -> 79 k Mpixels/s)
            c.tune("prio", 0)
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
    c.set_calibration(*calibration(cw, ch, pw, ph, rig=SCENES[scene]["rig"]))
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



def ingest_leg(ctx, scene, N, H, W, proj_size, steps, mode_fused, maps, xyz, buffers=2):
    """The scan from the camera's BGR frames (src/3-capture_decode.py:66-70 + :75): slgc_scan_bgr_dev -- one kernel, luma inside the frame loads, 3 N + 12
    algorithmic bytes per pixel -- against the chain slgc_to_gray_dev (per frame, into a grey stack) + slgc_scan_dev, which moves 3 N + N + N + 12.  Rotated
    BGR captures; the ordinary scan's kernel time comes from HIP events bound to the dispatch, the chain's from events around the whole step."""
    px = H * W
    plane = 3 * px
    caps = []
    gray = ctx.alloc(N * px)
    for b in range(max(2, buffers)):
        synth_into(ctx, scene, gray.ptr, px, N, H, W, proj_size, 31 + b)
        c = ctx.alloc(N * plane)
        ctx.synth_bgr_dev(gray.ptr, px, N, H, W, c.ptr, plane)
        caps.append(c)
    from scanner import _native
    lib = _native.lib()

    def fused(i):
        ctx.scan_bgr_dev(caps[i % len(caps)].ptr, 1, N * plane, plane, N, H, W, 0, proj_size, xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode_fused)

    def separate(i):
        c = caps[i % len(caps)]
        ctx._ck(lib.slgc_to_gray_dev(ctx._h, c.ptr, N * px, 15, gray.ptr))          # the whole capture in one launch: frames are contiguous
        ctx.scan_dev(gray.ptr, 1, N * px, px, N, H, W, 0, proj_size, xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode_fused)

    K = max(5, steps)
    for i in range(3):
        fused(i)
    ctx.synchronize()
    ctx.prof_begin(K + 8, 1)
    t0 = time.perf_counter()
    for i in range(K):
        fused(i)
    ctx.synchronize()
    el_f = time.perf_counter() - t0
    kms, kn = ctx.prof_end()
    samples = ctx.prof_samples()
    path = ctx.last_scan_path()
    for i in range(3):
        separate(i)
    ctx.synchronize()
    ctx.event_record(2)
    t0 = time.perf_counter()
    for i in range(K):
        separate(i)
    ctx.event_record(3)
    ctx.synchronize()
    el_s = time.perf_counter() - t0
    sep_ms = ctx.event_elapsed_ms(2, 3) / K
    avg = kms / max(1, kn)
    per_px = 3 * N + 12
    out = {"scene": scene, "workload": f"{W}x{H} cam, {N} BGR frames", "executed": path,
           "ingest_fused": {"value": round(px / 1e6 * K / el_f, 1), "unit": "Mpixels/s", "ms_per_step": round(el_f / K * 1e3, 4),
                            "roofline": {"bound": "hbm", "achieved": round(per_px * px / (avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(per_px * px / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg, 5), "launches_timed": kn,
                                         "algorithmic_bytes_per_px": per_px, "algorithmic_bytes_per_launch": per_px * px,
                                         "kernel": "k_decode_pk<4,128,nt,FUSE,NS,BGR=1>: dwordx3 frame loads, cv2.cvtColor's 15-bit luma in registers (v_dot4_u32_u8)"}},
           "ingest_separate": {"value": round(px / 1e6 * K / el_s, 1), "unit": "Mpixels/s", "ms_per_step": round(el_s / K * 1e3, 4), "device_ms_per_step": round(sep_ms, 4),
                               "bytes_per_px_moved": 3 * N + N + N + 12,
                               "frac_on_3N_plus_12": round(per_px * px / (sep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "slgc_to_gray_dev (grey stack written to HBM) + slgc_scan_dev (reads it back)"},
           "speedup": round(el_s / el_f, 3)}
    for c in caps:
        c.free()
    gray.free()
    return out


def small_image_legs(_native, device, steps, mode_fused, workloads=("c2_1920x1080x44", "c1_1280x720x42"),
                     scenes=("physical", "s-scene", "s-uniform", "noisy-physical")):
    """BASELINE configs[1] / configs[0] sizes (one round of resident waves: head and tail of the kernel stay exposed) through the fused kernel on
    every synthetic capture, in the default run: kernel time from HIP events bound to the dispatches, rotated stacks > Infinity Cache."""
    out = {}
    for wl in workloads:
        cw, ch, pw, ph, n = WORKLOADS[wl]
        px = cw * ch
        c = _native.Context(device)
        maps, xyz, cnt = c.alloc(px * 4), c.alloc(px * 12), c.alloc(16)
        rows = {}
        for scene in scenes:
            c.set_calibration(*calibration(cw, ch, pw, ph, rig=SCENES[scene]["rig"]))
            stacks = []
            for b in range(max(2, -(-300_000_000 // (n * px)))):
                st = c.alloc(n * px)
                synth_into(c, scene, st.ptr, px, n, ch, cw, (pw, ph), 21 + b)
                stacks.append(st)

            def scan(i):
                c.scan_dev(stacks[i % len(stacks)].ptr, 1, n * px, px, n, ch, cw, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode_fused)

            for i in range(4):
                scan(i)
            c.synchronize()
            K = max(10, steps)
            c.prof_begin(K + 8, 1)
            t0 = time.perf_counter()
            for i in range(K):
                scan(i)
            c.synchronize()
            el = time.perf_counter() - t0
            kms, kn = c.prof_end()
            path = c.last_scan_path()
            cnt.zero()
            c.guard_count_dev(maps.at(0), maps.at(px * 2), ch, cw, 0, (pw, ph), cnt.ptr)
            c.synchronize()
            valid, flagged = (int(x) for x in cnt.download((2,), np.uint64))
            avg = kms / max(1, kn)
            rows[scene] = {"avg_launch_ms": round(avg, 5), "frac": round((n + 12) * px / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "ms_per_step": round(el / K * 1e3, 5), "value": round(px / 1e6 * K / el, 1), "valid_pixels_per_scan": valid,
                           "guard_flagged_pixels": flagged, "path": path["path"], "launches_timed": kn}
            for st in stacks:
                st.free()
        out[wl] = {"workload": f"{cw}x{ch} cam, {pw}x{ph} proj, {n} frames", "algorithmic_bytes_per_px": n + 12, "scenes": rows}
        c.close()
    return out


def physical_accuracy(ctx, N, H, W, row0, rows, plane, proj_size, scene, maps, xyz, band_px, mode_fused):
    """Recovered XYZ of one fused scan of a physical scene against the generator's TRUE surface points (not against the oracle): the error
    is the method's -- half a projector pixel of code quantisation seen through the triangulation geometry.  The context's calibration must
    be the scene's rig."""
    st, truth = ctx.alloc(max(16, N * plane)), ctx.alloc(max(16, band_px * 12))
    synth_into(ctx, scene, st.ptr, plane, N, H, W, proj_size, 1, row0=row0, rows=rows, d_truth_xyz=truth.ptr)
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
    return {"scene": scene, "pixels_compared": int(ok.sum()), "sampling": "every 5th pixel", "median_error_mm": round(float(np.median(err)) * 1e3, 4),
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

    prod32 = ctx.alloc_cloud_lists(band_px, colors=True, lists=True, f32=True)

    def via_cloud32(i):
        s = stacks[i % len(stacks)]
        ctx.cloud_dev(s.ptr, 1, N * plane, plane, N, rows, W, proj_size, white.ptr, prod32, d_h=maps.at(0), d_v=maps.at(band_px * 2))

    el_dense = run(via_dense)
    el_points = run(via_cloud_points)
    el_f32 = run(via_cloud32)
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
    # The stage is bound by what the chip can WRITE, not by the 8 TB/s read-side figure: measure that ceiling in the same run -- a fill of 1 GB
    # (hipMemsetAsync: one contiguous, perfectly coalesced store stream), HIP events around three fills
    fill = ctx.alloc(1 << 30)
    ctx.dev_memset(fill.ptr, 0, 1 << 30)
    ctx.event_record(2)
    for _ in range(3):
        ctx.dev_memset(fill.ptr, 0, 1 << 30)
    ctx.event_record(3)
    fill_gbs = 3 * (1 << 30) / (ctx.event_elapsed_ms(2, 3) * 1e-3) / 1e9
    fill.free()
    write_gbs = M * 64 / (stage_ms * 1e-3) / 1e9
    out = {"value": round(band_px / 1e6 * K / el, 1), "unit": "Mpixels/s", "ms_per_scan": round(el / K * 1e3, 4), "steps": K, "points": int(M),
           "executed": {**executed, "list_kernel": list_kernel},
           "list_stage_ms": round(stage_ms, 4), "list_stage_bytes": int(stage_bytes),
           "list_stage_roofline": {"bound": "hbm-write", "achieved": round(stage_bytes / (stage_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(stage_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "bytes_written": int(M * 64), "bytes_read": int(band_px * (8 + ray_b + 3)),
                                   "write_rate": round(write_gbs, 1), "write_ceiling": round(fill_gbs, 1),
                                   "frac_of_write_ceiling": round(write_gbs / fill_gbs, 4),
                                   "note": "the stage writes %d %% of its bytes: `frac` is against the 8 TB/s spec figure like every other roofline object here; "
                                           "frac_of_write_ceiling = its store rate against a 1 GB hipMemsetAsync fill timed in the same run (what this chip "
                                           "writes with one perfectly coalesced stream; the stage writes eight interleaved streams)" % round(100.0 * M * 64 / stage_bytes)},
           "points_and_colours_only": {"value": round(band_px / 1e6 * K / el_points, 1), "ms_per_scan": round(el_points / K * 1e3, 4),
                                       "note": "slgc_cloud_dev without the two correspondence lists (intermediates of src/4-triangulate.py:62-64; the script keeps "
                                               "pts_3d and colors, :67-68): 48 instead of 64 bytes written per point"},
           "float32_product": {"value": round(band_px / 1e6 * K / el_f32, 1), "ms_per_scan": round(el_f32 / K * 1e3, 4),
                               "note": "slgc_cloud32_dev: the same lists with the points and colours as float32 -- NOT the reference's dtypes -- 40 instead of 64 "
                                       "bytes written per point; for callers that do not need float64"},
           "via_dense_xyz": {"value": round(band_px / 1e6 * K / el_dense, 1), "ms_per_scan": round(el_dense / K * 1e3, 4),
                             "note": "round 2's route: fused scan (writes 12 B/pixel of dense XYZ) + slgc_cloud_lists_dev (reads it back)"},
           "note": "slgc_cloud_dev: decode kernel + x-major list build (count, column prefix, LDS-transposed scatter that triangulates each valid "
                   "pixel in-kernel, folds in the colour gather and the float64 (3,M) points and writes whole aligned 128-byte lines); everything stays in HBM, no dense XYZ; "
                   "list_stage_ms = the list build alone, mean of %d back-to-back builds" % K}
    white.free()
    lists.free()
    prod.free()
    prod32.free()
    return out


