"""bench.py: the CPU baseline leg (the only place of the bench that imports oracle/)."""
import os
import sys
import time

import numpy as np

from .common import ROOT, WORKLOADS, calibration


# ------------------------------------------------------------------------------------------------ CPU baseline (N = 1)
def _port_scan(onp, name, rows=None):
    """One decode + triangulate of a BASELINE configuration (rows = only its first `rows` image rows: a bounded sample, the cost is per pixel)
    through the reference-cost port; -> (seconds, seconds get_codes, seconds loops, points)"""
    cw, ch, pw, ph, n = WORKLOADS[name]
    K, cd, pk, pd, R, T = calibration(cw, ch, pw, ph)
    st, _, _ = onp.synth_scene_int(n, ch, cw, seed=1, rows=rows)
    ch = st.shape[1]
    white = np.repeat(st[1][:, :, None], 3, axis=2)
    t0 = time.perf_counter()
    hc, vc = onp.get_codes_loops(st.astype(np.float64))                # float64 stack like the reference driver (src/3:68-70)
    t_codes = time.perf_counter() - t0
    hp, vp = onp.codes_to_pixels_loops(hc, vc)
    t_pix = time.perf_counter() - t0 - t_codes
    cam, proj, _ = onp.cam_proj_pts_loops(hp, vp, (cw, ch), (pw, ph), white)
    pts = onp.triangulate(cam, proj, K, cd, pk, pd, R, T)
    return time.perf_counter() - t0, t_codes, t_pix, pts.shape[1]


def cpu_baseline(level="full"):
    """the reference-cost CPU path on BASELINE configs[0] (1280x720 camera, 1280x800 projector, 42 frames): get_codes with the
    reference's cost shape (fancy-index copies, np.repeat, ten np.where scatters), the per-pixel Python loops of src/3-capture_decode.py:99-100
    and triangulate.py:52-64, then the NumPy law of sines -- one thread, like the reference.
    level "lite" (the default run): configs[0] at FULL size (1280x720x42: 5-10 s of one core), the C oracle on a 1024x512 crop, threads swept
    over {8, 32, all}.  level "full": configs[0] AND configs[1] at full size, the C oracle on 2048x1024, the whole thread sweep."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    import oracle_np as onp
    lite = level != "full"
    with_c2 = not lite
    cw, ch, pw, ph, n = WORKLOADS["c1_1280x720x42"]
    dt, t_codes, t_pix, npts = _port_scan(onp, "c1_1280x720x42")
    mpix = cw * ch / 1e6
    c2 = None
    if with_c2:
        d2, c2_codes, c2_pix, c2_pts = _port_scan(onp, "c2_1920x1080x44")
        c2 = {"value": round(1920 * 1080 / 1e6 / d2, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
              "sample": f"BASELINE configs[1] at full size: 1920x1080 camera and projector, 44 frames, S-scene, decode + triangulate, {d2:.1f} s "
                        f"({c2_codes:.1f} s get_codes, {c2_pix:.1f} s gray_to_decimal loops), {c2_pts} points; same port, 1 thread"}
    # strong baseline: the plain-C oracle on the headline workload's own size class (a 2048x1024 crop of the 44-frame scene)
    c_w, c_h = (1024, 512) if lite else (2048, 1024)
    st2, _, _ = onp.synth_scene_int(44, c_h, c_w, seed=1)
    cal3 = calibration(4096, 3000, 1920, 1200)
    t1 = time.perf_counter()
    oc.scan_dense(st2, (1920, 1200), *cal3)
    dt_c = time.perf_counter() - t1
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    dt_mt, used, sweep = None, 1, {}
    for nthr in sorted({min(cores, c) for c in ((8, 32, cores) if lite else (8, 16, 32, 64, 128, cores))}):     # a container's CPU quota can be far below its visible cores
        oc.set_threads(nthr)
        oc.scan_dense(st2, (1920, 1200), *cal3)                        # thread pool start-up
        t2 = time.perf_counter()
        oc.scan_dense(st2, (1920, 1200), *cal3)
        d = time.perf_counter() - t2
        sweep[str(nthr)] = round(c_w * c_h / 1e6 / d, 2)
        if dt_mt is None or d < dt_mt:
            dt_mt, used = d, nthr
    oc.set_threads(1)
    mpix2 = c_w * c_h / 1e6
    size = f"configs[0] full size ({cw}x{ch}x{n})"
    out = {"value": round(mpix / dt, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port", "level": level,
           "sample": f"{size}, decode+triangulate, {dt:.1f} s, {npts} points; NumPy/Python port, reference's cost shape, 1 thread",
           "sample_detail": f"BASELINE configs[0]: {cw}x{ch} camera rows, {pw}x{ph} projector, {n} frames, synthetic scene, decode + "
                            f"triangulate, {dt:.1f} s ({t_codes:.1f} s get_codes, {t_pix:.1f} s gray_to_decimal loops), {npts} points; "
                            "NumPy/Python port with the reference's cost shape (oracle/oracle_np.py *_loops), 1 thread like the reference",
           "reference_measured": {"value": 0.046, "unit": "Mpixels/s", "note": "the reference itself, end to end at 1920x1080x44 in the "
                                  "survey container (BASELINE.md section 2); it cannot travel to the GPU box"},
           # measured in the build container (8 vCPU Xeon 2.1 GHz), reference imported from /root/reference, same stacks, results identical:
           # 1280x720x42 reference 0.071 / port 0.089 Mpixels/s; 1920x1080x44 reference 0.068 / port 0.083; get_codes alone, alternated in
           # one process: 3.6-4.3 s for both at 1920x1080 (notes/r06.md)
           "why_port_differs": "same cost shape as the reference: build container, same stack, reference 0.068-0.071 vs port 0.083-0.089 Mpix/s "
                               "(get_codes equal when alternated); the rest over the survey's 0.046 is this host's faster core and the survey's cold first calls",
           "c_oracle_value": round(mpix2 / dt_c, 3), "c_oracle_note": f"plain-C scalar oracle (oracle/slgc_oracle.c), 1 thread, {c_w}x{c_h}x44 crop of the headline scene",
           "c_oracle_all_cores_value": round(mpix2 / dt_mt, 3), "c_oracle_all_cores": used, "host_cores_visible": cores,
           "c_oracle_threads_sweep_mpix_s": sweep,
           "c_oracle_all_cores_note": "the same C oracle, per-pixel loops on host threads (OpenMP); the figure is the BEST of the sweep over 8/(16)/32/(64/128)/all "
                                      "visible cores (every point of the sweep is listed): the box shows the host's cores but the job's CPU quota and the "
                                      "2 MB crop's memory traffic stop the scaling well below them, so more threads than the best point are slower, not faster"}
    if c2:
        out["c2"] = c2
    return out
