"""bench.py support: workloads, calibrations, small helpers (no GPU work here)."""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))

WORKLOADS = {
    # name: (cam_w, cam_h, proj_w, proj_h, N)
    "c3_4096x3000x44": (4096, 3000, 1920, 1200, 44),
    "c2_1920x1080x44": (1920, 1080, 1920, 1080, 44),
    "c3_4096x3000x46": (4096, 3000, 1920, 1200, 46),
    "c1_1280x720x42": (1280, 720, 1280, 800, 42),
    "c2_1920x1080x46": (1920, 1080, 1920, 1080, 46),
    # a 4K / an 8K projector: the frame counts the reference's generator emits for them (generate_codes.py:22-25,53), specialised kernels since round 5
    "c3_4096x3000x50": (4096, 3000, 3840, 2160, 50),
    "c3_4096x3000x54": (4096, 3000, 7680, 4320, 54),
    # small test workloads (tests/test_gpu_rccl_multi.py): an odd height (ragged bands at any G) and an even one
    "t_516x1031x44": (516, 1031, 300, 200, 44),
    "t_512x1024x44": (512, 1024, 300, 200, 44),
    # one rank's band of the headline image at 2 / 4 / 8 ranks (timing the band kernels on one GPU: tools/ab_fused.py --workload ...)
    "b2_4096x1500x44": (4096, 1500, 1920, 1200, 44),
    "b4_4096x750x44": (4096, 750, 1920, 1200, 44),
    "b8_4096x375x44": (4096, 375, 1920, 1200, 44),
}
BASELINE_CONFIG = {"c1_1280x720x42": 0, "c2_1920x1080x44": 1, "c3_4096x3000x44": 2}      # --workload -> index into BASELINE.json "configs"


def workload_label(name, cam_w, cam_h, proj_w, proj_h, N, sharded_over=0, exchange="RCCL"):
    idx = BASELINE_CONFIG.get(name)
    if idx is None:
        tag = "not a BASELINE.json config: a variant / test / band workload"
    elif sharded_over and idx == 2:
        tag = f"BASELINE.json configs[2] row-sharded over {sharded_over} GPU(s) + {exchange} exchange = configs[3]"
    elif sharded_over:
        tag = f"BASELINE.json configs[{idx}] row-sharded over {sharded_over} GPU(s) + {exchange} exchange"
    else:
        tag = f"BASELINE.json configs[{idx}]"
    return f"{cam_w}x{cam_h} cam, {proj_w}x{proj_h} proj, {N} uint8 frames ({tag})"


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PREHEAT_S = 0.15       # untimed back-to-back scans before the counted warm-up: the clocks of a fresh box ramp for ~100 ms


def calibration(cam_w, cam_h, proj_w, proj_h, rig="survey"):
    """rig "survey" = SURVEY.md 8(d): repo intrinsics; fx=fy=3000 for the 4096x3000 camera; synthetic extrinsics.  Its projector (fx ~ 2650 px at
    1920, principal point (543, 272), a lens model that folds over at r^2 ~ 0.22) reaches 9-28 % of the camera image.
    rig "covering" = the same camera and the same stereo pose with a short-throw projector whose ADDRESSABLE raster -- 2^L x 2^L pixels, L = 10
    code bits at 42 / 44 frames -- covers the camera's field of view (f = 600 px, principal point at the raster centre, a mild lens): > 90 % of
    the camera pixels are lit by one physically consistent surface, which is what the reference's scanner is built to see."""
    from scanner import reference_calibration as rc
    K = rc.CAM_MTX.copy()
    if cam_w > 1920:
        K[0, 0] = K[1, 1] = 3000.0
        K[0, 2], K[1, 2] = cam_w / 2.0, cam_h / 2.0
    th = np.deg2rad(-20.0)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T = np.array([[0.25], [0.02], [0.04]])
    if rig == "covering":
        pk = np.array([[600.0, 0.0, 511.5], [0.0, 600.0, 511.5], [0.0, 0.0, 1.0]])
        return K, rc.CAM_DIST, pk, COVERING_PROJ_DIST.copy(), R, T
    if rig != "survey":
        raise ValueError(f"unknown rig {rig!r}")
    pk = rc.PROJ_MTX.copy()
    pk[0, :] *= proj_w / 1920.0          # Triangulate.__init__ scaling, triangulate.py:28-33
    pk[1, :] *= proj_h / 1080.0
    return K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T


COVERING_PROJ_DIST = np.array([0.04, -0.06, -0.0005, 0.0008, 0.02])      # k1 k2 p1 p2 k3: monotonic far beyond the camera's field of view

# Synthetic captures (all generated on the device, each with a bit-identical NumPy twin in oracle/oracle_np.py).
SCENES = {
    "physical": {"rig": "covering", "kind": "physical", "noise": 3, "gains": (140, 180), "r2_max": 1.5,
                 "label": "physical (plane + sphere seen by camera and projector through the covering rig: one consistent surface lights > 90 % of "
                          "the camera pixels; noise 3, surface gains 140 / 180)"},
    "noisy-physical": {"rig": "covering", "kind": "physical", "noise": 12, "gains": (46, 60), "r2_max": 1.5,
                       "label": "noisy physical (the same surface, dim and noisy: surface gains 46 / 60, noise 12 -- several per cent of the decoded "
                                "pixels carry a wrong code, i.e. (h, v) pairs no surface produces)"},
    "physical-survey": {"rig": "survey", "kind": "physical", "noise": 3, "gains": None, "r2_max": None,
                        "label": "physical on SURVEY.md 8(d)'s calibration (the reference's narrow projector reaches 9-28 % of the camera image; noise 3)"},
    "s-scene": {"rig": "survey", "kind": "s-scene", "noise": 3,
                "label": "S-scene (SURVEY.md 8(d): smooth synthetic code maps, shadow rectangle, noise 3; ~80 % decodable, epipolar-inconsistent)"},
    "s-uniform": {"rig": "survey", "kind": "uniform",
                  "label": "S-uniform (SURVEY.md 8(d): every byte of every frame uniform in 0..255 -- ~21 % of the pixels decode, to arbitrary codes; the worst case)"},
}


def synth_into(c, scene, d_ptr, plane, n, H, W, proj_size, seed, row0=0, rows=None, d_truth_xyz=None):
    """One synthetic capture of SCENES[scene] into HBM.  The physical kinds need the context's calibration to be the scene's rig (the caller's job)."""
    cfg = SCENES[scene]
    if cfg["kind"] == "physical":
        c.synth_physical_dev(d_ptr, plane, n, H, W, proj_size, row0=row0, rows=rows, seed=seed, noise=cfg["noise"], gains=cfg["gains"],
                             r2_max=cfg["r2_max"], d_truth_xyz=d_truth_xyz)
    elif cfg["kind"] == "uniform":
        c.synth_uniform_dev(d_ptr, plane, n, H, W, row0=row0, rows=rows, seed=seed)
    else:
        c.synth_scene_dev(d_ptr, plane, n, H, W, row0=row0, rows=rows, seed=seed, noise=cfg["noise"], shadow=True)


# ------------------------------------------------------------------------------------------------ helpers
def launch_stats(samples_ms):
    s = np.sort(np.asarray(samples_ms, dtype=np.float64))
    if s.size == 0:
        return {}
    q = lambda f: float(s[min(s.size - 1, int(round(f * (s.size - 1))))])     # noqa: E731
    med = float(np.median(s))
    return {"min_launch_ms": round(float(s[0]), 5), "median_launch_ms": round(med, 5), "p95_launch_ms": round(q(0.95), 5),
            "max_launch_ms": round(float(s[-1]), 5), "outliers": int((s > 2.0 * med).sum()), "outliers_rule": "launches above 2 x median"}


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


STAGE = ["start"]          # where the run is (printed by the watchdogs: a hang names its stage)
