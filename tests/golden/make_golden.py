#!/usr/bin/env python3
"""Generate the golden vectors in this directory by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; the GPU box has neither the
reference nor this need -- it only reads the committed .npz files).  The reference is pure
Python; its decode half imports once a dummy ``cv2`` module is registered (the hot functions
never call cv2 -- scanner/grayCode/decode_codes.py:2 is the only obstacle).  For
``Triangulate.triangulate`` (scanner/triangulation/triangulate.py:84-85) the two cv2 calls
are served by the oracle's restatement of OpenCV's algorithm, so the reference's OWN
law-of-sines code (:86-95) produces the stored XYZ; the undistortPoints half stays unpinned
(OpenCV 4.8.0.76 is not installed here) and the fixture says so.

The four driver lines of src/3-capture_decode.py:95-100 (a script body that opens a camera
on import) are restated around the imported functions, as SURVEY.md Appendix A describes.

Usage:  python tests/golden/make_golden.py      (writes tests/golden/*.npz)
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_np as onp  # noqa: E402  (only for the cv2 stand-ins and synthetic inputs)

cv2_stub = types.ModuleType("cv2")
cv2_stub.undistortPoints = lambda src, K, dist, R=None: onp.undistort_points(src, K, dist, R=R)
cv2_stub.convertPointsToHomogeneous = lambda src: onp.to_homogeneous(src)
cv2_stub.absdiff = lambda a, b: np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))   # float64 frames: exact |a-b|
sys.modules["cv2"] = cv2_stub
sys.path.insert(0, "/root/reference")
from scanner.grayCode import decode_codes as ref_dc  # noqa: E402
from scanner.grayCode import generate_codes as ref_gc  # noqa: E402
from scanner.triangulation import Triangulate as RefTriangulate  # noqa: E402

warnings.filterwarnings("ignore", category=RuntimeWarning)
CAL = "/root/reference/data/calib_results"


def ref_pixels(h_codes, v_codes):
    """src/3-capture_decode.py:99-100, verbatim semantics around the imported function."""
    L, H, W = h_codes.shape
    hp = np.array([ref_dc.gray_to_decimal(h_codes[:, y, x]) for y in range(H) for x in range(W)]).reshape(H, W)
    vp = np.array([ref_dc.gray_to_decimal(np.flip(v_codes[:, y, x])) for y in range(H) for x in range(W)]).reshape(H, W)
    return hp, vp


def adversarial(N, H, W, seed):
    """NaN pixels (white+black=0), saturated, all-equal frames, and the tie family
    D*w/(w+b) integer (e.g. black=2, white=1, D=3) -- SURVEY.md section 8(c)(iii)."""
    rng = np.random.default_rng(seed)
    st = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
    q = W // 4
    st[0, :, :q] = 0
    st[1, :, :q] = 0                                   # 0/0 -> NaN
    st[0, :, q:2 * q] = 0
    st[1, :, q:2 * q] = 255                            # b_inv = 1
    st[:, : H // 3, 2 * q:3 * q] = st[5, : H // 3, 2 * q:3 * q]          # all frames equal
    st[0, :, 3 * q:] = 2
    st[1, :, 3 * q:] = 1                               # b_inv = 1/3, ties 3*fl(1/3)
    st[2:, H // 2:, 3 * q:] = (rng.integers(0, 86, (N - 2, H - H // 2, W - 3 * q)) * 3).astype(np.uint8)
    # near-threshold pairs: |n - i| in {0,1,2} around eps=1
    half = (N - 2) // 4 * 2
    base = rng.integers(0, 254, (half, H // 3, q))
    st[2:2 + half, -(H // 3):, :q] = base
    st[2 + half:2 + 2 * half, -(H // 3):, :q] = base + rng.integers(0, 3, base.shape)
    st[0, -(H // 3):, :q] = 10
    st[1, -(H // 3):, :q] = 240
    return st


def decode_case(stack_u8, second_run_u8=None):
    st = stack_u8.astype(np.float64)                   # src/3-capture_decode.py:69-70 (float64 stack)
    L_d, L_g = ref_dc.get_direct_indirect(st)
    h_codes, v_codes = ref_dc.get_codes(st)
    hp, vp = ref_pixels(h_codes, v_codes)
    out = dict(stack=stack_u8, L_d=L_d, L_g=L_g, h_codes=h_codes, v_codes=v_codes,
               h_pixels=hp.astype(np.int64), v_pixels=vp.astype(np.int64))
    if second_run_u8 is not None:
        h2, v2 = ref_dc.get_codes(second_run_u8.astype(np.float64))
        bh = np.max([h_codes, h2], axis=0)             # src/3-capture_decode.py:95-96
        bv = np.max([v_codes, v2], axis=0)
        mh, mv = ref_pixels(bh, bv)
        out.update(stack2=second_run_u8, merged_h_pixels=mh.astype(np.int64), merged_v_pixels=mv.astype(np.int64))
    return out


def multirun():
    """src/3-capture_decode.py:48 (MAX_NB_RUNS), :75-100: get_codes per run, np.max over ALL runs' codes, then the two per-pixel loops -- with 4 and
    8 runs (the reference's script ships with 2; the merge is an N-way max).  Runs differ in noise, and every run but one has a block of
    pixels blacked out, so that the merged maps differ from every single run's.  -> tests/golden/multirun.npz (written on its own:
    `python tests/golden/make_golden.py multirun`)."""
    flat = {}
    for N, H, W, R in ((26, 20, 36, 4), (44, 16, 32, 8), (14, 9, 13, 8)):
        runs = np.stack([onp.synth_scene(N, H, W, seed=60 + r, noise=2 + r) for r in range(R)])
        for r in range(R):
            if r != 1:
                runs[r, :, :, (3 * r) % W:(3 * r) % W + 3] = 0
        codes = [ref_dc.get_codes(run.astype(np.float64)) for run in runs]          # :75 per run
        bh = np.max([c[0] for c in codes], axis=0)                                   # :95
        bv = np.max([c[1] for c in codes], axis=0)                                   # :96
        mh, mv = ref_pixels(bh, bv)                                                  # :99-100
        single = ref_pixels(*codes[0])
        assert not np.array_equal(single[0], mh)
        tag = f"runs{R}_N{N}_{H}x{W}"
        flat[f"{tag}/stacks"] = runs
        flat[f"{tag}/merged_h_codes"] = bh
        flat[f"{tag}/merged_v_codes"] = bv
        flat[f"{tag}/merged_h_pixels"] = mh.astype(np.int64)
        flat[f"{tag}/merged_v_pixels"] = mv.astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "multirun.npz"), **flat)
    print("multi-run cases:", len(flat) // 5)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "multirun":
        return multirun()
    multirun()
    # ------------------------------------------------------------------ decode cases
    cases = {}
    for N in (14, 15, 17, 26, 42, 44, 46):
        for (H, W) in ((24, 40), (17, 29)):
            if N in (15, 17) and (H, W) != (17, 29):
                continue
            tag = f"N{N}_{H}x{W}"
            rnd = np.random.default_rng(0).integers(0, 256, (N, H, W), dtype=np.uint8)
            rnd2 = np.random.default_rng(7).integers(0, 256, (N, H, W), dtype=np.uint8)
            cases[f"uniform_{tag}"] = decode_case(rnd, rnd2)
            sc = onp.synth_scene(N, H, W, seed=1)
            sc2 = onp.synth_scene(N, H, W, seed=2, noise=9)
            cases[f"scene_{tag}"] = decode_case(sc, sc2)
            cases[f"adversarial_{tag}"] = decode_case(adversarial(N, H, W, 3))
    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "decode_cases.npz"), **flat)
    print("decode cases:", len(cases))

    # ------------------------------------------------------------------ known-answer rule table
    # (d, g, n, i) -> code, fed to the reference's get_is_lit as 1x1 images (N=14, bit 0 of h).
    quads = [(100, 20, 150, 30), (100, 20, 30, 150), (100, 20, 60, 60), (5, 2, 150, 30), (20, 50, 150, 30),
             (20, 50, 10, 100), (20, 50, 100, 10), (50, 20, 30, 40), (100, 20, 21, 99), (100, 20, 22, 98),
             (50, 50, 200, 10), (np.nan, np.nan, 200, 10), (100, 20, 61, 60), (100, 20, 62, 60), (100, 20, 60, 62),
             (30.5, 29.5, 29, 31), (30.5, 29.49, 29, 31), (0, 0, 0, 0), (255, 0, 255, 0), (1e9, -1e9, 3, 3)]
    rng = np.random.default_rng(11)
    for _ in range(400):
        quads.append((float(rng.uniform(-20, 300)), float(rng.uniform(-20, 300)),
                      float(rng.integers(0, 256)), float(rng.integers(0, 256))))
    kat = []
    for eps in (1, 0, 2, 0.5, 1.0 - 2.0 ** -60, 3.75):
        for (d, g, n, i) in quads:
            st = np.zeros((14, 1, 1))
            st[2, 0, 0] = n                            # normal h bit 0
            st[2 + 6, 0, 0] = i                        # inverse h bit 0  (L=3 -> 2L=6)
            hc, _ = ref_dc.get_is_lit(st, np.array([[d]], dtype=np.float64), np.array([[g]], dtype=np.float64), eps=eps)
            kat.append((eps, d, g, n, i, int(hc[0, 0, 0])))
    np.savez_compressed(os.path.join(HERE, "rule_kat.npz"), kat=np.array(kat, dtype=np.float64))
    print("rule KAT rows:", len(kat))

    # ------------------------------------------------------------------ gray helpers
    np.savez_compressed(
        os.path.join(HERE, "gray_kat.npz"),
        gray_decode=np.array([ref_dc.gray_decode(n) for n in range(4096)], dtype=np.int64),
        g2d_in=np.array([[1, 0, 1, 1], [0, 0, 0, 0], [1, 1, 1, 1], [1, -1, 0, 1], [0, 1, 1, 0]], dtype=np.int8),
        g2d_out=np.array([ref_dc.gray_to_decimal(s) for s in
                          ([1, 0, 1, 1], [0, 0, 0, 0], [1, 1, 1, 1], [1, -1, 0, 1], [0, 1, 1, 0])], dtype=np.int64))

    # ------------------------------------------------------------------ pattern generator + identity scan
    gen = {}
    for (w, h) in ((64, 32), (40, 24), (16, 16)):
        codes = ref_gc.get_gray_codes(w, h)
        seq = ref_gc.get_image_sequence(codes, w, h)
        gen[f"codes_{w}x{h}"] = codes
        gen[f"seq_{w}x{h}"] = seq
        cap = (20 + 0.7 * seq.astype(np.float64)).astype(np.uint8)        # camera == projector, gain .7, ambient 20
        hc, vc = ref_dc.get_codes(cap.astype(np.float64))
        hp, vp = ref_pixels(hc, vc)
        gen[f"ident_h_{w}x{h}"] = hp.astype(np.int64)
        gen[f"ident_v_{w}x{h}"] = vp.astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "generator.npz"), **gen)

    # ------------------------------------------------------------------ calibration data held by the reference
    cal = dict(cam_mtx=np.load(f"{CAL}/cam_1080/cam_mtx.npy"), cam_dist=np.load(f"{CAL}/cam_1080/cam_dist.npy"),
               cam1440_mtx=np.load(f"{CAL}/cam_1440/cam_mtx.npy"), cam1440_dist=np.load(f"{CAL}/cam_1440/cam_dist.npy"),
               proj_mtx=np.load(f"{CAL}/proj/proj_mtx.npy"), proj_dist=np.load(f"{CAL}/proj/proj_dist.npy"))
    np.savez_compressed(os.path.join(HERE, "calib.npz"), **cal)

    # ------------------------------------------------------------------ Triangulate (a6-a9)
    th = np.deg2rad(-20.0)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T = np.array([[0.25], [0.02], [0.04]])
    tri = {}
    for name, (cw, ch), (pw, ph), (cpw, cph) in (("a", (64, 48), (1280, 800), (1920, 1080)),
                                                 ("b", (57, 33), (1920, 1080), (1920, 1080))):
        rng = np.random.default_rng(5)
        yy, xx = np.mgrid[0:ch, 0:cw]
        hmap = (xx * (pw + 40) // cw).astype(np.int64)          # some codes exceed the projector -> clamp (:60-61)
        vmap = (yy * (ph + 25) // ch).astype(np.int64)
        holes = rng.random((ch, cw)) < 0.25
        hmap[holes] = -1
        vmap[rng.random((ch, cw)) < 0.1] = -1
        white = rng.integers(0, 256, (ch, cw, 3), dtype=np.uint8)
        # place the tiny camera window around the real principal point so distortion is exercised
        cam_mtx = cal["cam_mtx"].copy()
        cam_mtx[0, 2], cam_mtx[1, 2] = cw / 2.0 + 0.3, ch / 2.0 - 0.2
        cam_mtx[0, 0] = cam_mtx[1, 1] = 45.0
        proj_mtx_in = cal["proj_mtx"].copy()
        t = RefTriangulate(hmap, vmap, (cw, ch), cam_mtx, cal["cam_dist"], (pw, ph), (cpw, cph),
                           proj_mtx_in, cal["proj_dist"], R, T, None)
        cam_pts, proj_pts, colors = t.get_cam_proj_pts(white)
        pts = t.triangulate(cam_pts, proj_pts)
        fp, fc = t.filter_3d_pts(pts, colors, threshold=0.5)
        tri.update({f"{name}/h": hmap, f"{name}/v": vmap, f"{name}/white": white, f"{name}/cam_size": np.array([cw, ch]),
                    f"{name}/proj_size": np.array([pw, ph]), f"{name}/proj_calib_size": np.array([cpw, cph]),
                    f"{name}/cam_mtx": cam_mtx, f"{name}/cam_dist": cal["cam_dist"], f"{name}/proj_mtx": cal["proj_mtx"],
                    f"{name}/proj_mtx_scaled": proj_mtx_in, f"{name}/proj_dist": cal["proj_dist"], f"{name}/R": R,
                    f"{name}/T": T, f"{name}/cam_pts": cam_pts, f"{name}/proj_pts": proj_pts, f"{name}/colors": colors,
                    f"{name}/pts": pts, f"{name}/filt_pts": fp, f"{name}/filt_colors": fc})
        print("triangulate", name, "M =", cam_pts.shape[0], "kept", fp.shape[1],
              "nan:", int(np.isnan(pts).any(axis=0).sum()))
    np.savez_compressed(os.path.join(HERE, "triangulate.npz"), **tri)
    # ------------------------------------------------------------------ remove_bad_images (decode_codes.py:34-68)
    ing = {}
    rng = np.random.default_rng(17)
    for name, n in (("a", 24), ("b", 31), ("c", 9)):
        # a capture: each projected pattern held for 2-3 frames with 0-1 blended transition frames in between
        pats = rng.integers(0, 256, (n, 6, 8, 3)).astype(np.float64)
        frames = []
        for k in range(n):
            if k and rng.random() < 0.6:
                a = rng.uniform(0.2, 0.8)
                frames.append(np.floor(a * pats[k - 1] + (1 - a) * pats[k]))
            frames.extend([pats[k] + rng.integers(-2, 3, pats[k].shape)] * int(rng.integers(1, 4)))
        seq = np.array(frames)
        ing[f"{name}/frames"] = seq
        ing[f"{name}/kept"] = np.array(ref_dc.remove_bad_images(seq), dtype=np.int64)
        print("remove_bad_images", name, len(seq), "->", len(ing[f"{name}/kept"]))
    np.savez_compressed(os.path.join(HERE, "ingest.npz"), **ing)

    # ------------------------------------------------------------------ config 1 at full size: 1280x720 camera, 1280x800 projector, 42 frames
    # (BASELINE.json configs[0]; the repo's own capture is not in the repo -- SURVEY.md D4 -- so the scene is synthetic, the
    # intrinsics are the repo's).  The reference itself decodes and triangulates it; only outputs are stored (the input is
    # regenerated from its seed and guarded by a SHA-256).
    import hashlib
    N1, H1, W1 = 42, 720, 1280
    stack1, _, _ = onp.synth_scene_int(N1, H1, W1, seed=5, noise=3, shadow=True)
    hc, vc = ref_dc.get_codes(stack1.astype(np.float64))
    hp1, vp1 = ref_pixels(hc, vc)
    white1 = np.repeat(stack1[1][:, :, None], 3, axis=2)
    th = np.deg2rad(-20.0)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    T = np.array([[0.25], [0.02], [0.04]])
    pm1 = cal["proj_mtx"].copy()
    t1 = RefTriangulate(hp1, vp1, (W1, H1), cal["cam_mtx"], cal["cam_dist"], (1280, 800), (1920, 1080), pm1, cal["proj_dist"], R, T, None)
    cam1, proj1, col1 = t1.get_cam_proj_pts(white1)
    pts1 = t1.triangulate(cam1, proj1)
    fp1, fc1 = t1.filter_3d_pts(pts1, col1, threshold=0.5)
    sel = np.arange(0, pts1.shape[1], 97)
    np.savez_compressed(os.path.join(HERE, "config1.npz"), params=np.array([N1, H1, W1, 5, 3]),
                        stack_sha256=np.frombuffer(hashlib.sha256(stack1.tobytes()).digest(), dtype=np.uint8),
                        h_pixels=hp1.astype(np.int16), v_pixels=vp1.astype(np.int16), M=np.array([cam1.shape[0], fp1.shape[1]]),
                        cam_sha256=np.frombuffer(hashlib.sha256(cam1.tobytes()).digest(), dtype=np.uint8),
                        proj_sha256=np.frombuffer(hashlib.sha256(proj1.tobytes()).digest(), dtype=np.uint8),
                        sample_index=sel, pts_sample=pts1[:, sel], colors_sample=col1[sel], R=R, T=T)
    print("config1: M =", cam1.shape[0], "kept", fp1.shape[1])

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
