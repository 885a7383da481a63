"""One chained, device-resident pass over everything either side of the hot path (-m gpu): SURVEY.md 8(f) rows 1-3 around rows a0-a9.

    BGR capture in HBM (two runs, every pattern shown three times with a blended transition frame in between)
      -> slgc_frame_diff_counts_dev + the keep / drop rule          /root/reference/scanner/grayCode/decode_codes.py:34-68
      -> slgc_to_gray_dev on the kept frames, straight into the stack   /root/reference/src/3-capture_decode.py:66-70
      -> slgc_cloud_dev: decode both runs, merge, maps, x-major lists, triangulation, colours   src/3:75-100, src/4-triangulate.py:50-64
      -> filter_3d_pts(0.5) -> statistical outlier removal (k-NN on the GPU) -> cloud.ply       src/4:71, scanner/utils/visualize.py:98-113

against the oracle chain on the CPU.  Integer products are compared bit for bit END TO END (kept frames, grey stack, maps, correspondence
lists, colours); the float64 points within 1e-4 of the oracle's triangulation; and every stage after the points (box filter, inlier
indices, PLY bytes) bit for bit against the oracle applied to the SAME points (a point 1e-7 off the oracle's may legitimately fall on the
other side of a threshold), with the end-to-end inlier sets required to agree to 99.9 %."""
import numpy as np
import pytest

import oracle_c as oc
import oracle_np as onp
from conftest import has_gpu
from test_gpu_physical import matched_rig

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]


def capture_bgr(gray_stack, rng):
    """[N,H,W] grey patterns -> a BGR capture [3N + N - 1, H, W, 3]: every pattern three times (sensor noise +-2 between the repeats),
    a half-and-half blend while the projector switches; channel offsets so that the luma conversion has something to do."""
    N, H, W = gray_stack.shape
    yy, xx = np.mgrid[0:H, 0:W]
    dB, dR = ((xx * 7 + yy * 3) % 11) - 5, ((xx * 5 + yy * 11) % 9) - 4
    base = np.stack([np.clip(gray_stack.astype(np.int64) + dB, 0, 255), gray_stack.astype(np.int64), np.clip(gray_stack.astype(np.int64) - dR, 0, 255)], axis=-1)
    frames = []
    for k in range(N):
        for _ in range(3):
            frames.append(np.clip(base[k] + rng.integers(-2, 3, base[k].shape), 0, 255))
        if k + 1 < N:
            frames.append((base[k] + base[k + 1]) // 2)
    return np.asarray(frames, dtype=np.uint8)


def test_capture_to_ply_chained_on_the_device(tmp_path):
    from scanner import _native, pointcloud as pc
    from scanner.grayCode import decode_codes as dc
    rng = np.random.default_rng(77)
    W, H, pw, ph, N = 512, 384, 512, 384, 46
    calib = matched_rig(W, H, pw, ph)
    px = W * H
    runs_bgr = [capture_bgr(onp.synth_physical(N, H, W, (pw, ph), calib, seed=21 + r, noise=3 + 3 * r)[0], rng) for r in range(2)]
    F = len(runs_bgr[0])
    ctx = _native.Context(0)
    ctx.set_calibration(*calib)
    d_stack = ctx.alloc(2 * N * px)
    kept_runs = []
    for r, bgr in enumerate(runs_bgr):
        d_bgr = ctx.alloc(bgr.nbytes).upload(bgr)
        d_counts = ctx.alloc(8 * F)
        ctx.frame_diff_counts_dev(d_bgr.ptr, F, px * 3, 50, d_counts.ptr)
        ctx.synchronize()
        counts = d_counts.download((F - 1,), np.uint64)
        assert np.array_equal(counts.astype(np.int64), onp.frame_diff_counts(bgr, 50))
        kept = dc.keep_from_counts(counts)
        assert kept == onp.remove_bad_images(bgr) and len(kept) == N                  # one settled frame per pattern, in order
        assert all(k // 4 == j for j, k in enumerate(kept))                            # (frame 4 j .. 4 j + 2 show pattern j)
        for j, k in enumerate(kept):                                                   # grey conversion of the kept frames, straight into the stack
            lib = _native.lib()
            ctx._ck(lib.slgc_to_gray_dev(ctx._h, d_bgr.at(k * px * 3), px, 15, d_stack.at((r * N + j) * px)))
        kept_runs.append(kept)
        ctx.synchronize()
        d_bgr.free()
        d_counts.free()
    # ---- oracle chain, CPU
    gray_ref = np.stack([onp.bgr_to_gray(bgr[kept]) for bgr, kept in zip(runs_bgr, kept_runs)])          # [2, N, H, W] uint8
    assert np.array_equal(d_stack.download((2, N, H, W), np.uint8), gray_ref)
    white_rgb = np.ascontiguousarray(runs_bgr[0][kept_runs[0][1]][:, :, ::-1])                             # frame_1, BGR -> RGB (src/4:27-30)
    ref_h, ref_v, ref_xyz = oc.scan_dense(gray_ref, (pw, ph), *calib)
    rcam, rproj, rcol = oc.cam_proj_pts(ref_h, ref_v, (W, H), (pw, ph), white_rgb, order="x")
    rpts = oc.triangulate(rcam, rproj, *calib)
    # ---- device chain
    d_white = ctx.alloc(px * 3).upload(white_rgb)
    maps = ctx.alloc(px * 4)
    lists = ctx.alloc_cloud_lists(px, colors=True)
    ctx.cloud_dev(d_stack.ptr, 2, N * px, px, N, H, W, (pw, ph), d_white.ptr, lists, d_h=maps.at(0), d_v=maps.at(px * 2))
    cam, proj, pts, col = lists.download()
    assert ctx.last_scan_path()["path"] == "cloud"
    assert np.array_equal(maps.download((H, W), np.int16), ref_h) and np.array_equal(maps.download((H, W), np.int16, px * 2), ref_v)
    assert len(rcam) > 0.8 * px                                                        # the matched rig lights most of the image
    assert np.array_equal(cam, rcam) and np.array_equal(proj, rproj) and np.array_equal(col, rcol)
    assert pts.shape == rpts.shape and np.isfinite(pts).all()
    err = np.abs(pts - rpts) / np.maximum(np.abs(rpts), 1e-300)
    assert float(err.max()) <= 1e-4
    # ---- filter -> outlier removal -> PLY: each stage bit for bit against the oracle applied to the same input
    fpts, fcol = ctx.filter_3d_pts(pts, col, 0.5)
    opts, ocol = onp.filter_3d_pts(pts, col, 0.5)
    assert np.array_equal(fpts, opts) and np.array_equal(fcol, ocol) and 0.2 * pts.shape[1] < fpts.shape[1] < pts.shape[1]    # the box really cuts
    inl, icol, ind = pc.save_point_cloud(fpts, fcol, str(tmp_path), ctx=ctx)
    oind = onp.remove_statistical_outlier(fpts.T.astype(np.float32), 20, 0.5)
    assert np.array_equal(ind, oind) and 0.5 * fpts.shape[1] < len(ind) < fpts.shape[1]
    want = np.empty(len(oind), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1"), ("g", "u1"), ("b", "u1")]))
    p32 = fpts.T.astype(np.float32)[oind].astype(np.float64)                          # Open3D holds float32 coordinates (visualize.py:98)
    want["x"], want["y"], want["z"] = p32[:, 0], p32[:, 1], p32[:, 2]
    c8 = np.round(np.clip(ocol[oind], 0, 1) * 255).astype(np.uint8)
    want["r"], want["g"], want["b"] = c8[:, 0], c8[:, 1], c8[:, 2]
    raw = open(tmp_path / "cloud.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert f"element vertex {len(oind)}".encode() in head and body == want.tobytes()
    # ---- end to end against the pure oracle chain: same box population and inlier set up to threshold ties
    e2e_pts, _ = onp.filter_3d_pts(rpts, rcol, 0.5)
    assert abs(e2e_pts.shape[1] - fpts.shape[1]) <= 2 + fpts.shape[1] // 100000
    if e2e_pts.shape[1] == fpts.shape[1]:
        e2e_ind = onp.remove_statistical_outlier(e2e_pts.T.astype(np.float32), 20, 0.5)
        common = len(np.intersect1d(e2e_ind, ind))
        assert common >= 0.999 * max(len(e2e_ind), len(ind))
    print(f"\nchain: {F} captured frames per run -> {N} kept, {pts.shape[1]} points, {fpts.shape[1]} inside the box, {len(ind)} inliers; "
          f"worst rel. XYZ error vs the oracle {float(err.max()):.2e}")
    ctx.close()
