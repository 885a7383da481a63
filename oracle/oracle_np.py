"""NumPy CPU restatement of the reference's Gray-code decode + triangulation path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline.  The product path
(``3dscanner-graycode_amd/scanner``) never imports this module and fails loudly when the HIP
library is missing.

Pinning status
--------------
* decode half (a0-a5, a7, a9): **pinned** -- bit-exact against outputs of the reference
  itself (imported in the build container by ``tests/golden/make_golden.py``), committed
  as ``tests/golden/*.npz`` and checked by ``tests/test_oracle_golden.py``.
* law-of-sines half of a8 (``triangulate.py:86-95``): **pinned** the same way (the
  reference's own ``Triangulate.triangulate`` is run with its two cv2 calls served by
  :func:`undistort_points` below).
* ``cv2.undistortPoints`` (OpenCV 4.8.0.76, ``requirements.txt:4``; not vendored, wheel not
  installed, no reference test holds vectors for it): **parity unpinned**.
  :func:`undistort_points` restates the published algorithm of
  ``cvUndistortPointsInternal`` (calib3d/undistort.dispatch.cpp, 4.8): 5 fixed-point
  iterations, ``icdist < 0`` bail-out, ``R`` applied after, float32 output.

All ``file:line`` citations are relative to ``/root/reference``.

Two flavours are kept on purpose:
* ``*_loops`` functions do what the reference does per pixel in Python (string join per
  pixel, nested x/y loops).  They are the reference-cost "port" that ``bench.py`` times.
* the vectorised twins produce identical arrays fast, so tests can use bigger cases; the
  test-suite checks loop == vectorised on small inputs.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------- a0/a1


def frame_ids(n_frames: int):
    """Index tables of ``get_direct_indirect`` (scanner/grayCode/decode_codes.py:109-111).

    The reference keeps ``pattern_len`` as a *float* here and truncates through a uint8
    cast, so N=44 gives hid=[19,17,15,40,38,36] (survey D2/H7).  Indices are relative to
    ``images[2:]``.
    """
    plf = (n_frames - 2) / 4
    hid = np.array([2 * plf - 2, 2 * plf - 4, 2 * plf - 6,
                    4 * plf - 2, 4 * plf - 4, 4 * plf - 6], dtype=np.uint8)
    vid = np.array([1, 3, 5, 2 * plf + 1, 2 * plf + 3, 2 * plf + 5], dtype=np.uint8)
    return hid, vid


def code_len(n_frames: int) -> int:
    """``int((N-2)/4)`` -- decode_codes.py:149."""
    return int((n_frames - 2) / 4)


def direct_indirect(stack: np.ndarray):
    """decode_codes.py:90-122.  ``stack`` is [N,H,W]; computed in float64.

    L_max runs over the six ``hid`` frames only, L_min over the six ``vid`` frames only
    (asymmetric in the reference; reproduced).  0/0 -> NaN, silently.
    """
    st = np.asarray(stack, dtype=np.float64)
    hid, vid = frame_ids(len(st))
    black, white = st[0], st[1]
    with np.errstate(invalid="ignore", divide="ignore"):
        b_inv = white / (white + black)                      # :113
    pat = st[2:]
    with np.errstate(invalid="ignore"):                      # inf - inf etc. on float64 stacks: NaN, silently, like the reference
        l_max = pat[hid].max(axis=0)                         # :116
        l_min = pat[vid].min(axis=0)                         # :117
        l_d = (l_max - l_min) * b_inv                        # :119
        l_g = 2.0 * (l_max - l_d) * b_inv                    # :120 -> (2*(..))*b_inv
    return l_d, l_g


# ----------------------------------------------------------------------------- a2


def is_lit(stack: np.ndarray, l_d: np.ndarray, l_g: np.ndarray, eps=1, m=10):
    """decode_codes.py:125-186 restated as a rule table, last matching rule wins.

    Rule 0 (``L_d < m -> -1``, :169-170) writes the value the array already holds, so
    ``m`` has no effect -- kept as an argument for signature parity only.
    """
    st = np.asarray(stack, dtype=np.float64)
    L = code_len(len(st))
    pat = st[2:]
    normal, inverse = pat[:2 * L], pat[2 * L:]
    out = []
    d = l_d[None]
    g = l_g[None]
    with np.errstate(invalid="ignore"):
        direct_dominates = d > (g + eps)
        for parity in (0, 1):                     # 0: column code "h" (:154), 1: row code "v" (:155)
            n = normal[parity:2 * L:2]
            i = inverse[parity:2 * L:2]
            c = np.full(n.shape, -1, dtype=np.int8)           # :162-163
            c[direct_dominates & (n > (i + eps))] = 1         # :172-173
            c[direct_dominates & ((n + eps) < i)] = 0         # :175-176
            c[((n + eps) < d) & (i > (g + eps))] = 0          # :178-179
            c[(n > (g + eps)) & ((i + eps) < d)] = 1          # :181-182
            out.append(c)
    return out[0], out[1]


def get_codes(stack: np.ndarray):
    """decode_codes.py:231-248."""
    l_d, l_g = direct_indirect(stack)
    return is_lit(stack, l_d, l_g)


def is_lit_loops(stack: np.ndarray, l_d: np.ndarray, l_g: np.ndarray, eps=1, m=10):
    """Same result as :func:`is_lit`, with the COST SHAPE of decode_codes.py:149-182, which is 98 % of the reference's
    ``get_codes`` time: four integer-list fancy-index copies of ``[L,H,W]`` float64 (:157-160), ``L_d`` / ``L_g`` materialised
    ``L`` times with ``np.repeat`` (:165-166), and ten ``codes[np.where(mask)] = value`` scatters through int64 index tuples
    (:169-182).  This is the "port" bench.py times as ``cpu_baseline``; tests check it equals the vectorised twin."""
    st = np.asarray(stack, dtype=np.float64)
    L = code_len(len(st))
    pat = st[2:]
    normal, inverse = pat[:2 * L], pat[2 * L:]
    col_ids, row_ids = list(range(0, 2 * L, 2)), list(range(1, 2 * L, 2))
    n_of = {0: normal[col_ids], 1: normal[row_ids]}                    # fancy indexing copies, like :157-158
    i_of = {0: inverse[col_ids], 1: inverse[row_ids]}                  # :159-160
    codes = {k: np.zeros(n_of[k].shape, dtype=np.int8) - 1 for k in (0, 1)}        # :162-163
    d = np.repeat(np.asarray(l_d)[np.newaxis], L, axis=0)              # :165
    g = np.repeat(np.asarray(l_g)[np.newaxis], L, axis=0)              # :166
    with np.errstate(invalid="ignore"):
        for k in (0, 1):
            codes[k][np.where(d < m)] = -1                             # :169-170 (no-op rule)
        for k in (0, 1):
            codes[k][np.where((d > (g + eps)) & (n_of[k] > (i_of[k] + eps)))] = 1          # :172-173
        for k in (0, 1):
            codes[k][np.where((d > (g + eps)) & ((n_of[k] + eps) < i_of[k]))] = 0          # :175-176
        for k in (0, 1):
            codes[k][np.where(((n_of[k] + eps) < d) & (i_of[k] > (g + eps)))] = 0          # :178-179
        for k in (0, 1):
            codes[k][np.where((n_of[k] > (g + eps)) & ((i_of[k] + eps) < d))] = 1          # :181-182
    return codes[0], codes[1]


def get_codes_loops(stack: np.ndarray):
    """decode_codes.py:231-248 at the reference's cost (see :func:`is_lit_loops`)."""
    l_d, l_g = direct_indirect(stack)
    return is_lit_loops(stack, l_d, l_g)


def merge_runs(code_runs):
    """src/3-capture_decode.py:95-96 -- elementwise max over runs (1 > 0 > -1)."""
    return np.max(np.asarray(code_runs), axis=0)


# ----------------------------------------------------------------------------- a5


def gray_decode(n: int) -> int:
    """decode_codes.py:189-207: prefix XOR from the MSB."""
    shift = n >> 1
    while shift:
        n ^= shift
        shift >>= 1
    return n


def gray_to_decimal(seq) -> int:
    """decode_codes.py:209-229: any -1 -> -1; else MSB-first bits -> Gray -> binary."""
    text = "".join(str(seq[k]) for k in range(len(seq)))
    if "-1" in text:
        return -1
    return gray_decode(int(text, 2))


def codes_to_pixels_loops(h_codes, v_codes):
    """src/3-capture_decode.py:99-100: per-pixel Python loops; v codes are flipped."""
    L, H, W = h_codes.shape
    hp = np.array([gray_to_decimal(h_codes[:, y, x]) for y in range(H) for x in range(W)]).reshape(H, W)
    vp = np.array([gray_to_decimal(np.flip(v_codes[:, y, x])) for y in range(H) for x in range(W)]).reshape(H, W)
    return hp, vp


def codes_to_pixels(h_codes, v_codes):
    """Vectorised twin of :func:`codes_to_pixels_loops` (int64 maps)."""
    L = h_codes.shape[0]

    def one(codes, msb_first):
        bad = (codes < 0).any(axis=0)
        word = np.zeros(codes.shape[1:], dtype=np.int64)
        for k in range(L):
            weight = (L - 1 - k) if msb_first else k
            word |= (codes[k].astype(np.int64) & 1) << weight
        s = 1
        while s < 64:
            word ^= word >> s
            s <<= 1
        word[bad] = -1
        return word

    return one(h_codes, True), one(v_codes, False)


def decode(stack_or_runs, eps=1, m=10):
    """Driver tail (src/3-capture_decode.py:75-100): runs -> codes -> max-merge -> maps."""
    runs = np.asarray(stack_or_runs)
    if runs.ndim == 3:
        runs = runs[None]
    hs, vs = [], []
    for st in runs:
        l_d, l_g = direct_indirect(st)
        h, v = is_lit(st, l_d, l_g, eps, m)
        hs.append(h)
        vs.append(v)
    return codes_to_pixels(merge_runs(hs), merge_runs(vs))


# ----------------------------------------------------------------------------- a6/a7


def scale_proj_mtx(proj_mtx, proj_size, proj_calib_size):
    """triangulate.py:28-33 (returns a scaled copy; the reference mutates in place)."""
    k = np.array(proj_mtx, dtype=np.float64, copy=True)
    k[0, :] = k[0, :] * (proj_size[0] / proj_calib_size[0])
    k[1, :] = k[1, :] * (proj_size[1] / proj_calib_size[1])
    return k


def cam_proj_pts_loops(h_pixels, v_pixels, cam_size, proj_size, img_white):
    """triangulate.py:39-71: x-major scan, drop (-1), clamp to projector, gather colour."""
    cam_w, cam_h = cam_size
    proj_w, proj_h = proj_size
    cam, proj, col = [], [], []
    for x in range(cam_w):
        for y in range(cam_h):
            hv, vv = h_pixels[y, x], v_pixels[y, x]
            if hv == -1 or vv == -1:
                continue
            cam.append([x, y])
            proj.append([min(proj_w - 1, hv), min(proj_h - 1, vv)])
            col.append(img_white[y, x, :])
    return (np.array(cam, dtype=np.float32), np.array(proj, dtype=np.float32),
            np.array(col).astype(np.float64) / 255.0)


def cam_proj_pts(h_pixels, v_pixels, cam_size, proj_size, img_white=None, order="x"):
    """Vectorised twin of :func:`cam_proj_pts_loops`.  ``order='x'`` is the reference's
    column-major scan; ``order='row'`` is row-major (the build's multi-GPU band order)."""
    cam_w, cam_h = cam_size
    proj_w, proj_h = proj_size
    h = np.asarray(h_pixels)[:cam_h, :cam_w]
    v = np.asarray(v_pixels)[:cam_h, :cam_w]
    ok = (h != -1) & (v != -1)
    if order == "x":
        xs, ys = np.nonzero(ok.T)
    else:
        ys, xs = np.nonzero(ok)
    cam = np.stack([xs, ys], axis=1).astype(np.float32).reshape(-1, 2)
    proj = np.stack([np.minimum(proj_w - 1, h[ys, xs]), np.minimum(proj_h - 1, v[ys, xs])],
                    axis=1).astype(np.float32).reshape(-1, 2)
    col = None
    if img_white is not None:
        col = img_white[ys, xs, :].astype(np.float64).reshape(-1, img_white.shape[2]) / 255.0
    return cam, proj, col


# ----------------------------------------------------------------------------- a8


def undistort_points(pts, cam_mtx, dist, R=None):
    """OpenCV 4.8 ``cv::undistortPoints(src, K, dist, R)`` restated (PARITY UNPINNED).

    pts: float32 [M,2] (or [M,1,2]).  Returns float32 [M,1,2] like OpenCV.
    Default TermCriteria(MAX_ITER, 5, 0.01): exactly five iterations, no EPS test.
    """
    p = np.asarray(pts, dtype=np.float32).reshape(-1, 2).astype(np.float64)
    A = np.asarray(cam_mtx, dtype=np.float64)
    k = np.zeros(14)
    dk = np.asarray(dist, dtype=np.float64).ravel()
    k[:dk.size] = dk
    RR = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    fx, fy, cx, cy = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy
    u, v = p[:, 0], p[:, 1]
    x = (u - cx) * ifx
    y = (v - cy) * ify
    x0, y0 = x.copy(), y.copy()
    live = np.ones(x.shape, dtype=bool)          # False once the icdist<0 bail-out fired
    with np.errstate(all="ignore"):
        for _ in range(5):
            r2 = x * x + y * y
            icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            bail = live & (icdist < 0)
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
            dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
            nx = (x0 - dx) * icdist
            ny = (y0 - dy) * icdist
            step = live & ~bail
            x = np.where(step, nx, np.where(bail, (u - cx) * ifx, x))
            y = np.where(step, ny, np.where(bail, (v - cy) * ify, y))
            live = step
        xx = RR[0, 0] * x + RR[0, 1] * y + RR[0, 2]
        yy = RR[1, 0] * x + RR[1, 1] * y + RR[1, 2]
        ww = 1.0 / (RR[2, 0] * x + RR[2, 1] * y + RR[2, 2])
        out = np.stack([xx * ww, yy * ww], axis=1).astype(np.float32)
    return out.reshape(-1, 1, 2)


def to_homogeneous(pts):
    """``cv2.convertPointsToHomogeneous``: append 1 (same dtype), shape [M,1,3]."""
    p = np.asarray(pts)
    p = p.reshape(-1, 1, p.shape[-1])
    return np.concatenate([p, np.ones(p.shape[:2] + (1,), dtype=p.dtype)], axis=2)


def triangulate(cam_pts, proj_pts, cam_mtx, cam_dist, proj_mtx, proj_dist, proj_R, proj_T):
    """triangulate.py:73-97: undistort both sides, then angle-angle-side on the camera ray.

    ``proj_mtx`` must already be scaled (:func:`scale_proj_mtx`).  Result float64 (3,M):
    point relative to the camera centre, expressed in projector axes.
    """
    cam_h = to_homogeneous(undistort_points(cam_pts, cam_mtx, cam_dist, R=proj_R))[:, 0].T    # :84 f32 (3,M)
    proj_h = to_homogeneous(undistort_points(proj_pts, proj_mtx, proj_dist))[:, 0].T          # :85
    T = np.asarray(proj_T, dtype=np.float64)[:, 0]                                            # :86
    with np.errstate(all="ignore"):
        t_len = np.linalg.norm(T)
        ray = cam_h / np.linalg.norm(cam_h, axis=0)                                           # :90 f32
        alpha = np.arccos(np.dot(-T, ray) / t_len)                                            # :91
        beta = np.arccos(np.dot(T, proj_h) / (t_len * np.linalg.norm(proj_h, axis=0)))        # :92
        gamma = np.pi - alpha - beta                                                          # :93
        rng = t_len * np.sin(beta) / np.sin(gamma)                                            # :94
        return ray * rng                                                                      # :95


def filter_3d_pts(pts, colors, threshold=0.5):
    """triangulate.py:99-122: strict box filter on |X|,|Y|,|Z| (NaN drops)."""
    with np.errstate(invalid="ignore"):
        keep = ((pts[2] < threshold) & (pts[2] > -threshold) & (pts[1] < threshold) &
                (pts[1] > -threshold) & (pts[0] < threshold) & (pts[0] > -threshold))
    return pts[:, keep], (None if colors is None else colors[keep])


# ----------------------------------------------------------------------------- synthetic inputs


def gray_code_table(width: int, height: int):
    """generate_codes.py:22-32: n_bits = ceil(log2(max(w,h))), g = i ^ (i>>1), MSB first."""
    size = max(width, height)
    n_bits = int(np.ceil(np.log2(size)))
    i = np.arange(size, dtype=np.int64)
    g = i ^ (i >> 1)
    return ((g[:, None] >> np.arange(n_bits - 1, -1, -1)[None]) & 1).astype(np.uint8)


def pattern_sequence(width: int, height: int):
    """generate_codes.py:34-81 frame-order contract (survey a0), for width >= height.

    images[0]=black, [1]=white; P[2k]=column-code bit k (MSB first); P[2k+1]=row-code bit
    L-1-k; second half = inverses.  uint8 [4L+2, height, width] with 0/255.
    """
    codes = gray_code_table(width, height)
    L = codes.shape[1]
    seq = np.zeros((4 * L + 2, height, width), dtype=np.uint8)
    seq[1] = 255
    stripe = width // len(codes)                       # :60 (1 when width >= height)
    for j in range(L):
        col_bits = np.zeros(width, dtype=np.uint8)
        n_cols = min(width, len(codes) * stripe)
        col_bits[:n_cols] = np.repeat(codes[:, j], stripe)[:n_cols] * 255
        seq[2 * j + 2] = col_bits[None, :]
        seq[2 * j + 2 + 2 * L] = 255 - seq[2 * j + 2]
        row_id = 2 * (L - (j + 1)) + 3
        rows = min(height, len(codes)) * stripe
        row_bits = np.zeros(height, dtype=np.uint8)
        row_bits[:rows] = np.repeat(codes[:min(height, len(codes)), j], stripe)[:rows] * 255
        seq[row_id, :rows, :] = row_bits[:rows, None]
        seq[row_id + 2 * L, :rows, :] = 255 - seq[row_id, :rows, :]
    return seq


def synth_scene(n_frames: int, H: int, W: int, seed: int = 1, shadow: bool = True, noise: int = 3):
    """S-scene synthetic capture of SURVEY.md section 8(d) (uint8 [N,H,W])."""
    L = (n_frames - 2) // 4
    yy, xx = np.mgrid[0:H, 0:W]
    xs = (0.9 * xx + 5 * np.sin(yy / 50.0)).astype(np.int64) % (1 << L)
    ys = (0.9 * yy + 5 * np.cos(xx / 50.0)).astype(np.int64) % (1 << L)
    gx, gy = xs ^ (xs >> 1), ys ^ (ys >> 1)
    img = np.full((n_frames, H, W), 15, dtype=np.int64)
    img[1] = 195
    for k in range(L):
        bx = (gx >> (L - 1 - k)) & 1
        by = (gy >> k) & 1
        img[2 + 2 * k] = 15 + 180 * bx
        img[3 + 2 * k] = 15 + 180 * by
        img[2 + 2 * L + 2 * k] = 15 + 180 * (1 - bx)
        img[3 + 2 * L + 2 * k] = 15 + 180 * (1 - by)
    if shadow:
        y0, y1 = int(0.30 * H), int(0.30 * H + 0.387 * H)
        x0, x1 = int(0.55 * W), int(0.55 * W + 0.387 * W)
        img[:, y0:y1, x0:x1] = 15
    if noise:
        img += np.random.default_rng(seed).integers(-noise, noise + 1, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_scene_int(n_frames: int, H: int, W: int, seed: int = 1, noise: int = 3, shadow: bool = True,
                    row0: int = 0, rows=None):
    """Bit-identical NumPy twin of the device generator csrc/synth.hip (all-integer warp, hash
    noise).  Returns (stack uint8 [N,rows,W], xs, ys) where xs/ys are the projector coordinates
    encoded at each pixel (mod 2^L)."""
    rows = H - row0 if rows is None else rows
    L = (n_frames - 2) // 4
    yy, xx = np.mgrid[row0:row0 + rows, 0:W].astype(np.int64)

    def tri(t):
        return np.abs(((t >> 5) % 10) - 5)

    msk = (1 << L) - 1
    xs = (((29 * xx) >> 5) + tri(yy)) & msk
    ys = (((29 * yy) >> 5) + tri(xx)) & msk
    gx, gy = xs ^ (xs >> 1), ys ^ (ys >> 1)
    img = np.full((n_frames, rows, W), 15, dtype=np.int64)
    img[1] = 195
    for k in range(L):
        bx = (gx >> (L - 1 - k)) & 1
        by = (gy >> k) & 1
        img[2 + 2 * k] = 15 + 180 * bx
        img[3 + 2 * k] = 15 + 180 * by
        img[2 + 2 * L + 2 * k] = 15 + 180 * (1 - bx)
        img[3 + 2 * L + 2 * k] = 15 + 180 * (1 - by)
    if shadow:
        sy0, sy1 = int(0.30 * H), int(0.30 * H + 0.387 * H)
        sx0, sx1 = int(0.55 * W), int(0.55 * W + 0.387 * W)
        inside = (yy >= sy0) & (yy < sy1) & (xx >= sx0) & (xx < sx1)
        img[:, inside] = 15
    if noise > 0:
        gp = (yy * W + xx).astype(np.uint32)
        f = np.arange(n_frames, dtype=np.uint32)[:, None, None]
        with np.errstate(over="ignore"):
            x = gp[None] * np.uint32(0x9E3779B1) + f * np.uint32(0x85EBCA77) + np.uint32(seed)
            x ^= x >> np.uint32(16)
            x *= np.uint32(0x7FEB352D)
            x ^= x >> np.uint32(15)
            x *= np.uint32(0x846CA68B)
            x ^= x >> np.uint32(16)
        img += (x % np.uint32(2 * noise + 1)).astype(np.int64) - noise
    return np.clip(img, 0, 255).astype(np.uint8), xs, ys


# ----------------------------------------------------------------------------- physically consistent synthetic capture
# One surface seen by the camera AND lit by the projector, the way /root/reference/src/4-triangulate.py:50-64 assumes its inputs were
# made: every camera pixel's ray is cast into a scene (a tilted back plane with a sphere in front of it, 0.35-0.6 m from the camera), the
# hit point goes through the stereo pose (x_p = R x_c + T) and the projector's FORWARD lens model (OpenCV's documented distortion
# polynomial) to the projector pixel that lights it, and that pixel's (column, row) is what the Gray-code frames encode.  Pixels the
# projector cannot reach (outside its raster, behind the sphere as seen from the projector, or where its lens model is no longer
# monotonic) stay at ambient level in every frame -- the decoder's shadow mask.  All float64, + - * / sqrt only, evaluated in ONE fixed
# order: the device generator (csrc/synth.hip: k_synth_physical_codes) is a bit-identical twin.
PHYS_PLANE_POINT = (0.0, 0.0, 0.56)          # a point of the back plane, camera coordinates, metres
PHYS_PLANE_NORMAL = (0.18, -0.10, -1.0)      # its normal (not normalised)
PHYS_SPHERE_CENTRE = (0.045, 0.015, 0.46)
PHYS_SPHERE_RADIUS = 0.06
PHYS_R2_MAX = 0.16                           # projector lens: lit only where x'^2 + y'^2 <= this (the reference's `proj` model folds over at ~0.22)
PHYS_UNDISTORT_ITERS = 20                    # fixed-point inverse of the camera lens run to convergence (the reference's own ray uses 5)


def _dist14(dist):
    k = np.zeros(14)
    d = np.asarray(dist, dtype=np.float64).ravel()
    k[:d.size] = d
    return k


def synth_physical_codes(H, W, proj_size, cam_mtx, cam_dist, proj_mtx, proj_dist, proj_R, proj_T, row0=0, rows=None, code_bits=15, r2_max=PHYS_R2_MAX):
    """-> (h int16 [rows,W], v int16 [rows,W], truth float64 [rows,W,3]): the projector pixel that lights each camera pixel (-1 = unlit) and
    the true surface point in the frame Triangulate.triangulate reports (relative to the camera centre, projector axes: R x_c).
    code_bits: a pattern of L bits per axis can only address projector pixels below 2^L (44 frames = 10 bits on a 1920-pixel projector:
    BASELINE.json configs[1], [2]); the rest of the raster stays dark."""
    rows = H - row0 if rows is None else rows
    A = np.asarray(cam_mtx, dtype=np.float64)
    P = np.asarray(proj_mtx, dtype=np.float64)
    R = np.asarray(proj_R, dtype=np.float64).reshape(3, 3)
    T = np.asarray(proj_T, dtype=np.float64).reshape(3)
    k, q = _dist14(cam_dist), _dist14(proj_dist)
    pw, ph = int(proj_size[0]), int(proj_size[1])
    yy, xx = np.mgrid[row0:row0 + rows, 0:W]
    u, v = xx.astype(np.float64), yy.astype(np.float64)
    with np.errstate(all="ignore"):
        # camera pixel -> ray (x, y, 1): fixed-point inverse of the forward model, PHYS_UNDISTORT_ITERS rounds, no early exit
        x0 = (u - A[0, 2]) / A[0, 0]
        y0 = (v - A[1, 2]) / A[1, 1]
        x, y = x0, y0
        for _ in range(PHYS_UNDISTORT_ITERS):
            r2 = x * x + y * y
            icd = (1.0 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            dx = ((2.0 * k[2]) * x * y + k[3] * (r2 + (2.0 * x) * x)) + (k[8] * r2 + (k[9] * r2) * r2)
            dy = (k[2] * (r2 + (2.0 * y) * y) + (2.0 * k[3]) * x * y) + (k[10] * r2 + (k[11] * r2) * r2)
            x = (x0 - dx) * icd
            y = (y0 - dy) * icd
        # nearest hit: sphere |t d - c|^2 = r^2 with d = (x, y, 1), else the plane n.(t d - p0) = 0
        cx, cy, cz = PHYS_SPHERE_CENTRE
        a = (x * x + y * y) + 1.0
        b = (x * cx + y * cy) + cz
        cc = ((cx * cx + cy * cy) + cz * cz) - PHYS_SPHERE_RADIUS * PHYS_SPHERE_RADIUS
        disc = b * b - a * cc
        on_sphere = disc > 0.0
        ts = (b - np.sqrt(np.where(on_sphere, disc, 0.0))) / a
        nx, ny, nz = PHYS_PLANE_NORMAL
        px_, py_, pz_ = PHYS_PLANE_POINT
        tp = ((nx * px_ + ny * py_) + nz * pz_) / ((nx * x + ny * y) + nz)
        t = np.where(on_sphere, ts, tp)
        X, Y, Z = t * x, t * y, t
        # into the projector frame
        Xp = ((R[0, 0] * X + R[0, 1] * Y) + R[0, 2] * Z) + T[0]
        Yp = ((R[1, 0] * X + R[1, 1] * Y) + R[1, 2] * Z) + T[1]
        Zp = ((R[2, 0] * X + R[2, 1] * Y) + R[2, 2] * Z) + T[2]
        # a plane point is in the sphere's shadow when the segment to the projector centre C = -R^T T crosses the sphere
        Cx = -((R[0, 0] * T[0] + R[1, 0] * T[1]) + R[2, 0] * T[2])
        Cy = -((R[0, 1] * T[0] + R[1, 1] * T[1]) + R[2, 1] * T[2])
        Cz = -((R[0, 2] * T[0] + R[1, 2] * T[1]) + R[2, 2] * T[2])
        ex, ey, ez = Cx - X, Cy - Y, Cz - Z                      # towards the projector
        fx_, fy_, fz_ = X - cx, Y - cy, Z - cz
        ea = (ex * ex + ey * ey) + ez * ez
        eb = (ex * fx_ + ey * fy_) + ez * fz_
        ec = ((fx_ * fx_ + fy_ * fy_) + fz_ * fz_) - PHYS_SPHERE_RADIUS * PHYS_SPHERE_RADIUS
        shadow = (~on_sphere) & ((eb * eb - ea * ec) > 0.0) & (eb < 0.0)
        # forward lens model of the projector (OpenCV: radial quotient, tangential, thin prism), skew ignored like undistortPoints does
        xn, yn = Xp / Zp, Yp / Zp
        r2 = xn * xn + yn * yn
        rad = (1.0 + ((q[4] * r2 + q[1]) * r2 + q[0]) * r2) / (1.0 + ((q[7] * r2 + q[6]) * r2 + q[5]) * r2)
        xd = (xn * rad + ((2.0 * q[2]) * xn * yn + q[3] * (r2 + (2.0 * xn) * xn))) + (q[8] * r2 + (q[9] * r2) * r2)
        yd = (yn * rad + (q[2] * (r2 + (2.0 * yn) * yn) + (2.0 * q[3]) * xn * yn)) + (q[10] * r2 + (q[11] * r2) * r2)
        pu = np.floor((P[0, 0] * xd + P[0, 2]) + 0.5)
        pv = np.floor((P[1, 1] * yd + P[1, 2]) + 0.5)
        top = float(min(1 << code_bits, 32767) - 1)
        lit = (Zp > 0.0) & (r2 <= r2_max) & (pu >= 0.0) & (pu <= pw - 1.0) & (pv >= 0.0) & (pv <= ph - 1.0) & (~shadow) & (t > 0.0) & \
              (pu <= top) & (pv <= top)
        h = np.where(lit, pu, -1.0).astype(np.int16)
        vv = np.where(lit, pv, -1.0).astype(np.int16)
        truth = np.stack([(R[0, 0] * X + R[0, 1] * Y) + R[0, 2] * Z, (R[1, 0] * X + R[1, 1] * Y) + R[1, 2] * Z,
                          (R[2, 0] * X + R[2, 1] * Y) + R[2, 2] * Z], axis=-1)
        truth[~lit] = np.nan
    return h, vv, truth


def render_codes(n_frames, H, W, h, v, seed=1, noise=3, row0=0, gains=(140, 180)):
    """Frames of a capture whose pixel (y, x) is lit by projector pixel (h, v) (-1 = unlit: ambient in every frame): the frame order of
    generate_codes.py:53-79, ambient 15, a 140 / 180 checker of surface gains, hash noise -- twin of csrc/synth.hip: k_synth_render."""
    rows = h.shape[0]
    L = (n_frames - 2) // 4
    yy, xx = np.mgrid[row0:row0 + rows, 0:W].astype(np.int64)
    lit = (h != -1) & (v != -1)
    gain = np.where((((xx >> 4) ^ (yy >> 4)) & 1) == 1, int(gains[1]), int(gains[0]))
    hs, vs = h.astype(np.int64) & ((1 << L) - 1), v.astype(np.int64) & ((1 << L) - 1)
    gx, gy = hs ^ (hs >> 1), vs ^ (vs >> 1)
    img = np.full((n_frames, rows, W), 15, dtype=np.int64)
    img[1] = np.where(lit, 15 + gain, 15)
    for k in range(L):
        bx = (gx >> (L - 1 - k)) & 1
        by = (gy >> k) & 1
        img[2 + 2 * k] = np.where(lit, 15 + gain * bx, 15)
        img[3 + 2 * k] = np.where(lit, 15 + gain * by, 15)
        img[2 + 2 * L + 2 * k] = np.where(lit, 15 + gain * (1 - bx), 15)
        img[3 + 2 * L + 2 * k] = np.where(lit, 15 + gain * (1 - by), 15)
    if noise > 0:
        gp = (yy * W + xx).astype(np.uint32)
        f = np.arange(n_frames, dtype=np.uint32)[:, None, None]
        with np.errstate(over="ignore"):
            x = gp[None] * np.uint32(0x9E3779B1) + f * np.uint32(0x85EBCA77) + np.uint32(seed)
            x ^= x >> np.uint32(16)
            x *= np.uint32(0x7FEB352D)
            x ^= x >> np.uint32(15)
            x *= np.uint32(0x846CA68B)
            x ^= x >> np.uint32(16)
        img += (x % np.uint32(2 * noise + 1)).astype(np.int64) - noise
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_physical(n_frames, H, W, proj_size, calib, seed=1, noise=3, row0=0, rows=None, gains=(140, 180), r2_max=PHYS_R2_MAX):
    """-> (stack uint8 [N,rows,W], h, v, truth): see synth_physical_codes / render_codes.  calib = (cam_mtx, cam_dist, proj_mtx (scaled),
    proj_dist, proj_R, proj_T) as slgc_set_calibration takes them.  gains / r2_max: slgc_synth_physical_ex_dev's knobs."""
    h, v, truth = synth_physical_codes(H, W, proj_size, *calib, row0=row0, rows=rows, code_bits=(n_frames - 2) // 4, r2_max=r2_max)
    return render_codes(n_frames, H, W, h, v, seed=seed, noise=noise, row0=row0, gains=gains), h, v, truth


def gray_to_bgr_capture(stack, row0=0):
    """[N,rows,W] grey frames -> [N,rows,W,3] BGR frames as csrc/synth.hip k_synth_bgr makes them: B = clip(g + ((7 x + 3 y) mod 11) - 5), G = g,
    R = clip(g - (((5 x + 11 y) mod 9) - 4)), y counted in the whole image."""
    n, rows, W = stack.shape
    yy, xx = np.mgrid[row0:row0 + rows, 0:W]
    g = stack.astype(np.int64)
    b = np.clip(g + ((7 * xx + 3 * yy) % 11) - 5, 0, 255)
    r = np.clip(g - (((5 * xx + 11 * yy) % 9) - 4), 0, 255)
    return np.stack([b, g, r], axis=-1).astype(np.uint8)


def synth_uniform(n_frames, H, W, seed=0, row0=0, rows=None):
    """SURVEY.md 8(d) S-uniform as the device generates it (csrc/synth.hip: k_synth_uniform): every byte uniform in 0..255 from a counter
    hash keyed by (frame, dword of the whole image, seed); little-endian bytes of one mix32 per 4 pixels.  W % 4 == 0."""
    rows = H - row0 if rows is None else rows
    assert W % 4 == 0
    q = (np.arange(rows * W // 4, dtype=np.uint64) + np.uint64(row0 * W // 4)).astype(np.uint32)
    f = np.arange(n_frames, dtype=np.uint32)[:, None]
    with np.errstate(over="ignore"):
        x = q[None] * np.uint32(0x9E3779B1) + f * np.uint32(0x85EBCA77) + np.uint32(seed)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return np.ascontiguousarray(x).astype("<u4").view(np.uint8).reshape(n_frames, rows, W)


# ----------------------------------------------------------------------------- "next" rows


def frame_diff_counts(images, thresh=50):
    """counts[j] = len(np.argwhere(cv2.absdiff(images[j+1], images[j]) > thresh)) -- decode_codes.py:49-53."""
    im = np.asarray(images, dtype=np.float64)
    return np.array([int((np.abs(im[j + 1] - im[j]) > thresh).sum()) for j in range(len(im) - 1)], dtype=np.int64)


def remove_bad_images(images):
    """decode_codes.py:34-68 (transition-frame filter); cv2.absdiff restated as |a - b| (OpenCV absent: that one call is
    unpinned, the state machine is pinned by tests/golden/ingest.npz)."""
    d = frame_diff_counts(images, 50)
    diff1, diff2 = int(d[0]), int(d[1])
    kept = []
    for i in range(len(images) - 3):
        diff3 = int(d[i + 2])
        if diff1 < diff2 and diff1 < diff3 and diff2 < diff3 and i + 1 not in kept:
            if len(kept) == 0 or (kept[-1] != i and kept[-1] != i + 2):
                kept.append(i + 1)
        elif diff2 <= diff1 and diff2 <= diff3 and i + 2 not in kept:
            if len(kept) == 0 or (kept[-1] != i + 1 and kept[-1] != i + 3):
                kept.append(i + 2)
                diff3 = -1
                diff2 = -1
        diff1, diff2 = diff2, diff3
    return kept


def bgr_to_gray(images, coeff_bits=15):
    """OpenCV 8-bit BGR2GRAY fixed-point luma (PARITY UNPINNED: cv2 not installed): 4.x uses 15-bit coefficients
    (R 9798, G 19235, B 3735), older releases 14-bit (4899, 9617, 1868); rounding = + half, then shift."""
    im = np.asarray(images).astype(np.int64)
    ry, gy, by = (9798, 19235, 3735) if coeff_bits == 15 else (4899, 9617, 1868)
    return ((im[..., 0] * by + im[..., 1] * gy + im[..., 2] * ry + (1 << (coeff_bits - 1))) >> coeff_bits).astype(np.uint8)


def knn_mean_distance(points, k=20):
    """Mean distance to the k nearest points (self included), exact k-NN through scipy's cKDTree -- the arithmetic of Open3D's
    remove_statistical_outlier (PARITY with Open3D UNPINNED: not installed).  points float32 [M,3]; float64 distances."""
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float32).astype(np.float64)
    d, _ = cKDTree(p).query(p, k=k)
    return d.reshape(len(p), -1).mean(axis=1)


def remove_statistical_outlier(points, nb_neighbors=20, std_ratio=0.5):
    avg = knn_mean_distance(points, nb_neighbors)
    valid = avg > 0
    mean = avg[valid].sum() / valid.sum()
    std = np.sqrt(((avg[valid] - mean) ** 2).sum() / (valid.sum() - 1))
    return np.nonzero(valid & (avg < mean + std_ratio * std))[0]
