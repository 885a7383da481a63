"""What CAN be checked here about the three third-party slices whose parity is unpinned (no cv2 / open3d wheel in the image):

* cv2.undistortPoints (triangulate.py:84-85): the restated inverse, pushed through OpenCV's DOCUMENTED forward distortion model
  (calib3d docs: x'' = x'(1 + k1 r^2 + k2 r^4 + k3 r^6) + 2 p1 x'y' + p2 (r^2 + 2x'^2), ...; u = fx x'' + cx), must give the pixel
  back wherever the 5-step fixed-point iteration has converged -- on all three calibrations the reference ships; pixels on the
  `icdist < 0` bail-out must come back as the plain pinhole normalisation; and the iteration count is what the documented default
  TermCriteria(MAX_ITER, 5, 0.01) says: the result sits on the 5th iterate, measurably away from the 4th and the 6th.
* cv2.cvtColor(BGR2GRAY) (decode_codes.py:86, src/3-capture_decode.py:66): hand-derived known answers from the documented weights
  0.299 R + 0.587 G + 0.114 B in OpenCV's 8-bit fixed point (15-bit coefficients 9798 / 19235 / 3735, round half up), incl. exact
  ties that separate rounding from truncation and triples that separate the 15-bit set from the older 14-bit one.
tools/pin_third_party.py turns all of this into pinned vectors on a machine that has the wheels.
"""
import numpy as np
import pytest

import oracle_c as oc
import oracle_np as onp
from conftest import has_gpu
from scanner import reference_calibration as rc

CALIBS = {"cam_1080": (rc.CAM_MTX, rc.CAM_DIST, (1920, 1080)), "cam_1440": (rc.CAM1440_MTX, rc.CAM1440_DIST, (2560, 1440)),
          "proj": (rc.PROJ_MTX, rc.PROJ_DIST, (1920, 1080))}


def forward_model(xy, K, dist):
    """OpenCV's documented projection with distortion (k1, k2, p1, p2, k3) -- written from the calib3d documentation, independent of
    the inverse under test."""
    k1, k2, p1, p2, k3 = np.asarray(dist, dtype=np.float64).reshape(-1)[:5]
    x, y = xy[:, 0].astype(np.float64), xy[:, 1].astype(np.float64)
    r2 = x * x + y * y
    radial = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], 1)


def fixed_point(pts, K, dist, n):
    """n plain fixed-point steps of the documented inverse (no bail-out) in float64 -- the yardstick for "which iterate is this"."""
    k1, k2, p1, p2, k3 = np.asarray(dist, dtype=np.float64).reshape(-1)[:5]
    x0 = (pts[:, 0].astype(np.float64) - K[0, 2]) / K[0, 0]
    y0 = (pts[:, 1].astype(np.float64) - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(n):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    return np.stack([x, y], 1)


def grid(w, h, step):
    yy, xx = np.mgrid[0:h:step, 0:w:step]
    return np.stack([xx.ravel(), yy.ravel()], 1).astype(np.float32)


def check_round_trip(undistort, name, step=5):
    K, dist, (w, h) = CALIBS[name]
    pts = grid(w, h, step)
    und = np.asarray(undistort(pts, K, dist)).reshape(-1, 2)
    assert und.dtype == np.float32                                             # OpenCV returns CV_32FC2 for float32 input
    pin = np.stack([(pts[:, 0].astype(np.float64) - K[0, 2]) / K[0, 0], (pts[:, 1].astype(np.float64) - K[1, 2]) / K[1, 1]], 1)
    bailed = np.all(und == pin.astype(np.float32), axis=1) & (np.hypot(pin[:, 0], pin[:, 1]) > 0.25)
    res = np.abs(forward_model(und, K, dist) - pts).max(axis=1)
    r = np.hypot(pin[:, 0], pin[:, 1])
    centre = r < 0.2                                                           # the iteration has converged here for all three models
    assert centre.sum() > 300 and not bailed[centre].any()
    assert res[centre].max() < 1e-4, f"{name}: {res[centre].max()} px"         # float32 ray resolution x focal length
    # the 5th iterate, not the 4th or the 6th
    ok = ~bailed                                                               # towards the corners the iteration is slow: iterates differ
    with np.errstate(all="ignore"):
        d4, d5, d6 = (np.abs(fixed_point(pts, K, dist, n).astype(np.float32) - und)[ok].max() for n in (4, 5, 6))
    assert d5 <= 2.4e-7 and d4 > 1e-4 and d6 > 1e-4, (name, d4, d5, d6)
    return bailed, res, r


@pytest.mark.parametrize("name", sorted(CALIBS))
def test_oracle_undistort_inverts_the_documented_forward_model(name):
    for undistort, step in ((lambda p, K, d: oc.undistort(p, K, d), 5), (lambda p, K, d: onp.undistort_points(p, K, d), 23)):   # NumPy twin: per-point loop, coarser grid
        bailed, res, r = check_round_trip(undistort, name, step)
        if name == "cam_1080":
            assert not bailed.any() and res.max() < 0.6                       # mild model: never bails out, half a pixel at the far corners
        if name == "proj":
            assert bailed.sum() > (1000 if step == 5 else 50)                  # k2 = 6.7, k3 = -31.6: the bail-out branch matters


def test_bgr2gray_known_answers_oracle():
    check_bgr2gray(lambda im, bits: onp.bgr_to_gray(im, bits))


BGR_KATS_15 = [  # (b, g, r) -> grey, 15-bit coefficients B 3735, G 19235, R 9798, (sum + 2^14) >> 15
    ((255, 255, 255), 255), ((0, 0, 0), 0), ((0, 0, 255), 76), ((0, 255, 0), 150), ((255, 0, 0), 29), ((128, 128, 128), 128),
    ((255, 0, 255), 105), ((17, 34, 51), 37), ((10, 200, 30), 128),
    # exact ties: sum = k * 32768 + 16384 -> round half UP (truncation would give one less)
    ((4, 12, 0), 8), ((8, 0, 12), 5), ((5, 13, 1), 9), ((9, 1, 13), 6),
    # triples on which the 15-bit set and the older 14-bit set (1868, 9617, 4899, (sum + 2^13) >> 14) disagree
    ((0, 70, 192), 99), ((0, 105, 33), 72), ((0, 105, 237), 133), ((0, 140, 78), 106),
]
BGR_KATS_14 = [((0, 70, 192), 98), ((0, 105, 33), 71), ((0, 105, 237), 132), ((0, 140, 78), 105), ((0, 0, 255), 76), ((255, 255, 255), 255)]


def check_bgr2gray(to_gray):
    for kats, bits in ((BGR_KATS_15, 15), (BGR_KATS_14, 14)):
        im = np.array([k for k, _ in kats], dtype=np.uint8).reshape(1, 1, len(kats), 3)
        got = np.asarray(to_gray(im, bits)).reshape(-1)
        assert list(got.astype(int)) == [v for _, v in kats], (bits, list(got))
    # the documented weights: the fixed-point result is within half a grey level (+ coefficient rounding) of 0.299 R + 0.587 G + 0.114 B
    rng = np.random.default_rng(4)
    im = rng.integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    y = 0.114 * im[..., 0] + 0.587 * im[..., 1] + 0.299 * im[..., 2]
    assert np.abs(np.asarray(to_gray(im, 15)).astype(np.float64) - y).max() <= 0.5 + 0.02


# ------------------------------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    if not has_gpu():
        pytest.skip("needs a HIP device")
    from scanner import _native
    c = _native.Context(0)
    yield c
    c.close()


@gpu
@pytest.mark.parametrize("name", sorted(CALIBS))
def test_gpu_undistort_points_bit_exact_and_round_trip(ctx, name):
    """slgc_undistort_points (the kernels' own undistortPoints) == the oracle bit for bit on the same pixels, for the camera form
    (R = proj_R applied after, triangulate.py:84) and the projector form (:85); and the same forward-model round trip."""
    K, dist, (w, h) = CALIBS[name]
    th = np.deg2rad(-20.0)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    ctx.set_calibration(K, dist, K, dist, R, np.array([[0.25], [0.02], [0.04]]))
    pts = grid(w, h, 3)
    assert np.array_equal(ctx.undistort_points(pts, projector=True), oc.undistort(pts, K, dist))
    assert np.array_equal(ctx.undistort_points(pts, projector=False), oc.undistort(pts, K, dist, R))
    sub = np.random.default_rng(1).uniform(0, [w, h], (20000, 2)).astype(np.float32)          # sub-pixel inputs (the list API takes any float32)
    assert np.array_equal(ctx.undistort_points(sub, projector=True), oc.undistort(sub, K, dist))
    check_round_trip(lambda p, K_, d_: ctx.undistort_points(p, projector=True), name)


@gpu
def test_gpu_bgr2gray_known_answers(ctx):
    check_bgr2gray(lambda im, bits: ctx.to_gray(im, bits))
