"""ctypes view of oracle/libslgc_oracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/slgc_oracle.c.  Imported by tests/, smoke() and the
cpu_baseline leg of bench.py; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SLGC_ORACLE_SO") or os.path.join(_HERE, "libslgc_oracle.so")      # SLGC_ORACLE_SO: the sanitizer build (make -C oracle san)


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _load():
    if "SLGC_ORACLE_SO" not in os.environ and (not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "slgc_oracle.c"))):
        build()
    lib = C.CDLL(_SO)
    lib.orc_cam_proj_pts.restype = C.c_int64
    lib.orc_filter.restype = C.c_int64
    lib.orc_gray_decode.restype = C.c_int64
    lib.orc_gray_decode.argtypes = [C.c_int64]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _stack(stack):
    st = np.asarray(stack)
    if st.dtype == np.uint8:
        return np.ascontiguousarray(st), 0
    return np.ascontiguousarray(st, dtype=np.float64), 1


def _d9(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def set_threads(n):
    """Host threads for the per-pixel loops (default 1 = scalar).  Returns the value in effect."""
    return int(lib().orc_set_threads(int(n)))


def frame_ids(n):
    h = (C.c_int * 6)()
    v = (C.c_int * 6)()
    rc = lib().orc_frame_ids(int(n), h, v)
    if rc:
        raise ValueError(f"N={n} unsupported")
    return np.array(h[:]), np.array(v[:])


def direct_indirect(stack):
    st, f = _stack(stack)
    N, H, W = st.shape
    ld = np.empty((H, W))
    lg = np.empty((H, W))
    assert lib().orc_direct_indirect(_p(st), f, N, H, W, _p(ld), _p(lg)) == 0
    return ld, lg


def is_lit(stack, l_d, l_g, eps=1, m=10):
    st, f = _stack(stack)
    N, H, W = st.shape
    L = int((N - 2) / 4)
    hc = np.empty((L, H, W), np.int8)
    vc = np.empty((L, H, W), np.int8)
    ld = np.ascontiguousarray(l_d, dtype=np.float64)
    lg = np.ascontiguousarray(l_g, dtype=np.float64)
    assert lib().orc_is_lit(_p(st), f, N, H, W, _p(ld), _p(lg), C.c_double(eps), C.c_double(m), _p(hc), _p(vc)) == 0
    return hc, vc


def get_codes(stack):
    return is_lit(stack, *direct_indirect(stack))


def codes_to_pixels(h_codes, v_codes):
    """h_codes/v_codes: [L,H,W] or [R,L,H,W] int8 (R runs are max-merged)."""
    hc = np.ascontiguousarray(h_codes, dtype=np.int8)
    vc = np.ascontiguousarray(v_codes, dtype=np.int8)
    if hc.ndim == 3:
        hc, vc = hc[None], vc[None]
    R, L, H, W = hc.shape
    hp = np.empty((H, W), np.int64)
    vp = np.empty((H, W), np.int64)
    assert lib().orc_codes_to_pixels(_p(hc), _p(vc), R, L, H, W, _p(hp), _p(vp)) == 0
    return hp, vp


def decode(stack_or_runs, eps=1, m=10):
    st = np.asarray(stack_or_runs)
    if st.ndim == 3:
        st = st[None]
    st, f = _stack(st)
    R, N, H, W = st.shape
    hp = np.empty((H, W), np.int64)
    vp = np.empty((H, W), np.int64)
    rc = lib().orc_decode(_p(st), f, R, N, H, W, C.c_double(eps), C.c_double(m), _p(hp), _p(vp))
    if rc:
        raise ValueError("orc_decode failed")
    return hp, vp


def cam_proj_pts(h_pixels, v_pixels, cam_size, proj_size, img_white=None, order="x"):
    h = np.ascontiguousarray(h_pixels, dtype=np.int64)
    v = np.ascontiguousarray(v_pixels, dtype=np.int64)
    cw, ch = int(cam_size[0]), int(cam_size[1])
    pw, ph = int(proj_size[0]), int(proj_size[1])
    assert h.shape == (ch, cw) and v.shape == (ch, cw)
    wh = None if img_white is None else np.ascontiguousarray(img_white, dtype=np.uint8)
    o = 0 if order == "x" else 1
    M = lib().orc_cam_proj_pts(_p(h), _p(v), cw, ch, pw, ph, _p(wh), o, None, None, None)
    cam = np.empty((M, 2), np.float32)
    proj = np.empty((M, 2), np.float32)
    col = np.empty((M, 3), np.float64) if wh is not None else None
    lib().orc_cam_proj_pts(_p(h), _p(v), cw, ch, pw, ph, _p(wh), o, _p(cam), _p(proj), _p(col))
    return cam, proj, col


def undistort(pts, K, dist, R=None):
    p = np.ascontiguousarray(np.asarray(pts, dtype=np.float32).reshape(-1, 2))
    out = np.empty_like(p)
    d = _d9(dist)
    r = None if R is None else _d9(R)
    assert lib().orc_undistort(_p(p), C.c_int64(len(p)), _p(_d9(K)), _p(d), len(d), _p(r), _p(out)) == 0
    return out


def triangulate(cam_pts, proj_pts, cam_mtx, cam_dist, proj_mtx, proj_dist, proj_R, proj_T):
    a = np.ascontiguousarray(np.asarray(cam_pts, dtype=np.float32).reshape(-1, 2))
    b = np.ascontiguousarray(np.asarray(proj_pts, dtype=np.float32).reshape(-1, 2))
    M = len(a)
    xyz = np.empty((3, M), np.float64)
    cd, pd = _d9(cam_dist), _d9(proj_dist)
    rc = lib().orc_triangulate(_p(a), _p(b), C.c_int64(M), _p(_d9(cam_mtx)), _p(cd), len(cd), _p(_d9(proj_mtx)), _p(pd),
                               len(pd), _p(_d9(proj_R)), _p(_d9(proj_T)), _p(xyz))
    assert rc == 0
    return xyz


def filter_3d_pts(pts, colors, threshold=0.5):
    x = np.ascontiguousarray(pts, dtype=np.float64)
    M = x.shape[1]
    c = None if colors is None else np.ascontiguousarray(colors, dtype=np.float64)
    kept = lib().orc_filter(_p(x), _p(c), C.c_int64(M), C.c_double(threshold), None, None)
    xo = np.empty((3, kept))
    co = None if c is None else np.empty((kept, 3))
    lib().orc_filter(_p(x), _p(c), C.c_int64(M), C.c_double(threshold), _p(xo), _p(co))
    return xo, co


def scan_dense(stack_or_runs, proj_size, cam_mtx, cam_dist, proj_mtx, proj_dist, proj_R, proj_T, eps=1, m=10):
    st = np.asarray(stack_or_runs)
    if st.ndim == 3:
        st = st[None]
    st, f = _stack(st)
    R, N, H, W = st.shape
    hp = np.empty((H, W), np.int64)
    vp = np.empty((H, W), np.int64)
    xyz = np.empty((3, H, W), np.float64)
    cd, pd = _d9(cam_dist), _d9(proj_dist)
    rc = lib().orc_scan_dense(_p(st), f, R, N, H, W, C.c_double(eps), C.c_double(m), int(proj_size[0]), int(proj_size[1]),
                              _p(_d9(cam_mtx)), _p(cd), len(cd), _p(_d9(proj_mtx)), _p(pd), len(pd), _p(_d9(proj_R)),
                              _p(_d9(proj_T)), _p(hp), _p(vp), _p(xyz))
    assert rc == 0
    return hp, vp, xyz
