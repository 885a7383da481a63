"""Drop-in for ``scanner/triangulation/triangulate.py`` of the reference.

``Triangulate`` keeps the reference's constructor (positional order of triangulate.py:5-17, as
called at src/4-triangulate.py:50-61), method names, return shapes/dtypes and point order; the
loops and the arithmetic run in HIP kernels (csrc/correspond.hip, csrc/triangulate.hip).
``Triangulation`` adds the fused ``compute()`` named by the build's north star.
"""
from __future__ import annotations

import numpy as np

from .. import _native
from .._native import default_context

__all__ = ["Triangulate", "Triangulation"]


class Triangulate(object):
    def __init__(self, h_pixels=None, v_pixels=None, cam_size=(1920, 1080), cam_mtx=None, cam_dist=None,
                 proj_size=(1920, 1080), proj_calib_size=(1920, 1080), proj_mtx=None, proj_dist=None, proj_R=None,
                 proj_T=None, image_folder=None, ctx=None):
        self.h_pixels = h_pixels
        self.v_pixels = v_pixels
        self.cam_w, self.cam_h = cam_size
        self.cam_mtx = cam_mtx
        self.cam_dist = cam_dist
        self.proj_w, self.proj_h = proj_size
        self.proj_mtx = proj_mtx
        self.proj_dist = proj_dist
        # Reference triangulate.py:28-33 scales the caller's matrix IN PLACE; kept for drop-in fidelity.
        if proj_mtx is not None:
            calib_w, calib_h = proj_calib_size
            self.proj_mtx[0, :] = self.proj_mtx[0, :] * (self.proj_w / calib_w)
            self.proj_mtx[1, :] = self.proj_mtx[1, :] * (self.proj_h / calib_h)
        self.proj_T = proj_T
        self.proj_R = proj_R
        self.image_folder = image_folder
        self._ctx = ctx
        self._calib_sent = False

    # -- plumbing
    def _context(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def _send_calibration(self):
        if self.cam_mtx is None or self.proj_mtx is None or self.proj_R is None or self.proj_T is None:
            raise ValueError("triangulate() needs cam_mtx, proj_mtx, proj_R and proj_T")
        self._context().set_calibration(self.cam_mtx, self.cam_dist, self.proj_mtx, self.proj_dist, self.proj_R, self.proj_T)
        self._calib_sent = True

    # -- reference API
    def get_cam_proj_pts(self, img_white=None, order="x"):
        """Reference triangulate.py:39-71 -> (cam_pts f32 [M,2], proj_pts f32 [M,2], colors f64 [M,3]).

        x-major point order like the reference (``order='row'`` gives row-major).  The reference
        raises UnboundLocalError when ``img_white`` is None (:71); here colors is None instead.  With no decodable pixel at all
        the reference's ``np.array([], dtype=np.float32)`` has shape (0,), not (0, 2): reproduced (:66-69).
        """
        o = _native.ORDER_X if order == "x" else _native.ORDER_ROW
        cam, proj, col = self._context().cam_proj_pts(self.h_pixels, self.v_pixels, (self.cam_w, self.cam_h),
                                                      (self.proj_w, self.proj_h), img_white, o)
        if len(cam) == 0:
            cam, proj = cam.reshape(0), proj.reshape(0)
            col = None if col is None else col.reshape(0)
        return cam, proj, col

    def triangulate(self, cam_pts, proj_pts, exact=True):
        """Reference triangulate.py:73-97 -> float64 (3,M), camera-centred, projector axes.

        ``exact=True`` evaluates acos/sin like the reference; ``exact=False`` uses the algebraically
        identical sqrt form (agrees to ~1e-12 relative away from degenerate geometry).
        """
        self._send_calibration()
        return self._context().triangulate(cam_pts, proj_pts, _native.TRI_EXACT if exact else _native.TRI_ALGEBRAIC)

    def filter_3d_pts(self, Pts, colors, threshold=.5):
        """Reference triangulate.py:99-122: keep |X|,|Y|,|Z| < threshold (strict), order preserved."""
        return self._context().filter_3d_pts(Pts, colors, threshold)


class Triangulation(Triangulate):
    """``Triangulate`` plus the fused ``compute()``: correspondences -> XYZ (-> box filter) in ONE device-resident call."""

    def compute(self, img_white=None, threshold=None, exact=True, order="x"):
        """What src/4-triangulate.py:62-71 does with a ``Triangulate`` -- ``get_cam_proj_pts(img_white)`` (triangulate.py:39-71), then
        ``triangulate`` (:73-97), then, with a ``threshold``, ``filter_3d_pts`` (:99-122) -- as one call: the maps go up once, the three
        kernels of the three methods run back to back in HBM, the kept points and colours come down once (slgc_compute_count / _fetch).
        Returns ``(pts, colors)`` exactly as the three-call composition does: float64 (3,M) x-major, float64 [M,3] or None; bit-identical
        to it (same kernels, same order).  ``compute_by_calls`` is that composition, kept for comparison."""
        self._send_calibration()
        o = _native.ORDER_X if order == "x" else _native.ORDER_ROW
        pts, colors, n_raw = self._context().compute(self.h_pixels, self.v_pixels, (self.cam_w, self.cam_h), (self.proj_w, self.proj_h),
                                                     img_white, threshold, o, _native.TRI_EXACT if exact else _native.TRI_ALGEBRAIC)
        if n_raw == 0 and colors is not None:
            colors = colors.reshape(0)          # the reference's empty colour array has shape (0,) (triangulate.py:66-69), and :121 keeps it
        return pts, colors

    def compute_by_calls(self, img_white=None, threshold=None, exact=True, order="x"):
        """The three methods one after the other (three host round trips): the definition ``compute`` is tested against."""
        cam_pts, proj_pts, colors = self.get_cam_proj_pts(img_white, order=order)
        pts = self.triangulate(cam_pts, proj_pts, exact=exact)
        if threshold is not None:
            if colors is None:
                pts, _ = self.filter_3d_pts(pts, None, threshold)
            else:
                pts, colors = self.filter_3d_pts(pts, colors, threshold)
        return pts, colors
