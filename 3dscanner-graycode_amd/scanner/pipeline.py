"""One-call form of the reference's two driver scripts (the "caller glue" rows of SURVEY.md section 2).

``scan_to_cloud`` does what ``src/3-capture_decode.py:75-100`` (get_codes per run, max-merge, gray_to_decimal) and
``src/4-triangulate.py:50-71`` (Triangulate, get_cam_proj_pts, triangulate, filter_3d_pts) do between them, with ONE
host->device upload of the image stacks and no intermediate round trips: same outputs (int64 maps, x-major point order,
float64 (3,M) points, float64 colours / 255).
"""
from __future__ import annotations

import numpy as np

from . import _native
from ._native import default_context

__all__ = ["scan_to_cloud"]


def scan_to_cloud(runs, cam_mtx, cam_dist, proj_size, proj_calib_size, proj_mtx, proj_dist, proj_R, proj_T, img_white=None,
                  threshold=None, eps=1, m=10, exact=True, order="x", ctx=None, return_lists=False):
    """runs: one ``[N,H,W]`` stack or a sequence / ``[R,N,H,W]`` of runs (uint8 or float64).

    Unlike ``Triangulate.__init__`` this does NOT modify ``proj_mtx``: the row scaling of triangulate.py:28-33 is applied
    to a copy.  Returns a dict with ``pts`` (3,M) float64, ``colors`` (or None), ``h_pixels``, ``v_pixels`` and, with
    ``return_lists``, the unfiltered ``cam_pts`` / ``proj_pts``.
    """
    c = ctx or default_context()
    pk = np.array(proj_mtx, dtype=np.float64, copy=True)
    pk[0, :] = pk[0, :] * (proj_size[0] / proj_calib_size[0])
    pk[1, :] = pk[1, :] * (proj_size[1] / proj_calib_size[1])
    c.set_calibration(cam_mtx, cam_dist, pk, proj_dist, proj_R, proj_T)
    return c.pipeline(runs, proj_size, img_white=img_white, threshold=threshold, eps=eps, m=m,
                      order=_native.ORDER_X if order == "x" else _native.ORDER_ROW,
                      mode=_native.TRI_EXACT if exact else _native.TRI_ALGEBRAIC, want_lists=return_lists)
