"""Point-cloud post-processing of ``scanner/utils/visualize.py:91-113`` of the reference, without the GUI:
statistical outlier removal (the k-NN arithmetic on the GPU) and the PLY file.

The reference delegates both to Open3D 0.17 (``remove_statistical_outlier(nb_neighbors=20, std_ratio=0.5)``,
``o3d.io.write_point_cloud``).  Open3D is third-party and not installable in the build container, so parity with it is
UNPINNED; what is restated is its published algorithm: per point the mean distance to its ``nb_neighbors`` nearest points
(the KD-tree query returns the point itself, distance 0), then keep points whose mean distance is below
``mean + std_ratio * std`` (sample standard deviation).  The interactive viewer (:115-131) is out of scope.
"""
from __future__ import annotations

import os

import numpy as np

from ._native import default_context

__all__ = ["remove_statistical_outlier", "write_ply", "save_point_cloud"]


def _as_points(pts):
    p = np.asarray(pts)
    if p.ndim != 2:
        raise ValueError("points must be (3,M) or (M,3)")
    if p.shape[0] == 3 and p.shape[1] != 3:
        p = p.T
    if p.shape[1] != 3:
        raise ValueError("points must be (3,M) or (M,3)")
    return p


def remove_statistical_outlier(pts, nb_neighbors=20, std_ratio=0.5, ctx=None):
    """-> (inlier_points float64 [M',3], ind int64 [M']).  ``pts`` is (3,M) as ``Triangulate.triangulate`` returns it, or (M,3).

    Like the reference's call chain the coordinates are rounded to float32 first (``o3c.Tensor(Pts.T, o3c.float32)``,
    visualize.py:98) and distances are evaluated in float64."""
    p32 = np.ascontiguousarray(_as_points(pts), dtype=np.float32)
    if nb_neighbors < 1 or std_ratio <= 0:
        raise ValueError("nb_neighbors must be >= 1 and std_ratio > 0")
    if len(p32) == 0:
        return np.zeros((0, 3)), np.zeros(0, np.int64)
    avg = (ctx or default_context()).knn_mean_distance(p32, nb_neighbors)
    valid = avg > 0
    n_valid = int(valid.sum())
    if n_valid < 2:
        return p32.astype(np.float64), np.arange(len(p32), dtype=np.int64)
    cloud_mean = avg[valid].sum() / n_valid
    std = np.sqrt(((avg[valid] - cloud_mean) ** 2).sum() / (n_valid - 1))          # Bessel's correction, as Open3D
    ind = np.nonzero(valid & (avg < cloud_mean + std_ratio * std))[0].astype(np.int64)
    return p32[ind].astype(np.float64), ind


def write_ply(path, points, colors=None):
    """Binary little-endian PLY in the layout Open3D writes for a legacy PointCloud: double x/y/z, then uchar red/green/blue
    (colour * 255, clamped and rounded) when colours are present."""
    p = np.ascontiguousarray(_as_points(points), dtype="<f8")
    n = len(p)
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    header = ["ply", "format binary_little_endian 1.0", "comment Created by slgc (layout of Open3D write_point_cloud)",
              f"element vertex {n}", "property double x", "property double y", "property double z"]
    c8 = None
    if colors is not None:
        c = np.asarray(colors, dtype=np.float64)
        if c.shape != (n, 3):
            raise ValueError("colors must be (M,3)")
        c8 = np.round(np.clip(c, 0.0, 1.0) * 255.0).astype(np.uint8)
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        header += ["property uchar red", "property uchar green", "property uchar blue"]
    header.append("end_header")
    rec = np.empty(n, dtype=np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = p[:, 0], p[:, 1], p[:, 2]
    if c8 is not None:
        rec["red"], rec["green"], rec["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())
    return n


def save_point_cloud(Pts, colors, save_to, nb_neighbors=20, std_ratio=0.5, ctx=None):
    """What ``plot_point_cloud`` (visualize.py:91-113) does before it opens the viewer: outlier removal, then ``cloud.ply``
    in ``save_to``.  Returns (inlier_points, inlier_colors, ind)."""
    pts, ind = remove_statistical_outlier(Pts, nb_neighbors, std_ratio, ctx=ctx)
    col = None if colors is None else np.asarray(colors)[ind]
    os.makedirs(save_to, exist_ok=True)
    write_ply(os.path.join(save_to, "cloud.ply"), pts, col)
    return pts, col, ind
