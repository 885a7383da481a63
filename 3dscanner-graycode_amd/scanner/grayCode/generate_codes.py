"""Pattern generator (reference scanner/grayCode/generate_codes.py:5-81), host side.

Runs once per projector; it defines the frame order ("wire format") the decode kernels consume:
images[0] black, [1] white, then column-code bit j (MSB first) at 2+2j, row-code bit j at
3+2(L-1-j), and the inverses 2L frames later.  The fullscreen display loop (:83-120) is GUI and
out of scope.
"""
from __future__ import annotations

import numpy as np

__all__ = ["get_gray_codes", "get_image_sequence"]


def get_gray_codes(width, height):
    """uint8 [max(w,h), n_bits] table of g = i ^ (i >> 1), MSB first; n_bits = ceil(log2(max(w,h)))."""
    size = max(width, height)
    n_bits = int(np.ceil(np.log2(size)))
    i = np.arange(size, dtype=np.uint16)
    g = (i >> 1) ^ i
    shifts = np.arange(n_bits - 1, -1, -1, dtype=np.uint16)
    return ((g[:, None] >> shifts[None, :]) & 1).astype(np.uint8)


def get_image_sequence(gray_codes, width, height):
    """uint8 [4*n_bits+2, height, width] stack of 0/255 frames in the reference's order."""
    n_codes, L = gray_codes.shape
    seq = np.zeros((4 * L + 2, height, width), dtype=np.uint8)
    seq[1] = 255
    stripe = width // n_codes          # 1 for width >= height; 0 otherwise (reference :60 behaves the same)
    for j in range(L):
        id_v, id_h = 2 * j + 2, 2 * (L - (j + 1)) + 3
        cols = np.zeros(width, dtype=np.uint8)
        span = min(width, n_codes * stripe)
        cols[:span] = np.repeat(gray_codes[:, j], stripe)[:span] * 255
        seq[id_v, :, :span] = cols[None, :span]
        seq[id_v + 2 * L, :, :span] = 255 - seq[id_v, :, :span]
        n_rows_codes = min(height, n_codes)
        rspan = min(height, n_rows_codes * stripe)
        rows = np.repeat(gray_codes[:n_rows_codes, j], stripe)[:rspan] * 255
        seq[id_h, :rspan, :] = rows[:, None]
        seq[id_h + 2 * L, :rspan, :] = 255 - seq[id_h, :rspan, :]
    return seq
