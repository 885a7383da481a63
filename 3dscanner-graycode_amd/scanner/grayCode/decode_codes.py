"""Drop-in for ``scanner/grayCode/decode_codes.py`` of the reference (hot functions only).

Same names, argument meaning and return shapes/dtypes as the reference; the arithmetic runs in
HIP kernels (csrc/decode.hip) through the ctypes C-ABI.  Inputs may be the reference's float64
``[N,H,W]`` stack (src/3-capture_decode.py:68-70) or a uint8 stack (8x less PCIe traffic; a uint8
stack means "these grey levels", i.e. the reference called with ``images.astype(float64)``).

Added on top of the reference's names: :func:`decode` fuses
get_codes -> run merge -> gray_to_decimal (src/3-capture_decode.py:75-100) into one call, and
:func:`codes_to_pixels` is that driver tail alone.
"""
from __future__ import annotations

from .._native import default_context

__all__ = ["get_direct_indirect", "get_is_lit", "get_codes", "gray_decode", "gray_to_decimal",
           "codes_to_pixels", "decode"]


def get_direct_indirect(images, ctx=None):
    """Direct / global illumination, reference decode_codes.py:90-122 -> (L_d, L_g) float64 [H,W]."""
    return (ctx or default_context()).direct_indirect(images)


def get_is_lit(images, L_d, L_g, eps=1, m=10, ctx=None):
    """Robust per-bit classification, reference decode_codes.py:125-186 -> (h_codes, v_codes) int8 [L,H,W].

    ``m`` is accepted for signature parity; in the reference its rule (:169-170) rewrites the
    value the arrays already hold, so it has no effect there either.
    """
    return (ctx or default_context()).is_lit(images, L_d, L_g, eps, m)


def get_codes(images, ctx=None):
    """Reference decode_codes.py:231-248: get_is_lit(images, *get_direct_indirect(images))."""
    return (ctx or default_context()).codes(images, 1, 10)


def gray_decode(n):
    """Gray code -> integer (reference decode_codes.py:189-207).  Scalar host helper."""
    n = int(n)
    shift = n >> 1
    while shift:
        n ^= shift
        shift >>= 1
    return n


def gray_to_decimal(gray_code_list):
    """One pixel's code list -> projector coordinate (reference decode_codes.py:209-229):
    -1 if any entry is -1, else MSB-first bits -> Gray -> binary.  Scalar host helper; use
    :func:`codes_to_pixels` / :func:`decode` for whole images."""
    word = 0
    for k in range(len(gray_code_list)):
        c = int(gray_code_list[k])
        if c == -1:
            return -1
        if c not in (0, 1):
            raise ValueError(f"invalid literal for int() with base 2: {c!r}")
        word = (word << 1) | c
    return gray_decode(word)


def codes_to_pixels(h_codes, v_codes, ctx=None):
    """Driver tail of src/3-capture_decode.py:95-100 on the GPU: ``h_codes``/``v_codes`` are
    ``[L,H,W]`` or ``[R,L,H,W]`` (R runs, max-merged) -> (h_pixels, v_pixels) int64 [H,W]."""
    return (ctx or default_context()).codes_to_pixels(h_codes, v_codes)


def decode(images_or_runs, eps=1, m=10, ctx=None):
    """Fused decode: one ``[N,H,W]`` stack or several runs (sequence / ``[R,N,H,W]``) ->
    (h_pixels, v_pixels) int64 [H,W], -1 where undecodable.  Valid mask = (h != -1) & (v != -1)."""
    return (ctx or default_context()).decode(images_or_runs, eps, m)
