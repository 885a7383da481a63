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

import numpy as np

from .._native import default_context

__all__ = ["read_images", "read_images_order", "remove_bad_images", "keep_from_counts", "to_gray", "get_direct_indirect", "get_is_lit", "get_codes", "gray_decode", "gray_to_decimal",
           "codes_to_pixels", "decode"]


def read_images_order(file_names):
    """File order of the reference's read_images (decode_codes.py:22): ``sorted(os.listdir(folder), key=len)`` -- a stable
    sort by NAME LENGTH only, so frame_2.jpg sorts before frame_10.jpg but equal-length names keep listdir order.  The image
    decoding itself (cv2.imread) stays with the caller: JPEG decoding is outside this build's scope."""
    return sorted(file_names, key=len)


def read_images(folder, dtype=np.float64):
    """Reference decode_codes.py:6-32: every image of ``folder`` in ``sorted(os.listdir(folder), key=len)`` order ->
    ``(images [n,H,W,3] BGR, file_names)`` -- float64 like the reference's ``np.empty`` buffer (``dtype=np.uint8`` keeps the bytes).

    Host-side ingest helper, like the reference's (cv2.imread runs on the CPU there too): the JPEG decoding is Pillow's here -- the ROCm
    image ships no rocJPEG, and OpenCV is not installed -- so byte-for-byte parity with ``cv2.imread`` (another libjpeg build, another
    IDCT) is UNPINNED; the file order, shapes, channel order and return types are the reference's.  Feed the images to
    ``Context.to_gray`` (GPU BGR -> grey, uint8 stack) and then to ``decode``."""
    import os
    try:
        from PIL import Image
    except ImportError as e:                                   # no silent fallback: say what is missing
        raise ImportError("read_images needs Pillow to decode the image files (JPEG decoding is not part of libslgc)") from e
    names = read_images_order(os.listdir(folder))
    images = None
    for i, name in enumerate(names):
        with Image.open(os.path.join(folder, name)) as im:
            bgr = np.asarray(im.convert("RGB"), dtype=np.uint8)[:, :, ::-1]          # cv2.imread returns BGR
        if images is None:
            images = np.empty((len(names),) + bgr.shape, dtype=dtype)               # :28
        images[i] = bgr
    if images is None:
        images = np.empty((0, 0, 0, 3), dtype=dtype)
    return images, np.array(names)


def keep_from_counts(counts):
    """The keep / drop rule of decode_codes.py:52-66 over the change counts ``counts[j]`` = number of elements that differ by more than 50
    between frames j and j + 1 (``len(np.argwhere(cv2.absdiff(a, b) > 50))``): indexes of the frames to keep.  Separate from the counting so
    that a capture already in HBM (``Context.frame_diff_counts_dev``) goes through the same rule."""
    e = [int(x) for x in counts]
    n = len(e) + 1
    kept = []

    def far_from_last(*idx):            # the frame kept last is none of idx (or nothing is kept yet)
        return not kept or kept[-1] not in idx

    # Window (e[i], e[i+1], e[i+2]) slides over the change counts (the reference walks images[2:-1], :52-66):
    #   counts rising across the window  -> frame i+1 is the settled frame before a transition (:56-58)
    #   middle count is the window's low -> frame i+2 sits in a quiet trough (:59-63); its two counts are then spent (-1) so the
    #                                       next two windows cannot pick the same trough again
    # A frame is never kept twice, nor directly next to the frame kept last.  The two cases exclude each other (rising needs
    # e[i+1] > e[i], a trough needs e[i+1] <= e[i]).
    for i in range(n - 3):
        a, b, c = e[i], e[i + 1], e[i + 2]
        if a < b < c:
            if i + 1 not in kept and far_from_last(i, i + 2):
                kept.append(i + 1)
        elif b <= a and b <= c and i + 2 not in kept:
            if far_from_last(i + 1, i + 3):
                kept.append(i + 2)
                e[i + 1] = e[i + 2] = -1
    return kept


def remove_bad_images(images, ctx=None):
    """Indexes of the frames to keep (transition frames dropped), reference decode_codes.py:34-68.

    The per-pair ``len(np.argwhere(cv2.absdiff(a, b) > 50))`` counts are one GPU reduction over all consecutive pairs
    (csrc/ingest.hip); the keep/drop rule of :56-66 is :func:`keep_from_counts`.  ``images`` is the reference's ``[n,H,W,3]`` (or
    ``[n,H,W]``) array, float64 or uint8."""
    return keep_from_counts((ctx or default_context()).frame_diff_counts(images, 50))


def to_gray(images, ctx=None, coeff_bits=15):
    """Reference decode_codes.py:70-87: BGR -> grey per frame, returned as the float64 ``[n,H,W]`` array the reference
    returns.  The arithmetic is OpenCV's 8-bit fixed-point luma (see csrc/ingest.hip); OpenCV 4.8.0.76 is not available
    in the build container, so this function's parity with cv2.cvtColor is UNPINNED.  For the decode path prefer
    ``Context.to_gray`` which keeps the uint8 stack."""
    return (ctx or default_context()).to_gray(images, coeff_bits).astype(np.float64)


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
