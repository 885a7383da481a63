"""MI355X-native drop-in for the hot path of guillaume-charron/3DScanner-GrayCode.

Same module paths as the reference's ``scanner`` package for the functions on the decode +
triangulation path (``scanner.grayCode.decode_codes``, ``scanner.triangulation``); everything
runs as hand-written HIP kernels behind libslgc.so (ctypes, see ``_native``).  Camera I/O,
calibration and visualisation are out of scope and stay with the reference.
"""
from . import _native  # noqa: F401
from ._native import Context, SlgcError, default_context  # noqa: F401
from .pipeline import scan_to_cloud  # noqa: F401,E402
