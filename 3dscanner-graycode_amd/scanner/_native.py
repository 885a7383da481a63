"""ctypes binding of libslgc.so (C-ABI in include/slgc.h) -- the only door to the HIP kernels.

There is no CPU fallback: if the library is missing or no HIP device is usable, every compute
entry point raises.  No torch, no numpy-side arithmetic on the data path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SLGC_LIB", os.path.join(_PKG_ROOT, "lib", "libslgc.so"))

U8, F64 = 0, 1
ORDER_X, ORDER_ROW = 0, 1
WIRE_MAX_CODE_BITS = 11           # slgc.h SLGC_WIRE_MAX_CODE_BITS
TRI_EXACT, TRI_ALGEBRAIC, TRI_DIRECT, TRI_SPLIT = 0, 1, 2, 4
UNIQUE_ID_BYTES = 128
BUS_ID_BYTES = 16


class SlgcError(RuntimeError):
    pass


_vp, _i, _i64, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol include/slgc.h declares
SIGNATURES = {
    "slgc_version": (_i, []),
    "slgc_backend": (C.c_char_p, []),
    "slgc_strerror": (C.c_char_p, [_i]),
    "slgc_device_count": (_i, []),
    "slgc_create": (_i, [_i, C.POINTER(_vp)]),
    "slgc_destroy": (_i, [_vp]),
    "slgc_last_error": (C.c_char_p, [_vp]),
    "slgc_synchronize": (_i, [_vp]),
    "slgc_last_input_path": (_i, [_vp]),
    "slgc_last_scan_path": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "slgc_last_list_kernel": (_i, [_vp]),
    "slgc_last_scan_ragged": (_i, [_vp]),
    "slgc_tune": (_i, [_vp, C.c_char_p, _i]),
    "slgc_device_name": (_i, [_vp, C.c_char_p, _i]),
    "slgc_device_pci_bus_id": (_i, [_vp, C.c_char_p, _i]),
    "slgc_direct_indirect": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "slgc_is_lit": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _d, _d, _vp, _vp]),
    "slgc_codes": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _vp, _vp]),
    "slgc_codes_to_pixels": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "slgc_decode": (_i, [_vp, C.POINTER(_vp), _i, _i, _i, _i, _i, _d, _d, _vp, _vp]),
    "slgc_set_calibration": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "slgc_cam_proj_pts_count": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, C.POINTER(_i64)]),
    "slgc_cam_proj_pts_fetch": (_i, [_vp, _vp, _vp, _vp]),
    "slgc_triangulate": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "slgc_undistort_points": (_i, [_vp, _i, _vp, _i64, _vp]),
    "slgc_filter_count": (_i, [_vp, _vp, _vp, _i64, _d, C.POINTER(_i64)]),
    "slgc_filter_fetch": (_i, [_vp, _vp, _vp]),
    "slgc_compute_count": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _d, C.POINTER(_i64), C.POINTER(_i64)]),
    "slgc_compute_fetch": (_i, [_vp, _vp, _vp]),
    "slgc_to_gray": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "slgc_to_gray_dev": (_i, [_vp, _vp, _sz, _i, _vp]),
    "slgc_tiled_stack_bytes": (_i, [_i, _sz, _i, C.POINTER(_sz)]),
    "slgc_tile_stack_dev": (_i, [_vp, _vp, _sz, _i, _sz, _i, _vp]),
    "slgc_to_gray_tiled_dev": (_i, [_vp, _vp, _i, _sz, _i, _i, _vp]),
    "slgc_frame_diff_counts": (_i, [_vp, _vp, _i, _i, _sz, _d, _vp]),
    "slgc_frame_diff_counts_dev": (_i, [_vp, _vp, _i, _i, _sz, _d, _vp]),
    "slgc_knn_mean_distance": (_i, [_vp, _vp, _i64, _i, _vp]),
    "slgc_knn_mean_distance_dev": (_i, [_vp, _vp, _i64, _i, _vp]),
    "slgc_move_only_dev": (_i, [_vp, _vp, C.c_size_t, _i, C.c_size_t, _vp, _vp, _vp]),
    "slgc_pipeline_count": (_i, [_vp, C.POINTER(_vp), _i, _i, _i, _i, _i, _d, _d, _i, _i, _vp, _i, _i, _d, C.POINTER(_i64)]),
    "slgc_pipeline_fetch": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), _vp, _vp]),
    "slgc_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "slgc_dev_free": (_i, [_vp, _vp]),
    "slgc_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "slgc_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "slgc_dev_memset": (_i, [_vp, _vp, _i, _sz]),
    "slgc_decode_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _d, _d, _vp, _vp, _i]),
    "slgc_scan_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp, _vp]),
    "slgc_scan_bgr_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp, _vp]),
    "slgc_decode_bgr_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _i, _d, _d, _vp, _vp]),
    "slgc_selftest_thresholds": (_i, [_vp, _i, _i, _i, C.POINTER(C.c_uint64)]),
    "slgc_selftest_classify": (_i, [_vp, _i, C.POINTER(C.c_uint64)]),
    "slgc_triangulate_maps_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slgc_cloud_lists_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "slgc_cloud_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slgc_cloud32_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slgc_compact_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "slgc_compact_records_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "slgc_pack_hv24_dev": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "slgc_unpack_hv24_dev": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "slgc_triangulate_wire_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "slgc_synth_scene_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _i, C.c_uint32, _i, _i]),
    "slgc_synth_physical_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, C.c_uint32, _i, _vp, _vp, _vp]),
    "slgc_synth_physical_ex_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, C.c_uint32, _i, _i, _i, _d, _vp, _vp, _vp]),
    "slgc_synth_bgr_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _i, _vp, _sz]),
    "slgc_synth_uniform_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _i, C.c_uint32]),
    "slgc_event_record": (_i, [_vp, _i]),
    "slgc_event_elapsed_ms": (_i, [_vp, _i, _i, C.POINTER(C.c_float)]),
    "slgc_prof_begin": (_i, [_vp, _i, _i]),
    "slgc_prof_end": (_i, [_vp, C.POINTER(_d), C.POINTER(_i)]),
    "slgc_prof_samples": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "slgc_build_ray_tables_dev": (_i, [_vp, _i, _i, _i, _i, _i]),
    "slgc_ray_table_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_d)]),
    "slgc_guard_count_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "slgc_comm_unique_id": (_i, [_vp]),
    "slgc_comm_init": (_i, [_vp, _i, _i, _vp]),
    "slgc_comm_destroy": (_i, [_vp]),
    "slgc_comm_barrier": (_i, [_vp]),
    "slgc_comm_allreduce_max_f64": (_i, [_vp, C.POINTER(_d)]),
    "slgc_comm_allgather_i64": (_i, [_vp, _i64, C.POINTER(_i64)]),
    "slgc_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "slgc_comm_allgather_bus_ids": (_i, [_vp, C.c_char_p, C.POINTER(_i)]),
    "slgc_comm_allgatherv": (_i, [_vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "slgc_comm_allgatherv_begin": (_i, [_vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _i]),
    "slgc_comm_allgatherv_pair_begin": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _i]),
    "slgc_comm_wait": (_i, [_vp, _i]),
    "slgc_direct_init": (_i, [_vp, _i, _i, C.c_char_p]),
    "slgc_direct_destroy": (_i, [_vp]),
    "slgc_direct_register": (_i, [_vp, _vp, _sz]),
    "slgc_direct_unregister": (_i, [_vp, _vp]),
    "slgc_direct_allgatherv_begin": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(C.POINTER(_i64)), C.POINTER(C.POINTER(_i64)), _i]),
    "slgc_direct_wait": (_i, [_vp, _i]),
    "slgc_direct_release": (_i, [_vp, _i, C.POINTER(_vp)]),
    "slgc_direct_barrier": (_i, [_vp]),
    "slgc_direct_allgather_i64": (_i, [_vp, _i64, C.POINTER(_i64)]),
    "slgc_shard_band": (_i, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "slgc_scan_batch_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp]),
    "slgc_scan_sharded_dev": (_i, [_vp, _vp, _i, _sz, _sz, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load libslgc.so once; raise (loudly) if it is not there."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SlgcError(
                    f"HIP library not found at {LIB_PATH}: build it with `make -C 3dscanner-graycode_amd` "
                    "(or __graft_entry__.build()).  There is no CPU fallback.")
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def shard_band(H: int, nranks: int, rank: int):
    """(row0, rows) of a rank's band: slgc_shard_band, the plan slgc_scan_sharded_dev uses."""
    r0, rn = C.c_int(), C.c_int()
    if lib().slgc_shard_band(H, nranks, rank, C.byref(r0), C.byref(rn)):
        raise ValueError("bad H / nranks / rank")
    return r0.value, rn.value


def device_count() -> int:
    return int(lib().slgc_device_count())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _ResultPool:
    """Recycled memory behind the large arrays the host-buffer API returns.  A device-to-host copy into memory nobody has touched runs
    at the page-fault rate of the host (11-15 GB/s measured; the link does 56), into pages that exist already at the rate of the link --
    page-locked or not (tools/ubench/host_alloc.hip).  So the binding keeps the buffers of results the caller has let go of and hands
    them out again: a result array is an ordinary writable NumPy array over such a buffer, and when the array and every view of it are
    gone the buffer returns to the pool instead of to the operating system.  A script that handles one scan sees exactly np.empty's cost;
    a loop gets its results at the rate of the link from its second or third pass on, with nothing page-locked and no warm-up to pay.
    Buffers are cached up to SLGC_RESULT_POOL_MB (default 1024: two 4096x3000 scans' worth of int64 maps); ``result_pool_clear()`` returns
    everything cached to the operating system; SLGC_RESULT_POOL=0 returns plain np.empty arrays.  A pooled array does not own its memory
    (``arr.flags.owndata`` is False, ``ndarray.resize`` refuses): copy it if you need an owning array."""

    MIN_BYTES = 8 << 20
    GRANULE = 2 << 20

    def __init__(self):
        self.free = {}                      # rounded size -> [uint8 buffer, ...]
        self.cached = 0
        self.lock = threading.Lock()
        self.enabled = os.environ.get("SLGC_RESULT_POOL", "1") != "0"
        self.limit = int(os.environ.get("SLGC_RESULT_POOL_MB", "1024")) << 20

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if not self.enabled or nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype)
        size = -(-nbytes // self.GRANULE) * self.GRANULE
        with self.lock:
            bufs = self.free.get(size)
            base = bufs.pop() if bufs else None
            if base is not None:
                self.cached -= size
        if base is None:
            base = np.empty(size, np.uint8)
        raw = (C.c_char * nbytes).from_address(base.ctypes.data)
        weakref.finalize(raw, self._release, base, size)          # fires when the array and all its views are gone; holds `base` until then
        return np.frombuffer(raw, dtype=dtype).reshape(shape)

    def _release(self, base, size):
        with self.lock:
            if self.cached + size <= self.limit:
                self.free.setdefault(size, []).append(base)
                self.cached += size


    def clear(self):
        with self.lock:
            self.free.clear()
            self.cached = 0


_pool = _ResultPool()


def result_pool_clear():
    """Give the cached result buffers back to the operating system (they are re-created on demand)."""
    _pool.clear()


def result_pool_bytes() -> int:
    return _pool.cached


def _out(shape, dtype=np.float64):
    """A result array of the host-buffer API (see _ResultPool)."""
    return _pool.empty(shape, dtype)


class DeviceBuffer:
    """Device allocation owned by a Context (freed with it or by .free())."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._ck(lib().slgc_dev_alloc(ctx._h, self.nbytes, C.byref(p)))
        self.ptr = p.value
        ctx._buffers.append(self)

    def at(self, byte_offset: int) -> int:
        assert 0 <= byte_offset <= self.nbytes
        return self.ptr + int(byte_offset)

    def upload(self, arr: np.ndarray, byte_offset: int = 0):
        a = np.ascontiguousarray(arr)
        assert byte_offset + a.nbytes <= self.nbytes
        self.ctx._ck(lib().slgc_h2d(self.ctx._h, self.at(byte_offset), _ptr(a), a.nbytes))
        return self

    def download(self, shape, dtype, byte_offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert byte_offset + out.nbytes <= self.nbytes
        self.ctx._ck(lib().slgc_d2h(self.ctx._h, _ptr(out), self.at(byte_offset), out.nbytes))
        return out

    def zero(self):
        self.ctx._ck(lib().slgc_dev_memset(self.ctx._h, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self.ptr:
            self.ctx._ck(lib().slgc_dev_free(self.ctx._h, self.ptr))          # refused (SlgcError) while the buffer is registered with the direct exchange
            self.ptr = None
            if self in self.ctx._buffers:
                self.ctx._buffers.remove(self)


class CloudLists:
    """Device buffers of the reference-shaped product (slgc_cloud_lists_dev): capacity ``npix`` entries each."""

    def __init__(self, ctx: "Context", npix: int, colors: bool = True, points: bool = True, lists: bool = True, f32: bool = False):
        self.ctx, self.npix, self.f32 = ctx, int(npix), bool(f32)      # f32: points / colours as float32 (slgc_cloud32_dev only; NOT the reference's dtypes)
        self.cam = ctx.alloc(max(16, self.npix * 8)) if lists else None           # lists=False (slgc_cloud_dev only): points + colours, no correspondence lists
        self.proj = ctx.alloc(max(16, self.npix * 8)) if lists else None
        self.pts = ctx.alloc(max(16, self.npix * (12 if f32 else 24))) if points else None
        self.colors = ctx.alloc(max(16, self.npix * (12 if f32 else 24))) if colors else None
        self.count = ctx.alloc(8).zero()

    def total(self) -> int:
        """M of the last build (synchronises)."""
        self.ctx.synchronize()
        return int(self.count.download((1,), np.uint64)[0])

    def download(self):
        """-> (cam_pts f32 [M,2], proj_pts f32 [M,2], Pts f64 (3,M) or None, colors f64 [M,3] or None), as the reference returns them."""
        M = self.total()
        cam = self.cam.download((M, 2), np.float32) if self.cam is not None else None
        proj = self.proj.download((M, 2), np.float32) if self.proj is not None else None
        ft = np.float32 if self.f32 else np.float64
        pts = self.pts.download((3, M), ft) if self.pts is not None else None
        col = self.colors.download((M, 3), ft) if self.colors is not None else None
        return cam, proj, pts, col

    def free(self):
        for b in (self.cam, self.proj, self.pts, self.colors, self.count):
            if b is not None:
                b.free()


class Context:
    """(device, HIP stream, workspace).  One per thread; calls on one context are serialised by the caller."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        rc = lib().slgc_create(int(device), C.byref(h))
        if rc:
            raise SlgcError(f"slgc_create(device={device}) failed: {lib().slgc_strerror(rc).decode()} "
                            "(an MI355X / HIP device is required; there is no CPU fallback)")
        self._h = h
        self.device = device
        self._buffers = []

    # ---- plumbing
    def _ck(self, rc: int):
        if rc:
            msg = lib().slgc_last_error(self._h).decode(errors="replace")
            text = f"{lib().slgc_strerror(rc).decode()}: {msg}"
            raise (ValueError if rc == -1 else SlgcError)(text)

    def close(self):
        if self._h:
            lib().slgc_direct_destroy(self._h)          # closes the peer mappings: registered buffers become ordinary ones (callers barrier before closing, see slgc.h)
            for b in list(self._buffers):
                b.free()
            lib().slgc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def synchronize(self):
        self._ck(lib().slgc_synchronize(self._h))

    def tune(self, name: str, value: int):
        """Same-process A/B knobs (slgc_tune): results never change."""
        self._ck(lib().slgc_tune(self._h, name.encode(), int(value)))

    def last_input_path(self) -> int:
        """0 = uint8 stack as given, 1 = float64 stack narrowed to uint8 on the host, 2 = float64 kernel (slgc_last_input_path)."""
        return int(lib().slgc_last_input_path(self._h))

    SCAN_PATHS = {0: "none", 1: "fused", 2: "split", 3: "split-ragged", 4: "batch-fused", 5: "cloud", 6: "fused-bgr"}

    def last_scan_path(self) -> dict:
        """Which kernels the last scan_dev / scan_batch_dev call launched (slgc_last_scan_path): {"path": "fused" | "split" |
        "split-ragged" | "batch-fused" | "cloud" | "fused-bgr" | "none", "ns_frames": 42 | 44 | 46 | 0 (generic kernel), "node_table": bool, "guard": bool,
        "fallback_kernels": {"decode": bool, "triangulation": bool} (slgc_last_scan_ragged: byte-wide / per-pixel fallback kernels took part)}."""
        ns, nodes, guard = C.c_int(), C.c_int(), C.c_int()
        rc = lib().slgc_last_scan_path(self._h, C.byref(ns), C.byref(nodes), C.byref(guard))
        if rc < 0:
            self._ck(rc)
        rg = int(lib().slgc_last_scan_ragged(self._h))
        return {"path": self.SCAN_PATHS.get(rc, str(rc)), "ns_frames": int(ns.value), "node_table": bool(nodes.value), "guard": guard.value == 1,
                "fallback_kernels": {"decode": bool(rg & 1), "triangulation": bool(rg & 2)}}

    LIST_KERNELS = {0: "none", 1: "tile-runs", 2: "whole-lines"}

    def last_list_kernel(self) -> str:
        """Which scatter the last x-major list build launched (slgc_last_list_kernel): "tile-runs" | "whole-lines" | "none"."""
        rc = lib().slgc_last_list_kernel(self._h)
        if rc < 0:
            self._ck(rc)
        return self.LIST_KERNELS.get(rc, str(rc))

    def dev_memset(self, dptr: int, value: int, nbytes: int):
        self._ck(lib().slgc_dev_memset(self._h, dptr, int(value), int(nbytes)))

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self._ck(lib().slgc_device_name(self._h, buf, 256))
        return buf.value.decode()

    def device_pci_bus_id(self) -> str:
        buf = C.create_string_buffer(64)
        self._ck(lib().slgc_device_pci_bus_id(self._h, buf, 64))
        return buf.value.decode().lower()

    def event_record(self, idx: int):
        self._ck(lib().slgc_event_record(self._h, idx))

    def event_elapsed_ms(self, a: int, b: int) -> float:
        ms = C.c_float()
        self._ck(lib().slgc_event_elapsed_ms(self._h, a, b, C.byref(ms)))
        return float(ms.value)

    # ---- host-buffer entry points (reference-shaped)
    @staticmethod
    def _stack(images):
        st = np.asarray(images)
        if st.ndim != 3:
            raise ValueError("image stack must be [N,H,W]")
        if st.dtype == np.uint8:
            return np.ascontiguousarray(st), U8
        return np.ascontiguousarray(st, dtype=np.float64), F64

    def direct_indirect(self, images):
        st, dt = self._stack(images)
        N, H, W = st.shape
        ld, lg = _out((H, W)), _out((H, W))
        self._ck(lib().slgc_direct_indirect(self._h, _ptr(st), dt, N, H, W, _ptr(ld), _ptr(lg)))
        return ld, lg

    def is_lit(self, images, L_d, L_g, eps=1, m=10):
        st, dt = self._stack(images)
        N, H, W = st.shape
        L = int((N - 2) / 4)
        ld = np.ascontiguousarray(L_d, dtype=np.float64)
        lg = np.ascontiguousarray(L_g, dtype=np.float64)
        if ld.shape != (H, W) or lg.shape != (H, W):
            raise ValueError("L_d / L_g must be [H,W]")
        hc, vc = _out((L, H, W), np.int8), _out((L, H, W), np.int8)
        self._ck(lib().slgc_is_lit(self._h, _ptr(st), dt, N, H, W, _ptr(ld), _ptr(lg), float(eps), float(m), _ptr(hc), _ptr(vc)))
        return hc, vc

    def codes(self, images, eps=1, m=10):
        st, dt = self._stack(images)
        N, H, W = st.shape
        L = int((N - 2) / 4)
        hc, vc = _out((max(L, 0), H, W), np.int8), _out((max(L, 0), H, W), np.int8)
        self._ck(lib().slgc_codes(self._h, _ptr(st), dt, N, H, W, float(eps), float(m), _ptr(hc), _ptr(vc)))
        return hc, vc

    def codes_to_pixels(self, h_codes, v_codes):
        hc = np.ascontiguousarray(h_codes, dtype=np.int8)
        vc = np.ascontiguousarray(v_codes, dtype=np.int8)
        if hc.ndim == 3:
            hc, vc = hc[None], vc[None]
        if hc.shape != vc.shape or hc.ndim != 4:
            raise ValueError("codes must be [L,H,W] or [R,L,H,W], same shape for h and v")
        R, L, H, W = hc.shape
        hp, vp = _out((H, W), np.int64), _out((H, W), np.int64)
        self._ck(lib().slgc_codes_to_pixels(self._h, _ptr(hc), _ptr(vc), R, L, H, W, _ptr(hp), _ptr(vp)))
        return hp, vp

    def decode(self, runs, eps=1, m=10):
        """runs: [N,H,W] or a sequence / [R,N,H,W] array of runs (uint8 or float64)."""
        arr = runs if isinstance(runs, np.ndarray) else None
        if arr is not None and arr.ndim == 3:
            stacks = [arr]
        else:
            stacks = list(runs)
        prepared = [self._stack(s) for s in stacks]
        dts = {dt for _, dt in prepared}
        if len(dts) != 1 or len({s.shape for s, _ in prepared}) != 1:
            raise ValueError("all runs must share dtype and shape")
        dt = dts.pop()
        N, H, W = prepared[0][0].shape
        ptrs = (C.c_void_p * len(prepared))(*[s.ctypes.data for s, _ in prepared])
        hp, vp = _out((H, W), np.int64), _out((H, W), np.int64)
        self._ck(lib().slgc_decode(self._h, ptrs, dt, len(prepared), N, H, W, float(eps), float(m), _ptr(hp), _ptr(vp)))
        return hp, vp

    def set_calibration(self, cam_mtx, cam_dist, proj_mtx, proj_dist, proj_R, proj_T):
        def vec(a, n=None):
            v = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
            if n is not None and v.size != n:
                raise ValueError(f"expected {n} values, got {v.size}")
            return v
        ck, pk, r, t = vec(cam_mtx, 9), vec(proj_mtx, 9), vec(proj_R, 9), vec(proj_T, 3)
        cd = vec(cam_dist if cam_dist is not None else [])
        pd = vec(proj_dist if proj_dist is not None else [])
        self._ck(lib().slgc_set_calibration(self._h, _ptr(ck), _ptr(cd), cd.size, _ptr(pk), _ptr(pd), pd.size, _ptr(r), _ptr(t)))

    def cam_proj_pts(self, h_pixels, v_pixels, cam_size, proj_size, img_white=None, order=ORDER_X):
        cw, ch = int(cam_size[0]), int(cam_size[1])
        h = np.ascontiguousarray(np.asarray(h_pixels)[:ch, :cw], dtype=np.int64)
        v = np.ascontiguousarray(np.asarray(v_pixels)[:ch, :cw], dtype=np.int64)
        if h.shape != (ch, cw) or v.shape != (ch, cw):
            raise ValueError("h_pixels / v_pixels smaller than cam_size")
        wh = None
        if img_white is not None:
            wh = np.ascontiguousarray(np.asarray(img_white)[:ch, :cw, :3], dtype=np.uint8)
            if wh.shape != (ch, cw, 3):
                raise ValueError("img_white must be [H,W,3]")
        M = C.c_int64()
        self._ck(lib().slgc_cam_proj_pts_count(self._h, _ptr(h), _ptr(v), cw, ch, int(proj_size[0]), int(proj_size[1]),
                                               _ptr(wh), int(order), C.byref(M)))
        n = M.value
        cam, proj = _out((n, 2), np.float32), _out((n, 2), np.float32)
        col = _out((n, 3), np.float64) if wh is not None else None
        self._ck(lib().slgc_cam_proj_pts_fetch(self._h, _ptr(cam), _ptr(proj), _ptr(col)))
        return cam, proj, col

    def triangulate(self, cam_pts, proj_pts, mode=TRI_EXACT):
        a = np.ascontiguousarray(np.asarray(cam_pts, dtype=np.float32).reshape(-1, 2))
        b = np.ascontiguousarray(np.asarray(proj_pts, dtype=np.float32).reshape(-1, 2))
        if a.shape != b.shape:
            raise ValueError("cam_pts and proj_pts must have the same length")
        xyz = _out((3, len(a)), np.float64)
        self._ck(lib().slgc_triangulate(self._h, _ptr(a), _ptr(b), len(a), int(mode), _ptr(xyz)))
        return xyz

    def undistort_points(self, pts, projector: bool = False):
        """cv2.undistortPoints as triangulate.py:84 (camera, R = proj_R) / :85 (projector) call it; float32 [M,2] -> float32 [M,2]."""
        a = np.ascontiguousarray(np.asarray(pts, dtype=np.float32).reshape(-1, 2))
        out = np.empty_like(a)
        self._ck(lib().slgc_undistort_points(self._h, 1 if projector else 0, _ptr(a), len(a), _ptr(out)))
        return out

    def filter_3d_pts(self, pts, colors, threshold=0.5):
        x = np.ascontiguousarray(pts, dtype=np.float64)
        if x.ndim != 2 or x.shape[0] != 3:
            raise ValueError("Pts must be (3,M)")
        M = x.shape[1]
        c = None if colors is None else np.ascontiguousarray(colors, dtype=np.float64)
        if M == 0 and c is not None and c.size == 0:
            # a scan with no decodable pixel: the reference's get_cam_proj_pts hands on colours of shape (0,) (triangulate.py:66-69), and its
            # filter (:119-121) returns `colors[filter]` in whatever empty shape it was given
            return np.empty((3, 0), np.float64), c.copy()
        if c is not None and c.shape != (M, 3):
            raise ValueError("colors must be (M,3)")
        kept = C.c_int64()
        self._ck(lib().slgc_filter_count(self._h, _ptr(x), _ptr(c), M, float(threshold), C.byref(kept)))
        xo = _out((3, kept.value), np.float64)
        co = None if c is None else _out((kept.value, 3), np.float64)
        self._ck(lib().slgc_filter_fetch(self._h, _ptr(xo), _ptr(co)))
        return xo, co

    def compute(self, h_pixels, v_pixels, cam_size, proj_size, img_white=None, threshold=None, order=ORDER_X, mode=TRI_EXACT):
        """get_cam_proj_pts -> triangulate -> filter_3d_pts (triangulate.py:39-122) as one device-resident chain (slgc_compute_count /
        _fetch; set_calibration first): one upload of the maps (int16 on the link when they fit) and the white image, one download of
        the kept points (3,M) float64 and colours [M,3] float64 (None without img_white).  Returns (pts, colors, n_unfiltered)."""
        cw, ch = int(cam_size[0]), int(cam_size[1])
        h = np.ascontiguousarray(np.asarray(h_pixels)[:ch, :cw], dtype=np.int64)
        v = np.ascontiguousarray(np.asarray(v_pixels)[:ch, :cw], dtype=np.int64)
        if h.shape != (ch, cw) or v.shape != (ch, cw):
            raise ValueError("h_pixels / v_pixels smaller than cam_size")
        wh = None
        if img_white is not None:
            wh = np.ascontiguousarray(np.asarray(img_white)[:ch, :cw, :3], dtype=np.uint8)
            if wh.shape != (ch, cw, 3):
                raise ValueError("img_white must be [H,W,3]")
        M, raw = C.c_int64(), C.c_int64()
        thr = float("nan") if threshold is None else float(threshold)
        self._ck(lib().slgc_compute_count(self._h, _ptr(h), _ptr(v), cw, ch, int(proj_size[0]), int(proj_size[1]), _ptr(wh), int(order),
                                          int(mode), thr, C.byref(M), C.byref(raw)))
        pts = _out((3, M.value), np.float64)
        col = _out((M.value, 3), np.float64) if wh is not None else None
        self._ck(lib().slgc_compute_fetch(self._h, _ptr(pts), _ptr(col)))
        return pts, col, int(raw.value)

    def to_gray(self, images, coeff_bits=15):
        im = np.ascontiguousarray(np.asarray(images).astype(np.uint8, copy=False))
        if im.ndim != 4 or im.shape[3] != 3:
            raise ValueError("images must be [n,H,W,3]")
        n, H, W, _ = im.shape
        out = _out((n, H, W), np.uint8)
        self._ck(lib().slgc_to_gray(self._h, _ptr(im), n, H, W, int(coeff_bits), _ptr(out)))
        return out

    # ---- tile-interleaved stack layout (device-resident scans; an A/B kept reproducible, not a product layout: include/slgc_bench.h)
    @staticmethod
    def tiled_stack_bytes(N: int, npix: int, tile_log2: int) -> int:
        n = _sz()
        if lib().slgc_tiled_stack_bytes(int(N), int(npix), int(tile_log2), C.byref(n)):
            raise ValueError("tile_log2 must be 8..24")
        return int(n.value)

    def tile_stack_dev(self, d_planar: int, plane_stride: int, N: int, npix: int, tile_log2: int, d_tiled: int):
        self._ck(lib().slgc_tile_stack_dev(self._h, d_planar, int(plane_stride), int(N), int(npix), int(tile_log2), d_tiled))

    def to_gray_tiled_dev(self, d_bgr: int, n_frames: int, npix: int, tile_log2: int, d_tiled: int, coeff_bits: int = 15):
        self._ck(lib().slgc_to_gray_tiled_dev(self._h, d_bgr, int(n_frames), int(npix), int(coeff_bits), int(tile_log2), d_tiled))

    def frame_diff_counts(self, frames, thresh):
        fr = np.asarray(frames)
        fr = np.ascontiguousarray(fr) if fr.dtype == np.uint8 else np.ascontiguousarray(fr, dtype=np.float64)
        n = fr.shape[0]
        elems = int(np.prod(fr.shape[1:])) if n else 0
        out = np.zeros(max(n - 1, 0), np.int64)
        self._ck(lib().slgc_frame_diff_counts(self._h, _ptr(fr), U8 if fr.dtype == np.uint8 else F64, n, elems, float(thresh), _ptr(out)))
        return out

    def frame_diff_counts_dev(self, d_frames: int, n_frames, elems_per_frame, thresh, d_counts: int, dtype=U8):
        """remove_bad_images' counts for frames already in HBM (asynchronous; d_counts = n_frames - 1 uint64 in device memory)."""
        self._ck(lib().slgc_frame_diff_counts_dev(self._h, d_frames, int(dtype), int(n_frames), int(elems_per_frame), float(thresh), d_counts))

    def knn_mean_distance(self, pts, k=20):
        p = np.ascontiguousarray(np.asarray(pts, dtype=np.float32).reshape(-1, 3))
        out = _out(len(p), np.float64)
        self._ck(lib().slgc_knn_mean_distance(self._h, _ptr(p), len(p), int(k), _ptr(out)))
        return out

    def move_only_dev(self, d_stack: int, plane_stride: int, N: int, npix: int, d_h=None, d_v=None, d_xyz=None):
        """The movement yardstick (slgc_move_only_dev): a kernel that only moves one scan's bytes -- N planes in, maps / 12 B per pixel out."""
        self._ck(lib().slgc_move_only_dev(self._h, d_stack, int(plane_stride), int(N), int(npix), d_h, d_v, d_xyz))

    def knn_mean_distance_dev(self, d_pts: int, M: int, k: int, d_mean: int):
        """The same for a cloud already in HBM: d_pts float32 [M][3], d_mean float64 [M] (device pointers; enqueued, not finished, on return)."""
        self._ck(lib().slgc_knn_mean_distance_dev(self._h, d_pts, int(M), int(k), d_mean))

    def pipeline(self, runs, proj_size, img_white=None, threshold=None, eps=1, m=10, order=ORDER_X, mode=TRI_EXACT,
                 want_maps=True, want_lists=False):
        """Whole reference pipeline with one upload (set_calibration first).  Returns a dict: pts (3,M) float64, colors
        [M,3] or None, h_pixels / v_pixels int64 (if want_maps), cam_pts / proj_pts (unfiltered lists, if want_lists)."""
        arr = runs if isinstance(runs, np.ndarray) else None
        stacks = [arr] if (arr is not None and arr.ndim == 3) else list(runs)
        prepared = [self._stack(s) for s in stacks]
        if len({dt for _, dt in prepared}) != 1 or len({s.shape for s, _ in prepared}) != 1:
            raise ValueError("all runs must share dtype and shape")
        dt = prepared[0][1]
        N, H, W = prepared[0][0].shape
        wh = None
        if img_white is not None:
            wh = np.ascontiguousarray(np.asarray(img_white)[:H, :W, :3], dtype=np.uint8)
            if wh.shape != (H, W, 3):
                raise ValueError("img_white must be [H,W,3]")
        ptrs = (C.c_void_p * len(prepared))(*[s.ctypes.data for s, _ in prepared])
        M = C.c_int64()
        thr = float("nan") if threshold is None else float(threshold)
        self._ck(lib().slgc_pipeline_count(self._h, ptrs, dt, len(prepared), N, H, W, float(eps), float(m), int(proj_size[0]),
                                           int(proj_size[1]), _ptr(wh), int(order), int(mode), thr, C.byref(M)))
        n = M.value
        out = {"pts": _out((3, n), np.float64), "colors": _out((n, 3), np.float64) if wh is not None else None}
        hp = vp = cam = proj = None
        if want_maps:
            hp, vp = _out((H, W), np.int64), _out((H, W), np.int64)
        raw = C.c_int64()
        # first fetch learns the unfiltered length when the lists are wanted
        self._ck(lib().slgc_pipeline_fetch(self._h, _ptr(hp), _ptr(vp), _ptr(out["pts"]), _ptr(out["colors"]), C.byref(raw), None, None))
        if want_lists:
            cam, proj = _out((raw.value, 2), np.float32), _out((raw.value, 2), np.float32)
            self._ck(lib().slgc_pipeline_fetch(self._h, None, None, None, None, None, _ptr(cam), _ptr(proj)))
        out.update(h_pixels=hp, v_pixels=vp, cam_pts=cam, proj_pts=proj, n_unfiltered=int(raw.value))
        return out

    # ---- device-resident entry points (enqueue only)
    def decode_dev(self, d_stack: int, n_runs, run_stride, plane_stride, N, rows, W, d_h: int, d_v: int, eps=1, m=10, variant=0):
        self._ck(lib().slgc_decode_dev(self._h, d_stack, n_runs, run_stride, plane_stride, N, rows, W, float(eps), float(m),
                                       d_h, d_v, int(variant)))

    def scan_dev(self, d_stack: int, n_runs, run_stride, plane_stride, N, rows, W, row0, proj_size, d_xyz: int, d_count=None,
                 d_h=None, d_v=None, eps=1, m=10, mode=TRI_ALGEBRAIC):
        self._ck(lib().slgc_scan_dev(self._h, d_stack, n_runs, run_stride, plane_stride, N, rows, W, row0, int(proj_size[0]),
                                     int(proj_size[1]), float(eps), float(m), int(mode), d_h, d_v, d_xyz, d_count))

    def scan_bgr_dev(self, d_bgr: int, n_runs, run_stride, plane_stride, N, rows, W, row0, proj_size, d_xyz: int, d_count=None, d_h=None, d_v=None,
                     coeff_bits=15, eps=1, m=10, mode=TRI_ALGEBRAIC):
        """scan_dev straight from BGR frames [n_runs][N][rows][W][3] (plane_stride / run_stride in bytes): cv2.cvtColor's luma inside the frame loads."""
        self._ck(lib().slgc_scan_bgr_dev(self._h, d_bgr, int(n_runs), int(run_stride), int(plane_stride), int(N), int(rows), int(W), int(row0),
                                         int(proj_size[0]), int(proj_size[1]), int(coeff_bits), float(eps), float(m), int(mode), d_h, d_v, d_xyz, d_count))

    def decode_bgr_dev(self, d_bgr: int, n_runs, run_stride, plane_stride, N, rows, W, d_h: int, d_v: int, coeff_bits=15, eps=1, m=10):
        """decode_dev straight from BGR frames [n_runs][N][rows][W][3] (strides in bytes) -> int16 maps."""
        self._ck(lib().slgc_decode_bgr_dev(self._h, d_bgr, int(n_runs), int(run_stride), int(plane_stride), int(N), int(rows), int(W), int(coeff_bits),
                                           float(eps), float(m), d_h, d_v))

    def selftest_thresholds(self, eps: int = 1, black_lo: int = 0, black_hi: int = 256) -> int:
        """Exhaustive check of the decode kernels' integer-threshold folding over the uint8 domain; returns the mismatch count."""
        bad = C.c_uint64()
        self._ck(lib().slgc_selftest_thresholds(self._h, int(eps), int(black_lo), int(black_hi), C.byref(bad)))
        return int(bad.value)

    def selftest_classify(self, negative_control: bool = False) -> int:
        """Exhaustive check of the packed-16 rule evaluation against the scalar rule table; returns the mismatch count."""
        bad = C.c_uint64()
        self._ck(lib().slgc_selftest_classify(self._h, int(bool(negative_control)), C.byref(bad)))
        return int(bad.value)

    def triangulate_maps_dev(self, d_h: int, d_v: int, rows, W, row0, proj_size, d_xyz: int, d_count=None, mode=TRI_ALGEBRAIC):
        self._ck(lib().slgc_triangulate_maps_dev(self._h, d_h, d_v, rows, W, row0, int(proj_size[0]), int(proj_size[1]),
                                                 int(mode), d_xyz, d_count))

    def alloc_cloud_lists(self, npix: int, colors: bool = True, points: bool = True, lists: bool = True, f32: bool = False) -> CloudLists:
        return CloudLists(self, npix, colors, points, lists, f32)

    def cloud_lists_dev(self, d_h: int, d_v: int, d_xyz, d_white, cam_w, cam_h, proj_size, lists: CloudLists):
        """int16 maps + dense float32 XYZ (+ device-resident uint8 RGB white image) -> the reference's x-major lists, float64 (3,M)
        points and colours, all in HBM (asynchronous).  d_xyz / d_white may be None."""
        if lists.cam is None or lists.f32:
            raise ValueError("cloud_lists_dev always writes the correspondence lists and float64 products: allocate the CloudLists with lists=True, f32=False")
        self._ck(lib().slgc_cloud_lists_dev(self._h, d_h, d_v, d_xyz if lists.pts is not None else None,      # d_xyz None + points wanted: triangulated in-kernel
                                            d_white if lists.colors is not None else None, int(cam_w), int(cam_h), int(proj_size[0]),
                                            int(proj_size[1]), lists.cam.ptr, lists.proj.ptr, lists.pts.ptr if lists.pts is not None else None,
                                            lists.colors.ptr if lists.colors is not None else None, lists.count.ptr))

    def cloud_dev(self, d_stack: int, n_runs, run_stride, plane_stride, N, cam_h, cam_w, proj_size, d_white, lists: CloudLists, d_h=None, d_v=None,
                  eps=1, m=10):
        """Whole scan -> the reference-shaped lists in one call (slgc_cloud_dev): decode kernel, then the x-major list build with the
        triangulation inside it -- no dense XYZ.  lists.pts / lists.colors may be absent."""
        fn = lib().slgc_cloud32_dev if lists.f32 else lib().slgc_cloud_dev      # lists.f32: float32 points / colours (not the reference's dtypes)
        self._ck(fn(self._h, d_stack, int(n_runs), int(run_stride), int(plane_stride), int(N), int(cam_h), int(cam_w),
                                      int(proj_size[0]), int(proj_size[1]), float(eps), float(m), d_white if lists.colors is not None else None, d_h, d_v,
                                      lists.cam.ptr if lists.cam is not None else None, lists.proj.ptr if lists.proj is not None else None,
                                      lists.pts.ptr if lists.pts is not None else None,
                                      lists.colors.ptr if lists.colors is not None else None, lists.count.ptr))

    def compact_dev(self, d_xyz: int, rows, W, row0, d_points: int, d_keys, d_count: int):
        self._ck(lib().slgc_compact_dev(self._h, d_xyz, rows, W, row0, d_points, d_keys, d_count))

    def compact_records_dev(self, d_xyz: int, rows, W, row0, d_records: int, d_count: int):
        self._ck(lib().slgc_compact_records_dev(self._h, d_xyz, rows, W, row0, d_records, d_count))

    def pack_hv24_dev(self, d_h: int, d_v: int, npix: int, code_bits: int, d_wire: int):
        """int16 maps -> 3 bytes per pixel (12 + 12 bits, 0xFFF = -1) for the multi-GPU exchange; code_bits <= WIRE_MAX_CODE_BITS."""
        self._ck(lib().slgc_pack_hv24_dev(self._h, d_h, d_v, int(npix), int(code_bits), d_wire))

    def unpack_hv24_dev(self, d_wire: int, npix: int, d_h: int, d_v: int):
        self._ck(lib().slgc_unpack_hv24_dev(self._h, d_wire, int(npix), d_h, d_v))

    def triangulate_wire_dev(self, d_wire: int, rows, W, row0, proj_size, d_h: int, d_v: int, d_xyz: int, d_count=None, mode=TRI_ALGEBRAIC):
        """triangulate_maps_dev on wire-format maps; also writes the int16 maps d_h / d_v."""
        self._ck(lib().slgc_triangulate_wire_dev(self._h, d_wire, rows, W, row0, int(proj_size[0]), int(proj_size[1]), int(mode) & 1,
                                                 d_h, d_v, d_xyz, d_count))

    def synth_scene_dev(self, d_stack: int, plane_stride, N, H, W, row0=0, rows=None, seed=1, noise=3, shadow=True):
        rows = H if rows is None else rows
        self._ck(lib().slgc_synth_scene_dev(self._h, d_stack, plane_stride, N, H, W, row0, rows, seed, noise, int(bool(shadow))))

    def synth_physical_dev(self, d_stack, plane_stride, N, H, W, proj_size, row0=0, rows=None, seed=1, noise=3, d_h_true=None, d_v_true=None,
                           d_truth_xyz=None, gains=None, r2_max=None):
        """Physically consistent synthetic capture (set_calibration first): frames encode the projector pixel that really lights each
        camera pixel of a plane + sphere scene; optional device outputs: the encoded codes (int16, -1 = unlit) and the true surface points.
        gains = (lo, hi) surface gains of the checker (default 140, 180); r2_max = trusted radius^2 of the projector lens model (default 0.16)."""
        rows = H if rows is None else rows
        if gains is None and r2_max is None:
            self._ck(lib().slgc_synth_physical_dev(self._h, d_stack, int(plane_stride), int(N), int(H), int(W), int(row0), int(rows), int(proj_size[0]),
                                                   int(proj_size[1]), int(seed), int(noise), d_h_true, d_v_true, d_truth_xyz))
            return
        lo, hi = (140, 180) if gains is None else gains
        self._ck(lib().slgc_synth_physical_ex_dev(self._h, d_stack, int(plane_stride), int(N), int(H), int(W), int(row0), int(rows), int(proj_size[0]),
                                                  int(proj_size[1]), int(seed), int(noise), int(lo), int(hi), float(0.16 if r2_max is None else r2_max),
                                                  d_h_true, d_v_true, d_truth_xyz))

    def synth_bgr_dev(self, d_gray, gray_plane_stride, N, H, W, d_bgr, bgr_plane_stride, row0=0, rows=None):
        """A BGR capture [N][rows][W][3] of a grey stack in HBM (channel offsets so that the luma has work to do; twin: oracle_np.gray_to_bgr_capture)."""
        rows = H if rows is None else rows
        self._ck(lib().slgc_synth_bgr_dev(self._h, d_gray, int(gray_plane_stride), int(N), int(H), int(W), int(row0), int(rows), d_bgr, int(bgr_plane_stride)))

    def synth_uniform_dev(self, d_stack, plane_stride, N, H, W, row0=0, rows=None, seed=0):
        """SURVEY.md 8(d) S-uniform: every byte uniform in 0..255 (counter hash; twin: oracle_np.synth_uniform)."""
        rows = H if rows is None else rows
        self._ck(lib().slgc_synth_uniform_dev(self._h, d_stack, int(plane_stride), int(N), int(H), int(W), int(row0), int(rows), int(seed)))

    def prof_begin(self, max_launches: int = 4096, stride: int = 1):
        self._ck(lib().slgc_prof_begin(self._h, int(max_launches), int(stride)))

    def prof_end(self):
        """-> (summed decode-kernel milliseconds, launches sampled)"""
        ms, n = C.c_double(), C.c_int()
        self._ck(lib().slgc_prof_end(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def prof_samples(self) -> np.ndarray:
        """Kernel durations (ms) of the launches sampled between the last prof_begin / prof_end, in launch order."""
        n = C.c_int()
        self._ck(lib().slgc_prof_samples(self._h, None, 0, C.byref(n)))
        out = np.empty(max(n.value, 0), np.float32)
        if n.value:
            self._ck(lib().slgc_prof_samples(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def scan_batch_dev(self, d_stacks: int, n_scans, scan_stride, plane_stride, N, rows, W, row0, proj_size, d_xyz: int, d_h=None, d_v=None,
                       eps=1.0, m=10.0, mode=TRI_ALGEBRAIC):
        """n_scans independent single-run scans of one geometry in one launch (throughput mode); outputs back to back per scan.
        d_h = d_v = None: XYZ only (the fused kernel then stores no maps)."""
        self._ck(lib().slgc_scan_batch_dev(self._h, d_stacks, int(n_scans), int(scan_stride), int(plane_stride), int(N), int(rows), int(W), int(row0),
                                           int(proj_size[0]), int(proj_size[1]), float(eps), float(m), int(mode), d_h, d_v, d_xyz))

    def build_ray_tables_dev(self, rows, W, row0, proj_size):
        """Per-calibration ray tables of the dense path (asynchronous); built on first use otherwise."""
        self._ck(lib().slgc_build_ray_tables_dev(self._h, int(rows), int(W), int(row0), int(proj_size[0]), int(proj_size[1])))

    def ray_table_info(self):
        """-> (node table in use, max relative difference between the rays interpolated from it and the exact ones; -1.0 if none was built)."""
        use, err = C.c_int(), C.c_double()
        self._ck(lib().slgc_ray_table_info(self._h, C.byref(use), C.byref(err)))
        return bool(use.value), err.value

    def guard_count_dev(self, d_h: int, d_v: int, rows, W, row0, proj_size, d_counts: int):
        """d_counts[0] += decodable pixels, d_counts[1] += pixels on the guarded (float32-mirror) triangulation path."""
        self._ck(lib().slgc_guard_count_dev(self._h, d_h, d_v, int(rows), int(W), int(row0), int(proj_size[0]), int(proj_size[1]), d_counts))

    # ---- RCCL
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = lib().slgc_comm_unique_id(buf)
        if rc:
            raise SlgcError("slgc_comm_unique_id failed (librccl.so not loadable?)")
        return buf.raw

    def comm_init(self, rank: int, nranks: int, uid: bytes):
        assert len(uid) == UNIQUE_ID_BYTES
        self._ck(lib().slgc_comm_init(self._h, rank, nranks, C.create_string_buffer(uid, UNIQUE_ID_BYTES)))
        self.rank, self.nranks = rank, nranks

    def comm_destroy(self):
        self._ck(lib().slgc_comm_destroy(self._h))

    def comm_barrier(self):
        self._ck(lib().slgc_comm_barrier(self._h))

    def comm_allreduce_max(self, value: float) -> float:
        v = C.c_double(value)
        self._ck(lib().slgc_comm_allreduce_max_f64(self._h, C.byref(v)))
        return float(v.value)

    def comm_allgather_i64(self, mine: int):
        out = (C.c_int64 * self.nranks)()
        self._ck(lib().slgc_comm_allgather_i64(self._h, int(mine), out))
        return [int(x) for x in out]

    def comm_info(self) -> dict:
        """RCCL's own view of the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), not what comm_init was told."""
        n, r, d = _i(), _i(), _i()
        self._ck(lib().slgc_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)))
        return {"nranks": n.value, "rank": r.value, "device": d.value}

    def comm_rank_devices(self):
        """-> ([PCI bus id of rank 0's GPU, rank 1's, ...], number of distinct GPUs), all-gathered over the communicator (collective)."""
        n = self.comm_info()["nranks"]
        buf = C.create_string_buffer(BUS_ID_BYTES * n)
        distinct = _i()
        self._ck(lib().slgc_comm_allgather_bus_ids(self._h, buf, C.byref(distinct)))
        raw = buf.raw
        return [raw[i * BUS_ID_BYTES:(i + 1) * BUS_ID_BYTES].split(b"\0", 1)[0].decode() for i in range(n)], distinct.value

    def comm_allgatherv(self, d_send: int, d_recv: int, counts, displs):
        n = self.nranks
        c = (C.c_int64 * n)(*[int(x) for x in counts])
        d = (C.c_int64 * n)(*[int(x) for x in displs])
        self._ck(lib().slgc_comm_allgatherv(self._h, d_send, d_recv, c, d))

    def comm_allgatherv_begin(self, d_send: int, d_recv: int, counts, displs, slot: int):
        n = self.nranks
        c = (C.c_int64 * n)(*[int(x) for x in counts])
        d = (C.c_int64 * n)(*[int(x) for x in displs])
        self._ck(lib().slgc_comm_allgatherv_begin(self._h, d_send, d_recv, c, d, int(slot)))

    def comm_allgatherv_pair_begin(self, d_send_a: int, d_recv_a: int, d_send_b: int, d_recv_b: int, counts, displs, slot: int):
        n = self.nranks
        c = (C.c_int64 * n)(*[int(x) for x in counts])
        d = (C.c_int64 * n)(*[int(x) for x in displs])
        self._ck(lib().slgc_comm_allgatherv_pair_begin(self._h, d_send_a, d_recv_a, d_send_b, d_recv_b, c, d, int(slot)))

    def comm_wait(self, slot: int):
        self._ck(lib().slgc_comm_wait(self._h, int(slot)))

    # ---- direct exchange (every rank pushes its band into every peer's buffer over xGMI: csrc/direct.hip)
    def direct_init(self, rank: int, nranks: int, key: str):
        self._ck(lib().slgc_direct_init(self._h, int(rank), int(nranks), str(key).encode()))
        self.direct_rank, self.direct_nranks = rank, nranks

    def direct_destroy(self):
        self._ck(lib().slgc_direct_destroy(self._h))

    def direct_register(self, d_base: int, nbytes: int):
        self._ck(lib().slgc_direct_register(self._h, d_base, int(nbytes)))

    def direct_unregister(self, d_base: int):
        """Collective: the buffer leaves the exchange on every rank (do this before freeing it)."""
        self._ck(lib().slgc_direct_unregister(self._h, _vp(d_base)))

    def direct_allgatherv_begin(self, d_bases, layouts, slot: int):
        """d_bases: 1..3 registered buffers (base device pointers); layouts: [(counts, displs)] per buffer, bytes per rank."""
        n, k = self.direct_nranks, len(d_bases)
        bases = (C.c_void_p * k)(*[int(b) for b in d_bases])
        cs = [(C.c_int64 * n)(*[int(x) for x in c]) for c, _ in layouts]
        ds = [(C.c_int64 * n)(*[int(x) for x in d]) for _, d in layouts]
        cp = (C.POINTER(C.c_int64) * k)(*[C.cast(a, C.POINTER(C.c_int64)) for a in cs])
        dp = (C.POINTER(C.c_int64) * k)(*[C.cast(a, C.POINTER(C.c_int64)) for a in ds])
        self._ck(lib().slgc_direct_allgatherv_begin(self._h, k, bases, cp, dp, int(slot)))

    def direct_wait(self, slot: int):
        self._ck(lib().slgc_direct_wait(self._h, int(slot)))

    def direct_release(self, d_bases):
        k = len(d_bases)
        self._ck(lib().slgc_direct_release(self._h, k, (C.c_void_p * k)(*[int(b) for b in d_bases])))

    def direct_barrier(self):
        self._ck(lib().slgc_direct_barrier(self._h))

    def direct_allgather_i64(self, mine: int):
        out = (C.c_int64 * self.direct_nranks)()
        self._ck(lib().slgc_direct_allgather_i64(self._h, int(mine), out))
        return [int(x) for x in out]

    def scan_sharded_dev(self, d_band_stack: int, n_runs, run_stride, plane_stride, N, H, W, proj_size, d_h_full: int, d_v_full: int,
                         d_xyz_full: int, eps=1, m=10, mode=TRI_ALGEBRAIC):
        """One row-sharded scan in one C call (decode band -> in-place map all-gatherv -> full triangulation), asynchronous."""
        self._ck(lib().slgc_scan_sharded_dev(self._h, d_band_stack, n_runs, run_stride, plane_stride, N, H, W, proj_size[0], proj_size[1],
                                             float(eps), float(m), mode, d_h_full, d_v_full, d_xyz_full))


_default_ctx = None
_default_lock = threading.Lock()


def default_context() -> Context:
    """Process-wide context on device SLGC_DEVICE (default 0) for the reference-shaped module functions."""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context(int(os.environ.get("SLGC_DEVICE", "0")))
        return _default_ctx
