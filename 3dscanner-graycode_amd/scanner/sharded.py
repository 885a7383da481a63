"""Row-sharded scan across the GPUs of one node (BASELINE.json configs[3]).

The reference is single-process; this module exists because the build shards ONE scan by image rows:
rank r decodes + triangulates rows [row0, row0+rows) of every frame, compacts its band row-major
(16-byte records: float32 XYZ + uint32 linear pixel key) and the ranks all-gatherv the records so every rank holds the
reassembled cloud.  Band-major concatenation == row-major order of the full image; the reference's
x-major order (triangulate.py:52-53) is recovered from the keys (:func:`x_major_permutation`).

Three exchange strategies, all ending with the whole cloud on every rank:

* ``exchange="maps"`` (default): the ranks all-gather(v) their int16 (h, v) map bands -- 4 B/pixel, sizes known from the
  plan, so no count exchange and no host synchronisation -- and every rank triangulates the full maps itself (the dense
  triangulation kernel is ~70 us for 4096x3000, far cheaper than moving 12 B/pixel of XYZ over xGMI).
* ``exchange="xyz"``: each rank runs the FUSED kernel (decode with the triangulation tail) on its band, straight into its slot
  of full-size maps and XYZ, and the ranks all-gather the three band sets in place (16 B/pixel, fixed sizes, no counts, no
  replicated triangulation).
* ``exchange="records"``: each rank triangulates and compacts its band into 16-byte records {x, y, z, key} and the ranks
  all-gatherv those (a count all-gather first; 16 B per VALID pixel).

The exchange is behind a three-method protocol (``allgather_i64``, ``allgatherv``, ``barrier``):
:class:`RcclExchange` is the product implementation (RCCL over xGMI through the C-ABI, device pointers);
the CPU test-suite drives the same planning / layout / reassembly code over gloo with host arrays.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

RECORD_BYTES = 16  # float32 x, y, z + uint32 linear pixel index (y * W + x)
RECORD_DTYPE = np.dtype([("xyz", np.float32, 3), ("key", np.uint32)])


@dataclass(frozen=True)
class ShardPlan:
    """Contiguous row bands: the first H % G ranks get one extra row."""
    H: int
    W: int
    G: int

    def band(self, rank: int):
        if not 0 <= rank < self.G:
            raise ValueError("rank out of range")
        base, extra = divmod(self.H, self.G)
        rows = base + (1 if rank < extra else 0)
        row0 = rank * base + min(rank, extra)
        return row0, rows

    def bands(self):
        return [self.band(r) for r in range(self.G)]

    def band_offset_bytes(self, rank: int) -> int:
        """Offset of the band inside one uint8 frame plane."""
        return self.band(rank)[0] * self.W


def gather_layout(counts, record_bytes: int):
    """Per-rank byte counts and displacements of an all-gatherv of ``counts[r]`` records."""
    c = [int(x) for x in counts]
    if any(x < 0 for x in c):
        raise ValueError("negative count")
    byte_counts = [x * record_bytes for x in c]
    displs, acc = [], 0
    for b in byte_counts:
        displs.append(acc)
        acc += b
    return byte_counts, displs, acc


def x_major_permutation(keys: np.ndarray, W: int, H: int) -> np.ndarray:
    """Permutation that sorts records from row-major (key order) into the reference's x-major order."""
    k = np.asarray(keys, dtype=np.int64)
    return np.argsort((k % W) * H + (k // W), kind="stable")


def to_reference_lists(records: np.ndarray, W: int, H: int):
    """Reassembled cloud (RECORD_DTYPE array) -> (cam_pts float32 [M,2] x-major, Pts float64 (3,M)) as the reference returns them."""
    perm = x_major_permutation(records["key"], W, H)
    k = records["key"].astype(np.int64)[perm]
    cam = np.stack([k % W, k // W], axis=1).astype(np.float32)
    return cam, np.ascontiguousarray(records["xyz"].astype(np.float64)[perm].T)


def exchange_records(exchange, send_records, count: int, recv_records):
    """Counts all-gather, then ONE all-gatherv of the 16-byte records.  Buffers are whatever the exchange understands
    (device pointers for RCCL, NumPy arrays for the gloo test double).  Returns (counts, total)."""
    counts = exchange.allgather_i64(int(count))
    bc, bd, _ = gather_layout(counts, RECORD_BYTES)
    exchange.allgatherv(send_records, recv_records, bc, bd)
    return counts, sum(counts)


def share_unique_id(rank: int, make_uid, key: str = None, timeout_s: float = 300.0, directory: str = "/tmp"):
    """Hand rank 0's RCCL unique id (128 bytes) to the other ranks of one node through a file.

    ``torch.distributed.run`` starts the ranks as children of one agent process and exports MASTER_PORT, so
    ``MASTER_PORT + parent pid`` names a rendezvous that is unique per launch (no torch import needed in the workers).
    Rank 0 writes ``<file>.tmp`` and renames it (atomic), the others poll.  Returns (uid, path); rank 0 should remove the file
    once every rank has passed its first collective."""
    import os
    import time
    if key is None:
        key = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    path = os.path.join(directory, f"slgc_uid_{key}")
    if rank == 0:
        uid = make_uid()
        if len(uid) != 128:
            raise ValueError("unique id must be 128 bytes")
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid, path
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == 128:
                return uid, path
        except FileNotFoundError:
            pass
        time.sleep(0.02)
    raise RuntimeError(f"timed out waiting for the RCCL unique id from rank 0 ({path})")


class RcclExchange:
    """RCCL over xGMI through libslgc.so (one context = one rank = one GPU)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.rank, self.nranks = ctx.rank, ctx.nranks

    def allgather_i64(self, value: int):
        return self.ctx.comm_allgather_i64(value)

    @staticmethod
    def _ptr(b):
        return b.ptr if hasattr(b, "ptr") else int(b)                   # DeviceBuffer or raw device pointer

    def allgatherv(self, d_send, d_recv, byte_counts, byte_displs):
        self.ctx.comm_allgatherv(self._ptr(d_send), self._ptr(d_recv), byte_counts, byte_displs)

    def allgatherv_begin(self, d_send, d_recv, byte_counts, byte_displs, slot: int):
        """Enqueue on the communication stream (after the compute stream's work so far) and return; see wait()."""
        self.ctx.comm_allgatherv_begin(self._ptr(d_send), self._ptr(d_recv), byte_counts, byte_displs, slot)

    def allgatherv_pair_begin(self, d_send_a, d_recv_a, d_send_b, d_recv_b, byte_counts, byte_displs, slot: int):
        """Two buffers with one shard layout (the h and v maps) in one RCCL group."""
        self.ctx.comm_allgatherv_pair_begin(self._ptr(d_send_a), self._ptr(d_recv_a), self._ptr(d_send_b), self._ptr(d_recv_b),
                                            byte_counts, byte_displs, slot)

    def wait(self, slot: int):
        """Make the compute stream wait for the exchange started in ``slot``."""
        self.ctx.comm_wait(slot)

    def barrier(self):
        self.ctx.comm_barrier()

    def register(self, buffers):
        """(RCCL needs no registration; the direct exchange does)"""

    def release(self, buffers):
        """(RCCL overwrites a rank's buffers from that rank's own communication stream, ordered after its compute stream: nothing to release)"""


class DirectExchange:
    """The same exchange protocol with every band pushed straight into every peer's buffer over xGMI, all links at once (csrc/direct.hip;
    hipIpcMemHandle mappings, flags in one shared-memory segment, no RCCL).  Buffers that take part are the scanner's full-size buffers,
    registered collectively (same order on every rank); every exchange is in place: the rank's band sits at displs[rank] of the buffer.

    EXPERIMENTAL until a node has timed it against RcclExchange: RCCL stays the default of :class:`ShardedScanner` users (bench.py times this
    one as a ``sharded_alternatives`` entry and verifies it bit for bit)."""

    def __init__(self, ctx, rank: int, nranks: int, key: str):
        self.ctx, self.rank, self.nranks = ctx, rank, nranks
        ctx.direct_init(rank, nranks, key)
        self._base = {}                                                # registered base pointer -> size

    def register(self, buffers):
        for b in buffers:
            if b.ptr not in self._base:
                self.ctx.direct_register(b.ptr, b.nbytes)
                self._base[b.ptr] = b.nbytes

    def unregister(self, buffers):
        """Collective, same order on every rank: the buffers leave the exchange (before they are freed; their slots can be registered again)."""
        for b in buffers:
            if b.ptr in self._base:
                self.ctx.direct_unregister(b.ptr)
                del self._base[b.ptr]

    def _in_place(self, d_send, d_recv, displs):
        recv = RcclExchange._ptr(d_recv)
        if recv not in self._base:
            raise ValueError("DirectExchange: the receive buffer is not registered")
        if RcclExchange._ptr(d_send) != recv + displs[self.rank]:
            raise ValueError("DirectExchange exchanges in place: the band must sit at displs[rank] of the receive buffer")
        return recv

    def allgather_i64(self, value: int):
        return self.ctx.direct_allgather_i64(value)

    def allgatherv_begin(self, d_send, d_recv, byte_counts, byte_displs, slot: int):
        self.ctx.direct_allgatherv_begin([self._in_place(d_send, d_recv, byte_displs)], [(byte_counts, byte_displs)], slot)

    def allgatherv_pair_begin(self, d_send_a, d_recv_a, d_send_b, d_recv_b, byte_counts, byte_displs, slot: int):
        self.ctx.direct_allgatherv_begin([self._in_place(d_send_a, d_recv_a, byte_displs), self._in_place(d_send_b, d_recv_b, byte_displs)],
                                         [(byte_counts, byte_displs)] * 2, slot)

    def allgatherv(self, d_send, d_recv, byte_counts, byte_displs):
        self.allgatherv_begin(d_send, d_recv, byte_counts, byte_displs, 3)
        self.wait(3)

    def wait(self, slot: int):
        self.ctx.direct_wait(slot)

    def release(self, buffers):
        """Compute stream: everything enqueued so far is done with what the last exchange left in these buffers; peers may overwrite them."""
        bases = [b.ptr for b in buffers if b.ptr in self._base]
        for i in range(0, len(bases), 3):
            self.ctx.direct_release(bases[i:i + 3])

    def barrier(self):
        self.ctx.direct_barrier()


def map_band_layout(plan: ShardPlan):
    """Byte counts / displacements of the int16 map bands inside one full [H][W] int16 map."""
    counts = [rows * plan.W * 2 for _, rows in plan.bands()]
    displs = [row0 * plan.W * 2 for row0, _ in plan.bands()]
    return counts, displs


def band_layout(plan: ShardPlan, bytes_per_px: int):
    """Byte counts / displacements of the row bands inside one full [H][W] array of ``bytes_per_px`` bytes per pixel."""
    return ([rows * plan.W * bytes_per_px for _, rows in plan.bands()], [row0 * plan.W * bytes_per_px for row0, _ in plan.bands()])


def exchange_bands(exchange, plan: ShardPlan, arrays, band_view):
    """In-place all-gatherv of the row bands of several full-size arrays: ``arrays`` = [(buffer, bytes per pixel), ...] -- the "xyz"
    strategy sends (h, 2), (v, 2), (xyz, 12).  One exchange per array, fixed sizes from the plan: no counts, no host round trip."""
    for buf, bpp in arrays:
        counts, displs = band_layout(plan, bpp)
        exchange.allgatherv(band_view(buf, displs[exchange.rank]), buf, counts, displs)


def exchange_map_bands(exchange, plan: ShardPlan, h_full, v_full, band_view):
    """In-place all-gatherv of the (h, v) map bands: every rank has written its own band into the full-size maps;
    afterwards both maps are complete everywhere.  ``band_view(buf, byte_offset)`` addresses a buffer of the exchange's
    kind (device pointer arithmetic for RCCL, a NumPy view for the gloo test double)."""
    counts, displs = map_band_layout(plan)
    if hasattr(exchange, "allgatherv_pair_begin"):                 # both maps in one RCCL group
        exchange.allgatherv_pair_begin(band_view(h_full, displs[exchange.rank]), h_full, band_view(v_full, displs[exchange.rank]), v_full,
                                       counts, displs, 3)
        exchange.wait(3)
        return
    for full in (h_full, v_full):
        exchange.allgatherv(band_view(full, displs[exchange.rank]), full, counts, displs)


class ShardedScanner:
    """Device-resident sharded scan: owns the per-rank buffers, runs one band per call.

    With the "maps" strategy consecutive scans can be pipelined (:meth:`submit` / :meth:`flush`): the all-gatherv of scan i runs
    on the communication stream while the compute stream triangulates scan i-1 and decodes scan i+1 (two sets of map buffers)."""

    def __init__(self, ctx, exchange, plan: ShardPlan, proj_size, n_frames: int, mode: int = 1, exchange_kind: str = "maps",
                 wire: str = "int16"):
        """wire ("maps" strategy): "int16" (default) sends the two int16 maps as they are (4 B/pixel); "hv24" packs them into
        3 B/pixel for the exchange (codes of <= 11 bits) and unpacks on arrival -- fewer bytes on the links for two small streaming
        kernels; "auto" = hv24 when there is more than one rank and the codes fit.  EXPERIMENTAL until measured on real xGMI: int16
        stays the default because no multi-GPU box has run either yet."""
        if exchange_kind not in ("maps", "records", "xyz"):
            raise ValueError("exchange_kind must be 'maps', 'xyz' or 'records'")
        if wire not in ("auto", "int16", "hv24"):
            raise ValueError("wire must be 'auto', 'int16' or 'hv24'")
        from ._native import WIRE_MAX_CODE_BITS
        self.code_bits = int((n_frames - 2) / 4)
        if wire == "auto":
            wire = "hv24" if (plan.G > 1 and self.code_bits <= WIRE_MAX_CODE_BITS) else "int16"
        if wire == "hv24" and self.code_bits > WIRE_MAX_CODE_BITS:
            raise ValueError(f"the 3-byte wire format holds codes of at most {WIRE_MAX_CODE_BITS} bits (N={n_frames} has {self.code_bits})")
        self.wire = wire
        self.ctx, self.exchange, self.plan = ctx, exchange, plan
        self.proj_size, self.N, self.mode, self.kind = proj_size, n_frames, mode, exchange_kind
        # the camera-ray table decision (per-pixel or node table) is taken for the whole image, not for this rank's band: a pixel's XYZ is
        # then bit-identical for any number of ranks (and equal to a single-GPU scan of the image).  Set around every call that launches
        # kernels (_whole_image) and cleared afterwards: the context may serve other scanners and plain scans in between.
        self.rank = exchange.rank
        self.row0, self.rows = plan.band(self.rank)
        band_px, full_px = self.rows * plan.W, plan.H * plan.W
        self.count = ctx.alloc(8)
        self.last_counts = None
        if exchange_kind == "maps":
            self._sets = [(ctx.alloc(max(16, full_px * 2)), ctx.alloc(max(16, full_px * 2))) for _ in range(2)]
            self._wire = [ctx.alloc(max(16, full_px * 3)) for _ in range(2)] if wire == "hv24" else None
            self.h_full, self.v_full = self._sets[0]                 # where scan_maps() / the last finished submit() left the maps
            self.xyz_full = ctx.alloc(max(16, full_px * 12))
            self._submitted = 0
            self._pending = None
            self._pair = False
            # what travels: the packed wire buffers, or the maps themselves (registered with an exchange that maps peers' buffers)
            self._exchanged = [[self._wire[s]] if wire == "hv24" else list(self._sets[s]) for s in range(2)]
        elif exchange_kind == "xyz":
            self.wire = "int16"
            self._sets = [(ctx.alloc(max(16, full_px * 2)), ctx.alloc(max(16, full_px * 2)), ctx.alloc(max(16, full_px * 12))) for _ in range(2)]
            self.h_full, self.v_full, self.xyz_full = self._sets[0]
            self._submitted = 0
            self._pending = None
            self._exchanged = [list(self._sets[s]) for s in range(2)]
        else:
            self.maps = ctx.alloc(max(16, band_px * 4))
            self.xyz = ctx.alloc(max(16, band_px * 12))
            self.records = ctx.alloc(max(16, band_px * RECORD_BYTES))
            self.all_records = ctx.alloc(max(16, full_px * RECORD_BYTES))
            self._exchanged = [[], []]
        if hasattr(exchange, "register"):                          # collective, same order on every rank
            for s in range(2):
                exchange.register(self._exchanged[s])
        self._used = [False, False]                                # set s has held an exchange's result (its re-use must be announced)

    def close(self):
        """Collective (same point on every rank): finish what is in flight, take the scanner's buffers out of the exchange -- a DirectExchange
        holds peer mappings of them on every rank, and has room for 16 buffers at a time -- and free them.  The scanner is unusable afterwards."""
        if getattr(self, "_closed", False):
            return
        self.flush()
        self.ctx.synchronize()
        if hasattr(self.exchange, "unregister"):
            for s in range(2):
                self.exchange.unregister(self._exchanged[s])
        bufs = [self.count]
        if self.kind in ("maps", "xyz"):
            bufs += [b for st in self._sets for b in st]
            bufs += list(self._wire or []) if self.kind == "maps" else []
            if self.kind == "maps":
                bufs.append(self.xyz_full)
        else:
            bufs += [self.maps, self.xyz, self.records, self.all_records]
        seen = set()
        for b in bufs:
            if b is not None and id(b) not in seen:
                seen.add(id(b))
                b.free()
        self._closed = True

    def _reuse(self, s: int):
        """This rank is about to overwrite buffer set s: with an exchange whose peers write into this rank's buffers (DirectExchange) they
        must not do so before everything enqueued so far -- the kernels that read the set's previous contents -- has run."""
        if self._used[s] and hasattr(self.exchange, "release"):
            self.exchange.release(self._exchanged[s])
        self._used[s] = True

    class _WholeImage:
        def __init__(self, scanner):
            self.s = scanner

        def __enter__(self):
            self.s._wi_depth = getattr(self.s, "_wi_depth", 0) + 1
            if self.s._wi_depth == 1:
                self.s.ctx.tune("image_rows", self.s.plan.H)

        def __exit__(self, *exc):
            self.s._wi_depth -= 1
            if self.s._wi_depth == 0:
                self.s.ctx.tune("image_rows", 0)

    def _whole_image(self):
        """slgc_tune("image_rows", H) for the duration of a call (re-entrant)."""
        return ShardedScanner._WholeImage(self)

    def scan_maps(self, d_band_stack: int, plane_stride: int, n_runs: int = 1, run_stride: int = 0, eps=1):
        """Band decode -> map all-gatherv -> full-image triangulation on every rank.  Nothing here synchronises with the host.
        Leaves int16 maps (h_full, v_full) and dense float32 XYZ [H][W][3] (NaN = undecodable) on every rank."""
        with self._whole_image():
            return self._scan_maps(d_band_stack, plane_stride, n_runs, run_stride, eps)

    def _scan_maps(self, d_band_stack, plane_stride, n_runs, run_stride, eps):
        c, W, H = self.ctx, self.plan.W, self.plan.H
        self.flush()
        self.h_full, self.v_full = self._sets[0]
        self._reuse(0)
        off = self.row0 * W * 2
        c.decode_dev(d_band_stack, n_runs, run_stride or self.N * plane_stride, plane_stride, self.N, self.rows, W,
                     self.h_full.at(off), self.v_full.at(off), eps=eps)
        if self.wire == "hv24":
            self._wire_begin(0, self.h_full, self.v_full, 3)
            self.exchange.wait(3)
            c.triangulate_wire_dev(self._wire[0].ptr, H, W, 0, self.proj_size, self.h_full.ptr, self.v_full.ptr, self.xyz_full.ptr, None,
                                   mode=self.mode & 1)           # unpacks while it loads: int16 maps + XYZ in one pass
            return
        exchange_map_bands(self.exchange, self.plan, self.h_full, self.v_full, lambda buf, o: buf.at(o))
        c.triangulate_maps_dev(self.h_full.ptr, self.v_full.ptr, H, W, 0, self.proj_size, self.xyz_full.ptr, None, mode=self.mode & 3)

    def _wire_begin(self, s: int, h_full, v_full, slot: int):
        """Pack this rank's band into its slot of wire buffer ``s`` and start the in-place all-gatherv of the packed bands."""
        W = self.plan.W
        band_px = self.rows * W
        wire = self._wire[s]
        if band_px:
            self.ctx.pack_hv24_dev(h_full.at(self.row0 * W * 2), v_full.at(self.row0 * W * 2), band_px, self.code_bits, wire.at(self.row0 * W * 3))
        counts = [rows * W * 3 for _, rows in self.plan.bands()]
        displs = [row0 * W * 3 for row0, _ in self.plan.bands()]
        self.exchange.allgatherv_begin(wire.at(displs[self.rank]), wire, counts, displs, slot)

    def _band_layouts(self, *bytes_per_px):
        return [band_layout(self.plan, b) for b in bytes_per_px]

    def _submit_xyz(self, d_band_stack: int, plane_stride: int, n_runs: int, run_stride: int, eps):
        """Fused kernel on the band, straight into this rank's slot of set s; in-place all-gather of the three band sets (slots 2s,
        2s+1 of the communication stream's events); the previous scan only has to be waited for."""
        c, W = self.ctx, self.plan.W
        s = self._submitted % 2
        h_full, v_full, xyz_full = self._sets[s]
        self._reuse(s)
        px0 = self.row0 * W
        if self.rows:
            c.scan_dev(d_band_stack, n_runs, run_stride or self.N * plane_stride, plane_stride, self.N, self.rows, W, self.row0, self.proj_size,
                       xyz_full.at(px0 * 12), None, h_full.at(px0 * 2), v_full.at(px0 * 2), eps=eps, mode=self.mode)
        (mc, md), (xc, xd) = self._band_layouts(2, 12)
        self.exchange.allgatherv_pair_begin(h_full.at(md[self.rank]), h_full, v_full.at(md[self.rank]), v_full, mc, md, 2 * s)
        self.exchange.allgatherv_begin(xyz_full.at(xd[self.rank]), xyz_full, xc, xd, 2 * s + 1)
        if self._pending is not None:
            self._finish(self._pending)
        self._pending = s
        self._submitted += 1

    def compute_only(self, d_band_stack: int, plane_stride: int, n_runs: int = 1, run_stride: int = 0, eps=1):
        """The kernels of one sharded scan on this rank WITHOUT the exchange (bench.py: what the links have to keep up with).  The
        maps / XYZ it leaves behind are not a reassembled result."""
        with self._whole_image():
            return self._compute_only(d_band_stack, plane_stride, n_runs, run_stride, eps)

    def _compute_only(self, d_band_stack, plane_stride, n_runs, run_stride, eps):
        c, W = self.ctx, self.plan.W
        px0 = self.row0 * W
        if self.kind == "xyz":
            h_full, v_full, xyz_full = self._sets[0]
            if self.rows:
                c.scan_dev(d_band_stack, n_runs, run_stride or self.N * plane_stride, plane_stride, self.N, self.rows, W, self.row0, self.proj_size,
                           xyz_full.at(px0 * 12), None, h_full.at(px0 * 2), v_full.at(px0 * 2), eps=eps, mode=self.mode)
            return
        if self.kind != "maps":
            raise ValueError("compute_only() covers the 'maps' and 'xyz' strategies")
        h_full, v_full = self._sets[0]
        if self.rows:
            c.decode_dev(d_band_stack, n_runs, run_stride or self.N * plane_stride, plane_stride, self.N, self.rows, W,
                         h_full.at(px0 * 2), v_full.at(px0 * 2), eps=eps)
        if self.wire == "hv24":
            if self.rows:
                c.pack_hv24_dev(h_full.at(px0 * 2), v_full.at(px0 * 2), self.rows * W, self.code_bits, self._wire[0].at(px0 * 3))
            c.triangulate_wire_dev(self._wire[0].ptr, self.plan.H, W, 0, self.proj_size, h_full.ptr, v_full.ptr, self.xyz_full.ptr, None, mode=self.mode & 1)
        else:
            c.triangulate_maps_dev(h_full.ptr, v_full.ptr, self.plan.H, W, 0, self.proj_size, self.xyz_full.ptr, None, mode=self.mode & 3)

    def submit(self, d_band_stack: int, plane_stride: int, n_runs: int = 1, run_stride: int = 0, eps=1):
        """Pipelined scan ("maps": decode this scan's band, start its exchange on the communication stream, then finish the
        PREVIOUS scan -- wait for its exchange, triangulate; "xyz": fused kernel on the band, start the exchange, wait for the
        previous one).  Call :meth:`flush` after the last one.  No host synchronisation."""
        if self.kind not in ("maps", "xyz"):
            raise ValueError("submit()/flush() pipeline the 'maps' and 'xyz' strategies")
        with self._whole_image():
            if self.kind == "xyz":
                return self._submit_xyz(d_band_stack, plane_stride, n_runs, run_stride, eps)
            return self._submit_maps(d_band_stack, plane_stride, n_runs, run_stride, eps)

    def _submit_maps(self, d_band_stack, plane_stride, n_runs, run_stride, eps):
        c, W = self.ctx, self.plan.W
        s = self._submitted % 2
        h_full, v_full = self._sets[s]
        self._reuse(s)
        off = self.row0 * W * 2
        c.decode_dev(d_band_stack, n_runs, run_stride or self.N * plane_stride, plane_stride, self.N, self.rows, W,
                     h_full.at(off), v_full.at(off), eps=eps)
        counts, displs = map_band_layout(self.plan)
        if self.wire == "hv24":
            self._wire_begin(s, h_full, v_full, 2 * s)
            self._pair = True
        elif hasattr(self.exchange, "allgatherv_pair_begin"):         # one RCCL group for both maps
            self.exchange.allgatherv_pair_begin(h_full.at(displs[self.rank]), h_full, v_full.at(displs[self.rank]), v_full, counts, displs, 2 * s)
            self._pair = True
        else:
            self.exchange.allgatherv_begin(h_full.at(displs[self.rank]), h_full, counts, displs, 2 * s)
            self.exchange.allgatherv_begin(v_full.at(displs[self.rank]), v_full, counts, displs, 2 * s + 1)
            self._pair = False
        if self._pending is not None:
            self._finish(self._pending)
        self._pending = s
        self._submitted += 1

    def _finish(self, s: int):
        if self.kind == "xyz":
            self.exchange.wait(2 * s)
            self.exchange.wait(2 * s + 1)
            self.h_full, self.v_full, self.xyz_full = self._sets[s]
            return
        h_full, v_full = self._sets[s]
        self.exchange.wait(2 * s)
        if not self._pair:
            self.exchange.wait(2 * s + 1)
        if self.wire == "hv24":
            self.ctx.triangulate_wire_dev(self._wire[s].ptr, self.plan.H, self.plan.W, 0, self.proj_size, h_full.ptr, v_full.ptr,
                                          self.xyz_full.ptr, None, mode=self.mode & 1)
        else:
            self.ctx.triangulate_maps_dev(h_full.ptr, v_full.ptr, self.plan.H, self.plan.W, 0, self.proj_size, self.xyz_full.ptr, None,
                                          mode=self.mode & 3)
        self.h_full, self.v_full = h_full, v_full

    def flush(self):
        """Finish the scan still in flight (its maps end up in h_full / v_full, its cloud in xyz_full)."""
        if self.kind in ("maps", "xyz") and self._pending is not None:
            with self._whole_image():
                self._finish(self._pending)
            self._pending = None

    def scan(self, d_band_stack: int, plane_stride: int, n_runs: int = 1, run_stride: int = 0, eps=1):
        """d_band_stack points at this rank's first row of frame 0.  "records": returns the total number of points (all
        ranks); "maps": returns None (the cloud is the dense XYZ map, count it with :meth:`count_valid`)."""
        if self.kind == "maps":
            return self.scan_maps(d_band_stack, plane_stride, n_runs, run_stride, eps)
        if self.kind == "xyz":
            self.flush()
            with self._whole_image():
                self._submit_xyz(d_band_stack, plane_stride, n_runs, run_stride, eps)
            return self.flush()
        with self._whole_image():
            return self._scan_records(d_band_stack, plane_stride, n_runs, run_stride, eps)

    def _scan_records(self, d_band_stack, plane_stride, n_runs, run_stride, eps):
        c, W = self.ctx, self.plan.W
        band_px = self.rows * W
        c.scan_dev(d_band_stack, n_runs, run_stride or self.N * plane_stride, plane_stride, self.N, self.rows, W, self.row0,
                   self.proj_size, self.xyz.ptr, None, self.maps.at(0), self.maps.at(band_px * 2), eps=eps, mode=self.mode)
        c.compact_records_dev(self.xyz.ptr, self.rows, W, self.row0, self.records.ptr, self.count.ptr)
        m = int(self.count.download((1,), np.uint64)[0])          # the host needs M_r for the displacements
        self.last_counts, total = exchange_records(self.exchange, self.records.ptr, m, self.all_records.ptr)
        return total

    def fetch(self, total: int):
        """Reassembled cloud on the host ("records"): RECORD_DTYPE array [M] in row-major (key) order."""
        self.ctx.synchronize()
        return self.all_records.download((total,), RECORD_DTYPE)

    def fetch_dense(self):
        """Reassembled products on the host ("maps" / "xyz"): (h int16 [H,W], v int16 [H,W], xyz float32 [H,W,3])."""
        self.ctx.synchronize()
        H, W = self.plan.H, self.plan.W
        return (self.h_full.download((H, W), np.int16), self.v_full.download((H, W), np.int16),
                self.xyz_full.download((H, W, 3), np.float32))
