"""N > 1 path with the real kernels, one process per rank, all ranks on the one GPU of the test box.

RCCL refuses two ranks on one device ("Duplicate GPU detected", ncclInvalidUsage), so the transport here is a host-staged double with
the RcclExchange protocol (device band -> shared host memory -> the other ranks' device buffers).  Everything else is the product path:
ShardPlan bands, decode_dev / scan_dev on a band with its row offset, the in-place map layout, triangulate_maps_dev over the reassembled
maps, compact_records_dev with row0, submit()/flush() pipelining over two buffer sets and four event slots.

The double has three modes, because a double that completes inside ``allgatherv_begin`` hides exactly the hazards the buffer sets exist for:
  * ``sync``     the exchange is complete when ``*_begin`` returns (what round 2 tested);
  * ``deferred`` ``*_begin`` only records the request -- the copy happens in ``wait(slot)``: the LATEST moment the product may rely on.
                 A missing ``wait``, a slot reused before its wait, or a single-buffered send band show up as wrong pixels;
  * ``thread``   a helper thread (its own context / stream) performs the request after a delay, ``wait(slot)`` joins it: the exchange
                 runs concurrently with whatever the rank enqueues next, like RCCL on its communication stream.

Small shapes (world 2 / 3 / 4, an empty band) run all strategies in every mode against the C oracle.  BASELINE.json configs[3] runs at
its own size -- 4096x3000x44, bench.py's calibration and device-generated stacks, world 2, 8 and (ragged bands) 7, all three strategies,
EVERY pixel of the reassembled result on every rank against ``oracle_c.scan_dense`` -- reference: what every rank must hold is the
content of /root/reference/scanner/triangulation/triangulate.py:52-64 for the whole image."""
import ctypes as C
import os
import queue
import sys
import threading
import time
import uuid

import numpy as np
import pytest

from conftest import PKG, ROOT, ext, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]
XYZ_RTOL = 1e-4           # BASELINE.json north_star: XYZ within 1e-4 relative


class Board:
    """What the rank processes share: a host buffer every band passes through, two barriers, a word per rank (created by the parent)."""

    def __init__(self, mpc, world, nbytes):
        import tempfile
        self.world, self.nbytes = world, max(4096, int(nbytes))
        d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        self.path = os.path.join(d, f"slgc_x_{uuid.uuid4().hex[:12]}")
        with open(self.path, "wb") as f:
            f.truncate(self.nbytes)
        self.bar = mpc.Barrier(world)
        self.words = mpc.Array("q", world)

    def handle(self):
        return (self.path, self.nbytes, self.bar, self.words)

    def close(self):
        try:
            os.remove(self.path)
        except OSError:
            pass


class StagedExchange:
    """RcclExchange protocol over device pointers, staged through shared host memory.  mode: sync | deferred | thread (module docstring)."""

    def __init__(self, ctx, native, rank, world, handle, mode="sync", delay_s=0.004):
        path, nbytes, self.bar, self.words = handle
        self.ctx, self.native, self.rank, self.nranks, self.mode, self.delay = ctx, native, rank, world, mode, delay_s
        self.host = np.memmap(path, dtype=np.uint8, mode="r+", shape=(nbytes,))
        self.base = self.host.ctypes.data
        self.pending = {}                                   # slot -> [request, ...] (deferred) or [threading.Event, ...] (thread)
        self.stats = {"begun": 0, "waited": 0, "max_pending": 0}
        self.q = None
        if mode == "thread":
            self.q = queue.Queue()
            self.err = None
            self.worker = threading.Thread(target=self._serve, daemon=True)
            self.worker.start()

    # ---- the staged copy itself (all ranks execute the same sequence of these)
    @staticmethod
    def _ptr(b):
        return b.ptr if hasattr(b, "ptr") else int(b)

    def _copy(self, c, send, recv, counts, displs):
        lib, n_mine = self.native.lib(), int(counts[self.rank])
        assert int(displs[-1]) + int(counts[-1]) <= self.host.size, "exchange larger than the shared staging buffer"
        if n_mine:
            c._ck(lib.slgc_d2h(c._h, C.c_void_p(self.base + int(displs[self.rank])), C.c_void_p(send), n_mine))
        self.bar.wait(timeout=300)
        for r in range(self.nranks):
            n, d = int(counts[r]), int(displs[r])
            if n and (r != self.rank or send != recv + d):
                c._ck(lib.slgc_h2d(c._h, C.c_void_p(recv + d), C.c_void_p(self.base + d), n))
        self.bar.wait(timeout=300)                          # nobody writes the staging buffer for the next exchange before everyone has read this one

    def _serve(self):
        hctx = self.native.Context(0)                       # the helper's own stream: the exchange really overlaps the rank's kernels
        while True:
            item = self.q.get()
            if item is None:
                break
            fn, done, delay = item
            try:
                if delay:
                    time.sleep(delay)
                fn(hctx)
            except BaseException as e:  # noqa: BLE001
                self.err = e
            done.set()
        hctx.close()

    def _run(self, fn, slot=None, delay=0.0):
        """Execute `fn(context)` now (sync), at wait(slot) (deferred) or on the helper thread (thread)."""
        if self.mode == "thread":
            self.ctx.synchronize()                          # RCCL's stream order: the collective starts after the kernels enqueued so far
            done = threading.Event()
            self.q.put((fn, done, delay))
            if slot is None:
                self._join(done)
            else:
                self.pending.setdefault(slot, []).append(done)
        elif self.mode == "deferred" and slot is not None:
            self.pending.setdefault(slot, []).append(fn)
        else:
            fn(self.ctx)
        if slot is not None:
            self.stats["begun"] += 1
            self.stats["max_pending"] = max(self.stats["max_pending"], sum(len(v) for v in self.pending.values()))

    def _join(self, done):
        assert done.wait(timeout=300), "staged exchange: helper thread timed out"
        if self.err is not None:
            raise self.err

    # ---- RcclExchange protocol
    def allgather_i64(self, value):
        out = []

        def fn(_c):
            self.words[self.rank] = int(value)
            self.bar.wait(timeout=300)
            out.extend(int(x) for x in self.words[:])
            self.bar.wait(timeout=300)
        self._run(fn)
        return out

    def allgatherv(self, d_send, d_recv, byte_counts, byte_displs):
        s, r, c, d = self._ptr(d_send), self._ptr(d_recv), list(byte_counts), list(byte_displs)
        self._run(lambda cx: self._copy(cx, s, r, c, d))

    def allgatherv_begin(self, d_send, d_recv, byte_counts, byte_displs, slot):
        assert slot not in self.pending or not self.pending[slot], f"slot {slot} reused before its wait"
        s, r, c, d = self._ptr(d_send), self._ptr(d_recv), list(byte_counts), list(byte_displs)
        self._run(lambda cx: self._copy(cx, s, r, c, d), slot=slot, delay=self.delay)

    def allgatherv_pair_begin(self, d_send_a, d_recv_a, d_send_b, d_recv_b, byte_counts, byte_displs, slot):
        assert slot not in self.pending or not self.pending[slot], f"slot {slot} reused before its wait"
        sa, ra, sb, rb = (self._ptr(x) for x in (d_send_a, d_recv_a, d_send_b, d_recv_b))
        c, d = list(byte_counts), list(byte_displs)

        def fn(cx):                                          # RcclExchange sends both maps in one group
            self._copy(cx, sa, ra, c, d)
            self._copy(cx, sb, rb, c, d)
        self._run(fn, slot=slot, delay=self.delay)

    def wait(self, slot):
        for item in self.pending.pop(slot, []):
            if self.mode == "thread":
                self._join(item)
            else:
                item(self.ctx)
            self.stats["waited"] += 1

    def barrier(self):
        self._run(lambda _c: self.bar.wait(timeout=300))

    def close(self):
        assert not any(self.pending.values()), f"exchanges begun but never waited for: slots {[s for s, v in self.pending.items() if v]}"
        if self.q is not None:
            self.q.put(None)
            self.worker.join(timeout=60)
        del self.host


def _setup_paths():
    for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)


def _check_dense(got, ref, what):
    gh, gv, gx = got
    fh, fv, fx = ref                                         # int64 / int16 [H,W] maps, float64 (3,H,W) points
    assert np.array_equal(gh, fh), f"{what}: h map differs in {(gh != fh).sum()} pixels"
    assert np.array_equal(gv, fv), f"{what}: v map differs in {(gv != fv).sum()} pixels"
    ok = (np.asarray(fh) != -1) & (np.asarray(fv) != -1)
    assert np.array_equal(np.isfinite(gx[..., 0]), ok), f"{what}: finite mask differs"
    want = np.moveaxis(np.asarray(fx), 0, -1)[ok]
    err = np.abs(gx[ok].astype(np.float64) - want) / np.maximum(np.abs(want), 1e-300)
    assert float(err.max()) <= XYZ_RTOL, f"{what}: worst elementwise relative XYZ error {float(err.max()):.3e}"


def _pipelined(sc, bufs, plane):
    """len(bufs) captures through submit()/flush(); the dense results in submission order."""
    outs = []
    for j, b in enumerate(bufs):
        sc.submit(b.ptr, plane)
        if j:
            outs.append(sc.fetch_dense())
    sc.flush()
    outs.append(sc.fetch_dense())
    return outs


# ------------------------------------------------------------------------------------------------ small shapes, every mode
def _worker_small(rank, world, handle, mode, H, W, N, q):
    try:
        _setup_paths()
        import oracle_c as oc
        import oracle_np as onp
        from scanner import _native, sharded
        from scanner import reference_calibration as rc
        ctx = _native.Context(0)
        ex = StagedExchange(ctx, _native, rank, world, handle, mode)
        plan = sharded.ShardPlan(H, W, world)
        row0, rows = plan.band(rank)
        K = rc.CAM_MTX.copy()
        K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 250.0, 250.0
        psize = (160, 120)
        pk = onp.scale_proj_mtx(rc.PROJ_MTX, psize, (1920, 1080))
        th = np.deg2rad(-20.0)
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        T = np.array([[0.25], [0.02], [0.04]])
        ctx.set_calibration(K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T)
        caps = [onp.synth_scene_int(N, H, W, seed=40 + j, noise=3 + j)[0] for j in range(3)]
        refs = [oc.scan_dense(cp, psize, K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T) for cp in caps]

        def upload_band(st):                                     # this rank holds only its row band of every frame
            band = np.ascontiguousarray(st[:, row0:row0 + rows])
            return ctx.alloc(max(16, band.nbytes)).upload(band) if rows else ctx.alloc(16)

        bufs = [upload_band(cp) for cp in caps]
        plane = max(1, rows * W)
        for wire in ("auto", "int16"):                           # auto = the 3-byte wire format (world > 1, L = 6 bits)
            scm = sharded.ShardedScanner(ctx, ex, plan, psize, N, mode=_native.TRI_EXACT, wire=wire)
            assert scm.wire == ("hv24" if wire == "auto" else "int16")
            assert scm.scan(bufs[0].ptr, plane) is None
            _check_dense(scm.fetch_dense(), refs[0], f"maps/{scm.wire} single")
            for j, got in enumerate(_pipelined(scm, bufs, plane)):
                _check_dense(got, refs[j], f"maps/{scm.wire} pipelined {j}")
        scx = sharded.ShardedScanner(ctx, ex, plan, psize, N, mode=_native.TRI_ALGEBRAIC, exchange_kind="xyz")
        scx.scan(bufs[2].ptr, plane)
        _check_dense(scx.fetch_dense(), refs[2], "xyz single")
        for j, got in enumerate(_pipelined(scx, bufs, plane)):
            _check_dense(got, refs[j], f"xyz pipelined {j}")
        sc = sharded.ShardedScanner(ctx, ex, plan, psize, N, mode=_native.TRI_EXACT, exchange_kind="records")
        total = sc.scan(bufs[1].ptr, plane)
        rec = sc.fetch(total)
        fh, fv, fx = refs[1]
        ok = (fh != -1) & (fv != -1)
        assert total == int(ok.sum()) and sum(sc.last_counts) == total and len(sc.last_counts) == world
        assert np.array_equal(rec["key"], np.nonzero(ok.ravel())[0])                 # band-major concatenation == row-major
        np.testing.assert_allclose(rec["xyz"], np.moveaxis(fx, 0, -1)[ok], rtol=XYZ_RTOL, atol=0)
        cam, P = sharded.to_reference_lists(rec, W, H)
        rcam, _, _ = oc.cam_proj_pts(fh, fv, (W, H), psize, None, order="x")
        assert np.array_equal(cam, rcam) and P.shape == (3, total)
        ex.barrier()
        stats = dict(ex.stats)
        ex.close()
        ctx.close()
        q.put((rank, "ok", (list(sc.last_counts), stats)))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


def _run_ranks(target, world, nbytes, args, timeout=900):
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    board = Board(mpc, world, nbytes)
    q = mpc.Queue()
    procs = [mpc.Process(target=target, args=(r, world, board.handle()) + tuple(args) + (q,)) for r in range(world)]
    try:
        for p in procs:
            p.start()
        results, deadline = [], time.time() + timeout
        while len(results) < world and time.time() < deadline:
            try:
                results.append(q.get(timeout=1.0))
            except queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):                  # a rank died without a report: do not wait for the others
                    break
                continue
            if results[-1][1] != "ok":
                break                                                                 # the others sit in a barrier that will never fill
        for rank, status, info in results:
            assert status == "ok", f"rank {rank}: {info}"
        assert len(results) == world, f"only {len(results)} of {world} ranks reported (exit codes {[p.exitcode for p in procs]})"
        return results
    finally:
        t_end = time.time() + 10
        for p in procs:
            p.join(timeout=max(0.1, t_end - time.time()))
        for p in procs:
            if p.is_alive():
                p.kill()                                                              # exact processes this test started
        board.close()


@pytest.mark.parametrize("world,H,mode", [(2, 48, "sync"), (3, 50, "deferred"), (4, 3, "thread"), (3, 50, "thread"),      # every mode, every layout (even / ragged / empty bands)
                                          ext(2, 48, "deferred"), ext(2, 48, "thread"), ext(3, 50, "sync"), ext(4, 3, "sync"), ext(4, 3, "deferred")])
def test_sharded_scanner_ranks_share_one_gpu(world, H, mode):
    """(4, 3): more ranks than rows -> one rank owns an empty band."""
    W, N = 128, 26
    results = _run_ranks(_worker_small, world, H * W * 16, (mode, H, W, N))
    assert len({tuple(info[0]) for _, _, info in results}) == 1
    for _, _, (_, stats) in results:
        assert stats["begun"] > 0
        if mode != "sync":                                                            # (sync: complete on return, nothing left to wait for)
            assert stats["begun"] == stats["waited"]                                  # every exchange that was begun was waited for
            assert stats["max_pending"] >= 2                                          # ... and exchanges really were in flight across calls


# ------------------------------------------------------------------------------------------------ BASELINE configs[3] at its own size
FULL = "c3_4096x3000x44"
SEEDS = (1, 2)                                                                        # bench.py's stacks[0], stacks[1]


def _worker_full(rank, world, handle, mode, ref_dir, q):
    try:
        _setup_paths()
        import bench
        import oracle_c as oc
        from scanner import _native, sharded
        W, H, pw, ph, N = bench.WORKLOADS[FULL]
        ctx = _native.Context(0)
        ctx.set_calibration(*bench.calibration(W, H, pw, ph))
        ex = StagedExchange(ctx, _native, rank, world, handle, mode)
        plan = sharded.ShardPlan(H, W, world)
        row0, rows = plan.band(rank)
        plane = rows * W
        bufs = []
        for seed in SEEDS:                                                            # each rank generates (only) its band, as bench.py does
            b = ctx.alloc(max(16, N * plane))
            ctx.synth_scene_dev(b.ptr, plane, N, H, W, row0=row0, rows=rows, seed=seed, noise=3, shadow=True)
            bufs.append(b)
        refs = [tuple(np.load(os.path.join(ref_dir, f"{k}{seed}.npy"), mmap_mode="r") for k in ("h", "v", "xyz")) for seed in SEEDS]
        order = (0, 1, 0)
        report = {}
        for kind, wire in (("maps", "int16"), ("maps", "hv24"), ("xyz", "int16")):
            sc = sharded.ShardedScanner(ctx, ex, plan, (pw, ph), N, mode=_native.TRI_ALGEBRAIC, exchange_kind=kind, wire=wire)
            sc.scan(bufs[1].ptr, plane)
            _check_dense(sc.fetch_dense(), refs[1], f"{kind}/{wire} world {world} single scan")
            for j, got in enumerate(_pipelined(sc, [bufs[i] for i in order], plane)):
                _check_dense(got, refs[order[j]], f"{kind}/{wire} world {world} pipelined scan {j}")
            report[f"{kind}/{wire}"] = ctx.last_scan_path()
            if rank == world - 1:
                # ... and bit for bit what ONE GPU computes for the whole image (the ray-table choice is the whole image's for every band:
                # slgc_tune "image_rows"): the last rank scans the full stack alone with the fused kernel
                sc.scan(bufs[0].ptr, plane)
                gh, gv, gx = sc.fetch_dense()
                px = W * H
                full, m1, x1 = ctx.alloc(N * px), ctx.alloc(px * 4), ctx.alloc(px * 12)
                ctx.synth_scene_dev(full.ptr, px, N, H, W, row0=0, rows=H, seed=SEEDS[0], noise=3, shadow=True)
                ctx.scan_dev(full.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), x1.ptr, None, m1.at(0), m1.at(px * 2), mode=_native.TRI_ALGEBRAIC)
                ctx.synchronize()
                assert ctx.last_scan_path()["path"] == "fused"
                assert np.array_equal(gh, m1.download((H, W), np.int16)) and np.array_equal(gv, m1.download((H, W), np.int16, px * 2))
                one = x1.download((H, W, 3), np.float32)
                assert np.array_equal(gx.view(np.uint32), one.view(np.uint32)), f"{kind}/{wire}: XYZ of {world} ranks differs from the single-GPU scan"
                for b in (full, m1, x1):
                    b.free()
            else:
                sc.scan(bufs[0].ptr, plane)                                           # (every rank takes part in the exchange of that extra scan)
            del sc
        sc = sharded.ShardedScanner(ctx, ex, plan, (pw, ph), N, mode=_native.TRI_ALGEBRAIC, exchange_kind="records")
        total = sc.scan(bufs[0].ptr, plane)
        rec = sc.fetch(total)
        fh, fv, fx = refs[0]
        ok = (np.asarray(fh) != -1) & (np.asarray(fv) != -1)
        assert total == int(ok.sum()) and sum(sc.last_counts) == total
        assert np.array_equal(rec["key"], np.nonzero(ok.ravel())[0])
        want = np.moveaxis(np.asarray(fx), 0, -1)[ok]
        err = np.abs(rec["xyz"].astype(np.float64) - want) / np.maximum(np.abs(want), 1e-300)
        assert float(err.max()) <= XYZ_RTOL
        if rank == 0:                                                                 # the reference's x-major order, recovered from the keys
            cam, P = sharded.to_reference_lists(rec, W, H)
            rcam, _, _ = oc.cam_proj_pts(np.asarray(fh), np.asarray(fv), (W, H), (pw, ph), None, order="x")
            assert np.array_equal(cam, rcam) and P.shape == (3, total)
        ex.barrier()
        stats = dict(ex.stats)
        ex.close()
        ctx.close()
        q.put((rank, "ok", (int(total), stats, report)))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


@pytest.fixture(scope="module")
def full_size_reference(tmp_path_factory):
    """bench.py's calibration and its first two device-generated stacks at 4096x3000x44, decoded and triangulated by the C oracle on the
    host cores, once for all worlds: int16 maps + float64 (3,H,W) points as .npy files the rank processes map."""
    import bench
    import oracle_c as oc
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS[FULL]
    calib = bench.calibration(W, H, pw, ph)
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path_factory.mktemp("ref"))
    ref_dir = os.path.join(d, f"slgc_ref_{uuid.uuid4().hex[:10]}")
    os.makedirs(ref_dir)
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    oc.set_threads(max(1, min(64, n)))
    ctx = _native.Context(0)
    try:
        px = W * H
        stack = ctx.alloc(N * px)
        for seed in SEEDS:
            ctx.synth_scene_dev(stack.ptr, px, N, H, W, row0=0, rows=H, seed=seed, noise=3, shadow=True)
            ctx.synchronize()
            fh, fv, fx = oc.scan_dense(stack.download((N, H, W), np.uint8), (pw, ph), *calib)
            np.save(os.path.join(ref_dir, f"h{seed}.npy"), fh.astype(np.int16))
            np.save(os.path.join(ref_dir, f"v{seed}.npy"), fv.astype(np.int16))
            np.save(os.path.join(ref_dir, f"xyz{seed}.npy"), fx)
            del fh, fv, fx
    finally:
        ctx.close()
        oc.set_threads(1)
    yield ref_dir
    for f in os.listdir(ref_dir):
        os.remove(os.path.join(ref_dir, f))
    os.rmdir(ref_dir)


@pytest.mark.parametrize("world,mode", [(2, "thread"), (8, "deferred"), ext(7, "deferred")])
def test_configs3_full_size_every_pixel(full_size_reference, world, mode):
    """BASELINE.json configs[3]: 4096x3000x44 row-sharded over `world` ranks (7: ragged bands, 428 / 429 rows), maps (int16 and 3-byte
    wire), xyz and records strategies, one scan and three pipelined scans each, every pixel of what every rank ends up holding."""
    import bench
    W, H, _, _, _ = bench.WORKLOADS[FULL]
    results = _run_ranks(_worker_full, world, W * H * 16, (mode, full_size_reference), timeout=1500)
    assert len({info[0] for _, _, info in results}) == 1                              # same point count on every rank
    for _, _, (_, stats, report) in results:
        assert stats["begun"] == stats["waited"] > 0 and stats["max_pending"] >= 2
        assert report["xyz/int16"]["path"] == "fused" and report["xyz/int16"]["node_table"]          # band kernels: the whole image's ray-table choice
        assert report["maps/int16"]["node_table"]
