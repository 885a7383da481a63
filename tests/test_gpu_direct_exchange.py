"""The direct exchange (csrc/direct.hip, slgc_direct_*) driven through the C-ABI by several rank PROCESSES that share the one GPU of the test
box (-m gpu).  No RCCL anywhere in this file: set-up, flags and the small collectives go through the job's shared-memory segment, the bands
through hipIpcMemHandle mappings of the peers' buffers -- exactly what runs on a node, minus the xGMI links.

Covered: the host collectives; ragged layouts with an empty shard, odd offsets and lengths (head / tail bytes of the push kernel) and untouched
gaps; 25 exchanges in a row on the same buffers under the release protocol (a band that arrives early or a buffer overwritten before it was
consumed = a wrong byte); three buffers in one exchange; the pipelined ShardedScanner (maps int16 / hv24, xyz) against the C oracle on
every rank and every scan; argument errors; and a peer that never shows up: the call fails after the GPU-side timeout, the GPU is not hung."""
import os
import sys
import time
import uuid

import numpy as np
import pytest

from conftest import PKG, ROOT, ext, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]


def _setup(rank, world, key):
    for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
        sys.path.insert(0, p)
    from scanner import _native, sharded
    ctx = _native.Context(0)
    ex = sharded.DirectExchange(ctx, rank, world, key)
    return _native, sharded, ctx, ex


def _worker_exchange(rank, world, key, q):
    try:
        _native, sharded, ctx, ex = _setup(rank, world, key)
        assert ex.allgather_i64(100 + rank) == [100 + r for r in range(world)]
        ex.barrier()
        # ---- ragged layout: an empty shard, odd offsets / lengths, 8-byte gaps that must stay untouched
        counts = [0 if r == 1 else 1003 + 37 * r for r in range(world)]
        displs = [int(x) for x in np.cumsum([3] + [c + 8 for c in counts[:-1]])]
        total = displs[-1] + counts[-1] + 64
        bufs = [ctx.alloc(total).zero() for _ in range(3)]
        ex.register(bufs)
        for it in range(int(os.environ.get("SLGC_DIRECT_TEST_ROUNDS", "25"))):      # (soak: SLGC_DIRECT_TEST_ROUNDS=400)
            # the rank's band of every buffer gets this round's pattern; everything before (the previous round's readers: the downloads
            # below, already synchronised) is done with the buffers -> release, then push
            ex.release(bufs)
            for k, b in enumerate(bufs):
                if counts[rank]:
                    b.upload(np.full(counts[rank], (7 * it + 31 * rank + 3 * k) % 251 + 1, np.uint8), displs[rank])
            if it % 2:
                ctx.direct_allgatherv_begin([b.ptr for b in bufs], [(counts, displs)] * 3, 2)                  # three buffers, one exchange
                ex.wait(2)
            else:
                ex.allgatherv_pair_begin(bufs[0].at(displs[rank]), bufs[0], bufs[1].at(displs[rank]), bufs[1], counts, displs, 0)
                ex.allgatherv_begin(bufs[2].at(displs[rank]), bufs[2], counts, displs, 1)
                ex.wait(0)
                ex.wait(1)
            ctx.synchronize()
            for k, b in enumerate(bufs):
                got = b.download((total,), np.uint8)
                for r in range(world):
                    want = (7 * it + 31 * r + 3 * k) % 251 + 1
                    assert (got[displs[r]:displs[r] + counts[r]] == want).all(), (it, k, r, got[displs[r]:displs[r] + 8], want)
                    assert not got[displs[r] + counts[r]:displs[r] + counts[r] + 8].any(), (it, k, r)
                assert not got[:3].any()
        # ---- argument errors leave the exchange usable
        other = ctx.alloc(4096)
        with pytest.raises((ValueError, _native.SlgcError), match="not registered"):
            ctx.direct_allgatherv_begin([other.ptr], [(counts, displs)], 0)
        with pytest.raises((ValueError, _native.SlgcError), match="outside"):
            ctx.direct_allgatherv_begin([bufs[0].ptr], [([total + 1] * world, [0] * world)], 0)
        with pytest.raises((ValueError, _native.SlgcError), match="twice"):
            ctx.direct_allgatherv_begin([bufs[0].ptr, bufs[0].ptr], [(counts, displs)] * 2, 0)
        with pytest.raises(ValueError, match="in place"):
            ex.allgatherv_begin(bufs[0].at(displs[rank] + 1), bufs[0], counts, displs, 0)
        ex.barrier()
        ctx.close()
        q.put((rank, "ok", 0))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


def _worker_scanner(rank, world, key, H, q):
    try:
        _native, sharded, ctx, ex = _setup(rank, world, key)
        import oracle_c as oc
        import oracle_np as onp
        from scanner import reference_calibration as rc
        N, W = 26, 128
        K = rc.CAM_MTX.copy()
        K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 250.0, 250.0
        psize = (160, 120)
        pk = onp.scale_proj_mtx(rc.PROJ_MTX, psize, (1920, 1080))
        th = np.deg2rad(-20.0)
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        T = np.array([[0.25], [0.02], [0.04]])
        ctx.set_calibration(K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T)
        plan = sharded.ShardPlan(H, W, world)
        row0, rows = plan.band(rank)
        caps = [onp.synth_scene_int(N, H, W, seed=30 + j, noise=3 + j)[0] for j in range(3)]
        refs = [oc.scan_dense(cp, psize, K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T) for cp in caps]
        bufs = []
        for cp in caps:
            b = np.ascontiguousarray(cp[:, row0:row0 + rows])
            bufs.append(ctx.alloc(max(16, b.nbytes)).upload(b) if rows else ctx.alloc(16))
        total_ok = 0
        # two rounds of three scanners = 24 registrations against 16 slots: close() must hand the slots back (slgc_direct_unregister)
        for kind, wire in (("maps", "int16"), ("maps", "hv24"), ("xyz", "int16")) * 2:
            sc = sharded.ShardedScanner(ctx, ex, plan, psize, N, mode=_native.TRI_ALGEBRAIC, exchange_kind=kind, wire=wire)
            outs = []
            for j in range(7):                                                       # seven submits: every buffer set is re-used three times
                sc.submit(bufs[j % 3].ptr, max(1, rows * W))
                if j:
                    outs.append(sc.fetch_dense())
            sc.flush()
            outs.append(sc.fetch_dense())
            sc.scan(bufs[1].ptr, max(1, rows * W))                                   # ... and the unpipelined form on top
            outs.append(sc.fetch_dense())
            for j, (gh, gv, gx) in enumerate(outs):
                rh, rv, rx = refs[j % 3] if j < 7 else refs[1]
                okj = (rh != -1) & (rv != -1)
                assert np.array_equal(gh, rh) and np.array_equal(gv, rv), (kind, wire, j)
                assert np.array_equal(np.isfinite(gx[..., 0]), okj), (kind, wire, j)
                np.testing.assert_allclose(gx[okj], np.moveaxis(rx, 0, -1)[okj], rtol=1e-4, atol=0)
                total_ok += int(okj.sum())
            with pytest.raises(_native.SlgcError, match="registered"):              # peers hold mappings of it: freeing it now is refused
                sc._exchanged[0][0].free()
            sc.close()                                                               # collective: unregister on every rank, then free
        ex.barrier()
        ctx.close()
        q.put((rank, "ok", total_ok))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


def _worker_lost_peer(rank, world, key, q):
    """Rank 1 registers and then never exchanges: rank 0's wait gives up (its peer's host never submits: the START deadline), and from then on
    nothing of that exchange comes back as a result -- synchronize, the download and the next exchange all fail; the GPU itself is fine."""
    try:
        os.environ["SLGC_DIRECT_TIMEOUT_S"] = "1.0"
        os.environ["SLGC_DIRECT_START_TIMEOUT_S"] = "1.0"
        _native, sharded, ctx, ex = _setup(rank, world, key)
        buf = ctx.alloc(8192).zero()
        ex.register([buf])
        if rank == 0:
            t0 = time.time()
            ex.allgatherv_begin(buf.at(0), buf, [4096, 4096], [0, 4096], 0)
            ex.wait(0)
            with pytest.raises(_native.SlgcError, match="timed out"):               # the polling kernel has given up: the stream drains, the scan has FAILED
                ctx.synchronize()
            waited = time.time() - t0
            assert 0.8 < waited < 15.0, waited
            with pytest.raises(_native.SlgcError, match="timed out"):
                buf.download((16,), np.uint8)                                        # bytes of a timed-out exchange are not a result
            with pytest.raises(_native.SlgcError, match="timed out"):
                ex.allgatherv_begin(buf.at(0), buf, [4096, 4096], [0, 4096], 1)
            # the GPU itself is fine: once the dead exchange is gone, ordinary calls work on this context
            ctx.direct_destroy()
            assert buf.download((16,), np.uint8).sum() == 0
            ctx.synchronize()
        else:
            time.sleep(4.0)                                                         # stays alive (mappings valid), never pushes
        ctx.close()
        q.put((rank, "ok", 0))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


def _worker_late_peer(rank, world, key, q):
    """Hosts out of step by more than the (short) GPU-side deadline: rank 1 submits its side 3 s after rank 0, with SLGC_DIRECT_TIMEOUT_S = 1.
    The short deadline only runs once the peer's host has submitted, so the exchange completes (RCCL would simply have waited, too) -- on the
    first exchange (rank 0's wait polls for a band nobody has started) and on the second (rank 0's GATE polls for a release nobody has submitted)."""
    try:
        os.environ["SLGC_DIRECT_TIMEOUT_S"] = "1.0"
        _native, sharded, ctx, ex = _setup(rank, world, key)
        buf = ctx.alloc(8192).zero()
        ex.register([buf])
        for it in range(2):
            if rank == 1:
                time.sleep(3.0)
            if it:
                ex.release([buf])
            buf.upload(np.full(4096, 10 * it + rank + 1, np.uint8), 4096 * rank)
            ex.allgatherv_begin(buf.at(4096 * rank), buf, [4096, 4096], [0, 4096], 0)
            ex.wait(0)
            ctx.synchronize()
            got = buf.download((8192,), np.uint8)
            assert (got[:4096] == 10 * it + 1).all() and (got[4096:] == 10 * it + 2).all(), (it, got[::2048])
        ex.barrier()
        ex.unregister([buf])
        buf.free()
        ctx.close()
        q.put((rank, "ok", 0))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


def _worker_register_failure(rank, world, key, q):
    """Ranks register buffers of different sizes: every rank gets the error, nothing of the failed registration stays behind (the same buffers
    register fine afterwards, in the same slot), and unregistered buffers can be freed."""
    try:
        _native, sharded, ctx, ex = _setup(rank, world, key)
        a, b = ctx.alloc(4096 * (1 + rank)).zero(), ctx.alloc(8192).zero()
        with pytest.raises((ValueError, _native.SlgcError), match="different sizes"):
            ctx.direct_register(a.ptr, a.nbytes)
        a.free()                                                                    # not registered: free is allowed
        for _ in range(20):                                                         # 20 > 16 slots: register / unregister recycles them
            ctx.direct_register(b.ptr, b.nbytes)
            with pytest.raises((ValueError, _native.SlgcError), match="already registered"):
                ctx.direct_register(b.ptr, b.nbytes)
            ctx.direct_unregister(b.ptr)
        with pytest.raises((ValueError, _native.SlgcError), match="not registered"):
            ctx.direct_unregister(b.ptr)
        ctx.direct_register(b.ptr, b.nbytes)
        b.upload(np.full(4096, rank + 1, np.uint8), 4096 * rank)
        ctx.direct_allgatherv_begin([b.ptr], [([4096, 4096], [0, 4096])], 0)
        ctx.direct_wait(0)
        ctx.synchronize()
        got = b.download((8192,), np.uint8)
        assert (got[:4096] == 1).all() and (got[4096:] == 2).all()
        ex.barrier()
        ctx.direct_unregister(b.ptr)
        b.free()
        ctx.close()
        q.put((rank, "ok", 0))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


def _run(target, world, *args, timeout=120):
    import multiprocessing as mp
    import queue as pyqueue
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    key = f"t{uuid.uuid4().hex[:12]}"
    procs = [mpc.Process(target=target, args=(r, world, key, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    results, deadline = [], time.time() + timeout
    try:
        while len(results) < world and time.time() < deadline:
            try:
                results.append(q.get(timeout=1.0))
            except pyqueue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs) and not any(p.is_alive() for p in procs):
                    break
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()                                                             # exact processes this test started
    assert len(results) == world, f"only {len(results)} of {world} ranks reported: {results}"
    bad = [r for r in results if r[1] != "ok"]
    assert not bad, "\n".join(str(b[2]) for b in bad)
    return results


@pytest.mark.parametrize("world", [ext(2), 4] + ([8] if os.environ.get("SLGC_DIRECT_TEST_ROUNDS") else []))
def test_direct_exchange_layouts_and_release_protocol(world):
    _run(_worker_exchange, world, timeout=120 + 2 * int(os.environ.get("SLGC_DIRECT_TEST_ROUNDS", "25")))


@pytest.mark.parametrize("world,H", [ext(2, 48), (3, 50), (5, 3)])
def test_sharded_scanner_over_the_direct_exchange_matches_the_oracle(world, H):
    """(5, 3): more ranks than rows -> ranks with an empty band take part in every exchange."""
    res = _run(_worker_scanner, world, H)
    assert all(r[2] > 0 for r in res)


def test_a_lost_peer_is_a_failed_call_not_a_hung_gpu():
    _run(_worker_lost_peer, 2, timeout=60)


def test_hosts_out_of_step_wait_for_each_other():
    _run(_worker_late_peer, 2, timeout=60)


def test_registration_is_failure_atomic_and_slots_are_recycled():
    _run(_worker_register_failure, 2, timeout=60)
