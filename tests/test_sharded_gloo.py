"""N > 1 path on CPU: world_size-2 (and 3) gloo run of the row-sharding plan, the counts + all-gatherv exchange protocol
and the reassembly / x-major reordering code of scanner/sharded.py.  The per-band compute is done by the oracle here
(this is a CPU test of the host logic); on the GPU box the same functions drive RcclExchange with device pointers."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class GlooExchange:
    """Test double with the RcclExchange protocol, over torch.distributed(gloo) and NumPy buffers."""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()

    def allgather_i64(self, value):
        t = self.torch.tensor([value], dtype=self.torch.int64)
        out = [self.torch.zeros(1, dtype=self.torch.int64) for _ in range(self.nranks)]
        self.dist.all_gather(out, t)
        return [int(x[0]) for x in out]

    def allgatherv(self, send, recv, byte_counts, byte_displs):
        """Same contract as slgc_comm_allgatherv: one broadcast per contributing rank into recv[displ : displ+count]."""
        rb = recv.view(np.uint8).reshape(-1)
        sb = send.view(np.uint8).reshape(-1)
        for r in range(self.nranks):
            n = byte_counts[r]
            if n == 0:
                continue
            if r == self.rank:
                rb[byte_displs[r]:byte_displs[r] + n] = sb[:n]
            t = self.torch.from_numpy(rb[byte_displs[r]:byte_displs[r] + n])
            self.dist.broadcast(t, src=r)

    def allgatherv_begin(self, send, recv, byte_counts, byte_displs, slot):
        self.allgatherv(send, recv, byte_counts, byte_displs)          # CPU double: completes immediately

    def wait(self, slot):
        pass

    def barrier(self):
        self.dist.barrier()


def _worker(rank, world, port, H, W, N, q):
    try:
        for p in (PKG, os.path.join(ROOT, "oracle")):
            sys.path.insert(0, p)
        import torch
        import torch.distributed as dist
        import oracle_c as oc
        import oracle_np as onp
        from scanner import reference_calibration as rc
        from scanner import sharded
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        ex = GlooExchange(dist, torch)
        plan = sharded.ShardPlan(H, W, world)
        row0, rows = plan.band(rank)
        st, _, _ = onp.synth_scene_int(N, H, W, seed=4)
        K = rc.CAM_MTX.copy()
        K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 200.0, 200.0
        psize = (200, 150)
        pk = onp.scale_proj_mtx(rc.PROJ_MTX, psize, (1920, 1080))
        th = np.deg2rad(-20.0)
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        T = np.array([[0.25], [0.02], [0.04]])
        # ---- per-band compute (oracle stands in for scan_dev + compact_dev): band of the stack only
        band = np.ascontiguousarray(st[:, row0:row0 + rows])
        Kb = K.copy()
        Kb[1, 2] -= row0                                     # camera y of local row 0 is row0
        hp, vp, xyz = oc.scan_dense(band, psize, Kb, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T) if rows else (None, None, np.zeros((3, 0, W)))
        dense = np.moveaxis(xyz, 0, -1).astype(np.float32).reshape(-1, 3)
        ok = np.isfinite(dense[:, 0])
        rec = np.zeros(int(ok.sum()), sharded.RECORD_DTYPE)
        rec["xyz"] = dense[ok]
        rec["key"] = (np.nonzero(ok)[0] + row0 * W).astype(np.uint32)
        # ---- the code under test: layout + exchange + reassembly
        all_rec = np.zeros(H * W, sharded.RECORD_DTYPE)
        counts, total = sharded.exchange_records(ex, rec, len(rec), all_rec)
        all_pts, all_keys = all_rec["xyz"], all_rec["key"]
        ex.barrier()
        # ---- reference: the whole image in one go
        fh, fv, fxyz = oc.scan_dense(st, psize, K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T)
        fdense = np.moveaxis(fxyz, 0, -1).astype(np.float32).reshape(-1, 3)
        fok = np.isfinite(fdense[:, 0])
        assert total == int(fok.sum()) and sum(counts) == total and len(counts) == world
        assert np.array_equal(all_keys[:total], np.nonzero(fok)[0].astype(np.uint32))       # band-major == row-major
        np.testing.assert_allclose(all_pts[:total], fdense[fok], rtol=1e-6)
        cam, P = sharded.to_reference_lists(all_rec[:total], W, H)
        rcam, _, _ = oc.cam_proj_pts(fh, fv, (W, H), psize, None, order="x")
        assert np.array_equal(cam, rcam) and P.shape == (3, total) and P.dtype == np.float64
        # ---- default strategy: in-place all-gatherv of the int16 map bands, then every rank triangulates the full maps
        h_full = np.full((H, W), -7, np.int16)
        v_full = np.full((H, W), -7, np.int16)
        if rows:
            h_full[row0:row0 + rows] = hp
            v_full[row0:row0 + rows] = vp
        mc, md = sharded.map_band_layout(plan)
        assert sum(mc) == H * W * 2 and md[rank] == row0 * W * 2
        sharded.exchange_map_bands(ex, plan, h_full, v_full, lambda buf, o: buf.reshape(-1).view(np.uint8)[o:])
        ex.barrier()
        assert np.array_equal(h_full, fh) and np.array_equal(v_full, fv)                  # bit-exact maps on every rank
        # ---- "xyz" strategy: every rank fills its band of full-size maps AND dense XYZ, three in-place band all-gathers
        h2, v2 = np.full((H, W), -9, np.int16), np.full((H, W), -9, np.int16)
        x2 = np.full((H, W, 3), 123.0, np.float32)
        if rows:
            h2[row0:row0 + rows], v2[row0:row0 + rows] = hp, vp
            x2[row0:row0 + rows] = np.moveaxis(xyz, 0, -1).astype(np.float32)
        xc, xd = sharded.band_layout(plan, 12)
        assert sum(xc) == H * W * 12 and xd[rank] == row0 * W * 12
        sharded.exchange_bands(ex, plan, [(h2, 2), (v2, 2), (x2, 12)], lambda buf, o: buf.reshape(-1).view(np.uint8)[o:])
        ex.barrier()
        assert np.array_equal(h2, fh) and np.array_equal(v2, fv)
        assert np.array_equal(x2.reshape(-1, 3), fdense, equal_nan=True)                   # band by band == the whole image in one go
        dist.destroy_process_group()
        q.put((rank, "ok", counts))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world,H", [(2, 48), (3, 50)])
def test_sharded_exchange_gloo(world, H):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, 64, 26, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in results:
        assert status == "ok", f"rank {rank}: {info}"
    assert len({tuple(info) for _, _, info in results}) == 1          # every rank saw the same counts


def test_c_shard_band_matches_python_plan():
    """slgc_shard_band (the plan slgc_scan_sharded_dev uses) == ShardPlan.band for every rank, ragged and empty bands."""
    from scanner import _native, sharded
    for H in (0, 1, 3, 7, 48, 50, 1080, 3000):
        for G in (1, 2, 3, 4, 7, 8):
            plan = sharded.ShardPlan(H, 16, G)
            assert [_native.shard_band(H, G, r) for r in range(G)] == plan.bands()
    with pytest.raises(ValueError):
        _native.shard_band(10, 2, 2)


def test_shard_plan_and_layout():
    from scanner import sharded
    for H, G in ((3000, 8), (3000, 7), (5, 8), (1080, 4)):
        plan = sharded.ShardPlan(H, 64, G)
        bands = plan.bands()
        assert bands[0][0] == 0 and sum(r for _, r in bands) == H
        for (a0, ar), (b0, _) in zip(bands, bands[1:]):
            assert a0 + ar == b0
        assert max(r for _, r in bands) - min(r for _, r in bands) <= 1
    bc, bd, tot = sharded.gather_layout([3, 0, 5], sharded.RECORD_BYTES)
    assert bc == [48, 0, 80] and bd == [0, 48, 48] and tot == 128 and sharded.RECORD_DTYPE.itemsize == sharded.RECORD_BYTES
    with pytest.raises(ValueError):
        sharded.gather_layout([1, -1], 4)
    keys = np.array([0 * 4 + 1, 0 * 4 + 3, 1 * 4 + 0, 2 * 4 + 1], dtype=np.uint32)      # W=4: (x,y) = (1,0),(3,0),(0,1),(1,2)
    perm = sharded.x_major_permutation(keys, 4, 3)
    assert list(perm) == [2, 0, 3, 1]                                                     # x=0 | x=1 (y=0,2) | x=3


def _uid_worker(rank, key, tmpdir, q):
    try:
        sys.path.insert(0, PKG)
        from scanner import sharded
        uid, path = sharded.share_unique_id(rank, lambda: bytes(range(128)), key=key, timeout_s=30, directory=tmpdir)
        q.put((rank, uid == bytes(range(128)), path))
    except Exception as e:  # noqa: BLE001
        q.put((rank, False, repr(e)))


def test_unique_id_rendezvous_file(tmp_path):
    """The torch-free rendezvous bench.py uses under torch.distributed.run: rank 0 publishes atomically, late and early ranks poll."""
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uid_worker, args=(r, "testkey", str(tmp_path), q)) for r in (2, 1)]     # non-zero ranks first
    for p in procs:
        p.start()
    time.sleep(0.5)
    p0 = ctx.Process(target=_uid_worker, args=(0, "testkey", str(tmp_path), q))
    p0.start()
    res = [q.get(timeout=60) for _ in range(3)]
    for p in procs + [p0]:
        p.join(timeout=30)
    assert all(ok for _, ok, _ in res), res
    assert len({path for _, _, path in res}) == 1
    from scanner import sharded
    with pytest.raises(RuntimeError, match="timed out"):
        sharded.share_unique_id(1, None, key="nobody", timeout_s=0.2, directory=str(tmp_path))
    with pytest.raises(ValueError):
        sharded.share_unique_id(0, lambda: b"short", key="bad", directory=str(tmp_path))
