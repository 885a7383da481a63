"""The tile-interleaved stack layout ([tile][N][2^k], slgc_tune "stack_tile_log2"; include/slgc_bench.h: an A/B of the layout question, not a product layout) against the planar stack the reference's
[N][H][W] maps onto (src/3-capture_decode.py:68-70): the same kernels must give the same bits -- maps from the decode kernel, maps + XYZ from the
fused and the two-kernel scan -- at BASELINE configs[1] / [2], on ragged sizes, with two runs, for several tile sizes; the two ways into the
layout (slgc_tile_stack_dev, slgc_to_gray_tiled_dev) agree byte for byte; host-buffer entry points are unaffected by the setting."""
import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

if conftest.has_gpu():
    import bench
    from scanner import _native


@pytest.fixture()
def ctx():
    c = _native.Context(0)
    yield c
    c.close()


def _tiled_numpy(st, k):
    """NumPy statement of the layout: pixel p of frame f at ((p >> k) * N + f) << k | (p & (2^k - 1))."""
    N = st.shape[0]
    flat = st.reshape(N, -1)
    npix = flat.shape[1]
    piece = 1 << k
    tiles = -(-npix // piece)
    out = np.zeros((tiles, N, piece), np.uint8)
    pad = np.zeros((N, tiles * piece), np.uint8)
    pad[:, :npix] = flat
    out[:] = pad.reshape(N, tiles, piece).transpose(1, 0, 2)
    return out.reshape(-1)


@pytest.mark.parametrize("W,H,N,k", [(1920, 1080, 44, 12), (4096, 3000, 44, 12), (4096, 3000, 44, 10), (516, 1031, 46, 8), (128, 50, 42, 16), (64, 48, 26, 9)])
def test_tiled_stack_same_bits_as_planar(ctx, W, H, N, k):
    px = W * H
    pw, ph = (1920, 1200) if W > 1920 else (1280, 800)
    ctx.set_calibration(*bench.calibration(W, H, pw, ph))
    planar = ctx.alloc(N * px)
    ctx.synth_scene_dev(planar.ptr, px, N, H, W, seed=5, noise=3, shadow=True)
    tb = ctx.tiled_stack_bytes(N, px, k)
    assert tb == -(-px // (1 << k)) * N * (1 << k)
    tiled = ctx.alloc(tb).zero()
    ctx.tile_stack_dev(planar.ptr, px, N, px, k, tiled.ptr)
    ctx.synchronize()
    if px <= 1920 * 1080:                                        # the device layout against its NumPy statement
        st = planar.download((N, H, W), np.uint8)
        assert np.array_equal(tiled.download((tb,), np.uint8), _tiled_numpy(st, k))
    maps_p, maps_t = ctx.alloc(px * 4), ctx.alloc(px * 4)
    xyz_p, xyz_t = ctx.alloc(px * 12), ctx.alloc(px * 12)
    piece = 1 << k
    for mode in (_native.TRI_ALGEBRAIC, _native.TRI_ALGEBRAIC | _native.TRI_SPLIT):
        for b in (maps_p, maps_t, xyz_p, xyz_t):
            b.zero()
        ctx.tune("stack_tile_log2", 0)
        ctx.scan_dev(planar.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz_p.ptr, None, maps_p.at(0), maps_p.at(px * 2), mode=mode)
        path_p = ctx.last_scan_path()["path"]
        ctx.tune("stack_tile_log2", k)
        ctx.scan_dev(tiled.ptr, 1, tb, piece, N, H, W, 0, (pw, ph), xyz_t.ptr, None, maps_t.at(0), maps_t.at(px * 2), mode=mode)
        assert ctx.last_scan_path()["path"] == path_p and path_p in ("fused", "split")
        ctx.synchronize()
        a, b = maps_p.download((2, H, W), np.int16), maps_t.download((2, H, W), np.int16)
        assert np.array_equal(a, b) and (a != -1).any()
        assert np.array_equal(xyz_p.download((px * 3,), np.uint32), xyz_t.download((px * 3,), np.uint32))
    # the decode kernel alone
    maps_t.zero()
    ctx.decode_dev(tiled.ptr, 1, tb, piece, N, H, W, maps_t.at(0), maps_t.at(px * 2))
    ctx.synchronize()
    assert np.array_equal(maps_t.download((2, H, W), np.int16), a)
    # wrong plane_stride with the setting on: refused, not misread
    with pytest.raises((ValueError, _native.SlgcError), match="plane piece"):
        ctx.decode_dev(tiled.ptr, 1, tb, px, N, H, W, maps_t.at(0), maps_t.at(px * 2))
    ctx.tune("stack_tile_log2", 0)
    for b in (planar, tiled, maps_p, maps_t, xyz_p, xyz_t):
        b.free()


def test_tiled_two_runs_and_host_entry_points_unaffected(ctx):
    W, H, N, k = 256, 96, 44, 10
    px = W * H
    ctx.set_calibration(*bench.calibration(W, H, 1280, 800))
    piece, tb = 1 << k, ctx.tiled_stack_bytes(N, px, k)
    planar, tiled = ctx.alloc(2 * N * px), ctx.alloc(2 * tb).zero()
    for r in range(2):
        ctx.synth_scene_dev(planar.at(r * N * px), px, N, H, W, seed=11 + r, noise=4 + r, shadow=True)
        ctx.tile_stack_dev(planar.at(r * N * px), px, N, px, k, tiled.at(r * tb))
    mp, mt = ctx.alloc(px * 4), ctx.alloc(px * 4)
    ctx.decode_dev(planar.ptr, 2, N * px, px, N, H, W, mp.at(0), mp.at(px * 2))
    ctx.tune("stack_tile_log2", k)
    ctx.decode_dev(tiled.ptr, 2, tb, piece, N, H, W, mt.at(0), mt.at(px * 2))
    ctx.synchronize()
    want = mp.download((2, H, W), np.int16)
    assert np.array_equal(mt.download((2, H, W), np.int16), want)
    # host-buffer decode on the same context while the setting is on: its uploaded stack is planar and is read as such
    st = planar.download((2, N, H, W), np.uint8)
    hp, vp = ctx.decode([st[0], st[1]])
    assert np.array_equal(hp, want[0]) and np.array_equal(vp, want[1])
    # ... and a ragged band (not a multiple of 4 pixels) is refused in this layout
    with pytest.raises((ValueError, _native.SlgcError), match="multiple of 4"):
        ctx.decode_dev(tiled.ptr, 1, tb, piece, N, 3, 3, mt.at(0), mt.at(px * 2))
    with pytest.raises((ValueError, _native.SlgcError)):
        ctx.tune("stack_tile_log2", 5)
    ctx.tune("stack_tile_log2", 0)


@pytest.mark.parametrize("W,H,N,k", [(1920, 1080, 44, 12), (130, 70, 42, 8)])
def test_ingest_emits_the_tiled_stack(ctx, W, H, N, k):
    """slgc_to_gray_tiled_dev (BGR frames -> tile-interleaved grey stack) == slgc_to_gray_dev followed by slgc_tile_stack_dev, byte for byte."""
    px = W * H
    gray0 = ctx.alloc(N * px)
    ctx.synth_scene_dev(gray0.ptr, px, N, H, W, seed=3, noise=3, shadow=True)
    bgr = ctx.alloc(3 * N * px)
    ctx.synth_bgr_dev(gray0.ptr, px, N, H, W, bgr.ptr, 3 * px)
    gray = ctx.alloc(N * px)
    assert _native.lib().slgc_to_gray_dev(ctx._h, bgr.ptr, N * px, 15, gray.ptr) == 0
    tb = ctx.tiled_stack_bytes(N, px, k)
    t1, t2 = ctx.alloc(tb).zero(), ctx.alloc(tb).zero()
    ctx.tile_stack_dev(gray.ptr, px, N, px, k, t1.ptr)
    ctx.to_gray_tiled_dev(bgr.ptr, N, px, k, t2.ptr)
    ctx.synchronize()
    a = t1.download((tb,), np.uint8)
    assert np.array_equal(a, t2.download((tb,), np.uint8)) and a.any()
    for b in (gray0, bgr, gray, t1, t2):
        b.free()
