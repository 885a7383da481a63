"""Every synthetic capture bench.py times (benchlib/common.py SCENES) at BASELINE.json's sizes, EVERY pixel against the CPU oracle (-m gpu).

SURVEY.md 8(d) names two inputs -- S-uniform (every byte random: the worst case, ~21 % of the pixels decode to arbitrary codes) and the
S-scene; bench.py adds the physically consistent capture on a rig whose projector covers the camera's field of view (> 90 % lit) and a dim,
noisy version of it in which several per cent of the decoded pixels carry wrong codes.  Scattered wrong codes are what puts a flat
triangle into nearly every wave of the fused kernel: its two forms of the float64 redo (compacted over the wave / lane by lane:
slgc_tune "guard_list") must give bit-identical XYZ, and both must match the oracle within BASELINE.json's 1e-4
(reference: decode_codes.py:90-229, triangulate.py:56-61, 84-95)."""
import os

import numpy as np
import pytest

import bench
import oracle_c as oc
import oracle_np as onp
from conftest import ext, has_gpu
from test_gpu_fullsize import compare_scan

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]


@pytest.fixture(scope="module")
def ctx():
    from scanner import _native
    c = _native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", autouse=True)
def oracle_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    oc.set_threads(max(1, min(64, n)))
    yield
    oc.set_threads(1)


@pytest.mark.parametrize("case", ["full", "band", "padded"])
def test_uniform_generator_equals_numpy_twin(ctx, case):
    N, H, W = 18, 96, 260
    row0, rows, pad = {"full": (0, H, 0), "band": (37, 41, 0), "padded": (5, 64, 48)}[case]
    plane = rows * W + pad
    buf = ctx.alloc(N * plane).zero()
    ctx.synth_uniform_dev(buf.ptr, plane, N, H, W, row0=row0, rows=rows, seed=7)
    ctx.synchronize()
    got = buf.download((N, plane), np.uint8)
    ref = onp.synth_uniform(N, H, W, seed=7, row0=row0, rows=rows)
    assert np.array_equal(got[:, :rows * W].reshape(N, rows, W), ref)
    assert not got[:, rows * W:].any()                                                  # the padding between planes is not touched
    whole = onp.synth_uniform(N, H, W, seed=7)
    assert np.array_equal(ref, whole[:, row0:row0 + rows])                               # a band holds the bytes of the same rows of the whole image
    hist = np.bincount(whole.ravel(), minlength=256)
    assert hist.min() > 0.8 * whole.size / 256 and hist.max() < 1.2 * whole.size / 256
    buf.free()


@pytest.mark.parametrize("scene", ["physical", "noisy-physical"])
def test_covering_rig_generator_equals_numpy_twin(ctx, scene):
    """slgc_synth_physical_ex_dev (gains, r2_max) through benchlib's synth_into against oracle_np.synth_physical, on the covering rig."""
    W, H, pw, ph, N = 512, 376, 1920, 1200, 44
    cfg = bench.SCENES[scene]
    calib = list(bench.calibration(4096, 3000, pw, ph, rig=cfg["rig"]))
    calib[0] = calib[0].copy()
    calib[0][:2] /= 8.0                                                                  # the 4096x3000 camera at 1/8 scale: same field of view
    ctx.set_calibration(*calib)
    px = W * H
    stack, hv, truth = ctx.alloc(N * px), ctx.alloc(px * 4), ctx.alloc(px * 12)
    ctx.synth_physical_dev(stack.ptr, px, N, H, W, (pw, ph), seed=3, noise=cfg["noise"], gains=cfg["gains"], r2_max=cfg["r2_max"], d_h_true=hv.at(0),
                           d_v_true=hv.at(px * 2), d_truth_xyz=truth.ptr)
    ctx.synchronize()
    st, h, v, tr = onp.synth_physical(N, H, W, (pw, ph), calib, seed=3, noise=cfg["noise"], gains=cfg["gains"], r2_max=cfg["r2_max"])
    assert np.array_equal(hv.download((H, W), np.int16), h) and np.array_equal(hv.download((H, W), np.int16, px * 2), v)
    assert np.array_equal(stack.download((N, H, W), np.uint8), st)
    got = truth.download((H, W, 3), np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(tr)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(tr.astype(np.float32)))
    assert (h != -1).mean() > 0.9                                                        # the projector covers the camera's field of view
    for b in (stack, hv, truth):
        b.free()


# N = 46 (L = 11 code bits) is the only frame count the reference's own generator emits for the projectors of BASELINE.json
# (generate_codes.py:22-25,53: 4 * ceil(log2(max(w, h))) + 2): codes reach 2047, so the projector clamp of triangulate.py:60-61 fires on every
# pixel decoded right of / below the projector's edge, and the 46-frame kernels are separate compiled specialisations.
SCENE_CASES = [(w, s) for w in ("c3_4096x3000x44", "c2_1920x1080x44", "c1_1280x720x42") for s in ("physical", "noisy-physical", "s-uniform", "s-scene")] + \
              [("c3_4096x3000x46", "s-scene"), ("c2_1920x1080x46", "physical")] + \
              [ext(w, s) for w in ("c3_4096x3000x46", "c2_1920x1080x46") for s in ("physical", "noisy-physical", "s-uniform", "s-scene")
               if (w, s) not in (("c3_4096x3000x46", "s-scene"), ("c2_1920x1080x46", "physical"))]


@pytest.mark.parametrize("workload,scene", SCENE_CASES)
def test_bench_scene_every_pixel(ctx, workload, scene):
    from scanner import _native
    W, H, pw, ph, N = bench.WORKLOADS[workload]
    calib = bench.calibration(W, H, pw, ph, rig=bench.SCENES[scene]["rig"])
    ctx.set_calibration(*calib)
    px = W * H
    stack = ctx.alloc(N * px)
    bench.synth_into(ctx, scene, stack.ptr, px, N, H, W, (pw, ph), 1)                    # bench.py's stacks[0]
    maps, xyz, cnt = ctx.alloc(px * 4), ctx.alloc(px * 12), ctx.alloc(16).zero()
    ctx.synchronize()
    st = stack.download((N, H, W), np.uint8)
    ref_h, ref_v, ref_xyz = oc.scan_dense(st, (pw, ph), *calib)
    got = {}
    try:
        for glist in (1, 0):
            ctx.tune("guard_list", glist)
            maps.zero()
            xyz.zero()
            ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=_native.TRI_ALGEBRAIC)
            ctx.synchronize()
            assert ctx.last_scan_path()["path"] == "fused"
            got[glist] = xyz.download((H, W, 3), np.float32)
            valid, worst = compare_scan(maps.download((H, W), np.int16), maps.download((H, W), np.int16, px * 2), got[glist], ref_h, ref_v, ref_xyz,
                                        f"{workload} {scene} guard_list={glist}")
    finally:
        ctx.tune("guard_list", 1)
    assert np.array_equal(got[0].view(np.uint32), got[1].view(np.uint32)), "the two forms of the flat-triangle redo must agree bit for bit"
    ctx.guard_count_dev(maps.at(0), maps.at(px * 2), H, W, 0, (pw, ph), cnt.ptr)
    ctx.synchronize()
    n_ok, n_flat = (int(x) for x in cnt.download((2,), np.uint64))
    assert n_ok == valid
    frac = valid / px
    print(f"\n{workload} {scene}: {valid} / {px} decodable ({frac:.3f}), {n_flat} on the guarded path ({100.0 * n_flat / max(valid, 1):.3f} % of them), "
          f"worst rel. XYZ error {worst:.2e}")
    if N == 46:
        clamped = int((((ref_h >= pw) | (ref_v >= ph)) & (ref_h != -1) & (ref_v != -1)).sum())
        print(f"  N = 46: {clamped} decodable pixels beyond the projector's edge (clamped, triangulate.py:60-61)")
        if scene == "s-uniform" or (scene == "s-scene" and W > 1920):      # (the S-scene's codes stay inside a 1920x1080 projector at 1920x1080)
            assert clamped > 0.02 * valid
    if scene == "physical" and N != 46:
        assert frac >= 0.8 and n_flat <= 0.001 * n_ok       # the headline capture: most pixels decode, one consistent surface, (almost) no flat triangle
    if scene == "s-uniform":
        assert 0.1 < frac < 0.35 and n_flat > 0
    if scene == "noisy-physical":
        assert frac >= 0.6
    for b in (stack, maps, xyz, cnt):
        b.free()
