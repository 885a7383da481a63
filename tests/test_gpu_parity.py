"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C-ABI of libslgc.so
(via the ctypes mirror) and is compared with the golden vectors produced by the reference itself and
with the CPU oracle on seeded inputs.  Integer/byte/index outputs must be bit-exact; XYZ within
1e-4 relative (BASELINE.json north_star), in practice ~1e-12."""
import os

import numpy as np
import pytest

import oracle_c as oc
import oracle_np as onp
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

XYZ_RTOL = 1e-4  # tolerance stated by BASELINE.json north_star


@pytest.fixture(scope="module")
def ctx():
    from scanner import _native
    c = _native.Context(0)
    yield c
    c.close()


def rot_y(deg):
    th = np.deg2rad(deg)
    return np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])


# ----------------------------------------------------------------------------------------- decode vs golden
def test_golden_decode_cases(ctx, decode_cases):
    for name, c in decode_cases.items():
        st = c["stack"]
        for view in (st, st.astype(np.float64)):
            ld, lg = ctx.direct_indirect(view)
            assert np.array_equal(ld, c["L_d"], equal_nan=True), name
            assert np.array_equal(lg, c["L_g"], equal_nan=True), name
            hc, vc = ctx.codes(view)
            assert hc.dtype == np.int8 and np.array_equal(hc, c["h_codes"]) and np.array_equal(vc, c["v_codes"]), name
            hc2, vc2 = ctx.is_lit(view, c["L_d"], c["L_g"])
            assert np.array_equal(hc2, c["h_codes"]) and np.array_equal(vc2, c["v_codes"]), name
            hp, vp = ctx.decode(view)                      # uint8 -> integer-threshold kernel; float64 -> literal kernel
            assert hp.dtype == np.int64
            assert np.array_equal(hp, c["h_pixels"]) and np.array_equal(vp, c["v_pixels"]), name
        hp, vp = ctx.codes_to_pixels(c["h_codes"], c["v_codes"])
        assert np.array_equal(hp, c["h_pixels"]) and np.array_equal(vp, c["v_pixels"]), name
        if "stack2" in c:
            for cast in (np.uint8, np.float64):
                hp, vp = ctx.decode([st.astype(cast), c["stack2"].astype(cast)])
                assert np.array_equal(hp, c["merged_h_pixels"]) and np.array_equal(vp, c["merged_v_pixels"]), name


def test_module_level_api_matches_golden(decode_cases):
    from scanner.grayCode.decode_codes import decode, get_codes, get_direct_indirect, get_is_lit
    c = decode_cases["scene_N44_24x40"]
    st = c["stack"].astype(np.float64)                      # the reference's own stack dtype
    ld, lg = get_direct_indirect(st)
    hc, vc = get_is_lit(st, ld, lg)
    hc2, vc2 = get_codes(st)
    assert np.array_equal(hc, c["h_codes"]) and np.array_equal(vc2, c["v_codes"]) and np.array_equal(hc, hc2)
    hp, vp = decode(st)
    assert np.array_equal(hp, c["h_pixels"]) and np.array_equal(vp, c["v_pixels"])


def test_rule_known_answers(ctx):
    kat = np.load(os.path.join(GOLDEN, "rule_kat.npz"))["kat"]
    for eps in np.unique(kat[:, 0]):
        rows = kat[kat[:, 0] == eps]
        n = len(rows)
        st = np.zeros((14, 1, n))
        st[2, 0], st[8, 0] = rows[:, 3], rows[:, 4]
        hc, _ = ctx.is_lit(st, rows[:, 1].reshape(1, n), rows[:, 2].reshape(1, n), eps=eps)
        assert np.array_equal(hc[0, 0], rows[:, 5].astype(np.int8)), eps


def test_identity_scan(ctx):
    from scanner.grayCode import generate_codes as gc
    for (w, h) in ((64, 32), (128, 96)):
        seq = gc.get_image_sequence(gc.get_gray_codes(w, h), w, h)
        cap = (20 + 0.7 * seq.astype(np.float64)).astype(np.uint8)
        hp, vp = ctx.decode(cap)
        yy, xx = np.mgrid[0:h, 0:w]
        assert np.array_equal(hp, xx) and np.array_equal(vp, yy)


# ----------------------------------------------------------------------------------------- decode vs oracle, bigger seeded inputs
@pytest.mark.parametrize("N", [14, 17, 26, 42, 44, 46, 62])
def test_decode_random_vs_oracle(ctx, N):
    rng = np.random.default_rng(N)
    for (H, W) in ((67, 131), (120, 256)):                 # ragged (not a multiple of 16) and aligned
        st = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
        st[:, :7, :9] = 0                                   # NaN pixels
        st2 = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
        ref = oc.decode(st)
        for view in (st, st.astype(np.float64)):
            hp, vp = ctx.decode(view)
            assert np.array_equal(hp, ref[0]) and np.array_equal(vp, ref[1])
        ref2 = oc.decode(np.stack([st, st2]))
        hp, vp = ctx.decode([st, st2])
        assert np.array_equal(hp, ref2[0]) and np.array_equal(vp, ref2[1])


@pytest.mark.parametrize("eps", [0, 1, 2, 5, 0.5, 1.0 - 2.0 ** -60, 3.75, -1])
def test_decode_eps_variants_vs_oracle(ctx, eps):
    rng = np.random.default_rng(3)
    st = rng.integers(0, 256, (26, 50, 64), dtype=np.uint8)
    # near-tie pairs so eps matters
    st[8:14] = np.clip(st[2:8].astype(int) + rng.integers(-3, 4, st[2:8].shape), 0, 255).astype(np.uint8)
    ref = oc.decode(st, eps=eps)
    hp, vp = ctx.decode(st, eps=eps)
    assert np.array_equal(hp, ref[0]) and np.array_equal(vp, ref[1])
    ld, lg = oc.direct_indirect(st)
    hc, vc = ctx.is_lit(st, ld, lg, eps=eps)
    rhc, rvc = oc.is_lit(st, ld, lg, eps=eps)
    assert np.array_equal(hc, rhc) and np.array_equal(vc, rvc)


def test_decode_f64_non_integer_stack(ctx):
    rng = np.random.default_rng(5)
    st = rng.uniform(0, 255, (18, 33, 47))
    st[0, :3] = np.nan
    ref = oc.decode(st)
    hp, vp = ctx.decode(st)
    assert np.array_equal(hp, ref[0]) and np.array_equal(vp, ref[1])
    a, b = ctx.direct_indirect(st)
    ra, rb = oc.direct_indirect(st)
    assert np.array_equal(a, ra, equal_nan=True) and np.array_equal(b, rb, equal_nan=True)


def test_float64_stack_of_grey_levels_is_narrowed_on_the_host(ctx, calib):
    """The reference's unchanged caller (src/3-capture_decode.py:66-70) hands over float64 holding uint8 grey levels: the C-ABI
    narrows it to uint8 before the upload (path 1); anything else -- a fraction, NaN, a negative, 256 -- takes the float64 kernel
    (path 2).  Both branches, every host entry point, against the oracle."""
    rng = np.random.default_rng(21)
    N, H, W = 44, 70, 133
    u8, _, _ = onp.synth_scene_int(N, H, W, seed=5)
    u8[:, :3, :9] = 0                                       # black = white = 0 -> NaN pixels survive the narrowing
    f64 = u8.astype(np.float64)
    f64[7, 10, 11] = -0.0 if u8[7, 10, 11] == 0 else f64[7, 10, 11]
    ref = oc.decode(u8)
    for stack, want_path in ((u8, 0), (f64, 1)):
        hp, vp = ctx.decode(stack)
        assert ctx.last_input_path() == want_path
        assert np.array_equal(hp, ref[0]) and np.array_equal(vp, ref[1])
    hc, vc = ctx.codes(f64)
    assert ctx.last_input_path() == 1
    rhc, rvc = oc.get_codes(u8)
    assert np.array_equal(hc, rhc) and np.array_equal(vc, rvc)
    ld, lg = ctx.direct_indirect(f64)
    rld, rlg = oc.direct_indirect(u8)
    assert ctx.last_input_path() == 1 and np.array_equal(ld, rld, equal_nan=True) and np.array_equal(lg, rlg, equal_nan=True)
    # two runs: narrowed together
    u8b, _, _ = onp.synth_scene_int(N, H, W, seed=6, noise=7)
    ref2 = oc.decode(np.stack([u8, u8b]))
    hp, vp = ctx.decode([f64, u8b.astype(np.float64)])
    assert ctx.last_input_path() == 1 and np.array_equal(hp, ref2[0]) and np.array_equal(vp, ref2[1])
    # the first sample that is not a grey level sends the whole call to the float64 kernel -- wherever it sits
    for where, val in (((0, 0, 0), 0.5), ((N - 1, H - 1, W - 1), 255.0000001), ((20, 33, 64), np.nan), ((3, 2, 1), -1.0), ((9, 9, 9), 256.0),
                       ((12, 0, 5), np.inf), ((43, 69, 0), 1e300)):
        bad = f64.copy()
        bad[where] = val
        r = oc.decode(bad)
        hp, vp = ctx.decode(bad)
        assert ctx.last_input_path() == 2, (where, val)
        assert np.array_equal(hp, r[0]) and np.array_equal(vp, r[1]), (where, val)
    bad2 = u8b.astype(np.float64)
    bad2[30, 40, 50] += 0.25                                # only the SECOND run carries the fraction
    r = oc.decode(np.stack([f64, bad2]))
    hp, vp = ctx.decode([f64, bad2])
    assert ctx.last_input_path() == 2 and np.array_equal(hp, r[0]) and np.array_equal(vp, r[1])
    # whole pipeline through the narrowed path
    from scanner.pipeline import scan_to_cloud
    K = calib["cam_mtx"].copy()
    K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 250.0, 250.0
    R, T = rot_y(-20.0), np.array([[0.25], [0.02], [0.04]])
    a = scan_to_cloud(f64, K, calib["cam_dist"], (160, 120), (1920, 1080), calib["proj_mtx"], calib["proj_dist"], R, T, ctx=ctx)
    assert ctx.last_input_path() == 1
    b = scan_to_cloud(u8, K, calib["cam_dist"], (160, 120), (1920, 1080), calib["proj_mtx"], calib["proj_dist"], R, T, ctx=ctx)
    assert np.array_equal(a["pts"], b["pts"]) and np.array_equal(a["h_pixels"], b["h_pixels"]) and a["pts"].shape[1] > 1000
    # an empty stack and float32 / int input still work
    assert ctx.decode(np.zeros((14, 0, 5)))[0].shape == (0, 5)
    hp, vp = ctx.decode(u8.astype(np.float32))
    assert ctx.last_input_path() == 1 and np.array_equal(hp, ref[0])


def test_c99_host_decodes_through_the_c_abi(tmp_path):
    """A C host (no Python between it and libslgc.so) runs slgc_decode on a capture read from a file; the maps must be the oracle's."""
    import subprocess
    from conftest import ROOT
    from scanner import _native
    N, H, W = 26, 24, 40
    st, _, _ = onp.synth_scene_int(N, H, W, seed=8)
    st.tofile(tmp_path / "stack.u8")
    src = tmp_path / "host.c"
    src.write_text(r"""
#include <stdio.h>
#include <stdlib.h>
#include "slgc.h"
int main(int argc, char **argv) {
    int N = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]);
    size_t n = (size_t)N * H * W, npix = (size_t)H * W;
    unsigned char *stack = malloc(n);
    int64_t *h = malloc(npix * 8), *v = malloc(npix * 8);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(stack, 1, n, f) != n) return 2;
    fclose(f);
    slgc_ctx *ctx;
    const void *runs[1];
    runs[0] = stack;
    if (slgc_device_count() < 1) return 3;
    if (slgc_create(0, &ctx)) return 4;
    if (slgc_decode(ctx, runs, SLGC_U8, 1, N, H, W, 1.0, 10.0, h, v)) { fprintf(stderr, "%s\n", slgc_last_error(ctx)); return 5; }
    if (slgc_decode(ctx, runs, SLGC_U8, 1, 10, H, W, 1.0, 10.0, h, v) != SLGC_EINVAL) return 6;     /* N < 14: the reference raises too */
    f = fopen(argv[2], "wb");
    fwrite(h, 8, npix, f); fwrite(v, 8, npix, f);
    fclose(f);
    return slgc_destroy(ctx);
}
""")
    exe = tmp_path / "host"
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lslgc", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([str(exe), str(tmp_path / "stack.u8"), str(tmp_path / "maps.i64"), str(N), str(H), str(W)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "maps.i64", dtype=np.int64).reshape(2, H, W)
    ref = oc.decode(st)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_bad_arguments(ctx):
    with pytest.raises(ValueError, match="diagnostic build"):
        dev_decode(ctx, np.zeros((1, 44, 8, 64), np.uint8), variant=11104128)      # timing-only ablation kernels are not in the shipped library
    with pytest.raises(ValueError):
        ctx.decode(np.zeros((10, 4, 4), np.uint8))          # the reference raises for N < 14 too
    with pytest.raises(ValueError):
        ctx.decode(np.zeros((70, 4, 4), np.uint8))
    with pytest.raises(ValueError):
        ctx.decode(np.zeros((14, 4), np.uint8))
    hp, vp = ctx.decode(np.zeros((14, 0, 5), np.uint8))    # empty image
    assert hp.shape == (0, 5)


# ----------------------------------------------------------------------------------------- device-resident decode
def dev_decode(ctx, st_runs, variant=0, band=None, eps=1):
    R, N, H, W = st_runs.shape
    buf = ctx.alloc(st_runs.nbytes).upload(st_runs)
    r0, r1 = band or (0, H)
    rows = r1 - r0
    out = ctx.alloc(rows * W * 2 * 2 + 64)
    voff = (rows * W * 2 + 31) // 32 * 32
    ctx.decode_dev(buf.at(r0 * W), R, N * H * W, H * W, N, rows, W, out.at(0), out.at(voff), eps=eps, variant=variant)
    ctx.synchronize()
    h = out.download((rows, W), np.int16)
    v = out.download((rows, W), np.int16, voff)
    buf.free()
    out.free()
    return h, v


@pytest.mark.parametrize("variant", [0, 16256, 16128, 16064, 8256, 8128, 8064, 4256, 4128, 4064, 1256,
                                     104064, 104128, 104256, 108064, 108128, 108256, 116128, 1104128, 1108256, 1116256])
def test_decode_dev_variants(ctx, variant):
    rng = np.random.default_rng(9)
    st = rng.integers(0, 256, (2, 44, 96, 256), dtype=np.uint8)
    st[1] = onp.synth_scene(44, 96, 256, seed=3)
    ref = oc.decode(st)
    h, v = dev_decode(ctx, st, variant)
    assert np.array_equal(h, ref[0]) and np.array_equal(v, ref[1])
    # row band of a taller image (multi-GPU sharding): rows 32..80
    rb = oc.decode(st[:, :, 32:80])
    h, v = dev_decode(ctx, st, variant, band=(32, 80))
    assert np.array_equal(h, rb[0]) and np.array_equal(v, rb[1])


def _fuzz_stack(rng, R, N, H, W):
    kind = rng.integers(0, 5)
    if kind == 0:
        return rng.integers(0, 256, (R, N, H, W), dtype=np.uint8)
    if kind == 1:                                               # tiny range: exact ties in (L_max-L_min)*w/(w+b) and in n == i +- eps
        return rng.integers(0, int(rng.integers(2, 12)), (R, N, H, W), dtype=np.uint8)
    if kind == 2:                                               # saturated extremes, black == white == 0 (NaN) included
        return rng.choice(np.array([0, 1, 254, 255], np.uint8), (R, N, H, W))
    if kind == 3:                                               # two levels + noise: thresholds sit next to the data
        lo, hi = sorted(int(x) for x in rng.integers(0, 256, 2))
        st = np.where(rng.integers(0, 2, (R, N, H, W)) == 1, hi, lo) + rng.integers(-2, 3, (R, N, H, W))
        return np.clip(st, 0, 255).astype(np.uint8)
    return np.stack([onp.synth_scene_int(N, H, W, seed=int(rng.integers(1 << 30)), noise=int(rng.integers(0, 12)))[0] for _ in range(R)])


def test_decode_fuzz_vs_oracle(ctx):
    """400 seeded random cases over N (14..62), image shape (ragged widths, single rows), run count, eps and five value
    distributions built to sit on the rule table's boundaries: device-resident kernels and the host API, bit-exact vs the oracle."""
    rng = np.random.default_rng(2024)
    for case in range(400 * int(os.environ.get("SLGC_FUZZ_SCALE", "1"))):      # soak: SLGC_FUZZ_SCALE=50 -> 20 000 cases
        N = int(rng.integers(14, 63))
        H, W = int(rng.integers(1, 24)), int(rng.integers(1, 140))
        R = int(rng.integers(1, 4))
        eps = int(rng.choice([0, 1, 1, 1, 2, 5, 40]))
        st = _fuzz_stack(rng, R, N, H, W)
        ref = oc.decode(st, eps=eps)
        tag = f"case {case}: N={N} H={H} W={W} R={R} eps={eps}"
        if case % 3 == 2:                                       # host API (float64 stack as the reference driver builds it, or uint8)
            got = ctx.decode(st.astype(np.float64) if case % 2 else st, eps=eps)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), tag
        else:
            h, v = dev_decode(ctx, st, 0, eps=eps)
            assert np.array_equal(h, ref[0]) and np.array_equal(v, ref[1]), tag


@pytest.mark.parametrize("eps", [0, 1, 2, 5, 40, 255])
def test_threshold_folding_exhaustive(ctx, eps):
    """All 2^32 (black, white, L_max, L_min) tuples x all 256 grey levels: the integer thresholds the device-resident kernels
    compare against give exactly the literal fp64 predicates of decode_codes.py:172-182 (device evaluation of the same
    expressions k_decode_generic uses, which the golden vectors pin).  2.2e12 comparisons per eps."""
    assert ctx.selftest_thresholds(eps) == 0
    if eps == 1:
        assert ctx.selftest_thresholds(eps | 0x100, 0, 8) > 1_000_000          # negative control: literal side with eps + 1


def test_packed_classification_exhaustive(ctx):
    """Every threshold triple x every (normal, inverse) grey-level pair through the packed-16 rule evaluation of the shipped
    kernel (classify_pk) == the scalar rule table; the other half of each register carries different data."""
    assert ctx.selftest_classify() == 0
    assert ctx.selftest_classify(negative_control=True) > 1_000_000


def _pack_hv24_np(h, v):
    p = (h.astype(np.int64) & 0xfff) | ((v.astype(np.int64) & 0xfff) << 12)
    return np.stack([p & 0xff, (p >> 8) & 0xff, (p >> 16) & 0xff], axis=-1).astype(np.uint8).reshape(-1)


def test_wire_format_hv24_round_trip(ctx):
    """3-byte wire format of the maps (multi-GPU exchange): pack == the bit layout slgc.h states, unpack(pack(x)) == x, for
    aligned, misaligned and ragged sizes, -1 included, 11-bit codes at their maximum."""
    rng = np.random.default_rng(99)
    for npix, misalign in ((4096, 0), (4099, 0), (1, 0), (3, 0), (5, 0), (1000, 2), (1001, 6), (64, 4)):
        h = rng.integers(-1, 2048, npix).astype(np.int16)
        v = rng.integers(-1, 2048, npix).astype(np.int16)
        h[rng.random(npix) < 0.2] = -1
        v[-1] = 2047
        dh, dv = ctx.alloc(npix * 2 + 16), ctx.alloc(npix * 2 + 16)
        dh.upload(h, misalign)
        dv.upload(v, misalign)
        wire = ctx.alloc(npix * 3 + 16).zero()
        woff = misalign // 2 if misalign else 0
        ctx.pack_hv24_dev(dh.at(misalign), dv.at(misalign), npix, 11, wire.at(woff))
        ctx.synchronize()
        assert np.array_equal(wire.download((npix * 3,), np.uint8, woff), _pack_hv24_np(h, v)), (npix, misalign)
        oh, ov = ctx.alloc(npix * 2 + 16).zero(), ctx.alloc(npix * 2 + 16).zero()
        ctx.unpack_hv24_dev(wire.at(woff), npix, oh.at(misalign), ov.at(misalign))
        ctx.synchronize()
        assert np.array_equal(oh.download((npix,), np.int16, misalign), h) and np.array_equal(ov.download((npix,), np.int16, misalign), v)
        for b in (dh, dv, wire, oh, ov):
            b.free()
    buf = ctx.alloc(64)
    with pytest.raises(ValueError):
        ctx.pack_hv24_dev(buf.ptr, buf.ptr, 4, 12, buf.ptr)            # 12-bit codes do not fit


@pytest.mark.parametrize("H,W", [(64, 256), (270, 256), (7, 33)])
def test_triangulate_wire_equals_unpack_then_triangulate(ctx, calib, H, W):
    """The triangulation kernel fed with wire-format maps (unpacks in its map load, also writes the int16 maps) == the separate
    unpack kernel followed by the ordinary triangulation, bit for bit; (7, 33) takes the unpack-first fallback (ragged size)."""
    N = 26
    st, _, _ = onp.synth_scene_int(N, H, W, seed=H + W)
    K = calib["cam_mtx"].copy()
    K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 300.0, 300.0
    psize = (60, 50)
    pk = onp.scale_proj_mtx(calib["proj_mtx"], psize, (1920, 1080))
    ctx.set_calibration(K, calib["cam_dist"], pk, calib["proj_dist"], rot_y(-20.0), np.array([[0.25], [0.02], [0.04]]))
    hp, vp = oc.decode(st)
    npix = H * W
    dh, dv = ctx.alloc(npix * 2).upload(hp.astype(np.int16)), ctx.alloc(npix * 2).upload(vp.astype(np.int16))
    wire = ctx.alloc(npix * 3 + 16)
    ctx.pack_hv24_dev(dh.ptr, dv.ptr, npix, 6, wire.ptr)
    x_ref, x_wire, cnt = ctx.alloc(npix * 12), ctx.alloc(npix * 12), ctx.alloc(8)
    oh, ov = ctx.alloc(npix * 2).zero(), ctx.alloc(npix * 2).zero()
    for mode in (0, 1):
        ctx.triangulate_maps_dev(dh.ptr, dv.ptr, H, W, 0, psize, x_ref.ptr, None, mode=mode)
        cnt.zero(); oh.zero(); ov.zero()
        ctx.triangulate_wire_dev(wire.ptr, H, W, 0, psize, oh.ptr, ov.ptr, x_wire.ptr, cnt.ptr, mode=mode)
        ctx.synchronize()
        assert np.array_equal(oh.download((H, W), np.int16), hp) and np.array_equal(ov.download((H, W), np.int16), vp)
        assert np.array_equal(x_wire.download((npix, 3), np.float32), x_ref.download((npix, 3), np.float32), equal_nan=True)
        assert int(cnt.download((1,), np.uint64)[0]) == int(((hp != -1) & (vp != -1)).sum())
    for b in (dh, dv, wire, x_ref, x_wire, cnt, oh, ov):
        b.free()


def test_decode_dev_misaligned_band_falls_back_to_narrow_loads(ctx):
    rng = np.random.default_rng(10)
    st = rng.integers(0, 256, (1, 42, 37, 101), dtype=np.uint8)       # W odd: bands start at odd byte offsets
    ref = oc.decode(st[:, :, 5:30])
    h, v = dev_decode(ctx, st, 0, band=(5, 30))
    assert np.array_equal(h, ref[0]) and np.array_equal(v, ref[1])
    with pytest.raises(ValueError):
        dev_decode(ctx, st, 16256, band=(5, 30))


def test_synth_matches_numpy_twin(ctx):
    for (N, H, W, noise, shadow) in ((44, 64, 200, 3, True), (46, 37, 101, 0, False), (14, 16, 40, 9, True)):
        buf = ctx.alloc(N * H * W)
        ctx.synth_scene_dev(buf.ptr, H * W, N, H, W, seed=7, noise=noise, shadow=shadow)
        ctx.synchronize()
        got = buf.download((N, H, W), np.uint8)
        ref, _, _ = onp.synth_scene_int(N, H, W, seed=7, noise=noise, shadow=shadow)
        assert np.array_equal(got, ref)
        buf.free()


@pytest.mark.parametrize("N,W,H", [(44, 4096, 3000), (46, 1920, 1080)])
def test_full_size_round_trip_properties(ctx, N, W, H):
    """BASELINE.json full sizes: encode (device generator) -> decode must return the encoded projector
    coordinates wherever a pixel is decodable; shadowed pixels must be undecodable; and a band of the result
    must equal the CPU oracle bit for bit."""
    L = (N - 2) // 4
    stack = ctx.alloc(N * H * W)
    ctx.synth_scene_dev(stack.ptr, H * W, N, H, W, seed=1, noise=3, shadow=True)
    out = ctx.alloc(H * W * 4)
    ctx.decode_dev(stack.ptr, 1, N * H * W, H * W, N, H, W, out.at(0), out.at(H * W * 2))
    ctx.synchronize()
    h = out.download((H, W), np.int16).astype(np.int64)
    v = out.download((H, W), np.int16, H * W * 2).astype(np.int64)
    yy, xx = np.mgrid[0:H, 0:W]
    msk = (1 << L) - 1
    tri = lambda t: np.abs(((t >> 5) % 10) - 5)           # noqa: E731
    xs, ys = (((29 * xx) >> 5) + tri(yy)) & msk, (((29 * yy) >> 5) + tri(xx)) & msk
    okh, okv = h != -1, v != -1
    assert okh.mean() > 0.5 and okv.mean() > 0.5
    assert np.array_equal(h[okh], xs[okh]) and np.array_equal(v[okv], ys[okv])
    sy0, sy1, sx0, sx1 = int(0.30 * H), int(0.30 * H + 0.387 * H), int(0.55 * W), int(0.55 * W + 0.387 * W)
    assert not okh[sy0:sy1, sx0:sx1].any() and not okv[sy0:sy1, sx0:sx1].any()
    assert h.min() >= -1 and h.max() <= msk
    # oracle on a band that crosses the shadow edge
    r0, r1 = sy0 - 8, sy0 + 8
    band = stack.download((N, H, W), np.uint8)[:, r0:r1]
    ref = oc.decode(band)
    assert np.array_equal(h[r0:r1], ref[0]) and np.array_equal(v[r0:r1], ref[1])
    # idempotence: a second launch gives the same bytes
    ctx.decode_dev(stack.ptr, 1, N * H * W, H * W, N, H, W, out.at(0), out.at(H * W * 2))
    ctx.synchronize()
    assert np.array_equal(out.download((H, W), np.int16), h.astype(np.int16))
    stack.free()
    out.free()


# ----------------------------------------------------------------------------------------- correspondences / triangulation
def test_golden_triangulate_cases(ctx, tri_cases):
    from scanner.triangulation import Triangulate
    for name, c in tri_cases.items():
        pm = c["proj_mtx"].copy()
        t = Triangulate(c["h"], c["v"], tuple(int(x) for x in c["cam_size"]), c["cam_mtx"], c["cam_dist"],
                        tuple(int(x) for x in c["proj_size"]), tuple(int(x) for x in c["proj_calib_size"]), pm,
                        c["proj_dist"], c["R"], c["T"], None, ctx=ctx)
        assert np.array_equal(pm, c["proj_mtx_scaled"])                        # in-place scaling, like the reference
        cam, proj, col = t.get_cam_proj_pts(c["white"])
        assert cam.dtype == np.float32 and proj.dtype == np.float32 and col.dtype == np.float64
        assert np.array_equal(cam, c["cam_pts"]) and np.array_equal(proj, c["proj_pts"]) and np.array_equal(col, c["colors"])
        pts = t.triangulate(cam, proj)
        assert pts.dtype == np.float64 and pts.shape == c["pts"].shape
        np.testing.assert_allclose(pts, c["pts"], rtol=XYZ_RTOL, atol=0)
        np.testing.assert_allclose(pts, c["pts"], rtol=1e-9, atol=1e-12)       # what the kernel actually achieves
        alg = t.triangulate(cam, proj, exact=False)
        np.testing.assert_allclose(alg, c["pts"], rtol=XYZ_RTOL, atol=0)
        fp, fc = t.filter_3d_pts(c["pts"], c["colors"], threshold=0.5)
        assert np.array_equal(fp, c["filt_pts"]) and np.array_equal(fc, c["filt_colors"])
        cam_r, proj_r, col_r = t.get_cam_proj_pts(c["white"], order="row")
        ref_r = oc.cam_proj_pts(c["h"], c["v"], c["cam_size"], c["proj_size"], c["white"], order="row")
        assert np.array_equal(cam_r, ref_r[0]) and np.array_equal(proj_r, ref_r[1]) and np.array_equal(col_r, ref_r[2])
        cam_n, proj_n, col_n = t.get_cam_proj_pts(None)
        assert col_n is None and np.array_equal(cam_n, c["cam_pts"])
        # no decodable pixel at all: the reference's np.array([], dtype=float32) has shape (0,) (triangulate.py:66-69), and triangulate() of it is empty
        t0 = Triangulate(np.full_like(c["h"], -1), c["v"], tuple(int(x) for x in c["cam_size"]), c["cam_mtx"], c["cam_dist"],
                         tuple(int(x) for x in c["proj_size"]), tuple(int(x) for x in c["proj_size"]), c["proj_mtx"].copy(), c["proj_dist"], c["R"], c["T"], None, ctx=ctx)
        e_cam, e_proj, e_col = t0.get_cam_proj_pts(c["white"])
        assert e_cam.shape == (0,) and e_cam.dtype == np.float32 and e_proj.shape == (0,) and e_col.shape == (0,) and e_col.dtype == np.float64
        assert t0.triangulate(e_cam, e_proj).shape == (3, 0)
        # ... and the box filter of that empty scan returns the colours in the empty shape it was given (triangulate.py:119-121), through the
        # method and through the fused Triangulation.compute()
        f_pts, f_col = t0.filter_3d_pts(t0.triangulate(e_cam, e_proj), e_col, 0.5)
        assert f_pts.shape == (3, 0) and f_col.shape == (0,)
        from scanner.triangulation.triangulate import Triangulation
        tc = Triangulation(np.full_like(c["h"], -1), c["v"], tuple(int(x) for x in c["cam_size"]), c["cam_mtx"], c["cam_dist"],
                           tuple(int(x) for x in c["proj_size"]), tuple(int(x) for x in c["proj_size"]), c["proj_mtx"].copy(), c["proj_dist"], c["R"], c["T"], None, ctx=ctx)
        c_pts, c_col = tc.compute(c["white"], threshold=0.5)
        assert c_pts.shape == (3, 0) and c_col.shape == (0,)
        assert tc.compute(None, threshold=0.5)[0].shape == (3, 0)


def test_cam_proj_pts_large_vs_oracle(ctx):
    rng = np.random.default_rng(12)
    H, W = 301, 517
    h = rng.integers(-1, 1500, (H, W)).astype(np.int64)
    v = rng.integers(-1, 900, (H, W)).astype(np.int64)
    h[rng.random((H, W)) < 0.3] = -1
    h[:, 100:140] = -1                                      # empty columns
    v[50:90] = -1                                           # empty rows
    white = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for order, o in (("x", 0), ("row", 1)):
        ref = oc.cam_proj_pts(h, v, (W, H), (1280, 800), white, order=order)
        got = ctx.cam_proj_pts(h, v, (W, H), (1280, 800), white, order=o)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
    none = ctx.cam_proj_pts(np.full((H, W), -1), v, (W, H), (1280, 800), white)
    assert none[0].shape == (0, 2) and none[2].shape == (0, 3)
    full = ctx.cam_proj_pts(np.zeros((H, W), np.int64), np.zeros((H, W), np.int64), (W, H), (1280, 800), None)
    assert full[0].shape == (H * W, 2) and np.array_equal(full[0][:3], [[0, 0], [0, 1], [0, 2]])   # x-major


@pytest.mark.parametrize("W,H", [(517, 301), (64, 32), (65, 33), (3, 2), (1, 1), (200, 1)])
def test_cloud_lists_dev_ragged_vs_oracle(ctx, W, H):
    """slgc_cloud_lists_dev on maps / XYZ / white images of its own making: widths that are not a multiple of 4 (white image read byte by byte),
    tiles cut by both edges, empty rows and columns, every byte value in the colours (b / 255.0 must be the reference's division)."""
    rng = np.random.default_rng(W * 1000 + H)
    px = W * H
    h = rng.integers(-1, 1500, (H, W)).astype(np.int16)
    v = rng.integers(-1, 900, (H, W)).astype(np.int16)
    h[rng.random((H, W)) < 0.3] = -1
    if W > 140:
        h[:, 100:140] = -1
    if H > 90:
        v[50:90] = -1
    dense = rng.standard_normal((H, W, 3)).astype(np.float32)
    white_h = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    white_h.reshape(-1)[:min(256, px * 3)] = np.arange(min(256, px * 3), dtype=np.uint8)
    maps = ctx.alloc(max(16, px * 4))
    maps.upload(np.concatenate([h.reshape(-1), v.reshape(-1)]))
    xyz = ctx.alloc(max(16, px * 12)).upload(dense)
    raw = ctx.alloc(px * 3 + 16)
    for shift in (0, 1):                                     # white image on a dword boundary and one byte off it
        raw.upload(np.concatenate([np.zeros(shift, np.uint8), white_h.reshape(-1)]))
        lists = ctx.alloc_cloud_lists(px, colors=True)
        ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), xyz.ptr, raw.at(shift), W, H, (1280, 800), lists)
        cam, proj, pts, col = lists.download()
        rcam, rproj, rcol = oc.cam_proj_pts(h.astype(np.int64), v.astype(np.int64), (W, H), (1280, 800), white_h, order="x")
        assert np.array_equal(cam, rcam) and np.array_equal(proj, rproj) and np.array_equal(col, rcol)
        xs, ys = rcam[:, 0].astype(np.int64), rcam[:, 1].astype(np.int64)
        assert pts.shape == (3, len(rcam)) and np.array_equal(pts, dense[ys, xs].astype(np.float64).T)
        lists.free()
    for b in (maps, xyz, raw):
        b.free()


def test_triangulate_random_vs_oracle(ctx, calib):
    rng = np.random.default_rng(13)
    M = 200_003
    cam = np.stack([rng.integers(0, 1920, M), rng.integers(0, 1080, M)], 1).astype(np.float32)
    proj = np.stack([rng.integers(0, 1920, M), rng.integers(0, 1080, M)], 1).astype(np.float32)
    pk = onp.scale_proj_mtx(calib["proj_mtx"], (1920, 1080), (1920, 1080))
    R, T = rot_y(-20.0), np.array([[0.25], [0.02], [0.04]])
    ctx.set_calibration(calib["cam_mtx"], calib["cam_dist"], pk, calib["proj_dist"], R, T)
    ref = oc.triangulate(cam, proj, calib["cam_mtx"], calib["cam_dist"], pk, calib["proj_dist"], R, T)
    got = ctx.triangulate(cam, proj)
    fin = np.isfinite(ref).all(axis=0)
    assert fin.mean() > 0.99 and np.array_equal(np.isfinite(got).all(axis=0), fin)
    np.testing.assert_allclose(got[:, fin], ref[:, fin], rtol=XYZ_RTOL, atol=0)
    rel = np.abs(got[:, fin] - ref[:, fin]) / np.maximum(np.abs(ref[:, fin]), 1e-12)
    assert np.median(rel) < 1e-12
    # the sqrt form agrees except where rays are nearly parallel to the baseline (ill-conditioned in both forms)
    alg = ctx.triangulate(cam, proj, mode=1)
    rel = np.abs(alg[:, fin] - ref[:, fin]) / np.maximum(np.abs(ref[:, fin]), 1e-12)
    assert np.quantile(rel, 0.999) < XYZ_RTOL
    # filter
    col = rng.random((M, 3))
    fp, fc = ctx.filter_3d_pts(ref, col, 0.5)
    rp, rc = oc.filter_3d_pts(ref, col, 0.5)
    assert np.array_equal(fp, rp) and np.array_equal(fc, rc)
    e = ctx.filter_3d_pts(np.zeros((3, 0)), np.zeros((0, 3)), 0.5)
    assert e[0].shape == (3, 0)


def test_scan_dev_vs_oracle(ctx, calib):
    N, H, W = 44, 96, 256
    st, _, _ = onp.synth_scene_int(N, H, W, seed=2)
    K = calib["cam_mtx"].copy()
    K[0, 2], K[1, 2] = W / 2, H / 2
    K[0, 0] = K[1, 1] = 300.0
    psize = (300, 200)
    pk = onp.scale_proj_mtx(calib["proj_mtx"], psize, (1920, 1080))
    R, T = rot_y(-20.0), np.array([[0.25], [0.02], [0.04]])
    ctx.set_calibration(K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
    hp, vp, ref = oc.scan_dense(st, psize, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
    stack = ctx.alloc(st.nbytes).upload(st)
    xyz = ctx.alloc(H * W * 12)
    cnt = ctx.alloc(8).zero()
    maps = ctx.alloc(H * W * 4)
    for mode in (0, 1):
        cnt.zero()
        ctx.scan_dev(stack.ptr, 1, st.nbytes, H * W, N, H, W, 0, psize, xyz.ptr, cnt.ptr, maps.at(0), maps.at(H * W * 2), mode=mode)
        ctx.synchronize()
        got = xyz.download((H, W, 3), np.float32)
        ok = (hp != -1) & (vp != -1)
        assert int(cnt.download((1,), np.uint64)[0]) == ok.sum() > 1000
        assert np.array_equal(maps.download((H, W), np.int16), hp) and np.array_equal(maps.download((H, W), np.int16, H * W * 2), vp)
        assert np.isnan(got[~ok]).all() and np.isfinite(got[ok]).all()
        np.testing.assert_allclose(got[ok], np.moveaxis(ref, 0, -1)[ok], rtol=XYZ_RTOL, atol=0)
    # fused single-kernel path (no count requested, algebraic): same maps, same cloud
    maps.zero()
    ctx.scan_dev(stack.ptr, 1, st.nbytes, H * W, N, H, W, 0, psize, xyz.ptr, None, maps.at(0), maps.at(H * W * 2), mode=1)
    ctx.synchronize()
    got = xyz.download((H, W, 3), np.float32)
    assert np.array_equal(maps.download((H, W), np.int16), hp) and np.array_equal(maps.download((H, W), np.int16, H * W * 2), vp)
    assert np.isnan(got[~ok]).all() and np.isfinite(got[ok]).all()
    np.testing.assert_allclose(got[ok], np.moveaxis(ref, 0, -1)[ok], rtol=XYZ_RTOL, atol=0)
    # fused, two runs, band rows 16..80 without caller-provided maps
    st2 = np.stack([st, onp.synth_scene_int(N, H, W, seed=9, noise=6)[0]])
    hp2, vp2, ref2 = oc.scan_dense(st2[:, :, 16:80], psize, np.array([[K[0, 0], 0, K[0, 2]], [0, K[1, 1], K[1, 2] - 16], [0, 0, 1]]),
                                   calib["cam_dist"], pk, calib["proj_dist"], R, T)
    stack2 = ctx.alloc(st2.nbytes).upload(st2)
    ctx.scan_dev(stack2.at(16 * W), 2, st.nbytes, H * W, N, 64, W, 16, psize, xyz.ptr, None, mode=1)
    ctx.synchronize()
    got2 = xyz.download((64, W, 3), np.float32)
    ok2 = (hp2 != -1) & (vp2 != -1)
    assert np.array_equal(np.isfinite(got2[..., 0]), ok2)
    np.testing.assert_allclose(got2[ok2], np.moveaxis(ref2, 0, -1)[ok2], rtol=XYZ_RTOL, atol=0)
    stack2.free()
    # band with row0 (multi-GPU shard): rows 32..64
    cnt.zero()
    ctx.scan_dev(stack.at(32 * W), 1, st.nbytes, H * W, N, 32, W, 32, psize, xyz.ptr, cnt.ptr, mode=0)
    ctx.synchronize()
    got = xyz.download((32, W, 3), np.float32)
    okb = ok[32:64]
    np.testing.assert_allclose(got[okb], np.moveaxis(ref, 0, -1)[32:64][okb], rtol=XYZ_RTOL, atol=0)
    # row-major compaction of the dense band, keyed by linear pixel index
    pts = ctx.alloc(32 * W * 12)
    keys = ctx.alloc(32 * W * 4)
    ctx.compact_dev(xyz.ptr, 32, W, 32, pts.ptr, keys.ptr, cnt.ptr)
    ctx.synchronize()
    m = int(cnt.download((1,), np.uint64)[0])
    assert m == okb.sum()
    k = keys.download((m,), np.uint32)
    yy, xx = np.nonzero(okb)
    assert np.array_equal(k, (yy + 32) * W + xx)
    assert np.array_equal(pts.download((m, 3), np.float32), got[okb])
    for b in (stack, xyz, cnt, maps, pts, keys):
        b.free()


def test_rccl_single_rank(ctx):
    """nranks = 1 exercises library loading, communicator creation and every collective wrapper."""
    from scanner import _native
    uid = _native.Context.comm_unique_id()
    ctx.comm_init(0, 1, uid)
    try:
        ctx.comm_barrier()
        assert ctx.comm_allreduce_max(3.5) == 3.5
        assert ctx.comm_allgather_i64(42) == [42]
        src = ctx.alloc(1000).upload(np.arange(1000, dtype=np.uint8))
        dst = ctx.alloc(1000).zero()
        ctx.comm_allgatherv(src.ptr, dst.ptr, [1000], [0])                     # equal shards back to back: ncclAllGather
        ctx.synchronize()
        assert np.array_equal(dst.download((1000,), np.uint8), np.arange(1000, dtype=np.uint8))
        dst2 = ctx.alloc(1200).zero()
        ctx.comm_allgatherv(src.ptr, dst2.ptr, [1000], [64])                   # any other layout: grouped ncclBroadcast
        ctx.synchronize()
        got = dst2.download((1200,), np.uint8)
        assert np.array_equal(got[64:1064], np.arange(1000, dtype=np.uint8)) and not got[:64].any() and not got[1064:].any()
        ctx.comm_allgatherv(dst.ptr, dst.ptr, [1000], [0])                     # in place
        ctx.synchronize()
        assert np.array_equal(dst.download((1000,), np.uint8), np.arange(1000, dtype=np.uint8))
    finally:
        ctx.comm_destroy()


def test_sharded_scanner_single_rank_rccl(ctx, calib):
    """ShardedScanner + RcclExchange end to end on one GPU (nranks = 1): band scan, compaction, counts all-gather and
    the all-gatherv must reproduce the oracle's cloud, and the keys must give back the reference's x-major lists."""
    from scanner import _native, sharded
    N, H, W = 26, 40, 128
    st, _, _ = onp.synth_scene_int(N, H, W, seed=6)
    K = calib["cam_mtx"].copy()
    K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 250.0, 250.0
    psize = (160, 120)
    pk = onp.scale_proj_mtx(calib["proj_mtx"], psize, (1920, 1080))
    R, T = rot_y(-20.0), np.array([[0.25], [0.02], [0.04]])
    ctx.set_calibration(K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
    hp, vp, ref = oc.scan_dense(st, psize, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
    ctx.comm_init(0, 1, _native.Context.comm_unique_id())
    try:
        stack = ctx.alloc(st.nbytes).upload(st)
        scm = sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), sharded.ShardPlan(H, W, 1), psize, N, mode=_native.TRI_EXACT)
        assert scm.scan(stack.ptr, H * W) is None                                           # default: map-band exchange
        gh, gv, gx = scm.fetch_dense()
        okm = (hp != -1) & (vp != -1)
        assert np.array_equal(gh, hp) and np.array_equal(gv, vp) and np.array_equal(np.isfinite(gx[..., 0]), okm)
        np.testing.assert_allclose(gx[okm], np.moveaxis(ref, 0, -1)[okm], rtol=XYZ_RTOL, atol=0)
        # pipelined form: three different captures in flight, results must come out in order
        caps = [onp.synth_scene_int(N, H, W, seed=60 + j, noise=3 + j)[0] for j in range(3)]
        bufs = [ctx.alloc(cp.nbytes).upload(cp) for cp in caps]
        outs = []
        for j, b in enumerate(bufs):
            scm.submit(b.ptr, H * W)
            if j:
                outs.append(scm.fetch_dense())                       # scan j-1 is complete once submit(j) returned its stream work
        scm.flush()
        outs.append(scm.fetch_dense())
        for cp, (gh2, gv2, gx2) in zip(caps, outs):
            rh2, rv2, rx2 = oc.scan_dense(cp, psize, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
            ok2 = (rh2 != -1) & (rv2 != -1)
            assert np.array_equal(gh2, rh2) and np.array_equal(gv2, rv2) and np.array_equal(np.isfinite(gx2[..., 0]), ok2)
            np.testing.assert_allclose(gx2[ok2], np.moveaxis(rx2, 0, -1)[ok2], rtol=XYZ_RTOL, atol=0)
        # the 3-byte wire format forced on (auto keeps int16 for a single rank): same products
        scw = sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), sharded.ShardPlan(H, W, 1), psize, N, mode=_native.TRI_EXACT, wire="hv24")
        assert scm.wire == "int16" and scw.wire == "hv24"
        scw.scan(stack.ptr, H * W)
        wh, wv, wx = scw.fetch_dense()
        assert np.array_equal(wh, hp) and np.array_equal(wv, vp) and np.array_equal(wx, gx, equal_nan=True)
        scw.submit(stack.ptr, H * W)
        scw.submit(bufs[0].ptr, H * W)
        scw.flush()
        wh, wv, wx = scw.fetch_dense()
        rh0, rv0, _ = oc.scan_dense(caps[0], psize, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
        assert np.array_equal(wh, rh0) and np.array_equal(wv, rv0)
        # 41 scans back to back without touching the host in between (double-buffered maps / wire buffers, two streams, events):
        # the last result must be the last input's
        for sc_ in (scm, scw):
            for j in range(41):
                sc_.submit(bufs[j % 3].ptr, H * W)
            sc_.flush()
            lh, lv, lx = sc_.fetch_dense()
            rhl, rvl, rxl = oc.scan_dense(caps[40 % 3], psize, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
            okl = (rhl != -1) & (rvl != -1)
            assert np.array_equal(lh, rhl) and np.array_equal(lv, rvl) and np.array_equal(np.isfinite(lx[..., 0]), okl)
            np.testing.assert_allclose(lx[okl], np.moveaxis(rxl, 0, -1)[okl], rtol=XYZ_RTOL, atol=0)
        with pytest.raises(ValueError):
            sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), sharded.ShardPlan(H, W, 1), psize, 62, wire="hv24")      # L = 15 does not fit
        # the same scan through the one-call C entry point (slgc_scan_sharded_dev)
        for wire_knob in (0, 1):                       # int16 maps (default) / the 3-byte wire format
            ctx.tune("wire", wire_knob)
            dh, dv, dx = ctx.alloc(H * W * 2).zero(), ctx.alloc(H * W * 2).zero(), ctx.alloc(H * W * 12).zero()
            ctx.scan_sharded_dev(stack.ptr, 1, st.nbytes, H * W, N, H, W, psize, dh.ptr, dv.ptr, dx.ptr, mode=_native.TRI_EXACT)
            ctx.synchronize()
            assert np.array_equal(dh.download((H, W), np.int16), hp) and np.array_equal(dv.download((H, W), np.int16), vp)
            cx = dx.download((H, W, 3), np.float32)
            assert np.array_equal(np.isfinite(cx[..., 0]), okm)
            np.testing.assert_allclose(cx[okm], np.moveaxis(ref, 0, -1)[okm], rtol=XYZ_RTOL, atol=0)
        ctx.tune("wire", 0)
        sc = sharded.ShardedScanner(ctx, sharded.RcclExchange(ctx), sharded.ShardPlan(H, W, 1), psize, N, mode=_native.TRI_EXACT,
                                    exchange_kind="records")
        total = sc.scan(stack.ptr, H * W)
        rec = sc.fetch(total)
        ok = (hp != -1) & (vp != -1)
        assert total == ok.sum() and np.array_equal(rec["key"], np.nonzero(ok.ravel())[0])
        np.testing.assert_allclose(rec["xyz"], np.moveaxis(ref, 0, -1)[ok], rtol=XYZ_RTOL, atol=0)
        cam, P = sharded.to_reference_lists(rec, W, H)
        rcam, _, _ = oc.cam_proj_pts(hp, vp, (W, H), psize, None, order="x")
        assert np.array_equal(cam, rcam) and P.shape == (3, total)
    finally:
        ctx.comm_destroy()


# ----------------------------------------------------------------------------------------- robustness
def test_call_order_and_error_paths(calib):
    from scanner import _native
    c = _native.Context(0)
    try:
        with pytest.raises(_native.SlgcError, match="set_calibration"):
            c.triangulate(np.zeros((4, 2), np.float32), np.zeros((4, 2), np.float32))
        buf = c.alloc(64)
        with pytest.raises(_native.SlgcError, match="set_calibration"):
            c.scan_dev(buf.ptr, 1, 64, 4, 14, 1, 4, 0, (4, 4), buf.ptr)
        with pytest.raises(ValueError):
            c.decode_dev(buf.ptr, 1, 64, 4, 14, 1, 4, buf.ptr, buf.ptr, eps=0.5)      # device path needs an integer eps
        with pytest.raises(ValueError):
            c.decode_dev(buf.ptr, 1, 64, 2, 14, 1, 4, buf.ptr, buf.ptr)               # plane_stride smaller than the band
        with pytest.raises(ValueError):
            c.set_calibration(np.eye(3), np.zeros(15), np.eye(3), None, np.eye(3), np.zeros(3))
        with pytest.raises(ValueError):
            c.set_calibration(np.eye(3), [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.1, 0], np.eye(3), None, np.eye(3), np.zeros(3))   # tilt
        with pytest.raises(_native.SlgcError):
            c.comm_barrier()                                                           # no communicator
        with pytest.raises(ValueError):
            c.decode([np.zeros((14, 4, 4), np.uint8)] * 9)                              # more runs than SLGC_MAX_RUNS
        with pytest.raises(ValueError):
            c.decode([np.zeros((14, 4, 4), np.uint8), np.zeros((14, 4, 5), np.uint8)])
    finally:
        c.close()


def test_two_contexts_in_two_threads(decode_cases):
    """Contexts are independent (own stream + workspace); ctypes releases the GIL, so two Python threads overlap."""
    import threading
    from scanner import _native
    rng = np.random.default_rng(21)
    stacks = [rng.integers(0, 256, (44, 200, 320), dtype=np.uint8) for _ in range(2)]
    refs = [oc.decode(s) for s in stacks]
    errs = []

    def worker(i):
        try:
            c = _native.Context(0)
            for _ in range(5):
                hp, vp = c.decode(stacks[i])
                assert np.array_equal(hp, refs[i][0]) and np.array_equal(vp, refs[i][1])
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("N,eps", [(65, 1), (14, 0), (30, 3), (58, 255)])
def test_decode_extreme_frame_counts_and_eps(ctx, N, eps):
    rng = np.random.default_rng(N + eps)
    st = rng.integers(0, 256, (2, N, 41, 52), dtype=np.uint8)
    st[1, :, 10:30] = np.clip(st[0, :, 10:30].astype(int) + rng.integers(-2, 3, st[0, :, 10:30].shape), 0, 255)
    ref = oc.decode(st, eps=eps)
    hp, vp = ctx.decode(list(st), eps=eps)
    assert np.array_equal(hp, ref[0]) and np.array_equal(vp, ref[1])
    h, v = dev_decode(ctx, st, 0, eps=eps)
    assert np.array_equal(h, ref[0]) and np.array_equal(v, ref[1])
    assert hp.max() <= (1 << int((N - 2) / 4)) - 1


@pytest.mark.parametrize("H,W", [(270, 256), (300, 1000), (64, 1024)])
def test_triangulate_maps_xcd_remap_sizes(ctx, calib, H, W):
    """Dense triangulation with >= 64 workgroups takes the XCD-aware workgroup -> tile map (slgc_internal.h: xcd_block); sizes
    whose workgroup count is / is not a multiple of 8, last workgroup partial.  Every pixel must still land in its own slot."""
    N = 26
    st, _, _ = onp.synth_scene_int(N, H, W, seed=H)
    K = calib["cam_mtx"].copy()
    K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 400.0, 400.0
    psize = (200, 150)
    pk = onp.scale_proj_mtx(calib["proj_mtx"], psize, (1920, 1080))
    R, T = rot_y(-20.0), np.array([[0.25], [0.02], [0.04]])
    ctx.set_calibration(K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
    hp, vp, ref = oc.scan_dense(st, psize, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
    ok = (hp != -1) & (vp != -1)
    dh, dv = ctx.alloc(H * W * 2).upload(hp.astype(np.int16)), ctx.alloc(H * W * 2).upload(vp.astype(np.int16))
    xyz, cnt = ctx.alloc(H * W * 12), ctx.alloc(8)
    for mode in (0, 1):
        cnt.zero()
        xyz.zero()
        ctx.triangulate_maps_dev(dh.ptr, dv.ptr, H, W, 0, psize, xyz.ptr, cnt.ptr, mode=mode)
        ctx.synchronize()
        got = xyz.download((H, W, 3), np.float32)
        assert int(cnt.download((1,), np.uint64)[0]) == ok.sum()
        assert np.array_equal(np.isfinite(got[..., 0]), ok) and np.isnan(got[~ok]).all()
        want = np.moveaxis(ref, 0, -1)       # this geometry has a few near-parallel ray pairs (ranges of 1e3..1e4 baselines): guarded form
        np.testing.assert_allclose(got[ok], want[ok], rtol=XYZ_RTOL, atol=0)
    for b in (dh, dv, xyz, cnt):
        b.free()


def test_scan_dev_fuzz_vs_oracle(ctx, calib):
    """40 seeded random scans: N, band shape (ragged and 4-aligned), run count, projector size and calibration drawn at
    random; maps bit-exact, validity identical, XYZ within 1e-4 (exact mode everywhere; the cancelled form away from
    near-parallel rays), for the fused kernel and for the two-kernel path."""
    rng = np.random.default_rng(77)
    worst = [0.0]
    for case in range(40 * int(os.environ.get("SLGC_FUZZ_SCALE", "1"))):
        N = int(rng.choice([14, 18, 26, 30, 42, 44, 46]))
        L = (N - 2) // 4
        H, W = int(rng.integers(2, 40)), int(rng.integers(8, 200))
        if case % 2:
            W = (W + 3) // 4 * 4                                 # fused-kernel eligible
        R_ = int(rng.integers(1, 3))
        st = np.stack([onp.synth_scene_int(N, H, W, seed=int(rng.integers(1 << 30)), noise=int(rng.integers(0, 8)))[0] for _ in range(R_)])
        psize = (int(rng.integers(4, (2 << L) + 8)), int(rng.integers(4, (2 << L) + 8)))      # smaller and larger than the code range (clamp)
        K = calib["cam_mtx"].copy()
        f = float(rng.uniform(0.8, 2.0)) * max(W, H)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = f, f * float(rng.uniform(0.95, 1.05)), W / 2 + float(rng.uniform(-3, 3)), H / 2 + float(rng.uniform(-3, 3))
        pk = onp.scale_proj_mtx(calib["proj_mtx"], psize, (1920, 1080))
        cd = calib["cam_dist"] * float(rng.uniform(0.0, 1.5))
        pd = calib["proj_dist"] * float(rng.uniform(0.0, 1.5))
        Rm, T = rot_y(float(rng.uniform(-30, -10))), np.array([[float(rng.uniform(0.15, 0.4))], [float(rng.uniform(-0.05, 0.05))], [float(rng.uniform(-0.05, 0.08))]])
        ctx.set_calibration(K, cd, pk, pd, Rm, T)
        ctx.tune("cam_nodes", 2 if case % 4 == 1 else 1)         # every other 4-aligned case: camera rays from the node table whenever it is accepted
        hp, vp, ref = oc.scan_dense(st, psize, K, cd, pk, pd, Rm, T)
        want = np.moveaxis(ref, 0, -1)
        ok = (hp != -1) & (vp != -1)
        stack = ctx.alloc(st.nbytes + 64).upload(st)
        xyz, maps = ctx.alloc(H * W * 12 + 64), ctx.alloc(H * W * 4 + 64)
        voff = (H * W * 2 + 31) // 32 * 32
        for mode in (0, 1, 1 | 4):                               # exact (two kernels), algebraic (fused when eligible), algebraic split
            xyz.zero()
            maps.zero()
            ctx.scan_dev(stack.ptr, R_, N * H * W, H * W, N, H, W, 0, psize, xyz.ptr, None, maps.at(0), maps.at(voff), mode=mode)
            ctx.synchronize()
            tag = f"case {case} mode {mode}: N={N} H={H} W={W} R={R_} proj={psize}"
            assert np.array_equal(maps.download((H, W), np.int16), hp) and np.array_equal(maps.download((H, W), np.int16, voff), vp), tag
            got = xyz.download((H, W, 3), np.float32)
            assert np.array_equal(np.isfinite(got[..., 0]), ok), tag
            # every decodable pixel, elementwise, in every mode: the cancelled form switches to the reference's own float32
            # intermediates where an angle of the triangle is below ~3.6 degrees (tri_math.h: law_of_sines_guarded)
            np.testing.assert_allclose(got[ok], want[ok], rtol=XYZ_RTOL, atol=0, err_msg=tag)
            if mode != 0 and ok.any():
                err = np.abs(got[ok].astype(np.float64) - want[ok]) / np.abs(want[ok])
                worst[0] = max(worst[0], float(err.max()))
        for b in (stack, xyz, maps):
            b.free()
    ctx.tune("cam_nodes", 1)
    print(f"cancelled form, worst elementwise relative error over the fuzz scans: {worst[0]:.3e}")


def test_host_api_fuzz_vs_oracle(ctx, calib):
    """Seeded random cases through the reference-shaped host API: get_codes, codes_to_pixels (1-3 runs), get_cam_proj_pts
    (both orders, with / without colours, maps with -1 holes and codes beyond the projector size), triangulate (both modes, flat
    triangles included), filter_3d_pts (NaN / inf / boundary values).  Integer and index outputs bit-exact, XYZ within 1e-4."""
    rng = np.random.default_rng(4242)
    for case in range(60 * int(os.environ.get("SLGC_FUZZ_SCALE", "1"))):
        N = int(rng.integers(14, 63))
        L = (N - 2) // 4
        H, W = int(rng.integers(1, 20)), int(rng.integers(1, 70))
        tag = f"case {case}: N={N} H={H} W={W}"
        # ---- codes and maps
        st = _fuzz_stack(rng, 1, N, H, W)[0]
        hc, vc = ctx.codes(st if case % 2 else st.astype(np.float64))
        rhc, rvc = oc.get_codes(st)
        assert np.array_equal(hc, rhc) and np.array_equal(vc, rvc), tag
        R_ = int(rng.integers(1, 4))
        hcs = rng.integers(-1, 2, (R_, L, H, W)).astype(np.int8)
        vcs = rng.integers(-1, 2, (R_, L, H, W)).astype(np.int8)
        gp = ctx.codes_to_pixels(hcs if R_ > 1 else hcs[0], vcs if R_ > 1 else vcs[0])
        rp = oc.codes_to_pixels(hcs, vcs)
        assert np.array_equal(gp[0], rp[0]) and np.array_equal(gp[1], rp[1]), tag
        # ---- correspondences
        psize = (int(rng.integers(2, (2 << L) + 8)), int(rng.integers(2, (2 << L) + 8)))
        hp = rng.integers(-1, 1 << L, (H, W)).astype(np.int64)
        vp = rng.integers(-1, 1 << L, (H, W)).astype(np.int64)
        hp[rng.random((H, W)) < 0.3] = -1
        white = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if case % 3 else None
        for order in ("x", "row"):
            g = ctx.cam_proj_pts(hp, vp, (W, H), psize, white, order={"x": 0, "row": 1}[order])
            r = oc.cam_proj_pts(hp, vp, (W, H), psize, white, order=order)
            assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1]), tag
            assert (g[2] is None and r[2] is None) or np.array_equal(g[2], r[2]), tag
        # ---- triangulation of the list (random calibration; some pairs nearly parallel)
        K = calib["cam_mtx"].copy()
        f = float(rng.uniform(0.8, 2.0)) * max(W, H, 8)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = f, f, W / 2, H / 2
        pk = onp.scale_proj_mtx(calib["proj_mtx"], psize, (1920, 1080))
        cd, pd = calib["cam_dist"] * float(rng.uniform(0, 1.5)), calib["proj_dist"] * float(rng.uniform(0, 1.5))
        if case % 5 == 0:                                      # violent distortion: the icdist < 0 bail-out of undistortPoints fires for some pixels
            cd = calib["cam_dist"] * float(rng.uniform(-60, 60))
            pd = calib["proj_dist"] * float(rng.uniform(-60, 60))
        Rm, T = rot_y(float(rng.uniform(-30, 30))), np.array([[float(rng.uniform(0.1, 0.4))], [float(rng.uniform(-0.05, 0.05))], [float(rng.uniform(-0.1, 0.1))]])
        ctx.set_calibration(K, cd, pk, pd, Rm, T)
        cam, proj, _ = oc.cam_proj_pts(hp, vp, (W, H), psize, None, order="x")
        want = oc.triangulate(cam, proj, K, cd, pk, pd, Rm, T)
        for mode in (0, 1):
            got = ctx.triangulate(cam, proj, mode=mode)
            fin = np.isfinite(want)
            if mode == 0:
                assert np.array_equal(np.isfinite(got), fin), tag           # the exact chain reproduces the reference's NaN / inf too
            np.testing.assert_allclose(got[fin], want[fin], rtol=XYZ_RTOL, atol=0, err_msg=f"{tag} mode {mode}")
        # ---- box filter
        M = int(rng.integers(0, 200))
        pts = rng.uniform(-0.8, 0.8, (3, M))
        if M:
            pts[:, rng.integers(0, M, 3)] = np.array([[np.nan], [0.5], [-0.5]])       # NaN, exactly on the threshold (strict <)
            pts[0, rng.integers(0, M)] = np.inf
        col = rng.uniform(0, 1, (M, 3)) if case % 2 else None
        gf, rf = ctx.filter_3d_pts(pts, col, 0.5), oc.filter_3d_pts(pts, col, 0.5)
        assert np.array_equal(gf[0], rf[0]) and ((gf[1] is None and rf[1] is None) or np.array_equal(gf[1], rf[1])), tag


def test_scan_dev_ragged_sizes(ctx, calib):
    """Bands whose pixel count is not a multiple of 4 (or whose width is odd) take the two-kernel path with byte-wide tails."""
    N = 26
    K = calib["cam_mtx"].copy()
    psize = (200, 150)
    pk = onp.scale_proj_mtx(calib["proj_mtx"], psize, (1920, 1080))
    R, T = rot_y(-20.0), np.array([[0.25], [0.02], [0.04]])
    for (H, W) in ((7, 33), (5, 64), (9, 130), (4, 33), (8, 131), (12, 17)):      # the last three: fused kernel, 4-pixel groups straddle rows
        st, _, _ = onp.synth_scene_int(N, H, W, seed=H)
        K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 150.0, 150.0
        ctx.set_calibration(K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
        hp, vp, ref = oc.scan_dense(st, psize, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
        stack = ctx.alloc(st.nbytes + 64).upload(st)
        xyz = ctx.alloc(H * W * 12 + 64)
        ctx.scan_dev(stack.ptr, 1, st.nbytes, H * W, N, H, W, 0, psize, xyz.ptr, None, mode=1)
        ctx.synchronize()
        got = xyz.download((H, W, 3), np.float32)
        ok = (hp != -1) & (vp != -1)
        assert np.array_equal(np.isfinite(got[..., 0]), ok)
        np.testing.assert_allclose(got[ok], np.moveaxis(ref, 0, -1)[ok], rtol=XYZ_RTOL, atol=0)
        stack.free()
        xyz.free()


def test_scan_to_cloud_matches_reference_pipeline(calib):
    """One-upload pipeline == the reference's script 3 tail + script 4 (restated by the oracle) on the same captures."""
    from scanner.pipeline import scan_to_cloud
    N, H, W = 42, 72, 160
    rng = np.random.default_rng(31)
    run0, _, _ = onp.synth_scene_int(N, H, W, seed=3, noise=4)
    run1, _, _ = onp.synth_scene_int(N, H, W, seed=4, noise=9)
    white = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    K = calib["cam_mtx"].copy()
    K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 220.0, 220.0
    psize, pcal = (200, 150), (1920, 1080)
    R, T = rot_y(-20.0), np.array([[0.25], [0.02], [0.04]])
    pm = calib["proj_mtx"].copy()
    pk = onp.scale_proj_mtx(pm, psize, pcal)
    for runs in ([run0], [run0, run1], [run0.astype(np.float64), run1.astype(np.float64)]):
        rh, rv = oc.decode(np.stack(runs))
        rcam, rproj, rcol = oc.cam_proj_pts(rh, rv, (W, H), psize, white, order="x")
        rpts = oc.triangulate(rcam, rproj, K, calib["cam_dist"], pk, calib["proj_dist"], R, T)
        for thr in (None, 0.5):
            out = scan_to_cloud(runs if len(runs) > 1 else runs[0], K, calib["cam_dist"], psize, pcal, pm, calib["proj_dist"], R, T,
                                img_white=white, threshold=thr, return_lists=True)
            assert np.array_equal(pm, calib["proj_mtx"])                                   # not mutated
            assert np.array_equal(out["h_pixels"], rh) and np.array_equal(out["v_pixels"], rv)
            assert np.array_equal(out["cam_pts"], rcam) and np.array_equal(out["proj_pts"], rproj)
            if thr is None:
                epts, ecol = rpts, rcol
            else:
                epts, ecol = oc.filter_3d_pts(rpts, rcol, thr)
                # the box filter is a strict compare on fp64 values that may differ in the last bits: compare on the oracle's mask
                keep = (np.abs(rpts) < thr).all(axis=0)
                borderline = np.abs(np.abs(rpts) - thr).min(axis=0) < 1e-9
                assert not borderline[keep].any()
            assert out["pts"].shape == epts.shape and out["pts"].dtype == np.float64
            np.testing.assert_allclose(out["pts"], epts, rtol=1e-9, atol=1e-12)
            assert np.array_equal(out["colors"], ecol)
    out = scan_to_cloud(run0, K, calib["cam_dist"], psize, pcal, pm, calib["proj_dist"], R, T)      # no colours, no filter
    assert out["colors"] is None and out["pts"].shape[1] == out["n_unfiltered"] > 100


def test_results_live_in_recycled_buffers(ctx):
    """scanner/_native.py: large results are ordinary writable arrays over buffers that go back to a pool when the array and all its
    views are gone -- and not before."""
    import gc
    st = onp.synth_scene_int(26, 1024, 1100, seed=5)[0]                    # 9 MB per int64 map: above the pool's 8 MB floor
    h, v = ctx.decode(st)
    assert h.dtype == np.int64 and h.flags.writeable and h.flags.c_contiguous
    ref_h, ref_v = oc.decode(st)
    assert np.array_equal(h, ref_h) and np.array_equal(v, ref_v)
    h[0, 0] = 123                                                          # the caller owns it
    first = {h.ctypes.data, v.ctypes.data}
    keep = h[10:20]                                                        # a view keeps the buffer out of the pool
    keep_base = h.ctypes.data
    del h, v
    gc.collect()
    h2, v2 = ctx.decode(st)
    assert keep_base not in (h2.ctypes.data, v2.ctypes.data) and np.array_equal(keep, ref_h[10:20])
    assert np.array_equal(h2, ref_h) and np.array_equal(v2, ref_v)
    second = {h2.ctypes.data, v2.ctypes.data}
    del keep, h2, v2
    gc.collect()
    h3, v3 = ctx.decode(st)
    assert {h3.ctypes.data, v3.ctypes.data} <= (first | second)           # recycled buffers
    assert np.array_equal(h3, ref_h) and np.array_equal(v3, ref_v)
    # the pool can be emptied, and a float64 stack that does not hold grey levels leaves no pinned staging behind
    from scanner import _native
    del h3, v3
    gc.collect()
    assert _native.result_pool_bytes() > 0
    _native.result_pool_clear()
    assert _native.result_pool_bytes() == 0
    for bad_at in (0, st.size - 1):                                        # caught by the look at the head / only by the full pass
        f = st.astype(np.float64)
        f.reshape(-1)[bad_at] += 0.5
        hf, vf = ctx.decode(f)
        assert ctx.last_input_path() == 2
        rh, rv = oc.decode(f)
        assert np.array_equal(hf, rh) and np.array_equal(vf, rv)
    hn, vn = ctx.decode(st.astype(np.float64))
    assert ctx.last_input_path() == 1 and np.array_equal(hn, ref_h) and np.array_equal(vn, ref_v)


# ----------------------------------------------------------------------------------------- ingest ("next" rows)
def test_remove_bad_images_and_diff_counts(ctx):
    from conftest import load_cases
    from scanner.grayCode.decode_codes import remove_bad_images
    for name, c in load_cases("ingest.npz").items():
        fr = c["frames"]
        assert np.array_equal(ctx.frame_diff_counts(fr, 50), onp.frame_diff_counts(fr, 50)), name
        assert remove_bad_images(fr, ctx=ctx) == list(c["kept"]), name                   # the reference's own output
        u8 = np.clip(fr, 0, 255).astype(np.uint8)
        assert np.array_equal(ctx.frame_diff_counts(u8, 50), onp.frame_diff_counts(u8, 50)), name
    rng = np.random.default_rng(41)
    big = rng.integers(0, 256, (5, 301, 517, 3), dtype=np.uint8)
    assert np.array_equal(ctx.frame_diff_counts(big, 50), onp.frame_diff_counts(big, 50))
    assert ctx.frame_diff_counts(big[:1], 50).shape == (0,)
    # the 16-pixels-per-lane kernel (uint8, a multiple of 16 pixels per frame) against the generic one and NumPy, every kind of threshold
    fr = rng.integers(0, 256, (9, 64, 112), dtype=np.uint8)
    fr[3], fr[4] = 0, 255                                                    # the extreme differences, both signs
    fr[6] = fr[5]                                                            # and none at all
    fr[7, :, ::2] = np.where(fr[6, :, ::2] > 128, fr[6, :, ::2] - 51, fr[6, :, ::2] + 51)   # exactly threshold + 1 / on odd and even bytes
    fr[7, :, 1::2] = np.where(fr[6, :, 1::2] > 128, fr[6, :, 1::2] - 50, fr[6, :, 1::2] + 50)
    for thresh in (-1.0, 0.0, 0.5, 1.0, 49.9, 50.0, 50.5, 127.0, 254.0, 254.5, 255.0, 300.0, float("nan")):
        ref = onp.frame_diff_counts(fr, thresh)
        assert np.array_equal(ctx.frame_diff_counts(fr, thresh), ref), thresh
        assert np.array_equal(ctx.frame_diff_counts(fr[:, :, :111], thresh), onp.frame_diff_counts(fr[:, :, :111], thresh)), thresh   # generic kernel
        assert np.array_equal(ctx.frame_diff_counts(fr.astype(np.float64), thresh), ref), thresh
    assert np.array_equal(ctx.frame_diff_counts(fr[:2], 50), onp.frame_diff_counts(fr[:2], 50))
    many = rng.integers(0, 256, (70, 16, 16), dtype=np.uint8)                # more pairs than lanes in a wave
    assert np.array_equal(ctx.frame_diff_counts(many, 50), onp.frame_diff_counts(many, 50))
    # frames that already sit in HBM
    d_fr, d_cnt = ctx.alloc(fr.nbytes).upload(fr), ctx.alloc(8 * 16)
    ctx.frame_diff_counts_dev(d_fr.ptr, len(fr), fr[0].size, 50, d_cnt.ptr)
    ctx.synchronize()
    assert np.array_equal(d_cnt.download((len(fr) - 1,), np.uint64).astype(np.int64), onp.frame_diff_counts(fr, 50))
    d_fr.free()
    d_cnt.free()


def test_to_gray_fixed_point_luma(ctx):
    """OpenCV's BGR2GRAY arithmetic (parity with cv2 itself is unpinned: OpenCV is not installed in the build container)."""
    from scanner.grayCode.decode_codes import to_gray
    rng = np.random.default_rng(42)
    for shape in ((3, 40, 64, 3), (2, 37, 53, 3), (1, 1, 1, 3)):
        im = rng.integers(0, 256, shape, dtype=np.uint8)
        for bits in (15, 14):
            assert np.array_equal(ctx.to_gray(im, bits), onp.bgr_to_gray(im, bits))
    im = np.zeros((1, 2, 4, 3), np.uint8)
    im[0, 0] = [[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]
    g = to_gray(im.astype(np.float64), ctx=ctx)
    assert g.dtype == np.float64 and list(g[0, 0]) == [255.0, 29.0, 150.0, 76.0]


def test_statistical_outlier_removal_and_ply(ctx, tmp_path):
    """k-NN mean distances vs scipy's exact cKDTree on a scanner-like surface with outliers, clusters and duplicates."""
    from scanner import pointcloud as pc
    rng = np.random.default_rng(51)
    yy, xx = np.mgrid[0:240, 0:320]
    surf = np.stack([xx * 1e-3, yy * 1e-3, 0.4 + 0.05 * np.sin(xx / 40.0) * np.cos(yy / 30.0)], -1).reshape(-1, 3)
    surf = surf[rng.random(len(surf)) < 0.8] + rng.normal(0, 2e-5, (1, 3))
    out = rng.uniform([-0.1, -0.1, 0.2], [0.45, 0.35, 0.7], (400, 3))                 # sparse outliers
    clump = rng.normal([0.5, 0.5, 0.5], 1e-6, (300, 3))                               # one dense far-away clump
    pts = np.concatenate([surf, out, clump, surf[:50]]).astype(np.float32)            # + exact duplicates
    pts = pts[rng.permutation(len(pts))]
    for k in (20, 8, 33):
        got = ctx.knn_mean_distance(pts, k)
        ref = onp.knn_mean_distance(pts, k)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-15)
    inl, ind = pc.remove_statistical_outlier(pts.T, 20, 0.5, ctx=ctx)                 # (3,M) like triangulate() returns
    rind = onp.remove_statistical_outlier(pts, 20, 0.5)
    assert np.array_equal(ind, rind) and inl.shape == (len(ind), 3) and inl.dtype == np.float64
    assert 0.5 * len(pts) < len(ind) < len(pts)
    col = rng.random((len(pts), 3))
    p2, c2, i2 = pc.save_point_cloud(pts.T.astype(np.float64), col, str(tmp_path), ctx=ctx)
    raw = open(tmp_path / "cloud.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert f"element vertex {len(i2)}".encode() in head and b"property double x" in head and b"property uchar blue" in head
    rec = np.frombuffer(body, dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1"), ("g", "u1"), ("b", "u1")]))
    assert len(rec) == len(i2) and np.array_equal(rec["x"], p2[:, 0]) and np.array_equal(rec["g"], np.round(c2[:, 1] * 255).astype(np.uint8))
    with pytest.raises(ValueError):
        ctx.knn_mean_distance(pts[:5], 20)                                           # k > number of points
    # small clouds with k close to M: the grid collapses to one cell, where the scan is exhaustive whatever the k-th distance is
    for M, k in ((25, 20), (20, 20), (64, 33), (3, 1), (1, 1)):
        small = rng.uniform(-1, 1, (M, 3)).astype(np.float32) * np.array([1.0, 0.02, 5.0], np.float32)
        np.testing.assert_allclose(ctx.knn_mean_distance(small, k), onp.knn_mean_distance(small, k), rtol=1e-12, atol=1e-15)
    # many cells: the counting sort's exclusive scan runs over several tile levels (own reduce-then-scan, csrc/cloud.hip)
    big = rng.uniform(0, 1, (300_000, 3)).astype(np.float32) * np.array([1.0, 1.0, 1e-3], np.float32)
    np.testing.assert_allclose(ctx.knn_mean_distance(big, 4), onp.knn_mean_distance(big, 4), rtol=1e-12, atol=1e-15)
    assert ctx.knn_mean_distance(np.zeros((0, 3), np.float32), 20).shape == (0,)
    # the device-resident entry: the same numbers bit for bit on a cloud that already sits in HBM; a non-finite coordinate there is refused too
    d_pts, d_mean = ctx.alloc(mixed_bytes := len(big) * 12).upload(big), ctx.alloc(len(big) * 8)
    ctx.knn_mean_distance_dev(d_pts.ptr, len(big), 4, d_mean.ptr)
    ctx.synchronize()
    assert np.array_equal(d_mean.download((len(big),), np.float64), ctx.knn_mean_distance(big, 4)) and mixed_bytes
    ctx.dev_memset(d_pts.ptr + 12 * 777 + 4, 0xff, 4)                              # 0xffffffff = a NaN
    with pytest.raises(Exception, match="non-finite"):
        ctx.knn_mean_distance_dev(d_pts.ptr, len(big), 4, d_mean.ptr)
    d_pts.free()
    d_mean.free()
    # a non-finite coordinate is refused, and named (the check runs on the device behind the upload)
    for bad_value in (np.nan, np.inf, -np.inf):
        poisoned = big[:50_000].copy()
        poisoned[31_337, 1] = bad_value
        with pytest.raises(Exception, match="non-finite coordinate at point 31337"):
            ctx.knn_mean_distance(poisoned, 4)
    # outliers far from a dense sheet and k = 33 / 64: the stragglers' pass (16 lanes per query, wider blocks of the same grid), every K
    sheet = rng.uniform(0, 1, (60_000, 3)).astype(np.float32) * np.array([1.0, 1.0, 2e-3], np.float32)
    far = rng.uniform(-3, 4, (300, 3)).astype(np.float32)
    mixed = np.concatenate([sheet, far, far[:40] + np.float32(1e-4)])
    for k in (20, 33, 64):
        np.testing.assert_allclose(ctx.knn_mean_distance(mixed, k), onp.knn_mean_distance(mixed, k), rtol=1e-12, atol=1e-15)


def test_decode_randomised_configurations(ctx):
    """40 random (N, H, W, runs, eps, dtype) draws through the host API and the device API against the C oracle."""
    rng = np.random.default_rng(2026)
    for trial in range(40):
        N = int(rng.integers(14, 66))
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 150))
        R = int(rng.integers(1, 4))
        eps = [1, 1, 1, 0, 2, 7, 0.5, 1.25][int(rng.integers(0, 8))]
        kind = int(rng.integers(0, 3))
        if kind == 0:
            st = rng.integers(0, 256, (R, N, H, W), dtype=np.uint8)
        elif kind == 1:
            st = np.stack([onp.synth_scene(N, H, W, seed=int(rng.integers(1, 1000)), noise=int(rng.integers(0, 12))) for _ in range(R)])
        else:
            st = rng.integers(0, 256, (R, N, H, W), dtype=np.uint8)
            st[:, 0] = rng.integers(0, 3, (R, H, W))
            st[:, 1] = rng.integers(0, 3, (R, H, W))              # many 0/0 and tie pixels
        ref = oc.decode(st, eps=eps)
        as_f64 = bool(rng.integers(0, 2))
        runs = [s.astype(np.float64) if as_f64 else s for s in st]
        hp, vp = ctx.decode(runs if R > 1 else runs[0], eps=eps)
        assert np.array_equal(hp, ref[0]) and np.array_equal(vp, ref[1]), (trial, N, H, W, R, eps, kind, as_f64)
        if float(eps).is_integer() and eps >= 0:
            h, v = dev_decode(ctx, st, 0, eps=eps)
            assert np.array_equal(h, ref[0]) and np.array_equal(v, ref[1]), (trial, N, H, W, R, eps, kind)


def test_config1_full_size_against_the_reference(calib):
    """BASELINE.json configs[0] at full size through the one-call pipeline, against outputs of the reference itself."""
    import hashlib
    from scanner.pipeline import scan_to_cloud
    z = np.load(os.path.join(GOLDEN, "config1.npz"))
    N, H, W, seed, noise = (int(x) for x in z["params"])
    stack, _, _ = onp.synth_scene_int(N, H, W, seed=seed, noise=noise, shadow=True)
    assert hashlib.sha256(stack.tobytes()).digest() == z["stack_sha256"].tobytes()
    white = np.repeat(stack[1][:, :, None], 3, axis=2)
    for st in (stack, stack.astype(np.float64)):
        out = scan_to_cloud(st, calib["cam_mtx"], calib["cam_dist"], (1280, 800), (1920, 1080), calib["proj_mtx"], calib["proj_dist"],
                            z["R"], z["T"], img_white=white, return_lists=True)
        assert np.array_equal(out["h_pixels"], z["h_pixels"]) and np.array_equal(out["v_pixels"], z["v_pixels"])
        assert out["pts"].shape == (3, int(z["M"][0]))
        assert hashlib.sha256(out["cam_pts"].tobytes()).digest() == z["cam_sha256"].tobytes()
        assert hashlib.sha256(out["proj_pts"].tobytes()).digest() == z["proj_sha256"].tobytes()
        sel = z["sample_index"]
        np.testing.assert_allclose(out["pts"][:, sel], z["pts_sample"], rtol=XYZ_RTOL, atol=0)
        np.testing.assert_allclose(out["pts"][:, sel], z["pts_sample"], rtol=1e-9, atol=1e-12)
        assert np.array_equal(out["colors"][sel], z["colors_sample"])
    kept = scan_to_cloud(stack, calib["cam_mtx"], calib["cam_dist"], (1280, 800), (1920, 1080), calib["proj_mtx"], calib["proj_dist"],
                         z["R"], z["T"], threshold=0.5)
    assert kept["pts"].shape[1] == int(z["M"][1])
