"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/slgc.h declares; the Python mirror exposes the reference's names; the product path has no
CPU fallback and never imports the oracle."""
import ast
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import PKG, ROOT


def header_functions(name=None):
    """name = "slgc.h" (the reference-facing ABI) or "slgc_bench.h" (measurement / diagnostic exports); None = both."""
    out = set()
    for h in ([name] if name else ["slgc.h", "slgc_bench.h"]):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(slgc_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_library_exports_every_declared_symbol():
    from scanner import _native
    lib = _native.lib()
    declared = header_functions()
    abi, bench = header_functions("slgc.h"), header_functions("slgc_bench.h")
    assert len(abi) >= 35 and not set(abi) & set(bench), set(abi) & set(bench)
    # the reference-facing header stays free of measurement / diagnostic exports
    assert not [n for n in abi if re.search(r"synth|selftest|move_only|prof_|event_|guard_count", n)], abi
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert sorted(_native.SIGNATURES) == declared, "ctypes signature table out of sync with slgc.h"
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (slgc_[a-z0-9_]+)", out)))
    assert exported == declared


def test_library_identity_no_gpu_needed():
    from scanner import _native
    lib = _native.lib()
    assert lib.slgc_backend() == b"hip:gfx950"
    assert lib.slgc_version() == 100
    assert lib.slgc_strerror(-2) == b"no HIP device"
    assert lib.slgc_device_count() >= 0


def test_fails_loudly_without_device():
    from scanner import _native
    if _native.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_native.SlgcError, match="no CPU fallback"):
        _native.Context(0)
    from scanner.grayCode.decode_codes import get_codes
    with pytest.raises(_native.SlgcError):
        get_codes(np.zeros((14, 4, 4), np.uint8))


def test_python_surface_matches_reference_names():
    import inspect
    from scanner.grayCode import decode_codes as dc, generate_codes as gc
    from scanner.triangulation import Triangulate, Triangulation
    for fn in ("get_direct_indirect", "get_is_lit", "get_codes", "gray_decode", "gray_to_decimal", "decode"):
        assert callable(getattr(dc, fn))
    assert list(inspect.signature(dc.get_is_lit).parameters)[:5] == ["images", "L_d", "L_g", "eps", "m"]
    assert inspect.signature(dc.get_is_lit).parameters["eps"].default == 1
    assert inspect.signature(dc.get_is_lit).parameters["m"].default == 10
    params = list(inspect.signature(Triangulate.__init__).parameters)
    assert params[1:13] == ["h_pixels", "v_pixels", "cam_size", "cam_mtx", "cam_dist", "proj_size", "proj_calib_size",
                            "proj_mtx", "proj_dist", "proj_R", "proj_T", "image_folder"]
    assert inspect.signature(Triangulate.filter_3d_pts).parameters["threshold"].default == .5
    assert issubclass(Triangulation, Triangulate) and callable(Triangulation.compute)
    assert callable(gc.get_gray_codes) and callable(gc.get_image_sequence)


def test_host_helpers_match_golden():
    from conftest import GOLDEN
    from scanner.grayCode import decode_codes as dc, generate_codes as gc
    z = np.load(os.path.join(GOLDEN, "gray_kat.npz"))
    assert [dc.gray_decode(n) for n in range(4096)] == list(z["gray_decode"])
    assert [dc.gray_to_decimal(s) for s in z["g2d_in"]] == list(z["g2d_out"])
    g = np.load(os.path.join(GOLDEN, "generator.npz"))
    for (w, h) in ((64, 32), (40, 24), (16, 16)):
        codes = gc.get_gray_codes(w, h)
        assert np.array_equal(codes, g[f"codes_{w}x{h}"])
        assert np.array_equal(gc.get_image_sequence(codes, w, h), g[f"seq_{w}x{h}"])


def test_triangulate_ctor_scales_proj_mtx_in_place(calib):
    from scanner.triangulation import Triangulate
    pm = calib["proj_mtx"].copy()
    Triangulate(None, None, (1920, 1080), calib["cam_mtx"], calib["cam_dist"], (1280, 800), (1920, 1080), pm,
                calib["proj_dist"], np.eye(3), np.ones((3, 1)), None)
    assert np.isclose(pm[0, 0], calib["proj_mtx"][0, 0] * 1280 / 1920) and np.isclose(pm[1, 2], calib["proj_mtx"][1, 2] * 800 / 1080)


def test_product_never_touches_oracle_or_torch():
    """The shipped package must not import oracle/, torch, or any CPU compute fallback."""
    for dirpath, _, files in os.walk(os.path.join(PKG, "scanner")):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom) and node.module:
                    names = [node.module]
                for n in names:
                    assert not n.startswith(("oracle", "torch", "cv2")), f"{f} imports {n}"
    for f in os.listdir(os.path.join(PKG, "csrc")):
        if not os.path.isfile(os.path.join(PKG, "csrc", f)):
            continue
        text = open(os.path.join(PKG, "csrc", f)).read()
        assert "slgc_oracle" not in text and "orc_" not in text, f


def test_header_is_plain_c_and_links(tmp_path):
    """include/slgc.h is the boundary a non-Python host binds: it must compile as C99 and a C program must link against
    libslgc.so and call the entry points that need no GPU."""
    from scanner import _native
    src = tmp_path / "host.c"
    src.write_text('#include <stdio.h>\n#include "slgc.h"\n#include "slgc_bench.h"\nint main(void){ int r0, rn;\n'
                   ' if (slgc_shard_band(3000, 8, 7, &r0, &rn)) return 2;\n'
                   ' printf("%d %s %d %d\\n", slgc_version(), slgc_backend(), r0, rn); return 0; }\n')
    exe = tmp_path / "host"
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lslgc", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out == ["100", "hip:gfx950", "2625", "375"]


def test_host_only_helpers(tmp_path):
    """Helpers that need no GPU: PLY layout, read_images file order, shard plan."""
    from scanner.grayCode.decode_codes import read_images_order
    from scanner import pointcloud as pc
    names = ["frame_10.jpg", "frame_2.jpg", "frame_1.jpg", "frame_100.jpg", "frame_11.jpg"]
    assert read_images_order(names) == ["frame_2.jpg", "frame_1.jpg", "frame_10.jpg", "frame_11.jpg", "frame_100.jpg"]   # stable, by length only
    pts = np.array([[0.0, 0.5, -1.25], [1.0, 2.0, 3.0]])
    col = np.array([[0.0, 0.5, 1.0], [1.2, -0.1, 0.25]])
    n = pc.write_ply(tmp_path / "c.ply", pts.T, col)                       # (3,M) input like triangulate() returns
    raw = open(tmp_path / "c.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert n == 2 and b"format binary_little_endian 1.0" in head and b"element vertex 2" in head
    rec = np.frombuffer(body, dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1"), ("g", "u1"), ("b", "u1")]))
    assert list(rec["z"]) == [-1.25, 3.0] and list(rec["r"]) == [0, 255] and list(rec["g"]) == [128, 0] and list(rec["b"]) == [255, 64]
    n = pc.write_ply(tmp_path / "n.ply", pts)
    assert b"red" not in open(tmp_path / "n.ply", "rb").read().split(b"end_header")[0]
    with pytest.raises(ValueError):
        pc.write_ply(tmp_path / "bad.ply", pts, col[:1])


def test_namespace_package_merges_with_the_reference_tree(tmp_path):
    """INTEGRATION.md option A: with this package FIRST on sys.path and the reference checkout behind it, the import lines of the
    reference's own scripts (src/3-capture_decode.py:5-8, src/4-triangulate.py:5-6) must resolve -- hot-path modules to this
    build, everything else (acquisition, utils.visualize) to the reference.  cv2 / open3d are not installed: empty stand-in
    modules satisfy the reference's module-level imports (nothing is called)."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "scanner")):
        pytest.skip("reference checkout not present (GPU box)")
    assert not os.path.exists(os.path.join(PKG, "scanner", "__init__.py")), "scanner/ must stay a namespace package like the reference's"
    code = f"""
import sys, types
for name in ("cv2", "open3d", "open3d.core"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["open3d"].core = sys.modules["open3d.core"]
sys.path[:0] = [{PKG!r}, {ref!r}]
from scanner.acquisition import Camera
from scanner.grayCode.generate_codes import get_gray_codes, get_image_sequence
from scanner.grayCode.decode_codes import get_codes, gray_to_decimal
from scanner.utils import visualize
from scanner.triangulation import Triangulate
import scanner.grayCode.decode_codes as dc, scanner.triangulation.triangulate as tr, scanner.utils.visualize as vz, scanner.acquisition.camera as cam
assert dc.__file__.startswith({PKG!r}) and tr.__file__.startswith({PKG!r}), (dc.__file__, tr.__file__)
assert vz.__file__.startswith({ref!r}) and cam.__file__.startswith({ref!r}), (vz.__file__, cam.__file__)
from scanner._native import Context
from scanner.pipeline import scan_to_cloud
from scanner import pointcloud, sharded
print("ok")
"""
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_remove_bad_images_window_rule_equals_the_reference_state_machine():
    """Host logic only: the product's sliding-window form of decode_codes.py:52-66 against the oracle's literal restatement (which
    tests/golden/ingest.npz pins to the reference's own output), on 20 000 random change-count sequences full of ties."""
    import oracle_np as onp
    from scanner.grayCode import decode_codes as dc
    rng = np.random.default_rng(0)

    class Counts:                                   # stands in for the GPU reduction: hands back the counts under test
        def __init__(self, d):
            self.d = d

        def frame_diff_counts(self, images, thresh):
            return self.d

    orig = onp.frame_diff_counts
    try:
        for case in range(20000):
            n = int(rng.integers(4, 40))
            hi = int(rng.choice([2, 3, 5, 50, 100000]))
            d = rng.integers(0, hi, n - 1)
            onp.frame_diff_counts = lambda images, thresh=50, d=d: d
            frames = [None] * n
            assert dc.remove_bad_images(frames, ctx=Counts(d)) == onp.remove_bad_images(frames), (case, d)
    finally:
        onp.frame_diff_counts = orig


def test_read_images_order_shapes_and_types(tmp_path):
    """read_images (decode_codes.py:6-32): sorted(key=len) file order, BGR channel order, float64 [n,H,W,3] + file names like the
    reference.  (Decoded values are Pillow's: parity with cv2.imread is unpinned, PNG is used here so the bytes are exact.)"""
    PIL = pytest.importorskip("PIL.Image")
    from scanner.grayCode.decode_codes import read_images
    rng = np.random.default_rng(3)
    frames = {}
    for name in ("frame_10.png", "frame_2.png", "frame_1.png", "frame_100.png"):
        rgb = rng.integers(0, 256, (6, 9, 3), dtype=np.uint8)
        PIL.fromarray(rgb).save(tmp_path / name)
        frames[name] = rgb
    images, names = read_images(str(tmp_path))
    want = sorted(os.listdir(tmp_path), key=len)
    assert list(names) == want and images.shape == (4, 6, 9, 3) and images.dtype == np.float64
    for i, n in enumerate(want):
        assert np.array_equal(images[i], frames[n][:, :, ::-1])                      # BGR
    u8, _ = read_images(str(tmp_path), dtype=np.uint8)
    assert u8.dtype == np.uint8 and np.array_equal(u8, images.astype(np.uint8))
    empty = tmp_path / "none"
    empty.mkdir()
    assert read_images(str(empty))[0].shape[0] == 0


def test_byte_over_255_by_one_refinement_step_is_the_correctly_rounded_quotient():
    """correspond.hip: unit_of_byte(b) = fma(fma(-q, 255, b), 1/255, q) with q = b * (1/255) must equal the reference's float64 b / 255.0
    (triangulate.py:64,:69) for every byte; the plain product b * (1/255) does not (that is what the negative control shows)."""
    from fractions import Fraction
    rcp = 1.0 / 255.0

    def fma(a, b, c):                                        # exact product-sum, rounded once (Fraction -> float rounds to nearest even)
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    plain_wrong = 0
    for b in range(256):
        q = float(b) * rcp
        plain_wrong += q != b / 255.0
        assert fma(fma(-q, 255.0, float(b)), rcp, q) == b / 255.0
    assert plain_wrong > 0


def test_result_arrays_come_from_recycled_buffers():
    """scanner/_native.py: large results are ordinary writable arrays over buffers that return to a pool when the array and all its views
    are gone (a device-to-host copy into pages that exist already runs at the rate of the link); small ones are plain np.empty arrays."""
    import gc
    from scanner import _native
    a = _native._out((1100, 1024), np.int64)                                 # 9 MB: above the pool's floor
    assert a.shape == (1100, 1024) and a.dtype == np.int64 and a.flags.writeable and a.flags.c_contiguous
    a[3, 4] = 7
    addr = a.ctypes.data
    view = a[10:20]
    del a
    gc.collect()
    b = _native._out((1100, 1024), np.int64)                                 # the view keeps the first buffer out of the pool
    assert b.ctypes.data != addr
    del view, b
    gc.collect()
    c = _native._out((1024, 1100), np.int64)                                 # same size class: one of the two buffers comes back
    d = _native._out((1100, 1024), np.int64)
    assert addr in (c.ctypes.data, d.ctypes.data)
    small = _native._out((4, 4), np.float32)
    assert small.flags.owndata


def test_tools_and_bench_compile():
    """The measurement scripts are part of the evidence: at least they must parse (they need the GPU box to run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) >= 15
    for f in files:
        ast.parse(open(f).read(), filename=f)
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")) + glob.glob(os.path.join(ROOT, "tools", "jobs", "*.sh"))):
        assert subprocess.run(["bash", "-n", f]).returncode == 0, f
