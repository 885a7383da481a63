import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
import numpy as np, oracle_c as oc, oracle_np as onp
from scanner import reference_calibration as rc, _native
H, W, N = 270, 256, 26
st, _, _ = onp.synth_scene_int(N, H, W, seed=H)
K = rc.CAM_MTX.copy(); K[0, 2], K[1, 2], K[0, 0], K[1, 1] = W / 2, H / 2, 400.0, 400.0
psize = (200, 150); pk = onp.scale_proj_mtx(rc.PROJ_MTX, psize, (1920, 1080))
th = np.deg2rad(-20.0); R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]); T = np.array([[0.25], [0.02], [0.04]])
hp, vp, ref = oc.scan_dense(st, psize, K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T)
want = np.moveaxis(ref, 0, -1); ok = (hp != -1) & (vp != -1)
ctx = _native.Context(0); ctx.set_calibration(K, rc.CAM_DIST, pk, rc.PROJ_DIST, R, T)
dh, dv = ctx.alloc(H * W * 2).upload(hp.astype(np.int16)), ctx.alloc(H * W * 2).upload(vp.astype(np.int16))
xyz = ctx.alloc(H * W * 12)
for mode in (0, 1):
    ctx.triangulate_maps_dev(dh.ptr, dv.ptr, H, W, 0, psize, xyz.ptr, None, mode=mode); ctx.synchronize()
    got = xyz.download((H, W, 3), np.float32).astype(np.float64)
    rel = np.abs(got - want) / np.abs(want)
    rel[~ok] = 0
    idx = np.argsort(rel.max(-1).ravel())[-5:]
    for i in idx:
        y, x = divmod(int(i), W)
        print("mode", mode, "px", (y, x), "want", want[y, x], "got", got[y, x], "rel", rel[y, x].max(), "f32(want)", want[y, x].astype(np.float32))
