#!/usr/bin/env python3
"""Runs the ctypes stub printed in INTEGRATION.md section B (kept in sync by hand) against the oracle on a small stack."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes as C, numpy as np
lib = C.CDLL("3dscanner-graycode_amd/lib/libslgc.so")
lib.slgc_last_error.restype = C.c_char_p
ctx = C.c_void_p()
assert lib.slgc_create(0, C.byref(ctx)) == 0                      # fails with SLGC_ENODEV (-2) without an MI355X
def get_codes(images):                                            # images: float64 or uint8 [N,H,W]
    st = np.ascontiguousarray(images)
    dtype = 0 if st.dtype == np.uint8 else 1                      # SLGC_U8 / SLGC_F64
    st = st if dtype == 0 else st.astype(np.float64, copy=False)
    N, H, W = st.shape
    L = int((N - 2) / 4)
    h = np.empty((L, H, W), np.int8); v = np.empty((L, H, W), np.int8)
    rc = lib.slgc_codes(ctx, st.ctypes.data_as(C.c_void_p), dtype, N, H, W, C.c_double(1), C.c_double(10),
                        h.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))
    if rc: raise RuntimeError(lib.slgc_last_error(ctx).decode())
    return h, v

import oracle_c as oc
st = np.random.default_rng(0).integers(0, 256, (26, 20, 33), dtype=np.uint8)
for stack in (st, st.astype(np.float64)):
    h, v = get_codes(stack)
    rh, rv = oc.get_codes(st)
    assert np.array_equal(h, rh) and np.array_equal(v, rv)
try:
    get_codes(np.zeros((10, 4, 4), np.uint8))
    raise SystemExit("expected an error for N < 14")
except RuntimeError as e:
    print("error path:", e)
print("INTEGRATION.md stub ok")
