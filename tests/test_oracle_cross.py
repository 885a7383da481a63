"""The two restatements (NumPy and plain C) must agree with each other bit for bit on inputs far outside the golden set:
every frame count 14..65, ragged shapes, integer and non-integer eps, float64 stacks with NaN/inf, multi-run merges."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle_c as oc
import oracle_np as onp


@pytest.mark.parametrize("N", list(range(14, 66, 3)) + [65])
def test_every_frame_count(N):
    rng = np.random.default_rng(N)
    stack = rng.integers(0, 256, (2, N, 9, 13), dtype=np.uint8)
    stack[0, :, :2, :3] = 0
    for eps in (1, 0, 2.5):
        a, b = onp.decode(stack, eps=eps), oc.decode(stack, eps=eps)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    h, v = onp.frame_ids(N), oc.frame_ids(N)
    assert np.array_equal(h[0], v[0]) and np.array_equal(h[1], v[1])
    assert onp.code_len(N) == oc.lib().orc_code_len(N)


@settings(max_examples=40, deadline=None)
@given(N=st.integers(14, 40), H=st.integers(1, 12), W=st.integers(1, 17), seed=st.integers(0, 2 ** 31 - 1),
       eps=st.sampled_from([1, 0, 3, 0.5, -1, 1e-9, 254.5]))
def test_random_float_stacks(N, H, W, seed, eps):
    rng = np.random.default_rng(seed)
    stack = rng.uniform(-5, 260, (N, H, W))
    stack[rng.random(stack.shape) < 0.02] = np.nan
    stack[rng.random(stack.shape) < 0.01] = np.inf
    la, lb = onp.direct_indirect(stack), oc.direct_indirect(stack)
    assert np.array_equal(la[0], lb[0], equal_nan=True) and np.array_equal(la[1], lb[1], equal_nan=True)
    ca, cb = onp.is_lit(stack, *la, eps=eps), oc.is_lit(stack, *lb, eps=eps)
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
    pa, pb = onp.codes_to_pixels(*ca), oc.codes_to_pixels(*cb)
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])


@settings(max_examples=25, deadline=None)
@given(H=st.integers(1, 20), W=st.integers(1, 20), seed=st.integers(0, 2 ** 31 - 1), frac=st.floats(0, 1))
def test_correspondence_orders(H, W, seed, frac):
    rng = np.random.default_rng(seed)
    h = rng.integers(-1, 2000, (H, W)).astype(np.int64)
    v = rng.integers(-1, 1200, (H, W)).astype(np.int64)
    h[rng.random((H, W)) < frac] = -1
    white = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for order in ("x", "row"):
        a = onp.cam_proj_pts(h, v, (W, H), (1280, 800), white, order=order)
        b = oc.cam_proj_pts(h, v, (W, H), (1280, 800), white, order=order)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    loops = onp.cam_proj_pts_loops(h, v, (W, H), (1280, 800), white)           # the reference-style nested loops (x-major)
    xmaj = oc.cam_proj_pts(h, v, (W, H), (1280, 800), white, order="x")
    assert np.array_equal(loops[0].reshape(-1, 2), xmaj[0]) and np.array_equal(loops[1].reshape(-1, 2), xmaj[1])
    assert np.array_equal(np.asarray(loops[2]).reshape(-1, 3), xmaj[2])
