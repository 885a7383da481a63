"""The per-column ownership rule of the whole-lines list build (csrc/correspond.hip, k_xmajor_lines), restated in Python and checked
exhaustively on random columns: every record of a column is written by exactly one tile, from rows that tile has in LDS, and -- where the
column is dense -- as whole groups of 16 records aligned on the absolute record index.  (The kernel itself is compared bit for bit with the
tile-run kernel and the oracle in tests/test_gpu_fullsize.py; this test pins the rule it evaluates, on the CPU.)"""
import numpy as np
import pytest

NR = 32            # nominal rows of a tile = one chunk of the prefix counts (kLinesNominal); the tile also loads the NR rows below


def tile_window(B0, Bp, cs, n1, n2, ty):
    """-> (ka, kb): the ranks (record index = B0 + rank) tile `ty` writes of a column; B0 = records above its first nominal row (absolute),
    Bp = the same for the tile above, cs = the column's first record, n1 / n2 = valid pixels in its nominal / halo rows.  Mirrors phase 2 (a)."""
    phase = B0 & 15
    to_next = 16 - phase
    S = max(B0 & ~15, cs)                              # first record (of this column) of the group B0 is in
    head_end = min(to_next, n1)
    ka = 0
    if S != B0 and ty > 0 and S >= Bp:
        ka = head_end                                  # the tile above began that group and sees these rows
    owns = n1 > 0 and (S == B0 or to_next < n1)
    kb = min(((phase + n1 - 1) & ~15) + 16 - phase, n1 + n2) if owns else head_end
    return ka, kb


@pytest.mark.parametrize("density", [0.0, 0.01, 0.05, 0.2, 0.5, 0.81, 0.97, 1.0])
def test_every_record_written_once_and_mostly_in_whole_groups(density):
    rng = np.random.default_rng(int(density * 1000) + 1)
    for trial in range(300):
        H = int(rng.integers(1, 400))
        cs = int(rng.integers(0, 1000))                # where the column starts in the x-major order (any alignment)
        structured = trial % 7 == 0
        valid = rng.random(H) < density
        if structured:                                 # bands of valid rows with gaps up to several tiles
            valid = (np.arange(H) % int(rng.integers(33, 120))) < int(rng.integers(1, 60))
        ntiles = -(-H // NR)
        above = np.concatenate([[0], np.cumsum(valid)])            # valid pixels above row y
        written = np.zeros(int(valid.sum()), np.int32)
        partial = 0                                                 # groups a tile touches without writing all 16 of their records
        for ty in range(ntiles):
            y0 = ty * NR
            n1 = int(valid[y0:y0 + NR].sum())
            n2 = int(valid[y0 + NR:y0 + 2 * NR].sum())
            B0 = cs + int(above[y0])
            Bp = cs + int(above[max(0, y0 - NR)]) if ty > 0 else B0
            ka, kb = tile_window(B0, Bp, cs, n1, n2, ty)
            assert 0 <= ka <= kb <= n1 + n2                         # only records of rows the tile has loaded
            written[B0 - cs + ka:B0 - cs + kb] += 1
            if kb > ka:
                a, b = B0 + ka, B0 + kb
                touched = ((b - 1) >> 4) - (a >> 4) + 1
                whole = max(0, (b >> 4) - ((a + 15) >> 4))
                partial += touched - whole
        assert (written == 1).all(), (density, trial, H, cs)
        if density >= 0.81 and not structured:                      # dense columns: whole lines but for the two ends of the column
            assert partial <= 2, (density, trial, H, cs, partial)


def test_dense_column_seams_are_whole_lines():
    """All pixels valid: every tile but the first starts its window on a group boundary and every tile but the last ends on one."""
    H, cs = 10 * NR, 5
    for ty in range(10):
        B0 = cs + ty * NR
        ka, kb = tile_window(B0, cs + max(0, ty - 1) * NR if ty else B0, cs, NR, NR if ty < 9 else 0, ty)
        if ty > 0:
            assert (B0 + ka) % 16 == 0
        if ty < 9:
            assert (B0 + kb) % 16 == 0
