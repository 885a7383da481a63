"""Host logic of the direct exchange that needs no GPU: scanner.sharded.DirectExchange against a recording stand-in for the context -- in-place
checks, the (h, v) pair and the single-buffer forms mapping onto slgc_direct_allgatherv_begin, release batching (at most three buffers per call),
registration happening once per buffer -- and ShardedScanner's announce-before-reuse protocol over both buffer sets."""
import pytest

from scanner import sharded


class FakeBuf:
    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def at(self, off):
        assert 0 <= off <= self.nbytes
        return self.ptr + off


class FakeCtx:
    def __init__(self):
        self.calls = []
        self._next = 0x10000

    def alloc(self, n):
        b = FakeBuf(self._next, n)
        self._next += (n + 0xfff) & ~0xfff
        return b

    def __getattr__(self, name):                      # every other C-ABI call is recorded
        def rec(*a, **k):
            self.calls.append((name, a, k))
        return rec


def test_direct_exchange_maps_the_protocol_onto_the_c_abi():
    ctx = FakeCtx()
    ex = sharded.DirectExchange(ctx, rank=1, nranks=3, key="k")
    assert ctx.calls[0][0] == "direct_init" and ctx.calls[0][1] == (1, 3, "k")
    h, v, x, other = ctx.alloc(6000), ctx.alloc(6000), ctx.alloc(36000), ctx.alloc(64)
    ex.register([h, v, x])
    ex.register([h])                                                        # already registered: not again
    assert [c[0] for c in ctx.calls].count("direct_register") == 3
    counts, displs = [2000, 2000, 2000], [0, 2000, 4000]
    ex.allgatherv_pair_begin(h.at(2000), h, v.at(2000), v, counts, displs, 2)
    name, a, _ = ctx.calls[-1]
    assert name == "direct_allgatherv_begin" and a[0] == [h.ptr, v.ptr] and a[1] == [(counts, displs)] * 2 and a[2] == 2
    ex.allgatherv(x.at(12000), x, [12000] * 3, [0, 12000, 24000])          # begin in slot 3 + wait
    assert [c[0] for c in ctx.calls[-2:]] == ["direct_allgatherv_begin", "direct_wait"] and ctx.calls[-1][1] == (3,)
    with pytest.raises(ValueError, match="in place"):
        ex.allgatherv_begin(h.at(0), h, counts, displs, 0)                  # rank 1's band sits at 2000
    with pytest.raises(ValueError, match="not registered"):
        ex.allgatherv_begin(other.at(0), other, [16, 16, 16], [0, 16, 32], 0)
    n = len(ctx.calls)
    ex.release([h, v, x, other, h])                                         # registered ones only, three per call
    rel = [c for c in ctx.calls[n:] if c[0] == "direct_release"]
    assert [len(c[1][0]) for c in rel] == [3, 1] and rel[0][1][0] == [h.ptr, v.ptr, x.ptr]


class FakeScanCtx(FakeCtx):
    rank, nranks = 0, 2


@pytest.mark.parametrize("kind,wire", [("maps", "int16"), ("maps", "hv24"), ("xyz", "int16")])
def test_sharded_scanner_announces_every_reuse_of_a_buffer_set(kind, wire):
    ctx = FakeScanCtx()
    ex = sharded.DirectExchange(ctx, 0, 2, "k")
    plan = sharded.ShardPlan(48, 64, 2)
    sc = sharded.ShardedScanner(ctx, ex, plan, (160, 120), 26, exchange_kind=kind, wire=wire)
    per_set = {"maps": 1 if wire == "hv24" else 2, "xyz": 3}[kind]
    assert [c[0] for c in ctx.calls].count("direct_register") == 2 * per_set      # both sets, each buffer once, same order on every rank
    for _ in range(5):
        sc.submit(0x1234, 64 * 24)
    sc.flush()
    names = [c[0] for c in ctx.calls]
    # set 0 is used by submits 0, 2, 4 and set 1 by 1, 3: the first use of a set needs no release, every later one is announced BEFORE its kernels
    rel = [i for i, n_ in enumerate(names) if n_ == "direct_release"]
    assert len(rel) == 3
    band = "scan_dev" if kind == "xyz" else "decode_dev"
    kern = [i for i, n_ in enumerate(names) if n_ == band]
    assert len(kern) == 5 and all(any(r < k for r in rel) for k in kern[2:])
    for r in rel:
        assert names[r + 1] == band                                          # release -> the band kernel that overwrites the set
    begins = [c for c in ctx.calls if c[0] == "direct_allgatherv_begin"]
    assert len(begins) == 5 * (2 if kind == "xyz" else 1)
    # close(): every registered buffer leaves the exchange (collectively, registration order) BEFORE anything is freed; a second close is a no-op
    n = len(ctx.calls)
    freed = []
    for b in [x for st in sc._sets for x in st] + list(getattr(sc, "_wire", None) or []) + [sc.count, sc.xyz_full]:
        b.free = (lambda b=b: freed.append(b.ptr))
    sc.close()
    tail = [c[0] for c in ctx.calls[n:]]
    assert tail.count("direct_unregister") == 2 * per_set and "synchronize" in tail and tail.index("synchronize") < tail.index("direct_unregister")
    assert len(freed) == len(set(freed)) >= 2 * per_set + 1
    sc.close()
    assert [c[0] for c in ctx.calls[n:]].count("direct_unregister") == 2 * per_set
    waits = [c for c in ctx.calls if c[0] == "direct_wait"]
    assert len(waits) == len(begins)                                         # every exchange is waited for exactly once
