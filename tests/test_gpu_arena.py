"""lsdr_arena (include/lsdr_hip.h, leansdr_amd/csrc/arena.hip): placed stream buffers — one large allocation handed out in windows, the
fastest ones under a probe first; lsdr_ctx_set_arena routes lsdr_malloc through it (the host framework's device pipes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_windows_are_disjoint_aligned_and_come_back(capi, ctx):
    a = capi.Arena(ctx, 1 << 30)
    assert a.nbytes == 1 << 30
    ws = a.place(64 << 20, n_best=3, max_windows=8)
    log = a.probe_log()
    assert 3 <= len(log) <= 8 and all(t > 0 for t in log)
    assert [w.probe_ms for w in ws] == sorted(w.probe_ms for w in ws) and abs(ws[0].probe_ms - min(log)) < 1e-6
    ptrs = sorted(w.ptr for w in ws)
    assert all(p % (2 << 20) == 0 for p in ptrs) and all(q - p >= 64 << 20 for p, q in zip(ptrs, ptrs[1:]))
    assert all(capi.lib.lsdr_arena_owns(a.h, capi.vp(p)) for p in ptrs)
    # windows hold data like any allocation
    x = np.arange(1 << 20, dtype=np.uint32)
    capi.check(capi.lib.lsdr_memcpy_h2d(ctx.h, ws[1].at(0), x.ctypes.data_as(capi.vp), x.nbytes))
    back = np.empty_like(x)
    capi.check(capi.lib.lsdr_memcpy_d2h(ctx.h, back.ctypes.data_as(capi.vp), ws[1].at(0), x.nbytes))
    ctx.sync()
    assert np.array_equal(x, back)
    # more windows (from the tail where the arena knows nothing better: it tries stretches it has measured fast first): no overlap with what is taken
    tail = a.place(10 << 20, n_best=2, max_windows=4, from_tail=True)
    assert all(t.ptr + (10 << 20) <= p or t.ptr >= p + (64 << 20) for t in tail for p in ptrs) and tail[0].ptr != tail[1].ptr
    # a full arena says so; released windows are handed out again
    with pytest.raises(Exception):
        a.place(900 << 20, n_best=2)
    for w in ws:
        w.free()
    again = a.place(64 << 20, n_best=3, max_windows=3)
    assert len({w.ptr for w in again}) == 3
    assert all(w.ptr + (64 << 20) <= t.ptr or w.ptr >= t.ptr + (12 << 20) for w in again for t in tail)      # (10 MiB windows occupy 12 MiB: 2 MiB granules)
    a.close()


def test_probe_callback_decides_and_errors_surface(capi, ctx):
    a = capi.Arena(ctx, 256 << 20)
    seen = []
    src = ctx.upload(np.full(32 << 20, 7, np.uint8))          # (fill_from must hold `bytes` bytes)

    def probe(w):      # "fast" = the third candidate: its probe queues nothing, the others a big memset
        seen.append(w)
        if len(set(seen)) != 3:
            capi.check(capi.lib.lsdr_memset(ctx.h, capi.vp(w), 0, 32 << 20))
    w, = a.place(32 << 20, n_best=1, max_windows=6, fill_from=src.ptr, probe=probe)
    order = list(dict.fromkeys(seen))
    assert w.ptr == order[2] and len(seen) == 9 * len(order)          # 3 untimed + 6 timed calls per candidate
    assert ctx.download(w, np.uint8, 1 << 20).tobytes() == bytes([7]) * (1 << 20)      # fill_from reached the window the probe left alone

    # the same probe over a buffer the caller already has (an incumbent to compare the arena's windows with)
    other = ctx.alloc(32 << 20)
    n0 = len(seen)
    ms = a.time(other.ptr, lambda p: capi.check(capi.lib.lsdr_memset(ctx.h, capi.vp(p), 0, 32 << 20)) or seen.append(p))
    assert ms > 0 and len(seen) == n0 + 9 and seen[-1] == other.ptr
    other.free()

    def bad(w):
        raise RuntimeError("probe failed")
    with pytest.raises(RuntimeError):
        a.place(32 << 20, probe=bad)
    src.free()
    a.close()


def test_attached_arena_serves_lsdr_malloc(capi, ctx):
    a = capi.Arena(ctx, 512 << 20)
    a.attach()
    big = ctx.alloc(100 << 20)          # ≥ 1 MiB: a window
    small = ctx.alloc(4096)             # small: an ordinary allocation
    assert capi.lib.lsdr_arena_owns(a.h, capi.vp(big.ptr)) and not capi.lib.lsdr_arena_owns(a.h, capi.vp(small.ptr))
    more = [ctx.alloc(100 << 20) for _ in range(5)]      # the arena runs full: the rest are ordinary allocations, no error
    assert sum(bool(capi.lib.lsdr_arena_owns(a.h, capi.vp(m.ptr))) for m in more) == 4          # five windows of 100 MiB fit into 512 MiB
    x = np.arange(1000, dtype=np.float32)
    capi.check(capi.lib.lsdr_memcpy_h2d(ctx.h, big.at(0), x.ctypes.data_as(capi.vp), x.nbytes))
    assert np.array_equal(ctx.download(big, np.float32, 1000), x)
    for b in [big, small] + more:
        b.free()
    a.attach(False)
    assert not capi.lib.lsdr_arena_owns(a.h, capi.vp(ctx.alloc(2 << 20).ptr))
    a.close()
