"""lsdr_capture_batch (BASELINE config 1 / 4): B independent cu8 captures, each from its first sample to TS — leandvb's default `--u8`
graph per capture (auto_notch(1) → cstln_receiver(linear) → deconvol_sync → mpeg_sync → deinterleaver → rs_decoder → derandomizer) — in
shared launches with every data-dependent count on the device.

  * whole job: every capture's TS is the reference BINARY's TS for the same IQ (`leandvb --u8 -f 2400e3 --sr 2000e3 --cr 1/2`, defaults
    otherwise — and the `--anf 0` variant), byte for byte;
  * front end: the packed decisions against the oracle's exact chain cconverter → auto_notch → cstln_receiver (sequential, the
    reference's arithmetic) under leansdr_amd.tolerance, the detected bins EQUAL to the oracle's, with a CW interferer so that the
    notch matters and a lowered detect period so that several detect intervals and a bin change are inside a small capture;
  * FEC tail: given the batch's own packed decisions, the device-resident control flow must produce the BYTES the one-block-per-call
    C ABI produces when the host drives it (bench_c1.Worker's loop): deconvolved bytes, mpeg_sync output, TS — including captures that
    need next_sync() (rotated constellation), lose the lock in the middle (a burst of garbage) and never lock (noise only).
"""
import ctypes as C
import sys

import numpy as np
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)


def _job(capi, n_caps, msamples, groups, tile, warm, anf, seed0=4000):
    import bench_c1
    return bench_c1.C1Job(capi, 0, n_caps, msamples, groups, tile, warm, seed0=seed0, anf=anf)


@pytest.mark.parametrize("anf,tile", [(1, 4096), (1, 2048), (0, 4096)])
def test_every_capture_decodes_to_the_reference_ts(capi, anf, tile):
    job = _job(capi, 3, 6, 2, tile, 512, anf)
    try:
        job.run(1)
        consumed = job.run(2, timed=True)
        chunks = (job.n - 1) // 128
        assert consumed == 2 * 3 * chunks * 128
        v = job.verify()
        assert v["pass"], v
        for c in v["per_capture"]:
            assert c["ts_packets"] > 3000 and c["same_count_every_step"]
            if "whole_ts_identical" in c:                     # the reference binary is on this machine
                assert c["whole_ts_identical"], c
        assert all(r["locked"] == 1 and r["seam_bad"] == 0 for r in job.results)
        if anf:
            # 6 Mi samples: one detect point (block 1023) — the notch is in force for the last third of the capture
            assert len(job.groups[0]["cb"].bins(0)) == 1
        ms, n = job.tile_kernel_ms()
        assert n == 2 * len(job.groups) and ms > 0
    finally:
        job.close()


def _chain_reference(capi, ctx, words, nsym, byte_cap, window):
    """The FEC tail driven by the HOST through the one-block-per-call C ABI (bench_c1.Worker.finish's loop): the checker of the
    device-resident control flow.  `words`: the packed decisions on the device.  Returns (deconvolved bytes, mpeg bytes, TS bytes, stats)."""
    lib = capi.lib
    dec, msync, derand = capi.Deconv(ctx, capi.FEC12), capi.MpegSync(ctx), capi.Derandomizer(ctx)
    pk_cap = byte_cap // 204 + 64
    d_bytes, d_mpeg = ctx.alloc(byte_cap + 64), ctx.alloc(byte_cap + 64)
    d_rs, d_rts, d_ts = ctx.alloc(pk_cap * 204), ctx.alloc(pk_cap * 188), ctx.alloc(pk_cap * 188)
    pos = bw = br = mw = 0
    next_sync = 0
    while True:
        cap = byte_cap - bw if msync.locked else min(window, byte_cap - bw)
        c, p = dec.run_dev_hs2(words, pos, nsym - pos, d_bytes.at(bw), cap)
        if not p:
            break
        pos += c; bw += p
        while True:
            c3, p3, _, _, cns = msync.run_dev(d_bytes.at(br), bw - br, d_mpeg.at(mw), byte_cap - mw)
            if cns:
                dec.next_sync(); next_sync += 1
            if not c3 and not p3:
                break
            br += c3; mw += p3
    cons, prod = C.c_size_t(), C.c_size_t()
    capi.check(lib.lsdr_deinterleaver_run(ctx.h, d_mpeg.ptr, mw, d_rs.ptr, pk_cap, C.byref(cons), C.byref(prod)))
    npk, n_ts, errs = prod.value, 0, 0
    if npk:
        b, e = C.c_long(), C.c_long()
        capi.check(lib.lsdr_rs_decoder_run(ctx.h, d_rs.ptr, npk, d_rts.ptr, C.byref(b), C.byref(e)))
        errs = e.value
        c2, p2 = C.c_size_t(), C.c_size_t()
        capi.check(lib.lsdr_derandomizer_run(derand.h, d_rts.ptr, npk, d_ts.ptr, pk_cap, C.byref(c2), C.byref(p2)))
        n_ts = p2.value
    out = (ctx.download(d_bytes, np.uint8, bw), ctx.download(d_mpeg, np.uint8, mw), ctx.download(d_ts, np.uint8, n_ts * 188),
           dict(next_sync=next_sync, npk=npk, errs=errs, locked=int(msync.locked)))
    for d in (d_bytes, d_mpeg, d_rs, d_rts, d_ts):
        d.free()
    dec.close(); msync.close(); derand.close()
    return out


def _rotate_u8(iq, quarter_turns):
    """(I, Q) → rotated by quarter_turns·90° on the cu8 grid (x ↦ 255 − x stands for the sign flip)."""
    a = iq.reshape(-1, 2).copy()
    for _ in range(quarter_turns % 4):
        a = np.stack([255 - a[:, 1], a[:, 0]], axis=1)
    return np.ascontiguousarray(a).reshape(-1)


@pytest.mark.parametrize("window", [0, 65536])
def test_device_resident_tail_equals_the_host_driven_block_chain(capi, ctx, window):
    import bench_c1
    n = 8 << 20
    gen = bench_c1.Generator(capi, ctx, n, 1)
    d0, _ = gen.capture(0, 777)
    gen.close()
    base = ctx.download(d0, np.uint8, 2 * n)
    d0.free()
    rng = np.random.default_rng(5)
    variants = []
    variants.append(("as generated", base))
    for q in (1, 2, 3):
        variants.append((f"rotated {90 * q} deg (needs next_sync)", _rotate_u8(base, q)))
    burst = base.copy()
    burst[2 * (n // 2): 2 * (n // 2 + 300000)] = rng.integers(100, 156, 600000, dtype=np.uint8)      # the lock drops, then comes back
    variants.append(("garbage burst in the middle", burst))
    variants.append(("noise only (never locks)", rng.integers(96, 160, 2 * n, dtype=np.uint8).astype(np.uint8)))
    short = base[: 2 * 70000].copy()
    bufs = [ctx.upload(v) for _, v in variants]
    cb = capi.CaptureBatch(ctx, len(variants), n, bench_c1.OMEGA, anf=0, tile_len=2048, tile_warmup=512, unlocked_window=window)
    try:
        res, ts = cb.decode([b.ptr for b in bufs], n)
        for i, (name, _) in enumerate(variants):
            r = res[i]
            words_dev = capi.lib.lsdr_capture_batch_words_dev(cb.h, i)
            byte_cap = int(n * 0.94 + 65536) // 8 + 65536
            want_bytes, want_mpeg, want_ts, st = _chain_reference(capi, ctx, words_dev, r["symbols"], byte_cap, cb.unlocked_window)
            got_bytes = cb.stage_bytes(i, "deconv", r["bytes_deconv"])
            got_mpeg = cb.stage_bytes(i, "mpeg", r["bytes_mpeg"])
            assert r["bytes_deconv"] == len(want_bytes) and got_bytes.tobytes() == want_bytes.tobytes(), name
            assert r["bytes_mpeg"] == len(want_mpeg) and got_mpeg.tobytes() == want_mpeg.tobytes(), name
            assert r["next_sync_calls"] == st["next_sync"] and r["rs_packets"] == st["npk"] and r["locked"] == st["locked"], (name, r, st)
            assert r["rs_bit_errors"] == st["errs"], (name, r, st)
            assert ts[i] == want_ts.tobytes(), name
        assert res[0]["ts_packets"] > 4000 and res[0]["next_sync_calls"] == 0
        assert any(res[i]["next_sync_calls"] > 0 and res[i]["ts_packets"] > 300 for i in (1, 2, 3))      # a rotated capture went through next_sync()
        assert 500 < res[4]["ts_packets"] < res[0]["ts_packets"]                                          # the burst cost packets, both halves decoded
        assert res[5]["ts_packets"] == 0 and res[5]["locked"] == 0
        # a SHORT run on the same object (fewer samples than it was created for; fewer than one tile's worth of packets)
        sb = ctx.upload(short)
        res2, ts2 = cb.decode([sb.ptr] * len(variants), len(short) // 2)
        assert all(r["samples"] == (len(short) // 2 - 1) // 128 * 128 for r in res2)
        assert len({t for t in ts2}) == 1                                                                 # the same capture six times: the same TS
        sb.free()
    finally:
        cb.close()
        for b in bufs:
            b.free()


@pytest.mark.parametrize("tile,cw_amp", [(4096, 14.0), (2048, 14.0), (4096, 0.0)])
def test_front_end_against_the_oracle_chain(capi, ctx, oracle, tile, cw_amp):
    """cconverter → auto_notch(1) → cstln_receiver(linear): the batch's packed decisions against the oracle's sequential exact chain.
    (CW amplitude 14 against a signal of RMS 75: until the first detect point nothing notches it, and much more than that throws the
    loops of the reference's receiver itself — its decisions then are noise to compare with.)"""
    import bench_c1
    import pyoracle as po
    from leansdr_amd import tolerance
    n = 1 << 20
    dec_period = 64 * 4096                       # auto_notch::decimation lowered (a public member of the reference's block): 3 detect points in 1 Mi samples
    gen = bench_c1.Generator(capi, ctx, n, 2)
    caps = []
    for k in range(2):
        d, _ = gen.capture(k, 900 + k)
        iq = ctx.download(d, np.uint8, 2 * n)
        d.free()
        if cw_amp:
            # a CW interferer that MOVES after the second detect point: the slot's bin changes, its estimator restarts (sdr.h:99-101)
            t = np.arange(n)
            f = np.where(t < 2 * dec_period + 4096 * 5, 0.1234, -0.31)
            ph = 2 * np.pi * np.cumsum(f)
            x = iq.reshape(-1, 2).astype(np.float64) - 128 + cw_amp * np.stack([np.cos(ph), np.sin(ph)], axis=1)
            iq = np.clip(np.rint(x + 128), 0, 255).astype(np.uint8).reshape(-1)
        caps.append(iq)
    gen.close()
    bufs = [ctx.upload(c) for c in caps]
    cb = capi.CaptureBatch(ctx, 2, n, bench_c1.OMEGA, anf=1, tile_len=tile, tile_warmup=512, notch_decimation=dec_period)
    try:
        cb.run_async([b.ptr for b in bufs], n)
        res = cb.wait()
        for i in range(2):
            xf = oracle.cconverter_u8(caps[i])
            bins_by_block = oracle.auto_notch_bins(xf, decimation=dec_period)
            want_bins = [bins_by_block[b] for b in range(len(bins_by_block)) if (b + 1) * 4096 % dec_period == 0 and (b + 1) * 4096 >= dec_period]
            got_bins = cb.bins(i)
            assert got_bins == want_bins, (got_bins, want_bins)
            if cw_amp:
                assert len(set(got_bins)) >= 2                     # the bin did change inside the capture
            notched, _ = oracle.auto_notch(xf, 1, dec_period)
            o = oracle.rx(po.rx_params(sampler=1, cstln=1, omega=bench_c1.OMEGA, meas_decimation=1 << 20), notched)
            got = cb.words(i, res[i]["symbols"])
            assert res[i]["samples"] == o["consumed"]
            sym = np.zeros(len(got), po.SOFTSYM); sym["symbol"] = got
            ref_sym = o["sym"].copy(); ref_sym["symbol"] &= 3
            sym["cost"] = ref_sym["cost"][: len(sym)] if len(sym) <= len(ref_sym) else 0      # (packed decisions carry no cost)
            rep = tolerance.check_tiled(sym, ref_sym, dict(tiles=res[i]["tiles"], bad_seams=res[i]["seam_bad"], dup=res[i]["seam_dup"], miss=res[i]["seam_miss"]),
                                        first_exact=0)
            assert rep["pass"], rep
            # tile 0 is the reference's arithmetic: its decisions are the oracle's
            first = int(512 / bench_c1.OMEGA) - 8
            assert got[:first].tobytes() == (o["sym"]["symbol"][:first] & 3).tobytes()
    finally:
        cb.close()
        for b in bufs:
            b.free()


@pytest.mark.parametrize("tile", [4096, 2048])
def test_notched_stream_against_the_oracles_auto_notch(capi, ctx, oracle, tile):
    """The samples the tiles see — x − S with S from the estimator pre-pass at every tile start, the interval switches at the detect points,
    S[n] = p·S[n−1] + k·x[n] in between — against auto_notch<f32> (sdr.h:64-138) run sequentially by the oracle, with a STRONG interferer
    that changes its frequency (a bin change: the estimator restarts) and a detect period of 32 blocks.  Bounds: 4e-5 of full scale where
    the bin is below 2048, 1e-3 above.  What is left is the REFERENCE's phasor table — cosf/sinf of (float)(2π·bin·i/4096), an angle of up to
    1834 rad for bin 292, i.e. ±6e-5 rad of table noise times an estimate of amplitude 40: 2e-5 of full scale, the figure measured; the
    exact-phase recurrence here does not reproduce that noise (include/lsdr_hip.h, lsdr_notch_fir's tolerance class)."""
    n = 1 << 20
    dec_period = 32 * 4096
    rng = np.random.default_rng(11)
    t = np.arange(n)
    f = np.where(t < 5 * dec_period + 4096 * 3, 0.0712, np.where(t < 6 * dec_period + 4096 * 7, -0.2203, 0.0712))    # bins ≈ 292, 3194, 292
    ph = 2 * np.pi * np.cumsum(f)
    x = 40.0 * np.stack([np.cos(ph), np.sin(ph)], axis=1) + rng.normal(0, 20.0, (n, 2))
    iq = np.clip(np.rint(x + 128), 0, 255).astype(np.uint8).reshape(-1)
    buf = ctx.upload(iq)
    cb = capi.CaptureBatch(ctx, 1, n, 1.2, anf=1, tile_len=tile, tile_warmup=512, notch_decimation=dec_period)
    try:
        cb.run_async([buf.ptr], n)
        cb.wait()
        xf = oracle.cconverter_u8(iq)
        want, _ = oracle.auto_notch(xf, 1, dec_period)
        bins_by_block = oracle.auto_notch_bins(xf, decimation=dec_period)
        got_bins = cb.bins(0)
        assert got_bins == [bins_by_block[b] for b in range(len(bins_by_block)) if (b + 1) * 4096 % dec_period == 0]
        assert len(set(got_bins)) == 2 and min(got_bins) < 2048 <= max(got_bins)
        got = cb.notched(0, n)
        full = 128.0
        err = np.abs(got - want[:n]) / full
        blk_bin = np.repeat(np.array(bins_by_block), 4096)[:n]
        assert err[:dec_period - 4096].max() == 0.0                      # in front of the first detect point the notch passes its input through
        lo, hi = blk_bin < 2048, blk_bin >= 2048
        assert err[lo].max() <= 4e-5, err[lo].max()        # (measured 2.04e-5)
        assert err[hi].max() <= 1e-3, err[hi].max()
        # … and it notches: the interferer is 20 dB down where an estimator has settled
        settled = slice(4 * dec_period, 5 * dec_period)
        assert np.mean(np.abs(got[settled]) ** 2) < np.mean(np.abs(xf[settled]) ** 2) - 0.9 * 40.0 ** 2
    finally:
        cb.close()
        buf.free()
