"""bench_more.py — the secondary configurations reported next to bench.py's headline (its `more` object).  Every entry is
measured after the headline's timed region, on the same GPU, each with its own clock and (where a kernel dominates) its
own roofline.  N=1 only.

  four_captures  batched streams: FOUR independent captures on the GPU, 64 Mi samples each per batch (the headline's batch split four
                 ways), one fir_filter launch and one set of receiver launches for all of them (rounds 2-4's headline arrangement).
  anf1           config 2 with the reference's default `--anf 1` in front: auto_notch (throughput mode: single-pass scan,
                 detect() on the device) → fir_filter → cstln_receiver; the notch is its own HBM pass (8 B in + 8 B out).
  c2_offset      config 2 with the carrier 1 MHz off: the receiver is biased there (leandvb --tune), its freq_tap feeds
                 fir_filter::track() after every batch (dsp.h:236-244) → shifted, COMPLEX taps (5 packed ops per tap).
  c2_fma         config 2 with the opt-in fused-multiply-add filter arithmetic (LSDR_FIR_FMA; not bit-exact: tolerance).
  c3             config 3: the full DVB-S chain on a framed signal — fir_filter → cstln_receiver → viterbi_sync → mpeg_sync →
                 deinterleaver → rs_decoder → derandomizer — every TS packet checked against what was transmitted.
  c5_rescoped    config 5 as far as the reference implements it: 8PSK + convolutional 2/3 + viterbi_sync at
                 4 samples/symbol ("30 MS/s symbol rate" = 120 MS/s input), TS checked.
  c1             BASELINE config 1 / 4 (bench_c1.py): independent cu8 captures at 1.2 samples/symbol decoded from the first sample to
                 TS, every capture's TS checked against the reference binary's for the same IQ.
  c1_hs          the same input shape on the reference's --hs receiver (fast_qpsk_receiver<u8>, tiled).
  exact_batch    the bit-exact receiver, one GPU lane per independent capture (lsdr_rx_batch: config 4's shape scaled up):
                 65 536 independent cu8 captures at 1.2 samples/symbol, aggregate rate; 64 randomly chosen captures are checked
                 against the oracle bit for bit.
  end_to_end     PCIe-inclusive: the same config-2 chain fed from pinned HOST memory through the copy engine (uploads on a
                 side stream, double-buffered), C2 (cf32, 8 B/sample) and a C1-shaped cu8 stream (2 B/sample).
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MIN_SECONDS = float(os.environ.get("LSDR_BENCH_MIN_SECONDS", 0.6))          # every entry is timed for at least this long (clocks settle, launch overheads amortise)
HBM = 8000.0               # GB/s, MI355X spec peak
REFBIN = os.path.join(ROOT, "oracle", "_ref", "leandvb")


def hbm_frac(samples_per_s, bytes_per_sample):
    """Fraction of the HBM peak on the ALGORITHMIC bytes of SURVEY §8(d)."""
    return round(samples_per_s * bytes_per_sample / 1e9 / HBM, 5)


def batches_for(seconds_per_batch, lo=8):
    return max(lo, int(np.ceil(MIN_SECONDS / max(seconds_per_batch, 1e-6))))


def timed_at_least(run_once, sync):
    """Repeat run_once() (returns the units it processed) until MIN_SECONDS of wall time have been timed.  Returns (units, seconds, calls)."""
    units, calls, t0 = 0, 0, time.perf_counter()
    while True:
        units += run_once()
        sync()
        calls += 1
        dt = time.perf_counter() - t0
        if dt >= MIN_SECONDS:
            return units, dt, calls


def reference_ts(iq, flags, timeout=600):
    """TS bytes the reference binary writes for `iq` (None where oracle/_ref was not built)."""
    if not (os.path.exists(REFBIN) and os.access(REFBIN, os.X_OK)):
        return None
    with tempfile.NamedTemporaryFile(prefix="lsdr_ref_", suffix=".iq", delete=False) as f:
        iq.tofile(f)
    try:
        r = subprocess.run([REFBIN] + flags, stdin=open(f.name, "rb"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout)
        return r.stdout
    finally:
        os.unlink(f.name)


class ReferenceStream:
    """The reference binary decoding `n_samples` of the periodic stream `period` (cf32, repeated) in the BACKGROUND — one host core, fed through
    a pipe while the GPU chain is timed; result() joins and returns its TS bytes (None where oracle/_ref was not built)."""
    def __init__(self, period, n_samples, flags):
        import threading
        self.proc = None
        if not (os.path.exists(REFBIN) and os.access(REFBIN, os.X_OK)):
            return
        self.out = tempfile.NamedTemporaryFile(prefix="lsdr_ref_ts_", suffix=".ts", delete=False)
        self.proc = subprocess.Popen([REFBIN] + flags, stdin=subprocess.PIPE, stdout=self.out, stderr=subprocess.DEVNULL)
        raw = memoryview(np.ascontiguousarray(period).view(np.uint8))
        self.n_samples = int(n_samples)

        def feed():
            left = self.n_samples * 8
            try:
                while left > 0:
                    n = min(left, len(raw))
                    self.proc.stdin.write(raw[:n])
                    left -= n
            except BrokenPipeError:
                pass
            finally:
                try:
                    self.proc.stdin.close()
                except BrokenPipeError:
                    pass
        self.t0 = time.perf_counter()
        self.th = threading.Thread(target=feed, daemon=True)
        self.th.start()

    def result(self, timeout=900):
        if self.proc is None:
            return None, 0.0
        try:
            self.proc.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            self.proc.kill()
        self.th.join(timeout=10)
        dt = time.perf_counter() - self.t0
        self.out.close()
        try:
            with open(self.out.name, "rb") as f:
                return f.read(), dt
        finally:
            os.unlink(self.out.name)


def ts_contains(got_packets, ref_bytes, skip=16, min_packets=24):
    """Every packet the reference wrote after its first `skip` is in `got_packets`, in order and without gaps."""
    rpk = [ref_bytes[i:i + 188] for i in range(0, len(ref_bytes) - 187, 188)]
    tail = rpk[skip:]
    if len(tail) < min_packets or tail[0] not in got_packets:
        return dict(ref_packets=len(rpk), compared=0, equal=False)
    i0 = got_packets.index(tail[0])
    m = min(len(tail), len(got_packets) - i0)
    return dict(ref_packets=len(rpk), compared=m, equal=bool(m >= min_packets and got_packets[i0:i0 + m] == tail[:m]))


def headline_arith(capi, args):
    """The fir_filter arithmetic of the headline (bench.py --fir-arith): the secondary configurations run the same filter."""
    return {"exact": capi.FIR_EXACT, "fma": capi.FIR_FMA, "mfma": capi.FIR_MFMA, "blk": capi.FIR_MFMA_BLK}[getattr(args, "fir_arith", "blk")]


def four_captures(capi, synth, device, args):
    import bench
    # four captures whose batches together are the headline's one (256 Mi samples per GPU and batch)
    # (tiles of 256 samples as in rounds 4-5: a quarter of the headline's samples per capture and launch leaves too few 512-sample tiles: 610 vs 592 GS/s)
    pipe = bench.C2Pipeline(capi, synth, device, 4, max(16, args.batch_msamples * args.captures // 4), args.period_msamples, (256, args.tile_warmup),
                            seed0=77, rx_cus=args.rx_cus, cu_pattern=args.cu_pattern, fir_arith=headline_arith(capi, args))
    t0 = time.perf_counter()
    pipe.run(8, False)
    pipe.sync()
    nb = batches_for((time.perf_counter() - t0) / 8)
    consumed, dt, calls = timed_at_least(lambda: pipe.run(nb, True, snapshot_last=not args.no_verify), pipe.sync)
    nb *= calls
    out = dict(value=round(consumed / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), captures_per_gpu=4, batches=nb,
               roofline=pipe.roofline())
    if not args.no_verify:
        out["verified"] = pipe.verify_last_batch()
        out["pass"] = out["verified"]["pass"]
    pipe.close()
    return out


def c2_variant(capi, synth, device, args, arith, note, n_caps=None):
    """Config 2 with another fir_filter arithmetic than the headline's; verified like the headline (filter output bit for bit
    against the arithmetic's oracle restatement, soft symbols under TOL against the exact chain)."""
    import bench
    pipe = bench.C2Pipeline(capi, synth, device, n_caps or args.captures, args.batch_msamples, args.period_msamples,
                            (args.tile_len, args.tile_warmup), seed0=91, fir_arith=arith)
    t0 = time.perf_counter()
    pipe.run(16, False)
    pipe.sync()
    nb = batches_for((time.perf_counter() - t0) / 16)
    consumed, dt, calls = timed_at_least(lambda: pipe.run(nb, True, snapshot_last=not args.no_verify), pipe.sync)
    nb *= calls
    out = dict(value=round(consumed / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), captures_per_gpu=len(pipe.caps), batches=nb,
               arithmetic=note, roofline=pipe.roofline())
    if not args.no_verify:
        out["verified"] = pipe.verify_last_batch()
        out["pass"] = out["verified"]["pass"]
    pipe.close()
    return out


def c2_rrc(capi, synth, device, args):
    """Config 2 with the polyphase fractional resampler north_star names: leandvb --sampler rrc = fir_sampler on a root-raised-cosine
    of 16 steps per sample and 167 taps (sdr.h:635-689, leandvb.cc:440-458) in place of the linear interpolator; tiled, every capture
    verified against the oracle's exact fir_filter -> exact serial receiver with the same sampler under TOL."""
    import bench
    # (the RRC tiles pace this pipeline.  Rounds 4-5: 256-sample tiles after 512 of warm-up — 3 symbol steps per symbol; with the taps in LDS and the samples in
    # pairs (cstln_receiver.hip, round 6) 343–350 -> 406–414 GS/s, and at 384 / 256 — the warm-up of every other tiled receiver here, 64 symbols; deviations from the
    # serial loop within a third of TOL either way — 442–479)
    tile = (int(os.environ.get("LSDR_RRC_TILE", 384)), int(os.environ.get("LSDR_RRC_WARMUP", max(args.tile_warmup, 256))))
    pipe = bench.C2Pipeline(capi, synth, device, args.captures, args.batch_msamples, args.period_msamples, tile, seed0=61,
                            fir_arith=headline_arith(capi, args), sampler="rrc")
    t0 = time.perf_counter()
    pipe.run(16, False)
    pipe.sync()
    nb = batches_for((time.perf_counter() - t0) / 16)
    consumed, dt, calls = timed_at_least(lambda: pipe.run(nb, True, snapshot_last=not args.no_verify), pipe.sync)
    nb *= calls
    out = dict(value=round(consumed / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), captures_per_gpu=len(pipe.caps), batches=nb,
               sampler=dict(kind="fir_sampler (RRC)", rx_tile=dict(tile_len=tile[0], warmup=tile[1]), **pipe.rrc), roofline=pipe.roofline())
    if not args.no_verify:
        out["verified"] = pipe.verify_last_batch()
        out["pass"] = out["verified"]["pass"]
    pipe.close()
    return out


def c2_cnr(capi, synth, device, args):
    """Config 2 with the two FFT blocks leandvb wires onto the preprocessed stream at a 1 Hz cadence: cnr_fft (carrier-to-noise
    estimate, sdr.h:1273-1345, leandvb.cc:322-329) and spectrum (always in the graph, sdr.h:1347-1404, leandvb.cc:335-343) on capture
    0's decimated stream.  Their outputs are compared with the oracle's blocks run over the same decimated stream (the stream is
    B-periodic, so the last batch's buffer, downloaded, IS every batch) bit for bit."""
    import bench
    st = dict(cnr=[], spec=[], fed=0)
    ctx_m = capi.Ctx(device)
    fs_dec = bench.FS / 30
    dec1hz = int(fs_dec)                                          # decimation(Fs, 1), leandvb.cc:138-141 (after the resampler: Fs/decim)
    cnr = capi.CnrFft(ctx_m, float(np.float32(bench.FM / fs_dec)), 4096, dec1hz, 0.1)
    spec = capi.Spectrum(ctx_m, dec1hz, 0.5)

    def hook(pipe, i, prod, done):
        g = pipe.geo
        ctx_m.wait_event(done)
        vals, c1 = cnr.run_dev(pipe.caps[0].dec[i].ptr, g["n_out"], 0.0, 1.0 / g["decim"])
        rows, c2 = spec.run_dev(pipe.caps[0].dec[i].ptr, g["n_out"])
        assert c1 == g["n_out"] and c2 == g["n_out"]
        st["cnr"].extend(vals.tolist()); st["spec"].extend(rows); st["fed"] += 1

    pipe = bench.C2Pipeline(capi, synth, device, args.captures, args.batch_msamples, args.period_msamples, (args.tile_len, args.tile_warmup),
                            seed0=71, fir_arith=headline_arith(capi, args), batch_hook=hook)
    t0 = time.perf_counter()
    pipe.run(16, False)
    pipe.sync()
    nb = batches_for((time.perf_counter() - t0) / 16)
    consumed, dt, calls = timed_at_least(lambda: pipe.run(nb, True, snapshot_last=not args.no_verify), pipe.sync)
    nb *= calls
    g = pipe.geo
    out = dict(value=round(consumed / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), captures_per_gpu=len(pipe.caps), batches=nb,
               blocks="cnr_fft(bandwidth Fm/Fs, 4096 points) + spectrum(1024 points, kavg 0.5) on capture 0, one transform each per second of signal "
                      f"(every {dec1hz} decimated samples = {dec1hz / g['n_out']:.2f} batches)",
               cnr_values=len(st["cnr"]), spectrum_rows=len(st["spec"]), cnr_db_last=st["cnr"][-1] if st["cnr"] else None, roofline=pipe.roofline())
    if not args.no_verify:
        out["verified"] = pipe.verify_last_batch()
        po = bench._oracle()
        O = po.Oracle()
        y = pipe.ctx.download(pipe.caps[0].dec[pipe.snap[1]], np.complex64, g["n_out"])
        stream = np.tile(y, st["fed"])
        want_c = O.cnr_fft(stream, np.float32(bench.FM / fs_dec), 4096, dec1hz, 0.0, 1.0 / g["decim"])
        want_s = O.spectrum(stream, dec1hz, 0.5)
        got_c, got_s = np.array(st["cnr"], np.float32), np.array(st["spec"], np.float32).reshape(-1, 1024)
        ok_c = len(got_c) == len(want_c) and len(got_c) > 0 and got_c.tobytes() == want_c.tobytes()
        ok_s = got_s.shape == want_s.shape and len(got_s) > 0 and got_s.tobytes() == want_s.tobytes()
        out["fft_blocks_checked"] = dict(cnr_values=int(len(want_c)), cnr_bit_exact=bool(ok_c), spectrum_rows=int(len(want_s)), spectrum_bit_exact=bool(ok_s),
                                         checker="oracle lo_cnr_fft / lo_spectrum over the same decimated stream")
        out["pass"] = bool(out["verified"]["pass"] and ok_c and ok_s)
    pipe.close()
    cnr.close(); spec.close(); ctx_m.close()
    return out


def c2_exact(capi, synth, device, args):
    return c2_variant(capi, synth, device, args, capi.FIR_EXACT,
                      "LSDR_FIR_EXACT: the reference's arithmetic (two roundings per tap, i ascending) on the vector ALUs, k_fir_persist — "
                      "filter output bit-exact against the oracle")


def c2_fma(capi, synth, device, args):
    return c2_variant(capi, synth, device, args, capi.FIR_FMA,
                      "LSDR_FIR_FMA: one fmaf chain per output on the vector ALUs (v_pk_fma_f32) — bit-identical to oracle lo_fir_filter_fma")


def c2_mfma(capi, synth, device, args):
    return c2_variant(capi, synth, device, args, capi.FIR_MFMA,
                      "LSDR_FIR_MFMA: the same fmaf chain as a banded Toeplitz block on v_mfma_f32_16x16x4_f32 (k_fir_mfma) — bit-identical to "
                      "oracle lo_fir_filter_fma and to c2_fma")


def anf1(capi, synth, device, args):
    """The reference's DEFAULT graph (leandvb.cc:103,296-301: auto_notch(1 slot) in front of fir_filter) on the FUSED block
    lsdr_notch_fir: one complex-tap matrix-pipe filter pass over the raw samples + a recurrence at the decimated rate — the notched
    stream never exists (8 B per sample instead of the 24 B of `anf1_scan`).  notch_fir(k+1) runs next to cstln_receiver(k−1).
    Checked: the bin; the filter output from the stream start through the first detect against the oracle's chain
    scaler → auto_notch → fir_filter in the reference's arithmetic (≤ 2e-5 of full scale); later runs against an earlier one at the same
    phase of the periodic signal (the block's state carries over consistently)."""
    import bench
    pipe = bench.C2Pipeline(capi, synth, device, 1, args.batch_msamples, args.period_msamples, (args.tile_len, args.tile_warmup), seed0=33,
                            cw=(0.0137, 3.0), fir_arith=headline_arith(capi, args))
    g, cp = pipe.geo, pipe.caps[0]
    B, period, n_out, N, D, EXTRA = g["B"], g["period"], g["n_out"], g["N"], g["decim"], bench.EXTRA
    ctx = pipe.ctx
    # the endless stream: reps + 2 periods resident, run k reads from stream position F (≡ F mod period) on
    # (the resident capture: the buffer hipMalloc returns or a window of the pipeline's arena — whichever the pipeline's plain filter launch, as a stand-in for
    # the fused block's pass, reads faster into the output ring)
    NB = 8
    ring = ctx.alloc((NB * n_out + EXTRA + 64) * 8)
    d_x = ctx.alloc((B + 2 * period) * 8)
    if getattr(pipe, "arena", None) is not None and os.environ.get("LSDR_BENCH_PLACE_INPUT", "1") != "0":
        try:
            probe = lambda p: pipe.fir.run_dev(p, B + EXTRA * D + N, ring.ptr, n_out + EXTRA)
            t_alloc = pipe.arena.time(d_x.ptr, probe)
            w = pipe.arena.place((B + 2 * period) * 8, n_best=1, max_windows=8, probe=probe)[0]
            if w.probe_ms < t_alloc:
                d_x.free()
                d_x = w
            else:
                w.free()
        except Exception:
            pass
    dp = ctx.upload(cp.x)
    for r in range(g["reps"] + 2):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_x.at(r * period * 8), dp.ptr, period * 8))
    ctx.sync(); dp.free()
    nf = capi.NotchFir(ctx, pipe.coeffs, D, in_scale=75.0)
    if os.environ.get("LSDR_NF_OVERLAP", "1") != "0":      # (the capture is resident: the promise lsdr_notch_fir_set_overlap asks for holds; the block's detect chain
        nf.set_overlap(True)                               # and pass on a stream of their own, its tail beside the next pass: 357 -> 372–377 GS/s)
    # the block's output pipe: a ring of NB batch slots in ONE allocation, so that the head of batch k sits behind batch k − 1 (the receiver's
    # read-ahead) without a copy — except where the ring wraps (one batch in NB: its head is copied behind the last slot)

    class _Slot:
        def __init__(self, j):
            self.ptr = ring.ptr + j * n_out * 8

        def at(self, off):
            return self.ptr + int(off)
    dec = [_Slot(j) for j in range(NB)]
    ev_nf = [ctx.event() for _ in range(NB)]
    ev_rx = [cp.ctx_rx.event() for _ in range(NB)]
    st = dict(F=0, k=0)

    def nf_run(j):
        cons, prod = nf.run_dev(d_x.at((st["F"] % period) * 8), B + 400, dec[j].ptr, n_out)
        st["F"] += cons
        return prod

    # run 0 (the stream start: pass-through, then the first detect at block 1023): B/D − 11 outputs, kept for the check below
    prod0 = nf_run(0)
    assert prod0 == n_out - 11 and st["F"] == B - 330, (prod0, st["F"])
    ctx.sync()
    n_chk = min(prod0, (12 << 20) // D)
    y0 = ctx.download(dec[0], np.complex64, n_chk)

    def run(nb, timed):
        for _ in range(nb):
            st["k"] += 1
            k = st["k"]; j = k % NB; jp = (k - 1) % NB
            if k > NB:
                ctx.wait_event(ev_rx[j])                      # the receiver run that read this buffer (batch k − NB) is done
            prod = nf_run(j)
            assert prod == n_out and st["F"] == (k + 1) * B - 330, (prod, st["F"])
            if k >= 2:       # batch k−1 is complete once the head of batch k sits behind it (the receiver's read-ahead)
                if j == 0:
                    capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, dec[jp].at(n_out * 8), dec[j].ptr, EXTRA * 8))
                ctx.event_record(ev_nf[jp])
                cp.ctx_rx.wait_event(ev_nf[jp])
                used = cp.rx.run_async(dec[jp].ptr, n_out + EXTRA, cp.d_sym.ptr, n_out + EXTRA + 256)
                assert used == n_out
                cp.ctx_rx.event_record(ev_rx[jp])
                cp.queued += 1
                cp.retire(False, keep=2)
        cp.retire(False, keep=0)
        pipe.sync()

    t0 = time.perf_counter()
    run(12, False)
    nb = batches_for((time.perf_counter() - t0) / 12)
    nf.pass_time(True)
    def once():
        run(nb, True)
        return nb
    nb_total, dt, _ = timed_at_least(once, lambda: None)
    kms, klaunches = nf.pass_time(False)
    # checks (after the clock): the bin, the stream start against the oracle chain, two runs one period-multiple apart against each other
    want_bin = int(round(0.0137 * period) / period * 4096 + 0.5) % 4096
    got_bin = nf.bin()
    po = bench._oracle()
    O = po.Oracle()
    n_s = N + n_chk * D
    xs = np.tile(cp.x, -(-n_s // period))[:n_s]
    xn, _ = O.auto_notch(O.scaler(75.0, xs), 1, 1024 * 4096, 0.002, 0.0)
    yr = O.fir_filter(pipe.coeffs, D, xn)[0][:n_chk]
    m = min(len(yr), n_chk)
    err = float(np.abs(y0[:m].astype(np.complex128) - yr[:m]).max() / np.abs(yr).max()) if m else 1.0
    ja, jb = st["k"] % NB, (st["k"] - 2) % NB                  # batches k and k − 2: the same phase of the B-periodic signal
    ya, yb = ctx.download(dec[ja], np.complex64, 1 << 20), ctx.download(dec[jb], np.complex64, 1 << 20)
    drift = float(np.abs(ya - yb).max() / np.abs(ya).max())
    ok = bool(got_bin == want_bin and m > (4096 * 1100) // D and err <= 2e-5 and drift <= 1e-5)
    alg = int(B * bench.ALG_BYTES_PER_SAMPLE_C2)
    out = dict(value=round(nb_total * B / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), batches=nb_total, notch_bin=got_bin, expected_bin=want_bin,
               **{"pass": ok}, interferer="CW at 0.0137 cycles/sample, 3x the signal amplitude",
               block="lsdr_notch_fir (auto_notch(1) fused into fir_filter: k_fir_mfma_stream<30,1,12,IV> + k_nf_scan), then cstln_receiver (tiled)",
               checked=dict(outputs_vs_oracle_chain=int(m), max_rel_err_vs_reference_arithmetic=err, bound=2e-5,
                            run_to_run_same_phase_max_rel_diff=drift, bound_run_to_run=1e-5),
               pipeline_hbm_bytes_per_sample=round(8 + 3 * 8 / D, 3),
               roofline={"kernel": "k_fir_mfma_stream<30,1,12,IV> (notch_fir filter pass, complex taps per detect interval)", "bound": "hbm",
                         "achieved": round(alg / (kms * 1e-3) / 1e9, 2) if kms else None, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (kms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4) if kms else None,
                         "hbm_frac": hbm_frac(nb_total * B / dt, bench.ALG_BYTES_PER_SAMPLE_C2),
                         "avg_launch_ms": round(kms, 4), "launches_timed": klaunches, "algorithmic_bytes_per_launch": alg, "traffic": None,
                         "note": "HIP events around the filter pass on its stream (lsdr_notch_fir_time), the receiver running next to it; the "
                                 "decimated-rate kernels (taps, head, scan, state) add 3 x 8/30 B per sample"})
    nf.close()
    ring.free()
    d_x.free()
    pipe.close()
    return out


def anf1_scan(capi, synth, device, args):
    """One capture: auto_notch(scan) on its own stream → fir_filter → cstln_receiver (queued), three stages in flight (the separate blocks:
    24 B per sample; ≤ 2e-5 of the reference's arithmetic for every bin)."""
    import bench
    pipe = bench.C2Pipeline(capi, synth, device, 1, getattr(args, "more_batch_msamples", 64), args.period_msamples, (args.tile_len, args.tile_warmup), seed0=33,
                            cw=(0.0137, 3.0), fir_arith=headline_arith(capi, args))
    g, cp = pipe.geo, pipe.caps[0]
    ctx_n = capi.Ctx(device)
    notch = capi.AutoNotch(ctx_n, 1, 0.0, mode=capi.NOTCH_SCAN)
    if os.environ.get("LSDR_ANF_OVERLAP"):     # detect chain of batch k+1 ‖ scan of batch k: measured 187 vs 195 GS/s — the pipeline is bound by the
        capi.check(capi.lib.lsdr_auto_notch_set_overlap(notch.h, 1))      # 24 B/sample it moves (4.7 TB/s), not by the chain's 0.1 ms
    assert g["B"] % 4096 == 0
    nblk = g["B"] // 4096 + 2          # the batch plus the two blocks fir_filter's history reaches into
    d_notched = [pipe.ctx.alloc(nblk * 4096 * 8) for _ in range(2)]
    ev_n = [ctx_n.event() for _ in range(2)]
    ev_f = [pipe.ctx.event() for _ in range(2)]
    n_fir_in = g["B"] + bench.EXTRA * g["decim"] + g["N"]
    notch_ms, fir_ms = [], []
    pool = []

    def run(nb, timed):
        for k in range(nb):
            i = k & 1
            if k >= 2:
                ctx_n.wait_event(ev_f[i])                     # fir_filter(k-2) has read this notched buffer
            if timed:
                pool.append((ctx_n.event(), ctx_n.event(), pipe.ctx.event(), pipe.ctx.event()))
                ctx_n.event_record(pool[-1][0])
            notch.run_dev(cp.d_in.ptr, nblk * 4096, d_notched[i].ptr, nblk * 4096)
            if timed:
                ctx_n.event_record(pool[-1][1])
            ctx_n.event_record(ev_n[i])
            pipe.ctx.wait_event(ev_n[i])
            j = pipe.batch_no % g["nbuf"]
            if timed:
                pipe.ctx.event_record(pool[-1][2])
            _, prod = pipe.fir.run_dev(d_notched[i].ptr, n_fir_in, cp.dec[j].ptr, g["n_out"] + bench.EXTRA)
            if timed:
                pipe.ctx.event_record(pool[-1][3])
            pipe.ctx.event_record(ev_f[i])
            pipe.ctx.event_record(pipe.ev_fir[j])
            cp.ctx_rx.wait_event(pipe.ev_fir[j])
            used = cp.rx.run_async(cp.dec[j].ptr, prod, cp.d_sym.ptr, g["n_out"] + bench.EXTRA + 256)
            assert used == g["n_out"]
            cp.queued += 1
            cp.retire(False, keep=2)
            pipe.batch_no += 1
        cp.retire(False, keep=0)
        pipe.sync(); ctx_n.sync()

    t0 = time.perf_counter()
    run(12, False)
    nb = batches_for((time.perf_counter() - t0) / 12)
    notch.scan_time(True)                 # HIP events around the k_notch_scan launches (the last 16 are kept)
    def once():
        run(nb, True)
        return nb
    nb_total, dt, _ = timed_at_least(once, lambda: None)
    nb = nb_total
    kms, klaunches = notch.scan_time(False)
    for e in pool:
        notch_ms.append(ctx_n.event_elapsed_ms(e[0], e[1])); fir_ms.append(pipe.ctx.event_elapsed_ms(e[2], e[3]))
    nms = float(np.mean(notch_ms))
    alg = nblk * 4096 * 16
    n_traffic = n_traffic_src = None
    try:      # recorded, not measured in this run (separate rocprofv3 --pmc passes of the kernel alone, per sample)
        rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r02_bench", "notch_scan_pmc_traffic.json")))
        n_traffic = int(rec["traffic_bytes_per_sample"] * nblk * 4096)
        n_traffic_src = "recorded, not measured in this run: profiles/r02_bench/notch_scan_pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE of the kernel alone, scaled per sample)"
    except Exception:
        pass
    want_bin = int(round(0.0137 * g["period"]) / g["period"] * 4096 + 0.5) % 4096
    out = dict(value=round(nb * g["B"] / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), batches=nb, notch_bin=notch.bins(),
               **{"pass": bool(want_bin in [b % 4096 for b in notch.bins()])}, expected_bin=want_bin,
               interferer="CW at 0.0137 cycles/sample, 3x the signal amplitude", fir_filter_avg_launch_ms=round(float(np.mean(fir_ms)), 4),
               auto_notch_run_avg_ms=round(nms, 4),
               pipeline_hbm_bytes_per_sample=24, pipeline_hbm_frac=round(nb * g["B"] * 24 / dt / 1e9 / bench.HBM_PEAK_GBS, 4),
               roofline={"kernel": "k_notch_scan (auto_notch, 1 slot)", "bound": "hbm", "achieved": round(alg / (kms * 1e-3) / 1e9, 2),
                         "hbm_frac": hbm_frac(nb * g["B"] / dt, bench.ALG_BYTES_PER_SAMPLE_C2),
                         "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (kms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4),
                         "avg_launch_ms": round(kms, 4), "launches_timed": klaunches, "algorithmic_bytes_per_launch": alg,
                         "traffic": n_traffic, "traffic_source": n_traffic_src,
                         "note": "HIP events around the k_notch_scan launch on its stream (lsdr_auto_notch_scan_time), fir_filter and the "
                                 "receiver running next to it; auto_notch_run_avg_ms is the whole run incl. the batched detect FFTs / "
                                 "peak search / table build"})
    notch.close(); ctx_n.close()
    for d in d_notched:
        d.free()
    pipe.close()
    return out


def c2_offset(capi, synth, device, args):   # (complex taps under the headline's arithmetic: k_fir_mfma_stream<CP = 1>)
    """Carrier 1 MHz off: fir_filter runs with complex (frequency-shifted) taps and keeps following the receiver's carrier
    estimate (fir_filter::track, dsp.h:236-244; the scheduler's feedback of leandvb.cc:506-510).  The estimate comes from the
    newest receiver run that has COMPLETED (lsdr_rx_retired_freq_tap) while later ones are still queued: same queued pipeline as
    the headline, feedback latency = queue depth (two batches) instead of a host wait after every batch."""
    import bench
    f0 = 1.0e6 / bench.FS                 # cycles per input sample
    pipe = bench.C2Pipeline(capi, synth, device, 1, args.batch_msamples, args.period_msamples, (args.tile_len, args.tile_warmup),
                            seed0=55, freq=f0, rx_freq=f0 * 30, fir_arith=headline_arith(capi, args))
    g = pipe.geo
    tol = float(np.float32(bench.FM / bench.FS * 0.1))
    t0 = time.perf_counter()
    pipe.run(16, False, track_tol=tol)
    pipe.sync()
    nb = batches_for((time.perf_counter() - t0) / 16)
    # the timed run must exercise the re-shift edge: the filter starts 1.2·tol away from the carrier the receiver reports, so the
    # first track() inside the timed region moves it (dsp.h:236-244)
    pipe.fir.set_freq(float(np.float32(f0 + 1.2 * tol)))
    pipe.reshifts = 0
    consumed, dt, calls = timed_at_least(lambda: pipe.run(nb, True, track_tol=tol), pipe.sync)
    nb *= calls
    cp = pipe.caps[0]
    followed = abs(pipe.fir.current_freq - f0) < tol
    out = dict(value=round(consumed / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), batches=nb, carrier_offset_hz=1.0e6,
               filter_freq=pipe.fir.current_freq, filter_reshifts=int(pipe.reshifts), receiver_freq_tap=cp.rx.state().freq_tap,
               symbols_per_batch=cp.nsym // nb, **{"pass": bool(followed and pipe.reshifts > 0)},
               mode="queued like the headline; freq_tap of the newest completed receiver run -> fir_filter::track() on the host",
               roofline=dict(pipe.roofline(), kernel=pipe.roofline()["kernel"] + ", complex taps", traffic=None, traffic_source=None))
    pipe.close()
    return out


# ---- full chains on framed signals ------------------------------------------------------------------------------------
class DevPipe:
    """A linear device buffer used like a pipebuf between two synchronous C-ABI calls."""

    def __init__(self, capi, ctx, item, cap):
        self.capi, self.ctx, self.item, self.cap = capi, ctx, item, cap
        self.buf = ctx.alloc(cap * item)
        self.rd = self.n = 0

    def room(self, need):
        if self.cap - self.rd - self.n < need and self.rd:
            self.capi.check(self.capi.lib.lsdr_memcpy_d2d(self.ctx.h, self.buf.ptr, self.buf.at(self.rd * self.item), self.n * self.item))
            self.rd = 0
        return self.cap - self.rd - self.n

    def wr(self):
        return self.buf.at((self.rd + self.n) * self.item)

    def rp(self):
        return self.buf.at(self.rd * self.item)

    def push(self, k):
        self.n += k

    def pop(self, k):
        self.rd += k
        self.n -= k
        if self.n == 0:
            self.rd = 0

    def free(self):
        self.buf.free()


def framed_period(capi, ctx, cstln, rate, sps, snr_db, seed, decim=1, groups=1):
    """One period (8·groups TS packets: whole energy-dispersal cycles) of circular DVB-S baseband at `sps`/`decim` samples per
    symbol, unit RMS, from this repo's GPU transmit chain (randomizer → RS → interleaver → convolutional coder → mapper → RRC
    interpolator → decimator); the stream is run until the interleaver and the filters are in steady state and one period is
    cut out.  `groups` makes the period a whole number of samples when `decim` > 1."""
    from leansdr_amd import synth_dvbs
    ts8 = synth_dvbs.ts_packets(8 * groups)
    periods = 10
    tx = capi.TxChain(ctx, interp=sps, decim=decim, amp=1.0, cstln=cstln, rate=rate)
    y = tx.run(np.tile(ts8, (periods, 1)))
    tx.close()
    bits_in, bits_out = C.c_int(), C.c_int()
    capi.check(capi.lib.lsdr_fec_spec(capi.FEC46 if (rate == capi.FEC23 and capi.CSTLN_BITS[cstln] in (2, 6)) else rate,
                                      C.byref(bits_in), C.byref(bits_out), None))
    nsym = 8 * groups * 204 * 8 * bits_out.value // bits_in.value // capi.CSTLN_BITS[cstln]
    assert (nsym * sps) % decim == 0
    P = nsym * sps // decim
    assert len(y) >= 7 * P, (len(y), P)
    a, b = y[5 * P:6 * P], y[6 * P:7 * P]
    assert np.allclose(a, b, atol=1e-4 * np.abs(a).max()), "transmit chain not periodic"
    x = a / np.sqrt(np.mean(np.abs(a) ** 2))
    rng = np.random.default_rng(seed)
    nstd = np.sqrt(0.5 * (sps / decim) / (10 ** (snr_db / 10)))
    x = x + (rng.standard_normal(P) + 1j * rng.standard_normal(P)) * nstd
    x = x / np.sqrt(1 + 2 * nstd ** 2)
    return x.astype(np.complex64), ts8


def full_chain(capi, synth, device, args, cstln, rate, sps, use_fir, batch_msamples, label, ref_flags, ref_packets=0):
    """ref_packets > 0: the reference binary decodes that many TS packets' worth of the SAME (periodic) stream from sample 0 on one host core
    while the GPU chain is timed, and every packet it writes is compared; 0: a prefix of one batch (a few dozen packets)."""
    import bench
    lib = capi.lib
    ctx = capi.Ctx(device)
    x, ts8 = framed_period(capi, ctx, cstln, rate, sps, 20.0 if cstln == capi.QPSK else 24.0, seed=3)
    P = len(x)
    ref_stream = None
    if ref_packets:
        # samples that carry ref_packets packets (204·8 coded bits each over bits-per-symbol × code rate data bits per symbol) + the head the reference spends locking
        spp = 204 * 8 * sps / (capi.CSTLN_BITS[cstln] * {capi.FEC12: 1 / 2, capi.FEC23: 2 / 3}[rate])
        ref_stream = ReferenceStream(x if use_fir else x * np.float32(75.0), int((ref_packets + 600) * spp), ref_flags)
    if use_fir:
        coeffs, decim = bench.c2_filter(capi)
        N = len(coeffs)
        assert sps == int(bench.FS / bench.FM) and P % (128 * decim) == 0
    else:
        decim, N = 1, 0
        assert P % 128 == 0
    reps = max(1, (batch_msamples << 20) // P)
    B = P * reps
    n_out = B // decim
    # (the resident batch: the buffer hipMalloc returns or a window of an lsdr_arena — whichever the filter launch reads faster; one 32 GB
    # hipMalloc lands as it lands.  LSDR_BENCH_PLACE_INPUT=0: the ordinary allocation)
    fir = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=headline_arith(capi, args)) if use_fir else None
    d_decs = [ctx.alloc((n_out + bench.EXTRA) * 8) for _ in range(2)] if use_fir else None
    d_in = ctx.alloc((B + P) * 8)
    c3_arena = None
    if use_fir and os.environ.get("LSDR_BENCH_PLACE_INPUT", "1") != "0" and not os.environ.get("LSDR_RANK_DEVICES"):
        try:
            c3_arena = capi.Arena(ctx, int(os.environ.get("LSDR_BENCH_ARENA_GIB", 160)) << 30)
            probe_in = lambda p: fir.run_dev(p, B + bench.EXTRA * decim + N, d_decs[0].ptr, n_out + bench.EXTRA)      # (speed does not depend on the data)
            t_alloc = c3_arena.time(d_in.ptr, probe_in)
            w = c3_arena.place((B + P) * 8, n_best=1, max_windows=4, probe=probe_in)[0]
            if w.probe_ms < t_alloc:
                d_in.free()
                d_in = w
            else:
                w.free()
        except Exception:
            if c3_arena is not None:
                c3_arena.close()
            c3_arena = None
    dp = ctx.upload(x if use_fir else x * np.float32(75.0))
    for r in range(reps + 1):
        capi.check(lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * P * 8), dp.ptr, P * 8))
    ctx.sync()
    dp.free()
    omega = float(sps / decim)
    rx_kw = dict(sampler=capi.SAMP_LINEAR, cstln=cstln, fec=rate, omega=omega, meas_decimation=1 << 22, pll_adjustment=1 / 6.0)
    if use_fir and c3_arena is not None:
        # the decimated-stream buffers: the two hipMalloc returned, or two arena windows — whichever pair the filter launch over THIS input writes faster (the
        # output side goes with the input window it is paired with, DESIGN §5: 549 against 600–607 GS/s on one box)
        probe = lambda p: fir.run_dev(d_in.ptr, B + bench.EXTRA * decim + N, p, n_out + bench.EXTRA)
        try:
            t_alloc = max(c3_arena.time(d.ptr, probe) for d in d_decs)
            wins = c3_arena.place((n_out + bench.EXTRA) * 8, n_best=2, max_windows=8, from_tail=True, probe=probe)
            if max(w.probe_ms for w in wins) < t_alloc:
                for d in d_decs:
                    d.free()
                d_decs = wins
            else:
                for w in wins:
                    w.free()
        except Exception:
            pass
    d_dec = d_decs[0] if use_fir else None
    # the receiver has its own stream: fir_filter(k+1) runs while cstln_receiver(k) (queued) works on the other decimated buffer
    ctx_rx = capi.Ctx(device)
    ev_fir = [ctx.event() for _ in range(2)]
    rx = capi.CstlnReceiver(ctx_rx, mode=capi.RX_TILED, tile_len=int(os.environ.get("LSDR_CHAIN_TILE", 1024)),
                            tile_warmup=max(args.tile_warmup, 512), **rx_kw)
    # The FEC tail lives on its own context (stream) and its own host thread: every block of it returns data-dependent counts
    # (a host synchronisation per call), so the only way to keep the front end busy meanwhile is a second thread — the
    # C ABI is thread-safe per context and ctypes releases the GIL.  Two symbol buffers go back and forth between the threads.
    ctx_t = capi.Ctx(device)
    vit = capi.Viterbi(ctx_t, cstln, rate)
    msync = capi.MpegSync(ctx_t)
    derand = capi.Derandomizer(ctx_t)
    sym_cap = int(n_out / omega * 1.1) + 4096
    d_stage = [ctx.alloc(sym_cap * 4) for _ in range(2)]
    p_sym = DevPipe(capi, ctx_t, 4, 2 * sym_cap)
    p_bytes = DevPipe(capi, ctx_t, 1, sym_cap)
    p_mpeg = DevPipe(capi, ctx_t, 1, sym_cap)
    pk_cap = sym_cap // 204 + 64
    d_rs = ctx_t.alloc(pk_cap * 204)
    d_rts = ctx_t.alloc(pk_cap * 188)
    d_ts = ctx_t.alloc(pk_cap * 188)
    e0, e1 = ctx.event(), ctx.event()
    fir_ms, ts_out, bits, errs = [], [], [0], [0]

    fir_ev = []

    def front_enqueue(timed, dst, k):
        """fir_filter + (queued) receiver of batch k into the staging buffer `dst`; front_wait() yields the symbol count."""
        if use_fir:
            probe = timed and len(fir_ev) < 64          # HIP events around the first 64 timed filter launches
            if probe:
                fir_ev.append((ctx.event(), ctx.event()))
                ctx.event_record(fir_ev[-1][0])
            _, prod = fir.run_dev(d_in.ptr, B + bench.EXTRA * decim + N, d_decs[k & 1].ptr, n_out + bench.EXTRA)
            if probe:
                ctx.event_record(fir_ev[-1][1])
            ctx.event_record(ev_fir[k & 1])
            ctx_rx.wait_event(ev_fir[k & 1])
            src, n_src = d_decs[k & 1].ptr, prod
        else:
            src, n_src = d_in.ptr, n_out + bench.EXTRA
        used = rx.run_async(src, n_src, dst.ptr, sym_cap)
        assert used == n_out, (used, n_out)

    def front_wait():
        return rx.wait()

    stage_s = {"front": 0.0, "viterbi": 0.0, "mpeg_sync": 0.0, "rest": 0.0}

    def tail(keep, src, n_sym):
        t0 = time.perf_counter()
        p_sym.room(n_sym)
        capi.check(lib.lsdr_memcpy_d2d(ctx_t.h, p_sym.wr(), src.ptr, n_sym * 4))
        p_sym.push(n_sym)
        while True:
            p_bytes.room(p_sym.n // 4 + 256)
            c, p = vit.run_dev(p_sym.rp(), p_sym.n, p_bytes.wr(), p_bytes.room(0))
            if not c and not p:
                break
            p_sym.pop(c); p_bytes.push(p)
        t1 = time.perf_counter()
        while True:
            p_mpeg.room(p_bytes.n + 4096)
            c, p, _, _, _ = msync.run_dev(p_bytes.rp(), p_bytes.n, p_mpeg.wr(), p_mpeg.room(0))
            if not c and not p:
                break
            p_bytes.pop(c); p_mpeg.push(p)
        t2 = time.perf_counter()
        stage_s["viterbi"] += t1 - t0; stage_s["mpeg_sync"] += t2 - t1
        t3 = time.perf_counter()
        cons, prod = C.c_size_t(), C.c_size_t()
        capi.check(lib.lsdr_deinterleaver_run(ctx_t.h, p_mpeg.rp(), p_mpeg.n, d_rs.ptr, pk_cap, C.byref(cons), C.byref(prod)))
        p_mpeg.pop(cons.value)
        npk = prod.value
        n_ts = 0
        if npk:
            b, e = C.c_long(), C.c_long()
            capi.check(lib.lsdr_rs_decoder_run(ctx_t.h, d_rs.ptr, npk, d_rts.ptr, C.byref(b), C.byref(e)))
            bits[0] += b.value; errs[0] += e.value
            c2, p2 = C.c_size_t(), C.c_size_t()
            capi.check(lib.lsdr_derandomizer_run(derand.h, d_rts.ptr, npk, d_ts.ptr, pk_cap, C.byref(c2), C.byref(p2)))
            if keep and p2.value:
                ts_out.append(ctx_t.download(d_ts, np.uint8, p2.value * 188).reshape(-1, 188).copy())
            n_ts = p2.value
        stage_s["rest"] += time.perf_counter() - t3
        return n_ts

    # acquisition (exact serial loop on the head of the stream), then tracking
    acq = capi.CstlnReceiver(ctx, mode=capi.RX_SERIAL, **rx_kw)
    if use_fir:
        _, p0 = fir.run_dev(d_in.ptr, min(B, 1 << 22), d_dec.ptr, n_out)
        ctx.sync()
        acq.run_dev(d_dec.ptr, p0, d_stage[0].ptr, sym_cap, meas=False)
    else:
        acq.run_dev(d_in.ptr, min(n_out, 1 << 18), d_stage[0].ptr, sym_cap, meas=False)
    rx.set_state(acq.state())
    acq.close()

    import queue, threading

    def pipeline(n_batches, timed):
        """front end on this thread, FEC tail on a second one; the two staging buffers circulate through two queues."""
        free_q, full_q = queue.Queue(), queue.Queue()
        for i in range(2):
            free_q.put(i)
        failure = []

        def tail_thread():
            try:
                while True:
                    item = full_q.get()
                    if item is None:
                        return
                    i, n_sym = item
                    tail(timed, d_stage[i], n_sym)
                    ctx_t.sync()
                    free_q.put(i)
            except BaseException as e:      # surfaces in the main thread
                failure.append(e)
                free_q.put(0); free_q.put(1)

        th = threading.Thread(target=tail_thread)
        th.start()
        try:
            pending = None
            for k in range(n_batches):
                i = free_q.get()
                if failure:
                    break
                tf = time.perf_counter()
                front_enqueue(timed, d_stage[i], k)
                if pending is not None:
                    n_sym = front_wait()
                    full_q.put((pending, n_sym))
                pending = i
                stage_s["front"] += time.perf_counter() - tf
            if pending is not None and not failure:
                full_q.put((pending, front_wait()))
        finally:
            full_q.put(None)
            th.join()
        if failure:
            raise failure[0]

    t0 = time.perf_counter()
    pipeline(4, False)
    nb = batches_for((time.perf_counter() - t0) / 4)
    for k in stage_s:
        stage_s[k] = 0.0
    def once():
        pipeline(nb, True)
        return nb
    nb, dt, _ = timed_at_least(once, lambda: (ctx.sync(), ctx_rx.sync(), ctx_t.sync()))
    fir_ms = [ctx.event_elapsed_ms(a, b) for a, b in fir_ev]
    got = np.concatenate(ts_out) if ts_out else np.zeros((0, 188), np.uint8)
    # every packet must be the next one of the transmitted 8-packet cycle
    ok = bad = 0
    if len(got):
        cyc = [bytes(t) for t in ts8]
        first = [k for k in range(8) if cyc[k] == bytes(got[0])]
        ph = first[0] if first else 0
        want = np.tile(ts8, (len(got) // 8 + 2, 1))[ph:ph + len(got)]
        eq = (got == want).all(axis=1)
        ok, bad = int(eq.sum()), int(len(got) - eq.sum())
    # … and the reference's own leandvb binary decodes a prefix of the same IQ (from sample 0) to the same packets
    ref_check = None
    if ref_stream is not None:
        ref, ref_dt = ref_stream.result()
        n_pref = ref_stream.n_samples
        if ref is not None:
            ref_check = ts_contains([bytes(t) for t in got[:ref_packets + 4096]], ref, skip=8, min_packets=min(ref_packets, 16))
            ref_check["reference_seconds_one_core"] = round(ref_dt, 1)
    else:
        n_pref = min(B, (24 << 20) if use_fir else (3 << 20))
        pref = ctx.download(d_in, np.complex64, n_pref)
        ref = reference_ts(pref, ref_flags)
        if ref is not None:
            ref_check = ts_contains([bytes(t) for t in got[:4096]], ref, skip=8, min_packets=16)
    out = dict(value=round(nb * B / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), batches=nb, chain=label, samples_per_symbol=sps,
               symbols_per_s=round(nb * B / sps / dt / 1e6, 3), ts_packets=int(len(got)), ts_packets_per_s=round(len(got) / dt, 1),
               ts_check={"packets_equal_to_the_transmitted_sequence": ok, "different": bad, "pass": bool(bad == 0 and ok > 8),
                         "reference_binary": ("oracle/_ref/leandvb " + " ".join(ref_flags) + f" on the first {n_pref} samples") if ref is not None else None,
                         "reference": ref_check},
               vber=(errs[0] / bits[0] if bits[0] else None), viterbi=vit.stats(), rx_tiles=rx.tiled_stats(),
               host_seconds_per_stage={k: round(v, 4) for k, v in stage_s.items()},
               mode="front end (fir_filter + receiver) and FEC tail on two host threads / two streams, two symbol buffers in "
                    "flight; host_seconds_per_stage are busy times per thread (front | viterbi+mpeg_sync+rest)")
    out["pass"] = bool(out["ts_check"]["pass"] and (ref_check is None or ref_check["equal"]))
    alg_per_sample = 8.0 + 188.0 / (204 * 8 * sps * {capi.QPSK: 1.0, capi.PSK8: 0.5}.get(cstln, 1.0))     # cf32 in + TS out
    # viterbi_sync's kernel (k_viterbi_q4 at these batch sizes) is bound by vector-instruction issue — one wave64 VALU instruction
    # per 4 cycles and SIMD (SQ_ACTIVE_INST_VALU = 1 quad-cycle per instruction, profiles/r03_bench/viterbi_q4.txt): `valu_issue`
    # prices the trellis steps of the CURRENT alignment alone against 1024 SIMDs x 2.4 GHz / (instructions per tile step x 4
    # cycles); the other alignments' resync chunks (3 of them for QPSK, 15 for 8PSK) and the tiles' warm-up come on top.
    instr_per_tile_step = {capi.FEC12: 190 / 16.0, capi.FEC23: 590 / 16.0}.get(rate)
    steps_per_s = nb * B / sps / dt
    valu = None
    if instr_per_tile_step:
        peak_steps = 1024 * 2.4e9 / (instr_per_tile_step * 4.0)
        valu = {"kernel": "k_viterbi_q4 (viterbi_sync, four lanes per tile, sixteen tiles per wavefront)", "instructions_per_tile_step": round(instr_per_tile_step, 2),
                "peak_trellis_steps_per_s": round(peak_steps, 1), "achieved_trellis_steps_per_s": round(steps_per_s, 1), "frac": round(steps_per_s / peak_steps, 4)}
    if use_fir and fir_ms:
        n_launch_out = n_out + bench.EXTRA
        kb = n_launch_out * decim * 8 + n_launch_out * 8
        ms = float(np.mean(fir_ms))
        kname = {capi.FIR_MFMA: "k_fir_mfma", capi.FIR_MFMA_BLK: "k_fir_mfma_stream"}.get(headline_arith(capi, args), "k_fir_persist")
        out["fir_arith"] = getattr(args, "fir_arith", "blk")
        out["roofline"] = {"kernel": kname + " (fir_filter)", "bound": "hbm", "achieved": round(B * alg_per_sample / (ms * 1e-3) / 1e9, 2), "peak": bench.HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(B * alg_per_sample / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4),
                           "algorithmic_bytes_per_launch": int(B * alg_per_sample), "kernel_bytes": kb, "traffic": None,
                           "hbm_frac": hbm_frac(nb * B / dt, alg_per_sample), "viterbi_valu_issue": valu}
    else:
        st = vit.stats()
        out["roofline"] = {"kernel": "k_viterbi_q4 (viterbi_sync, four lanes per tile, sixteen tiles per wavefront)",
                           "bound": "valu issue (64-state add-compare-select per symbol; not an HBM stream)",
                           "trellis_steps_per_s": round(steps_per_s, 1), "valu_issue": valu, "hbm_frac": hbm_frac(nb * B / dt, alg_per_sample),
                           "frac": hbm_frac(nb * B / dt, alg_per_sample), "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                           "achieved": round(nb * B / dt * alg_per_sample / 1e9, 2), "algorithmic_bytes_per_sample": round(alg_per_sample, 3),
                           "viterbi_stats": st}
    for p in (p_sym, p_bytes, p_mpeg):
        p.free()
    for d in (d_in, d_rs, d_rts, d_ts, d_stage[0], d_stage[1]):
        d.free()
    if d_dec:
        d_dec.free()
    vit.close(); msync.close(); derand.close(); rx.close()
    if fir:
        fir.close()
    if use_fir:
        d_decs[1].free()
    if c3_arena is not None:
        c3_arena.close()
    ctx_t.close(); ctx_rx.close(); ctx.close()
    return out


def c3(capi, synth, device, args):
    # viterbi_sync's tiles are long dependent chains (one wavefront walks thousands of trellis steps): a call costs a tile's
    # latency however few tiles there are, and next to fir_filter's persistent workgroups its wavefronts get few slots.  Large
    # batches give a call enough tiles to fill what is left of the chip: 512 Mi samples per batch 227 GS/s, 1 Gi 247, 2 Gi 277,
    # 4 Gi 307 (32 GB of input per batch, resident; 288 GB of HBM is what makes this the natural batch).
    return full_chain(capi, synth, device, args, capi.QPSK, capi.FEC12, 120, True, int(os.environ.get("LSDR_C3_BATCH_MSAMPLES", 4096)),
                      "QPSK 1/2 @ 120 sps cf32: scaler+fir_filter(313,/30) -> cstln_receiver(tiled) -> viterbi_sync -> mpeg_sync -> deinterleaver -> rs_decoder -> derandomizer",
                      ["--f32", "--float-scale", "75", "-f", "240e6", "--sr", "2000e3", "--cr", "1/2", "--resample", "--anf", "0", "--viterbi"],
                      ref_packets=int(os.environ.get("LSDR_C3_REF_PACKETS", 10000)))


def c5_rescoped(capi, synth, device, args):
    # batch: 128 Mi symbols per viterbi_sync call let k_viterbi_q4 keep 32-chunk tiles on every SIMD (64 Mi samples per batch
    # 9.7 GS/s — the lane = state kernel does 11.2 there —, 256 Mi 15.1, 512 Mi 17–20, 1 Gi 14.8)
    return full_chain(capi, synth, device, args, capi.PSK8, capi.FEC23, 4, False, int(os.environ.get("LSDR_C5_BATCH_MSAMPLES", 512)),
                      "8PSK 2/3 @ 4 sps cf32 (30 MS/s symbols = 120 MS/s input): cstln_receiver(PSK8, tiled) -> viterbi_sync(2/3) -> mpeg_sync -> deinterleaver -> rs_decoder -> derandomizer",
                      ["--f32", "--float-scale", "1", "-f", "120e6", "--sr", "30000e3", "--const", "8PSK", "--cr", "2/3", "--anf", "0", "--viterbi"])


def c1(capi, synth, device, args):
    """BASELINE config 1 / 4 on one GPU = `bench.py --workload c1` (also its own line, and with --gpus N the multi-GPU form of
    config 4), run as a process of its own so that its record is exactly what that command prints: lsdr_capture_batch — the
    reference's default graph (--anf 1) for all captures of a GPU in shared launches, counts on the device, one host thread, the
    runtime's default hardware queues."""
    if _C1_EARLY is not None:      # (bench.py ran it before this process touched the GPU: see c1_early)
        return _C1_EARLY
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "c1", "--no-cpu", "--steps", "45", "--warmup", "2"]
    if args.no_verify:
        cmd.append("--no-verify")
    import tempfile
    with tempfile.TemporaryDirectory() as td:       # the child's stdout line is the compact record; its full record goes to a file of ours
        full = os.path.join(td, "c1_full.json")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, env=dict(os.environ, LSDR_BENCH_FULL=full))
        if not os.path.exists(full):
            return dict(value=None, unit="MS/s", **{"pass": False}, error=(r.stderr or r.stdout)[-1500:])
        j = json.load(open(full))
    out = dict(value=j["value"], unit="MS/s", seconds=round(j["steps"] * j["ms_per_step"] / 1e3, 3), steps=j["steps"], ms_per_step=j["ms_per_step"],
               captures=j["config"]["captures_per_gpu"], engine=j["config"].get("engine"), samples_per_capture=j["config"]["samples_per_capture"],
               ts_packets_per_capture=j["config"].get("ts_packets_per_capture"), rs_bit_errors_corrected=j["config"].get("rs_bit_errors_corrected"),
               receiver_seams=j["config"].get("receiver_seams"), config4_one_capture_per_gpu=j.get("config4_one_capture_per_gpu"),
               chain=j["config"]["workload"],
               mode="`bench.py --workload c1 --no-cpu --steps 45` in a process of its own, exit code %d" % r.returncode)
    rl = dict(j["roofline"])
    rl["tile_kernel_avg_launch_ms"] = rl.pop("avg_launch_ms", None)
    rl["achieved"] = round(j["value"] * 1e6 * bench_alg_c1() / 1e9, 2)
    rl["frac"] = rl["hbm_frac"] = hbm_frac(j["value"] * 1e6, bench_alg_c1())
    rl["traffic_source"] = "not collected for this kernel: it is bound by vector-instruction issue (valu_issue below, profiles/r06_bench/c1_pmc_sq.txt), 0.04 of the HBM rate"
    out["roofline"] = rl
    if "verified" in j:
        v = j["verified"]
        out["pass"] = bool(v["pass"]) and r.returncode == 0
        if len(v.get("per_capture", [])) > 2:
            v["per_capture"] = v["per_capture"][:2] + [{"...": f"{len(v['per_capture']) - 2} more, all in `pass`"}]
        out["verified"] = v
    else:
        out["pass"] = r.returncode == 0
    return out


_C1_EARLY = None


def c1_early(args):
    """`bench.py --workload c1` as a process of its own BEFORE the calling process initialises HIP.  Run from a parent that already holds a context on the GPU
    (idle queues and all) the same command comes out 12 % slower — 261–267 GS/s four times in a row against 296–303 eight times standalone, the tile kernel at the same
    5.65 ms: the second process's queues share the hardware scheduler's run list with the parent's.  The record is kept for run_all's `c1` entry."""
    global _C1_EARLY
    try:
        _C1_EARLY = c1(None, None, 0, args)
        _C1_EARLY["mode"] += "; started before the parent process initialised HIP"
    except Exception as e:
        _C1_EARLY = None
        print(f"bench_more.c1_early: {type(e).__name__}: {e}", file=sys.stderr)


def bench_alg_c1():
    import bench_c1
    return bench_c1.ALG_BYTES_PER_SAMPLE


def c1_hs(capi, synth, device, args, hs=True, streams=4):
    """BASELINE config 1's shape (QPSK 1/2, cu8 IQ at 1.2 samples/symbol, 2 B/sample), device-resident, TS checked against the
    transmitted packet sequence.  Default chain (leandvb.cc:205-600 without options): cconverter<u8> → cstln_receiver (linear
    sampler, tiled) → deconvol_sync → mpeg_sync → deinterleaver → rs_decoder → derandomizer; hs: the reference's "maximum
    throughput" receiver: fast_qpsk_receiver<u8> (tiled) → dvb_deconvol_sync<u8> → the same tail.
    `streams` independent decoders (own context = HIP stream, own host thread, each at its own place of the resident capture)
    run at once: every block of the chain returns data-dependent counts, i.e. a host round trip, and one chain alone leaves
    the GPU idle most of the time."""
    import threading
    lib = capi.lib
    ctx = capi.Ctx(device)
    groups = 5
    x, ts = framed_period(capi, ctx, capi.QPSK, capi.FEC12, 6, 20.0, seed=5, decim=5, groups=groups)
    P = len(x)
    u8 = capi.cconv_f32_u8(ctx, x * np.float32(75.0)).reshape(-1)          # leanchansim --ou8: amplitude 75 around 128
    reps = max(1, (64 << 20) // P)
    B = P * reps
    extra = 8192
    d_in = ctx.alloc((B + 2 * P + extra) * 2)
    dp = ctx.upload(u8)
    for r in range(reps + 2):
        capi.check(lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * P * 2), dp.ptr, P * 2))
    capi.check(lib.lsdr_memcpy_d2d(ctx.h, d_in.at((reps + 2) * P * 2), dp.ptr, extra * 2))
    ctx.sync(); dp.free()
    sym_cap = int(B / 1.2 * 1.05) + 65536
    pk_cap = sym_cap // 8 // 204 + 64

    class Chain:
        def __init__(self, w):
            self.w = w
            c = self.ctx = capi.Ctx(device)
            if hs:
                self.rx = capi.FastQpsk(c, 1.2)
                self.dec = capi.HsDeconv(c)
            else:
                self.rx_kw = dict(sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, fec=capi.FEC12, omega=1.2, meas_decimation=1 << 22)
                self.rx = capi.CstlnReceiver(c, mode=capi.RX_SERIAL, **self.rx_kw)
                self.dec = capi.Deconv(c, capi.FEC12)
                self.d_cf = c.alloc((B + extra) * 8)
            self.msync = capi.MpegSync(c)
            self.derand = capi.Derandomizer(c)
            self.p_sym = DevPipe(capi, c, 1 if hs else 4, 2 * sym_cap)
            self.p_bytes = DevPipe(capi, c, 1, sym_cap // 4)
            self.p_mpeg = DevPipe(capi, c, 1, sym_cap // 4)
            self.d_rs, self.d_rts, self.d_ts = c.alloc(pk_cap * 204), c.alloc(pk_cap * 188), c.alloc(pk_cap * 188)
            self.ts_out, self.bits, self.errs = [], 0, 0
            self.pos = (w * (P // max(1, streams))) // 2 * 2 if w else 0     # every decoder at its own place of the stream
            self.stage_s = {"receiver": 0.0, "deconvol": 0.0, "mpeg_sync": 0.0, "rest": 0.0}

        def batch(self, keep, n=None):
            c, rx, dec = self.ctx, self.rx, self.dec
            p_sym, p_bytes, p_mpeg = self.p_sym, self.p_bytes, self.p_mpeg
            t0 = time.perf_counter()
            p_sym.room(sym_cap)
            if hs:
                cns, p = rx.run_dev(d_in.at(self.pos * 2), (n or B) + extra, p_sym.wr(), p_sym.room(0))
            else:
                capi.check(lib.lsdr_cconverter_u8_run(c.h, d_in.at(self.pos * 2), (n or B) + extra, self.d_cf.ptr))
                o = rx.run_dev(self.d_cf.ptr, (n or B) + extra, p_sym.wr(), p_sym.room(0), meas=False)
                cns, p = o["consumed"], o["produced"]
            assert cns > 0
            self.pos = (self.pos + cns) % P
            p_sym.push(p)
            t1 = time.perf_counter()
            while True:
                p_bytes.room(p_sym.n // 8 + 256)
                c2, p2 = dec.run_dev(p_sym.rp(), p_sym.n, p_bytes.wr(), p_bytes.room(0))
                if not p2:
                    break
                p_sym.pop(c2); p_bytes.push(p2)
            t2 = time.perf_counter()
            while True:
                p_mpeg.room(p_bytes.n + 4096)
                c3_, p3, _, _, _ = self.msync.run_dev(p_bytes.rp(), p_bytes.n, p_mpeg.wr(), p_mpeg.room(0))
                if not c3_ and not p3:
                    break
                p_bytes.pop(c3_); p_mpeg.push(p3)
            t3 = time.perf_counter()
            st = self.stage_s
            st["receiver"] += t1 - t0; st["deconvol"] += t2 - t1; st["mpeg_sync"] += t3 - t2
            cons, prod = C.c_size_t(), C.c_size_t()
            capi.check(lib.lsdr_deinterleaver_run(c.h, p_mpeg.rp(), p_mpeg.n, self.d_rs.ptr, pk_cap, C.byref(cons), C.byref(prod)))
            p_mpeg.pop(cons.value)
            npk = prod.value
            if npk:
                b_, e_ = C.c_long(), C.c_long()
                capi.check(lib.lsdr_rs_decoder_run(c.h, self.d_rs.ptr, npk, self.d_rts.ptr, C.byref(b_), C.byref(e_)))
                self.bits += b_.value; self.errs += e_.value
                cc, pp = C.c_size_t(), C.c_size_t()
                capi.check(lib.lsdr_derandomizer_run(self.derand.h, self.d_rts.ptr, npk, self.d_ts.ptr, pk_cap, C.byref(cc), C.byref(pp)))
                if keep and pp.value:
                    self.ts_out.append(c.download(self.d_ts, np.uint8, pp.value * 188).reshape(-1, 188).copy())
            st["rest"] += time.perf_counter() - t3
            return cns

        def track(self):
            """acquisition done by the exact serial loop on the head of the stream: switch to the tiled (throughput) receiver"""
            if hs:
                self.rx.set_tiled(1, int(os.environ.get("LSDR_HS_TILE", 1024)), int(os.environ.get("LSDR_HS_WARM", 512)))   # no unreconciled seams at 512
            else:
                st = self.rx.state()
                self.rx.close()
                self.rx = capi.CstlnReceiver(self.ctx, mode=capi.RX_TILED, tile_len=int(os.environ.get("LSDR_C1_TILE", 1024)),
                                             tile_warmup=max(args.tile_warmup, 512), **self.rx_kw)
                self.rx.set_state(st)

        def close(self):
            for p_ in (self.p_sym, self.p_bytes, self.p_mpeg):
                p_.free()
            for d in (self.d_rs, self.d_rts, self.d_ts) + (() if hs else (self.d_cf,)):
                d.free()
            self.rx.close(); self.dec.close(); self.msync.close(); self.derand.close(); self.ctx.close()

    chains = [Chain(w) for w in range(max(1, streams))]

    def all_chains(fn):
        """fn(chain) on every chain at once (one host thread each: ctypes releases the GIL); returns the results, re-raises"""
        res, err = [None] * len(chains), []

        def run(i):
            try:
                res[i] = fn(chains[i])
            except BaseException as e:
                err.append(e)
        ths = [threading.Thread(target=run, args=(i,)) for i in range(len(chains))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if err:
            raise err[0]
        return res

    def warm(ch):
        ch.batch(False, n=1 << 20)         # acquisition: the exact serial loop on the head of the stream
        ch.track()
        for _ in range(2):
            ch.batch(False)
    all_chains(warm)
    t0 = time.perf_counter()
    all_chains(lambda ch: [ch.batch(False) for _ in range(2)])
    for ch in chains:
        ch.ctx.sync()
    nb = batches_for((time.perf_counter() - t0) / 2)
    for ch in chains:
        ch.ts_out.clear()
        for k in ch.stage_s:
            ch.stage_s[k] = 0.0

    def timed_once():
        return sum(all_chains(lambda ch: sum(ch.batch(True) for _ in range(nb))))

    def sync_all():
        for ch in chains:
            ch.ctx.sync()
    consumed, dt, calls = timed_at_least(timed_once, sync_all)
    nb *= calls
    # every chain's TS against the transmitted sequence; the reference binary on a prefix of the same IQ against chain 0 (from sample 0)
    ok = bad = n_ts = 0
    got0 = None
    for ch in chains:
        got = np.concatenate(ch.ts_out) if ch.ts_out else np.zeros((0, 188), np.uint8)
        if ch.w == 0:
            got0 = got
        n_ts += len(got)
        if len(got):
            first = [k for k in range(len(ts)) if bytes(ts[k]) == bytes(got[0])]
            ph = first[0] if first else 0
            want = np.tile(ts, (len(got) // len(ts) + 2, 1))[ph:ph + len(got)]
            eq = (got == want).all(axis=1)
            ok += int(eq.sum()); bad += int(len(got) - eq.sum())
    pref = ctx.download(d_in, np.uint8, 2 * min(B, 6 << 20))
    ref = reference_ts(pref, ["--u8", "-f", "2400e3", "--sr", "2000e3", "--cr", "1/2", "--anf", "0"] + (["--hs"] if hs else []))
    ref_check = ts_contains([bytes(t) for t in got0[:8192]], ref, skip=16, min_packets=32) if ref is not None else None
    alg = 2.0 + 188.0 / (204 * 8 * 1.2)
    stage_s = {k: round(sum(ch.stage_s[k] for ch in chains), 4) for k in chains[0].stage_s}
    out = dict(value=round(consumed / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), batches=nb * len(chains), decoders=len(chains),
               **{"pass": bool(bad == 0 and ok > 8 and (ref_check is None or ref_check["equal"]))},
               roofline={"kernel": "k_fastqpsk_tiles (fast_qpsk_receiver<u8>, tiled)" if hs else "k_rx_tiles (cstln_receiver, tiled)", "bound": "hbm",
                         "achieved": round(consumed / dt * alg / 1e9, 2),
                         "peak": HBM, "unit": "GB/s", "frac": hbm_frac(consumed / dt, alg), "hbm_frac": hbm_frac(consumed / dt, alg),
                         "algorithmic_bytes_per_sample": round(alg, 4), "traffic": None,
                         "note": "whole-job rate on the algorithmic bytes; each chain is synchronous per batch and latency-bound, "
                                 f"{len(chains)} of them overlap"},
               chain="QPSK 1/2 @ 1.2 sps cu8 (2 B/sample): " + ("fast_qpsk_receiver<u8>(tiled) -> dvb_deconvol_sync<u8>" if hs else
                                                                "cconverter<u8> -> cstln_receiver(tiled) -> deconvol_sync")
                     + " -> mpeg_sync -> deinterleaver -> rs_decoder -> derandomizer",
               symbols_per_s=round(consumed / 1.2 / dt / 1e6, 3), ts_packets=int(n_ts), ts_packets_per_s=round(n_ts / dt, 1),
               ts_check={"packets_equal_to_the_transmitted_sequence": ok, "different": bad, "pass": bool(bad == 0 and ok > 8), "reference": ref_check},
               rs_byte_errors_corrected=sum(ch.errs for ch in chains), rx_tiles=chains[0].rx.tiled_stats(),
               host_seconds_per_stage_summed_over_decoders=stage_s,
               cpu_reference_one_core_MSps=29.7 if hs else 17.6,
               mode=f"{len(chains)} independent decoders (own stream and host thread each, each at its own place of the resident capture); a chain is "
                    "synchronous per batch (every block returns data-dependent counts)")
    for ch in chains:
        ch.close()
    d_in.free(); ctx.close()
    return out


def c1_hs_entry(capi, synth, device, args):
    return c1_hs(capi, synth, device, args, hs=True, streams=int(os.environ.get("LSDR_HS_STREAMS", 8)))      # (8 decoders: 40 GS/s; 4: 35.5; 2: 33 — the tile kernel is bound by its table look-ups per symbol)


def exact_batch(capi, synth, device, args):
    """The bit-exact receiver with one GPU LANE per capture (lsdr_rx_batch) on config 4's input shape: 65 536 independent cu8
    captures at 1.2 samples/symbol, 128 Ki samples each, all resident (17 GB in, 32 GB of soft symbols out).  Every capture is its
    own stretch of a long signal — a different part of the modulated stream (start packet) and a different part of one long noise
    realisation — generated on the device by this repo's transmit / channel blocks.  64 randomly chosen captures are checked
    against the CPU oracle's serial receiver, bit for bit (soft symbols and final loop state)."""
    import bench, bench_c1
    po = bench._oracle()
    lib = capi.lib
    ctx = capi.Ctx(device)
    n_streams, L = int(os.environ.get("LSDR_EXACT_STREAMS", 65536)), int(os.environ.get("LSDR_EXACT_KSAMPLES", 128)) * 1024
    period = 32 << 20                                   # clean baseband period (samples) the captures are cut from
    gen = bench_c1.Generator(capi, ctx, period + 2 * L, 1)
    total = n_streams * L + L
    d_all = ctx.alloc(total * 2 + 64)
    w = C.c_void_p()
    capi.check(lib.lsdr_wgn_create(ctx.h, 1, 4242, C.byref(w)))
    stddev = float(np.float32(10 ** (17.5 / 20)))
    d_tmp = ctx.alloc(period * 8)
    base0 = 40 // 5 * gen.spp5                           # past the modulator's transients
    pos = 0
    while pos < total:                                   # ONE noise stream (the generator's state carries over), period by period
        m = min(period, total - pos)
        capi.check(lib.lsdr_wgn_run(w, stddev, gen.d_base.at(base0 * 8), d_tmp.ptr, m))
        capi.check(lib.lsdr_cconverter_f32_u8_run(ctx.h, d_tmp.ptr, m, d_all.at(pos * 2)))
        pos += m
    ctx.sync()
    lib.lsdr_wgn_destroy(w); d_tmp.free(); gen.close()
    n = L + 1
    cap = (L // 128) * 119 + 256        # the batch form reserves ⌈128/(omega−0.1)⌉+2 symbol slots per chunk
    d_out = ctx.alloc(n_streams * cap * 4)
    kw = dict(sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=bench_c1.OMEGA, in_format=capi.IN_CU8)
    ins = [d_all.at(i * L * 2) for i in range(n_streams)]
    outs = [d_out.at(i * cap * 4) for i in range(n_streams)]
    b = capi.RxBatch(ctx, n_streams, **kw)
    b.run_dev(ins, 128 * 8 + 1, outs, cap)            # warm-up launch (its state is discarded below)
    b.close()
    runs, dts = 0, 0.0
    while dts < MIN_SECONDS:                            # every launch decodes all captures from their first sample (fresh loop states)
        b2 = capi.RxBatch(ctx, n_streams, **kw)
        t0 = time.perf_counter()
        cons, prod = b2.run_dev(ins, n, outs, cap)
        dts += time.perf_counter() - t0
        runs += 1
        if dts < MIN_SECONDS:
            b2.close()
    # 64 random captures against the oracle
    rng = np.random.default_rng(99)
    picks = sorted(set([0, n_streams - 1] + [int(v) for v in rng.integers(0, n_streams, 62)]))
    O = po.Oracle()
    p = po.rx_params(sampler=1, cstln=1, omega=bench_c1.OMEGA)
    good = 0
    for i in picks:
        iq = ctx.download(d_all, np.uint8, 2 * n, byte_offset=i * L * 2)
        ref = O.rx(p, O.cconverter_u8(iq))
        g = ctx.download(d_out, capi.SOFTSYM, prod[i], byte_offset=i * cap * 4)
        st = b2.state(i)
        same_state = all(np.float32(getattr(st, k)).tobytes() == np.float32(getattr(ref["state"], k)).tobytes() for k in ("mu", "phase", "freqw", "agc_gain", "est_insp"))
        good += bool(ref["consumed"] == cons and len(g) == len(ref["sym"]) and g["cost"].tobytes() == ref["sym"]["cost"].tobytes()
                     and g["symbol"].tobytes() == ref["sym"]["symbol"].tobytes() and same_state)
    rate = runs * n_streams * cons / dts
    alg = 2.0 + 4.0 / bench_c1.OMEGA                    # cu8 in + one softsymbol per 1.2 samples out (this block's own algorithmic bytes)
    out = dict(value=round(rate / 1e6, 3), unit="MS/s", seconds=round(dts, 3), launches=runs, captures=n_streams, samples_per_capture=cons,
               per_capture_MSps=round(cons / (dts / runs) / 1e6, 3), symbols_per_launch=int(sum(prod)), input_bytes=int(n_streams * L * 2),
               output_bytes=int(sum(prod)) * 4, captures_checked=len(picks), captures_bit_exact_vs_oracle=good, **{"pass": bool(good == len(picks))},
               roofline={"kernel": "k_rx_batch<linear, cu8> (exact cstln_receiver, one lane per capture)", "bound": "hbm", "peak": HBM, "unit": "GB/s",
                         "achieved": round(rate * alg / 1e9, 2), "frac": hbm_frac(rate, alg), "hbm_frac": hbm_frac(rate, alg),
                         "algorithmic_bytes_per_sample": round(alg, 3), "traffic": None,
                         "note": "the reference's exact per-symbol recurrence: latency-bound by construction (one wavefront per SIMD at 65 536 captures)"},
               note="independent captures (own stretch of signal and noise each), config 4's cu8 / 1.2 sps shape, exact arithmetic per lane; "
                    "64 captures per wavefront")
    b2.close(); d_all.free(); d_out.free(); ctx.close()
    return out


def end_to_end(capi, synth, device, args):
    """Host-resident input: pinned staging buffers, uploads on the context's side stream (lsdr_copy_h2d_async), the compute
    stream waits on the GPU (lsdr_copy_fence) — chunk k+1 crosses PCIe while chunk k is filtered and demodulated."""
    import bench
    lib = capi.lib
    out = {}
    for fmt in ("cf32", "cu8"):
        ctx = capi.Ctx(device)
        coeffs, decim = bench.c2_filter(capi)
        N = len(coeffs)
        item = 8 if fmt == "cf32" else 2
        chunk = 16 << 20                                  # samples per upload
        chunk = chunk // (128 * decim) * (128 * decim)
        n_out = chunk // decim
        x, _ = synth.qpsk_baseband(chunk // 8 // 120 * 120, 120, seed=9, rms=1.0, snr_db=20.0)
        reps = chunk // len(x)
        chunk = reps * len(x); n_out = chunk // decim
        if fmt == "cf32":
            host = np.tile(x, reps + 1)
            fir = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0)
        else:
            xi = np.tile(x, reps + 1) * 40.0 + (128 + 128j)
            host = np.empty((len(xi), 2), np.uint8)
            host[:, 0] = np.clip(xi.real, 0, 255); host[:, 1] = np.clip(xi.imag, 0, 255)
            fir = capi.FirFilter(ctx, coeffs, decim, in_format=capi.IN_CU8)
        nbytes = (chunk + bench.EXTRA * decim + N) * item
        pin = []
        for _ in range(2):
            p = C.c_void_p()
            capi.check(lib.lsdr_malloc_host(nbytes, C.byref(p)))
            C.memmove(p, host.ctypes.data, nbytes)
            pin.append(p)
        d_in = [ctx.alloc(nbytes) for _ in range(2)]
        d_dec = ctx.alloc((n_out + bench.EXTRA) * 8)
        d_sym = ctx.alloc((n_out + bench.EXTRA + 256) * 4)
        rx = capi.CstlnReceiver(ctx, sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=4.0, mode=capi.RX_TILED, tile_len=args.tile_len,
                                tile_warmup=args.tile_warmup)
        nb = 24

        def run(n):
            capi.check(lib.lsdr_copy_h2d_async(ctx.h, d_in[0].ptr, pin[0], nbytes))
            for k in range(n):
                capi.check(lib.lsdr_copy_fence(ctx.h))                               # compute waits for upload k (on the GPU)
                if k + 1 < n:
                    capi.check(lib.lsdr_copy_h2d_async(ctx.h, d_in[(k + 1) & 1].ptr, pin[(k + 1) & 1], nbytes))
                _, prod = fir.run_dev(d_in[k & 1].ptr, chunk + bench.EXTRA * decim + N, d_dec.ptr, n_out + bench.EXTRA)
                rx.run_dev(d_dec.ptr, prod, d_sym.ptr, n_out + bench.EXTRA + 256, meas=False)   # synchronises: buffer k&1 is free again
        t0 = time.perf_counter()
        run(6)
        ctx.sync()
        nb = batches_for((time.perf_counter() - t0) / 6)
        def once():
            run(nb)
            return nb
        nb, dt, _ = timed_at_least(once, ctx.sync)
        # what arrived and what came out of it, after the clock stopped: the last upload's device buffer byte for byte against the host
        # data, the filter's output of the last chunk bit for bit against the oracle (a 200 000-output head), the receiver's seams
        ver = None
        if not args.no_verify:
            po = bench._oracle()
            O = po.Oracle()
            last = (nb - 1) & 1
            got_in = ctx.download(d_in[last], np.uint8, nbytes)
            in_ok = bool(got_in.tobytes() == host.view(np.uint8).reshape(-1)[:nbytes].tobytes())
            nchk = min(200000, n_out)
            y = ctx.download(d_dec, np.complex64, nchk)
            head = host[: nchk * decim + N]
            xin = O.scaler(75.0, head) if fmt == "cf32" else O.cconverter_u8(head.reshape(-1))
            y_ref = O.fir_filter(coeffs, decim, xin)[0][:nchk]
            fir_ok = bool(len(y_ref) == nchk and y_ref.tobytes() == y.tobytes())
            stt = rx.tiled_stats()
            ver = dict(uploaded_bytes_identical=in_ok, fir_outputs_checked=int(nchk), fir_bit_exact=fir_ok, rx_tiles=stt["tiles"], rx_bad_seams=stt["bad_seams"],
                       **{"pass": bool(in_ok and fir_ok and stt["tiles"] > 0 and stt["bad_seams"] == 0)})
        out[fmt] = dict(value=round(nb * chunk / dt / 1e6, 3), unit="MS/s", bytes_per_sample=item, seconds=round(dt, 3), chunks=nb, verified=ver,
                        roofline={"bound": "pcie", "peak": 63.0, "unit": "GB/s", "achieved": round(nb * chunk * item / dt / 1e9, 2),
                                  "frac": round(nb * chunk * item / dt / 1e9 / 63.0, 4), "hbm_frac": hbm_frac(nb * chunk / dt, item + 4.0 / 120),
                                  "note": "host-resident input: bounded by the PCIe Gen5 x16 link (63 GB/s spec), not by HBM"},
                        **{"pass": None if ver is None else ver["pass"]},
                        pcie_GBps=round(nb * chunk * item / dt / 1e9, 2), chunk_samples=chunk)
        rx.close(); fir.close()
        for p in pin:
            lib.lsdr_free_host(p)
        for d in d_in:
            d.free()
        d_dec.free(); d_sym.free()
        ctx.close()
    out["note"] = "host pinned -> HBM on the upload stream, double-buffered against fir_filter + cstln_receiver; bounded by the PCIe link, not by the kernels"
    return out


def run_all(capi, synth, device, args):
    more = {}
    for name, fn in (("four_captures", four_captures), ("c2_exact", c2_exact), ("c2_fma", c2_fma), ("c2_mfma", c2_mfma), ("c2_rrc", c2_rrc),
                     ("c2_cnr", c2_cnr), ("anf1", anf1), ("anf1_scan", anf1_scan),
                     ("c2_offset", c2_offset), ("c3", c3),
                     ("c5_rescoped", c5_rescoped), ("c1", c1), ("c1_hs", c1_hs_entry), ("exact_batch", exact_batch), ("end_to_end", end_to_end)):
        only = os.environ.get("LSDR_BENCH_MORE_ONLY")       # development: a comma-separated subset
        if only and name not in only.split(","):
            continue
        t0 = time.perf_counter()
        try:
            more[name] = fn(capi, synth, device, args)
        except Exception as e:      # a failing extra must not take the headline line down; it is reported as failed
            import traceback
            more[name] = {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-600:]}
        more[name]["bench_seconds"] = round(time.perf_counter() - t0, 1)
    return more
