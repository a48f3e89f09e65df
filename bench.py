#!/usr/bin/env python3
"""bench.py — IQ MSamples/s demodulated on BASELINE config 2 (see BASELINE.json).

Workload at N=1 ("configs[1]"): leandvb DVB-S QPSK 1/2, Fs = 240 MS/s cf32 input at
120 samples/symbol, device-resident synthetic signal:
    scaler(x75, fused) -> fir_filter(N=313, D=30) -> cstln_receiver(omega=4, linear sampler)
One step = one pass of that hot path over one batch of `--batch-msamples` Mi input samples
(the receiver's loop state is carried from step to step, as on an endless stream).
`value` = input IQ samples consumed per second over all ranks (inputs already in HBM).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N>1: independent captures, one per GPU, no data-path collective -> weak scaling.

Extra JSON objects: `roofline` for the dominant kernel (fir_filter; algorithmic bytes =
8 B read per input sample + 8/30 B written) timed with HIP events on the
kernel's own stream, and `cpu_baseline` = the oracle (plain-C port of the reference)
timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, MI355X_MICROARCH.md
FS, FM, ROLLOFF, REJ = 240e6, 2e6, 0.35, 10.0


def c2_filter(capi):
    """Filter design of leandvb.cc:353-378 for Fs=240e6, Fm=2e6 -> order 312, decim 30."""
    decim = int(FS / (FM * 4))
    transition = (FM / 2) * ROLLOFF
    order = int(REJ * FS / (22 * transition))
    order = ((order + 1) // 2) * 2
    fcut = np.float32((FM / 2) * (1 + ROLLOFF / 2) / FS)
    return capi.lowpass(order, fcut), decim


def cpu_baseline(x, coeffs, decim, budget_s):
    """CPU baseline over a bounded sample, single thread: scaler -> fir_filter -> cstln_receiver.  When the real reference
    was built (oracle/_ref/libleansdr_ref.so: the reference's own headers behind oracle/ref_harness.cc) it is what gets
    timed (kind "reference"); otherwise the plain-C oracle (kind "port").  Test infrastructure used as a yardstick only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    use_ref = po.have_ref()
    O = po.Ref() if use_ref else po.Oracle()
    p = po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=1 << 20)
    n_done, t0 = 0, time.perf_counter()
    passes = 0
    while True:
        xs = O.scaler(75.0, x)
        y = O.fir_filter(coeffs, decim, xs)
        y = y[0] if isinstance(y, tuple) else y
        O.rx(p, y)
        n_done += len(x)
        passes += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    lib = "oracle/_ref/libleansdr_ref.so (pabr/leansdr blocks, g++ -O3)" if use_ref else "oracle/liblsdr_oracle.so"
    return dict(value=round(n_done / dt / 1e6, 3), unit="MS/s", cores=1, kind="reference" if use_ref else "port",
                sample=f"{passes} pass(es) over {len(x)} samples of the same workload, {dt:.1f} s, "
                       f"{lib} (scaler+fir_filter+cstln_receiver), 1 thread")


def pmc_traffic(batch_samples):
    """HBM bytes per fir_filter launch from the committed rocprofv3 PMC passes (profiles/r01_bench/pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE corrected ×2 for gfx950 as the microarch guide
    prescribes).  Counters cannot be read from inside a normal run, so this is the recorded per-launch figure for the
    same workload; None when the batch size differs from the profiled one."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_bench", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d["traffic_bytes_per_launch"] if d["batch_samples"] == batch_samples else None
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-msamples", type=int, default=64, help="Mi input samples per step per GPU")
    ap.add_argument("--period-msamples", type=int, default=4, help="unique synthetic period (Mi samples, tiled)")
    ap.add_argument("--rx-mode", choices=["serial", "tiled"], default=os.environ.get("LSDR_BENCH_RX", "tiled"))
    ap.add_argument("--tile-len", type=int, default=128)
    ap.add_argument("--tile-warmup", type=int, default=256)
    ap.add_argument("--no-overlap", action="store_true",
                    help="run fir_filter and cstln_receiver back to back on one stream (default: two HIP streams, "
                         "fir_filter of batch k+1 overlaps cstln_receiver of batch k)")
    ap.add_argument("--captures", type=int, default=6,
                    help="independent captures demodulated concurrently on each GPU (own streams, buffers and block handles; "
                         "tiled receiver with overlapped streams only); a step is then one batch of every capture")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    from leansdr_amd.shard import Shard
    shard = Shard()                       # one process per GPU; torch.distributed (RCCL) only when WORLD_SIZE > 1
    rank, local_rank, world = shard.rank, shard.local_rank, shard.world

    import leansdr_amd.capi as capi
    from leansdr_amd import synth

    barrier = shard.barrier

    ctx = capi.Ctx(local_rank)
    coeffs, decim = c2_filter(capi)
    N = len(coeffs)

    # ---- synthetic input, resident in HBM ----------------------------------
    sps = int(FS / FM)
    period = (args.period_msamples << 20) // sps * sps
    reps = max(1, (args.batch_msamples << 20) // period)
    B = period * reps
    x, _ = synth.qpsk_baseband(period, sps, seed=shard.capture_seed(), rms=1.0, snr_db=20.0)
    d_in = ctx.alloc(B * 8)
    d_per = ctx.upload(x)
    for r in range(reps):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, d_in.at(r * period * 8), d_per.ptr, period * 8))
    ctx.sync()
    d_per.free()
    n_out_max = (B - N) // decim
    d_dec = ctx.alloc(n_out_max * 8)
    d_sym = ctx.alloc((n_out_max + 256) * 4)

    fir = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0)
    rx_kw = dict(sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=float(FS / decim / FM), meas_decimation=int(FS / decim))
    ctx_rx = capi.Ctx(local_rank)          # second HIP stream on the same device
    rx = capi.CstlnReceiver(ctx_rx, mode=capi.RX_TILED if args.rx_mode == "tiled" else capi.RX_SERIAL,
                            tile_len=args.tile_len, tile_warmup=args.tile_warmup, **rx_kw)
    if args.rx_mode == "tiled":
        # Acquisition: the exact serial loop locks on the head of the stream, then the tiled
        # (tracking) receiver takes over from that state.  Not timed (warm-up happens after it).
        acq = capi.CstlnReceiver(ctx, mode=capi.RX_SERIAL, **rx_kw)
        cons0, prod0 = fir.run_dev(d_in.ptr, min(B, 1 << 22), d_dec.ptr, n_out_max)
        ctx.sync()
        acq.run_dev(d_dec.ptr, prod0, d_sym.ptr, n_out_max + 256, meas=False)
        rx.set_state(acq.state())
        acq.close()

    # Two HIP streams: ctx carries fir_filter, ctx_rx carries cstln_receiver.  The decimated
    # stream is double-buffered so that fir_filter(batch k+1) runs while cstln_receiver(batch k)
    # (latency-bound, few wavefronts) is still tracking.  Every batch still goes through both
    # blocks inside the timed region; the pipeline is drained before the clock stops.
    overlap = not args.no_overlap
    d_dec2 = ctx.alloc(n_out_max * 8) if overlap else None
    NBUF = 3 if (overlap and args.rx_mode == "tiled") else 2   # decimated-stream buffers per capture (queued receiver: three)
    d_dec3 = ctx.alloc(n_out_max * 8) if NBUF == 3 else None
    dec = [d_dec, d_dec2] + ([d_dec3] if NBUF == 3 else [])
    ev_fir = [ctx.event() for _ in range(3)]
    e0 = [ctx.event(), ctx.event()]
    e1 = [ctx.event(), ctx.event()]
    ev_rx = [ctx_rx.event() for _ in range(3)]
    ev_pool = []
    fir_ms = []
    nsym = [0]
    dbg = []

    class Lane:
        """One capture: input in HBM, fir_filter on its own stream, cstln_receiver on another, double-buffered decimated
        stream.  Lane 0 wraps the objects created above; further lanes (--captures) get their own of everything."""

        def __init__(self, idx):
            self.idx = idx
            if idx == 0:
                self.ctx, self.ctx_rx, self.d_in, self.dec, self.d_sym, self.fir, self.rx = ctx, ctx_rx, d_in, dec, d_sym, fir, rx
                self.ev_fir, self.ev_rx = ev_fir, ev_rx
            else:
                # fir_filter of every capture goes through ONE stream (two persistent fir kernels side by side only fight for the
                # LDS of the same CUs); every capture's receiver has its own stream
                self.ctx, self.ctx_rx = ctx, capi.Ctx(local_rank)
                xs, _ = synth.qpsk_baseband(period, sps, seed=shard.capture_seed() + 1000 * idx, rms=1.0, snr_db=20.0)
                self.d_in = self.ctx.alloc(B * 8)
                dp = self.ctx.upload(xs)
                for r in range(reps):
                    capi.check(capi.lib.lsdr_memcpy_d2d(self.ctx.h, self.d_in.at(r * period * 8), dp.ptr, period * 8))
                self.ctx.sync()
                dp.free()
                self.dec = [self.ctx.alloc(n_out_max * 8) for _ in range(NBUF)]
                self.d_sym = self.ctx.alloc((n_out_max + 256) * 4)
                self.fir = fir                                          # one filter object: every capture goes through the same launch
                self.rx = capi.CstlnReceiver(self.ctx_rx, mode=capi.RX_TILED, tile_len=args.tile_len, tile_warmup=args.tile_warmup, **rx_kw)
                a = capi.CstlnReceiver(self.ctx, mode=capi.RX_SERIAL, **rx_kw)     # acquisition on the head of this capture
                _, p0 = self.fir.run_dev(self.d_in.ptr, min(B, 1 << 22), self.dec[0].ptr, n_out_max)
                self.ctx.sync()
                a.run_dev(self.dec[0].ptr, p0, self.d_sym.ptr, n_out_max + 256, meas=False)
                self.rx.set_state(a.state())
                a.close()
                self.ev_fir = [self.ctx.event() for _ in range(3)]
                self.ev_rx = [self.ctx_rx.event() for _ in range(3)]
            self.queued = []

        def retire(self, timed, keep=2):
            while len(self.queued) > keep:
                self.queued.pop(0)
                nprod = self.rx.wait()
                if timed:
                    nsym[0] += nprod

        def close(self):
            if self.idx:
                self.rx.close()
                self.d_in.free(); self.d_sym.free()
                for d in self.dec:
                    d.free()
                self.ctx_rx.close()

    n_captures = args.captures if (overlap and args.rx_mode == "tiled") else 1
    lanes = [Lane(c) for c in range(n_captures)]

    def run_steps(k_steps, timed):
        consumed = 0
        if not overlap:
            for _ in range(k_steps):
                ctx.event_record(e0[0])
                cons, prod = fir.run_dev(d_in.ptr, B, d_dec.ptr, n_out_max)
                ctx.event_record(e1[0])
                ctx.event_record(ev_fir[0])
                rx.ctx.wait_event(ev_fir[0])
                o = rx.run_dev(d_dec.ptr, prod, d_sym.ptr, n_out_max + 256, meas=False)
                if timed:
                    fir_ms.append(ctx.event_elapsed_ms(e0[0], e1[0]))
                    nsym[0] += o["produced"]
                consumed += cons
            return consumed
        if args.rx_mode == "tiled":
            # Queued receiver runs (lsdr_rx_run_async): the host only enqueues.  Per step: fir_filter(k) on the fir
            # stream (after the receiver has released that decimated buffer), cstln_receiver(k) on the rx stream
            # (after fir_filter(k)); results are retired two steps later, so the GPU never waits for the host.
            while len(ev_pool) < 2 * k_steps:
                ev_pool.append(ctx.event())
            for k in range(k_steps):
                _t_step = time.perf_counter()
                i = k % NBUF
                # dec[i] of every capture is free once its receiver run k-NBUF has read it.  With three buffers that run has
                # already been retired on the host (at most two runs stay queued), so no GPU-side wait is needed — and none is
                # wanted: an event recorded on a receiver stream sits behind whatever else shares its hardware queue.
                if k >= NBUF and NBUF < 3:
                    for ln in lanes:
                        ctx.wait_event(ln.ev_rx[i])
                ctx.event_record(ev_pool[2 * k])
                if len(lanes) == 1:
                    cons, prod = fir.run_dev(d_in.ptr, B, dec[i].ptr, n_out_max)
                else:                                            # one launch filters batch k of every capture
                    cons, prod = fir.run_multi_dev([ln.d_in.ptr for ln in lanes], B, [ln.dec[i].ptr for ln in lanes], n_out_max)
                ctx.event_record(ev_pool[2 * k + 1])
                ctx.event_record(ev_fir[i])
                consumed += cons * len(lanes)
                for ln in lanes:
                    ln.rx.ctx.wait_event(ev_fir[i])
                    ln.rx.run_async(ln.dec[i].ptr, prod, ln.d_sym.ptr, n_out_max + 256)
                    ln.rx.ctx.event_record(ln.ev_rx[i])
                    ln.queued.append(i)
                _t_enq = time.perf_counter()
                for ln in lanes:
                    ln.retire(timed)
                if timed and os.environ.get("LSDR_BENCH_DEBUG"):
                    _t_ret = time.perf_counter()
                    dbg.append((_t_enq - _t_step, _t_ret - _t_enq))
            for ln in lanes:
                ln.retire(timed, keep=0)
            if timed:
                ctx.sync()
                for k in range(k_steps):       # HIP events around every fir_filter launch, on its own stream
                    fir_ms.append(ctx.event_elapsed_ms(ev_pool[2 * k], ev_pool[2 * k + 1]))
            return consumed
        pending = None                      # (buffer index, produced) of the batch waiting for the receiver
        for k in range(k_steps + 1):
            cur = None
            if k < k_steps:
                i = k & 1
                ctx.event_record(e0[i])
                cons, prod = fir.run_dev(d_in.ptr, B, dec[i].ptr, n_out_max)   # async on the fir stream
                ctx.event_record(e1[i])
                ctx.event_record(ev_fir[i])
                consumed += cons
                cur = (i, prod)
            if pending is not None:
                i, prod = pending
                rx.ctx.wait_event(ev_fir[i])
                o = rx.run_dev(dec[i].ptr, prod, d_sym.ptr, n_out_max + 256, meas=False)   # syncs the rx stream
                if timed:
                    fir_ms.append(ctx.event_elapsed_ms(e0[i], e1[i]))
                    nsym[0] += o["produced"]
            pending = cur
        return consumed

    def sync_all():
        for ln in lanes:
            ln.ctx.sync(); ln.rx.ctx.sync()

    run_steps(args.warmup, False)
    sync_all()
    barrier()
    t0 = time.perf_counter()
    consumed = run_steps(args.steps, True)
    sync_all()
    barrier()
    dt = time.perf_counter() - t0

    if dbg:
        print("host per step: enqueue %.1f us, blocked in retire %.1f us" % (1e6 * np.mean([d[0] for d in dbg]), 1e6 * np.mean([d[1] for d in dbg])), file=sys.stderr)
    total, dt, _ = shard.aggregate(consumed, dt)   # all ranks' samples ÷ the slowest rank's time

    if rank == 0:
        per_launch_samples = (n_out_max * decim) * n_captures           # input samples one fir launch processes (all captures)
        alg_bytes = per_launch_samples * 8 + n_out_max * n_captures * 8  # cf32 in + cf32 out
        fir_avg_ms = float(np.mean(fir_ms))
        achieved = alg_bytes / (fir_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "IQ MSamples/s demodulated (leandvb QPSK 1/2)",
            "value": round(total / dt / 1e6, 3),
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: QPSK 1/2, Fs 240 MS/s cf32 (120 sps), device-resident; "
                                   "scaler(x75 fused) + fir_filter(N=313,D=30) + cstln_receiver(omega 4, linear sampler)",
                       "batch_samples_per_gpu": B * n_captures, "captures_per_gpu": n_captures, "rx_mode": args.rx_mode,
                       "streams": ("fir_filter(k+1) of all captures in one launch (lsdr_fir_filter_run_multi) || cstln_receiver(k), one HIP stream per capture, "
                                   "receiver runs queued (lsdr_rx_run_async)") if overlap else "single stream",
                       "rx_tile": {"tile_len": args.tile_len, "warmup": args.tile_warmup},
                       "rx_tiles": rx.tiled_stats() if args.rx_mode == "tiled" else None,
                       "parallelism": f"{world * n_captures} independent capture(s), {n_captures} per GPU, no collectives",
                       "symbols_per_step": nsym[0] // max(1, args.steps)},
            "roofline": {"kernel": "k_fir (fir_filter)", "bound": "hbm", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(B * n_captures), "avg_launch_ms": round(fir_avg_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu and world == 1:   # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(x, coeffs, decim, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    for ln in lanes:
        ln.close()
    fir.close(); rx.close(); ctx_rx.close()
    d_in.free(); d_dec.free(); d_sym.free()
    if d_dec2 is not None:
        d_dec2.free()
    if d_dec3 is not None:
        d_dec3.free()
    ctx.close()
    shard.close()


if __name__ == "__main__":
    main()
