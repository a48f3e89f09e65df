#!/usr/bin/env python3
"""bench.py — IQ MSamples/s demodulated on BASELINE config 2 (see BASELINE.json).

Workload at N=1 ("configs[1]"): leandvb DVB-S QPSK 1/2, Fs = 240 MS/s cf32 input at
120 samples/symbol, device-resident synthetic signal:
    scaler(x75, fused) -> fir_filter(N=313, D=30) -> cstln_receiver(omega=4, linear sampler)

The input of every capture is ONE ENDLESS STREAM: the synthetic signal is circular with period B (one batch), every
batch consumes exactly B samples (fir_filter keeps its N-sample history by reading N + 128·D samples past the batch end,
the receiver consumes exactly B/D decimated samples and carries its loop state on the device), so batch k+1 continues
where batch k stopped — no restart, no dropped samples, no timing jump at batch boundaries.

A *step* is `--batches-per-step` batches (default 96 ≈ 50 ms of GPU work) of every capture, so that the driver's
`--steps 20` times ≈ 1 s and the clocks settle.  `value` = input IQ samples consumed per second over all ranks (inputs
already in HBM).

After the clock stops the LAST batch of the timed region is verified against the CPU oracle (`verified` in the JSON
line): the fir_filter output bit for bit, the soft symbols against the oracle's serial receiver started from the very
loop state the device used for that batch (snapshot taken in stream order), under the tolerance of the tiled mode
(tests/test_gpu_rx_tiled.py).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
`--gpus N` without a launcher starts the N ranks itself.  N>1: independent captures, one process per GPU, no data-path
collective, no RCCL (gloo carries the barrier and three scalars) -> weak scaling.

Extra JSON objects: `roofline` for the dominant kernel (fir_filter; algorithmic bytes = 8 B read per input sample +
8/30 B written) timed with HIP events on the kernel's own stream, `cpu_baseline` = the reference's blocks
(oracle/_ref, or the plain-C oracle) timed on this host's cores on a bounded sample of the same workload, and `more` =
the other configurations (single stream, carrier offset, full chain ...), each measured after the headline region.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, MI355X_MICROARCH.md
FS, FM, ROLLOFF, REJ = 240e6, 2e6, 0.35, 10.0
EXTRA = 128                # decimated samples fir_filter produces past the batch end (receiver read-ahead + less than a chunk)
EXTRA_RRC = 384            # ... with the RRC fir_sampler (read-ahead 166): the receiver is handed n_out + 256 of them

from leansdr_amd.tolerance import TOL, check_tiled      # THE tolerance of the tiled receiver (stated once, shared with tests/)

DEFAULT_TILE = (512, 256)  # receiver tile geometry of the headline (tile_len, warm-up), samples of the decimated stream: 64 symbols of warm-up as in
                           # rounds 2-5, tiles of 128 symbols instead of 64 (half the seams, 3/4 of the receiver's symbol steps: 669-672 -> 689-695 GS/s over six
                           # processes each, profiles/r06_bench/headline_repeat_tiles.txt; longer tiles make the receiver the pacer: 640+ samples 590-620)
ALG_BYTES_PER_SAMPLE_C2 = 8.0 + 4.0 / 120.0      # cf32 in + one softsymbol per 120 samples = 8.03


def c2_design():
    """Filter design of leandvb.cc:353-378 for Fs=240e6, Fm=2e6 -> order 312, cut-off, decimation 30."""
    decim = int(FS / (FM * 4))
    transition = (FM / 2) * ROLLOFF
    order = int(REJ * FS / (22 * transition))
    order = ((order + 1) // 2) * 2
    fcut = np.float32((FM / 2) * (1 + ROLLOFF / 2) / FS)
    return order, fcut, decim


def c2_filter(capi):
    order, fcut, decim = c2_design()
    return capi.lowpass(order, fcut), decim


def c2_geometry(batch_msamples, period_msamples, ncoeffs, decim, extra=EXTRA):
    """A batch = whole symbols and whole receiver chunks (so that every batch consumes exactly B samples)."""
    sps = int(FS / FM)
    unit = int(128 * decim * sps // np.gcd(128 * decim, sps))
    period = max(1, (period_msamples << 20) // unit) * unit
    reps = max(1, (batch_msamples << 20) // period)
    B = period * reps
    assert B % decim == 0 and (B // decim) % 128 == 0 and extra * decim + ncoeffs <= period
    return dict(period=period, reps=reps, B=B, n_out=B // decim, N=ncoeffs, decim=decim, sps=sps, nbuf=3)


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    return po


_CPU_JOB = {}     # inherited by the forked workers (no 32 MB pickle per worker)


def _cpu_worker(seconds):
    x, coeffs, decim, use_ref = _CPU_JOB["x"], _CPU_JOB["coeffs"], _CPU_JOB["decim"], _CPU_JOB["use_ref"]
    po = _oracle()
    O = po.Ref() if use_ref else po.Oracle()
    p = po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=1 << 20)
    n_done, t0 = 0, time.perf_counter()
    while True:
        xs = O.scaler(75.0, x)
        y = O.fir_filter(coeffs, decim, xs)
        y = y[0] if isinstance(y, tuple) else y
        O.rx(p, y)
        n_done += len(x)
        if time.perf_counter() - t0 >= seconds:
            break
    return n_done, time.perf_counter() - t0


def cpu_baseline(x, coeffs, decim, budget_s):
    """CPU baseline over a bounded sample: scaler -> fir_filter -> cstln_receiver on independent streams, one PROCESS per
    host core (the reference is single-threaded by design; `nproc` processes is how it scales).  When the real reference
    was built (oracle/_ref/libleansdr_ref.so: the reference's own headers behind oracle/ref_harness.cc) it is what gets
    timed (kind "reference"); otherwise the plain-C oracle (kind "port").  Test infrastructure used as a yardstick only.
    Runs BEFORE the process touches the GPU (the workers are forked)."""
    import multiprocessing as mp
    po = _oracle()
    use_ref = po.have_ref()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ctx = mp.get_context("fork")
    _CPU_JOB.update(x=x, coeffs=coeffs, decim=decim, use_ref=use_ref)
    one_n, one_t = _cpu_worker(budget_s * 0.3)
    one = one_n / one_t / 1e6
    passes = []
    if cores > 1:
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            for _ in range(3):          # the all-core figure swings from pass to pass (memory-bound): median of three
                res = pool.map(_cpu_worker, [budget_s * 0.7 / 3] * cores, chunksize=1)
                passes.append(sum(r[0] for r in res) / max(r[1] for r in res) / 1e6)
        wall = time.perf_counter() - t0
        allc = sorted(passes)[1]
    else:
        allc, wall = one, 0.0
    lib = "oracle/_ref/libleansdr_ref.so (pabr/leansdr blocks, g++ -O3)" if use_ref else "oracle/liblsdr_oracle.so"
    return dict(value=round(allc, 3), unit="MS/s", cores=cores, kind="reference" if use_ref else "port",
                one_core=round(one, 3), all_core_passes=[round(v, 1) for v in passes],
                sample=f"{lib}: scaler+fir_filter+cstln_receiver over {len(x)}-sample passes of the same workload; "
                       f"1 process for {one_t:.1f} s, then {cores} independent streams (one process per core), median of 3 passes of "
                       f"{budget_s * 0.7 / 3:.1f} s ({wall:.1f} s incl. start-up)")


def pmc_traffic(kernel, batch_samples):
    """HBM bytes per launch of `kernel`, RECORDED (not measured in this run): the committed rocprofv3 PMC passes under
    profiles/ (FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE corrected ×2 for gfx950 as the
    microarch guide prescribes).  Counters cannot be read from inside a normal run.  A record counts only when it was taken
    on the SAME kernel (by name) at the SAME launch size; otherwise None."""
    for d in ("r06_bench", "r05_bench", "r04_bench", "r03_bench", "r02_bench", "r01_bench"):
        try:
            with open(os.path.join(ROOT, "profiles", d, "pmc_traffic.json")) as f:
                j = json.load(f)
            if j["batch_samples"] == batch_samples and re.search(r"\b%s\b" % re.escape(kernel), j["kernel"]):
                return j["traffic_bytes_per_launch"], f"profiles/{d}/pmc_traffic.json"
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def measured_copy_ceiling():
    """The practical HBM ceiling SURVEY §8(d) asks for next to the spec peak: a plain streaming-read kernel's rate on this chip (the
    filter reads 30 bytes for every byte it writes), RECORDED from the newest profiles/**/membench*.txt (tools/membench.hip; not measured in this run)."""
    best = None
    for pat in ("profiles/r05_bench/membench.txt", "profiles/membench_r01.txt"):
        try:
            txt = open(os.path.join(ROOT, pat)).read()
        except OSError:
            continue
        vals = [float(v) for v in re.findall(r"([0-9]+(?:\.[0-9]+)?)\s*GB/s", txt)]
        if vals:
            best = (max(vals), pat)
            break
    return best


def resolve_defaults(args):
    """The C2 geometry that was not given: 256 Mi samples per batch and GPU shared by its captures, 24 Gi samples per step and GPU —
    one capture: 96 batches of 256 Mi; four captures: 96 batches of 4 × 64 Mi."""
    caps = max(1, args.captures)
    if args.batch_msamples is None:
        args.batch_msamples = max(16, 256 // caps)
    if args.batches_per_step is None:
        args.batches_per_step = max(1, 96 * 256 // (args.batch_msamples * caps))
    args.more_batch_msamples = 64         # the one-capture secondary configurations keep round 3's batch (comparable numbers)
    return args


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks (torch.distributed.run, rendezvous on 127.0.0.1)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    return subprocess.call(cmd, env=env)


class Capture:
    """One capture: its endless input in HBM, the decimated-stream buffers, the symbol buffer, its receiver (own HIP
    stream).  fir_filter of all captures of a GPU goes through ONE launch on the fir stream."""

    def __init__(self, capi, synth, device, fir_ctx, idx, seed, geo, rx_kw, tile, freq=0.0, rx_cus=None, cw=None, shared_rx_ctx=None, extra=EXTRA):
        self.capi, self.idx, self.geo = capi, idx, geo
        EXTRA = extra
        period, reps, B, n_out, N, decim = geo["period"], geo["reps"], geo["B"], geo["n_out"], geo["N"], geo["decim"]
        self.x, _ = synth.qpsk_baseband(period, geo["sps"], seed=seed, rms=1.0, snr_db=20.0, freq=freq)
        if cw:          # a CW interferer (cycles/sample rounded to a whole number of cycles per period, amplitude): auto_notch's job
            f = round(cw[0] * period) / period
            self.x = (self.x + np.float32(cw[1]) * np.exp(2j * np.pi * f * np.arange(period))).astype(np.complex64)
        self.ctx = fir_ctx
        self.own_rx_ctx = shared_rx_ctx is None
        self.ctx_rx = capi.Ctx(device, cu_mask=rx_cus) if shared_rx_ctx is None else shared_rx_ctx
        self.d_in2 = None       # a second copy of the resident capture, read by the odd batches (C2Pipeline.place_buffers decides)
        self.d_in = self.ctx.alloc((B + period) * 8)
        dp = self.ctx.upload(self.x)
        for r in range(reps + 1):
            capi.check(capi.lib.lsdr_memcpy_d2d(self.ctx.h, self.d_in.at(r * period * 8), dp.ptr, period * 8))
        self.ctx.sync()
        dp.free()
        self.dec = [self.ctx.alloc((n_out + EXTRA) * 8) for _ in range(geo["nbuf"])]
        self.d_sym = self.ctx.alloc((n_out + EXTRA + 256) * 4)
        if os.environ.get("LSDR_BENCH_PTRS"):       # diagnostic: where the buffers landed
            print("ptrs d_in %x dec %s d_sym %x" % (self.d_in.ptr, " ".join("%x" % d.ptr for d in self.dec), self.d_sym.ptr), file=sys.stderr)
        self.rx = capi.CstlnReceiver(self.ctx_rx, mode=capi.RX_TILED, tile_len=tile[0], tile_warmup=tile[1], **rx_kw)
        self.ev_rx = [self.ctx_rx.event() for _ in range(geo["nbuf"])]
        self.d_sym_mid = None       # copy of the symbols of the MIDDLE batch of a verified region (made in stream order)
        self.queued = 0
        self.retired = 0
        self.nsym = 0
        self.last_produced = 0
        self.mid_run = -1           # retire index of that batch's run; mid_produced = its symbol count
        self.mid_produced = 0

    def acquire(self, fir, rx_kw):
        """The exact serial loop locks on the head of the stream; the tiled (tracking) receiver takes over from that
        state.  Not timed."""
        capi, g = self.capi, self.geo
        acq = capi.CstlnReceiver(self.ctx, mode=capi.RX_SERIAL, **rx_kw)
        _, p0 = fir.run_dev(self.d_in.ptr, min(g["B"], 1 << 22), self.dec[0].ptr, g["n_out"])
        self.ctx.sync()
        acq.run_dev(self.dec[0].ptr, p0, self.d_sym.ptr, g["n_out"] + 256, meas=False)
        self.rx.set_state(acq.state())
        acq.close()

    def retire(self, timed, keep):
        while self.queued > keep:
            self.queued -= 1
            self.last_produced = self.rx.wait()
            if self.retired == self.mid_run:
                self.mid_produced = self.last_produced
            self.retired += 1
            if timed:
                self.nsym += self.last_produced

    def close(self):
        self.rx.close()
        self.d_in.free(); self.d_sym.free()
        if self.d_in2 is not None:
            self.d_in2.free()
        if self.d_sym_mid is not None:
            self.d_sym_mid.free()
        for d in self.dec:
            d.free()
        if self.own_rx_ctx:
            self.ctx_rx.close()


class C2Pipeline:
    """scaler(fused) -> fir_filter -> cstln_receiver(tiled) over `n_captures` endless captures on one GPU."""

    def __init__(self, capi, synth, device, n_captures, batch_msamples, period_msamples, tile, seed0, freq=0.0, rx_cus=0,
                 cu_pattern="xcd_major", rx_freq=0.0, fir_arith=None, cw=None, rx_multi=True, sampler="linear", batch_hook=None, rx_groups=1, placement_candidates=40, place=True):
        self.capi = capi
        self.batch_hook = batch_hook          # called per batch after fir_filter was queued: hook(pipe, dec buffer index, outputs, done event)
        # rx_multi: the receivers of all captures live on ONE stream and share their launches (lsdr_rx_run_multi_async): two
        # streams in all — with a stream per capture, five streams on the runtime's four hardware queues let two receivers
        # share a queue, and that queue paced the pipeline
        self.rx_multi = bool(rx_multi) and n_captures > 1
        # CU partition: the latency-bound receiver tiles get rx_cus compute units of their own (the same number from every
        # XCD), the HBM-streaming fir_filter the rest, so that neither disturbs the other's issue slots / L1
        fir_mask = rx_mask = None
        if rx_cus:
            total, per_xcd = 256, max(1, rx_cus // 8)
            if cu_pattern == "xcd_major":        # mask bit i = CU i%32 of XCD i/32
                rx_mask = [x * 32 + k for x in range(8) for k in range(per_xcd)]
            else:                                # mask bit i = CU i/8 of XCD i%8
                rx_mask = list(range(8 * per_xcd))
            fir_mask = [i for i in range(total) if i not in set(rx_mask)]
        self.rx_cus = len(rx_mask) if rx_mask else 0
        self.ctx = capi.Ctx(device, cu_mask=fir_mask)
        coeffs, decim = c2_filter(capi)
        self.coeffs, self.decim = coeffs, decim
        self.geo = c2_geometry(batch_msamples, period_msamples, len(coeffs), decim, extra=EXTRA_RRC if sampler == "rrc" else EXTRA)
        self.fir_arith = capi.FIR_EXACT if fir_arith is None else fir_arith
        self.fir = capi.FirFilter(self.ctx, coeffs, decim, in_scale=75.0, arith=self.fir_arith)
        self.rx_kw = dict(sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, omega=float(FS / decim / FM), meas_decimation=int(FS / decim),
                          freq=float(rx_freq))
        self.oracle_rx_kw = dict(sampler=1)
        if sampler == "rrc":      # leandvb --sampler rrc (leandvb.cc:440-458): fir_sampler on an RRC of ≥ 64 steps per symbol
            fs_dec = FS / decim
            steps = max(1, int(64 * FM / fs_dec))
            order = int(REJ * fs_dec * steps / (22 * (FM / 2) * ROLLOFF))
            rrc = capi.root_raised_cosine(order, float(np.float32(FM) / np.float32(fs_dec * steps)), ROLLOFF)
            self.rx_kw.update(sampler=capi.SAMP_FIR, coeffs=rrc, subsampling=steps)
            self.oracle_rx_kw = dict(sampler=2, coeffs=rrc, subsampling=steps)
            self.rrc = dict(ncoeffs=len(rrc), subsampling=steps)
        self.extra = EXTRA_RRC if sampler == "rrc" else EXTRA
        self.rx_extra = 256 if sampler == "rrc" else EXTRA     # consumed = ⌊(n_in − read-ahead)/128⌋·128 must be exactly n_out
        if rx_freq:   # what fir_filter::run does on its first call: the receiver's initial freq_tap moves the filter (dsp.h:236-244)
            self.fir.track(float(np.float32(rx_freq)), 1.0 / decim, float(np.float32(FM / FS * 0.1)))
        self.tile = tile
        # receiver streams: the captures are dealt to `rx_groups` streams, each with shared launches (1: all on one stream)
        self.rx_groups = max(1, min(int(os.environ.get("LSDR_BENCH_RX_GROUPS", rx_groups)), n_captures)) if self.rx_multi else 0
        self.ctx_rxs = [capi.Ctx(device, cu_mask=rx_mask) for _ in range(self.rx_groups)]
        self.ctx_rx = self.ctx_rxs[0] if self.ctx_rxs else None
        self.caps = [Capture(capi, synth, device, self.ctx, c, seed0 + 1000 * c, self.geo, self.rx_kw, tile, freq=freq, rx_cus=rx_mask, cw=cw,
                             shared_rx_ctx=self.ctx_rxs[c % self.rx_groups] if self.rx_multi else None, extra=self.extra) for c in range(n_captures)]
        self.arena, self.placement = None, None
        for cp in self.caps:
            cp.acquire(self.fir, self.rx_kw)
        self.ev_fir = [self.ctx.event() for _ in range(self.geo["nbuf"])]
        self.ev_pool = []
        self.fir_ms = []
        self.batch_no = 0
        self.reshifts = 0
        self.snap = None            # (capture, dec buffer index) of the batch whose loop state was snapshotted
        self.snap_mid, self.last_k = None, -1
        if len(self.caps) == 1 and place:
            self.placement = self.place_buffers(int(os.environ.get("LSDR_BENCH_PLACEMENT", placement_candidates)))

    def place_buffers(self, candidates):
        """The capture's input buffer and its decimated-stream buffers become windows of an lsdr_arena (leansdr_amd/csrc/arena.hip — the
        library's placed stream buffers: ONE large allocation, the fastest windows under a probe first).  WHERE a resident buffer lands
        decides how fast fir_filter streams it (0.37 ms over one 2 GiB buffer, 0.42 over the next, reproducibly per buffer:
        profiles/r05_bench/placement_probe.txt; ±8 % on the C2 headline from process to process).  The probe is the filter launch itself:
        over each candidate input window (filled with the capture) into the first decimated buffer, then over the chosen input into each
        candidate output window (the arena's tail).  Not timed; the data in the buffers is the same.  Returns the record for the JSON."""
        if candidates <= 1:
            return None
        capi, g, cp, ctx = self.capi, self.geo, self.caps[0], self.ctx
        n_in = g["B"] + self.extra * g["decim"] + g["N"]
        n_dec = g["n_out"] + self.extra
        nbytes = (g["B"] + g["period"]) * 8
        # (not when several ranks share this GPU: 160 GiB each would not fit)
        shared = os.environ.get("LSDR_RANK_DEVICES") and not os.environ.get("LSDR_BENCH_PLACE_SHARED")      # (PLACE_SHARED: a test of the multi-rank placement path on one GPU, small arenas)
        arena_gib = int(os.environ.get("LSDR_BENCH_ARENA_GIB", 160)) if not shared else 0
        if arena_gib <= 0:
            return None
        try:
            self.arena = capi.Arena(ctx, arena_gib << 30)
        except Exception as e:
            return dict(arena_gib=arena_gib, error=str(e)[-200:])
        # Input windows: the three fastest under the filter launch into the first decimated-stream buffer as it is.
        ins = self.arena.place(nbytes, n_best=3, max_windows=candidates, fill_from=cp.d_in.ptr,
                               probe=lambda w: self.fir.run_dev(w, n_in, cp.dec[0].ptr, n_dec))
        t_in = self.arena.probe_log()
        # The decimated-stream buffers (70 MB each, WRITTEN by the launch) matter as much, and what is fast there goes with the input window it is
        # paired with: one process — the launch 0.353 ms over the chosen input window into the buffer hipMalloc had returned, 0.408–0.417 into
        # every one of 64 arena windows; the next process all 64 at 0.352–0.362 (profiles/r06_bench/headline_repeat_before_pairing.txt).  So the PAIR is
        # chosen: for the fastest input windows in turn, the buffers there are (the incumbents) and arena windows are timed under the launch over THAT input
        # … and the launch alone is not the pipeline: the receiver reads decimated buffer k while the filter writes k + 1.  One run — input window 0.353 ms,
        # three arena output windows 0.352–0.353 each under the probe — had the launch at 0.414 ms in the pipeline where eight others (the buffers hipMalloc
        # had returned, 0.355–0.363 under the probe) had 0.393.  So per input window the output SETS play the pipeline itself — the three fastest under the
        # probe, the incumbents, arena windows from its start and from its end — 24 batches each after 4 untimed; the next input window is tried only
        # while the best set's batch takes more than 1.13 × the input window's launch alone (a good pair: ≈ 1.10–1.12 by this wall-clock measure).
        nd = len(cp.dec)
        inc = list(cp.dec)
        orig_in = cp.d_in
        made, played, t_dec_all = [], [], []           # arena windows handed out here; (ms, input window, set name, set)

        def pipeline_ms(w, dset, nb=24):
            cp.d_in, cp.dec = w, list(dset)
            self.run(4, False); self.sync()
            t0 = time.perf_counter()
            self.run(nb, False); self.sync()
            return (time.perf_counter() - t0) / nb * 1e3
        # (the buffers as hipMalloc returned them play too: one box had them at 0.374 ms in the pipeline — 699 GS/s unplaced — where the best arena pair made 0.41)
        played.append((pipeline_ms(orig_in, inc), orig_in, "input and outputs as allocated", list(inc)))
        for w in ins:
            probe_d = lambda p, w=w: self.fir.run_dev(w.ptr, n_in, p, n_dec)
            pool = [(self.arena.time(d.ptr, probe_d), d) for d in inc]
            sets = [("the buffers as allocated", list(inc))]
            for name, from_tail, nmax in (("arena windows from its end", True, 32), ("arena windows from its start", False, 8)):
                try:
                    ws = self.arena.place(n_dec * 8, n_best=nd, max_windows=nmax, from_tail=from_tail, probe=probe_d)
                except Exception:
                    continue
                t_dec_all.append([round(v, 4) for v in self.arena.probe_log()])
                made += ws
                sets.append((name, ws))
                pool += [(d.probe_ms, d) for d in ws]
            pool.sort(key=lambda e: e[0])
            fastest = [d for _, d in pool[:nd]]
            if all({id(d) for d in fastest} != {id(d) for d in ds} for _, ds in sets):
                sets.append(("fastest under the launch alone", fastest))
            here = [(pipeline_ms(w, ds), w, name, ds) for name, ds in sets]
            played += here
            if min(h[0] for h in here) <= 1.13 * w.probe_ms:
                break
        best_ms, w_best, best_name, best_set = min(played, key=lambda e: e[0])
        keep = {id(d) for d in best_set}
        for d in inc + made:
            if id(d) not in keep:
                d.free()
        if w_best is not orig_in:
            orig_in.free()
        cp.d_in, cp.dec = w_best, list(best_set)
        # The capture lives in ONE window — or in TWO read alternately where the candidates cannot be told apart (all within 4 %: none is known to be
        # of the fast kind; tools/placement_probe4.py: the same launch re-reading one buffer of the slow kind back to back streams 4.9–5.1 TB/s,
        # alternating between two of them 5.4; a fast one 5.75 either way).
        two = (max(t_in) - min(t_in)) < 0.04 * min(t_in) and os.environ.get("LSDR_BENCH_ALTERNATE", "1") != "0"
        two = two and w_best is not orig_in
        rest = [w for w in ins if w is not w_best]
        if two:
            cp.d_in2 = rest.pop(0)
        for w in rest:
            w.free()
        return dict(engine="lsdr_arena_place / lsdr_arena_time (include/lsdr_hip.h)", arena_gib=arena_gib, input_windows_tried=len(t_in), input_buffers_in_use=2 if two else 1,
                    filter_launch_ms_by_input_window=[round(float(v), 4) for v in t_in],
                    input_windows_paired=len({id(e[1]) for e in played}), chosen_pair={"input_window_launch_ms_alone": round(getattr(w_best, "probe_ms", 0.0) or 0.0, 4), "pipeline_ms_per_batch": round(best_ms, 4),
                                                                                       "output_buffers_from_the_arena": int(sum(1 for d in best_set if isinstance(d, capi.ArenaWindow)))},
                    pipeline_ms_per_batch_by_output_set=[{"input_window": round(getattr(w, "probe_ms", 0.0) or 0.0, 4), "set": name, "ms": round(ms, 4)} for ms, w, name, _ in played], output_set_in_use=best_name,
                    filter_launch_ms_by_decimated_window=t_dec_all)

    def run(self, n_batches, timed, snapshot_last=False, track_tol=None):
        """Queue n_batches batches of every capture.  Per batch: fir_filter(k) of all captures in one launch on the fir
        stream, cstln_receiver(k) of each capture on its own stream (after fir_filter(k)); results are retired two
        batches later, so the GPU never waits for the host; the pipeline is drained before returning."""
        capi, g, caps = self.capi, self.geo, self.caps
        B, n_out, N, decim, NBUF = g["B"], g["n_out"], g["N"], g["decim"], g["nbuf"]
        EXTRA, rx_n_in = self.extra, g["n_out"] + self.rx_extra
        n_in_fir = B + EXTRA * decim + N
        while timed and len(self.ev_pool) < 2 * n_batches:
            self.ev_pool.append(self.ctx.event())
        consumed = 0
        # verified regions keep two batches: the LAST one (loop-state snapshot slot 0, symbols still in d_sym afterwards) and one in
        # the MIDDLE (slot 1; its symbols are copied aside on the receiver's stream right behind its run — 9 MB, device to device)
        k_mid = n_batches // 2 - 1 if snapshot_last and n_batches >= 4 else -1
        self.snap_mid = None
        prof = os.environ.get("LSDR_BENCH_HOSTPROF") and timed        # host-side seconds per call category (diagnostic)
        tp = [0.0, 0.0, 0.0, 0.0]
        pc = time.perf_counter
        for k in range(n_batches):
            if prof: t_a = pc()
            i = self.batch_no % NBUF
            # dec[i] of every capture is free: its receiver run (batch_no − NBUF) was retired on the host (≤ 2 stay queued)
            if timed:
                self.ctx.event_record(self.ev_pool[2 * k])
            if len(caps) == 1:
                src = caps[0].d_in2 if (caps[0].d_in2 is not None and (self.batch_no & 1)) else caps[0].d_in
                cons, prod = self.fir.run_dev(src.ptr, n_in_fir, caps[0].dec[i].ptr, n_out + EXTRA)
            else:
                cons, prod = self.fir.run_multi_dev([c.d_in.ptr for c in caps], n_in_fir, [c.dec[i].ptr for c in caps], n_out + EXTRA)
            done = self.ev_pool[2 * k + 1] if timed else self.ev_fir[i]   # (timed: the stop event doubles as the "filtered" event)
            self.ctx.event_record(done)
            assert prod == n_out + EXTRA, (prod, n_out)
            if self.batch_hook is not None:
                self.batch_hook(self, i, prod, done)
            if prof: t_b = pc(); tp[0] += t_b - t_a
            if self.rx_multi:
                if snapshot_last and k == n_batches - 1:
                    self.snap = (0, i)
                    self.last_k = k
                for gi, cx in enumerate(self.ctx_rxs):
                    grp = caps[gi::self.rx_groups]
                    cx.wait_event(done)
                    if (snapshot_last and k == n_batches - 1) or k == k_mid:
                        for c in grp:
                            c.rx.snapshot_async(1 if k == k_mid else 0)
                    used = capi.CstlnReceiver.run_multi_async([c.rx for c in grp], [c.dec[i].ptr for c in grp], rx_n_in,
                                                              [c.d_sym.ptr for c in grp], n_out + EXTRA + 256)
                    assert all(u == n_out for u in used), (used, n_out)          # the stream continues exactly at the next batch
                    if k == k_mid:
                        for c in grp:
                            self._keep_mid(c, k, i)
                for c in caps:
                    c.queued += 1
            for c in ([] if self.rx_multi else caps):
                c.ctx_rx.wait_event(done)
                if snapshot_last and k == n_batches - 1:
                    c.rx.snapshot_async(0)
                    self.snap = (0, i)
                    self.last_k = k
                if k == k_mid:
                    c.rx.snapshot_async(1)
                used = c.rx.run_async(c.dec[i].ptr, rx_n_in, c.d_sym.ptr, n_out + EXTRA + 256)
                assert used == n_out, (used, n_out)          # the stream continues exactly at the next batch
                if k == k_mid:
                    self._keep_mid(c, k, i)
                c.queued += 1
            consumed += B * len(caps)
            if prof: t_c = pc(); tp[1] += t_c - t_b
            for c in caps:
                c.retire(timed, keep=2)
            if prof: tp[2] += pc() - t_c
            if track_tol is not None and caps[0].retired:
                # fir_filter follows the receiver's carrier estimate (dsp.h:236-244) from the newest run that has COMPLETED:
                # the feedback of leandvb.cc:506-510 with the queue depth as latency, no host wait between two batches
                self.reshifts += int(self.fir.track(caps[0].rx.retired_freq_tap, 1.0 / decim, track_tol))
            self.batch_no += 1
        for c in caps:
            c.retire(timed, keep=0)
        if prof:
            print(f"hostprof: per batch: fir enqueue {tp[0] / n_batches * 1e6:.1f} us, receiver enqueue ({len(caps)} captures) {tp[1] / n_batches * 1e6:.1f} us, "
                  f"retire (waits for the GPU) {tp[2] / n_batches * 1e6:.1f} us", file=sys.stderr)
        if timed:
            self.ctx.sync()
            for k in range(n_batches):       # HIP events around every fir_filter launch, on its own stream
                self.fir_ms.append(self.ctx.event_elapsed_ms(self.ev_pool[2 * k], self.ev_pool[2 * k + 1]))
        return consumed

    def _keep_mid(self, c, k, i):
        """Right behind the middle batch's receiver run, on its stream: its symbols aside (the next run overwrites d_sym)."""
        g = self.geo
        nbytes = (g["n_out"] + self.extra + 256) * 4
        if c.d_sym_mid is None:
            c.d_sym_mid = c.ctx_rx.alloc(nbytes)
        self.capi.check(self.capi.lib.lsdr_memcpy_d2d(c.ctx_rx.h, c.d_sym_mid.ptr, c.d_sym.ptr, nbytes))
        c.mid_run = c.retired + c.queued          # this run's place in the capture's retire order
        self.snap_mid = (k, i)

    def sync(self):
        self.ctx.sync()
        for c in self.caps:
            c.ctx_rx.sync()

    def verify_last_batch(self):
        """The LAST queued batch and one from the MIDDLE of the region, of EVERY capture, against the CPU oracle (test
        infrastructure, used as the checker only): fir_filter output bit for bit; soft symbols vs the oracle's exact serial
        receiver started from the loop state the device used for that very batch (snapshots taken in stream order), under
        leansdr_amd.tolerance.TOL.  A drift that built up over the queue would show in the last batch's state-restarted check
        only if it broke that batch; the middle batch is a second, independent sample of the queue."""
        caps = [self._verify_capture(ci) for ci in range(len(self.caps))]
        batches = []
        if self.snap_mid is not None:
            batches.append(dict(batch_index=self.snap_mid[0], which="middle", pass_=all(c["mid"]["pass"] for c in caps),
                                max_abs_dcost=max(c["mid"]["max_abs_dcost"] for c in caps), equal_decisions=min(c["mid"]["equal_decisions"] for c in caps)))
        batches.append(dict(batch_index=self.last_k, which="last", pass_=all(c["last_pass"] for c in caps)))
        for b in batches:
            b["pass"] = bool(b.pop("pass_"))
        out = dict(batch="last batch of the timed region + one middle batch", batches=batches, captures_checked=len(caps), tolerance=TOL, per_capture=caps,
                   checker="oracle/liblsdr_oracle.so (fir_filter; serial receiver from the device's loop state)",
                   checker_seconds=round(sum(c.pop("checker_seconds") for c in caps), 2))
        for k in ("fir_bit_exact", "count_equal", "first_tile_bit_exact"):
            out[k] = bool(all(c.get(k) for c in caps))
        out["equal_decisions"] = min(c.get("equal_decisions", 0.0) for c in caps)
        for k in ("mean_abs_dcost", "p99_abs_dcost", "max_abs_dcost", "bad_seams"):
            out[k] = max(c.get(k, 1e9) for c in caps)
        if any("fir_max_rel_err_vs_exact" in c for c in caps):
            out["fir_max_rel_err_vs_exact"] = max(c.get("fir_max_rel_err_vs_exact") or 1.0 for c in caps)
        out["pass"] = bool(all(c["pass"] for c in caps))
        return out

    def _verify_capture(self, ci):
        po = _oracle()
        O = po.Oracle()
        capi, g = self.capi, self.geo
        cp = self.caps[ci]
        B, n_out, N, decim = g["B"], g["n_out"], g["N"], g["decim"]
        EXTRA = self.extra
        st_dev = cp.rx.snapshot()
        y = self.ctx.download(cp.dec[self.snap[1]], np.complex64, n_out + EXTRA)
        sym = self.ctx.download(cp.d_sym, capi.SOFTSYM, cp.last_produced)
        t0 = time.perf_counter()
        x_full = np.concatenate([np.tile(cp.x, g["reps"]), cp.x[:EXTRA * decim + N]])
        if self.fir.current_freq:
            y_ref = O.fir_filter(self.coeffs, decim, O.scaler(75.0, x_full), freq=self.fir.current_freq)
        else:
            y_ref = O.fir_filter(self.coeffs, decim, O.scaler(75.0, x_full))
        y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
        # What the filter's output must equal bit for bit: the reference's arithmetic (LSDR_FIR_EXACT), or — tolerance modes
        # LSDR_FIR_FMA / LSDR_FIR_MFMA — the same loop with fused multiply-adds (oracle lo_fir_filter_fma).  The receiver is
        # checked against the oracle's EXACT filter → exact serial receiver either way.
        fir_extra = {}
        if self.fir_arith == capi.FIR_EXACT:
            y_want = y_ref
        else:
            if self.fir_arith == capi.FIR_MFMA_BLK:      # the scaler rides on the taps in this mode
                y_want = O.fir_filter(self.coeffs, decim, x_full, freq=self.fir.current_freq, fma="blk", scale=75.0)[0]
            else:
                y_want = O.fir_filter(self.coeffs, decim, O.scaler(75.0, x_full), freq=self.fir.current_freq, fma=True)[0]
            scale = float(np.abs(y_ref).max())
            fir_extra = dict(fir_arith={capi.FIR_FMA: "fma", capi.FIR_MFMA: "mfma", capi.FIR_MFMA_BLK: "mfma_blk"}[self.fir_arith],
                             fir_max_abs_err_vs_exact=float(np.abs(y - y_ref).max()) if len(y) == len(y_ref) else None,
                             fir_max_rel_err_vs_exact=float(np.abs(y - y_ref).max() / scale) if len(y) == len(y_ref) else None,
                             fir_rel_err_bound=1e-5)
        fir_ok = len(y_want) == len(y) and np.array_equal(y_want, y)
        # the stream is B-periodic: the other decimated-stream buffers of this capture (the two batches before) hold the same bits
        others_ok = all(np.array_equal(self.ctx.download(d, np.complex64, n_out + EXTRA), y_want)
                        for k, d in enumerate(cp.dec) if k != self.snap[1]) if self.batch_no >= len(cp.dec) else True
        fir_ok = fir_ok and others_ok
        if fir_extra:
            fir_ok = fir_ok and fir_extra["fir_max_rel_err_vs_exact"] is not None and fir_extra["fir_max_rel_err_vs_exact"] <= fir_extra["fir_rel_err_bound"]
        st = po.RxState()
        for k, _ in st._fields_:
            setattr(st, k, getattr(st_dev, k))
        p = po.rx_params(cstln=1, omega=float(FS / decim / FM), meas_decimation=int(FS / decim), **self.oracle_rx_kw)
        ref = O.rx(p, y_ref[:n_out + self.rx_extra], state_in=st)
        rep = check_tiled(sym, ref["sym"], cp.rx.tiled_stats(), first_exact=self.tile[1] // 4 - 8)   # tile 0 (exact) is one warm-up long
        rep.update(fir_extra)
        rep.update(capture=ci, fir_outputs=int(len(y)), fir_bit_exact=bool(fir_ok), fir_buffers_checked=len(cp.dec),
                   consumed_equal=bool(ref["consumed"] == n_out), checker_seconds=time.perf_counter() - t0)
        if not fir_ok and len(y_want) == len(y):
            bad = np.flatnonzero((y_want.real != y.real) | (y_want.imag != y.imag))
            rep["fir_diff"] = dict(outputs_different=int(len(bad)), first=int(bad[0]) if len(bad) else None,
                                   note=None if len(bad) else "the last batch's buffer is right; an earlier batch's buffer differs")
        rep["last_pass"] = bool(rep["pass"] and fir_ok and rep["consumed_equal"])
        rep["pass"] = rep["last_pass"]
        if self.snap_mid is not None and cp.d_sym_mid is not None:
            # the middle batch: same input bits (the stream is B-periodic; all buffers were just compared), its own loop state, its own symbols
            st_mid = cp.rx.snapshot(1)
            sym_mid = cp.ctx_rx.download(cp.d_sym_mid, capi.SOFTSYM, cp.mid_produced)
            stm = po.RxState()
            for k, _ in stm._fields_:
                setattr(stm, k, getattr(st_mid, k))
            ref_mid = O.rx(p, y_ref[:n_out + self.rx_extra], state_in=stm)
            mid = check_tiled(sym_mid, ref_mid["sym"], None, first_exact=self.tile[1] // 4 - 8)
            mid["state_differs_from_last"] = bool(any(getattr(st_mid, k) != getattr(st_dev, k) for k, _ in stm._fields_))
            rep["mid"] = {k: mid[k] for k in ("pass", "count_equal", "first_tile_bit_exact", "equal_decisions", "mean_abs_dcost", "p99_abs_dcost",
                                               "max_abs_dcost", "state_differs_from_last") if k in mid}
            rep["pass"] = bool(rep["pass"] and mid["pass"])
        rep["checker_seconds"] = time.perf_counter() - t0
        return rep

    def roofline(self):
        """fir_filter launch = the batch of every capture: algorithmic bytes per SURVEY §8(d) (8.03 B per input sample: cf32 in,
        soft symbols out — the decimated stream between the two kernels is not algorithmic traffic) over the mean launch
        duration (HIP events on the filter's stream).  kernel_bytes = what this kernel itself must move (cf32 in + cf32/D out)."""
        g = self.geo
        EXTRA = self.extra
        n_launch_out = (g["n_out"] + EXTRA) * len(self.caps)
        kernel_bytes = n_launch_out * g["decim"] * 8 + n_launch_out * 8          # cf32 in + cf32 out
        alg_bytes = int(g["B"] * len(self.caps) * ALG_BYTES_PER_SAMPLE_C2)
        ms = float(np.mean(self.fir_ms))
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        kname = {self.capi.FIR_MFMA: "k_fir_mfma", self.capi.FIR_MFMA_BLK: "k_fir_mfma_stream"}.get(self.fir_arith, "k_fir_persist")
        traffic, src = pmc_traffic(kname, g["B"] * len(self.caps))
        ceil = measured_copy_ceiling()
        return {"kernel": kname + " (fir_filter)", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "ceiling": ceil[0] if ceil else None, "ceiling_source": (f"measured streaming-read kernel (tools/membench.hip), recorded: {ceil[1]}" if ceil else None),
                "frac_of_ceiling": round(achieved / ceil[0], 4) if ceil else None, "traffic": traffic,
                "traffic_source": (f"recorded, not measured in this run: {src}" if src else None),
                "avg_launch_ms": round(ms, 4), "launches_timed": len(self.fir_ms), "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_per_sample": round(ALG_BYTES_PER_SAMPLE_C2, 4), "kernel_bytes": kernel_bytes,
                "kernel_bytes_frac": round(kernel_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    def close(self):
        for c in self.caps:
            c.close()
        for cx in self.ctx_rxs:
            cx.close()
        if getattr(self, "arena", None) is not None:
            self.arena.close(); self.arena = None
        self.fir.close()
        self.ctx.close()


def summary_of(out):
    """name → [MS/s, fraction of the HBM peak on the algorithmic bytes of SURVEY §8(d), pass flag]; compact (≤ 1.5 kB)."""
    def row(e):
        if not isinstance(e, dict) or "value" not in e:
            return ["failed", None, False]
        r = e.get("roofline") or {}
        ok = e.get("pass")
        if ok is None and isinstance(e.get("verified"), dict):
            ok = e["verified"].get("pass")
        return [round(e["value"]), r.get("hbm_frac", r.get("frac")), ok]
    sm = {"headline_c2": row(out)}
    for k, v in (out.get("more") or {}).items():
        if k == "end_to_end" and isinstance(v, dict) and "cf32" in v:
            sm["e2e_cf32"] = row(v["cf32"]); sm["e2e_cu8"] = row(v["cu8"])
        else:
            sm[k] = row(v)
    if "cpu_baseline" in out:
        sm["cpu_ref_all_cores"] = [round(out["cpu_baseline"]["value"]), None, None]
    sm["_cols"] = "MS/s, frac of 8 TB/s on SURVEY 8(d) bytes, pass"
    return sm



LINE_LIMIT = 4096     # bytes of the one JSON line on stdout (the driver keeps an 8 KB tail and parses the last line)


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "…"


def compact_line(out):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline` + the verdict of the in-run verification + the
    one-row-per-configuration summary, ≤ LINE_LIMIT bytes.  Everything else (`more`, per-capture verification reports, long
    descriptions) goes to bench_full.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in keep if k in out}
    cfg = out.get("config") or {}
    line["config"] = {k: (_short(v, 200) if k == "workload" else _short(v, 60)) for k, v in cfg.items()
                      if k in ("workload", "fir_arith", "batches_per_step", "batch_samples_per_capture", "captures_per_gpu", "samples_per_step_per_gpu",
                               "rx_mode", "rx_tile", "parallelism", "symbols_per_step", "c1_captures", "capture_samples", "engine")}
    r = out.get("roofline")
    if isinstance(r, dict):
        line["roofline"] = {k: _short(r[k], 80) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_unplaced", "ceiling", "frac_of_ceiling", "traffic",
                                                          "traffic_source", "avg_launch_ms", "launches_timed", "algorithmic_bytes_per_launch",
                                                          "algorithmic_bytes_per_sample") if k in r}
        if isinstance(r.get("valu_issue"), dict):      # c1: the bound that applies to the receiver's tiles (vector-instruction issue)
            vi = r["valu_issue"]
            line["roofline"]["valu_issue"] = {k: vi[k] for k in ("valu_instructions_per_symbol_step", "wave_instructions_per_launch", "peak_wave_instructions_per_s",
                                                                 "achieved_wave_instructions_per_s", "frac", "whole_job_frac") if k in vi}
    u = out.get("unplaced")
    if isinstance(u, dict):      # the same pipeline over the buffers as hipMalloc returned them (before lsdr_arena_place chose the headline's)
        line["unplaced"] = {k: u[k] for k in ("value", "unit", "ms_per_step", "avg_launch_ms", "frac") if k in u}
    c = out.get("cpu_baseline")
    if isinstance(c, dict):
        line["cpu_baseline"] = {k: _short(c[k], 230) for k in ("value", "unit", "cores", "kind", "one_core", "sample") if k in c}
    v = out.get("verified")
    if isinstance(v, dict):
        line["verified"] = {k: v[k] for k in ("pass", "batches", "captures_checked", "fir_bit_exact", "count_equal", "first_tile_bit_exact", "equal_decisions",
                                              "mean_abs_dcost", "p99_abs_dcost", "max_abs_dcost", "bad_seams", "fir_max_rel_err_vs_exact",
                                              "ts_identical", "ranks_passed", "ranks") if k in v}
    if "summary" in out:
        line["summary"] = out["summary"]
    if "full" in out:
        line["full"] = out["full"]
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt.encode()) > LINE_LIMIT:            # never outgrow the driver's parser again: drop the optional parts, largest first
        for k in ("summary", "verified", "full"):
            line.pop(k, None)
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt.encode()) <= LINE_LIMIT:
                break
    return txt


def emit(out):
    """Full record → bench_full.json (repo root, and gpurun_out/ when that exists: it is what travels back from the GPU box);
    compact record → the single stdout line."""
    paths = []
    override = os.environ.get("LSDR_BENCH_FULL")      # a caller that runs bench.py as a sub-process (bench_more.c1) names its own file
    for fn in ([override] if override else [os.path.join(d, "bench_full.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]):
        try:
            with open(fn, "w") as f:
                json.dump(out, f)
            paths.append(os.path.relpath(fn, ROOT))
        except OSError:
            pass
    out = dict(out, full=paths[-1] if paths else None)
    print(compact_line(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batches-per-step", type=int, default=None, help="batches of every capture in one step (default: 24 Gi samples per step and GPU)")
    ap.add_argument("--batch-msamples", type=int, default=None, help="Mi input samples per batch per capture (default: 256 per GPU, shared by its captures)")
    ap.add_argument("--period-msamples", type=int, default=4, help="unique synthetic period (Mi samples, circular)")
    ap.add_argument("--tile-len", type=int, default=DEFAULT_TILE[0])
    ap.add_argument("--tile-warmup", type=int, default=DEFAULT_TILE[1])
    ap.add_argument("--captures", type=int, default=1,
                    help="independent captures demodulated concurrently on each GPU (own buffers and block handles, shared launches); "
                         "1 = north_star's one capture per GPU (`more.four_captures` is the batched-streams case)")
    ap.add_argument("--fir-arith", choices=["exact", "fma", "mfma", "blk"], default="blk",
                    help="fir_filter arithmetic of the headline.  blk (default) = block-polyphase on the f32 matrix pipe (LSDR_FIR_MFMA_BLK: output "
                         "bit-identical to the oracle's restatement lo_fir_filter_blk, ≤ 1e-5 of full scale from the reference's arithmetic; "
                         "north_star's contract: soft symbols within the stated tolerance, TS bit-exact — both checked); exact = the reference's "
                         "arithmetic (bit-exact filter output: `more.c2_exact`); fma / mfma = one fmaf chain per output (VALU / matrix pipe)")
    ap.add_argument("--rx-per-stream", action="store_true", help="one HIP stream and one set of launches per capture's receiver (round 3's arrangement) "
                                                                 "instead of shared launches on one stream (lsdr_rx_run_multi_async)")
    ap.add_argument("--rx-cus", type=int, default=0, help="compute units reserved for the receiver streams (0: no partition)")
    ap.add_argument("--cu-pattern", choices=["xcd_major", "interleaved"], default="xcd_major")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-more", action="store_true", help="skip the secondary configurations (`more` in the JSON line)")
    ap.add_argument("--dry-run", action="store_true", help="launch/aggregation plumbing only: no GPU work")
    ap.add_argument("--workload", choices=["c2", "c1"], default="c2",
                    help="c2: BASELINE config 2 (the headline: cf32 at 120 sps, fir_filter + receiver); c1: BASELINE config 1 / 4 — independent "
                         "cu8 captures at 1.2 sps decoded to TS (bench_c1.py); with --gpus N that is config 4 (captures sharded over the GPUs)")
    ap.add_argument("--c1-captures", type=int, default=32, help="c1: independent captures resident per GPU (each decoded once per step)")
    ap.add_argument("--c1-msamples", type=int, default=128, help="c1: Mi samples per capture")
    ap.add_argument("--c1-workers", type=int, default=16, help="c1 --c1-mode chain: host threads / HIP streams decoding captures concurrently per GPU")
    ap.add_argument("--no-single", action="store_true", help="c1: skip the one-capture-per-GPU latency figure")
    ap.add_argument("--c1-mode", choices=["batch", "chain"], default="batch",
                    help="c1: batch = lsdr_capture_batch (shared launches, counts on the device, one host thread); chain = the C-ABI blocks one by one (rounds 2-5)")
    ap.add_argument("--c1-groups", type=int, default=2, help="c1 batch: capture groups per GPU (one lsdr_capture_batch + stream each)")
    ap.add_argument("--c1-aux-cus", type=int, default=0, help="c1 batch: compute units reserved for everything but the receiver's tiles (lsdr_capture_batch_cfg::aux_cus; 0: no partition)")
    ap.add_argument("--c1-anf", type=int, default=1, help="c1 batch: auto_notch slots (1 = leandvb's default, 0 = --anf 0)")
    ap.add_argument("--c1-tile", type=int, default=0, help="c1: receiver tile length in samples (0: 4096 batch / 2048 chain)")
    ap.add_argument("--c1-warmup", type=int, default=512)
    args = resolve_defaults(ap.parse_args())

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from leansdr_amd.shard import Shard
    shard = Shard()                       # one process per GPU; gloo (no RCCL) only when WORLD_SIZE > 1
    rank, world = shard.rank, shard.world
    local_rank = shard.device_index()     # LOCAL_RANK (or the rank → GPU map of LSDR_RANK_DEVICES)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")

    if args.dry_run:
        shard.barrier()
        total, dt, _ = shard.aggregate(1000.0 * (rank + 1), 1.0 + rank)
        if args.workload == "c1":       # the noise seeds every rank WOULD give its captures (the same functions the real run calls)
            import bench_c1
            seeds = shard.gather_ints(bench_c1.capture_seeds(rank, args.c1_captures))
        else:
            seeds = shard.gather_ints([shard.capture_seed() + 1000 * c for c in range(args.captures)])
        if rank == 0:
            print(json.dumps({"dry_run": True, "workload": args.workload, "n_gpus": world, "ranks": world, "units": total, "seconds": dt,
                              "capture_seeds_by_rank": seeds}), flush=True)
        shard.close()
        return

    if args.workload == "c1":
        if args.c1_mode == "chain":
            # rounds 2-5: 16 worker streams — by default the HIP runtime multiplexes a process's streams onto 4 hardware queues, and two
            # captures whose kernels share a queue run one after the other.  One queue per worker: 123 -> 139 GS/s.  The batch engine
            # (two streams, one host thread) runs on the runtime's defaults.
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        import leansdr_amd.capi as capi
        import bench_c1
        if capi.lib.lsdr_device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, only {capi.lib.lsdr_device_count()} visible")
        if world > 1:       # the 16 worker threads of a rank inherit its affinity: the CPUs next to its GPU
            shard.pin_to_gpu_numa(capi.device_pci_bus_id(local_rank))
        out, rc = bench_c1.run_workload(capi, local_rank, args, shard)
        if rank == 0:
            emit(out)
        shard.close()
        sys.exit(rc)

    from leansdr_amd import synth
    if rank == 0 and world == 1 and not args.no_more and (not os.environ.get("LSDR_BENCH_MORE_ONLY") or "c1" in os.environ["LSDR_BENCH_MORE_ONLY"].split(",")):
        import bench_more
        bench_more.c1_early(args)      # (the config-1/4 entry of `more`: a process of its own, before this one holds a context on the GPU)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:     # before the GPU is touched: the CPU workers are forked
        order, fcut, decim0 = c2_design()
        c0 = _oracle().Oracle().lowpass(order, fcut)
        g0 = c2_geometry(args.batch_msamples, args.period_msamples, len(c0), decim0)
        x_cpu, _ = synth.qpsk_baseband(g0["period"], g0["sps"], seed=shard.capture_seed(), rms=1.0, snr_db=20.0)
        cpu = cpu_baseline(x_cpu, c0, decim0, args.cpu_seconds)
        del x_cpu

    import leansdr_amd.capi as capi

    if capi.lib.lsdr_device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, only {capi.lib.lsdr_device_count()} visible")
    numa = shard.pin_to_gpu_numa(capi.device_pci_bus_id(local_rank)) if world > 1 else None      # (after the CPU baseline's all-core pass)
    tile = (args.tile_len, args.tile_warmup)
    pipe = C2Pipeline(capi, synth, local_rank, args.captures, args.batch_msamples, args.period_msamples, tile,
                      seed0=shard.capture_seed(), rx_cus=args.rx_cus, cu_pattern=args.cu_pattern,
                      fir_arith={"exact": capi.FIR_EXACT, "fma": capi.FIR_FMA, "mfma": capi.FIR_MFMA, "blk": capi.FIR_MFMA_BLK}[args.fir_arith],
                      rx_multi=not args.rx_per_stream, place=False)
    bps = args.batches_per_step
    # Buffer placement: the headline runs over windows of an lsdr_arena chosen by the library (C2Pipeline.place_buffers).  What the SAME
    # pipeline does over the buffers as hipMalloc returns them is measured first, the same number of steps, and reported beside it
    # (`unplaced`; roofline.frac_unplaced) — not part of the timed region below.
    unplaced = None
    n_cand = int(os.environ.get("LSDR_BENCH_PLACEMENT", 40))
    if args.captures == 1 and n_cand > 1 and (not os.environ.get("LSDR_RANK_DEVICES") or os.environ.get("LSDR_BENCH_PLACE_SHARED")):
        if world == 1 and os.environ.get("LSDR_BENCH_UNPLACED", "1") != "0":      # (0: profiling runs — fewer launches that are not the timed region's)
            pipe.run(args.warmup * bps, False)
            pipe.sync()
            t0 = time.perf_counter()
            c_un = pipe.run(args.steps * bps, True)
            pipe.sync()
            dt_un = time.perf_counter() - t0
            r_un = pipe.roofline()
            unplaced = {"value": round(c_un / dt_un / 1e6, 3), "unit": "MS/s", "ms_per_step": round(dt_un / args.steps * 1e3, 4),
                        "avg_launch_ms": r_un["avg_launch_ms"], "frac": r_un["frac"], "what": "the same pipeline and step count over the buffers as hipMalloc returned them, before placement"}
            pipe.fir_ms = []
        pipe.placement = pipe.place_buffers(n_cand)

    pipe.run(args.warmup * bps, False)
    pipe.sync()
    shard.barrier()
    t0 = time.perf_counter()
    consumed = pipe.run(args.steps * bps, True, snapshot_last=not args.no_verify)
    pipe.sync()
    shard.barrier()
    dt = time.perf_counter() - t0

    total, dt, _ = shard.aggregate(consumed, dt)   # all ranks' samples ÷ the slowest rank's time

    # EVERY rank checks its own captures against the oracle and contributes its verdict (gloo); rank 0 reports the job's
    rc = 0
    verified = None
    if not args.no_verify:
        verified = pipe.verify_last_batch()
        ranks_ok, ranks = shard.all_ranks_ok(verified["pass"])
        if not verified["pass"]:
            print(f"bench.py: rank {rank}: VERIFICATION FAILED: " + json.dumps(verified)[:3000], file=sys.stderr)
            rc = 3
    if rank == 0:
        g = pipe.geo
        nsym = sum(c.nsym for c in pipe.caps)
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
            "config": {"workload": "BASELINE config 2: QPSK 1/2, Fs 240 MS/s cf32 (120 sps), device-resident endless stream; "
                                   "scaler(x75 fused) + fir_filter(N=313,D=30) + cstln_receiver(omega 4, linear sampler)",
                       "fir_arith": {"exact": "LSDR_FIR_EXACT (the reference's arithmetic, bit-exact)", "fma": "LSDR_FIR_FMA", "mfma": "LSDR_FIR_MFMA",
                                     "blk": "LSDR_FIR_MFMA_BLK (f32 matrix pipe, block-polyphase; tolerance mode: filter output pinned bit for bit to "
                                            "oracle lo_fir_filter_blk, soft symbols under leansdr_amd.tolerance.TOL vs the exact chain)"}[args.fir_arith],
                       "rx_launches": "per capture (one stream each)" if args.rx_per_stream or len(pipe.caps) == 1 else "shared by the captures of a GPU (lsdr_rx_run_multi_async, one stream)",
                       "batches_per_step": bps, "batch_samples_per_capture": g["B"], "captures_per_gpu": len(pipe.caps),
                       "samples_per_step_per_gpu": g["B"] * len(pipe.caps) * bps,
                       "rx_mode": "tiled", "rx_tile": {"tile_len": tile[0], "warmup": tile[1]},
                       "cu_partition": {"receiver_cus": pipe.rx_cus, "fir_filter_cus": 256 - pipe.rx_cus} if pipe.rx_cus else None,
                       "rx_tiles_last_run": pipe.caps[0].rx.tiled_stats(), "rx_decisions": pipe.caps[0].rx.decision_mode(),
                       "streams": ("fir_filter(k+1) || cstln_receiver(k): two HIP streams, receiver runs queued" if len(pipe.caps) == 1 else
                                   "fir_filter(k+1) of all captures in one launch (lsdr_fir_filter_run_multi) || cstln_receiver(k) of all captures in "
                                   "shared launches (lsdr_rx_run_multi_async): two HIP streams, receiver runs queued"),
                       "parallelism": f"{world * len(pipe.caps)} independent capture(s), {len(pipe.caps)} per GPU, no collectives, no RCCL",
                       "rank0_numa": numa, "buffer_placement": pipe.placement,
                       "symbols_per_step": nsym // max(1, args.steps)},
            "roofline": pipe.roofline(),
        }
        if unplaced is not None:
            out["unplaced"] = unplaced
            out["roofline"]["frac_unplaced"] = unplaced["frac"]
        if verified is not None:
            out["verified"] = verified                              # rank 0's own captures, in full
            out["verified"]["ranks_passed"], out["verified"]["ranks"] = ranks_ok, ranks
            out["verified"]["pass"] = bool(verified["pass"] and ranks_ok == ranks)
            if not out["verified"]["pass"]:
                rc = 3
    coeffs, decim = pipe.coeffs, pipe.decim
    pipe.close()

    if rank == 0:
        if world == 1 and not args.no_more:
            import bench_more
            out["more"] = bench_more.run_all(capi, synth, local_rank, args)
        if cpu is not None:   # reported at N=1 only
            assert np.array_equal(c0, coeffs), "cpu_baseline used other filter coefficients than the GPU path"
            out["cpu_baseline"] = cpu
        out["summary"] = summary_of(out)
        emit(out)
    shard.close()
    sys.exit(rc)


if __name__ == "__main__":
    main()
