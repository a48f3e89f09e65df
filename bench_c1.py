"""bench_c1.py — BASELINE config 1 / config 4: independent 2 MS/s-class QPSK 1/2 captures (cu8 IQ at 1.2 samples/symbol, the
README's `leandvb --u8 -f 2400e3 --sr 2000e3 --cr 1/2` case WITH ITS DEFAULTS: `--anf 1`, linear sampler, algebraic deconvolution)
decoded from the first sample to transport-stream packets:

    cconverter<u8> → auto_notch(1 slot) → cstln_receiver (linear sampler) → deconvol_sync → mpeg_sync → deinterleaver →
    rs_decoder → derandomizer → TS packets in host memory

Unit of work = one CAPTURE = one scheduler instance of the reference (leandvb.cc:163, 205-600): every decode starts from freshly
constructed blocks (acquisition included) and ends with the capture's TS in pinned host memory.  `captures` captures live in HBM per
GPU; a step decodes each of them once.

C1Job (round 6): the captures of a GPU are decoded by lsdr_capture_batch objects (include/lsdr_hip.h) — ALL captures of a group share
every launch (the notch's detect chain, the receiver's tiles with the notch inside, the seam pass, the compaction, the FEC tail), every
data-dependent count stays on the device, the host reads one result record per capture and step.  Two groups on two streams, driven by
ONE host thread: one group's FEC tail and TS download run under the other group's tiles.  No GPU_MAX_HW_QUEUES, no worker threads.

ChainJob (rounds 2-5, `--c1-mode chain`): one host thread + stream per capture calling the blocks of the C ABI one by one — kept as the
checker of the device-resident control flow (tests/test_gpu_capture_batch.py) and as a bench row.

Verification after the clock stops (`verified`): the IQ of EVERY capture goes through the reference's own binary (oracle/_ref/leandvb
--u8 -f 2400e3 --sr 2000e3 --cr 1/2 [--anf 0 only in the anf-0 variant], built from /root/reference by oracle/Makefile; it travels with
the repo) on the host cores, and the TS must be the reference's, byte for byte; without the binary the check falls back to recorded
hashes / the transmitted packet sequence (and says so).
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, FM = 2400e3, 2000e3
OMEGA = float(np.float32(FS / FM))
ALG_BYTES_PER_SAMPLE = 2.0 + 188.0 / (204 * 8 * 1.2)       # cu8 in + TS out = 2.096
# vector instructions k_rxb_tiles issues per symbol step and lane [without, with the notch]: counted from the kernel's ISA (the symbol
# loop of `llvm-objdump -d`), confirmed by SQ_INSTS_VALU (profiles/r06_bench/c1_pmc_sq.txt); LSDR_C1_VALU_PER_SYMBOL overrides
VALU_PER_SYMBOL_STEP = (92.2, 112.7)
REFBIN = os.path.join(ROOT, "oracle", "_ref", "leandvb")
REF_ARGS_BASE = ["--u8", "-f", "2400e3", "--sr", "2000e3", "--cr", "1/2"]       # the README example: everything else at its default (--anf 1)
REF_ARGS = REF_ARGS_BASE + ["--anf", "0"]                                          # the variant rounds 2-5 measured


def ref_args(anf):
    return REF_ARGS_BASE if anf else REF_ARGS
SKIP_ACQ = 16            # packets of the reference's output skipped before the comparison (lock instants may differ by a few packets)
GOLDEN = os.path.join(ROOT, "tests", "golden", "c1_ts.json")   # recorded on an MI355X box with the reference binary next to it


def recorded_hashes():
    try:
        import json
        with open(GOLDEN) as f:
            return {(e["samples"], e["seed"], e["first_packet"]): e for e in json.load(f)["captures"]}
    except (OSError, ValueError, KeyError):
        return {}


def ts_packets(n, start=0):
    from leansdr_amd import synth_dvbs
    return synth_dvbs.ts_packets(n, start)


class Generator:
    """leantsgen | leandvbtx -f 6/5 --power 37.5 --agc | leanchansim --awgn 17.5 --ou8 with this repo's GPU blocks (tx.hip,
    chan.hip: each bit-exact against the reference class, tests/test_gpu_tx.py, test_gpu_chan.py), device-resident.  The clean
    baseband is modulated once; a capture is a stretch of it (its own start packet) plus its own noise (srand48 seed)."""

    def __init__(self, capi, ctx, n_samples, n_captures):
        self.capi, self.ctx, self.n = capi, ctx, n_samples
        lib = capi.lib
        sps_num, sps_den = 6, 5
        self.shift_packets = 3                                       # capture k starts 3k packets (5875.2 samples) later
        n_pk = int((n_samples + 64) * sps_den / (204 * 8 * sps_num)) + self.shift_packets * n_captures + 128
        n_pk = (n_pk + 7) // 8 * 8
        self.n_packets = n_pk
        ts = ts_packets(n_pk)
        cons, prod = C.c_size_t(), C.c_size_t()

        def call(fn, *a):
            capi.check(fn(*a, C.byref(cons), C.byref(prod)))
            return prod.value
        d_ts = ctx.upload(ts)
        rand = C.c_void_p(); capi.check(lib.lsdr_randomizer_create(ctx.h, C.byref(rand)))
        d_a = ctx.alloc(n_pk * 188)
        call(lib.lsdr_randomizer_run, rand, d_ts.ptr, n_pk, d_a.ptr, n_pk)
        d_b = ctx.alloc(n_pk * 204)
        call(lib.lsdr_rs_encoder_run, ctx.h, d_a.ptr, n_pk, d_b.ptr, n_pk)
        d_il = ctx.alloc(n_pk * 204)
        n_il = call(lib.lsdr_interleaver_run, ctx.h, d_b.ptr, n_pk, d_il.ptr, n_pk * 204)
        conv = C.c_void_p(); capi.check(lib.lsdr_convol_create(ctx.h, capi.FEC12, 2, C.byref(conv)))
        d_sym = ctx.alloc(n_il * 8 + 64)
        n_sym = call(lib.lsdr_convol_run, conv, d_il.ptr, n_il, d_sym.ptr, n_il * 8 + 64)
        d_iq = ctx.alloc(n_sym * 8)
        capi.check(lib.lsdr_cstln_transmitter_run(ctx.h, capi.QPSK, capi.FEC12, d_sym.ptr, n_sym, d_iq.ptr))
        amp = float(np.float32(10 ** (37.5 / 20)))                   # leandvbtx --power 37.5
        co = capi.root_raised_cosine(int(sps_num * 10.0), float(np.float32(1.0) / np.float32(sps_num)), 0.35)
        co = capi.normalize_power(co, float(np.float32(amp) / np.float32(75.0)))
        res = C.c_void_p(); capi.check(lib.lsdr_fir_resampler_create(ctx.h, len(co), capi._np(co), sps_num, C.byref(res)))
        d_up = ctx.alloc(n_sym * sps_num * 8)
        n_up = call(lib.lsdr_fir_resampler_run, res, d_iq.ptr, n_sym, d_up.ptr, n_sym * sps_num)
        d_dec = ctx.alloc((n_up // sps_den + 8) * 8)
        pr = C.c_size_t()
        capi.check(lib.lsdr_decimator_run(ctx.h, sps_den, d_up.ptr, n_up, d_dec.ptr, n_up // sps_den + 8, C.byref(pr)))
        n_dec = pr.value
        agc = C.c_void_p()
        capi.check(lib.lsdr_simple_agc_create(ctx.h, float(np.float32(amp) / np.sqrt(np.float32(np.float32(sps_num) / sps_den))),
                                              float(np.float32(0.001 * sps_den / sps_num)), C.byref(agc)))
        self.d_base = ctx.alloc(n_dec * 8)
        self.n_base = call(lib.lsdr_simple_agc_run, agc, d_dec.ptr, n_dec, self.d_base.ptr, n_dec)
        ctx.sync()
        for d in (d_ts, d_a, d_b, d_il, d_sym, d_iq, d_up, d_dec):
            d.free()
        lib.lsdr_randomizer_destroy(rand); lib.lsdr_convol_destroy(conv); lib.lsdr_fir_resampler_destroy(res); lib.lsdr_simple_agc_destroy(agc)
        self.ts = ts
        self.spp5 = 204 * 8 * sps_num                                 # samples per 5 packets (9792)
        self.d_noisy = ctx.alloc(n_samples * 8)

    def capture(self, k, seed):
        """cu8 capture k in a new device buffer; returns (buffer, first packet index of the stretch)."""
        capi, ctx, lib = self.capi, self.ctx, self.capi.lib
        # start on a whole number of samples: 5 packets = 9792 samples; the AGC / filter transients of the modulator are skipped
        first_pk = 40 + 5 * ((self.shift_packets * k + 4) // 5)
        start = first_pk // 5 * self.spp5
        assert start + self.n <= self.n_base, (start, self.n, self.n_base)
        w = C.c_void_p()
        capi.check(lib.lsdr_wgn_create(ctx.h, 1, int(seed), C.byref(w)))
        stddev = float(np.float32(10 ** (17.5 / 20)))                 # leanchansim --awgn 17.5
        capi.check(lib.lsdr_wgn_run(w, stddev, self.d_base.at(start * 8), self.d_noisy.ptr, self.n))
        d_u8 = ctx.alloc(self.n * 2 + 64)
        capi.check(lib.lsdr_cconverter_f32_u8_run(ctx.h, self.d_noisy.ptr, self.n, d_u8.ptr))
        ctx.sync()
        lib.lsdr_wgn_destroy(w)
        return d_u8, first_pk

    def close(self):
        self.d_base.free(); self.d_noisy.free()


class Worker:
    """One context (HIP stream) with the block handles and buffers of one capture's chain."""

    def __init__(self, capi, device, n_samples, tile, warm):
        self.capi, self.n = capi, n_samples
        ctx = self.ctx = capi.Ctx(device)
        self.sym_cap = int(n_samples * 0.94) + 65536          # the tiled run reserves ⌈128/(omega−0.1)⌉+3 symbol slots per chunk
        # Two receiver fronts (own stream, own receiver, own packed-symbol buffer each): the receiver of the worker's NEXT capture is
        # queued before the FEC tail of the current one starts, so the tail's data-dependent host round trips never leave this
        # worker without a tile kernel on the GPU.  fronts = 1: one capture at a time (LSDR_C1_FRONTS).
        self.fronts = []
        for i in range(max(1, int(os.environ.get("LSDR_C1_FRONTS", 2)))):
            fc = ctx if i == 0 else capi.Ctx(device)
            rx = capi.CstlnReceiver(fc, sampler=capi.SAMP_LINEAR, cstln=capi.QPSK, fec=capi.FEC12, omega=OMEGA, meas_decimation=int(FS / 5),
                                    mode=capi.RX_TILED, tile_len=tile, tile_warmup=warm, in_format=capi.IN_CU8, out_format=capi.SYM_HARD2)
            self.fronts.append(dict(ctx=fc, rx=rx, d_words=fc.alloc(self.sym_cap // 4 + 256), consumed=0, t0=0.0))
        self.rx = self.fronts[0]["rx"]
        self.dec = capi.Deconv(ctx, capi.FEC12)
        self.msync = capi.MpegSync(ctx)
        self.derand = capi.Derandomizer(ctx)
        self.byte_cap = self.sym_cap // 8 + 65536
        self.pk_cap = self.byte_cap // 204 + 64
        self.d_bytes = ctx.alloc(self.byte_cap)
        self.d_mpeg = ctx.alloc(self.byte_cap)
        self.d_rs = ctx.alloc(self.pk_cap * 204)
        self.d_rts = ctx.alloc(self.pk_cap * 188)
        self.d_ts = ctx.alloc(self.pk_cap * 188)
        self.stats = dict(bits=0, errs=0, next_sync=0, jobs=0)
        self.t_stage = dict(receiver=0.0, deconv_sync=0.0, rs_derand=0.0)

    def start(self, slot, d_iq):
        """Queue the receiver of one capture (first sample to packed decisions) on front `slot`."""
        f = self.fronts[slot]
        f["t0"] = time.perf_counter()
        f["rx"].reset()
        f["consumed"] = f["rx"].run_async_hs2(d_iq.ptr, self.n, f["d_words"].ptr, 0, self.sym_cap)

    def decode(self, d_iq, h_ts_ptr, h_ts_cap):
        """One capture, first sample to TS in host memory.  Returns (TS packets, samples consumed by the receiver)."""
        self.start(0, d_iq)
        return self.finish(0, h_ts_ptr, h_ts_cap)

    def finish(self, slot, h_ts_ptr, h_ts_cap):
        """The FEC tail of the capture whose receiver was queued on front `slot`: packed decisions to TS in host memory."""
        capi, lib, ctx = self.capi, self.capi.lib, self.ctx
        f = self.fronts[slot]
        d_words, consumed, t0 = f["d_words"], f["consumed"], f["t0"]
        self.dec.reset(); self.msync.reset()
        capi.check(lib.lsdr_derandomizer_reset(self.derand.h))
        nsym = f["rx"].wait()
        t1 = time.perf_counter()
        # deconvol_sync ↔ mpeg_sync: until mpeg_sync has locked the deconvolver is given small windows (the reference's pipe
        # sizes bound how far it runs ahead of a next_sync(), leandvb.cc:185-202); once locked, the rest of the capture in one call
        pos = bw = br = mw = 0
        while True:
            cap = self.byte_cap - bw if self.msync.locked else min(65536, self.byte_cap - bw)
            c, p = self.dec.run_dev_hs2(d_words.ptr, pos, nsym - pos, self.d_bytes.at(bw), cap)
            if not p:
                break
            pos += c; bw += p
            while True:
                c3, p3, _, _, cns = self.msync.run_dev(self.d_bytes.at(br), bw - br, self.d_mpeg.at(mw), self.byte_cap - mw)
                if cns:
                    self.dec.next_sync(); self.stats["next_sync"] += 1
                if not c3 and not p3:
                    break
                br += c3; mw += p3
        t2 = time.perf_counter()
        cons, prod = C.c_size_t(), C.c_size_t()
        capi.check(lib.lsdr_deinterleaver_run(ctx.h, self.d_mpeg.ptr, mw, self.d_rs.ptr, self.pk_cap, C.byref(cons), C.byref(prod)))
        npk, n_ts = prod.value, 0
        if npk:
            b, e = C.c_long(), C.c_long()
            capi.check(lib.lsdr_rs_decoder_run(ctx.h, self.d_rs.ptr, npk, self.d_rts.ptr, C.byref(b), C.byref(e)))
            self.stats["bits"] += b.value; self.stats["errs"] += e.value
            c2, p2 = C.c_size_t(), C.c_size_t()
            capi.check(lib.lsdr_derandomizer_run(self.derand.h, self.d_rts.ptr, npk, self.d_ts.ptr, self.pk_cap, C.byref(c2), C.byref(p2)))
            n_ts = p2.value
            assert n_ts * 188 <= h_ts_cap
            capi.check(lib.lsdr_memcpy_d2h(ctx.h, h_ts_ptr, self.d_ts.ptr, n_ts * 188))     # pinned destination: asynchronous
        t3 = time.perf_counter()
        self.t_stage["receiver"] += t1 - t0; self.t_stage["deconv_sync"] += t2 - t1; self.t_stage["rs_derand"] += t3 - t2
        self.stats["jobs"] += 1
        return n_ts, consumed

    def close(self):
        self.ctx.sync()
        for f in self.fronts:
            f["ctx"].sync(); f["rx"].close(); f["d_words"].free()
        self.dec.close(); self.msync.close(); self.derand.close()
        for d in (self.d_bytes, self.d_mpeg, self.d_rs, self.d_rts, self.d_ts):
            d.free()
        for f in self.fronts[1:]:
            f["ctx"].close()
        self.ctx.close()


def capture_seeds(rank, n_captures):
    """Noise seeds of rank `rank`'s captures: 1000·(rank+1)+k — no two captures of a job share one (bench.py --dry-run prints them)."""
    return [1000 * (rank + 1) + k for k in range(n_captures)]


class JobBase:
    """The captures of one GPU (generated on the device), their TS buffers in pinned host memory, and the check against the reference."""
    anf = 0

    def _make_captures(self, capi, device, n_captures, n_samples, seed0):
        self.capi, self.device = capi, device
        self.n = n_samples
        ctx = self.ctx = capi.Ctx(device)
        gen = Generator(capi, ctx, self.n, n_captures)
        self.ts_sent = gen.ts
        self.caps, self.first_pk, self.seeds = [], [], []
        for k in range(n_captures):
            d, fp = gen.capture(k, seed0 + k)
            self.caps.append(d); self.first_pk.append(fp); self.seeds.append(seed0 + k)
        gen.close()
        self.n_ts = [0] * n_captures
        self.counts = [[] for _ in range(n_captures)]

    def _make_ts_buffers(self, ts_cap):
        self.ts_cap = ts_cap
        self.h_ts = []
        for _ in range(len(self.caps)):
            p = C.c_void_p()
            self.capi.check(self.capi.lib.lsdr_malloc_host(self.ts_cap, C.byref(p)))
            self.h_ts.append(p)


class ChainJob(JobBase):
    """Rounds 2-5: the blocks of the C ABI called one by one, one host thread + stream per capture in flight (`--anf 0` graph)."""
    anf = 0

    def __init__(self, capi, device, n_captures, msamples, workers, tile, warm, seed0):
        self._make_captures(capi, device, n_captures, (msamples << 20) // 128 * 128 + 1, seed0)
        self.workers = [Worker(capi, device, self.n, tile, warm) for _ in range(max(1, workers))]
        self._make_ts_buffers(self.workers[0].pk_cap * 188)
        self.tile = (tile, warm)

    def run(self, steps, timed=False):
        """`steps` decodes of every capture, spread over the workers.  Returns the samples consumed."""
        jobs = [(s, k) for s in range(steps) for k in range(len(self.caps))]
        lock, nxt, total, failure = threading.Lock(), [0], [0], []

        def work(w):
            try:
                pending, turn = None, 0       # (capture, front) whose receiver is queued and whose tail is still to run
                while True:
                    with lock:
                        i = nxt[0]; nxt[0] += 1
                    cur = None
                    if i < len(jobs) and not failure:
                        _, k = jobs[i]
                        slot = turn % len(w.fronts); turn += 1
                        if pending is not None and pending[1] == slot:      # a single front: finish before it is reused
                            self._finish(w, pending, timed, lock, total); pending = None
                        w.start(slot, self.caps[k])
                        cur = (k, slot)
                    if pending is not None:
                        self._finish(w, pending, timed, lock, total)
                    pending = cur
                    if cur is None:
                        return
            except BaseException as e:
                failure.append(e)
        if timed:
            for w in self.workers:
                for f in w.fronts:
                    f["rx"].tile_time(True)
        ths = [threading.Thread(target=work, args=(w,)) for w in self.workers]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for w in self.workers:
            w.ctx.sync()
        if failure:
            raise failure[0]
        return total[0]

    def _finish(self, w, pend, timed, lock, total):
        k, slot = pend
        n_ts, cons = w.finish(slot, self.h_ts[k], self.ts_cap)
        self.n_ts[k] = n_ts
        if timed:
            self.counts[k].append(n_ts)
        with lock:
            total[0] += cons

    def tile_kernel_ms(self):
        ms, n = 0.0, 0
        for w in self.workers:
            for f in w.fronts:
                a, b = f["rx"].tile_time(False)
                ms += a * b; n += b
        return (ms / n if n else 0.0), n

    def ts_of(self, k):
        return bytes(C.string_at(self.h_ts[k], self.n_ts[k] * 188))

    def iq_of(self, k):
        return self.ctx.download(self.caps[k], np.uint8, self.n * 2)

    def verify(self, max_ref_workers=None):
        """Every capture: this path's TS of the last decode against the reference binary's TS for the same IQ."""
        out = dict(captures=len(self.caps), checker=None, per_capture=[])
        rec = recorded_hashes()
        have_ref = os.path.exists(REFBIN) and os.access(REFBIN, os.X_OK)
        t0 = time.perf_counter()
        refs = [None] * len(self.caps)
        if have_ref:
            out["checker"] = "oracle/_ref/leandvb " + " ".join(ref_args(self.anf)) + " (the reference binary, one process per capture on the host cores)"
            # the captures are read back one after the other (one context); hashing them and running the reference on them — one single-threaded
            # process per capture, fed through a pipe — goes to a pool of host threads (both release the interpreter lock), at most 12 captures in flight
            import threading
            from concurrent.futures import ThreadPoolExecutor
            self.iq_sha = {}
            room = threading.Semaphore(12)

            def ref_job(k, iq):
                try:
                    sha = hashlib.sha256(iq).hexdigest()
                    p = subprocess.Popen([REFBIN] + ref_args(self.anf), stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                    ts = p.communicate(memoryview(iq).cast("B"))[0]
                    return k, sha, ts
                finally:
                    room.release()
            with ThreadPoolExecutor(max_workers=12) as ex:
                futs = []
                for k in range(len(self.caps)):
                    room.acquire()
                    futs.append(ex.submit(ref_job, k, np.ascontiguousarray(self.iq_of(k))))
                for fu in futs:
                    k, sha, ts = fu.result()
                    self.iq_sha[k] = sha
                    refs[k] = ts
        else:
            out["checker"] = ("tests/golden/c1_ts.json (TS hashes recorded where the reference binary decoded the same IQ), else the transmitted packet "
                              "sequence (oracle/_ref/leandvb not present on this machine)")
        ok_all = True
        for k in range(len(self.caps)):
            got = self.ts_of(k)
            pk = [got[i:i + 188] for i in range(0, len(got), 188)]
            rep = dict(capture=k, seed=self.seeds[k], first_packet=self.first_pk[k], samples=self.n, ts_packets=len(pk),
                       ts_sha256=hashlib.sha256(got).hexdigest(), same_count_every_step=bool(len(set(self.counts[k])) <= 1))
            r0 = rec.get((self.n, self.seeds[k], self.first_pk[k]))
            if r0:      # this capture was decoded by the reference binary when tests/golden/c1_ts.json was recorded
                rep["recorded"] = dict(ts_sha256_equal=bool(r0["ts_sha256"] == rep["ts_sha256"]), ref_ts_sha256=r0["ref_ts_sha256"][:16],
                                       was_equal_to_reference_after_acquisition=bool(r0["equal_to_reference_after_acquisition"]))
            if refs[k] is not None:
                ref = refs[k]
                rpk = [ref[i:i + 188] for i in range(0, len(ref), 188)]
                tail = rpk[SKIP_ACQ:]
                rep["ref_packets"] = len(rpk)
                rep["ref_ts_sha256"] = hashlib.sha256(ref).hexdigest()
                rep["iq_sha256"] = self.iq_sha.get(k)
                rep["whole_ts_identical"] = bool(ref == got)
                ok = len(tail) > 100 and tail[0] in pk
                if ok:
                    i0 = pk.index(tail[0])
                    # the reference stops where its input ends; this path's last partial window may end a few packets earlier
                    m = min(len(tail), len(pk) - i0)
                    ok = pk[i0:i0 + m] == tail[:m] and len(tail) - m <= 16
                    rep["compared"] = m; rep["ref_tail_not_reached"] = len(tail) - m
                rep["equal_to_reference_after_acquisition"] = bool(ok)
            elif r0:
                ok = rep["recorded"]["ts_sha256_equal"] and rep["recorded"]["was_equal_to_reference_after_acquisition"]
                rep["equal_to_recorded_reference_checked_ts"] = bool(ok)
            else:
                sent = self.ts_sent
                first = [i for i in range(max(0, self.first_pk[k] - 64), min(len(sent), self.first_pk[k] + 4096)) if bytes(sent[i]) == pk[SKIP_ACQ]] if len(pk) > SKIP_ACQ else []
                ok = bool(first) and len(pk) > 100 and b"".join(pk[SKIP_ACQ:]) == sent[first[0]:first[0] + len(pk) - SKIP_ACQ].tobytes()
                rep["equal_to_transmitted_after_acquisition"] = bool(ok)
            rep["pass"] = bool(ok and rep["same_count_every_step"])
            ok_all = ok_all and rep["pass"]
            out["per_capture"].append(rep)
        out["checker_seconds"] = round(time.perf_counter() - t0, 2)
        out["pass"] = bool(ok_all)
        return out

    def close(self):
        self._close_engines()
        for p in self.h_ts:
            self.capi.lib.lsdr_free_host(p)
        for d in self.caps:
            d.free()
        self.ctx.close()

    def _close_engines(self):
        for w in self.workers:
            w.close()


class C1Job(ChainJob):
    """Round 6: lsdr_capture_batch — the captures of a group share every launch, counts stay on the device, one host thread."""

    def __init__(self, capi, device, n_captures, msamples, groups, tile, warm, seed0, anf=1, aux_cus=0):
        self.anf = 1 if anf else 0
        self.aux_cus = int(aux_cus)
        # auto_notch moves whole 4096-sample blocks: a capture of whole blocks is consumed entirely (but for the receiver's last chunk)
        n = (msamples << 20) if self.anf else (msamples << 20) // 128 * 128 + 1
        self._make_captures(capi, device, n_captures, n, seed0)
        groups = max(1, min(int(groups), n_captures))
        self.groups = []
        per = (n_captures + groups - 1) // groups
        for g in range(groups):
            ks = list(range(g * per, min(n_captures, (g + 1) * per)))
            if not ks:
                continue
            gctx = capi.Ctx(device)
            cb = capi.CaptureBatch(gctx, len(ks), self.n, OMEGA, fec=capi.FEC12, anf=self.anf, tile_len=tile, tile_warmup=warm, aux_cus=self.aux_cus)
            self.groups.append(dict(ctx=gctx, cb=cb, ks=ks, in_flight=False, ptrs=[self.caps[k].ptr for k in ks]))
        pk_cap = int(self.n / OMEGA / 8 / 204) + 512
        self._make_ts_buffers(pk_cap * 188)
        self.tile = (tile, warm)
        self.workers = []
        self.results = [None] * n_captures
        self.stats = dict(bits=0, errs=0, next_sync=0, jobs=0)

    def _retire(self, g, timed):
        res = g["cb"].wait()
        g["in_flight"] = False
        if not os.environ.get("LSDR_C1_NO_TS"):      # (diagnostic: what the job does without its TS going back to the host — never verified)
            g["cb"].ts_download_async([self.h_ts[k] for k in g["ks"]], self.ts_cap)
        tot = 0
        for k, r in zip(g["ks"], res):
            self.results[k] = r
            self.n_ts[k] = r["ts_packets"]
            if timed:
                self.counts[k].append(r["ts_packets"])
            tot += r["samples"]
            self.stats["errs"] += r["rs_bit_errors"]; self.stats["bits"] += r["rs_packets"] * 204 * 8
            self.stats["next_sync"] += r["next_sync_calls"]; self.stats["jobs"] += 1
        return tot

    def run(self, steps, timed=False):
        """`steps` decodes of every capture.  Returns the samples consumed.  One host thread: a group's batch is queued as soon as its
        previous one has been read back; the TS download of that one runs on a side stream under the new batch's kernels."""
        total = 0
        if timed:
            for g in self.groups:
                g["cb"].tile_time(True)
        for _ in range(steps):
            for g in self.groups:
                if g["in_flight"]:
                    total += self._retire(g, timed)
                g["cb"].run_async(g["ptrs"], self.n)
                g["in_flight"] = True
        for g in self.groups:
            if g["in_flight"]:
                total += self._retire(g, timed)
        for g in self.groups:
            g["cb"].ts_wait()
        return total

    def tile_kernel_ms(self):
        ms, n = 0.0, 0
        for g in self.groups:
            a, b = g["cb"].tile_time(False)
            ms += a * b; n += b
        return (ms / n if n else 0.0), n

    def _close_engines(self):
        for g in self.groups:
            g["cb"].close()
            g["ctx"].close()


def cpu_reference(job, budget_s=20.0):
    """The reference binary on the host cores: one process on a bounded head of capture 0, then one process per core on it
    (the reference is single-threaded; processes are how it scales), median of three passes."""
    if not os.path.exists(REFBIN):
        return None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(job.n, 16 << 20)
    iq = job.iq_of(0)[: 2 * n]
    f = tempfile.NamedTemporaryFile(prefix="lsdr_c1_cpu_", suffix=".u8", delete=False)
    iq.tofile(f); f.close()

    def run(np_):
        t0 = time.perf_counter()
        ps = [subprocess.Popen([REFBIN] + ref_args(job.anf), stdin=open(f.name, "rb"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(np_)]
        for p in ps:
            p.wait()
        return np_ * n / (time.perf_counter() - t0) / 1e6
    one = sorted(run(1) for _ in range(3))[1]
    allc = sorted(run(cores) for _ in range(3))[1] if cores > 1 else one
    os.unlink(f.name)
    return dict(value=round(allc, 3), unit="MS/s", cores=cores, kind="reference", one_core=round(one, 3),
                sample=f"oracle/_ref/leandvb {' '.join(ref_args(job.anf))} on the first {n} samples of capture 0 (whole chain to TS): 1 process, then "
                       f"{cores} processes (one per core); median of 3 passes each")


# VALU issue slots of one MI355X: 256 CUs × 4 SIMDs, one wave64 vector instruction per 2 cycles and SIMD (SIMD-32:
# /opt/skills/guides/MI355X_MICROARCH.md "Wave scheduling"; measured 0.94 ns per v_add_f32 and SIMD, profiles/NOTES.md) at the 2.4 GHz peak clock
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0


def run_workload(capi, device, args, shard):
    """bench.py --workload c1.  Returns the JSON object of rank 0 (None on the other ranks)."""
    rank, world = shard.rank, shard.world
    mode = getattr(args, "c1_mode", "batch")
    anf = int(getattr(args, "c1_anf", 1))
    seed0 = capture_seeds(rank, 1)[0]
    # The captures' TS goes back over PCIe into pinned host memory (412 MB per step at 32 captures = 29 GB/s at 300 GS/s): the job's buffers are allocated
    # while the process sits on the CPUs next to its GPU, so that they are pinned on that NUMA node — boxes where `c1` came out at 260–267 instead of
    # 300–308 GS/s had the tile kernel at the same 5.7 ms and 2 ms more per step outside it.  (With several ranks bench.py has pinned the rank already; with one, the
    # affinity is given back afterwards: the CPU baseline uses every core.)
    restore = None
    if world == 1 and hasattr(os, "sched_getaffinity"):
        restore = os.sched_getaffinity(0)
        shard.pin_to_gpu_numa(capi.device_pci_bus_id(device))
    if mode == "chain":
        job = ChainJob(capi, device, args.c1_captures, args.c1_msamples, args.c1_workers, args.c1_tile or 2048, args.c1_warmup, seed0=seed0)
    else:
        job = C1Job(capi, device, args.c1_captures, args.c1_msamples, args.c1_groups, args.c1_tile or 4096, args.c1_warmup, seed0=seed0, anf=anf,
                    aux_cus=getattr(args, "c1_aux_cus", 0))
    if restore is not None:
        os.sched_setaffinity(0, restore)
    job.run(max(1, args.warmup))
    shard.barrier()
    t0 = time.perf_counter()
    consumed = job.run(args.steps, timed=True)
    shard.barrier()
    dt = time.perf_counter() - t0
    total, dt, _ = shard.aggregate(consumed, dt)
    # every rank checks the TS of its own captures and contributes its verdict; rank 0 reports
    ver = None
    if not args.no_verify:
        ver = job.verify()
        ranks_ok, ranks = shard.all_ranks_ok(ver["pass"])
        if not ver["pass"]:
            print(f"bench.py: rank {rank}: C1 VERIFICATION FAILED: " + str(ver)[:3000], file=sys.stderr)
    if rank != 0:
        job.close()
        return None, (3 if ver is not None and not ver["pass"] else 0)
    kms, klaunch = job.tile_kernel_ms()
    graph = ("cconverter<u8> -> auto_notch(1) -> cstln_receiver(linear) -> deconvol_sync -> mpeg_sync -> deinterleaver -> rs_decoder -> derandomizer"
             if job.anf else "cconverter<u8> -> cstln_receiver(linear) -> deconvol_sync -> mpeg_sync -> deinterleaver -> rs_decoder -> derandomizer")
    out = {
        "metric": "IQ MSamples/s demodulated (leandvb QPSK 1/2)", "value": round(total / dt / 1e6, 3), "unit": "MS/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 1 / config 4: independent QPSK 1/2 captures, cu8 IQ at 1.2 samples/symbol (leandvb " + " ".join(ref_args(job.anf)) +
                               "), each decoded from its first sample to TS packets in host memory: " + graph,
                   "captures_per_gpu": len(job.caps), "samples_per_capture": job.n, "samples_per_step_per_gpu": job.n * len(job.caps),
                   "rx_tile": {"tile_len": job.tile[0], "warmup": job.tile[1]},
                   "parallelism": f"{world * len(job.caps)} independent capture(s), {len(job.caps)} per GPU, no collectives, no RCCL",
                   "ts_packets_per_capture": job.n_ts[0]},
    }
    if mode == "chain":
        W = len(job.workers)
        alg = job.n * ALG_BYTES_PER_SAMPLE
        out["config"]["engine"] = f"C-ABI blocks one by one, {W} host threads / streams per GPU (rounds 2-5)"
        out["config"]["host_seconds_per_stage_summed_over_workers"] = {k: round(sum(w.t_stage[k] for w in job.workers), 3) for k in job.workers[0].t_stage}
        out["config"]["rs_byte_errors_corrected"] = sum(w.stats["errs"] for w in job.workers)
        out["config"]["deconv_next_sync_calls"] = sum(w.stats["next_sync"] for w in job.workers)
        out["roofline"] = {"kernel": "k_rx_tiles<linear, arithmetic QPSK, cu8, LDS-staged, packed> (cstln_receiver tolerance tiles)", "bound": "hbm",
                           "achieved": round(alg / (kms * 1e-3) / 1e9, 2) if kms else None, "peak": 8000.0, "unit": "GB/s",
                           "frac": round(alg / (kms * 1e-3) / 1e9 / 8000.0, 5) if kms else None, "avg_launch_ms": round(kms, 4), "launches_timed": klaunch,
                           "algorithmic_bytes_per_launch": int(alg), "algorithmic_bytes_per_sample": round(ALG_BYTES_PER_SAMPLE, 4), "traffic": None,
                           "concurrent_launches": f"up to {W} (one per worker stream): a launch shares the chip, its duration is not the chip's rate",
                           "whole_job_frac": round(total / dt * ALG_BYTES_PER_SAMPLE / 1e9 / 8000.0 / world, 5)}
    else:
        per_launch = len(job.groups[0]["ks"])
        alg = job.n * ALG_BYTES_PER_SAMPLE * per_launch
        tiles = job.results[0]["tiles"] if job.results[0] else 0
        # symbol steps a launch executes: every tile walks its warm-up and its body (tile 0 of a capture is one lane of exact arithmetic)
        sym_steps = per_launch * max(tiles - 1, 0) * (job.tile[0] + job.tile[1]) / OMEGA
        ipss = float(os.environ.get("LSDR_C1_VALU_PER_SYMBOL", VALU_PER_SYMBOL_STEP[1 if job.anf else 0]))
        wave_inst = sym_steps / 64.0 * ipss
        out["config"]["engine"] = (f"lsdr_capture_batch: {len(job.groups)} group(s) of {per_launch} capture(s) per GPU, shared launches, counts on the device, "
                                   "ONE host thread, default hardware queues")
        out["config"]["rs_bit_errors_corrected"] = job.stats["errs"]
        out["config"]["deconv_next_sync_calls"] = job.stats["next_sync"]
        out["config"]["receiver_seams"] = {k: sum(r[k] for r in job.results if r) for k in ("seam_dup", "seam_miss", "seam_bad")}
        out["roofline"] = {"kernel": f"k_rxb_tiles<notch={bool(job.anf)}> (capture-batch tolerance tiles: cu8 LDS-staged, notch in the sample walk, QPSK by arithmetic, packed decisions)",
                           "bound": "hbm", "achieved": round(alg / (kms * 1e-3) / 1e9, 2) if kms else None, "peak": 8000.0, "unit": "GB/s",
                           "frac": round(alg / (kms * 1e-3) / 1e9 / 8000.0, 5) if kms else None, "avg_launch_ms": round(kms, 4), "launches_timed": klaunch,
                           "algorithmic_bytes_per_launch": int(alg), "algorithmic_bytes_per_sample": round(ALG_BYTES_PER_SAMPLE, 4), "traffic": None,
                           "captures_per_launch": per_launch,
                           "concurrent_launches": f"{len(job.groups)} (one per group stream): a launch shares the chip with the other group's launches",
                           "whole_job_frac": round(total / dt * ALG_BYTES_PER_SAMPLE / 1e9 / 8000.0 / world, 5),
                           # the bound that applies: vector-instruction issue (a decision-feedback recurrence per tile)
                           "valu_issue": {"bound": "valu issue", "symbol_steps_per_launch": int(sym_steps), "valu_instructions_per_symbol_step": ipss,
                                          "instructions_source": "profiles/r06_bench/c1_pmc_sq.txt (SQ_INSTS_VALU per launch ÷ symbol steps ÷ 64 lanes)",
                                          "wave_instructions_per_launch": int(wave_inst), "peak_wave_instructions_per_s": VALU_ISSUE_PEAK,
                                          "achieved_wave_instructions_per_s": round(wave_inst / (kms * 1e-3), 1) if kms else None,
                                          "frac": round(wave_inst / (kms * 1e-3) / VALU_ISSUE_PEAK, 4) if kms else None,
                                          "whole_job_frac": round(total / dt / world / (job.n * per_launch) * wave_inst / VALU_ISSUE_PEAK, 4)}}
    rc = 0
    if ver is not None:
        out["verified"] = ver
        out["verified"]["ranks_passed"], out["verified"]["ranks"] = ranks_ok, ranks
        out["verified"]["pass"] = bool(ver["pass"] and ranks_ok == ranks)
        if not out["verified"]["pass"]:
            rc = 3
    if world == 1 and args.c1_captures > 1 and not getattr(args, "no_single", False) and mode != "chain":
        # BASELINE config 4 as written — ONE capture per GPU: a latency figure (ms from the first sample to the last TS packet of a
        # capture), not the GPU's rate
        one = C1Job(capi, device, 1, args.c1_msamples, 1, args.c1_tile or 4096, args.c1_warmup, seed0=seed0, anf=anf)
        one.run(1)
        t1 = time.perf_counter()
        reps = 3
        n1 = one.run(reps, timed=True)
        d1 = time.perf_counter() - t1
        out["config4_one_capture_per_gpu"] = {"ms_per_capture": round(d1 / reps * 1e3, 3), "value": round(n1 / d1 / 1e6, 3), "unit": "MS/s",
                                              "samples_per_capture": one.n, "note": "--c1-captures 1: per-capture latency"}
        one.close()
    if world == 1 and not args.no_cpu:
        cpu = cpu_reference(job, args.cpu_seconds)
        if cpu:
            out["cpu_baseline"] = cpu
    job.close()
    return out, rc
