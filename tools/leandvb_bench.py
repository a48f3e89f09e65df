#!/usr/bin/env python3
"""tools/leandvb_bench.py — the sensitivity benchmark of the reference (test/leandvb_bench.sh) with every stage on the GPU:
    TS counter pattern | leandvbtx_amd -f RATIO --power P --agc | leanchansim_amd --awgn N --deterministic [--ou8]
                       | ref_graph/leandvb --f32 --float-scale S -f FS --sr 1e6 --anf 0 --fd-info 2 FLAGS
It parses the LOCK / VBER / CNR / SS / MER / LOCKTIME lines the same way (min/max VBER from the last lock until LOCKTIME
reaches MINPACKETS) and prints one row per SNR: ratio rxsnr cnr ss mer vbermin vbermax.
    python tools/leandvb_bench.py [--ref] [--packets N] [--min-packets M] [series ...]
--ref runs the reference binaries of oracle/_ref instead (build container only).  Because every block is bit-exact and the
noise is the reference's deterministic drand48 stream, both print the same rows."""
import math, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leansdr_amd import synth_dvbs

SERIES = {   # name: (ratio, SNRs, receiver flags) — test/leandvb_bench.sh:119-134
    "1.2sps-hs": ("6/5", [20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10], "--u8 --hs"),
    "1.2sps": ("6/5", [22, 21, 20, 19, 18, 17, 16, 15], ""),
    "4sps-viterbi-rrc": ("4", [6.5, 6.0, 5.5, 5.0, 4.5], "--viterbi --sampler rrc"),
    "4.2sps": ("21/5", [20, 19, 18, 17, 16, 15, 14], ""),
    "1.2sps-viterbi": ("6/5", [12, 11, 10.5, 10, 9.5, 9, 8.5], "--viterbi"),
    "2.4sps-viterbi-rrc": ("12/5", [8, 7, 6, 5.8, 5.6, 5.4, 5.2, 5.0, 4.8], "--viterbi --sampler rrc"),
}


RX_EXTRA = ""   # extra leandvb options (--rx-extra "--buf-factor 4": the reference's pipe sizes, hence its report cadence)
ANF_ARG = "--anf 0"   # test/leandvb_bench.sh runs without the notch; "" = leandvb's default (--anf 1)
RX_ENV = {}     # environment of the receiver process (LSDR_TILED=1, LSDR_FIR_ARITH=blk, LSDR_FUSE_NOTCH=1: the throughput modes)


def commands(ratio, snr, flags, ref=False):
    """The three command lines of one run (leandvb_bench.sh:20-56).  ref: False = this repo's apps, True = the reference
    binaries (oracle/_ref), "graph" = the reference's own app SOURCES compiled unchanged against this repo's host headers
    (leansdr_amd/host/ref_graph: every block a GPU block)."""
    num, _, den = ratio.partition("/")
    r = float(num) / float(den or 1)
    symbrate = 1000000
    samprate = int(symbrate * r)
    hs = flags == "--u8 --hs"
    if hs:      # the receiver gain is expected to put the u8 modulation amplitude at cstln_amp
        sigpow, noisepow, scale = 37.5, 37.5 - snr, None
    else:       # fixed noise floor, display scale adjusted
        sigpow, noisepow, scale = snr, 0, 10 * math.sqrt(r)
    if ref == "graph":
        d = os.path.join(ROOT, "leansdr_amd", "host", "ref_graph")
        tx, ch, rx = f"{d}/leandvbtx", f"{d}/leanchansim", f"{d}/leandvb"
    elif ref:
        d = os.path.join(ROOT, "oracle", "_ref")
        tx, ch, rx = f"{d}/leandvbtx", f"{d}/leanchansim", f"{d}/leandvb"
    else:
        d = os.path.join(ROOT, "leansdr_amd", "host", "apps")      # this repo's own generator-side builders; the receiver is always the
        tx, ch, rx = f"{d}/leandvbtx_amd", f"{d}/leanchansim_amd", os.path.join(ROOT, "leansdr_amd", "host", "ref_graph", "leandvb")   # reference's source
    cnr = "--cnr" if samprate > 3 * symbrate else ""
    c_tx = f"{tx} -f {ratio} --power {sigpow:g} --agc"
    c_ch = f"{ch} --awgn {noisepow:g} --deterministic {'--ou8' if hs else ''}"
    c_rx = (f"{rx} {'' if hs else f'--f32 --float-scale {scale:.10f}'} -f {samprate} --sr {symbrate} {ANF_ARG} {cnr} --fd-info 2 {flags} {'' if ref is True else RX_EXTRA}")
    return c_tx, c_ch, c_rx, sigpow - noisepow


def run_pipeline(ratio, snr, flags, npackets, ref=False):
    """Returns (info text, TS bytes).  The receiver reads the channel output from a file: its reports (and which SS/MER line
    is the last one before LOCKTIME reaches the threshold) depend on how stdin is cut into reads, and a file gives both
    implementations the same full-pipe reads."""
    import tempfile
    c_tx, c_ch, c_rx, _ = commands(ratio, snr, flags, ref)
    ts = synth_dvbs.ts_packets(npackets).tobytes()
    with tempfile.NamedTemporaryFile(suffix=".iq") as f:
        p = subprocess.run(f"{c_tx} | {c_ch} > {f.name}", shell=True, input=ts, stderr=subprocess.PIPE)
        if p.returncode:
            raise RuntimeError(p.stderr.decode()[-2000:])
        env = dict(os.environ)
        if ref is not True:
            env.update(RX_ENV)
        p = subprocess.run(f"{c_rx} < {f.name}", shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    if p.returncode:
        raise RuntimeError(p.stderr.decode()[-2000:])
    return p.stderr.decode(), p.stdout


def parse_info(text, min_packets):
    """leandvb_bench.sh:58-91."""
    vmin, vmax, cnr, ss, mer = 1000000, 0, 0.0, 0.0, 0.0
    for line in text.splitlines():
        k, _, arg = line.partition(" ")
        if k == "LOCK" and arg.strip() == "0":
            vmin, vmax, cnr, ss, mer = 1000000, 0, 0.0, 0.0, 0.0
        elif k == "VBER":
            v = int(float(arg) * 1000000)
            vmin, vmax = min(vmin, v), max(vmax, v)
        elif k == "CNR":
            cnr = float(arg)
        elif k == "SS":
            ss = float(arg)
        elif k == "MER":
            mer = float(arg)
        elif k == "LOCKTIME" and int(arg) >= min_packets:
            return dict(cnr=cnr, ss=ss, mer=mer, vbermin=vmin * 1e-6, vbermax=vmax * 1e-6)
    return None


if __name__ == "__main__":
    args = sys.argv[1:]
    ref = "--ref" in args
    npk = int(args[args.index("--packets") + 1]) if "--packets" in args else 3000
    minpk = int(args[args.index("--min-packets") + 1]) if "--min-packets" in args else 1000
    if "--rx-extra" in args:
        RX_EXTRA = args[args.index("--rx-extra") + 1]
    names = [a for a in args if a in SERIES] or list(SERIES)
    for name in names:
        ratio, snrs, flags = SERIES[name]
        print(f"# {name}.")
        for snr in snrs:
            text, _ = run_pipeline(ratio, snr, flags, npk, ref)
            r = parse_info(text, minpk)
            rr = eval(ratio) if "/" in ratio else float(ratio)
            rxsnr = commands(ratio, snr, flags)[3]
            print(f"{'ref' if ref else 'mi355x'} {rr:.2f} {rxsnr:.2f} " + ("no-lock" if r is None else
                  f"{r['cnr']:g} {r['ss']:g} {r['mer']:g} {r['vbermin']:.6f} {r['vbermax']:.6f}"), flush=True)
