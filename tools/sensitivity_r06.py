#!/usr/bin/env python3
"""tools/sensitivity_r06.py — round 6's sensitivity record: the reference's benchmark (test/leandvb_bench.sh, via tools/leandvb_bench.py)
for every mode bench.py times, next to the reference BINARIES' rows on the same deterministic inputs, written to gpurun_out/r06_sens/
(copied to profiles/r06_sensitivity/).  Run on the GPU box (oracle/_ref travels with the repo).

  rows_<series>.txt    one block per implementation: ref (oracle/_ref binaries), exact (ref_graph, exact modes), and the throughput
                       mode(s) the series exists for: tiled (LSDR_TILED=1: time-tiled receivers, scan notch), blk+tiled (LSDR_FIR_ARITH=blk),
                       fused (LSDR_FUSE_NOTCH=1)
  capture_batch.txt    lsdr_capture_batch (bench.py --workload c1's engine) against the reference binary at its defaults, per SNR:
                       TS packets, packets identical to the transmitted ones, RS-corrected bits per bit (the VBER the reference prints)
"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import leandvb_bench as lb
from leansdr_amd import synth_dvbs

OUT = os.path.join(ROOT, "gpurun_out", "r06_sens")
os.makedirs(OUT, exist_ok=True)

TILED = {"LSDR_TILED": "1"}
PLAN = [
    # name, (ratio, SNRs, flags), anf arg, packets, min packets, [(label, ref flag, env, rx extra)]
    ("1.2sps", lb.SERIES["1.2sps"], "--anf 0", 1500, 500, [("tiled", "graph", TILED, "--buf-factor 64")]),
    ("4sps-viterbi-rrc", lb.SERIES["4sps-viterbi-rrc"], "--anf 0", 1500, 500, [("tiled", "graph", TILED, "--buf-factor 64")]),
    ("1.2sps-hs", lb.SERIES["1.2sps-hs"], "--anf 0", 1500, 500, [("tiled", "graph", TILED, "--buf-factor 64")]),
    ("4.2sps", lb.SERIES["4.2sps"], "--anf 0", 1500, 500, [("tiled", "graph", TILED, "--buf-factor 64")]),
    ("1.2sps-anf1", ("6/5", [22, 20, 18, 17, 16, 15], ""), "", 4000, 3000, [("tiled+scan-notch", "graph", TILED, "--buf-factor 64")]),
    ("40sps-resample", ("40", [20, 17, 14, 12, 11, 10, 9], "--resample"), "--anf 0", 1000, 400,
     [("tiled", "graph", TILED, "--buf-factor 256"), ("blk", "graph", dict(LSDR_FIR_ARITH="blk"), "--buf-factor 256"),
      ("blk+tiled", "graph", dict(TILED, LSDR_FIR_ARITH="blk"), "--buf-factor 256")]),
    ("120sps-resample-anf1", ("120", [20, 17], "--resample"), "", 700, 250,
     [("tiled", "graph", TILED, "--buf-factor 256"), ("fused-notch", "graph", dict(LSDR_FUSE_NOTCH="1"), "--buf-factor 256"),
      ("fused-notch+tiled", "graph", dict(TILED, LSDR_FUSE_NOTCH="1"), "--buf-factor 256")]),
]


def rows(name, series, anf, npk, minpk, label, ref, env, extra):
    ratio, snrs, flags = series
    lb.ANF_ARG, lb.RX_ENV, lb.RX_EXTRA = anf, env, extra
    out = []
    for snr in snrs:
        t0 = time.time()
        try:
            text, ts = lb.run_pipeline(ratio, snr, flags, npk, ref)
            r = lb.parse_info(text, minpk)
        except Exception as e:          # a run that fails is a row that says so
            out.append(f"{label} {snr:.2f} FAILED {str(e)[-200:]!r}")
            continue
        rr = eval(ratio) if "/" in ratio else float(ratio)
        rxsnr = lb.commands(ratio, snr, flags)[3]
        npk_out = len(ts) // 188
        out.append(f"{label} {rr:.2f} {rxsnr:.2f} " + ("no-lock" if r is None else f"{r['cnr']:g} {r['ss']:g} {r['mer']:g} {r['vbermin']:.6f} {r['vbermax']:.6f}")
                   + f" ts_packets {npk_out} ({time.time() - t0:.1f} s)")
        print(name, out[-1], flush=True)
    return out


def series_files(which):
    for name, series, anf, npk, minpk, modes in PLAN:
        if which and name not in which:
            continue
        with open(os.path.join(OUT, f"rows_{name}.txt"), "w") as f:
            f.write(f"# {name}: leandvbtx -f {series[0]} | leanchansim --awgn … --deterministic | leandvb {series[2]} {anf} --fd-info 2; {npk} packets, "
                    f"VBER window closed at LOCKTIME >= {minpk}\n# columns: implementation, samples/symbol, SNR, CNR, SS, MER, min VBER, max VBER, TS packets written\n")
            for label, ref, env, extra in [("ref", True, {}, ""), ("exact", "graph", {}, "")] + modes:
                for line in rows(name, series, anf, npk, minpk, label, ref, env, extra):
                    f.write(line + "\n")
                f.flush()


def capture_batch_rows():
    import leansdr_amd.capi as capi
    import bench_c1
    ctx = capi.Ctx(0)
    npk = 6000
    ts_in = synth_dvbs.ts_packets(npk)
    sent = {bytes(ts_in[i]): i for i in range(npk)}
    refbin = os.path.join(ROOT, "oracle", "_ref", "leandvb")
    with open(os.path.join(OUT, "capture_batch.txt"), "w") as f:
        f.write("# lsdr_capture_batch (leandvb's default --u8 graph, anf 1) vs oracle/_ref/leandvb --u8 -f 1200000 --sr 1000000 on the same file:\n"
                "# leandvbtx -f 6/5 --power 37.5 --agc | leanchansim --awgn (37.5 - SNR) --deterministic --ou8, 6000 packets\n"
                "# columns: engine, tile/warm-up, SNR, TS packets, of them identical to a transmitted packet, RS bits corrected per bit\n")
        for snr in (20, 18, 17, 16, 15, 14, 13, 12, 11):
            c_tx, c_ch, _, _ = lb.commands("6/5", snr, "--u8 --hs", "graph")
            with tempfile.NamedTemporaryFile(suffix=".u8") as g:
                p = subprocess.run(f"{c_tx} | {c_ch} > {g.name}", shell=True, input=ts_in.tobytes(), stderr=subprocess.PIPE)
                assert p.returncode == 0, p.stderr.decode()[-500:]
                iq = np.fromfile(g.name, np.uint8)
                p = subprocess.run(f"{refbin} --u8 -f 1200000 --sr 1000000 --fd-info 2 < {g.name}", shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                ref_ts, info = p.stdout, p.stderr.decode()
            vb = [float(l.split()[1]) for l in info.splitlines() if l.startswith("VBER ")]
            rpk = [ref_ts[i:i + 188] for i in range(0, len(ref_ts), 188)]
            f.write(f"ref - {snr} {len(rpk)} {sum(1 for q in rpk if q in sent)} {np.mean(vb) if vb else float('nan'):.6f}\n")
            n = len(iq) // 2 // 4096 * 4096
            buf = ctx.upload(iq[: 2 * n])
            for tile, warm in ((4096, 512), (4096, 384), (4096, 256), (2048, 512)):
                cb = capi.CaptureBatch(ctx, 1, n, 1.2, anf=1, tile_len=tile, tile_warmup=warm)
                res, ts = cb.decode([buf.ptr], n)
                pk = [ts[0][i:i + 188] for i in range(0, len(ts[0]), 188)]
                r = res[0]
                vber = r["rs_bit_errors"] / max(1, r["rs_packets"] * 204 * 8)
                same_as_ref = ts[0] == ref_ts
                f.write(f"capture_batch {tile}/{warm} {snr} {len(pk)} {sum(1 for q in pk if q in sent)} {vber:.6f} ts_equals_reference {same_as_ref} "
                        f"seams dup {r['seam_dup']} miss {r['seam_miss']} bad {r['seam_bad']}\n")
                f.flush()
                cb.close()
            buf.free()
            print("capture_batch", snr, "done", flush=True)
    ctx.close()


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--no-series" not in sys.argv:
        series_files(which)
    if "--no-batch" not in sys.argv and (not which or "capture_batch" in which):
        capture_batch_rows()
