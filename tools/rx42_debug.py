"""Debug: the tiled receiver on the 4.2 sps series of tools/leandvb_bench.py (no-lock in profiles/r06_sensitivity).  GPU box."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import leansdr_amd.capi as capi
import pyoracle as po
import leandvb_bench as lb
from leansdr_amd import synth_dvbs
ratio = sys.argv[1] if len(sys.argv) > 1 else "21/5"
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 20
c_tx, c_ch, c_rx, _ = lb.commands(ratio, snr, "", "graph")
ts = synth_dvbs.ts_packets(600).tobytes()
with tempfile.NamedTemporaryFile(suffix=".iq") as f:
    subprocess.run(f"{c_tx} | {c_ch} > {f.name}", shell=True, input=ts, check=True)
    x = np.fromfile(f.name, np.complex64)
num, _, den = ratio.partition("/"); r = float(num) / float(den or 1)
x = (x * np.float32(10 * np.sqrt(r))).astype(np.complex64)
ppm = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # the receiver is told a symbol rate this far off the stream's
omega = float(np.float32(r * (1 + ppm * 1e-6)))
print("samples", len(x), "omega", omega, "rms", np.sqrt(np.mean(np.abs(x) ** 2)))
O = po.Oracle(); ctx = capi.Ctx(0)
md = int(1e6 * r / 5)
ref = O.rx(po.rx_params(sampler=1, cstln=1, omega=omega, meas_decimation=md), x)
print("oracle symbols", len(ref["sym"]), "consumed", ref["consumed"])
for run_len in (len(x), 262144):
    for tl, tw in ((0, 0), (1024, 512)):
        rx = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=omega, meas_decimation=md, mode=capi.RX_TILED, tile_len=tl, tile_warmup=tw)
        syms = []; pos = 0; bad = 0
        while pos + 129 <= len(x):
            chunk = x[pos:pos + run_len]
            out = rx.run(chunk)
            if not out["consumed"]:
                break
            syms.append(out["sym"]); pos += out["consumed"]; bad += rx.tiled_stats()["bad_seams"]
        s = np.concatenate(syms)
        n = min(len(s), len(ref["sym"]))
        eq = (s["symbol"][:n] == ref["sym"]["symbol"][:n])
        # first index where a window of 200 has < 90% agreement
        w = 200; first = None
        for i in range(0, n - w, w):
            if eq[i:i + w].mean() < 0.9:
                first = i; break
        print(f"run_len {run_len} tile {tl}/{tw}: symbols {len(s)} (ref {len(ref['sym'])}) equal {eq.mean():.4f} first bad window at symbol {first} bad_seams {bad}")
        rx.close()
