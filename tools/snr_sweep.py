#!/usr/bin/env python3
"""tools/snr_sweep.py — TS yield of leandvb_amd vs noise level, exact serial receiver vs tiled receiver, deconvol_sync vs viterbi_sync.
A packet counts as good when it is byte-identical to a generated packet (looked up by its counter field)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leansdr_amd import synth_dvbs
npk = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
extra = [a for a in sys.argv[2:] if a != "--tiled-only"]
tiled_only = "--tiled-only" in sys.argv
sps4 = "--sps4" in sys.argv   # 4 samples/symbol (the C2 receiver geometry) instead of 1.2
extra = [a for a in extra if a != "--sps4"]
app = os.path.join(ROOT, "leansdr_amd", "host", "apps", "leandvb_amd")
for noise in (7.5, 15.0, 20.0, 25.0, 30.0):
    iq, ts = synth_dvbs.capture_u8(n_packets=npk, seed=11, noise_std=noise, **(dict(sps_num=4, sps_den=1) if sps4 else {}))
    truth = {bytes(p[1:4]): bytes(p) for p in ts}
    path = "/tmp/snr.u8"
    iq.tofile(path)
    line = f"noise_std {noise:5.1f} (Es/N0 ≈ {20*np.log10(75/(noise*np.sqrt(2)))+10*np.log10(1.2):4.1f} dB): "
    for flags in ([["--tiled"], ["--tiled", "--viterbi"]] if tiled_only else [[], ["--tiled"], ["--viterbi"], ["--tiled", "--viterbi"]]):
        cmd = [app, "--u8", "-f", "8000e3" if sps4 else "2400e3", "--sr", "2000e3", "--cr", "1/2"] + flags + extra
        t0 = time.perf_counter()
        with open(path, "rb") as f:
            p = subprocess.run(cmd, stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        out = np.frombuffer(p.stdout, np.uint8).reshape(-1, 188)
        good = sum(1 for q in out if truth.get(bytes(q[1:4])) == bytes(q))
        line += f"[{' '.join(flags) or 'serial'}: {len(out)} pk, {good} good, {dt:.2f}s] "
    print(line, flush=True)
