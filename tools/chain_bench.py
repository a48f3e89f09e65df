#!/usr/bin/env python3
"""tools/chain_bench.py — end-to-end leandvb (the reference's source on this repo's host framework + all GPU blocks) on a synthetic DVB-S capture.
Writes the capture to /tmp, runs the app (optionally under rocprofv3), reports MS/s and checks the TS payload."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leansdr_amd import synth_dvbs
npk = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
flags = sys.argv[2:]
repeat = 1
if "--repeat" in flags:      # feed the capture N times back to back (steady-state throughput; the joints cost a re-lock)
    i = flags.index("--repeat")
    repeat = int(flags[i + 1])
    del flags[i:i + 2]
path = f"/tmp/cap_{npk}.u8"
if not os.path.exists(path):
    chunks = []
    # build in pieces of 1000 packets to bound memory; continuity across pieces is not needed for a throughput run
    iq, ts = synth_dvbs.capture_u8(n_packets=npk, seed=7)
    iq.tofile(path)
if repeat > 1:
    rpath = f"/tmp/cap_{npk}x{repeat}.u8"
    if not os.path.exists(rpath):
        data = open(path, "rb").read()
        with open(rpath, "wb") as f:
            for _ in range(repeat):
                f.write(data)
    path = rpath
n = os.path.getsize(path) // 2
which = "--ref-graph"      # default: the reference's leandvb.cc on this repo's headers (GPU blocks)
for w in ("--ref-graph", "--ref-cpu"):      # the reference's leandvb.cc on this repo's headers (GPU blocks) / the reference's CPU binary
    if w in flags:
        flags.remove(w); which = w
env = dict(os.environ)
if which == "--ref-graph":
    app = os.path.join(ROOT, "leansdr_amd", "host", "ref_graph", "leandvb")
    if "--tiled" in flags:
        flags.remove("--tiled"); env["LSDR_TILED"] = "1"
    flags += ["--buf-factor", "4096"]
elif which == "--ref-cpu":
    app = os.path.join(ROOT, "oracle", "_ref", "leandvb")
    flags = [f for f in flags if f != "--tiled"]
cmd = [app, "--u8", "-f", "2400e3", "--sr", "2000e3", "--cr", "1/2"] + flags
t0 = time.perf_counter()
with open(path, "rb") as f:
    p = subprocess.run(cmd, stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
dt = time.perf_counter() - t0
ts = np.frombuffer(p.stdout, np.uint8).reshape(-1, 188)
cnt = ts[:, 1].astype(int) * 65536 + ts[:, 2].astype(int) * 256 + ts[:, 3] if len(ts) else np.array([])
print(f"[{which}] {' '.join(flags) or '(default)'}: {n} samples in {dt:.3f} s = {n/dt/1e6:.1f} MS/s wall (incl. process start, file read, H2D/D2H); "
      f"{len(ts)} TS packets, {int((np.diff(cnt)==1).sum()) if len(cnt)>1 else 0} consecutive; rc={p.returncode} {p.stderr.decode()[-200:]}")
