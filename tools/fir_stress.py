#!/usr/bin/env python3
"""tools/fir_stress.py [iterations] [captures] — hunts for rare non-reproducible fir_filter outputs in the bench pipeline: the
endless stream is B-periodic, so EVERY batch of a capture must produce the same decimated stream, bit for bit, as the oracle.
After each burst of queued batches all decimated-stream buffers of all captures are compared with the oracle's result."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
import leansdr_amd.capi as capi
from leansdr_amd import synth
import pyoracle as po
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ncap = int(sys.argv[2]) if len(sys.argv) > 2 else 3
O = po.Oracle()
pipe = bench.C2Pipeline(capi, synth, 0, ncap, 64, 4, (128, 256), seed0=1)
g = pipe.geo
refs = []
for cp in pipe.caps:
    x_full = np.concatenate([np.tile(cp.x, g["reps"]), cp.x[:bench.EXTRA * g["decim"] + g["N"]]])
    y = O.fir_filter(pipe.coeffs, g["decim"], O.scaler(75.0, x_full))[0]
    refs.append(y.view(np.uint64).copy())
bad = 0
for it in range(iters):
    pipe.run(24, True)
    pipe.sync()
    for c, cp in enumerate(pipe.caps):
        for i, d in enumerate(cp.dec):
            y = pipe.ctx.download(d, np.complex64, g["n_out"] + bench.EXTRA).view(np.uint64)
            if not np.array_equal(y, refs[c]):
                w = np.flatnonzero(y != refs[c])
                bad += 1
                print(f"iteration {it} capture {c} buffer {i}: {len(w)} outputs differ, first {w[0]} last {w[-1]}; tiles {w[0] // 256}..{w[-1] // 256}", flush=True)
print(f"{iters} iterations x {ncap} captures x {g['nbuf']} buffers: {bad} mismatching buffers")
