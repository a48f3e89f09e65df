#!/usr/bin/env python3
"""tools/notch_fir_debug.py — where the fused auto_notch + fir_filter block differs from the oracle chain (error profile along the stream)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import leansdr_amd.capi as capi
import pyoracle as po
from test_gpu_notch_fir import signal, oracle_chain, c2_taps, DECIM, N_TAPS

O = po.Oracle()
ctx = capi.Ctx(0)
c = c2_taps(capi)
for name, dec, n, tones, step in (("pass_through_only", 4096 * 1024, 4096 * 20, [(0, 0.0031, 40.0)], None),
                                  ("positive_bin", 4096 * 8, 4096 * 60 + 1234, [(0, 0.0031, 40.0)], None),
                                  ("bin_changes", 4096 * 8, 4096 * 60 + 1234, [(0, 0.0031, 40.0), (4096 * 21 + 100, -0.0021, 55.0), (4096 * 43, 0.0007, 30.0)], None),
                                  ("small_runs", 4096 * 4, 4096 * 50, [(0, 0.0024, 35.0), (4096 * 17 + 2000, -0.0035, 45.0), (4096 * 34, 0.0024, 35.0)], 4096 + 400)):
    x = signal(n, 3, tones)
    ref, ref_bin, n_notched = oracle_chain(O, x, c, dec)
    nf = capi.NotchFir(ctx, c, DECIM, decimation=dec)
    y, cons = nf.run(x, step=step)
    m = min(len(y), len(ref))
    e = np.abs(y[:m].astype(np.complex128) - ref[:m]) / np.abs(ref).max()
    print(f"{name}: outputs {len(y)} (oracle {len(ref)}), bin {nf.bin()} (oracle {ref_bin}), max rel err {e.max():.3g} at output {int(e.argmax())} "
          f"(sample {N_TAPS + DECIM * int(e.argmax())}, block {(N_TAPS + DECIM * int(e.argmax())) / 4096:.2f})")
    per = dec // DECIM
    prof = [f"{e[i:i + per].max():.1e}" for i in range(0, m, per)]
    print("   per detect interval:", " ".join(prof[:24]))
    bad = np.flatnonzero(e > 5e-5)
    if len(bad):
        runs = np.split(bad, np.flatnonzero(np.diff(bad) > 1) + 1)
        print("   bad ranges (outputs):", [(int(r[0]), int(r[-1])) for r in runs[:12]], "tile =", 117)
    nf.close()
