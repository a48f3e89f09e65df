#!/usr/bin/env python3
"""tools/notch_poison_stress.py NSLOTS — stress test of k_notch_scan's cross-workgroup hand-off (value → vmcnt(0) → stamped flag,
bounded spins), on the MEASURE build of the library (LSDR_HIP_LIB=tools/variants/liblsdr_hip_measure.so: the poison hook is not in
the shipped one).  Before each of 1000 runs over the same input the hand-off buffers are filled with garbage (totals 3.4e38, flags
a stamp that no run carries).  A hand-off that ever read a total without the current run's stamp would put that garbage into a
carry-in; the output of every run must be bit-identical to the first one's.  The runs restart from the same state (a fresh
block each 250 runs also covers the first-run / re-allocation paths).  GPU box only; run by tests/test_gpu_notch.py."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leansdr_amd.capi as capi


def main():
    nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    assert hasattr(capi.lib, "lsdr_auto_notch_debug_poison"), "not the measure build: " + capi.LIB_PATH
    ctx = capi.Ctx(0)
    rng = np.random.default_rng(21)
    n = 4096 * 64
    t = np.arange(n)
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 12 + 70 * np.exp(2j * np.pi * 0.0713 * t)
         + 40 * np.exp(-2j * np.pi * 0.27 * t) + 25 * np.exp(2j * np.pi * 0.4 * t)).astype(np.complex64)
    d_in = ctx.upload(x)
    d_out = ctx.alloc(n * 8)
    first = None
    runs = 0
    for rep in range(4):
        a = capi.AutoNotch(ctx, nslots, 0.0, 4096 * 16, mode=capi.NOTCH_SCAN)
        sums = []
        for k in range(250):       # the k-th run of a block continues the carried state of run k−1: runs of the same k are compared across blocks
            capi.check(capi.lib.lsdr_auto_notch_debug_poison(a.h))
            a.run_dev(d_in.ptr, n, d_out.ptr, n)
            y = ctx.download(d_out, np.complex64, n)
            if rep == 0 and k in (0, 249):
                assert np.isfinite(y.view(np.float32)).all() and np.abs(y).max() < 1e4
            sums.append(hashlib.blake2b(y.tobytes(), digest_size=16).digest())
            runs += 1
        if first is None:
            first = sums
        else:
            bad = [k for k in range(250) if sums[k] != first[k]]
            assert not bad, (rep, bad[:10])
        aborted = C.c_uint(123)
        capi.check(capi.lib.lsdr_auto_notch_check(a.h, C.byref(aborted)))
        assert aborted.value == 0          # no look-back ever gave up
        a.close()
    d_in.free(); d_out.free(); ctx.close()
    print(f"stress OK: {runs} runs, {nslots} slot(s)")


if __name__ == "__main__":
    main()
