"""bench_more.py — the secondary configurations reported next to bench.py's headline (its `more` object).  Every entry is
measured after the headline's timed region, on the same GPU, each with its own clock and (where a kernel dominates) its
own roofline.  N=1 only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def single_stream(capi, synth, device, args):
    """north_star's N=1 case: ONE capture on the GPU (fir_filter(k+1) ‖ cstln_receiver(k) on two HIP streams)."""
    import bench
    pipe = bench.C2Pipeline(capi, synth, device, 1, args.batch_msamples, args.period_msamples, (args.tile_len, args.tile_warmup), seed0=77,
                            rx_cus=args.rx_cus, cu_pattern=args.cu_pattern)
    bps = args.batches_per_step
    pipe.run(bps, False)
    pipe.sync()
    t0 = time.perf_counter()
    consumed = pipe.run(max(1, args.steps // 2) * bps, True, snapshot_last=not args.no_verify)
    pipe.sync()
    dt = time.perf_counter() - t0
    out = dict(value=round(consumed / dt / 1e6, 3), unit="MS/s", seconds=round(dt, 3), captures_per_gpu=1,
               roofline=pipe.roofline())
    if not args.no_verify:
        out["verified"] = pipe.verify_last_batch()
    pipe.close()
    return out


def run_all(capi, synth, device, args):
    more = {}
    for name, fn in (("single_stream", single_stream),):
        try:
            more[name] = fn(capi, synth, device, args)
        except Exception as e:      # a failing extra must not take the headline line down; it is reported as failed
            more[name] = {"error": f"{type(e).__name__}: {e}"}
    return more
