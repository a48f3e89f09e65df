#!/usr/bin/env python3
"""tools/more_one.py NAME [NAME...] — run single secondary configurations of bench_more (anf1, c3, c5_rescoped, ...) with
bench.py's default geometry and print each result as JSON."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
from leansdr_amd import synth
import bench_more
a = argparse.Namespace(batch_msamples=int(os.environ.get('LSDR_MORE_BATCH', 256)), more_batch_msamples=64, period_msamples=4, tile_len=256, tile_warmup=256, batches_per_step=24, steps=20, no_verify=not os.environ.get("LSDR_MORE_VERIFY"),
                       rx_cus=0, cu_pattern="xcd_major", captures=int(os.environ.get('LSDR_MORE_CAPTURES', 1)), fir_arith=os.environ.get("LSDR_MORE_ARITH", "blk"))
for name in sys.argv[1:]:
    r = getattr(bench_more, name)(capi, synth, 0, a)
    r.pop("trace", None)
    print(name, json.dumps(r), flush=True)
