#!/usr/bin/env python3
"""tools/c3_profile.py — bench_more.c3 alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
from leansdr_amd import synth
import bench_more
a = argparse.Namespace(batch_msamples=64, period_msamples=4, tile_len=256, tile_warmup=256, batches_per_step=24, steps=4, no_verify=True, rx_cus=0, cu_pattern="xcd_major", captures=4)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
r = getattr(bench_more, which)(capi, synth, 0, a)
print(json.dumps({k: v for k, v in r.items() if k != "trace"})[:1500])
