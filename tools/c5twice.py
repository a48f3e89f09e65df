import os, sys, json, argparse
sys.path.insert(0, "/root/repo")
import leansdr_amd.capi as capi
from leansdr_amd import synth
import bench_more
a = argparse.Namespace(batch_msamples=64, period_msamples=4, tile_len=256, tile_warmup=256, batches_per_step=24, steps=4, no_verify=True, rx_cus=0, cu_pattern="xcd_major", captures=4)
for i in range(3):
    r = bench_more.c5_rescoped(capi, synth, 0, a)
    print(i, r["value"], r["seconds"], r["host_seconds_per_stage"], flush=True)
r = bench_more.c3(capi, synth, 0, a)
print("c3", r["value"], r["seconds"], r["host_seconds_per_stage"])
