"""Where does the 24.7 ms go after viterbi in the c5 chain?  MODE=sleep|dummy|none"""
import os, sys, time, argparse
sys.path.insert(0, "/root/repo")
import leansdr_amd.capi as capi
from leansdr_amd import synth
import bench_more
mode = os.environ.get("MODE", "none")
vorig = capi.Viterbi.run_dev
log = []
def vrun(self, *a):
    t0 = time.perf_counter()
    r = vorig(self, *a)
    t1 = time.perf_counter()
    if r[0] or r[1]:
        if mode == "sleep":
            time.sleep(0.03)
        elif mode == "dummy":
            d = self.ctx.alloc(256); self.ctx.sync()
            t2 = time.perf_counter()
            capi.check(capi.lib.lsdr_memcpy_d2d(self.ctx.h, d.ptr, d.at(128), 64)); self.ctx.sync()
            log.append(("dummy-after-vit ms", round((time.perf_counter() - t2) * 1e3, 3), "vit ms", round((t1 - t0) * 1e3, 3)))
            d.free()
        else:
            log.append(("vit ms", round((t1 - t0) * 1e3, 3)))
    return r
capi.Viterbi.run_dev = vrun
morig = capi.MpegSync.run_dev
def mrun(self, *a):
    t0 = time.perf_counter()
    r = morig(self, *a)
    if r[0] > 1000:
        log.append(("msync ms", round((time.perf_counter() - t0) * 1e3, 3)))
    return r
capi.MpegSync.run_dev = mrun
a = argparse.Namespace(batch_msamples=64, period_msamples=4, tile_len=256, tile_warmup=256, batches_per_step=24, steps=4, no_verify=True, rx_cus=0, cu_pattern="xcd_major", captures=4)
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
r = (bench_more.c5_rescoped if which == "c5" else bench_more.c3)(capi, synth, 0, a)
print(mode, which, r["value"], r["host_seconds_per_stage"])
print(log[-8:])
