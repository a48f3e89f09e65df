"""Debug aid for the capture-batch front end (tests/test_gpu_capture_batch.py::test_front_end_against_the_oracle_chain): where do the
packed decisions leave the oracle's?  python tools/cb_debug.py [tile] [cw_amp] [moving 0/1] [n_log2]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import leansdr_amd.capi as capi
import pyoracle as po
import bench_c1

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cw = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
moving = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 20)
anf = int(sys.argv[5]) if len(sys.argv) > 5 else 1
O = po.Oracle()
ctx = capi.Ctx(0)
dec_period = 64 * 4096
gen = bench_c1.Generator(capi, ctx, n, 1)
d, _ = gen.capture(0, 900)
iq = ctx.download(d, np.uint8, 2 * n); d.free(); gen.close()
if cw:
    t = np.arange(n)
    f = np.where(t < 2 * dec_period + 4096 * 5, 0.1234, -0.31) if moving else np.full(n, 0.1234)
    ph = 2 * np.pi * np.cumsum(f)
    x = iq.reshape(-1, 2).astype(np.float64) - 128 + cw * np.stack([np.cos(ph), np.sin(ph)], axis=1)
    iq = np.clip(np.rint(x + 128), 0, 255).astype(np.uint8).reshape(-1)
buf = ctx.upload(iq)
cb = capi.CaptureBatch(ctx, 1, n, bench_c1.OMEGA, anf=anf, tile_len=tile, tile_warmup=512, notch_decimation=dec_period)
cb.run_async([buf.ptr], n); res = cb.wait()
xf = O.cconverter_u8(iq)
if anf:
    notched, _ = O.auto_notch(xf, 1, dec_period)
    print("bins", cb.bins(0))
else:
    notched = xf
o = O.rx(po.rx_params(sampler=1, cstln=1, omega=bench_c1.OMEGA, meas_decimation=1 << 20), notched)
got = cb.words(0, res[0]["symbols"])
ref = o["sym"]["symbol"] & 3
print("result", res[0]); print("symbols got", len(got), "ref", len(ref))
m = min(len(got), len(ref))
neq = got[:m] != ref[:m]
print("mismatches (no re-alignment):", int(neq.sum()), "of", m)
# walk: find slips
W = 64
off = 0; i = 0; events = []
while i + W < m and i + off + W < len(got) and len(events) < 40:
    a = got[i + off:i + off + W]; b = ref[i:i + W]
    if (a != b).mean() > 0.3:
        best = None
        for dd in (-2, -1, 1, 2):
            if i + off + dd >= 0 and i + off + dd + W <= len(got):
                r = (got[i + off + dd:i + off + dd + W] != b).mean()
                if best is None or r < best[0]:
                    best = (r, dd)
        events.append((i, best))
        if best and best[0] < 0.3:
            off += best[1]
        i += W
    else:
        i += 16
sps = bench_c1.OMEGA
for (i, best) in events:
    samp = i * sps
    print(f"  symbol {i} (sample ~{samp:.0f}, tile ~{(samp - 512) / tile + 1:.2f}, block {samp / 4096:.2f}): best shift {best}")
print("final offset", off)
