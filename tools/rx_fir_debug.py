import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import leansdr_amd.capi as capi
from leansdr_amd import synth
import pyoracle as po
O = po.Oracle(); ctx = capi.Ctx(0)
g = np.load(os.path.join(ROOT, "tests", "golden", "cstln_receiver.npz"))
rrc = g["rrc_rx"]
x, _ = synth.qpsk_baseband(4 * 100000, 4, seed=7, rms=50.0, snr_db=20.0)
p = po.rx_params(sampler=2, coeffs=rrc, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096)
acq = 128 * 640
a = O.rx(p, x[:acq + len(rrc) - 1])
ref = O.rx(p, x[acq:], state_in=a["state"])
print("acq state agc", a["state"].agc_gain, "est_insp", a["state"].est_insp, "ref end agc", ref["state"].agc_gain)
for geo in ((256, 1024), (1024, 2048)):
    r = capi.CstlnReceiver(ctx, sampler=2, coeffs=rrc, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096, mode=capi.RX_TILED, tile_len=geo[0], tile_warmup=geo[1])
    st = capi.RxState()
    for k, _ in st._fields_:
        setattr(st, k, getattr(a["state"], k))
    r.set_state(st)
    out = r.run(x[acq:])
    print(geo, "n", len(out["sym"]), len(ref["sym"]), r.tiled_stats(), "end agc", out["state"].agc_gain)
    r.close()
    n = min(len(out["sym"]), len(ref["sym"]))
    oc, rc = out["sym"]["cost"][:n].astype(int), ref["sym"]["cost"][:n].astype(int)
    print("  mean cost out %.0f ref %.0f; eq %.4f" % (oc.mean(), rc.mean(), (out["sym"]["symbol"][:n] == ref["sym"]["symbol"][:n]).mean()))
    d = np.abs(oc - rc)
    for s0 in (0, 200, 400, 1000, 5000, 20000, 60000):
        print("   dcost[%d:%d] %.0f   out %.0f ref %.0f" % (s0, s0 + 200, d[s0:s0 + 200].mean(), oc[s0:s0+200].mean(), rc[s0:s0+200].mean()))
# serial GPU vs oracle with the same state hand-over (sanity)
r = capi.CstlnReceiver(ctx, sampler=2, coeffs=rrc, subsampling=16, cstln=1, omega=4.0, meas_decimation=4096)
r.set_state(st)
out = r.run(x[acq:])
r.close()
print("serial gpu from state: equal cost", (out["sym"]["cost"] == ref["sym"]["cost"][:len(out["sym"])]).mean())
