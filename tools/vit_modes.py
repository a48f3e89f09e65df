"""Debug helper: GPU viterbi_sync vs oracle for a (constellation, rate) mode; prints the first differing byte."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
from leansdr_amd import capi
o = po.Oracle()
ctx = capi.Ctx(0)
NS = {0: 2, 1: 4, 2: 8, 3: 16, 4: 32, 5: 64, 6: 16, 7: 64, 8: 256}
modes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(2, 2)]
for cst, rate in modes:
    rng = np.random.default_rng(4)
    n = 50000
    sym = np.zeros(n, capi.SOFTSYM)
    sym["symbol"] = rng.integers(0, NS[cst], n)
    sym["cost"] = -rng.integers(0, 9000, n)
    v = capi.Viterbi(ctx, cst, rate)
    got, cons = v.run_stream(sym)
    cur = v.current_sync
    v.close()
    want, wcons, wcur = o.viterbi_sync(sym, cst, rate)
    d = np.where(got[:min(len(got), len(want))] != want[:min(len(got), len(want))])[0]
    print(cst, rate, "len", len(got), len(want), "cons", cons, wcons, "cur", cur, wcur, "ndiff", len(d), "first", d[:10])
