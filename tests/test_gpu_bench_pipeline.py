"""bench.py's pipeline as a test: the endless-stream construction (every batch consumes exactly B samples and continues where
the previous one stopped), one fir_filter launch over all captures, queued tiled receivers — and the two checks bench.py makes
of its own timed output."""
import os
import sys
import numpy as np
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["exact", "blk"])
def pipe(capi, request):
    """Both filter arithmetics: `exact` (the reference's) and `blk` — what bench.py's headline times (--fir-arith blk)."""
    sys.path.insert(0, ROOT)
    import bench
    from leansdr_amd import synth
    arith = {"exact": capi.FIR_EXACT, "blk": capi.FIR_MFMA_BLK}[request.param]
    p = bench.C2Pipeline(capi, synth, 0, 3, 8, 4, (256, 256), seed0=5, fir_arith=arith)     # 3 captures, batches of 2 x 4 Mi samples
    yield p
    p.close()


def want_filter_output(pipe, oracle, x_full):
    """What the decimated stream must equal BIT FOR BIT: the reference's arithmetic, or (blk) the oracle's stated blocked sums."""
    g = pipe.geo
    if pipe.fir_arith == pipe.capi.FIR_MFMA_BLK:
        return oracle.fir_filter(pipe.coeffs, g["decim"], x_full, fma="blk", scale=75.0)[0]
    return oracle.fir_filter(pipe.coeffs, g["decim"], oracle.scaler(75.0, x_full))[0]


def test_every_batch_of_the_endless_stream_is_the_oracles_filter_output(pipe, oracle):
    """The stream is B-periodic, so EVERY batch of a capture must leave the same decimated stream in its buffer, bit for bit
    the oracle's fir_filter output — whichever of the three buffers it landed in, after any number of queued batches."""
    import bench
    g = pipe.geo
    refs = []
    for cp in pipe.caps:
        x_full = np.concatenate([np.tile(cp.x, g["reps"]), cp.x[:bench.EXTRA * g["decim"] + g["N"]]])
        refs.append(want_filter_output(pipe, oracle, x_full).view(np.uint64))
    for burst in (7, 12):
        pipe.run(burst, True)
        pipe.sync()
        for c, cp in enumerate(pipe.caps):
            for i, d in enumerate(cp.dec):
                y = pipe.ctx.download(d, np.complex64, g["n_out"] + bench.EXTRA).view(np.uint64)
                assert np.array_equal(y, refs[c]), (burst, c, i, int(np.count_nonzero(y != refs[c])))


def test_bench_self_verification_passes(pipe):
    """What bench.py reports as `verified`: the last queued batch against the oracle — fir_filter bit for bit, soft symbols
    vs the serial receiver from the device's own loop state under bench.TOL."""
    import bench
    consumed = pipe.run(9, True, snapshot_last=True)
    pipe.sync()
    assert consumed == 9 * pipe.geo["B"] * len(pipe.caps)
    v = pipe.verify_last_batch()
    assert v["pass"], v
    assert [b["batch_index"] for b in v["batches"]] == [3, 8]          # a MIDDLE batch and the last one, each from its own snapshot
    assert all(b["pass"] for b in v["batches"])
    if pipe.fir_arith != pipe.capi.FIR_EXACT:
        assert v["fir_max_rel_err_vs_exact"] <= 1e-5
    assert v["fir_bit_exact"] and v["count_equal"] and v["first_tile_bit_exact"]
    assert v["equal_decisions"] >= bench.TOL["min_equal_decisions"] and v["mean_abs_dcost"] <= bench.TOL["max_mean_abs_dcost"]
    r = pipe.roofline()
    assert r["launches_timed"] >= 9 and 0 < r["frac"] < 1
