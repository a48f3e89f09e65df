"""Parity of the HIP cstln_receiver (LSDR_RX_SERIAL, through the C ABI) with the
oracle and the reference's golden vectors: soft symbols, measurement outputs and
the full loop state are bit-exact."""
import numpy as np
import pytest
from conftest import gold, bits_equal, iq16_to_cf32
from test_oracle_golden import RX_CASES, rx_input, check_rx_against_golden, state_vec
import pyoracle as po

pytestmark = pytest.mark.gpu


def gpu_kwargs(capi, kw, g):
    kw = dict(kw)
    if kw.get("sampler") == 2:
        kw["coeffs"] = g["rrc_rx"]
    return kw


@pytest.mark.parametrize("tag,kw,src,limit", RX_CASES, ids=[c[0] for c in RX_CASES])
def test_golden(capi, ctx, oracle, tag, kw, src, limit):
    g = gold("cstln_receiver.npz")
    x = rx_input(oracle, g, src, limit)
    r = capi.CstlnReceiver(ctx, **gpu_kwargs(capi, kw, g))
    out = r.run(x)
    check_rx_against_golden(out, g, tag)
    r.close()


def test_streaming_state_carry(capi, ctx, oracle):
    """Feeding the stream in pieces through the C ABI (state lives in the handle) == one shot."""
    g = gold("cstln_receiver.npz")
    x = rx_input(oracle, g, "iq4", None)
    whole = oracle.rx(po.rx_params(sampler=1, cstln=1, omega=4.0, meas_decimation=4096), x)
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0, meas_decimation=4096)
    outs, pos = [], 0
    for piece in (5000, 129, 128, 100, 17000, 1 << 30):
        o = r.run(x[pos:pos + piece], meas=False)
        outs.append(o["sym"])
        pos += o["consumed"]
    got = np.concatenate(outs)
    assert bits_equal(got["cost"], whole["sym"]["cost"]) and bits_equal(got["symbol"], whole["sym"]["symbol"])
    sv, _ = state_vec(r.state())
    wv, _ = state_vec(whole["state"])
    assert bits_equal(sv, wv)
    r.close()


def test_output_capacity_gate(capi, ctx, oracle):
    """run() stops while fewer than 128 output slots remain (sdr.h:784)."""
    g = gold("cstln_receiver.npz")
    x = rx_input(oracle, g, "iq4", 8192)
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0)
    din = ctx.upload(x)
    dout = ctx.alloc(4 * 200)
    o = r.run_dev(din.ptr, len(x), dout.ptr, 200, meas=False)
    # 200 slots: chunks run while cap-produced >= 128 → stops after the chunk that brings produced above 72
    assert o["produced"] <= 200 and 200 - o["produced"] < 128 and o["consumed"] % 128 == 0
    din.free(); dout.free(); r.close()


def test_too_little_input_is_no_progress(capi, ctx):
    r = capi.CstlnReceiver(ctx, sampler=1, cstln=1, omega=4.0)
    o = r.run(np.zeros(128, np.complex64), meas=False)   # needs 128 + readahead(1)
    assert o["consumed"] == 0 and o["produced"] == 0
    r.close()
