"""Channel simulator on the GPU (leansdr_amd/csrc/chan.hip) through the C ABI: == the real `leanchansim` output and the
reference's wgn_c stream (tests/golden/chan.npz), == the pinned oracle for other seeds, call patterns and drift settings."""
import hashlib
import numpy as np
import pytest
import pyoracle as po
from conftest import gold, bits_equal

pytestmark = pytest.mark.gpu


def test_wgn_golden(capi, ctx):
    g = gold("chan.npz")
    w = capi.Wgn(ctx)
    y = w.run(200000, 0.7)
    assert bits_equal(y[:256], g["wgn_head"]) and hashlib.sha256(y.tobytes()).digest() == bytes(g["wgn_sha"])
    w.close()
    w = capi.Wgn(ctx, seed=77)
    y = w.run(5000, 2.0)
    assert bits_equal(y[:256], g["wgn77_head"]) and hashlib.sha256(y.tobytes()).digest() == bytes(g["wgn77_sha"])
    w.close()


@pytest.mark.parametrize("seed", [None, 1, 123456789])
def test_wgn_any_call_pattern(capi, ctx, oracle, seed):
    """The stream and the carried drand48 state do not depend on how it is cut into calls (1 … 3 M samples per call)."""
    sizes = [1, 7, 4096, 100000, 3000001, 63, 5000]
    ref, st = oracle.wgn(sum(sizes), 1.1, seed=seed)
    w = capi.Wgn(ctx, seed=seed)
    pos = 0
    for n in sizes:
        y = w.run(n, 1.1)
        assert bits_equal(y, ref[pos:pos + n]), (n, pos)
        pos += n
    assert w.state == st
    w.close()


def test_wgn_fused_add_and_adder(capi, ctx, oracle):
    x = po.chan_test_input(30000)
    n, _ = oracle.wgn(len(x), 0.5)
    w = capi.Wgn(ctx)
    assert bits_equal(w.run(len(x), 0.5, add=x), oracle.adder(x, n))
    w.close()
    assert bits_equal(capi.adder(ctx, x, n), oracle.adder(x, n))


def test_cconverter_f32_u8(capi, ctx, oracle):
    with np.errstate(all="ignore"):
        x = po.chan_test_input(30000) * 3
    assert np.array_equal(capi.cconv_f32_u8(ctx, x), oracle.cconv_f32_u8(x))


DRIFTS = [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),                   # pass-through (phase 0)
          ((float("nan"), 0.0, 0.0), (0.0, 0.0, 0.0)),          # leanchansim without -f: amp = 0/0
          ((0.015, 2.5e-4, 0.0), (5e-5, 3.5e-5, 0.0)),          # closed-form phase advance
          ((0.015, 2.5e-4, 0.01), (-5e-5, 3.5e-5, 0.3)),        # negative rate → sequential pre-pass
          ((1e6, 0.0, 0.0), (1e-3, 0.0, 0.0))]                  # float → int overflow inside run()


@pytest.mark.parametrize("amp,freq", DRIFTS)
@pytest.mark.parametrize("chunk", [4096, 1000, 0])
def test_drifter(capi, ctx, oracle, amp, freq, chunk):
    x = po.chan_test_input(40000)[8:]
    d = capi.Drifter(ctx, amp, freq)
    a = (0, 0, 0)
    for lo, hi in ((0, 12288), (12288, 39992)):   # carried component phases across calls
        y = d.run(x[lo:hi], chunk)
        ref, a = oracle.drifter(x[lo:hi], amp, freq, a, chunk)
        assert bits_equal(y, ref), (amp, freq, chunk, lo)
        assert d.phases == tuple(a)
    d.close()


@pytest.mark.parametrize("name,args,kw", po.CHAN_CASES)
def test_chansim_is_leanchansim(capi, ctx, oracle, name, args, kw):
    """scaler → + wgn_c → drifter → [cconverter] through the C ABI == the bytes of the real leanchansim."""
    g = gold("chan.npz")
    x = po.chan_test_input()
    kw = dict(kw)
    ou8 = kw.pop("ou8", False)
    scale = kw.pop("scale", 1.0)
    awgn = kw.pop("awgn_db", None)
    oracle.lib.lo_db_to_amp.restype = po.c_f
    oracle.lib.lo_db_to_amp.argtypes = [po.C.c_double]
    stddev = oracle.lib.lo_db_to_amp(awgn) if awgn is not None else 0.0   # option parsing (libm), not the data path
    amp, freq = po.chansim_drifts(**kw)
    w, d = capi.Wgn(ctx), capi.Drifter(ctx, amp, freq)
    y = d.run(w.run(len(x), stddev, add=ctx.scaler(scale, x)), 4096)
    if ou8:
        y = capi.cconv_f32_u8(ctx, y)
    y = y.reshape(-1)
    w.close(); d.close()
    assert len(y) == int(g[name + "_n"])
    assert bits_equal(y[:512], g[name + "_head"]) and bits_equal(y[-512:], g[name + "_tail"])
    assert hashlib.sha256(y.tobytes()).digest() == bytes(g[name + "_sha"])


import os
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APPS = os.path.join(ROOT, "leansdr_amd", "host", "apps")
RG = os.path.join(ROOT, "leansdr_amd", "host", "ref_graph")


@pytest.mark.parametrize("name,args,kw", po.CHAN_CASES)
@pytest.mark.parametrize("buf", ["4096", "1048576"])
def test_leanchansim_amd_is_leanchansim(name, args, kw, buf):
    """The drop-in app reproduces the bytes of the reference `leanchansim` binary, whatever the size of its device pipes."""
    g = gold("chan.npz")
    p = subprocess.run([os.path.join(APPS, "leanchansim_amd"), "--buf", buf] + args.split(), input=po.chan_test_input().tobytes(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    y = np.frombuffer(p.stdout, np.uint8 if "--ou8" in args else np.complex64)
    assert len(y) == int(g[name + "_n"])
    assert bits_equal(y[:512], g[name + "_head"]) and bits_equal(y[-512:], g[name + "_tail"])
    assert hashlib.sha256(y.tobytes()).digest() == bytes(g[name + "_sha"])


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "leansdr_amd", "host", "ref_graph", "leandvb")), reason="ref_graph/leandvb not built")
def test_generator_pipeline_on_gpu():
    """test/leandvb_bench.sh:52-56 with every stage on the GPU: TS → leandvbtx_amd → leanchansim_amd (noise + LO drift) →
    leandvb (the reference's source on the GPU headers, throughput receiver) returns the transmitted packets."""
    ts = gold("tx.npz")["ts"]
    ts = np.tile(ts, (60, 1))                      # 2400 packets
    ts[:, 3] = (np.arange(len(ts)) & 15) | 0x10    # distinct continuity counters
    ts[:, 4:8] = np.arange(len(ts), dtype=">u4").view(np.uint8).reshape(-1, 4)
    cmd = (f"{APPS}/leandvbtx_amd -f 6/5 --power 37.5 --agc | "
           f"{APPS}/leanchansim_amd --awgn 24 --deterministic -f 2.4e6 --lo 10e9 --ppm 0.0005 --drift-period 0.5 | "
           f"LSDR_TILED=1 {RG}/leandvb --f32 --float-scale 1 -f 2.4e6 --sr 2e6 --anf 0 --buf-factor 4096")
    p = subprocess.run(cmd, shell=True, input=ts.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = np.frombuffer(p.stdout, np.uint8).reshape(-1, 188)
    assert len(out) > 2000, len(out)
    sent = {bytes(r) for r in ts}
    bad = [i for i, r in enumerate(out) if bytes(r) not in sent]
    assert all(i < 8 for i in bad), bad[:20]       # only the packets before the derandomizer's first group start (the reference too)


@pytest.mark.parametrize("freq,a0", [(1e-7, 2**40 + 12345), (1e-7, 2**50 + 7), (3.3e-5, 2**51 - 10**7), (1.0 - 2.0**-24, 2**49),
                                     (3.3e-5, -5000000), (1e-7, 2**62), (-1e-7, 0), (-3.3e-5, -2**45 - 3), (-1e-7, 77)])
def test_drifter_large_phase_words(capi, ctx, oracle, freq, a0):
    """Component phase words far from zero: the closed-form advance (exact double sums, or rounding that cannot reach the
    next integer), and the sequential pre-pass where it cannot be proved (negative, ≥ 2^52, fraction next to 1)."""
    x = po.chan_test_input(30000)[8:]
    amp, fr = (0.01, 0.0, 0.0), (float(np.float32(freq)), 0.0, 0.0)
    d = capi.Drifter(ctx, amp, fr)
    d.phases = (a0, 0, 0)
    y = d.run(x, 4096)
    ref, a = oracle.drifter(x, amp, fr, (a0, 0, 0), 4096)
    assert bits_equal(y, ref) and d.phases == tuple(a)
    d.close()
