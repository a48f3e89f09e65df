"""The product's host-side table/coefficient design (leansdr_amd/csrc/host_tables.cpp,
through the C ABI) against the oracle and the goldens.  CPU only."""
import numpy as np
import pytest
from conftest import gold, bits_equal


def test_trig_and_constellations(capi, oracle):
    assert bits_equal(capi.trig16(), oracle.trig16())
    for pre, fec in [(0, 0), (1, 0), (2, 1), (3, 1), (3, 3), (3, 8), (4, 3), (4, 8), (5, 0), (6, 0), (7, 0), (8, 0)]:
        a, b = capi.cstln_lut(pre, fec), oracle.cstln_lut(pre, fec)
        assert a["nsymbols"] == b["nsymbols"] and a["nrotations"] == b["nrotations"]
        for k in ("symbols", "cost", "symbol", "phase_error"):
            assert bits_equal(a[k], b[k]), (pre, fec, k)


def test_unsupported_code_rate_is_an_error(capi):
    with pytest.raises(capi.LsdrError):
        capi.cstln_lut(capi.APSK16, capi.FEC12)   # "Code rate not supported with APSK16", dvb.h:60


def test_viterbi_quad_kernel_trellis_check(capi):
    """(CPU) lsdr_viterbi_q4_supported builds the trellis like trellis::init_convolutional (viterbi.h:59-92) and checks the
    structure k_viterbi_q4 relies on (predecessor registers, input symbols, coded symbols linear in the state bits, branch order):
    it must hold for the two codes the kernel is instantiated for, fail for every other rate, and report unsupported pairs."""
    sup = capi.lib.lsdr_viterbi_q4_supported
    assert sup(capi.QPSK, capi.FEC12) == 1 and sup(capi.PSK8, capi.FEC23) == 1
    for cstln, rate in [(capi.QPSK, capi.FEC34), (capi.QPSK, capi.FEC56), (capi.QPSK, capi.FEC78), (capi.QPSK, capi.FEC46),
                        (capi.BPSK, capi.FEC12), (capi.APSK16, capi.FEC34), (capi.QAM64, capi.FEC56)]:
        assert sup(cstln, rate) == 0, (cstln, rate)
    assert sup(capi.QPSK, capi.FEC23) == -1      # 3 coded bits do not fill QPSK symbols (dvb.h:1243)
    assert sup(capi.QPSK, capi.FEC910) == -1     # not a DVB-S code rate


def test_filtergen(capi, oracle):
    g = gold("tables.npz")
    assert bits_equal(capi.lowpass(312, float(g["lowpass_c2_fcut"])), g["lowpass_c2"])
    assert bits_equal(capi.root_raised_cosine(int(10 * 8e6 * 16 / (22 * (2e6 / 2) * 0.35)),
                                              np.float32(2e6 / (8e6 * 16)), np.float32(0.35)), g["rrc_rx"])
    for order, fc in [(14, 0.4895), (40, 0.1), (1, 0.3), (99, 0.02)]:
        assert bits_equal(capi.lowpass(order, np.float32(fc)), oracle.lowpass(order, np.float32(fc)))
        assert bits_equal(capi.lowpass(order, np.float32(fc), False), oracle.lowpass(order, np.float32(fc), False))
    for order, fs, ro in [(41, 0.25, 0.35), (64, 0.25, 0.25), (100, 0.5, 0.2), (33, 1 / 3.0, 0.35)]:
        assert bits_equal(capi.root_raised_cosine(order, np.float32(fs), np.float32(ro)),
                          oracle.rrc(order, np.float32(fs), np.float32(ro)))


def test_bench_more_lists_every_secondary_configuration():
    """(CPU) bench.py's `more` object is built from bench_more.run_all's list: every entry must be a function of the module
    (a typo would only show up on the GPU box, after the headline has been measured)."""
    import ast, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "bench_more.py")).read())
    funcs = {n.name for n in tree.body if isinstance(n, ast.FunctionDef)}
    run_all = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "run_all")
    names = [(e.elts[0].value, e.elts[1].id) for n in ast.walk(run_all) if isinstance(n, ast.Tuple) for e in [n]
             if len(n.elts) == 2 and isinstance(n.elts[0], ast.Constant) and isinstance(n.elts[1], ast.Name)]
    assert len(names) >= 9 and all(f in funcs for _, f in names)
    assert {"four_captures", "anf1", "c2_offset", "c3", "c5_rescoped", "c1", "c1_hs", "exact_batch", "end_to_end"} <= {k for k, _ in names}
