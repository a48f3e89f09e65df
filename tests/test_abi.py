"""The C-ABI library loads and exports every symbol include/lsdr_hip.h declares
(no compute: runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lsdr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef LSDR_MEASURE.*?#endif", "", src, flags=re.S)      # the measure build's extras are not part of the shipped ABI
    return sorted(set(re.findall(r"\b(lsdr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "lsdr_fir_filter_run" in syms and "lsdr_rx_run" in syms and len(syms) >= 30


def test_library_exports_every_declared_symbol(capi):
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"liblsdr_hip.so lacks: {missing}"


def test_abi_version(capi):
    assert capi.lib.lsdr_abi_version() == 2


def test_no_cpu_fallback(capi):
    """Without a GPU, context creation must fail loudly (never fall back)."""
    if capi.lib.lsdr_device_count() > 0:
        return
    import pytest
    with pytest.raises(capi.LsdrError):
        capi.Ctx(0)


def test_product_does_not_touch_oracle():
    """leansdr_amd/ must not import, link or reference anything under oracle/."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "leansdr_amd")):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".cc", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"pyoracle|lsdr_oracle|oracle/|liblsdr_oracle|_ref/", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_header_is_plain_c(tmp_path):
    """include/lsdr_hip.h is the C-ABI boundary: it must compile as C (gcc, no C++/HIP types) on its own."""
    import subprocess
    src = tmp_path / "h.c"
    src.write_text('#include "lsdr_hip.h"\nint main(void) { return LSDR_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


MEASURE_HOOKS = ("LSDR_FIR_SKIP", "LSDR_RX_SKIP", "LSDR_RX_DBG", "lsdr_auto_notch_debug_poison")


def test_no_work_skipping_hook_in_the_shipped_library(capi):
    """Hooks that skip or corrupt work for measurements (no filter launch, no receiver kernels, timing-only tiles, poisoned hand-off
    buffers) are compiled only into the measure build (-DLSDR_MEASURE): no environment variable can turn liblsdr_hip.so into a
    no-op that still reports consumed/produced."""
    blob = open(capi.LIB_PATH, "rb").read()
    found = [h for h in MEASURE_HOOKS if h.encode() in blob]
    assert not found, found
    lib = ctypes.CDLL(capi.LIB_PATH)
    assert not hasattr(lib, "lsdr_auto_notch_debug_poison")


def test_measure_build_has_them():
    """... and the measure build (tools/variants/liblsdr_hip_measure.so, built by __graft_entry__.build()) does."""
    path = os.path.join(ROOT, "tools", "variants", "liblsdr_hip_measure.so")
    assert os.path.exists(path), "run: make -C leansdr_amd/csrc measure"
    blob = open(path, "rb").read()
    missing = [h for h in MEASURE_HOOKS if h.encode() not in blob]
    assert not missing, missing
