"""The C-ABI library loads and exports every symbol include/lsdr_hip.h declares
(no compute: runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lsdr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lsdr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "lsdr_fir_filter_run" in syms and "lsdr_rx_run" in syms and len(syms) >= 30


def test_library_exports_every_declared_symbol(capi):
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"liblsdr_hip.so lacks: {missing}"


def test_abi_version(capi):
    assert capi.lib.lsdr_abi_version() == 1


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
