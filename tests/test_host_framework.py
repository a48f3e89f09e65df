"""Host data-flow runtime (leansdr_amd/host/leansdr/framework.h): pipe and scheduler semantics on host pipes, no GPU.
Compiles tests/host/framework_test.cc against the framework and runs it."""
import os
import subprocess
from conftest import ROOT


def test_framework_semantics(tmp_path):
    exe = tmp_path / "framework_test"
    host = os.path.join(ROOT, "leansdr_amd", "host")
    lib = os.path.join(ROOT, "leansdr_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", host, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "framework_test.cc"), "-o", str(exe),
                           "-L", lib, "-llsdr_hip", f"-Wl,-rpath,{lib}"])
    p = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 0 and p.stdout.decode().strip() == "ok", p.stdout.decode() + p.stderr.decode()


def test_same_program_against_the_reference_framework(tmp_path):
    """The same test program compiled against the reference's own framework.h (where /root/reference exists) passes too:
    the checks above state the reference's semantics, not just ours."""
    import pytest
    ref = "/root/reference/src"
    if not os.path.exists(os.path.join(ref, "leansdr", "framework.h")):
        pytest.skip("no reference sources on this machine")
    exe = tmp_path / "framework_test_ref"
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-w", "-I", ref, os.path.join(ROOT, "tests", "host", "framework_test.cc"), "-o", str(exe)])
    p = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 0 and p.stdout.decode().strip() == "ok", p.stdout.decode() + p.stderr.decode()


def _build_and_run(tmp_path, name, src, include, link_lsdr):
    exe = tmp_path / name
    cmd = ["g++", "-O1", "-std=c++14", "-w", "-I", include]
    if link_lsdr:
        lib = os.path.join(ROOT, "leansdr_amd")
        cmd += ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", src), "-o", str(exe), "-L", lib, "-llsdr_hip", f"-Wl,-rpath,{lib}"]
    else:
        cmd += [os.path.join(ROOT, "tests", "host", src), "-o", str(exe)]
    subprocess.check_call(cmd)
    p = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


def test_generic_reports(tmp_path):
    """file_printer (decimation, scale), rate_estimator, file_carrayprinter (fixed batches), file_vectorprinter: the text this
    build prints; where the reference sources exist, the same program compiled against them prints the same bytes."""
    mine = _build_and_run(tmp_path, "generic_mine", "generic_test.cc", os.path.join(ROOT, "leansdr_amd", "host"), True)
    lines = mine.splitlines()
    assert sum(l.startswith("F ") for l in lines) == 10 and "F -4.50" in lines          # items 3, 7, … of 40, scaled by 2
    assert sum(l.startswith("RATE ") for l in lines) >= 5
    assert sum(l.startswith("SYMBOLS 6 ") for l in lines) == 3                            # 23 items → three batches of 6
    assert "VEC [0.000,0.125,0.250,0.375]" in lines and "VEC [1.000,1.125,1.250,1.375]" in lines
    ref = "/root/reference/src"
    if os.path.exists(os.path.join(ref, "leansdr", "generic.h")):
        theirs = _build_and_run(tmp_path, "generic_ref", "generic_test.cc", ref, False)
        assert mine == theirs
