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
