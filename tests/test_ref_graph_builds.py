"""CPU-side half of the drop-in proof: the reference's own app sources compile and link, unchanged, against this repo's
host headers (`make -C leansdr_amd/host ref_graph`), and the one of them that has no GPU block (leantsgen: scheduler, pipebuf,
file_writer only) already runs here and writes the reference binary's bytes.  Skipped where /root/reference is absent."""
import os
import subprocess
import pytest
from conftest import ROOT

REF = "/root/reference/src/apps/leandvb.cc"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="/root/reference not present")


def test_reference_apps_compile_unchanged_against_the_gpu_headers():
    host = os.path.join(ROOT, "leansdr_amd", "host")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "leansdr_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", host, "ref_graph"])
    for app in ("leandvb", "leandvbtx", "leanchansim", "leantsgen"):
        assert os.path.exists(os.path.join(host, "ref_graph", app)), app
    # syntax-only compile straight from the reference tree: no local copy of the source is involved
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-DVERSION=\"t\"", "-I", host, "-I", os.path.join(ROOT, "include"), REF])


def test_leantsgen_on_this_framework_writes_the_reference_bytes():
    mine = subprocess.run([os.path.join(ROOT, "leansdr_amd", "host", "ref_graph", "leantsgen"), "-c", "500"], stdout=subprocess.PIPE, check=True).stdout
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "leantsgen")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref not built")
    theirs = subprocess.run([ref_bin, "-c", "500"], stdout=subprocess.PIPE, check=True).stdout
    assert mine == theirs and len(mine) == 500 * 188
