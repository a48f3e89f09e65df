"""leansdr_amd — MI355X-native leandvb IQ hot path (HIP/gfx950) behind a C ABI.

The product is `liblsdr_hip.so` (leansdr_amd/csrc, include/lsdr_hip.h) plus the
C++ host framework in leansdr_amd/host (scheduler / pipebuf / block shims with
the reference's class surface).  This Python package is only a thin ctypes
driver used by tests/, bench.py and __graft_entry__.py.  There is no CPU
fallback: importing `leansdr_amd.capi` raises if the HIP library is missing and
creating a context raises if no GPU is present.
"""
__version__ = "0.1.0"
