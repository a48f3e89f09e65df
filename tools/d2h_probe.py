#!/usr/bin/env python3
"""tools/d2h_probe.py — device-to-host rate of hipMemcpyAsync into pinned memory: 16 copies of 12.9 MB (what lsdr_capture_batch sends back per group) against one of 206 MB,
and both again while a second process holds a context on the GPU.  GPU box."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi

def run(tag):
    ctx = capi.Ctx(0)
    n1 = 68489 * 188
    dev = ctx.alloc(16 * n1 + 4096)
    host = C.c_void_p()
    capi.check(capi.lib.lsdr_malloc_host(16 * n1 + 4096, C.byref(host)))
    for label, chunks in (("16 x 12.9 MB", [(i * n1, n1) for i in range(16)]), ("1 x 206 MB", [(0, 16 * n1)])):
        best = 0.0
        for rep in range(6):
            ctx.sync(); t0 = time.perf_counter()
            for off, nb in chunks:
                capi.check(capi.lib.lsdr_memcpy_d2h(ctx.h, C.c_void_p(host.value + off), dev.at(off), nb))
            ctx.sync(); dt = time.perf_counter() - t0
            best = max(best, 16 * n1 / dt / 1e9)
        print(f"{tag}: {label}: {best:.1f} GB/s", flush=True)
    capi.lib.lsdr_free_host(host); dev.free(); ctx.close()

if len(sys.argv) > 1 and sys.argv[1] == "child":
    run("child of a process that holds a context")
else:
    run("alone")
    ctx_keep = capi.Ctx(0)          # this process keeps a context while the child measures
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"])
