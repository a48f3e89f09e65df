#!/usr/bin/env python3
"""tools/vmm_probe.py — are the fast / slow kinds of resident buffers a property of PHYSICAL chunks?  hipMemCreate makes NCH physical chunks of 512 MiB,
each is mapped alone and the headline's filter launch is timed over it (64 Mi samples); then the fastest four and the slowest four are mapped
back to back into one 2 GiB range and the 256 Mi-sample launch is timed over both — next to a hipMalloc'd buffer.  TB/s of algorithmic bytes."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leansdr_amd.capi as capi
import bench

hip = C.CDLL("libamdhip64.so")
class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p), ("allocFlags", Flags)]
class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]
def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")
hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(Prop), C.c_ulonglong]
hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
hip.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Access), C.c_size_t]
hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(C.c_size_t), C.POINTER(Prop), C.c_int]

ctx = capi.Ctx(0)
coeffs, decim = bench.c2_filter(capi)
prop = Prop(); prop.type = 1; prop.requestedHandleType = 0; prop.location = Loc(1, 0)
g = C.c_size_t()
ck(hip.hipMemGetAllocationGranularity(C.byref(g), C.byref(prop), 1), "granularity")
CH = 512 << 20
NCH = int(os.environ.get("NCH", 24))
print(f"recommended granularity {g.value} B; {NCH} chunks of {CH >> 20} MiB", flush=True)
acc = Access(Loc(1, 0), 3)
rng = np.random.default_rng(0)
blk = ((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) * 0.7).astype(np.complex64)
d_blk = ctx.upload(blk)
n1 = CH // 8
out = ctx.alloc((256 << 20) // decim * 8 + 1024)
f = capi.FirFilter(ctx, coeffs, decim, in_scale=75.0, arith=capi.FIR_MFMA_BLK)
e0, e1 = ctx.event(), ctx.event()
def t(ptr, n, reps=8):
    for _ in range(2):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.sync(); ctx.event_record(e0)
    for _ in range(reps):
        f.run_dev(ptr, n, out.ptr, n // decim)
    ctx.event_record(e1); ctx.sync()
    return n * 8.0333 / (ctx.event_elapsed_ms(e0, e1) / reps) / 1e9
def fill(ptr, nbytes):
    for r in range(nbytes // blk.nbytes):
        capi.check(capi.lib.lsdr_memcpy_d2d(ctx.h, C.c_void_p(ptr + r * blk.nbytes), d_blk.ptr, blk.nbytes))
    ctx.sync()
handles, va1 = [], C.c_void_p()
ck(hip.hipMemAddressReserve(C.byref(va1), CH, 0, None, 0), "reserve")
speed = []
for k in range(NCH):
    h = C.c_void_p()
    ck(hip.hipMemCreate(C.byref(h), CH, C.byref(prop), 0), "create")
    handles.append(h)
    ck(hip.hipMemMap(va1, CH, 0, h, 0), "map")
    ck(hip.hipMemSetAccess(va1, CH, C.byref(acc), 1), "access")
    fill(va1.value, CH)
    speed.append(t(va1, n1, 16))
    ctx.sync()
    ck(hip.hipMemUnmap(va1, CH), "unmap")
print("per chunk (64 Mi-sample launches):", " ".join(f"{s:.2f}" for s in speed), flush=True)
order = np.argsort(speed)
va4 = C.c_void_p()
ck(hip.hipMemAddressReserve(C.byref(va4), 4 * CH + (2 << 20), 0, None, 0), "reserve4")
def assemble(idx, label):
    for i, k in enumerate(idx):
        ck(hip.hipMemMap(C.c_void_p(va4.value + i * CH), CH, 0, handles[k], 0), "map4")
    ck(hip.hipMemSetAccess(va4, 4 * CH, C.byref(acc), 1), "access4")
    n = (4 * CH) // 8 - 4096
    r = [t(va4, n) for _ in range(3)]
    print(f"{label} chunks {list(map(int, idx))}: 256 Mi-sample launch", " ".join(f"{x:.2f}" for x in r), flush=True)
    ctx.sync()
    ck(hip.hipMemUnmap(va4, 4 * CH), "unmap4")
assemble(order[-4:], "fastest four")
assemble(order[:4], "slowest four")
assemble(order[-4:], "fastest four")
for _ in range(3):
    d = ctx.alloc(4 * CH + 4096); fill(d.ptr, 4 * CH)
    print("hipMalloc 2 GiB:", " ".join(f"{t(C.c_void_p(d.ptr), (4 * CH) // 8 - 4096):.2f}" for _ in range(2)), flush=True)
