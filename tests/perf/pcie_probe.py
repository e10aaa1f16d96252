#!/usr/bin/env python3
"""Host <-> device copy rates on this box: pageable, pinned, registered in place (decides how the host-pointer twins stage)."""
import ctypes
import time

import numpy as np

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
H2D, D2H = 1, 2


def chk(rc, what):
    assert rc == 0, (what, rc)


def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best


for mb in (16, 160):
    nbytes = mb << 20
    dev = ctypes.c_void_p(); chk(hip.hipMalloc(ctypes.byref(dev), nbytes), "malloc")
    dev2 = ctypes.c_void_p(); chk(hip.hipMalloc(ctypes.byref(dev2), nbytes), "malloc")
    page = np.ones(nbytes // 8, dtype=np.uint64)
    pin = ctypes.c_void_p(); chk(hip.hipHostMalloc(ctypes.byref(pin), nbytes, 0), "hostmalloc")
    pin2 = ctypes.c_void_p(); chk(hip.hipHostMalloc(ctypes.byref(pin2), nbytes, 0), "hostmalloc")
    ctypes.memset(pin, 1, nbytes)
    pa = page.ctypes.data
    print(f"--- {mb} MiB")
    for name, f in [
        ("H2D pageable", lambda: chk(hip.hipMemcpy(dev, pa, nbytes, H2D), "c")),
        ("D2H pageable", lambda: chk(hip.hipMemcpy(pa, dev, nbytes, D2H), "c")),
        ("H2D pinned", lambda: chk(hip.hipMemcpy(dev, pin, nbytes, H2D), "c")),
        ("D2H pinned", lambda: chk(hip.hipMemcpy(pin, dev, nbytes, D2H), "c")),
        ("host memcpy page->pinned (1 thread)", lambda: ctypes.memmove(pin, pa, nbytes)),
    ]:
        s = t(f)
        print(f"{name:40s} {s * 1e3:8.2f} ms  {nbytes / s / 1e9:6.1f} GB/s")
    s = t(lambda: (chk(hip.hipHostRegister(pa, nbytes, 0), "reg"), chk(hip.hipHostUnregister(pa), "unreg")), 3)
    print(f"{'register + unregister in place':40s} {s * 1e3:8.2f} ms")
    t0 = time.perf_counter(); chk(hip.hipHostRegister(pa, nbytes, 0), "reg"); t_reg = time.perf_counter() - t0
    s = t(lambda: chk(hip.hipMemcpy(dev, pa, nbytes, H2D), "c"))
    print(f"{'register only':40s} {t_reg * 1e3:8.2f} ms;  H2D registered {s * 1e3:.2f} ms {nbytes / s / 1e9:.1f} GB/s")
    chk(hip.hipHostUnregister(pa), "unreg")
    # both directions at once, pinned, two streams
    s1, s2 = ctypes.c_void_p(), ctypes.c_void_p()
    chk(hip.hipStreamCreate(ctypes.byref(s1)), "s"); chk(hip.hipStreamCreate(ctypes.byref(s2)), "s")

    def duplex():
        chk(hip.hipMemcpyAsync(dev, pin, nbytes, H2D, s1), "a")
        chk(hip.hipMemcpyAsync(pin2, dev2, nbytes, D2H, s2), "a")
        hip.hipStreamSynchronize(s1); hip.hipStreamSynchronize(s2)
    s = t(duplex)
    print(f"{'pinned H2D + D2H concurrently':40s} {s * 1e3:8.2f} ms  {2 * nbytes / s / 1e9:6.1f} GB/s total")
    page2 = np.ones(nbytes // 8, dtype=np.uint64)
    pb = page2.ctypes.data
    times = {}

    def duplex_pageable():
        t0 = time.perf_counter()
        chk(hip.hipMemcpyAsync(dev, pa, nbytes, H2D, s1), "a")
        times["after_h2d_call"] = time.perf_counter() - t0
        chk(hip.hipMemcpyAsync(pb, dev2, nbytes, D2H, s2), "a")
        times["after_d2h_call"] = time.perf_counter() - t0
        hip.hipStreamSynchronize(s1); hip.hipStreamSynchronize(s2)
    s = t(duplex_pageable)
    print(f"{'pageable H2D + D2H, two streams':40s} {s * 1e3:8.2f} ms  {2 * nbytes / s / 1e9:6.1f} GB/s total "
          f"(the H2D call returned after {times['after_h2d_call'] * 1e3:.2f} ms, the D2H call after {times['after_d2h_call'] * 1e3:.2f} ms)")
