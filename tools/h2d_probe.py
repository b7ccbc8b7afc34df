#!/usr/bin/env python3
"""Bare host-to-device copy loop: what does the link give a 640x480 frame (depth f32 + RGBA = 2.46 MB)

    page-locked by hipHostMalloc      vs   caller pages locked in place by hipHostRegister
    one hipMemcpyAsync per frame      vs   depth and colour on two streams (two SDMA queues)

Prints microseconds per frame and GB/s; each figure is the best of five runs of 200 frames, the copies of a run queued back to
back with ONE synchronisation at the end (= the asynchronous upload of the staging path) and, second column, with a
synchronisation per frame (= a registered call that returns when its own upload is through)."""
import ctypes as C
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hip error %d" % (what, rc))


W, H = 640, 480
nd, nc = W * H * 4, W * H * 4
N = 200
dev = vp()
ck(hip.hipMalloc(C.byref(dev), C.c_size_t(nd + nc)), "hipMalloc")
s1, s2 = vp(), vp()
ck(hip.hipStreamCreateWithFlags(C.byref(s1), 1), "stream")
ck(hip.hipStreamCreateWithFlags(C.byref(s2), 1), "stream")
ev = vp()
ck(hip.hipEventCreateWithFlags(C.byref(ev), 2), "event")


def run(src, two, sync_each):
    best = 1e9
    for _ in range(5):
        ck(hip.hipDeviceSynchronize(), "sync")
        t0 = time.perf_counter()
        for k in range(N):
            p = src + (k % 8) * (nd + nc)
            if two:
                ck(hip.hipMemcpyAsync(vp(dev.value + nd), vp(p + nd), C.c_size_t(nc), 1, s2), "copy")
                ck(hip.hipEventRecord(ev, s2), "record")
                ck(hip.hipMemcpyAsync(dev, vp(p), C.c_size_t(nd), 1, s1), "copy")
                ck(hip.hipStreamWaitEvent(s1, ev, 0), "wait")
            else:
                ck(hip.hipMemcpyAsync(dev, vp(p), C.c_size_t(nd + nc), 1, s1), "copy")
            if sync_each:
                ck(hip.hipStreamSynchronize(s1), "sync")
        ck(hip.hipStreamSynchronize(s1), "sync")
        best = min(best, (time.perf_counter() - t0) / N)
    return 1e6 * best, (nd + nc) / best / 1e9


pinned = vp()
ck(hip.hipHostMalloc(C.byref(pinned), C.c_size_t(8 * (nd + nc)), 0), "hipHostMalloc")
C.memset(pinned, 1, 8 * (nd + nc))
arr = np.ones(8 * (nd + nc), np.uint8)
ck(hip.hipHostRegister(vp(arr.ctypes.data), C.c_size_t(arr.nbytes), 0), "hipHostRegister")
print("%-44s %22s %26s" % ("2.46 MB per frame", "queued back to back", "one synchronisation per frame"))
for name, src in (("hipHostMalloc", pinned.value), ("hipHostRegister (numpy pages)", arr.ctypes.data)):
    for two in (False, True):
        a = run(src, two, False)
        b = run(src, two, True)
        print("%-30s %-13s %8.1f us %6.1f GB/s %12.1f us %6.1f GB/s" % (name, "two streams" if two else "one copy", a[0], a[1], b[0], b[1]))
