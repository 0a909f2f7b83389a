#!/usr/bin/env python3
"""Region time stamps of the fused rollout (analysis build: make EXTRA=-DMPPI_TRACE
OUT=../lib/trace/libmppi_gpu_amd.so OBJDIR=../lib/trace/obj; run with
MPPI_GPU_AMD_LIB=mppi_gpu_amd/lib/trace/libmppi_gpu_amd.so).  Prints, over the blocks of one
launch, when each region boundary is reached relative to the earliest block start (s_memtime
ticks, 100 MHz on gfx950 -> 10 ns)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel, _capi

A, K, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (2, 10000, 200)))
chunks = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lib = _capi.load()
lib.mppi_debug_trace.restype = C.c_int
lib.mppi_debug_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.mppi_debug_trace(None, 0) == 0
c = ol.make_case(A, 1, T, seed=0, u_scale=0.0)
m = PointMassModel(K, T, float(c["dt"]), 2 * A, A)
m.set_tuning(chunks=chunks, strict=0, max_blocks=0)
m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
for _ in range(50):
    m.solve_async()
m.sync_act()
geo = m.geometry()
grid = geo["grid"]
buf = np.zeros((grid, 16), np.uint64)
m.solve_async(); m.sync_act()
assert lib.mppi_debug_trace(buf.ctypes.data, grid) == 0
t = buf.astype(np.int64)
d = t[:, :11] - t[:, :1]          # per block: ticks since its own entry (counters differ per XCD)
names = ["entry", "pass1a done", "barrier 1", "1b+scan", "pass2", "min+exp", "nreduce", "barrier 2",
         "fold", "all tiles", "exit"]
print("geometry", geo, "blocks", grid)
print(f"{'stamp':14s} {'p10':>8s} {'median':>8s} {'p90':>8s}   median step")
prev = 0.0
for i, n in enumerate(names):
    r = d[:, i]
    med = float(np.median(r))
    print(f"{n:14s} {np.percentile(r, 10):8.0f} {med:8.0f} {np.percentile(r, 90):8.0f}   {med - prev:8.0f}")
    prev = med
m.close()
