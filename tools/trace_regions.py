#!/usr/bin/env python3
"""Region time stamps of the fused rollout (analysis build: make EXTRA=-DMPPI_TRACE
OUT=../lib/trace/libmppi_gpu_amd.so OBJDIR=../lib/trace/obj; run with
MPPI_GPU_AMD_LIB=mppi_gpu_amd/lib/trace/libmppi_gpu_amd.so).  Prints, over the blocks of one
launch, when each region boundary is reached relative to the earliest block start (s_memtime
ticks of the 100 MHz wall clock -> 10 ns)."""
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
packing = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lib = _capi.load()
lib.mppi_debug_trace.restype = C.c_int
lib.mppi_debug_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.mppi_debug_trace(None, 0) == 0
c = ol.make_case(A, 1, T, seed=0, u_scale=0.0)
m = PointMassModel(K, T, float(c["dt"]), 2 * A, A)
m.set_tuning(chunks=chunks, strict=0, max_blocks=0)
if packing:
    m.set_packing(packing)
m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
for _ in range(50):
    m.solve_async()
m.sync_act()
geo = m.geometry()
grid = geo["grid"]
ride = int(os.environ.get("TRACE_RIDE", "0"))       # 1: trace a launch that carries a combine
nb = 0
if ride:
    for _ in range(4):
        m.solve_async()                              # the last launch carried a combine
    m.sync_act()
    import math
    rs = min(8, max(1, math.ceil(grid / 640)))      # 16 row groups x 40 rows per lane and split
    nb = math.ceil(T * A / 16) * rs                  # kCombineCols = 16
else:
    m.solve_async(); m.sync_act()
buf = np.zeros((grid + nb, 16), np.uint64)
assert lib.mppi_debug_trace(buf.ctypes.data, grid + nb) == 0
if nb:
    tc = buf[:nb].astype(np.int64)
    t00 = min(tc[:, 0].min(), buf[nb:, 0].astype(np.int64).min())
    print("combine-role blocks:", nb, " start (10 ns ticks after first block) p50/max",
          np.median(tc[:, 0] - t00), (tc[:, 0] - t00).max(), " end p50/max",
          np.median(tc[:, 10] - t00), (tc[:, 10] - t00).max())
    for i, nm in [(8, "kernel arguments in"), (9, "m, s requested"), (1, "all loads requested"), (5, "beta known"), (2, "beta/nabla done"), (6, "row loads landed"),
                  (7, "row sums in LDS"), (3, "rows reduced"), (4, "splits met"), (10, "done")]:
        r = tc[:, i] - tc[:, 0]
        r = r[tc[:, i] > 0]
        if len(r):
            print(f"   combine {nm:20s} p50 {np.median(r):6.0f}  max {r.max():6d}  (n={len(r)})")
    buf = buf[nb:]
    print("rollout blocks start p50/max", np.median(buf[:, 0].astype(np.int64) - t00),
          (buf[:, 0].astype(np.int64) - t00).max(), " end max", (buf[:, 10].astype(np.int64) - t00).max())
t = buf.astype(np.int64)
if int(os.environ.get("MPPI_TRACE_TILE", "0")) > 0 and geo["packed"]:
    t = t[t[:, 11] > 0]           # packed kernel, a later tile: ticks since THAT tile's start
    t[:, 0] = t[:, 11]
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

# ---- where the slow blocks are: duration (entry -> all tiles done) by XCD (round-robin over block
#      index) and by dispatch half (second blocks of the CUs)
t = buf.astype(np.int64)
dur = (t[:, 9] - t[:, 0]).astype(np.float64)
start = t[:, 0] - t[:, 0].min()
print("block duration entry->all tiles: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f   (latest start %d)"
      % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), start.max()))
if geo["packed"] and (t[:, 12] > 0).all():
    print("  packed exit: wait for the block's slowest wave p50 %.0f p90 %.0f, merge + partial store p50 %.0f p90 %.0f"
          % (np.median(t[:, 12] - t[:, 9]), np.percentile(t[:, 12] - t[:, 9], 90),
             np.median(t[:, 10] - t[:, 12]), np.percentile(t[:, 10] - t[:, 12], 90)))
idx = np.arange(len(dur))
print("  by bid %% 8 (XCD):", " ".join("%.0f" % np.median(dur[idx % 8 == x]) for x in range(8)))
half = len(dur) // 2
print("  first half %.0f  second half %.0f" % (np.median(dur[:half]), np.median(dur[half:])))
order = np.argsort(dur)
print("  slowest blocks:", order[-12:], " fastest:", order[:12])

# ---- placement: blocks per CU (HW_ID of wave 0 at entry: CU_ID [11:8], SH_ID [12], SE_ID [15:13];
#      XCC_ID [3:0] of register 20), and the Philox-pass time by the number of blocks sharing the CU
hw = buf[:, 14].astype(np.int64)
xcc = buf[:, 15].astype(np.int64) & 0xF
cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
if hw.any():
    ids, cnt = np.unique(cu, return_counts=True)
    print("placement: %d distinct CUs hold the %d blocks; blocks per CU histogram:" % (len(ids), len(cu)),
          dict(zip(*np.unique(cnt, return_counts=True))))
    per = dict(zip(ids, cnt))
    share = np.array([per[x] for x in cu])
    for s in sorted(set(share)):
        sel = share == s
        print("   %d block(s) on the CU: n=%d  pass1a median %.0f  entry->all tiles median %.0f max %.0f  start median %.0f"
              % (s, sel.sum(), np.median(d[sel, 1]), np.median(dur[sel]), dur[sel].max(), np.median(start[sel])))
    print("   blocks per XCC:", dict(zip(*np.unique(xcc, return_counts=True))))
