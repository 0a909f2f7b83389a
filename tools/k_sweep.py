#!/usr/bin/env python3
"""Solve time and rollouts/s against the number of samples K (T = 200, 2-D and 3-D), solves back to
back on one GPU, with the geometry the engine chose: where the latency chain of the small launches
ends and the VALU-bound regime begins.   tools/k_sweep.py          (needs an MI355X)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel

T = 200
print("A      K  packed grid  tiles/block  us/solve  rollouts/s   noise GB/s (algorithmic)")
for A in (2, 3):
    for K in (1000, 3000, 10000, 20000, 30000, 50000, 100000, 200000, 400000, 1000000):
        c = ol.make_case(A, 1, T, seed=5, u_scale=0.0)
        with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
            m.set_seed(0)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.05:          # clocks up
                for _ in range(20):
                    m.solve_async()
                m.sync_act()
            n = max(50, min(4000, int(0.15 / (max(K, 10000) * 7e-10))))
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(n):
                    m.solve_async()
                m.sync_act()
                best = min(best, (time.perf_counter() - t0) / n)
            g = m.geometry()
            print("%d %7d  %-5s %5d  %6.2f      %8.2f  %.3e   %.0f" % (
                A, K, g["packed"], g["grid"], g["tile_groups"] / g["grid"], best * 1e6, K / best,
                4.0 * K * T * A / best / 1e9), flush=True)
