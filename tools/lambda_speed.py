#!/usr/bin/env python3
"""Does the solve time depend on lambda (i.e. on the DATA in the weighted-noise sums)?  C3 and C2,
alternating lambda, same engine.  python tools/lambda_speed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mppi_gpu_amd import PointMassModel
import bench

def run(m, n):
    for _ in range(50): m.solve_async()
    m.sync_act()
    m.set_profiling(8)
    t0 = time.perf_counter()
    for _ in range(n): m.solve_async()
    m.sync_act()
    dt = (time.perf_counter() - t0) / n
    k, kn = m.kernel_ms(0)
    m.set_profiling(0)
    return dt * 1e6, k * 1e3

for wl, n in (("c3", 600), ("c2", 3000)):
    A, K, T, _ = bench.WORKLOADS[wl]
    c = bench.make_inputs(A, T)
    with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
        m.set_seed(0)
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        for _ in range(300): m.solve_async()
        m.sync_act()
        for rep in range(2):
            for lam in (1.0, 3.0, 11.0, 100.0, 1e4, 0.05, 1.0):
                m.set_params(lam)
                us, kus = run(m, n)
                w = m.get_inf(x=False, u=False, e=False, cost=False, beta=False, nabla=False)["weight"]
                nz = int((w > 0).sum())
                print(f"{wl} lambda {lam:8.2f}: {us:7.2f} us/solve kernel {kus:7.2f} us  ess {bench.ess_of(w):9.1f} nonzero weights {nz}", flush=True)
        m.set_params(1.0)
        m.set_noise_store(False)
        for lam in (1.0, 11.0):
            m.set_params(lam)
            us, kus = run(m, n)
            print(f"{wl} NO STORE lambda {lam:6.2f}: {us:7.2f} us/solve kernel {kus:7.2f}", flush=True)
