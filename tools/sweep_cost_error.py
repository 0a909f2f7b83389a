#!/usr/bin/env python3
"""Cost agreement of the fused kernels with the oracle as a function of the horizon T (GPU box):
worst relative difference of a path cost over K samples, packed and row-aligned kernel, act_dim
1..4, T from 16 to the longest horizon the 64 KiB LDS budget takes.  The numbers DESIGN section 5
quotes and the bar tests/test_gpu_parity.py asserts (rtol <= max(3e-6, 0.25 * T * 2^-24)) come from
this script:   python tools/sweep_cost_error.py > gpurun_out/sweep_cost_error.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                      # noqa: E402  (the checker; this is a test tool)
from mppi_gpu_amd import PointMassModel, MppiError   # noqa: E402

NG = {1: 4, 2: 8, 3: 4, 4: 10}
K = 2000
print("# A  T     kernel   cost_rtol_max  cost_rtol_median  bar=max(3e-6,0.25*T*2^-24)  dU/scale")
for A in (1, 2, 3, 4):
    for T in (16, 50, 100, 200, 300, 400, 512, 640, 800, 1000, 1400, 2000):
        c = ol.make_case(A, K, T, seed=900 + A * 31 + T)
        ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=200.0)
        for kernel in ("packed", "row"):
            try:
                with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
                    m.set_packing(NG[A] if kernel == "packed" else -1)
                    m.set_params(200.0)
                    m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
                    m.set_noise(c["E"])
                    m.get_act()
                    inf = m.get_inf(x=False, e=False)
            except MppiError as ex:
                print(f"  {A}  {T:5d} {kernel:7s}  -- {str(ex)[:70]}")
                continue
            r = np.abs(inf["cost"] - ref["cost"]) / np.abs(ref["cost"])
            scale = max(float(np.abs(ref["U"]).max()), 0.025)
            bar = max(3e-6, 0.25 * T * 2.0 ** -24)
            print(f"  {A}  {T:5d} {kernel:7s}  {r.max():.2e}       {np.median(r):.2e}          "
                  f"{bar:.2e}   {np.abs(inf['u'] - ref['U']).max() / scale:.1e}"
                  f"{'   ABOVE BAR' if r.max() > bar else ''}")
