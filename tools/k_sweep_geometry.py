#!/usr/bin/env python3
"""For sizes around the row-aligned / packed transition: the engine's own geometry against each
kernel forced (mppi_set_packing -1 / n).   tools/k_sweep_geometry.py      (needs an MI355X)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel, MppiError

NGS = {1: [4], 2: [5, 8], 3: [4], 4: [10]}


def timed(A, K, T, packing):
    c = ol.make_case(A, 1, T, seed=5, u_scale=0.0)
    with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
        m.set_seed(0)
        try:
            m.set_packing(packing)
        except MppiError:
            return None, None
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.04:
            for _ in range(20):
                m.solve_async()
            m.sync_act()
        n = max(100, min(3000, int(0.1 / (max(K, 10000) * 7e-10))))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                m.solve_async()
            m.sync_act()
            best = min(best, (time.perf_counter() - t0) / n)
        g = m.geometry()
        return best * 1e6, g


for A, T in ((1, 200), (2, 200), (3, 200), (4, 200), (3, 50), (2, 50)):
    for K in (1000, 3000, 5000, 10000, 15000, 20000, 30000, 50000):
        row = "A %d T %3d K %6d:" % (A, T, K)
        ta, ga = timed(A, K, T, 0)
        row += "  auto %6.2f us (%s, grid %d)" % (ta, "packed" if ga["packed"] else "row", ga["grid"])
        tr, gr = timed(A, K, T, -1)
        row += "   row-aligned %s" % ("%6.2f" % tr if tr else "  n/a ")
        for ng in NGS[A]:
            tp, gp = timed(A, K, T, ng)
            row += "   packed/%d %s" % (ng, "%6.2f" % tp if tp else "  n/a ")
        print(row, flush=True)
