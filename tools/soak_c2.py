#!/usr/bin/env python3
"""Soak at the bench shape: 2-D K=1e4 T=200, riding vs flushed chains (see tools/soak.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
A, K, T = 2, 10000, 200
c = ol.make_case(A, 1, T, seed=0, u_scale=0.0)
res = []
for blocking in (False, True):
    with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        for i in range(n):
            m.get_act() if blocking else m.solve_async()
        res.append((m.sync_act().copy(), m.get_u().copy()))
same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
print("C2 soak", n, "solves: equal =", same, "finite =", bool(np.all(np.isfinite(res[0][1]))), "act", res[0][0])
sys.exit(0 if same else 1)
