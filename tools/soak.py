#!/usr/bin/env python3
"""Soak: long chains of solves whose combines ride (asynchronous enqueue) against the same chains
with every combine flushed on its own (blocking get_act).  Any stale or torn hand-over of the
controls would change the next solve and, through the chain, the final bits.
usage: tools/soak.py [n_solves] [rounds]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(7)
bad = 0
for r in range(rounds):
    A = int(rng.integers(1, 5))
    T = int(rng.choice([20, 50, 120, 200]))
    K = int(rng.choice([500, 3000, 10000, 20000]))
    c = ol.make_case(A, K, T, seed=300 + r, u_scale=0.02)
    res = []
    for blocking in (False, True):
        with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
            m.set_seed(11 + r)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            for i in range(n):
                if blocking:
                    m.get_act()
                else:
                    m.solve_async()
                    if i % 997 == 996:          # an occasional synchronisation in the middle
                        m.sync_act()
            act = m.sync_act()
            res.append((act.copy(), m.get_u().copy(), m.geometry()))
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    finite = bool(np.all(np.isfinite(res[0][1])))
    print(f"round {r}: A={A} K={K} T={T} grid={res[0][2]['grid']} solves={n} equal={same} finite={finite}",
          flush=True)
    bad += 0 if (same and finite) else 1
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad} rounds)")
sys.exit(1 if bad else 0)
