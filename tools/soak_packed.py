#!/usr/bin/env python3
"""Soak of the riding combine in PACKED launches of many tiles per block (round 3: a combine rides
in packed launches of any length, the launch may hold more blocks than the chip): long chains of
asynchronously enqueued solves against the same chains with every combine flushed on its own
(blocking get_act).  A stale or torn hand-over of the controls, or a rollout block that started
with controls of the wrong solve, would change the next solve and, through the chain, the final
bits.  Then the single-process sharded host (3 shard engines on this one device, direct exchange).
usage: tools/soak_packed.py [n_solves]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel
from mppi_gpu_amd.node import NodePointMassModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bad = 0
# (A, K, T, packing, lambda)
for A, K, T, packing, lam in ((3, 100000, 200, 0, 1.0), (3, 100000, 200, 0, 150.0), (3, 30011, 200, 4, 1.0),
                              (3, 60000, 50, 4, 1.0), (2, 40000, 200, 8, 30.0), (1, 50000, 203, 4, 1.0),
                              (4, 30000, 37, 10, 5.0)):
    c = ol.make_case(A, 1, T, seed=300 + A, u_scale=0.02)
    res = []
    for blocking in (False, True):
        with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
            if packing:
                m.set_packing(packing)
            m.set_seed(11 + A)
            m.set_params(lam)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            for i in range(n):
                if blocking:
                    m.get_act()
                else:
                    m.solve_async()
                    if i % 997 == 996:          # an occasional synchronisation in the middle
                        m.sync_act()
            act = m.sync_act()
            res.append((act.copy(), m.get_u().copy(), m.geometry(), m.launch_counts()))
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    finite = bool(np.all(np.isfinite(res[0][1])))
    g, cnt = res[0][2], res[0][3]
    print(f"A={A} K={K} T={T} lambda={lam} packed={g['packed']} grid={g['grid']} tile_groups={g['tile_groups']} "
          f"riding={cnt['riding']}/{cnt['rollout']} resident={cnt['resident_ride']} solves={n} equal={same} finite={finite}",
          flush=True)
    bad += 0 if (same and finite and g["packed"] and cnt["riding"] >= n - 10) else 1

# the single-process sharded host: 3 shards on this device, direct exchange riding
A, K, T = 2, 6000, 120
c = ol.make_case(A, 1, T, seed=9, u_scale=0.02)
res = []
for blocking in (False, True):
    with NodePointMassModel(K, T, float(c["dt"]), 2 * A, A, devices=[0, 0, 0], transport="direct") as m:
        m.set_seed(5)
        m.set_timeout(5.0)
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        for i in range(n):
            if blocking:
                m.get_act()
            else:
                m.solve_async()
                if i % 50 == 49:
                    m.sync_act()
        res.append((m.sync_act().copy(), m.get_u().copy()))
same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
print(f"node host, 3 shards on one device, direct: solves={n} equal={same}", flush=True)
bad += 0 if same else 1
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
