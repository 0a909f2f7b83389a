"""Seeded sweep of the engine's OWN geometry (no packing / chunks forced) against the oracle: random
act_dim, K (1 ... 40 000), horizon (1 ... 320), lambda, persistent-grid cap, cost weights (some zero),
injected noise, through tests/test_gpu_parity._check_solve.  The geometry rules of round 3 choose
among many more (kernel, lanes per trajectory) pairs than the fixed cases of the test suite visit.
Run from the repository root on a GPU box: python tools/sweep_auto.py [trials [first_seed]]"""
import collections
import os
import sys

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as ol
import test_gpu_parity as tg

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
seen = collections.Counter()
rng = np.random.default_rng(seed0)
for trial in range(trials):
    A = int(rng.integers(1, 5))
    T = int(rng.choice([1, 2, 3, 7, 16, 33, 50, 51, 64, 100, 128, 199, 200, 203, 256, 320]))
    K = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, 2500, 3000, 5000, 7000, 10000, 15000, 20011, 30000, 40000]))
    if K * T * A > 2.5e7:
        K = max(1, int(2.5e7 / (T * A)))
    lam = float(rng.choice([0.5, 1.0, 2.0, 50.0]))
    c = ol.make_case(A, K, T, seed=70000 + 1000 * seed0 + trial, u_scale=float(rng.choice([0.0, 0.05, 0.5])))
    c["goal"] = rng.standard_normal(2 * A).astype(np.float32)
    w = np.abs(rng.standard_normal(2 * A) * 5) * (rng.random(2 * A) > 0.2)
    if not w.any():
        w[0] = 1.0
    c["w"] = w.astype(np.float32)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam)
    with tg._model(None, A, K, T, c, max_blocks=int(rng.choice([0, 0, 0, 1, 3, 24]))) as m:
        m.set_params(lam)
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf(x=False)
        geo = m.geometry()
    seen[("packed/%d" % geo["groups_per_lane"]) if geo["packed"] else ("row C=%d" % geo["chunks"])] += 1
    try:
        tg._check_solve(act, inf, ref, cost_exact=False, lam=lam, T=T, tag=f"auto {seed0}/{trial} A{A} K{K} T{T}")
    except AssertionError as e:
        bad += 1
        msg = " ".join(str(e).split())
        print("FAIL", seed0, trial, "A%d K%d T%d lam %g max_blocks-grid %d %s:" % (
            A, K, T, lam, geo["grid"], ("packed/%d" % geo["groups_per_lane"]) if geo["packed"] else ("row C=%d" % geo["chunks"])),
            msg[:120], "...", msg[-160:], flush=True)
print("auto-geometry sweep: %d trials, failures: %d; geometries met: %s" % (trials, bad, dict(seen)))
