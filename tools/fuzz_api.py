#!/usr/bin/env python3
"""Seeded random walks over the engine's C ABI (through the Python mirror), each run under two
configurations that must end in the SAME BITS:

  A: a pending combine rides in the next rollout launch (pipeline mode 0) + noise prefetch forced
     (every blocking call draws the next solve's noise behind its combine)
  B: every combine flushed right behind its rollout (same combine code on its own launch) + no
     prefetch

The walk mixes blocking calls, asynchronous solves, state updates, parameter changes that keep a
prefetched buffer valid (lambda, inv_s) and changes that must discard it (sigma, seed, noise store,
geometry, injected noise on and off), an action limit, and every read-out call.  Whatever is handed out along the way (actions,
controls, noise, costs, beta, nabla, weights) is compared bit for bit.

    tools/fuzz_api.py [n_walks [steps_per_walk [first_seed]]]        (needs an MI355X)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol                                  # make_case only: shapes and constants
from mppi_gpu_amd import PointMassModel

SHAPES = [(2, 10000, 200), (3, 3000, 50), (1, 700, 33), (3, 20011, 200), (4, 2500, 64), (2, 1200, 17),
          (3, 60000, 200)]


def walk(seed, steps, config):
    rng = np.random.default_rng(seed)
    A, K, T = SHAPES[int(rng.integers(len(SHAPES)))]
    c = ol.make_case(A, 1, T, seed=int(rng.integers(1 << 30)), u_scale=0.03)
    out = []
    with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
        m.set_pipeline(0)
        m.set_noise_prefetch(2 if config == "A" else 0)
        m.set_seed(int(rng.integers(1 << 40)))
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        x = c["x0"].copy()
        lam, sig = 1.0, [0.025] * A
        solved = False                 # (read-outs of a solve need one since the last set_data)
        for _ in range(steps):
            op = int(rng.integers(100))
            if op >= 80 and op < 85 and not solved:
                op = 0
            if op < 30:
                out.append(("act", m.get_act()))
                solved = True
            elif op < 50:
                for _ in range(int(rng.integers(1, 5))):
                    m.solve_async()
                    if config == "B":
                        m.flush_async()
                solved = True
                if rng.integers(2):
                    out.append(("sync", m.sync_act()))
            elif op < 62:
                x = (x * np.float32(0.9) + rng.standard_normal(2 * A).astype(np.float32) * np.float32(0.01))
                m.set_x(x)
            elif op < 68:
                lam = float(rng.choice([0.5, 1.0, 3.0, 20.0]))
                m.set_params(lam, sigma=sig)
            elif op < 72:
                sig = [float(rng.choice([0.025, 0.05, 0.01]))] * A
                m.set_params(lam, sigma=sig)
            elif op < 75:
                m.set_seed(int(rng.integers(1 << 40)))
            elif op < 80:
                out.append(("u", m.get_u()))
            elif op < 85:
                inf = m.get_inf(x=False)
                for k in ("u", "e", "cost", "beta", "nabla", "weight"):
                    out.append((k, np.asarray(inf[k]).copy()))
            elif op < 88:
                m.set_noise_store(bool(rng.integers(2)))
            elif op < 91:
                if A != 4 or rng.integers(2):
                    m.set_tuning(chunks=0, strict=False, max_blocks=int(rng.choice([0, 0, 24, 3])))
            elif op < 93:
                m.set_packing(int(rng.choice([0, -1])))
            elif op < 96:
                m.memcpy_set_data(x, c["U"], c["goal"], c["w"])
                solved = False
            elif op < 97:
                m.flush_async()
            elif op < 98:
                if K * T * A <= 2_000_000:       # the caller's own noise for a while, or back to sampling
                    if rng.integers(2):
                        m.set_noise((rng.standard_normal((K, T, A)) * 0.025).astype(np.float32))
                    else:
                        m.set_noise(None)
                else:
                    lim = None if rng.integers(2) else [float(rng.choice([0.02, 0.2]))] * A
                    m.set_action_limit(lim)
            else:
                out.append(("act2", m.get_act()))
                out.append(("act3", m.get_act()))
                solved = True
        out.append(("final_act", m.get_act()))
        out.append(("final_u", m.get_u()))
        cnt = m.prefetch_counts()
    return (A, K, T), out, cnt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    bad = 0
    used = 0
    for seed in range(first, first + n):
        shape, a, cnt = walk(seed, steps, "A")
        _, b, _ = walk(seed, steps, "B")
        used += cnt["used"]
        ok = len(a) == len(b) and all(ka == kb and np.array_equal(va, vb) for (ka, va), (kb, vb) in zip(a, b))
        if not ok:
            bad += 1
            first_bad = next((i for i, ((ka, va), (kb, vb)) in enumerate(zip(a, b))
                              if ka != kb or not np.array_equal(va, vb)), None)
            print(f"seed {seed} shape {shape}: MISMATCH at read-out {first_bad} "
                  f"({a[first_bad][0] if first_bad is not None else 'length'})")
        else:
            print(f"seed {seed} shape {shape}: {len(a)} read-outs equal, {cnt['used']} rollouts on prefetched noise")
    print(f"{n} walks x {steps} calls: {bad} mismatches, {used} rollouts loaded prefetched noise")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
