#!/usr/bin/env python3
"""Seeded random walks over the single-process sharded host (libmppi_gpu_amd_sharded.so through
mppi_gpu_amd.node): 2 or 3 shard engines and worker threads on ONE device, the same walk through
the `direct` transport (peer stores from inside the combine kernel; the exchange of back-to-back
solves rides in the next rollout launch) and through `copy` (hipMemcpyPeerAsync + finish): every
read-out must be equal bit for bit.  Exercises the worker hand-over, the riding exchange and the
blocking path's noise prefetch of the shard engines.

    tools/fuzz_node.py [n_walks [steps_per_walk [first_seed]]]        (needs an MI355X)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol                                  # make_case only: shapes and constants
from mppi_gpu_amd.node import NodePointMassModel

SHAPES = [(2, 10000, 200), (3, 3000, 50), (1, 700, 33), (3, 20011, 200), (4, 2500, 64), (2, 1201, 17)]


def walk(seed, steps, transport):
    rng = np.random.default_rng(seed)
    A, K, T = SHAPES[int(rng.integers(len(SHAPES)))]
    n_shards = int(rng.integers(2, 4))
    c = ol.make_case(A, 1, T, seed=int(rng.integers(1 << 30)), u_scale=0.03)
    out = []
    with NodePointMassModel(K, T, float(c["dt"]), 2 * A, A, devices=[0] * n_shards,
                            transport=transport) as m:
        m.set_seed(int(rng.integers(1 << 40)))
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        x = c["x0"].copy()
        lam, sig = 1.0, [0.025] * A
        solved = False
        for _ in range(steps):
            op = int(rng.integers(100))
            if 80 <= op < 86 and not solved:
                op = 0
            if op < 35:
                out.append(("act", m.get_act().copy()))
                solved = True
            elif op < 55:
                for _ in range(int(rng.integers(1, 5))):
                    m.solve_async()
                solved = True
                if rng.integers(2):
                    out.append(("sync", m.sync_act().copy()))
            elif op < 67:
                x = (x * np.float32(0.9) + rng.standard_normal(2 * A).astype(np.float32) * np.float32(0.01))
                m.set_x(x)
            elif op < 72:
                lam = float(rng.choice([0.5, 1.0, 3.0, 20.0]))
                m.set_params(lam, sigma=sig)
            elif op < 75:
                sig = [float(rng.choice([0.025, 0.05, 0.01]))] * A
                m.set_params(lam, sigma=sig)
            elif op < 78:
                m.set_seed(int(rng.integers(1 << 40)))
            elif op < 80:
                out.append(("u", m.get_u().copy()))
            elif op < 86:
                inf = m.get_inf(x=False)
                for k in ("u", "e", "cost", "beta", "nabla", "weight"):
                    out.append((k, np.asarray(inf[k]).copy()))
            elif op < 92:
                m.memcpy_set_data(x, c["U"], c["goal"], c["w"])
                solved = False
            elif op < 95:
                if K * T * A <= 2_000_000:
                    if rng.integers(2):
                        m.set_noise((rng.standard_normal((K, T, A)) * 0.025).astype(np.float32))
                    else:
                        m.set_noise(None)
            else:
                lim = None if rng.integers(2) else [float(rng.choice([0.02, 0.2]))] * A
                m.set_action_limit(lim)
        out.append(("final_act", m.get_act().copy()))
        out.append(("final_u", m.get_u().copy()))
    return (A, K, T, n_shards), out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    bad = 0
    reads = 0
    for seed in range(first, first + n):
        shape, a = walk(seed, steps, "direct")
        _, b = walk(seed, steps, "copy")
        ok = len(a) == len(b) and all(ka == kb and np.array_equal(va, vb) for (ka, va), (kb, vb) in zip(a, b))
        reads += len(a)
        if not ok:
            bad += 1
            first_bad = next((i for i, ((ka, va), (kb, vb)) in enumerate(zip(a, b))
                              if ka != kb or not np.array_equal(va, vb)), None)
            print(f"seed {seed} shape {shape}: MISMATCH at read-out {first_bad} "
                  f"({a[first_bad][0] if first_bad is not None else 'length'})", flush=True)
        else:
            print(f"seed {seed} shape {shape}: {len(a)} read-outs equal", flush=True)
    print(f"{n} walks x {steps} calls: {bad} mismatches in {reads} read-outs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
