#!/usr/bin/env python3
"""Generate the committed golden fixtures. Run in the authoring container:

    python tests/golden/make_golden.py

cost_ref.npz      inputs + outputs of the REFERENCE's own Cost::step_cost / Cost::final_cost
                  (reference src/cost.cu compiled unmodified into oracle/_ref by oracle/Makefile,
                  called through oracle/ref_cost_shim.cpp).  This is real reference output and
                  pins the cost term of the oracle and, through it, of the HIP kernels.
solve_*.npz       inputs + outputs of full solves computed by the CPU oracle
                  (oracle/mppi_oracle.c).  Oracle-generated: they pin the HIP path against
                  regressions of the oracle itself; the reference holds no vector for these.
Only data is written: inputs and expected outputs, no reference source text.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def gen_cost_ref():
    ol.build()
    r = ol.ref_cost_lib()
    if r is None:
        raise SystemExit("oracle/_ref/libref_cost.so missing: /root/reference not available")
    rng = np.random.default_rng(20261004)
    rows = []
    for _ in range(600):
        A = int(rng.integers(1, 5))
        S = 2 * A
        scale = rng.choice([1e-3, 0.1, 1.0, 30.0])
        x = (rng.standard_normal(8) * scale).astype(np.float32)
        u = (rng.standard_normal(4) * scale).astype(np.float32)
        e = (rng.standard_normal(4) * 0.025).astype(np.float32)
        w = np.abs(rng.standard_normal(8) * 10).astype(np.float32)
        g = rng.standard_normal(8).astype(np.float32)
        inv = rng.uniform(0.5, 2.0, 4).astype(np.float32)
        lam = np.float32(rng.uniform(0.2, 3.0))
        p = lambda a: a.ctypes.data_as(ol.fp)  # noqa: E731
        sc = np.float32(r.ref_step_cost(p(x), p(u), p(e), p(w), p(g), C.c_float(lam), p(inv), S, A))
        fc = np.float32(r.ref_final_cost(p(x), p(w), p(g), S))
        rows.append((A, x, u, e, w, g, inv, lam, sc, fc))
    np.savez_compressed(
        os.path.join(HERE, "cost_ref.npz"),
        A=np.array([r_[0] for r_ in rows], np.int32),
        x=np.stack([r_[1] for r_ in rows]), u=np.stack([r_[2] for r_ in rows]),
        e=np.stack([r_[3] for r_ in rows]), w=np.stack([r_[4] for r_ in rows]),
        goal=np.stack([r_[5] for r_ in rows]), inv_s=np.stack([r_[6] for r_ in rows]),
        lam=np.array([r_[7] for r_ in rows], np.float32),
        step_cost=np.array([r_[8] for r_ in rows], np.float32),
        final_cost=np.array([r_[9] for r_ in rows], np.float32))


SOLVE_CASES = [
    # name, A, K, T, seed
    ("solve_1d_K100_T50", 1, 100, 50, 11),      # BASELINE config 1 shape
    ("solve_2d_K3_T12", 2, 3, 12, 12),          # the reference's mppi-config-test.yaml shape
    ("solve_2d_K257_T50", 2, 257, 50, 13),      # shipped horizon, ragged K
    ("solve_2d_K128_T200", 2, 128, 200, 14),    # BASELINE config 2 horizon
    ("solve_3d_K96_T200", 3, 96, 200, 15),      # BASELINE config 3 horizon
    ("solve_3d_K300_T51", 3, 300, 51, 16),      # odd horizon
    ("solve_2d_K64_T33", 2, 64, 33, 17),        # horizon not a multiple of the Philox block
]


def gen_solves():
    for name, A, K, T, seed in SOLVE_CASES:
        c = ol.make_case(A, K, T, seed)
        out = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], f64_update=True)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), A=A, K=K, T=T, dt=c["dt"],
                            x0=c["x0"], U=c["U"], E=c["E"], goal=c["goal"], w=c["w"],
                            cost=out["cost"], beta=out["beta"], nabla=out["nabla"],
                            weights=out["weights"], U_next=out["U"], next_act=out["next_act"])


if __name__ == "__main__":
    gen_cost_ref()
    gen_solves()
    print("golden fixtures written to", HERE)
