"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Floating point, so tolerances are stated here:

  strict kernel  (one lane per trajectory, sequential)   cost: BIT-EXACT, beta: exact
  fused kernels  (lanes share a trajectory, scan + tree) cost: rtol max(3e-6, 0.35 T 2^-24): 4.2e-6 at T = 200
                                                         (achieved <= 2.0e-6 there), 2.1e-5 at T = 1000
  both                                                   nabla: rtol 2e-6 (strict) / max(1e-4, 8 ulp(c)/lambda) (fused)
                                                         weights: rtol 2e-5 (strict) / max(1e-3, 16 ulp(c)/lambda) (fused)
                                                         U, action: max-norm rel 1e-5
  (w_k = exp(-(c_k-beta)/lambda)/nabla, so an absolute cost difference d moves a weight by the
  RELATIVE amount d/lambda: with costs of a few hundred, the fused kernel's ~1 ulp(cost)
  re-association differences show up as ~1e-4 relative in single weights; they average out
  in the controls, which is the quantity the 1e-5 bar is stated on.)
                                                         (north_star: "controls ... to 1e-5 rel")
The U criterion is |U_gpu - U_oracle|_inf <= 1e-5 * max(|U_oracle|_inf, sigma): element-wise
relative error is meaningless where a control crosses zero.  The strict kernel meets it always,
and so does the fused (benchmarked) kernel wherever the weights average over enough samples:
the PLAIN 1e-5 bar is asserted for the BASELINE-size cases (configs 2, 3, 4 whole and 4's last
shard: they achieve 1e-7 .. 3e-6) and for every case whose effective sample size 1/sum(w^2) is at
least ESS_PLAIN.  Only below that (nearly one-hot weights at lambda = 1: the underflow and
cost-ramp cases, tiny batches, the random sweep) the fused kernel gets max(that, 4 ulp(max cost)/lambda * max|E|): two correct fp32
evaluations of a path cost of a few hundred differ by a few ulp (1.5e-5 each at 230), the
weights by that amount RELATIVE, and with a handful of effective samples nothing averages the
difference out.  The same holds between the reference's own nvcc build (FMA-contracted) and its
host arithmetic.

Every checked solve appends what it ACHIEVED (|dU|/scale, worst weight / cost rtol) to
gpurun_out/parity_r03.json; the committed copy is profiles/parity_r03.json.

SPREAD-OUT WEIGHTS (round 3).  With lambda = 1 and path costs of a few hundred every case with a
long horizon is nearly one-hot (effective sample size 1/sum(w^2) of 1 .. 3): the update then only
copies the best sample's noise.  The `spread` tests choose lambda PER CASE (bisection on the
oracle's costs, `_lambda_for_ess`) so that the effective sample size lands near K/3 and near
K/8 .. K/100, at the BASELINE horizon T = 200 and the BASELINE sizes, and assert there -- with no
ulp escape -- U, action: 1e-5 * max(|U|, sigma);  nabla: rtol 1e-5;  weights: rtol 1e-4.
"""
import atexit
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

SIGMA = 0.025


def _model(gpu, A, K, T, case, chunks=0, strict=False, max_blocks=0):
    from mppi_gpu_amd import PointMassModel
    m = PointMassModel(K, T, float(case["dt"]), 2 * A, A)
    m.set_tuning(chunks=chunks, strict=strict, max_blocks=max_blocks)
    m.memcpy_set_data(case["x0"], case["U"], case["goal"], case["w"])
    return m


ESS_PLAIN = 32.0         # effective samples from which the fused kernel must meet the plain bar


def cost_rtol(T):
    """Bar on the path costs of the fused kernels against the oracle, as a function of the horizon."""
    return max(3e-6, 0.35 * T * 2.0 ** -24)
_RECORDS = []


def _dump_records():
    if not _RECORDS:
        return
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_r03.json"), "w") as f:
            json.dump({"bar": "|U_gpu - U_oracle|_inf / max(|U_oracle|_inf, sigma) <= 1e-5 "
                              "(plain) wherever ess >= %g" % ESS_PLAIN,
                       "cases": _RECORDS}, f, indent=1)
    except OSError:
        pass


atexit.register(_dump_records)


def _lambda_for_ess(cost1, c_ctrl, target):
    """lambda at which the softmax over the path costs has the effective sample size `target`.
    cost1 = path costs at lambda 1, c_ctrl = their control term sum_t u.inv_s.e (the stage cost
    multiplies it by lambda, src/cost.cu:46): cost(lambda) = cost1 + (lambda - 1) c_ctrl.
    The ESS grows with lambda; bisection in log(lambda).  Only CHOOSES the case: what the case
    achieved is taken from the oracle's weights afterwards."""
    c1 = np.asarray(cost1, np.float64)
    cc = np.asarray(c_ctrl, np.float64)

    def ess(lam):
        z = -(c1 + (lam - 1.0) * cc) / lam
        w = np.exp(z - z.max())
        return float(w.sum() ** 2 / np.sum(w * w))

    lo, hi = 1e-2, 1e7
    for _ in range(60):
        mid = float(np.sqrt(lo * hi))
        lo, hi = (mid, hi) if ess(mid) < target else (lo, mid)
    return float(np.float32(np.sqrt(lo * hi)))


def _ctrl_term(U, E):
    """sum_t sum_a U[t,a] E[k,t,a] per sample (inv_s = 1), float64."""
    K = E.shape[0]
    return E.reshape(K, -1).astype(np.float64) @ np.asarray(U, np.float64).reshape(-1)


def _check_solve(got_act, inf, ref, cost_exact, tag="", lam=1.0, emax=4.5 * SIGMA, plain=False,
                 spread=0.0, T=None):
    """plain=True: the plain 1e-5 bar whatever the effective sample size (the BASELINE-size
    cases: they hold it with a margin, profiles/parity_r03.json).
    spread=S > 0: a spread-out-weights case -- the oracle's effective sample size must be >= S,
    and the plain bars apply to everything: U / action 1e-5, nabla rtol 1e-5, weights rtol 1e-4."""
    plain = plain or spread > 0
    wref = ref["weights"].astype(np.float64)
    ess = float(1.0 / np.sum(wref * wref)) if wref.sum() > 0 else 0.0
    scale0 = max(float(np.abs(ref["U"]).max()), SIGMA)
    rec = {"case": tag, "kernel": "strict" if cost_exact else "fused", "K": int(ref["cost"].size),
           "T": int(T if T is not None else inf["u"].shape[0]), "lambda": float(lam),
           "ess": round(ess, 1), "spread": bool(spread > 0),
           "dU_over_scale": float(np.abs(inf["u"] - ref["U"]).max() / scale0),
           "dact_over_scale": float(np.abs(got_act - ref["next_act"]).max() / scale0),
           "cost_rtol": float(np.max(np.abs(inf["cost"] - ref["cost"]) / np.abs(ref["cost"]))),
           "weight_rtol": float(np.max(np.abs(inf["weight"] - ref["weights"])
                                       / np.maximum(ref["weights"], 1e-30)
                                       * (ref["weights"] > 1e-12))),
           "nabla_rtol": float(abs(inf["nabla"] - ref["nabla"]) / ref["nabla"]),
           "plain_bar": bool(cost_exact or plain or ess >= ESS_PLAIN)}
    _RECORDS.append(rec)
    if cost_exact:
        assert np.array_equal(inf["cost"], ref["cost"]), f"{tag}: cost not bit-exact"
        assert np.float32(inf["beta"]) == ref["beta"], tag
    else:
        # (the worst of a million samples reaches 2.0e-6 at T = 200: 600 rounded additions per
        #  path cost; the typical sample is at 4e-7).  Two fp32 evaluations of a T-step recurrence
        #  drift apart with T: the bar is COST_RTOL(T) = max(3e-6, 0.35 T 2^-24) -- 3e-6 up to
        #  T = 143, 4.2e-6 at 200, 1.1e-5 at 512, 2.1e-5 at 1000; tools/sweep_cost_error.py
        #  (profiles/r03_sweep_cost_error.txt) finds <= 0.23 T 2^-24 in both fused kernels alike.
        c_rtol = cost_rtol(int(T if T is not None else inf["u"].shape[0]))
        np.testing.assert_allclose(inf["cost"], ref["cost"], rtol=c_rtol, atol=0, err_msg=tag)
        np.testing.assert_allclose(inf["beta"], ref["beta"], rtol=c_rtol, err_msg=tag)
    # fused kernel: a weight moves by (cost difference)/lambda RELATIVE, i.e. by a few ulp(cost)/lambda
    ulp_c = float(np.spacing(np.float32(np.abs(ref["cost"]).max()))) / lam
    if spread > 0:
        assert ess >= spread, f"{tag}: effective sample size {ess:.1f} below the {spread:.0f} the case is for"
        np.testing.assert_allclose(inf["nabla"], ref["nabla"], rtol=1e-5, err_msg=tag)
        np.testing.assert_allclose(inf["weight"], ref["weights"], rtol=1e-4, atol=1e-12, err_msg=tag)
    else:
        np.testing.assert_allclose(inf["nabla"], ref["nabla"],
                                   rtol=2e-6 if cost_exact else max(1e-4, 8 * ulp_c), err_msg=tag)
        np.testing.assert_allclose(inf["weight"], ref["weights"],
                                   rtol=2e-5 if cost_exact else max(1e-3, 16 * ulp_c),
                                   atol=1e-12, err_msg=tag)
    scale = max(float(np.abs(ref["U"]).max()), SIGMA)
    tol = 1e-5 * scale
    if not cost_exact and not plain and ess < ESS_PLAIN:   # nearly one-hot weights only (docstring)
        if "e" in inf:
            emax = float(np.abs(inf["e"]).max())
        tol = max(tol, 4 * float(np.spacing(np.float32(ref["cost"].max()))) / lam * emax)
    err = float(np.abs(inf["u"] - ref["U"]).max())
    assert err <= tol, f"{tag}: U max err {err:.3e} vs tol {tol:.3e} (scale {scale:.3e})"
    err_a = float(np.abs(got_act - ref["next_act"]).max())
    assert err_a <= tol, f"{tag}: action err {err_a:.3e} vs tol {tol:.3e}"


CASES = [
    # A, K, T
    (1, 100, 50),      # BASELINE config 1 shape
    (2, 3, 12),        # mppi-config-test.yaml shape
    (2, 257, 50),
    (2, 1000, 200),    # config 2 horizon
    (3, 300, 51),
    (3, 1024, 200),    # config 3 horizon
    (4, 130, 37),
    (1, 513, 203),     # ragged horizon for 4 steps per Philox block
    (2, 64, 1),        # single step
    (1, 300, 512),     # long horizons: the cost bar grows with T (cost_rtol)
    (1, 200, 1000),
    (3, 100, 512),
    (3, 60, 1000),     # beyond 64 KiB of LDS per block (gfx950 grants a workgroup up to 160 KiB)
    (2, 50, 2000),
    (1, 40, 4000),
    (4, 40, 1200),
]


@pytest.mark.parametrize("A,K,T", CASES)
def test_strict_kernel_cost_bit_exact_and_update(gpu, A, K, T):
    c = ol.make_case(A, K, T, seed=100 + A * 7 + T)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    with _model(gpu, A, K, T, c, strict=True) as m:
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf()
        assert m.geometry()["strict"]
    assert np.array_equal(inf["e"], c["E"]), "injected noise must round-trip through the tile layout"
    _check_solve(act, inf, ref, cost_exact=True, tag=f"strict A{A} K{K} T{T}")


@pytest.mark.parametrize("A,K,T", CASES)
@pytest.mark.parametrize("chunks", [0, 1, 2, 8, 64])
def test_fused_kernel_matches_oracle(gpu, A, K, T, chunks):
    from mppi_gpu_amd import MppiError
    c = ol.make_case(A, K, T, seed=200 + A * 7 + T)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    try:
        m = _model(gpu, A, K, T, c, chunks=chunks)
    except MppiError as ex:       # chunks below the register-resident minimum for this horizon
        assert "chunks must be" in str(ex)
        pytest.skip(str(ex))
    with m:
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf()
        geo = m.geometry()
    assert not geo["strict"]
    assert np.array_equal(inf["e"], c["E"])
    _check_solve(act, inf, ref, cost_exact=False, tag=f"fused A{A} K{K} T{T} {geo}")


PACKED_CASES = [
    # A, K, T, groups per lane (0 = the engine's choice), max_blocks
    (3, 1024, 200, 4, 0),      # config 3 horizon: 5 trajectories per wavefront
    (3, 20300, 200, 0, 0),     # the engine's own choice for a launch of many tiles
    (3, 1000, 200, 4, 3),      # ... on a persistent grid of 3 blocks (rescale path)
    (2, 1000, 200, 8, 0),      # config 2 horizon, 8 groups per lane: 5 trajectories per wavefront
    (2, 1000, 200, 5, 0),      # ... 5 groups per lane: 3 trajectories per wavefront
    (2, 3, 12, 5, 0),          # mppi-config-test.yaml shape: 6 groups per trajectory
    (2, 257, 50, 5, 0),
    (3, 300, 52, 4, 0),        # 13 groups per trajectory: 19 trajectories per wavefront
    (1, 700, 204, 4, 0),
    (1, 64, 16, 4, 0),         # one lane per trajectory
    (4, 130, 37, 10, 0),
    (4, 50, 400, 10, 1),       # 400 groups per trajectory: one trajectory per wavefront
    (3, 7, 512, 4, 0),         # 128 groups per trajectory: two per wavefront
    (1, 9, 1024, 4, 0),        # 256 groups per trajectory = 64 lanes x 4: exactly one wavefront each
    (2, 21, 10, 5, 0),         # 5 groups per trajectory = one lane each, 64 trajectories per wavefront
    (1, 300, 512, 4, 0),       # long horizons: the cost bar grows with T (cost_rtol)
    (1, 200, 1000, 4, 0),
    (3, 100, 512, 4, 0),
    (2, 100, 1000, 8, 0),
    (3, 60, 1000, 4, 0),       # 74 KiB of LDS per block
    (3, 33, 1023, 4, 2),       # ... ragged, on a persistent grid
    # ragged horizons: the last group of a trajectory holds fewer steps than a group (masked)
    (3, 3000, 50, 4, 0),       # the reference's shipped config/point_mass3d.yaml: 12 groups + 2 steps
    (1, 513, 203, 4, 0),       # 50 groups + 3 steps
    (3, 300, 51, 4, 0),        # 12 groups + 3 steps
    (1, 3000, 50, 4, 0),       # the shipped config/point_mass1d.yaml: 12 groups + 2 steps
    (2, 3000, 51, 8, 0),       # 25 groups + 1 step (act_dim 2: groups of 2 steps)
    (2, 700, 11, 5, 2),        # 5 groups + 1 step: one lane and a bit per trajectory
    (3, 2000, 201, 4, 3),      # 50 groups + 1 step on a persistent grid
    (3, 900, 199, 4, 0),       # 49 groups + 3 steps
]


@pytest.mark.parametrize("A,K,T,ngl,max_blocks", PACKED_CASES)
def test_packed_kernel_matches_oracle(gpu, A, K, T, ngl, max_blocks):
    """The packed rollout (whole trajectories end to end over the lanes, scaled-state dynamics)
    against the oracle on injected noise, in every shape class: boundaries in mid-lane and at lane
    ends, idle slots at the end of a wavefront, several tiles per block."""
    c = ol.make_case(A, K, T, seed=400 + A * 7 + T)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    with _model(gpu, A, K, T, c, max_blocks=max_blocks) as m:
        if ngl:
            m.set_packing(ngl)
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf()
        geo = m.geometry()
    assert geo["packed"] and (ngl == 0 or geo["groups_per_lane"] == ngl), geo
    assert np.array_equal(inf["e"], c["E"]), "injected noise must round-trip through the packed layout"
    _check_solve(act, inf, ref, cost_exact=False, tag=f"packed A{A} K{K} T{T} {geo}")
    cost, X = ol.rollout(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], want_X=True)
    assert np.array_equal(inf["x"], X), "state trace reads the packed layout"


def test_packed_kernel_random_shapes_against_oracle(gpu):
    """Seeded sweep of the packed kernel over random (A, groups per lane, T a whole number of
    groups, K, persistent grid, lambda, goal, weights incl. zeros): every boundary pattern between
    trajectories and lanes that the shapes above do not name."""
    rng = np.random.default_rng(20261005)
    NGS = {1: [4], 2: [5, 8], 3: [4], 4: [10]}
    SGS = {1: 4, 2: 2, 3: 4, 4: 1}
    done = 0
    for trial in range(30):
        A = int(rng.integers(1, 5))
        ngl = int(rng.choice(NGS[A]))
        # groups per trajectory (the engine's LDS budget bounds the horizon at T*A ~ 1000)
        ngt = int(rng.integers(ngl, min(64 * ngl, 140, 1000 // (SGS[A] * A)) + 1))
        T = ngt * SGS[A]
        K = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1025, 2500]))
        lam = float(rng.choice([0.5, 1.0, 2.0]))
        c = ol.make_case(A, K, T, seed=3000 + trial, u_scale=float(rng.choice([0.0, 0.05, 0.5])))
        c["goal"] = rng.standard_normal(2 * A).astype(np.float32)
        c["w"] = (np.abs(rng.standard_normal(2 * A) * 5) * (rng.random(2 * A) > 0.2)).astype(np.float32)
        ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam)
        with _model(gpu, A, K, T, c, max_blocks=int(rng.choice([0, 1, 3]))) as m:
            m.set_packing(ngl)
            m.set_params(lam)
            m.set_noise(c["E"])
            act = m.get_act()
            inf = m.get_inf(x=False)
            geo = m.geometry()
        assert geo["packed"]
        assert np.array_equal(inf["e"], c["E"])
        _check_solve(act, inf, ref, cost_exact=False, lam=lam,
                     tag=f"packed trial {trial} A{A} K{K} T{T} {geo}")
        done += 1
    assert done == 30


def test_packed_kernel_random_ragged_shapes_against_oracle(gpu):
    """The sweep above with horizons that are NOT whole groups: T = (groups - 1) * steps per group
    + 1 .. steps per group - 1 more steps (act_dim 1, 2, 3; act_dim 4 has one step per group)."""
    rng = np.random.default_rng(20261006)
    NGS = {1: [4], 2: [5, 8], 3: [4]}
    SGS = {1: 4, 2: 2, 3: 4}
    for trial in range(24):
        A = int(rng.integers(1, 4))
        ngl = int(rng.choice(NGS[A]))
        ngt = int(rng.integers(ngl, min(64 * ngl, 140, 1000 // (SGS[A] * A)) + 1))
        T = (ngt - 1) * SGS[A] + int(rng.integers(1, SGS[A]))
        K = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1025, 2500]))
        lam = float(rng.choice([0.5, 1.0, 2.0, 50.0]))
        c = ol.make_case(A, K, T, seed=3500 + trial, u_scale=float(rng.choice([0.0, 0.05, 0.5])))
        c["goal"] = rng.standard_normal(2 * A).astype(np.float32)
        c["w"] = (np.abs(rng.standard_normal(2 * A) * 5) * (rng.random(2 * A) > 0.2)).astype(np.float32)
        ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam)
        with _model(gpu, A, K, T, c, max_blocks=int(rng.choice([0, 1, 3]))) as m:
            m.set_packing(ngl)
            m.set_params(lam)
            m.set_noise(c["E"])
            act = m.get_act()
            inf = m.get_inf()
            geo = m.geometry()
        assert geo["packed"]
        assert np.array_equal(inf["e"], c["E"])
        _check_solve(act, inf, ref, cost_exact=False, lam=lam,
                     tag=f"packed ragged trial {trial} A{A} K{K} T{T} {geo}")
        cost, X = ol.rollout(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam, want_X=True)
        assert np.array_equal(inf["x"], X)


@pytest.mark.parametrize("A,K,T,ngl", [(3, 3000, 50, 4), (1, 3000, 50, 4), (2, 2000, 51, 8)])
def test_packed_ragged_horizon_in_sampling_mode(gpu, A, K, T, ngl):
    """Sampling mode on a ragged horizon: the packed kernel draws the noise of the strict kernel bit
    for bit (the stream is defined on (seed, solve, sample, t*A + a), not on the layout), zeroes
    what lies past T, and the oracle re-run on that noise reproduces the solve; riding and flushed
    combines give equal bits there too."""
    c = ol.make_case(A, 1, T, seed=19, u_scale=0.05)
    res = {}
    for kind in ("strict", "packed"):
        with _model(gpu, A, K, T, c, strict=(kind == "strict")) as m:
            if kind == "packed":
                m.set_packing(ngl)
            m.set_seed(77)
            m.set_params(30.0)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            m.get_act()
            U1 = m.get_u()
            act = m.get_act()                       # second solve: solve index 1 in the counters
            res[kind] = (act, m.get_inf(x=False), U1, m.geometry())
    assert res["packed"][3]["packed"]
    assert np.array_equal(res["packed"][1]["e"], res["strict"][1]["e"])
    act, inf, U1, geo = res["packed"]
    ref = ol.solve(c["x0"], U1, inf["e"], c["goal"], c["w"], c["dt"], lam=30.0)
    _check_solve(act, inf, ref, cost_exact=False, lam=30.0, tag=f"packed ragged sampled A{A} K{K} T{T} {geo}")
    chains = []
    for blocking in (False, True):
        with _model(gpu, A, K, T, c) as m:
            m.set_packing(ngl)
            m.set_seed(78)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            for _ in range(6):
                m.get_act() if blocking else m.solve_async()
            chains.append((m.sync_act(), m.get_u(), m.launch_counts()))
    assert np.array_equal(chains[0][0], chains[1][0]) and np.array_equal(chains[0][1], chains[1][1])
    assert chains[0][2]["riding"] == 5, chains[0][2]


def test_packed_kernel_general_goal_and_zero_weights(gpu):
    """Velocity goals != 0 (the scaled position drifts by a constant per step), zero weights on
    single axes (scale 2^-60 instead of sqrt(w)), lambda and inv_s != 1."""
    A, K, T = 3, 900, 48
    c = ol.make_case(A, K, T, seed=71, u_scale=0.2)
    c["goal"] = np.array([1, .5, .75, .3, -.2, .1], np.float32)
    c["w"] = np.array([2, 0, 1, 5, 3, 0], np.float32)
    lam, inv_s = 1.7, np.array([2.0, 0.5, 1.25], np.float32)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam, inv_s=inv_s)
    with _model(gpu, A, K, T, c) as m:
        m.set_packing(4)
        m.set_params(lam, inv_s=inv_s)
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf(x=False, e=False)
        assert m.geometry()["packed"]
    _check_solve(act, inf, ref, cost_exact=False, tag="packed goal/zero-w", lam=lam)
    # the row-aligned kernel runs the same scaled dynamics (signs of the weights kept aside)
    for chunks in (0, 4, 64):
        with _model(gpu, A, K, T, c) as m:
            m.set_packing(-1)
            m.set_tuning(chunks=chunks)
            m.set_params(lam, inv_s=inv_s)
            m.set_noise(c["E"])
            act = m.get_act()
            inf = m.get_inf(x=False, e=False)
            assert not m.geometry()["packed"]
        _check_solve(act, inf, ref, cost_exact=False, tag=f"row-aligned goal/zero-w chunks={chunks}", lam=lam)
    # negative weights are not a cost the scaled form can carry: the row-aligned kernel takes over
    c["w"] = np.array([2, -1, 1, 5, 3, 0], np.float32)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    with _model(gpu, A, K, T, c) as m:
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf(x=False, e=False)
        assert not m.geometry()["packed"]
    _check_solve(act, inf, ref, cost_exact=False, tag="negative w falls back")


def test_packing_applies_to_ragged_horizons_and_is_refused_where_it_cannot(gpu):
    from mppi_gpu_amd import MppiError
    # T = 50 at act_dim 3 (the reference's shipped config/point_mass3d.yaml) is 12 groups of 4 steps
    # and one of 2: packed since round 3 (the steps past T are masked), 19 trajectories per wave
    c = ol.make_case(3, 100, 50, seed=3)
    with _model(gpu, 3, 100, 50, c) as m:
        m.set_packing(4)
        geo = m.geometry()
        assert geo["packed"] and geo["trajectories_per_wave"] == 19, geo
        m.set_packing(-1)
        assert not m.geometry()["packed"]
    c = ol.make_case(3, 100, 200, seed=3)
    with _model(gpu, 3, 100, 200, c) as m:
        assert not m.geometry()["packed"]                # a launch the chip holds at once: latency, not throughput
        m.set_packing(4)
        assert m.geometry()["packed"]
        with pytest.raises(MppiError):
            m.set_packing(7)                             # not an instantiated size
        assert m.geometry()["packed"]                    # (the refused call changed nothing)
        m.set_packing(-1)
        assert not m.geometry()["packed"]
    c = ol.make_case(3, 1, 10, seed=3)
    with _model(gpu, 3, 100, 10, c) as m:                # 3 groups per trajectory < 4 groups per lane
        with pytest.raises(MppiError):
            m.set_packing(4)
    c = ol.make_case(3, 1, 200, seed=3)
    with _model(gpu, 3, 24000, 200, c) as m:             # many tiles per block: packed by itself
        assert m.geometry()["packed"] and m.geometry()["trajectories_per_wave"] == 5
    c = ol.make_case(3, 1, 50, seed=3)
    with _model(gpu, 3, 120000, 50, c) as m:             # ... and so is a ragged horizon
        assert m.geometry()["packed"] and m.geometry()["trajectories_per_wave"] == 19


@pytest.mark.parametrize("A,K,T,packing", [(3, 24000, 200, 0), (3, 1500, 200, -1), (2, 3000, 200, 8),
                                            (2, 10000, 200, 0), (1, 700, 33, 0), (3, 3000, 50, 4)])
def test_noise_not_materialised_is_regenerated_bit_for_bit(gpu, A, K, T, packing):
    """mppi_set_noise_store(0): the rollout stores no noise (94 % fewer HBM bytes); the solve is
    the same bits, and the noise get_inf hands out -- regenerated from the Philox counters -- equals
    what the storing rollout wrote, bit for bit; so does the state trace computed from it."""
    c = ol.make_case(A, 1, T, seed=17, u_scale=0.05)
    out = []
    for store in (True, False):
        with _model(gpu, A, K, T, c) as m:
            if packing:
                m.set_packing(packing)
            m.set_noise_store(store)
            m.set_seed(99)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            m.get_act()
            act = m.get_act()                       # second solve: solve index 1 in the counters
            inf = m.get_inf(x=(K <= 3000))
            out.append((act, inf))
    (a1, i1), (a0, i0) = out
    assert np.array_equal(a1, a0) and np.array_equal(i1["u"], i0["u"])
    assert np.array_equal(i1["cost"], i0["cost"])
    assert np.array_equal(i1["e"], i0["e"]), "regenerated noise differs from stored noise"
    if "x" in i1:
        assert np.array_equal(i1["x"], i0["x"])


def test_states_trace_matches_oracle(gpu):
    A, K, T = 3, 200, 50
    c = ol.make_case(A, K, T, seed=5)
    cost, X = ol.rollout(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], want_X=True)
    with _model(gpu, A, K, T, c, chunks=4) as m:
        m.set_noise(c["E"])
        m.get_act()
        Xg, Eg = m.memcpy_get_data()
    assert np.array_equal(Eg, c["E"])
    assert np.array_equal(Xg, X), "state trace is sequential: bit-exact with the oracle"


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("solve_")))
def test_committed_golden_fixtures(gpu, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    A, K, T = int(g["A"]), int(g["K"]), int(g["T"])
    case = dict(x0=g["x0"], U=g["U"], goal=g["goal"], w=g["w"], dt=g["dt"])
    ref = dict(cost=g["cost"], beta=np.float32(g["beta"]), nabla=np.float32(g["nabla"]),
               weights=g["weights"], U=g["U_next"], next_act=g["next_act"])
    for strict in (True, False):
        with _model(gpu, A, K, T, case, strict=strict) as m:
            m.set_noise(g["E"])
            act = m.get_act()
            inf = m.get_inf()
        _check_solve(act, inf, ref, cost_exact=strict, tag=f"{name} strict={strict}")


def test_sampled_noise_matches_rocrand_host_stream(gpu):
    """Sampling mode: the device noise equals the stream stated over rocRAND's public host API
    (same Philox words; Box-Muller differs only by GPU-vs-libm transcendental accuracy),
    independent of the kernel geometry, and advances from solve to solve."""
    A, K, T = 3, 500, 200
    c = ol.make_case(A, K, T, seed=9)
    sig = np.array([0.025, 0.05, 0.0125], np.float32)
    Es = {}
    for chunks, strict in ((0, False), (8, False), (32, False), (0, True)):
        with _model(gpu, A, K, T, c, chunks=chunks, strict=strict) as m:
            m.set_params(1.0, sigma=sig)
            m.set_seed(1234)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            m.get_act()
            e0 = m.get_inf(x=False, u=False, cost=False, beta=False, nabla=False, weight=False)["e"]
            m.get_act()
            e1 = m.get_inf(x=False, u=False, cost=False, beta=False, nabla=False, weight=False)["e"]
        Es[(chunks, strict)] = (e0, e1)
    base0, base1 = Es[(0, False)]
    for key, (e0, e1) in Es.items():
        assert np.array_equal(e0, base0) and np.array_equal(e1, base1), f"noise depends on geometry {key}"
    h0 = ol.noise(1234, 0, 0, K, T, A, sig)
    h1 = ol.noise(1234, 1, 0, K, T, A, sig)
    # |z| <= 6.7; fast device sin/cos are accurate to ~1e-6 absolute on the unit circle
    np.testing.assert_allclose(base0, h0, rtol=0, atol=float(sig.max()) * 2e-5)
    np.testing.assert_allclose(base1, h1, rtol=0, atol=float(sig.max()) * 2e-5)
    assert not np.array_equal(base0, base1)
    z = base0 / sig
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01


def test_closed_loop_sequence_matches_oracle_chain(gpu):
    """Five consecutive solves in sampling mode with set_x in between: every solve is re-run
    by the oracle on the noise the device drew (read back with get_inf)."""
    A, K, T = 2, 2000, 50
    c = ol.make_case(A, K, T, seed=21, u_scale=0.0)
    x = c["x0"].copy()
    U = c["U"].copy()
    with _model(gpu, A, K, T, c) as m:
        for it in range(5):
            assert np.array_equal(m.get_u(), U) or it > 0
            U_before = m.get_u()
            act = m.get_act()
            inf = m.get_inf(x=False)
            ref = ol.solve(x, U_before, inf["e"], c["goal"], c["w"], c["dt"])
            _check_solve(act, inf, ref, cost_exact=False, tag=f"step {it}")
            # a stand-in plant: apply the action with the model itself
            xn = x.copy()
            xn[:A] = x[:A] + c["dt"] * x[A:] + np.float32(0.005) * act
            xn[A:] = x[A:] + c["dt"] * act
            x = xn.astype(np.float32)
            m.set_x(x)
            assert np.array_equal(m.get_x(), x)


@pytest.mark.gpu
@pytest.mark.parametrize("A,K,T", [(2, 10000, 200), (3, 3000, 50), (1, 700, 33), (3, 40000, 120)])
def test_deferred_combine_rides_and_flushes_with_equal_bits(gpu, A, K, T):
    """Mode 0 (default): back-to-back mppi_solve_async calls carry the previous solve's combine in
    the next rollout launch (rollout blocks poll the tagged 8-byte words {value, tag} the applying
    combine blocks publish for the new controls: no fence, no counter); get_act flushes it
    stand-alone.  Both are the same device function: a closed-loop chain of
    get_act calls and a chain of asynchronous solves must end in identical bits.  Against the
    eager mode (1024-thread combine, another summation order) the controls agree to rounding."""
    c = ol.make_case(A, K, T, seed=124)
    n = 6

    def chain(mode, blocking):
        with _model(gpu, A, K, T, c) as m:
            m.set_pipeline(mode)
            m.set_seed(78)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            acts = []
            for it in range(n):
                if blocking:
                    acts.append(m.get_act())
                else:
                    m.solve_async()
            if not blocking:
                acts.append(m.sync_act())
            inf = m.get_inf(x=False)
            return acts[-1], inf["u"], inf["cost"], inf["e"], inf["beta"], inf["nabla"]

    ride = chain(0, False)
    flush = chain(0, True)
    eager = chain(1, False)
    for a, b in zip(ride, flush):
        assert np.array_equal(a, b), "riding and flushed combine must give equal bits"
    # another summation order: one solve's controls move by a few ulp(cost)/lambda relative weight
    # change (module docstring); the chain of n solves feeds that back n times
    scale = max(float(np.abs(eager[1]).max()), SIGMA)
    cmax = float(np.abs(eager[2]).max())
    bar = n * max(1e-5 * scale, 4 * float(np.spacing(np.float32(cmax))) * 4.5 * SIGMA)
    assert np.abs(ride[1] - eager[1]).max() <= bar
    assert np.abs(ride[0] - eager[0]).max() <= bar
    assert abs(float(ride[4]) - float(eager[4])) <= 1e-5 * abs(float(eager[4]))


@pytest.mark.gpu
@pytest.mark.parametrize("A,K,T,ngl", [(3, 3000, 200, 4), (3, 24000, 200, 0), (2, 2500, 200, 8), (1, 900, 48, 4)])
def test_packed_kernel_rides_and_flushes_with_equal_bits(gpu, A, K, T, ngl):
    """The packed rollout in pipeline mode 0: back-to-back solves carry the previous solve's combine
    at the front of the packed grid (k_rollout_packed_ride); a chain of blocking get_act calls
    flushes every combine on its own.  Same device functions: identical bits, in sampling mode."""
    c = ol.make_case(A, 1, T, seed=131, u_scale=0.03)
    n = 7

    def chain(blocking):
        with _model(gpu, A, K, T, c) as m:
            if ngl:
                m.set_packing(ngl)
            m.set_seed(8)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            geo = m.geometry()
            assert geo["packed"], geo          # (a combine rides in a packed launch of any length)
            for _ in range(n):
                m.get_act() if blocking else m.solve_async()
            act = m.sync_act()
            inf = m.get_inf(x=False, e=False)
            return act, inf["u"], inf["cost"], inf["beta"], inf["nabla"]

    ride, flush = chain(False), chain(True)
    for a, b in zip(ride, flush):
        assert np.array_equal(a, b), "riding and flushed combine must give equal bits"
    assert np.all(np.isfinite(ride[1]))


@pytest.mark.gpu
def test_a_combine_rides_only_where_the_whole_launch_is_resident(gpu):
    """mppi_get_launch_counts: back-to-back solves of a short launch whose blocks all fit the chip
    (BASELINE config 2: 625 rollout + 25 combine blocks) are one launch each; a launch of the same
    tile length with more blocks than the chip holds at once launches its combines on their own --
    the rollout blocks of a riding launch WAIT for its combine blocks, so none may be left
    without a slot.  Equal results either way are the business of the equal-bits tests."""
    n = 12
    for A, K, T, expect_ride in [(2, 10000, 200, True), (2, 40000, 200, False)]:
        c = ol.make_case(A, 1, T, seed=5, u_scale=0.0)
        with _model(gpu, A, K, T, c) as m:
            m.set_packing(-1)
            geo = m.geometry()
            for _ in range(n):
                m.solve_async()
            m.sync_act()
            cnt = m.launch_counts()
        assert cnt["rollout"] == n and cnt["resident_ride"] > 0, cnt
        assert geo["tile_groups"] <= 2 * geo["grid"], geo            # short launches both
        if expect_ride:
            assert geo["grid"] + 25 <= cnt["resident_ride"], (geo, cnt)
            assert cnt["riding"] == n - 1 and cnt["combine"] == 1, cnt
        else:
            assert geo["grid"] > cnt["resident_ride"], (geo, cnt)
            assert cnt["riding"] == 0 and cnt["combine"] == n, cnt


def test_noise_mode_change_replans_the_riding_launch(gpu):
    """The sampling and the injected-noise instantiations of the riding kernel use different numbers
    of registers, so the chip holds different numbers of their blocks: after set_noise(E) ->
    solve -> set_noise(None) the co-residency test of a riding combine must use the occupancy of the
    SAMPLING variant again (it kept the injected variant's)."""
    A, K, T = 2, 10000, 200
    c = ol.make_case(A, K, T, seed=6)
    with _model(gpu, A, K, T, c) as m:
        fresh = m.launch_counts()["resident_ride"]
        geo_fresh = m.geometry()
        m.set_noise(c["E"])
        m.get_act()
        injected = m.launch_counts()["resident_ride"]
        m.set_noise(None)
        for _ in range(5):
            m.solve_async()
        m.sync_act()
        cnt = m.launch_counts()
        assert cnt["resident_ride"] == fresh > 0, (cnt, fresh, injected)
        assert m.geometry()["grid"] == geo_fresh["grid"]
        assert cnt["riding"] == 4, cnt


@pytest.mark.parametrize("A,K,T,packing", [(2, 10000, 200, 0), (3, 3000, 50, 0), (3, 3000, 50, 4),
                                            (3, 30011, 200, 0), (1, 700, 33, 0)])
def test_noise_prefetch_gives_the_bits_of_in_kernel_sampling(gpu, A, K, T, packing):
    """mppi_set_noise_prefetch: the combine launch of a blocking get_act carries low-priority
    blocks that draw the NEXT solve's noise; the next rollout loads it (injected-noise
    instantiation) instead of drawing it.  Same counters, same device functions: the chain must end in the bits of the
    chain that samples inside the rollout, with set_x / set_params / get_inf / solve_async in
    between, and the noise get_inf hands out must be the stream's."""
    c = ol.make_case(A, 1, T, seed=151, u_scale=0.03)
    outs = {}
    for mode in (0, 2):
        with _model(gpu, A, K, T, c) as m:
            if packing:
                m.set_packing(packing)
            m.set_noise_prefetch(mode)
            m.set_seed(21)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            log = []
            for it in range(4):
                log.append(m.get_act())
            m.set_x((c["x0"] * 0.8).astype(np.float32))
            log.append(m.get_act())
            e_mid = m.get_inf(x=False, u=False, cost=False, beta=False, nabla=False, weight=False)["e"]
            m.set_params(3.0)                     # lambda only: the prefetched noise stays valid
            log.append(m.get_act())
            m.solve_async(); m.solve_async()      # (the first of them may load a prefetch)
            log.append(m.sync_act())
            m.set_params(3.0, sigma=[0.05] * A)   # another sigma: a prefetch in flight is stale
            log.append(m.get_act())
            log.append(m.get_act())
            inf = m.get_inf(x=False)
            cnt = m.prefetch_counts()
        outs[mode] = (log, e_mid, inf, cnt)
    (l0, e0, i0, c0), (l2, e2, i2, c2) = outs[0], outs[2]
    assert c0 == {"launched": 0, "used": 0}
    assert c2["used"] >= 5 and c2["launched"] >= c2["used"], c2
    for a, b in zip(l0, l2):
        assert np.array_equal(a, b), "prefetched noise must give the bits of in-kernel sampling"
    assert np.array_equal(e0, e2) and np.array_equal(i0["e"], i2["e"])
    assert np.array_equal(i0["u"], i2["u"]) and np.array_equal(i0["cost"], i2["cost"])
    h = ol.noise(21, 4, 0, min(K, 64), T, A, [SIGMA] * A)          # solve index 4 = the fifth solve
    np.testing.assert_allclose(e2[:64], h, rtol=0, atol=SIGMA * 2e-5)


def test_noise_prefetch_auto_mode_follows_the_launch_length_and_the_think_time(gpu):
    """mode 1: a launch of one tile per block always prefetches; a long (VALU-bound) launch only
    when the host's think time between two blocking calls hides the whole draw."""
    import time
    for A, K, T in ((3, 3000, 50), (2, 10000, 200)):     # the shipped 3-D config, C2
        c = ol.make_case(A, 1, T, seed=5, u_scale=0.0)
        with _model(gpu, A, K, T, c) as m:
            for _ in range(6):
                m.get_act()
            assert m.prefetch_counts()["used"] >= 4
    c = ol.make_case(3, 1, 200, seed=5, u_scale=0.0)
    with _model(gpu, 3, 60000, 200, c) as m:             # packed, 6 tiles per block
        for _ in range(6):
            m.get_act()                                  # back to back: no think time
        assert m.prefetch_counts() == {"launched": 0, "used": 0}
        for _ in range(6):
            m.get_act()
            time.sleep(0.003)                            # a 300 Hz loop
        cnt = m.prefetch_counts()
        assert cnt["used"] >= 3, cnt


def test_deferred_combine_interleaved_with_everything_else(gpu):
    """A pending combine must be flushed by every call that reads or changes what it touches:
    set_x between asynchronous solves (rides on), get_u / get_inf / set_params / set_tuning /
    set_data in the middle (flush), persistent grids with several tiles per block."""
    A, K, T = 2, 9000, 60
    c = ol.make_case(A, K, T, seed=125)

    def script(mode):
        with _model(gpu, A, K, T, c, max_blocks=24) as m:
            m.set_pipeline(mode)
            m.set_seed(5)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            log = []
            m.solve_async(); m.solve_async()
            m.set_x((c["x0"] * 0.9).astype(np.float32))
            m.solve_async()
            log.append(m.get_u())                       # flush
            m.solve_async()
            m.set_params(2.0, None, None)               # flush, new lambda for the next solve
            m.solve_async(); m.solve_async()
            log.append(m.sync_act())
            m.set_tuning(chunks=8, strict=False, max_blocks=0)   # flush, new geometry
            m.solve_async(); m.solve_async()
            log.append(m.get_inf(x=False)["u"])
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            m.solve_async()
            log.append(m.get_act())
            return log

    a, b = script(0), script(1)
    for x, y in zip(a, b):
        scale = max(float(np.abs(y).max()), SIGMA)
        assert np.abs(x - y).max() <= 1e-4 * scale      # chains of up to 9 solves, see above


@pytest.mark.gpu
def test_long_riding_chain_equals_flushed_chain(gpu):
    """3 000 solves at the bench shape, enqueued back to back (every combine rides and hands the
    controls over through tagged words while the next rollout is already running) against the
    same chain with every combine launched on its own: a single stale or torn hand-over would
    change the next solve and, through the chain, the final bits.  (tools/soak.py runs 1e5.)"""
    A, K, T, n = 2, 10000, 200, 3000
    c = ol.make_case(A, 1, T, seed=0, u_scale=0.0)
    res = []
    for blocking in (False, True):
        with _model(gpu, A, K, T, c) as m:
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            for _ in range(n):
                if blocking:
                    m.get_act()
                else:
                    m.solve_async()
            res.append((m.sync_act().copy(), m.get_u().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.all(np.isfinite(res[0][1]))


@pytest.mark.gpu
def test_solves_that_change_stream_stay_ordered(gpu):
    """A held-back combine belongs to the stream of its solve; a next solve on ANOTHER stream must
    still start from its controls (flush + wait on the old stream), so hopping between two caller
    streams gives the bits of staying on one."""
    import torch
    A, K, T = 2, 6000, 80
    c = ol.make_case(A, K, T, seed=126)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(streams):
        with _model(gpu, A, K, T, c) as m:
            m.set_seed(3)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            for st in streams:
                m.solve_async(st.cuda_stream)
            act = m.sync_act()
            torch.cuda.synchronize()
            return act, m.get_u()

    one = run([s1] * 6)
    hop = run([s1, s1, s2, s1, s2, s2])
    for a, b in zip(one, hop):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_random_call_sequences_do_not_depend_on_how_solves_are_enqueued(gpu, seed):
    """Random shapes (tiny horizons and batches included) and random call sequences: the same
    sequence with a synchronisation forced after every solve (every combine flushed on its own)
    must give the bits of the free-running sequence (combines riding wherever they can), and the
    injected-noise mode must go through the same machinery."""
    rng = np.random.default_rng(1000 + seed)
    A = int(rng.integers(1, 5))
    T = int(rng.choice([1, 2, 3, 5, 17, 50, 200]))
    K = int(rng.choice([1, 3, 64, 257, 2000, 9000]))
    c = ol.make_case(A, K, T, seed=200 + seed, u_scale=0.05)
    inject = bool(rng.integers(0, 2))
    ops = [str(rng.choice(["solve", "solve", "solve", "set_x", "get_u", "params", "get_act"]))
           for _ in range(14)]
    xs = [(c["x0"] * np.float32(rng.uniform(0.5, 1.5))).astype(np.float32) for _ in ops]
    lams = [float(rng.uniform(0.5, 3.0)) for _ in ops]

    def run(force_sync):
        with _model(gpu, A, K, T, c) as m:
            m.set_seed(40 + seed)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            if inject:
                m.set_noise(c["E"])
            log = []
            for i, op in enumerate(ops):
                if op == "solve":
                    m.solve_async()
                    if force_sync:
                        m.sync_act()
                elif op == "set_x":
                    m.set_x(xs[i])
                elif op == "get_u":
                    log.append(m.get_u())
                elif op == "params":
                    m.set_params(lams[i], None, None)
                else:
                    log.append(m.get_act())
            log.append(m.sync_act() if any(o in ("solve", "get_act") for o in ops) else np.zeros(A))
            log.append(m.get_u())
            return log

    free, forced = run(False), run(True)
    assert len(free) == len(forced)
    for a, b in zip(free, forced):
        assert np.array_equal(a, b), (A, K, T, inject, ops)
    assert all(np.all(np.isfinite(a)) for a in free)


def test_persistent_grid_and_rescale_path(gpu):
    """max_blocks << tiles forces every block through several tile groups, i.e. through the
    running-minimum rescale branch; costs are spread so that block minima differ a lot."""
    A, K, T = 2, 20000, 50
    c = ol.make_case(A, K, T, seed=33, u_scale=0.3)
    # make sample costs vary strongly: scale the noise of later samples up
    ramp = np.linspace(1.0, 12.0, K, dtype=np.float32)[::-1].copy()
    c["E"] = (c["E"] * ramp[:, None, None]).astype(np.float32)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    assert ref["cost"].max() - ref["cost"].min() > 30
    for max_blocks in (1, 3, 64):
        with _model(gpu, A, K, T, c, chunks=4, max_blocks=max_blocks) as m:
            m.set_noise(c["E"])
            act = m.get_act()
            inf = m.get_inf(x=False, e=False)
            assert m.geometry()["grid"] == max_blocks
        _check_solve(act, inf, ref, cost_exact=False, tag=f"max_blocks={max_blocks}",
                     emax=float(np.abs(c["E"]).max()))


def test_extreme_cost_spread_underflow(gpu):
    """exp underflow: one sample far better than all others -> weights ~ one-hot, no NaN."""
    A, K, T = 2, 512, 20
    c = ol.make_case(A, K, T, seed=44)
    c["E"] = (c["E"] * 40).astype(np.float32)
    c["E"][7] = 0
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    with _model(gpu, A, K, T, c) as m:
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf(x=False, e=False)
    assert np.isfinite(inf["u"]).all()
    _check_solve(act, inf, ref, cost_exact=False, tag="underflow", emax=float(np.abs(c["E"]).max()))


def test_all_equal_costs_degenerate(gpu):
    """sigma = 0: all costs equal, weights uniform, dU = 0 -> U is shifted exactly."""
    A, K, T = 3, 777, 30
    c = ol.make_case(A, K, T, seed=55)
    with _model(gpu, A, K, T, c) as m:
        m.set_params(1.0, sigma=[0, 0, 0])
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        act = m.get_act()
        inf = m.get_inf(x=False)
    assert np.all(inf["e"] == 0)
    assert np.all(inf["cost"] == inf["cost"][0])
    np.testing.assert_allclose(inf["weight"], 1.0 / K, rtol=1e-6)
    assert np.array_equal(act, c["U"][0])
    assert np.array_equal(inf["u"][:-1], c["U"][1:]) and np.array_equal(inf["u"][-1], c["U"][-1])


def test_params_lambda_inv_s(gpu):
    A, K, T = 2, 400, 25
    c = ol.make_case(A, K, T, seed=66)
    lam, inv_s = 1.5, np.array([2.0, 0.5], np.float32)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam, inv_s=inv_s)
    for strict in (True, False):
        with _model(gpu, A, K, T, c, strict=strict) as m:
            m.set_params(lam, inv_s=inv_s)
            m.set_noise(c["E"])
            act = m.get_act()
            inf = m.get_inf(x=False, e=False)
        _check_solve(act, inf, ref, cost_exact=strict, tag=f"params strict={strict}", lam=lam)


@pytest.mark.parametrize("A,K,T,covered", [(3, 3000, 20, 2048), (1, 3000, 24, 768),
                                             (1, 1000, 50, 256), (1, 200, 16, 100), (2, 900, 10, 900)])
def test_ref_compat_reproduces_the_reference_update_coverage(gpu, A, K, T, covered):
    """with ref_compat the update sums only the samples the reference's
    update_act reaches (act_dim 3: the first 512*(K/768+1); act_dim 1: k even and (k/512) even;
    act_dim 2: all), while beta and nabla still use all K.  The expected controls come from the
    literal emulation of the reference's launch structure (oracle_lib.ref_update_emulated)."""
    c = ol.make_case(A, K, T, seed=77)
    full = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    assert int(ol.ref_update_mask(K, A).sum()) == covered
    U_upd = ol.ref_update_emulated(c["U"], full["weights"], c["E"])
    with _model(gpu, A, K, T, c) as m:
        m.set_ref_compat(True)
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf(x=False, e=False)
    np.testing.assert_allclose(inf["nabla"], full["nabla"], rtol=1e-4)
    scale = max(float(np.abs(U_upd).max()), SIGMA)
    assert np.abs(act - U_upd[0]).max() <= 1e-5 * scale
    assert np.abs(inf["u"][:-1] - U_upd[1:]).max() <= 1e-5 * scale
    # and it does differ from the correct update wherever the reference drops samples
    if covered < K:
        assert np.abs(inf["u"] - full["U"]).max() > 1e-6


def test_action_limit_is_opt_in(gpu):
    """max-a (reference src/main.cu:524,566-568: parsed, never applied): off by default; switched
    on, every updated control is clamped to +-max_a[axis] and the returned action with it."""
    A, K, T = 2, 1500, 40
    c = ol.make_case(A, K, T, seed=91, u_scale=0.3)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    lim = np.array([0.2, 0.05], np.float32)
    assert (np.abs(ref["U"]) > lim).any()
    exp_U = np.clip(ref["U"], -lim, lim)
    exp_act = np.clip(ref["next_act"], -lim, lim)
    for strict in (False, True):
        with _model(gpu, A, K, T, c, strict=strict) as m:
            m.set_noise(c["E"])
            m.set_action_limit(lim)
            act = m.get_act()
            U = m.get_u()
            inside = np.abs(exp_U) < lim * 0.999
            assert np.abs(U).max(axis=0)[0] <= lim[0] and np.abs(U).max(axis=0)[1] <= lim[1]
            assert np.abs(U - exp_U)[inside].max() <= 1e-5 * max(float(np.abs(ref["U"]).max()), SIGMA)
            assert np.abs(act - exp_act).max() <= 1e-5
            # off again: the plain update
            m.set_action_limit(None)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            m.set_noise(c["E"])
            m.get_act()
            assert np.abs(m.get_u() - ref["U"]).max() <= 1e-5 * max(float(np.abs(ref["U"]).max()), SIGMA)


def test_device_time_out_is_sticky_until_set_data(gpu):
    """A rank whose peer never shows up: the block that gave up publishes nothing, and every later
    solve / read-out reports MPPI_ESTATE until mppi_set_data starts over."""
    import torch
    from mppi_gpu_amd import PointMassModel, MppiError
    A, K, T = 2, 2000, 50
    c = ol.make_case(A, K, T, seed=63)
    with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        m.get_act()
        U_before = m.get_u()
        _, mine = m.xchg_open(0, 2)
        W = ((T * A + 2 + 15) // 16) * 16
        silent = torch.zeros(2 * 2 * W, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        m.xchg_connect(same_process=[mine, silent.data_ptr()])
        m.xchg_set_timeout(0.2)
        m.solve_exchange_async()
        for call in (m.sync_act, m.get_u, m.get_act, m.solve_async, m.sync_act):
            with pytest.raises(MppiError) as ei:
                call()
            assert ei.value.code == -4                      # MPPI_ESTATE, every time
        assert "switched to pipeline mode 1" in str(ei.value)
        assert m.pipeline() == {"mode": 1, "degraded": True}
        m.memcpy_set_data(c["x0"], U_before, c["goal"], c["w"])      # starts over
        assert np.array_equal(m.get_u(), U_before)
        assert np.all(np.isfinite(m.get_act()))
        # ... and stays degraded: solves enqueued back to back no longer ride (no block of a launch
        # waits for another block of it) until the caller asks for mode 0 again
        before = m.launch_counts()
        for _ in range(4):
            m.solve_async()
        m.sync_act()
        after = m.launch_counts()
        assert after["riding"] == before["riding"] and after["rollout"] == before["rollout"] + 4
        m.set_pipeline(0)
        assert m.pipeline() == {"mode": 0, "degraded": False}
        for _ in range(4):
            m.solve_async()
        m.sync_act()
        assert m.launch_counts()["riding"] == after["riding"] + 3


def test_sharded_engines_equal_single_engine(gpu):
    """Two shard engines on one GPU (k offsets 0 and K/2) + gather + finish == one engine."""
    import torch
    A, K, T = 3, 4000, 200
    c = ol.make_case(A, K, T, seed=88)
    from mppi_gpu_amd import PointMassModel
    with _model(gpu, A, K, T, c) as m:
        m.set_seed(5)
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        act1 = m.get_act()
        U1 = m.get_u()
        E1 = m.get_inf(x=False, u=False, cost=False, beta=False, nabla=False, weight=False)["e"]
        act1b = m.get_act()
    shards = []
    half = K // 2 + 100                      # uneven split
    for off, n in ((0, half), (half, K - half)):
        s = PointMassModel(n, T, float(c["dt"]), 2 * A, A, k_offset=off)
        s.set_seed(5)
        s.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        shards.append(s)
    L = shards[0].partial_len()
    assert L == T * A + 2
    for it in range(2):
        gathered = torch.zeros(2, L, device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        for i, s in enumerate(shards):
            s.solve_local_async(gathered[i].data_ptr())
            s.sync_act()
        acts = []
        for s in shards:
            s.solve_finish_async(gathered.data_ptr(), 2)
            acts.append(s.sync_act())
        assert np.array_equal(acts[0], acts[1]), "ranks must agree bit for bit"
        if it == 0:
            E = np.concatenate([s.get_inf(x=False, u=False, cost=False, beta=False, nabla=False,
                                          weight=False)["e"] for s in shards])
            assert np.array_equal(E, E1), "noise must not depend on the sharding"
            scale = max(float(np.abs(U1).max()), SIGMA)
            assert np.abs(acts[0] - act1).max() <= 2e-6 * scale
            assert np.abs(shards[0].get_u() - U1).max() <= 2e-6 * scale
            assert np.array_equal(shards[0].get_u(), shards[1].get_u())
        else:
            assert np.abs(acts[0] - act1b).max() <= 1e-5 * SIGMA * 4
    for s in shards:
        s.close()


@pytest.mark.gpu
def test_direct_exchange_in_process_equals_gathered_finish(gpu):
    """Two shard engines in ONE process, each on its own stream, exchanging through their
    inboxes (raw pointers instead of ipc handles) == local combine + gather + finish, bit for bit.
    The two combine kernels wait for one another, so they must be resident together: streams of
    different priority; the exchange time-out turns a scheduling surprise into a skip."""
    import torch
    from mppi_gpu_amd import PointMassModel
    A, K, T, G = 2, 6000, 120, 2
    c = ol.make_case(A, K, T, seed=61)
    bounds = [0, 2100, K]
    # two caller streams of DIFFERENT priority: HIP gives those separate hardware queues, so the
    # two engines' kernels can run together (streams of equal priority may share a queue)
    streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]

    def make():
        out = []
        for g in range(G):
            s = PointMassModel(bounds[g + 1] - bounds[g], T, float(c["dt"]), 2 * A, A,
                               k_offset=bounds[g])
            s.set_seed(9)
            s.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            out.append(s)
        return out

    ref = make()
    L = ref[0].partial_len()
    ref_acts = []
    for it in range(6):
        gathered = torch.zeros(G, L, device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        for g, s in enumerate(ref):
            s.solve_local_async(gathered[g].data_ptr())
            s.sync_act()
        for s in ref:
            s.solve_finish_async(gathered.data_ptr(), G)
        ref_acts.append([s.sync_act() for s in ref][0])
    U_ref = ref[0].get_u()
    for s in ref:
        s.close()

    eng = make()
    ptrs = [s.xchg_open(g, G)[1] for g, s in enumerate(eng)]
    for s in eng:
        s.xchg_connect(same_process=ptrs)
        s.xchg_set_timeout(3.0)

    # The kernels wait for one another, so they must RUN together; engines of one process can
    # share a hardware queue (HIP multiplexes streams onto a few), which serialises them.  That is
    # a property of this single-process arrangement, not of the exchange (one process per GPU in
    # production; the multi-process tests cover that).  Whether these two streams run kernels
    # side by side is PROBED first: only if they do not, a time-out below is a skip.
    def streams_run_concurrently():
        e_long, e_short = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(streams[0]):
            torch.cuda._sleep(400_000_000)          # ~0.2 s of spinning on stream 0
            e_long.record()
        with torch.cuda.stream(streams[1]):
            torch.zeros(8, device="cuda").add_(1)
            e_short.record()
        e_short.synchronize()
        side_by_side = not e_long.query()
        torch.cuda.synchronize()
        return side_by_side

    co_scheduled = streams_run_concurrently()

    def sync_all():
        try:
            return [s.sync_act() for s in eng]
        except RuntimeError as err:
            if "timed out" in str(err) and not co_scheduled:
                for s in eng:
                    s.close()
                pytest.skip("probe: the two streams share a hardware queue in this process, "
                            "their kernels cannot be co-scheduled")
            raise                 # co-scheduled streams and still a time-out: a real failure

    for it in range(3):
        for s, st in zip(eng, streams):
            s.solve_exchange_async(st.cuda_stream)
        for s in eng:
            s.flush_async()                   # one host thread drives both: launch all the
        acts = sync_all()                     # deferred exchanges before waiting on the first
        for a in acts:
            assert np.array_equal(a, ref_acts[it]), it
    # three more solves enqueued back to back: the exchange of solve j rides in the rollout launch
    # of solve j+1 of the same engine and waits there for the other engine's words (the grids are
    # small enough for the launches to be resident together)
    for it in range(3):
        for s, st in zip(eng, streams):
            s.solve_exchange_async(st.cuda_stream)
    for s in eng:
        s.flush_async()
    acts = sync_all()
    for a in acts:
        assert np.array_equal(a, ref_acts[5])
    for s in eng:
        assert np.array_equal(s.get_u(), U_ref)
        s.close()


@pytest.mark.gpu
def test_direct_exchange_times_out_instead_of_hanging(gpu):
    """A rank whose peer never shows up gives up after the time-out and reports it; the engine
    stays usable for ordinary solves."""
    import torch
    from mppi_gpu_amd import PointMassModel
    A, K, T = 2, 2000, 50
    c = ol.make_case(A, K, T, seed=62)
    with PointMassModel(K, T, float(c["dt"]), 2 * A, A) as m:
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        _, mine = m.xchg_open(0, 2)
        W = ((T * A + 2 + 15) // 16) * 16
        silent = torch.zeros(2 * 2 * W, dtype=torch.int64, device="cuda")   # rank 1 never writes
        torch.cuda.synchronize()
        m.xchg_connect(same_process=[mine, silent.data_ptr()])
        m.xchg_set_timeout(0.3)
        m.solve_exchange_async()
        with pytest.raises(RuntimeError, match="timed out"):
            m.sync_act()
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        a = m.get_act()
        assert np.all(np.isfinite(a))


# ---- spread-out weights: the exp-weighted update where MANY samples carry weight ----------------

# (A, K, T, kernel, groups per lane | chunks, max_blocks): the BASELINE horizon for every act_dim,
# both fused kernels, the strict kernel, a persistent grid of 3 blocks (every block folds dozens of
# tiles with comparable weights through its running minimum), ragged batch sizes
SPREAD_CASES = [
    (1, 9017, 200, "packed", 4, 0),
    (2, 10000, 200, "row", 0, 0),          # config 2 through the kernel the benchmark uses
    (2, 10000, 200, "packed", 8, 0),
    (3, 24000, 200, "packed", 0, 0),       # config 3's kernel by the engine's own choice
    (3, 9000, 200, "packed", 4, 3),        # persistent grid: 3 blocks x 150 tiles
    (4, 9000, 200, "packed", 10, 0),
    (3, 9000, 200, "row", 0, 0),
    (3, 9000, 200, "row", 16, 3),          # row-aligned on a persistent grid
    (1, 9000, 200, "row", 0, 0),
    (4, 9000, 200, "row", 64, 0),          # one wavefront per trajectory
    (2, 9000, 200, "strict", 0, 0),
    (3, 9001, 400, "packed", 4, 0),        # twice the horizon
]


@pytest.mark.parametrize("frac", [1 / 3, 1 / 8])
@pytest.mark.parametrize("A,K,T,kernel,shape,max_blocks", SPREAD_CASES)
def test_update_with_spread_out_weights(gpu, A, K, T, kernel, shape, max_blocks, frac):
    """lambda chosen so that the oracle's effective sample size is near frac*K (>= 1e3 everywhere):
    the per-wave running minimum / rescale, the wave merge, the block partials and the combine's
    row sums all carry thousands of comparable terms.  Plain bars, no ulp escape."""
    c = ol.make_case(A, K, T, seed=500 + A * 7 + T + max_blocks)
    cost1 = ol.rollout(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    lam = _lambda_for_ess(cost1, _ctrl_term(c["U"], c["E"]), frac * K)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam)
    with _model(gpu, A, K, T, c, chunks=shape if kernel == "row" else 0,
                strict=(kernel == "strict"), max_blocks=max_blocks) as m:
        if kernel == "packed" and shape:
            m.set_packing(shape)
        elif kernel == "row":
            m.set_packing(-1)
        m.set_params(lam)
        m.set_noise(c["E"])
        act = m.get_act()
        inf = m.get_inf(x=False, e=False)
        geo = m.geometry()
    assert geo["packed"] == (kernel == "packed") and geo["strict"] == (kernel == "strict"), geo
    if max_blocks:
        assert geo["grid"] == max_blocks, geo
    _check_solve(act, inf, ref, cost_exact=(kernel == "strict"), lam=lam, spread=0.6 * frac * K,
                 tag=f"spread {kernel} A{A} K{K} T{T} ess~K*{frac:.3f} {geo}")


@pytest.mark.parametrize("A,K,T,packing", [(2, 10000, 200, -1), (3, 9000, 200, 4)])
def test_riding_and_flushed_combine_equal_bits_with_spread_out_weights(gpu, A, K, T, packing):
    """The equal-bits chain of the riding combine at a lambda where a third of the batch carries
    weight (sampling mode): a hand-over that only ever moved the argmin's noise would pass the
    lambda = 1 chains and fail here."""
    c = ol.make_case(A, 1, T, seed=141, u_scale=0.03)
    n = 7
    out = []
    for blocking in (False, True):
        with _model(gpu, A, K, T, c) as m:
            if packing:
                m.set_packing(packing)
            m.set_seed(11)
            m.set_params(400.0)
            m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
            for _ in range(n):
                m.get_act() if blocking else m.solve_async()
            act = m.sync_act()
            inf = m.get_inf(x=False, e=False)
            cnt = m.launch_counts()
        w = inf["weight"].astype(np.float64)
        assert 1.0 / np.sum(w * w) >= K / 20, "the chain is meant to run with spread-out weights"
        if not blocking:
            assert cnt["riding"] == n - 1, cnt
        out.append((act, inf["u"], inf["cost"], inf["beta"], inf["nabla"], inf["weight"]))
    for a, b in zip(*out):
        assert np.array_equal(a, b), "riding and flushed combine must give equal bits"


# ---- BASELINE.json full sizes: size-independent properties + oracle on the device's noise ----

# (A, K, T, k_offset): configs 2 and 3, config 4 whole on one GPU (2.4 GB of noise, 64-bit tile
# offsets) and config 4's LAST of eight shards (global sample indices 875 000 .. 999 999)
FULL = [(2, 10_000, 200, None), (3, 100_000, 200, None), (3, 1_000_000, 200, None),
        (3, 125_000, 200, 875_000)]


@pytest.mark.parametrize("A,K,T,k_offset", FULL)
def test_full_size_parity_and_properties(gpu, A, K, T, k_offset):
    from mppi_gpu_amd import PointMassModel
    c = ol.make_case(A, 1, T, seed=300 + A, u_scale=0.02)

    def _model(gpu, A, K, T, c):          # a shard engine when the case names an offset
        m = PointMassModel(K, T, float(c["dt"]), 2 * A, A, k_offset=k_offset)
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        return m

    with _model(gpu, A, K, T, c) as m:
        m.set_seed(42)
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        act = m.get_act()
        inf = m.get_inf(x=False)
        geo = m.geometry()
    # (1) the oracle, run on the very noise the device drew, reproduces the solve
    ref = ol.solve(c["x0"], c["U"], inf["e"], c["goal"], c["w"], c["dt"])
    _check_solve(act, inf, ref, cost_exact=False, tag=f"full A{A} K{K} off={k_offset} {geo}",
                 plain=True)
    if k_offset is not None:
        # the shard draws the noise of the GLOBAL sample indices: rows 0..63 of the shard are
        # rows k_offset.. of the host statement of the stream
        h = ol.noise(42, 0, k_offset, 64, T, A, [SIGMA] * A)
        np.testing.assert_allclose(inf["e"][:64], h, rtol=0, atol=SIGMA * 2e-5)
    # (2) weights are a probability vector
    assert np.isclose(inf["weight"].astype(np.float64).sum(), 1.0, atol=2e-5)
    assert inf["weight"].min() >= 0 and np.isclose(inf["weight"].max(), 1.0 / inf["nabla"], rtol=1e-5)
    # (3) noise statistics of the whole batch
    z = inf["e"] / np.float32(SIGMA)
    assert abs(z.mean()) < 5e-4 and abs(z.std() - 1) < 5e-4
    # (4) determinism: same seed -> identical bits, different seed -> different action
    with _model(gpu, A, K, T, c) as m2:
        m2.set_seed(42)
        m2.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        act2 = m2.get_act()
        U2 = m2.get_u()
        m2.set_seed(43)
        m2.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        act3 = m2.get_act()
    assert np.array_equal(act, act2) and np.array_equal(U2, inf["u"])
    assert not np.array_equal(act, act3)


@pytest.mark.parametrize("frac", [1 / 3, 1 / 100])
@pytest.mark.parametrize("A,K,T,k_offset", FULL)
def test_full_size_update_with_spread_out_weights(gpu, A, K, T, k_offset, frac):
    """The BASELINE sizes on the device's OWN noise with lambda chosen so that K/3 and K/100 of
    the samples carry the weight (1e3 .. 3e5 effective samples): first solve at lambda 1 to learn
    the costs, choose lambda, start over with the same seed (same noise), solve, and re-run the
    oracle on the noise the device drew.  Plain bars: U / action 1e-5, nabla 1e-5, weights 1e-4."""
    from mppi_gpu_amd import PointMassModel
    c = ol.make_case(A, 1, T, seed=300 + A, u_scale=0.02)
    with PointMassModel(K, T, float(c["dt"]), 2 * A, A, k_offset=k_offset) as m:
        m.set_seed(42)
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        m.get_act()
        first = m.get_inf(x=False, u=False, beta=False, nabla=False, weight=False)
        lam = _lambda_for_ess(first["cost"], _ctrl_term(c["U"], first["e"]), frac * K)
        m.set_params(lam)
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])      # solve counter 0: the same noise
        act = m.get_act()
        inf = m.get_inf(x=False)
        geo = m.geometry()
    assert np.array_equal(inf["e"], first["e"])
    del first
    ref = ol.solve(c["x0"], c["U"], inf["e"], c["goal"], c["w"], c["dt"], lam=lam)
    _check_solve(act, inf, ref, cost_exact=False, lam=lam, spread=0.6 * frac * K,
                 tag=f"full spread A{A} K{K} off={k_offset} ess~K*{frac:.3f} {geo}")


def test_randomised_shapes_against_oracle(gpu):
    """Seeded sweep over random (A, K, T, chunks, max_blocks, lambda, x0, goal, w): strict kernel
    bit-exact in the costs, fused kernel within the stated bar, both on injected noise."""
    from mppi_gpu_amd import MppiError
    rng = np.random.default_rng(20261004)
    done = 0
    for trial in range(40):
        A = int(rng.integers(1, 5))
        K = int(rng.choice([1, 2, 63, 64, 65, 300, 1025, 4097]))
        T = int(rng.choice([1, 2, 3, 7, 16, 31, 50, 99, 128, 200, 257]))
        chunks = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64]))
        lam = float(rng.choice([0.5, 1.0, 2.0]))
        c = ol.make_case(A, K, T, seed=1000 + trial, u_scale=float(rng.choice([0.0, 0.05, 0.5])))
        c["goal"] = rng.standard_normal(2 * A).astype(np.float32)
        c["w"] = np.abs(rng.standard_normal(2 * A) * 5).astype(np.float32)
        ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=lam)
        for strict in (True, False):
            try:
                m = _model(gpu, A, K, T, c, chunks=0 if strict else chunks, strict=strict,
                           max_blocks=int(rng.choice([0, 1, 7])))
            except MppiError as ex:
                assert "chunks must be" in str(ex) or "horizon too long" in str(ex)
                continue
            with m:
                m.set_params(lam)
                m.set_noise(c["E"])
                act = m.get_act()
                inf = m.get_inf(x=False)
                geo = m.geometry()
            _check_solve(act, inf, ref, cost_exact=strict, lam=lam,
                         tag=f"trial {trial} A{A} K{K} T{T} strict={strict} {geo}")
            done += 1
    assert done >= 60


def test_random_api_walks_riding_and_prefetched_equal_flushed(gpu):
    """tools/fuzz_api.py: seeded random sequences of every C-ABI call that touches the solve state
    machine (blocking and asynchronous solves, set_x, lambda / sigma / seed changes, noise store,
    geometry, set_data, flush, every read-out), once with riding combines + forced noise prefetch
    and once with every combine flushed and no prefetch: every read-out along the way must be
    equal bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_api", os.path.join(ROOT, "tools", "fuzz_api.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    used = 0
    for seed in range(1000, 1012):
        shape, a, cnt = fz.walk(seed, 50, "A")
        _, b, _ = fz.walk(seed, 50, "B")
        used += cnt["used"]
        assert len(a) == len(b)
        for i, ((ka, va), (kb, vb)) in enumerate(zip(a, b)):
            assert ka == kb and np.array_equal(va, vb), (seed, shape, i, ka)
    assert used > 20


@pytest.mark.parametrize("inject", [False, True])
def test_read_outs_of_the_last_solve_survive_a_geometry_change(gpu, inject):
    """set_tuning / set_packing re-plan the tile layout and give the noise buffer a new shape; the
    noise (and the state trace worked out from it) of the solve BEFORE the change must still be
    what get_inf hands out: regenerated from the counters, or read from the caller's injected copy
    (found by tools/fuzz_api.py: the buffer used to come back zeroed)."""
    A, K, T = 3, 3000, 50
    c = ol.make_case(A, K, T, seed=77, u_scale=0.03)
    with _model(gpu, A, K, T, c) as m:
        m.set_seed(3)
        if inject:
            m.set_noise(c["E"])
        m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
        m.get_act()
        m.get_act()
        before = m.get_inf()
        assert np.abs(before["e"]).max() > 0
        for change in (lambda: m.set_tuning(chunks=0, strict=False, max_blocks=24),
                       lambda: m.set_packing(4), lambda: m.set_packing(-1)):
            change()
            after = m.get_inf()
            for k in ("e", "x", "u", "cost", "weight"):
                assert np.array_equal(before[k], after[k]), k
        m.get_act()                                    # and the next solve runs in the new geometry
        assert np.isfinite(m.get_u()).all()
