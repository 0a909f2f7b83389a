"""The serial CPU controller (reference `ControllerBase`, BASELINE config 1: point_mass1d,
K=100, T=50, no GPU) against the oracle.  Both are host float code compiled without FMA
contraction, so the path costs must agree bit for bit; weights/controls to rounding."""
import numpy as np
import pytest

import oracle_lib as ol


@pytest.mark.parametrize("A,K,T", [(1, 100, 50), (2, 64, 33), (3, 40, 200), (4, 17, 9)])
def test_controller_base_matches_oracle_on_injected_noise(A, K, T):
    from mppi_gpu_amd import ControllerBase
    c = ol.make_case(A, K, T, seed=A * 10 + 1)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"])
    cb = ControllerBase(K, T, float(c["dt"]), 2 * A, A)
    cb.setActions(c["U"])
    cb.setCost(c["goal"], c["w"])
    cb.setNoise(c["E"])
    act = cb.next(c["x0"])
    st = cb.state()
    assert np.array_equal(st["cost"], ref["cost"])
    assert np.float32(st["beta"]) == ref["beta"]
    np.testing.assert_allclose(st["nabla"], ref["nabla"], rtol=1e-6)
    np.testing.assert_allclose(st["weight"], ref["weights"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(st["u"], ref["U"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(act, ref["next_act"], rtol=0, atol=1e-7)


def test_controller_base_config1_closed_loop_and_noise_stream():
    """BASELINE config 1 shape, sampling mode: the noise is the engine's Philox stream (equal to
    the noise oracle bit for bit -- both evaluate it with libm), it advances per solve, and
    ten closed-loop iterations drive the 1-D point mass towards its goal."""
    from mppi_gpu_amd import ControllerBase
    A, K, T, dt = 1, 100, 50, 0.1
    goal, w = ol.PRESETS[1]["goal"], ol.PRESETS[1]["w"]
    cb = ControllerBase(K, T, dt, 2, 1)
    cb.setCost(goal, w)
    cb.setSeed(7)
    x = np.zeros(2, np.float32)
    d0 = abs(x[0] - goal[0])
    for it in range(10):
        act = cb.next(x)
        st = cb.state()
        assert np.array_equal(st["e"], ol.noise(7, it, 0, K, T, A, [0.025])), it
        assert np.isclose(st["weight"].astype(np.float64).sum(), 1.0, atol=1e-5)
        x = np.array([x[0] + dt * x[1] + 0.5 * dt * dt * act[0], x[1] + dt * act[0]], np.float32)
    assert np.isfinite(x).all()
    assert abs(x[0] - goal[0]) < d0          # moved towards the goal
    assert cb._lib.mppi_cpu_create(10, 5, 0.1, 3, 2) is None     # S != 2A rejected


def test_controller_base_inv_s_and_lambda():
    from mppi_gpu_amd import ControllerBase
    A, K, T = 2, 50, 20
    c = ol.make_case(A, K, T, seed=3)
    inv_s = np.array([2.0, 0.25], np.float32)
    ref = ol.solve(c["x0"], c["U"], c["E"], c["goal"], c["w"], c["dt"], lam=0.7, inv_s=inv_s)
    cb = ControllerBase(K, T, float(c["dt"]), 4, 2)
    cb.setActions(c["U"]); cb.setCost(c["goal"], c["w"]); cb.setParams(0.7, inv_s=inv_s)
    cb.setNoise(c["E"])
    act = cb.next(c["x0"])
    st = cb.state()
    assert np.array_equal(st["cost"], ref["cost"])
    np.testing.assert_allclose(st["u"], ref["U"], rtol=0, atol=1e-7)


def test_controller_base_threads_do_not_change_results():
    """setThreads: the sample loops run on worker threads; samples are independent and every
    control value is still summed over the samples in order, so 1, 3 and 8 threads must give
    identical bits (sampling mode, several iterations, state fed back)."""
    from mppi_gpu_amd import ControllerBase
    A, K, T = 2, 500, 40
    c = ol.make_case(A, K, T, seed=77)
    runs = []
    for threads in (1, 3, 8):
        ctl = ControllerBase(K, T, float(c["dt"]), 2 * A, A)
        ctl.setThreads(threads)
        ctl.setSeed(5)
        ctl.setActions(c["U"])
        ctl.setCost(c["goal"], c["w"])
        x = c["x0"].copy()
        acts = []
        for it in range(3):
            acts.append(ctl.next(x).copy())
            x = (x + np.float32(0.01)).astype(np.float32)
        st = ctl.state()
        runs.append((np.stack(acts), st["u"].copy(), st["cost"].copy(), st["e"].copy()))
        ctl.close()
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert np.array_equal(a, b)
