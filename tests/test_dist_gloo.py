"""World-size-2 rehearsal of the sharded path on CPU (gloo): the REAL orchestration code
(mppi_gpu_amd/sharded.py: shard ranges, partial layout, all-gather, call order) runs in two
processes; the two GPU entry points are replaced by a test double that computes the rank-local
partial and the final combine with the CPU oracle / numpy.  The result must equal the oracle's
single-process solve of the whole batch.  (The GPU kernels behind the same two entry points are
checked against a single engine in tests/test_gpu_parity.py::test_sharded_engines_equal_single_engine.)
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as ol  # noqa: E402


class _OracleShardEngine:
    """Test double for the two sharded entry points of the C ABI, on numpy + the oracle."""

    def __init__(self, k_local, k_offset, case, seed):
        self.c, self.seed = case, seed
        self.K, self.off = k_local, k_offset
        self.T, self.A = case["U"].shape
        self.U = case["U"].copy()
        self.x0 = case["x0"].copy()
        self.solve_idx = 0
        self.act = None

    def partial_len(self):
        return self.T * self.A + 2

    def memcpy_set_data(self, x, u, goal, w):
        self.x0, self.U = np.asarray(x, np.float32), np.asarray(u, np.float32).reshape(self.T, self.A)

    def solve_local_async(self, ptr, stream):
        E = ol.noise(self.seed, self.solve_idx, self.off, self.K, self.T, self.A, [0.025] * self.A)
        cost = ol.rollout(self.x0, self.U, E, self.c["goal"], self.c["w"], self.c["dt"])
        beta = cost.min()
        ex = np.exp(-(cost - beta).astype(np.float64))
        out = np.concatenate([[beta, ex.sum()], (ex[:, None] * E.reshape(self.K, -1)).sum(0)])
        self._out_tensor.copy_(torch.from_numpy(out.astype(np.float32)))

    def solve_finish_async(self, ptr, n_parts, stream):
        g = self._gathered_tensor.numpy().reshape(n_parts, -1).astype(np.float64)
        beta = g[:, 0].min()
        r = np.exp(-(g[:, 0] - beta))
        nabla = (r * g[:, 1]).sum()
        dU = (r[:, None] * g[:, 2:]).sum(0) / nabla
        full = (self.U.reshape(-1).astype(np.float64) + dU).astype(np.float32).reshape(self.T, self.A)
        self.act = full[0].copy()
        self.U = np.concatenate([full[1:], full[-1:]])
        self.solve_idx += 1

    def sync_act(self):
        return self.act

    def get_u(self):
        return self.U

    def close(self):
        pass


def _worker(rank, world, port, K, A, T, seed, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mppi_gpu_amd.sharded import ShardedPointMassModel, shard_range
    case = ol.make_case(A, 1, T, seed=5, u_scale=0.03)
    engines = {}

    def factory(k, off):
        engines["e"] = _OracleShardEngine(k, off, case, seed)
        return engines["e"]

    m = ShardedPointMassModel(K, T, float(case["dt"]), 2 * A, A, engine_factory=factory,
                              tensor_factory=lambda n: torch.zeros(n, dtype=torch.float32))
    engines["e"]._out_tensor = m._partial
    engines["e"]._gathered_tensor = m._gathered
    assert (m.k_begin, m.k_end) == shard_range(K, rank, world)
    m.memcpy_set_data(case["x0"], case["U"], case["goal"], case["w"])
    acts = [m.get_act().copy() for _ in range(3)]
    ret[rank] = (np.stack(acts), m.get_u().copy(), m.k_begin, m.k_end)
    dist.destroy_process_group()


def test_shard_range_is_a_partition():
    from mppi_gpu_amd.sharded import shard_range
    for K, W in ((10, 3), (1000000, 8), (8, 8), (100001, 4)):
        spans = [shard_range(K, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == K
        assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(3, 0, 4)


def test_two_rank_gloo_sharded_solve_equals_single_process_oracle():
    K, A, T, seed, world = 301, 3, 40, 17, 2          # uneven split: 151 + 150
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, K, A, T, seed, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    acts0, U0, b0, e0 = ret[0]
    acts1, U1, b1, e1 = ret[1]
    assert (b0, e0, b1, e1) == (0, 151, 151, 301)
    assert np.array_equal(acts0, acts1) and np.array_equal(U0, U1), "ranks must agree bitwise"
    # single-process oracle on the whole batch, same global noise stream
    case = ol.make_case(A, 1, T, seed=5, u_scale=0.03)
    U, x0 = case["U"].copy(), case["x0"]
    for it in range(3):
        E = ol.noise(seed, it, 0, K, T, A, [0.025] * A)
        ref = ol.solve(x0, U, E, case["goal"], case["w"], case["dt"])
        scale = max(float(np.abs(ref["U"]).max()), 0.025)
        assert np.abs(acts0[it] - ref["next_act"]).max() <= 1e-5 * scale, it
        U = ref["U"]
    assert np.abs(U0 - U).max() <= 1e-5 * max(float(np.abs(U).max()), 0.025)


def _gpu_worker(rank, world, port, K, A, T, seed, transport, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch as th
    th.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mppi_gpu_amd.sharded import ShardedPointMassModel
    case = ol.make_case(A, 1, T, seed=5, u_scale=0.03)
    m = ShardedPointMassModel(K, T, float(case["dt"]), 2 * A, A, transport=transport)
    m.engine.set_seed(seed)
    m.memcpy_set_data(case["x0"], case["U"], case["goal"], case["w"])
    acts = [m.get_act().copy() for _ in range(3)]
    m.set_x(case["x0"] * 0.5)                      # a state update between exchanges
    acts.append(m.get_act().copy())
    for _ in range(3):                             # back to back: the exchange rides (direct)
        m.solve_async()
    acts.append(m.sync_act().copy())
    ret[rank] = (np.stack(acts), m.get_u().copy(), m.transport)
    m.close()
    dist.destroy_process_group()


def _run_gpu_ranks(world, K, A, T, seed, transport, port_base):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = port_base + (os.getpid() % 2000)
    mp.spawn(_gpu_worker, args=(world, port, K, A, T, seed, transport, ret), nprocs=world,
             join=True)
    assert sorted(ret.keys()) == list(range(world))
    for r in range(1, world):
        assert np.array_equal(ret[0][0], ret[r][0]) and np.array_equal(ret[0][1], ret[r][1]), \
            "ranks must agree bitwise"
        assert ret[r][2] == ret[0][2]
    return ret[0]


@pytest.mark.gpu
def test_two_processes_sharing_one_gpu_equal_single_engine(gpu):
    """The real multi-process device path: two ranks (two processes, both on cuda:0, gloo as the
    transport because RCCL wants one GPU per rank) run the sharded engine kernels
    (solve_local / finish) and must reproduce the single-engine solve of the whole batch."""
    from mppi_gpu_amd import PointMassModel
    K, A, T, seed, world = 5001, 2, 200, 23, 2
    acts0, U0, tr = _run_gpu_ranks(world, K, A, T, seed, "collective", 29700)
    assert tr == "collective"
    case = ol.make_case(A, 1, T, seed=5, u_scale=0.03)
    with PointMassModel(K, T, float(case["dt"]), 2 * A, A) as m:
        m.set_seed(seed)
        m.memcpy_set_data(case["x0"], case["U"], case["goal"], case["w"])
        for it in range(4):
            if it == 3:
                m.set_x(case["x0"] * 0.5)
            a = m.get_act()
            scale = max(float(np.abs(m.get_u()).max()), 0.025)
            assert np.abs(a - acts0[it]).max() <= 2e-6 * scale, it
        for _ in range(3):
            m.solve_async()
        a = m.sync_act()
        scale = max(float(np.abs(m.get_u()).max()), 0.025)
        assert np.abs(a - acts0[4]).max() <= 1e-5 * scale
        assert np.abs(m.get_u() - U0).max() <= 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("world,K,A,T", [(2, 5001, 2, 200), (3, 4000, 3, 50)])
def test_direct_peer_exchange_equals_collective_bitwise(gpu, world, K, A, T):
    """The direct exchange (hipIpc-mapped inboxes, tagged 8-byte words written by the combine
    kernel of one PROCESS and polled by the combine kernel of another) must give the bits of
    local combine + all-gather + finish, on every rank; "auto" must have validated and kept it."""
    seed = 31
    acts_c, U_c, tr_c = _run_gpu_ranks(world, K, A, T, seed, "collective", 31700)
    acts_d, U_d, tr_d = _run_gpu_ranks(world, K, A, T, seed, "direct", 33700)
    acts_a, U_a, tr_a = _run_gpu_ranks(world, K, A, T, seed, "auto", 35700)
    assert (tr_c, tr_d, tr_a) == ("collective", "direct", "direct")
    assert np.array_equal(acts_c, acts_d) and np.array_equal(U_c, U_d)
    assert np.array_equal(acts_c, acts_a) and np.array_equal(U_c, U_a)


@pytest.mark.gpu
def test_four_ranks_on_one_gpu_do_not_wait_for_each_other_in_vain(gpu):
    """Four processes on ONE GPU at the bench's shard size (625 blocks each): their riding launches
    wait for each other's words, and all four do not fit the chip together -- the engine must see
    the peers' inboxes on its own device and launch those combines on their own instead of
    running into the exchange time-out."""
    seed, world, K, A, T = 7, 4, 40000, 2, 200
    acts_c, U_c, _ = _run_gpu_ranks(world, K, A, T, seed, "collective", 37700)
    acts_d, U_d, tr = _run_gpu_ranks(world, K, A, T, seed, "direct", 39700)
    assert tr == "direct"
    assert np.array_equal(acts_c, acts_d) and np.array_equal(U_c, U_d)
