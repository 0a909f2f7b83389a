"""Sample-sharded MPPI over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU rehearsal tests).

The K samples of a solve are independent, so rank r owns the contiguous range
[k_begin, k_end) of the global batch and draws its noise from the Philox subsequences of those
GLOBAL indices: the noise, and therefore the controls, do not depend on the number of ranks.
The only exchange per solve is T*A+2 floats per rank,
    [beta_g, S_g, N_g[T*A]]   beta_g = min cost, S_g = sum exp(-(c-beta_g)/lambda),
                              N_g    = sum exp(-(c-beta_g)/lambda) * E
after which every rank combines the G partials in rank order (bitwise identical on all ranks):
    beta = min beta_g, r_g = exp(-(beta_g-beta)/lambda), nabla = sum r_g S_g,
    U += sum r_g N_g / nabla, then the shift.
Two transports carry it, with identical results:
    "collective"  (default) rank-local combine, ONE RCCL all-gather over xGMI, final combine:
                  three launches.  What BASELINE.json's north_star names.
    "direct"      the rank-local combine kernel stores the partial straight into every rank's
                  inbox over xGMI (hipIpc-mapped uncached memory, 8-byte {value, tag} words) and
                  polls its own inbox: rollout + ONE launch per solve, no collective library on
                  the data path (torch.distributed only hands the ipc handles round once).
                  Fewer launches and no collective latency, but it leans on peer-mapped memory.
"auto" (what bench.py asks for) opens the direct exchange, checks on the first memcpy_set_data that
one solve through it reproduces the collective's bits on every rank, and otherwise stays on the
collective; the verdict is in `.validated`.  (A C++ host that is ONE process uses
include/point_mass_sharded.hpp instead: the same two transports without torch.)
The reference has no multi-GPU path; this is new.
"""
from . import PointMassModel


def shard_range(k_global, rank, world):
    """Contiguous, balanced split of range(k_global): the first k_global % world ranks get one
    sample more.  Returns (k_begin, k_end)."""
    if not (0 <= rank < world) or k_global < world:
        raise ValueError("need 0 <= rank < world <= k_global")
    base, extra = divmod(k_global, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class ShardedPointMassModel:
    """PointMassModel over all ranks of `group`.  Same call protocol as the reference class
    (memcpy_set_data once, then get_act / set_x per control step); every rank must make the
    same calls and gets the same action back.

    engine_factory / tensor_factory exist so that the CPU rehearsal tests can run the identical
    orchestration code (shard ranges, gather layout, call order) over gloo with a stand-in for
    the two GPU entry points; production code never passes them."""

    def __init__(self, nb_sim_global, steps, dt, state_dim, act_dim, group=None,
                 engine_factory=None, tensor_factory=None, transport="collective"):
        if transport not in ("auto", "direct", "collective"):
            raise ValueError("transport must be auto, direct or collective")
        import torch
        import torch.distributed as dist
        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.K_global = int(nb_sim_global)
        self.k_begin, self.k_end = shard_range(self.K_global, self.rank, self.world)
        k_local = self.k_end - self.k_begin
        make = engine_factory or (lambda k, off: PointMassModel(k, steps, dt, state_dim, act_dim,
                                                                k_offset=off))
        self.engine = make(k_local, self.k_begin)
        self.L = self.engine.partial_len()
        if tensor_factory is None:
            def tensor_factory(n):
                return torch.zeros(n, device="cuda", dtype=torch.float32)
        self._partial = tensor_factory(self.L)
        self._gathered = tensor_factory(self.L * self.world)
        self._stream = None
        self._tstream = None
        self._host_staging = None
        if self._partial.is_cuda:
            # A dedicated torch stream carries the engine's kernels AND the collective, in order.
            # (torch's default stream has the raw handle 0, which the C ABI reads as "use the
            # engine's own stream": kernels and all-gather would then run unordered.)
            self._tstream = torch.cuda.Stream()
            self._stream = self._tstream.cuda_stream
            assert self._stream != 0
            if dist.get_backend(group) != "nccl":
                # rehearsal transport (gloo): gather through pinned host tensors.  RCCL is the
                # production transport; this path exists so that the multi-process device code
                # can be exercised with several ranks on ONE GPU, which RCCL does not allow.
                self._host_staging = (torch.zeros(self.L, dtype=torch.float32).pin_memory(),
                                      torch.zeros(self.L * self.world, dtype=torch.float32).pin_memory())
        self.transport = "collective"
        self._validated = True
        self.validated = None       # verdict of the "auto" check: True = direct reproduced the
                                    # collective's bits on every rank, False = fell back
        if self._tstream is not None and transport != "collective":
            if self._open_direct():
                self.transport = "direct"
                self._validated = transport == "direct"     # "auto" checks it on first use
            elif transport == "direct":
                raise RuntimeError("direct peer exchange could not be opened on every rank")

    # -- control-plane helpers (setup only; nothing here runs per solve) ------------------------
    def _ctl_device(self):
        return "cuda" if self._dist.get_backend(self._group) == "nccl" else "cpu"

    def _all_ok(self, ok):
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._ctl_device())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self._group)
        return bool(t.item())

    def _open_direct(self):
        """Allocate the inbox, hand the ipc handles round, map the peers.  True on every rank or
        False on every rank."""
        import torch
        ok, handle = True, b""
        try:
            handle, _ = self.engine.xchg_open(self.rank, self.world)
        except RuntimeError:
            ok = False
        n = len(handle) if ok else 64
        mine = torch.frombuffer(bytearray(handle if ok else bytes(n)), dtype=torch.uint8)
        mine = mine.to(self._ctl_device())
        every = torch.zeros(n * self.world, dtype=torch.uint8, device=mine.device)
        self._dist.all_gather_into_tensor(every, mine, group=self._group)
        if self._all_ok(ok):
            try:
                self.engine.xchg_connect(handles=bytes(every.cpu().numpy().tobytes()))
            except RuntimeError:
                ok = False
        else:
            ok = False
        ok = self._all_ok(ok)
        if ok:
            # generous: the first exchange may meet a peer that is still loading its code objects
            self.engine.xchg_set_timeout(20.0)
        if not ok:
            try:
                self.engine.xchg_close()
            except RuntimeError:
                pass
        return ok

    def _validate_direct(self, x, u, goal, w):
        """One solve through each transport from the same state: equal bits on every rank or the
        collective stays."""
        import numpy as np
        self.engine.memcpy_set_data(x, u, goal, w)
        self._solve_collective()
        self.sync_act()
        u_ref = self.engine.get_u()
        self.engine.memcpy_set_data(x, u, goal, w)      # same state, same noise (solve index 0)
        same = False
        try:
            self._solve_direct()
            self.sync_act()
            same = bool(np.array_equal(u_ref, self.engine.get_u()))
        except RuntimeError:                            # exchange timed out
            same = False
        if not self._all_ok(same):
            self.transport = "collective"
        self._validated = True
        self.validated = self.transport == "direct"

    def memcpy_set_data(self, x, u, goal, w):
        if not self._validated:
            self._validate_direct(x, u, goal, w)
        self.engine.memcpy_set_data(x, u, goal, w)

    def set_x(self, x):
        self.engine.set_x(x)

    def solve_async(self):
        """Enqueue one sharded solve."""
        if self._tstream is None:                    # CPU rehearsal (test doubles)
            self.engine.solve_local_async(self._partial.data_ptr(), None)
            self._dist.all_gather_into_tensor(self._gathered, self._partial, group=self._group)
            self.engine.solve_finish_async(self._gathered.data_ptr(), self.world, None)
        elif self.transport == "direct":
            self._solve_direct()
        else:
            self._solve_collective()

    def _solve_direct(self):
        """rollout + one combine launch that exchanges the partials itself."""
        self.engine.solve_exchange_async(self._stream)

    def _solve_collective(self):
        """local rollout + reduction, all-gather, combine."""
        import torch
        with torch.cuda.stream(self._tstream):
            self.engine.solve_local_async(self._partial.data_ptr(), self._stream)
            if self._host_staging is None:
                self._dist.all_gather_into_tensor(self._gathered, self._partial, group=self._group)
            else:
                hp, hg = self._host_staging
                hp.copy_(self._partial, non_blocking=True)       # stream-ordered D2H
                self._tstream.synchronize()
                self._dist.all_gather_into_tensor(hg, hp, group=self._group)
                self._gathered.copy_(hg, non_blocking=True)
            self.engine.solve_finish_async(self._gathered.data_ptr(), self.world, self._stream)

    def get_act(self):
        self.solve_async()
        return self.sync_act()

    def sync_act(self):
        if self._tstream is not None:
            self._tstream.synchronize()
        return self.engine.sync_act()

    def get_u(self):
        return self.engine.get_u()

    def close(self):
        self.engine.close()
