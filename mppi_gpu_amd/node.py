"""Host-side mirror of the single-process multi-GPU controller (include/point_mass_sharded.hpp,
C ABI include/mppi_gpu_amd_sharded.h): one process, one shard engine and one host worker thread per
GPU, the per-solve exchange through RCCL's all-gather ("collective", default), peer stores from
inside the combine kernel ("direct") or peer copies ("copy").  For Python callers and the tests;
every call goes straight through the C ABI.  (mppi_gpu_amd.sharded is the OTHER arrangement:
one process per GPU under torch.distributed.)"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import MppiError

TRANSPORTS = {"collective": 0, "direct": 1, "copy": 2}


def _fp(a):
    return a.ctypes.data_as(_capi.c_float_p)


def _f32(a, n, name):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError(f"{name}: expected {n} floats, got {a.size}")
    return a


class NodePointMassModel:
    """reference `class PointMassModel` (include/point_mass.hpp:23-44) over `n_shards` engines of
    this process.  devices: HIP ordinal per shard (None: 0..n-1; n_shards=0: all visible)."""

    def __init__(self, nb_sim, steps, dt, state_dim, act_dim, n_shards=0, devices=None,
                 transport="collective", verbose=False):
        self._lib = _capi.load_sharded()
        self._h = _capi.sharded_p()
        self.K, self.T, self.S, self.A = int(nb_sim), int(steps), int(state_dim), int(act_dim)
        dev = None
        if devices is not None:
            dev = (C.c_int * len(devices))(*[int(d) for d in devices])
            n_shards = len(devices)
        rc = self._lib.mppi_sharded_create(self.K, self.T, float(dt), self.S, self.A, int(verbose),
                                           int(n_shards), dev, TRANSPORTS[transport],
                                           C.byref(self._h))
        if rc != 0:
            msg = self._lib.mppi_sharded_last_error().decode(errors="replace")
            if self._h:
                self._lib.mppi_sharded_destroy(self._h)
                self._h = _capi.sharded_p()
            raise MppiError(rc, msg)

    def _check(self, rc):
        if rc != 0:
            raise MppiError(rc, self._lib.mppi_sharded_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mppi_sharded_destroy(self._h)
            self._h = _capi.sharded_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- reference surface ------------------------------------------------------------------
    def memcpy_set_data(self, x, u, goal, w):
        x, u = _f32(x, self.S, "x"), _f32(u, self.T * self.A, "u")
        goal, w = _f32(goal, self.S, "goal"), _f32(w, self.S, "w")
        self._check(self._lib.mppi_sharded_set_data(self._h, _fp(x), _fp(u), _fp(goal), _fp(w)))

    def set_x(self, x):
        x = _f32(x, self.S, "x")
        self._check(self._lib.mppi_sharded_set_x(self._h, _fp(x)))

    def get_x(self):
        x = np.empty(self.S, np.float32)
        self._check(self._lib.mppi_sharded_get_x(self._h, _fp(x)))
        return x

    def get_act(self):
        act = np.empty(self.A, np.float32)
        self._check(self._lib.mppi_sharded_get_act(self._h, _fp(act)))
        return act

    def get_u(self):
        u = np.empty((self.T, self.A), np.float32)
        self._check(self._lib.mppi_sharded_get_u(self._h, _fp(u)))
        return u

    def get_inf(self, x=True, u=True, e=True, cost=True, beta=True, nabla=True, weight=True):
        null = C.cast(None, _capi.c_float_p)
        X = np.empty((self.K, self.T + 1, self.S), np.float32) if x else None
        U = np.empty((self.T, self.A), np.float32) if u else None
        E = np.empty((self.K, self.T, self.A), np.float32) if e else None
        cst = np.empty(self.K, np.float32) if cost else None
        b = np.empty(1, np.float32) if beta else None
        n = np.empty(1, np.float32) if nabla else None
        wt = np.empty(self.K, np.float32) if weight else None
        args = [(_fp(a) if a is not None else null) for a in (X, U, E, cst, b, n, wt)]
        self._check(self._lib.mppi_sharded_get_inf(self._h, *args))
        out = {k: a for k, a in (("x", X), ("u", U), ("e", E), ("cost", cst), ("weight", wt))
               if a is not None}
        if b is not None:
            out["beta"] = float(b[0])
        if n is not None:
            out["nabla"] = float(n[0])
        return out

    def memcpy_get_data(self):
        X = np.empty((self.K, self.T + 1, self.S), np.float32)
        E = np.empty((self.K, self.T, self.A), np.float32)
        self._check(self._lib.mppi_sharded_get_data(self._h, _fp(X), _fp(E)))
        return X, E

    # -- additions ----------------------------------------------------------------------------
    def solve_async(self):
        self._check(self._lib.mppi_sharded_solve_async(self._h))

    def sync_act(self):
        act = np.empty(self.A, np.float32)
        self._check(self._lib.mppi_sharded_sync_act(self._h, _fp(act)))
        return act

    def set_params(self, lam, sigma=None, inv_s=None):
        null = C.cast(None, _capi.c_float_p)
        sg = _f32(sigma, self.A, "sigma") if sigma is not None else None
        iv = _f32(inv_s, self.A, "inv_s") if inv_s is not None else None
        self._check(self._lib.mppi_sharded_set_params(self._h, float(lam),
                                                      _fp(sg) if sg is not None else null,
                                                      _fp(iv) if iv is not None else null))

    def set_seed(self, seed):
        self._check(self._lib.mppi_sharded_set_seed(self._h, int(seed)))

    def set_noise(self, e):
        if e is None:
            self._check(self._lib.mppi_sharded_set_noise(self._h, C.cast(None, _capi.c_float_p)))
            return
        e = _f32(e, self.K * self.T * self.A, "noise")
        self._check(self._lib.mppi_sharded_set_noise(self._h, _fp(e)))

    def set_action_limit(self, max_a):
        if max_a is None:
            self._check(self._lib.mppi_sharded_set_action_limit(self._h, C.cast(None, _capi.c_float_p)))
            return
        m = _f32(max_a, self.A, "max_a")
        self._check(self._lib.mppi_sharded_set_action_limit(self._h, _fp(m)))

    def set_timeout(self, seconds):
        self._check(self._lib.mppi_sharded_set_timeout(self._h, float(seconds)))

    @property
    def n_shards(self):
        return self._lib.mppi_sharded_n_shards(self._h)

    @property
    def transport(self):
        t = self._lib.mppi_sharded_transport(self._h)
        return {v: k for k, v in TRANSPORTS.items()}.get(t, "?")

    def shard_info(self, i):
        out = (C.c_longlong * 3)()
        self._check(self._lib.mppi_sharded_shard_info(self._h, int(i), out))
        return {"k_begin": int(out[0]), "k_end": int(out[1]), "device": int(out[2])}

    def engine_launch_counts(self, i):
        """launch counters of shard i's engine (mppi_get_launch_counts)."""
        eng = self._lib.mppi_sharded_engine(self._h, int(i))
        out = (C.c_longlong * 4)()
        _capi.check(_capi.load().mppi_get_launch_counts(eng, out))
        return {"rollout": int(out[0]), "riding": int(out[1]), "combine": int(out[2]),
                "resident_ride": int(out[3])}
