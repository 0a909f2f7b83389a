"""mppi_gpu_amd -- MI355X-native MPPI rollout-and-update engine (point-mass systems).

Host-side mirror of the reference's controller interface for tests, the benchmark and Python
callers.  `PointMassModel` has the public members of the reference's C++ class of the same
name (reference include/point_mass.hpp:23-44) with numpy arrays in place of raw float
pointers; every call goes straight through the C ABI (include/mppi_gpu_amd.h) into the HIP
library.  Nothing here computes anything on the CPU.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import MppiError, check  # noqa: F401

__all__ = ["PointMassModel", "ControllerBase", "MppiError", "device_count", "version"]


def _fp(a):
    return a.ctypes.data_as(_capi.c_float_p)


def _f32(a, n, name):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError(f"{name}: expected {n} floats, got {a.size}")
    return a


def device_count():
    return _capi.load().mppi_device_count()


def version():
    return _capi.load().mppi_version().decode()


class PointMassModel:
    """reference `class PointMassModel` (include/point_mass.hpp:23-44).

    PointMassModel(nb_sim, steps, dt, state_dim, act_dim, verbose=False)
    k_offset != None creates a shard of a larger global batch (multi-GPU path).
    """

    def __init__(self, nb_sim, steps, dt, state_dim, act_dim, verbose=False, k_offset=None):
        self._lib = _capi.load()
        self._h = _capi.engine_p()
        self.K, self.T, self.S, self.A = int(nb_sim), int(steps), int(state_dim), int(act_dim)
        if k_offset is None:
            rc = self._lib.mppi_create(self.K, self.T, float(dt), self.S, self.A, int(verbose),
                                       C.byref(self._h))
        else:
            rc = self._lib.mppi_create_shard(self.K, int(k_offset), self.T, float(dt), self.S,
                                             self.A, int(verbose), C.byref(self._h))
        if rc != 0:
            msg = self._lib.mppi_last_error().decode(errors="replace")
            if self._h:
                self._lib.mppi_destroy(self._h)
                self._h = _capi.engine_p()
            raise MppiError(rc, msg)
        # the closed-loop call's own buffer and argument objects (a.ctypes.data_as per call costs
        # more than a microsecond; the blocking solve is twenty)
        self._act = np.empty(self.A, np.float32)
        self._act_p = _fp(self._act)
        self._c_get_act = self._lib.mppi_get_act
        self._xbuf = np.empty(self.S, np.float32)
        self._xbuf_p = _fp(self._xbuf)
        self._null_stream = C.c_void_p(0)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mppi_destroy(self._h)
            self._h = _capi.engine_p()

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
        """reference src/point_mass.cu:205-228"""
        x = _f32(x, self.S, "x")
        u = _f32(u, self.T * self.A, "u")
        goal = _f32(goal, self.S, "goal")
        w = _f32(w, self.S, "w")
        check(self._lib.mppi_set_data(self._h, _fp(x), _fp(u), _fp(goal), _fp(w)))

    def set_x(self, x):
        """reference src/point_mass.cu:482-486"""
        self._xbuf[...] = _f32(x, self.S, "x")
        rc = self._lib.mppi_set_x(self._h, self._xbuf_p)
        if rc != 0:
            check(rc)

    def get_x(self):
        """declared in reference include/point_mass.hpp:34 (never defined there)"""
        x = np.empty(self.S, np.float32)
        check(self._lib.mppi_get_x(self._h, _fp(x)))
        return x

    def get_act(self, out=None):
        """One full MPPI solve; reference src/point_mass.cu:129-203. Returns next_act[A]
        (written into `out`, a float32 array of A elements, when given: no allocation per call)."""
        rc = self._c_get_act(self._h, self._act_p)
        if rc != 0:
            check(rc)
        if out is None:
            return self._act.copy()
        out[...] = self._act
        return out

    def get_u(self):
        """reference src/point_mass.cu:488-491"""
        u = np.empty((self.T, self.A), np.float32)
        check(self._lib.mppi_get_u(self._h, _fp(u)))
        return u

    def memcpy_get_data(self):
        """reference src/point_mass.cu:230-234 -> (X[K][T+1][S], E[K][T][A])"""
        X = np.empty((self.K, self.T + 1, self.S), np.float32)
        E = np.empty((self.K, self.T, self.A), np.float32)
        check(self._lib.mppi_get_data(self._h, _fp(X), _fp(E)))
        return X, E

    def get_inf(self, x=True, u=True, e=True, cost=True, beta=True, nabla=True, weight=True):
        """reference src/point_mass.cu:236-262; returns a dict of the requested arrays."""
        out = {}
        null = C.cast(None, _capi.c_float_p)
        X = np.empty((self.K, self.T + 1, self.S), np.float32) if x else None
        U = np.empty((self.T, self.A), np.float32) if u else None
        E = np.empty((self.K, self.T, self.A), np.float32) if e else None
        cst = np.empty(self.K, np.float32) if cost else None
        b = np.empty(1, np.float32) if beta else None
        n = np.empty(1, np.float32) if nabla else None
        wt = np.empty(self.K, np.float32) if weight else None
        args = [(_fp(a) if a is not None else null) for a in (X, U, E, cst, b, n, wt)]
        check(self._lib.mppi_get_inf(self._h, *args))
        for name, a in (("x", X), ("u", U), ("e", E), ("cost", cst), ("weight", wt)):
            if a is not None:
                out[name] = a
        if b is not None:
            out["beta"] = float(b[0])
        if n is not None:
            out["nabla"] = float(n[0])
        return out

    # -- extensions -------------------------------------------------------------------------
    def set_params(self, lam=1.0, sigma=None, inv_s=None):
        null = C.cast(None, _capi.c_float_p)
        s = _f32(sigma, self.A, "sigma") if sigma is not None else None
        i = _f32(inv_s, self.A, "inv_s") if inv_s is not None else None
        check(self._lib.mppi_set_params(self._h, float(lam), _fp(s) if s is not None else null,
                                        _fp(i) if i is not None else null))

    def set_seed(self, seed):
        check(self._lib.mppi_set_seed(self._h, int(seed)))

    def set_noise(self, E):
        if E is None:
            check(self._lib.mppi_set_noise(self._h, C.cast(None, _capi.c_float_p)))
            return
        E = _f32(E, self.K * self.T * self.A, "E")
        check(self._lib.mppi_set_noise(self._h, _fp(E)))

    def set_noise_store(self, on):
        """False: the rollout does not write the sampled noise to HBM; get_inf regenerates it."""
        check(self._lib.mppi_set_noise_store(self._h, int(bool(on))))

    def set_ref_compat(self, on):
        check(self._lib.mppi_set_ref_compat(self._h, int(bool(on))))

    def set_action_limit(self, max_a):
        """Opt-in clamp of the updated controls to +-max_a[axis] (None = off, the reference's
        effective behaviour: it parses max-a and drops it, src/main.cu:566-568)."""
        if max_a is None:
            check(self._lib.mppi_set_action_limit(self._h, C.cast(None, _capi.c_float_p)))
            return
        m = _f32(max_a, self.A, "max_a")
        check(self._lib.mppi_set_action_limit(self._h, _fp(m)))

    def set_tuning(self, chunks=0, strict=False, max_blocks=0):
        check(self._lib.mppi_set_tuning(self._h, int(chunks), int(bool(strict)), int(max_blocks)))

    def set_packing(self, groups_per_lane):
        """0 auto (default), -1 never, n > 0 force the packed kernel with n groups per lane."""
        check(self._lib.mppi_set_packing(self._h, int(groups_per_lane)))

    def set_pipeline(self, mode):
        """0 deferred combine (default), 1 eager; see the header."""
        check(self._lib.mppi_set_pipeline(self._h, int(mode)))

    def set_noise_prefetch(self, mode):
        """0 never, 1 auto (default), 2 whenever possible; see the header."""
        check(self._lib.mppi_set_noise_prefetch(self._h, int(mode)))

    def prefetch_counts(self):
        out = (C.c_longlong * 2)()
        check(self._lib.mppi_get_prefetch_counts(self._h, out))
        return {"launched": int(out[0]), "used": int(out[1])}

    def pipeline(self):
        """{"mode": 0 | 1, "degraded": the engine chose mode 1 itself after a watchdog trip}"""
        mode, deg = C.c_int(), C.c_int()
        check(self._lib.mppi_get_pipeline(self._h, C.byref(mode), C.byref(deg)))
        return {"mode": mode.value, "degraded": bool(deg.value)}

    def geometry(self):
        g = (C.c_int * 5)()
        check(self._lib.mppi_get_geometry(self._h, g))
        lay = (C.c_int * 4)()
        check(self._lib.mppi_get_layout(self._h, lay))
        return {"chunks": g[0], "nq": g[1], "grid": g[2], "block": g[3], "strict": bool(g[4]),
                "packed": bool(lay[0]), "groups_per_lane": lay[1], "trajectories_per_wave": lay[2],
                "tile_groups": lay[3]}

    def launch_counts(self):
        """Launches since creation: rollout launches, those that carried a riding combine,
        stand-alone combine launches, and the blocks of the riding kernel the chip holds at once."""
        c = (C.c_longlong * 4)()
        check(self._lib.mppi_get_launch_counts(self._h, c))
        return {"rollout": c[0], "riding": c[1], "combine": c[2], "resident_ride": c[3]}

    # -- asynchronous / sharded ---------------------------------------------------------------
    def solve_async(self, stream=None):
        rc = self._lib.mppi_solve_async(self._h, C.c_void_p(stream) if stream else self._null_stream)
        if rc != 0:
            check(rc)

    def flush_async(self):
        check(self._lib.mppi_flush_async(self._h))

    def sync_act(self):
        rc = self._lib.mppi_sync_act(self._h, self._act_p)
        if rc != 0:
            check(rc)
        return self._act.copy()

    def partial_len(self):
        return self._lib.mppi_partial_len(self._h)

    def solve_local_async(self, d_partial_ptr, stream=None):
        check(self._lib.mppi_solve_local_async(self._h, C.c_void_p(d_partial_ptr),
                                               C.c_void_p(stream or 0)))

    def solve_finish_async(self, d_gathered_ptr, n_parts, stream=None):
        check(self._lib.mppi_solve_finish_async(self._h, C.c_void_p(d_gathered_ptr), int(n_parts),
                                                C.c_void_p(stream or 0)))

    # -- direct peer exchange (include/mppi_gpu_amd.h "Direct peer exchange") -----------------
    def xchg_open(self, rank, world):
        """Allocate this rank's inbox; returns (ipc_handle_bytes, raw_device_pointer)."""
        n = self._lib.mppi_xchg_handle_bytes()
        buf = C.create_string_buffer(n)
        ptr = C.c_void_p()
        check(self._lib.mppi_xchg_open(self._h, int(rank), int(world), buf, C.byref(ptr)))
        return buf.raw, ptr.value

    def xchg_connect(self, handles=None, same_process=None):
        """handles: the world ipc handles concatenated in rank order (bytes) or None;
        same_process: list of raw inbox pointers (None/0 where the handle is to be used)."""
        hb = C.create_string_buffer(handles, len(handles)) if handles is not None else None
        arr = None
        if same_process is not None:
            arr = (C.c_void_p * len(same_process))(*[C.c_void_p(p or 0) for p in same_process])
        check(self._lib.mppi_xchg_connect(self._h, hb, arr))

    def xchg_set_timeout(self, seconds):
        check(self._lib.mppi_xchg_set_timeout(self._h, float(seconds)))

    def solve_exchange_async(self, stream=None):
        check(self._lib.mppi_solve_exchange_async(self._h, C.c_void_p(stream or 0)))

    def xchg_close(self):
        check(self._lib.mppi_xchg_close(self._h))

    # -- measurement ------------------------------------------------------------------------
    def set_profiling(self, every):
        """every > 0: record HIP events around the kernels of each `every`-th solve."""
        check(self._lib.mppi_set_profiling(self._h, int(every)))

    def kernel_ms(self, which):
        avg = C.c_double()
        n = C.c_int()
        check(self._lib.mppi_kernel_ms(self._h, int(which), C.byref(avg), C.byref(n)))
        return avg.value, n.value


class ControllerBase:
    """The serial CPU MPPI controller: reference `class ControllerBase`
    (include/controller_base.hpp:7-42; ctor (k, tau, dt, sDim, aDim), next(x), setActions) made
    real in C++ (mppi_gpu_amd/csrc/controller_base.cpp).  BASELINE config 1; needs no GPU and is
    never used as a fallback by PointMassModel."""

    def __init__(self, k, tau, dt, sDim, aDim):
        self._lib = _capi.load()
        self.K, self.T, self.S, self.A = int(k), int(tau), int(sDim), int(aDim)
        self._h = self._lib.mppi_cpu_create(self.K, self.T, float(dt), self.S, self.A)
        if not self._h:
            raise ValueError("ControllerBase: invalid dimensions")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mppi_cpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setActions(self, actions):
        u = _f32(actions, self.T * self.A, "actions")
        null = C.cast(None, _capi.c_float_p)
        check(self._lib.mppi_cpu_set_data(self._h, _fp(u), null, null))
        return True

    def setCost(self, goal, w):
        g = _f32(goal, self.S, "goal")
        ww = _f32(w, self.S, "w")
        check(self._lib.mppi_cpu_set_data(self._h, C.cast(None, _capi.c_float_p), _fp(g), _fp(ww)))

    def setParams(self, lam=1.0, sigma=None, inv_s=None):
        null = C.cast(None, _capi.c_float_p)
        s = _f32(sigma, self.A, "sigma") if sigma is not None else None
        i = _f32(inv_s, self.A, "inv_s") if inv_s is not None else None
        check(self._lib.mppi_cpu_set_params(self._h, float(lam), _fp(s) if s is not None else null,
                                            _fp(i) if i is not None else null))

    def setThreads(self, n):
        """Worker threads for the sample loops (default 1); results do not depend on the count."""
        check(self._lib.mppi_cpu_set_threads(self._h, int(n)))

    def setSeed(self, seed):
        check(self._lib.mppi_cpu_set_seed(self._h, int(seed)))

    def setNoise(self, E):
        if E is None:
            check(self._lib.mppi_cpu_set_noise(self._h, C.cast(None, _capi.c_float_p)))
        else:
            e = _f32(E, self.K * self.T * self.A, "E")
            check(self._lib.mppi_cpu_set_noise(self._h, _fp(e)))

    def next(self, x):
        xx = _f32(x, self.S, "x")
        act = np.empty(self.A, np.float32)
        check(self._lib.mppi_cpu_next(self._h, _fp(xx), _fp(act)))
        return act

    def state(self):
        u = np.empty((self.T, self.A), np.float32)
        e = np.empty((self.K, self.T, self.A), np.float32)
        cost = np.empty(self.K, np.float32)
        w = np.empty(self.K, np.float32)
        b = np.empty(1, np.float32)
        n = np.empty(1, np.float32)
        check(self._lib.mppi_cpu_get(self._h, _fp(u), _fp(e), _fp(cost), _fp(b), _fp(n), _fp(w)))
        return dict(u=u, e=e, cost=cost, weight=w, beta=float(b[0]), nabla=float(n[0]))
