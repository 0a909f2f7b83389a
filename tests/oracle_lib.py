"""ctypes access to the CPU oracle (oracle/*.so) for the tests, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Test infrastructure: nothing under mppi_gpu_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
fp = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(fp) if a is not None else C.cast(None, fp)


def build(force=False):
    """Compile the oracle if its libraries are missing (gcc / hipcc host-only; no GPU needed)."""
    need = force or not all(os.path.exists(os.path.join(ORACLE_DIR, n))
                            for n in ("liboracle.so", "libnoise_oracle.so"))
    if need:
        subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so", "libnoise_oracle.so"],
                       check=True, capture_output=True)
    if os.path.isdir("/root/reference/src") and (
            force or not os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref_cost.so"))):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)


_o = _n = _r = None


def oracle():
    global _o
    if _o is None:
        build()
        _o = C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so"))
        _o.orc_step_cost.restype = C.c_float
        _o.orc_final_cost.restype = C.c_float
        _o.orc_beta.restype = C.c_float
        _o.orc_nabla.restype = C.c_float
        _o.orc_nabla_tree.restype = C.c_float
        _o.orc_kat_exp_expected.restype = C.c_double
        _o.orc_kat_exp_expected.argtypes = [C.c_float, C.c_float, C.c_float]
    return _o


def noise_lib():
    global _n
    if _n is None:
        build()
        _n = C.CDLL(os.path.join(ORACLE_DIR, "libnoise_oracle.so"))
    return _n


def ref_cost_lib():
    """The reference's own cost.cu compiled here (oracle/_ref); None when it was not built."""
    global _r
    path = os.path.join(ORACLE_DIR, "_ref", "libref_cost.so")
    if _r is None and os.path.exists(path):
        _r = C.CDLL(path)
        _r.ref_step_cost.restype = C.c_float
        _r.ref_final_cost.restype = C.c_float
    return _r


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def step_cost(x, u, e, w, goal, lam, inv_s):
    x, u, e, w, goal, inv_s = map(f32, (x, u, e, w, goal, inv_s))
    return np.float32(oracle().orc_step_cost(_p(x), _p(u), _p(e), _p(w), _p(goal),
                                             C.c_float(lam), _p(inv_s), x.size, u.size))


def final_cost(x, w, goal):
    x, w, goal = map(f32, (x, w, goal))
    return np.float32(oracle().orc_final_cost(_p(x), _p(w), _p(goal), x.size))


def rollout(x0, U, E, goal, w, dt, lam=1.0, inv_s=None, want_X=False):
    E = f32(E)
    K, T, A = E.shape
    S = 2 * A
    x0, U, goal, w = f32(x0), f32(U).reshape(T, A), f32(goal), f32(w)
    inv_s = f32(np.ones(A) if inv_s is None else inv_s)
    cost = np.empty(K, np.float32)
    X = np.empty((K, T + 1, S), np.float32) if want_X else None
    oracle().orc_rollout(K, T, S, A, C.c_float(dt), _p(x0), _p(U), _p(E), _p(goal), _p(w),
                         C.c_float(lam), _p(inv_s), _p(cost), _p(X))
    return (cost, X) if want_X else cost


def solve(x0, U, E, goal, w, dt, lam=1.0, inv_s=None, f64_update=True, want_X=False):
    """One full solve on injected noise. Returns a dict; U is not modified."""
    E = f32(E)
    K, T, A = E.shape
    S = 2 * A
    x0, goal, w = f32(x0), f32(goal), f32(w)
    Uw = f32(U).reshape(T, A).copy()
    inv_s = f32(np.ones(A) if inv_s is None else inv_s)
    act = np.empty(A, np.float32)
    cost = np.empty(K, np.float32)
    wts = np.empty(K, np.float32)
    beta = C.c_float()
    nabla = C.c_float()
    X = np.empty((K, T + 1, S), np.float32) if want_X else None
    oracle().orc_solve(K, T, S, A, C.c_float(dt), _p(x0), _p(Uw), _p(E), _p(goal), _p(w),
                       C.c_float(lam), _p(inv_s), int(bool(f64_update)), _p(act), _p(cost),
                       C.byref(beta), C.byref(nabla), _p(wts), _p(X))
    out = dict(next_act=act, U=Uw, cost=cost, beta=np.float32(beta.value),
               nabla=np.float32(nabla.value), weights=wts)
    if want_X:
        out["X"] = X
    return out


def update(U, wts, E, f64=False):
    E = f32(E)
    K, T, A = E.shape
    Uw = f32(U).reshape(T, A).copy()
    wts = f32(wts)
    fn = oracle().orc_update_f64 if f64 else oracle().orc_update
    fn(_p(Uw), _p(wts), _p(E), K, T, A)
    return Uw


def noise(seed, solve_index, k_offset, K, T, A, sigma):
    sigma = f32(sigma)
    E = np.zeros((K, T, A), np.float32)
    rc = noise_lib().orc_noise_fill(C.c_ulonglong(seed), C.c_ulonglong(solve_index),
                                    C.c_ulonglong(k_offset), K, T, A, _p(sigma), _p(E))
    if rc != 0:
        raise ValueError("orc_noise_fill failed")
    return E


def noise_block_u32(seed, k_global, block_index):
    out = (C.c_uint * 4)()
    noise_lib().orc_noise_block_u32(C.c_ulonglong(seed), C.c_ulonglong(k_global),
                                    C.c_ulonglong(block_index), out)
    return np.array(list(out), dtype=np.uint32)


# problem presets of the reference's shipped configs (config/point_mass{1,2,3}d.yaml)
PRESETS = {
    1: dict(goal=[1, 0], w=[1, 5]),
    2: dict(goal=[1, 0, 0, 0], w=[1, 1, 50, 50]),
    3: dict(goal=[1, .5, .75, 0, 0, 0], w=[1, 1, 1, 5, 5, 5]),
    4: dict(goal=[1, .5, .75, .25, 0, 0, 0, 0], w=[1, 1, 1, 1, 5, 5, 5, 5]),
}


def make_case(A, K, T, seed, u_scale=0.05, sigma=0.025):
    """Seeded synthetic inputs of the shape SURVEY section 8(d) prescribes for parity runs."""
    rng = np.random.default_rng(seed)
    S = 2 * A
    x0 = (rng.standard_normal(S) * 0.1).astype(np.float32)
    U = (rng.standard_normal((T, A)) * u_scale).astype(np.float32)
    E = (rng.standard_normal((K, T, A)) * sigma).astype(np.float32)
    goal = np.array(PRESETS[A]["goal"], np.float32)
    w = np.array(PRESETS[A]["w"], np.float32)
    return dict(x0=x0, U=U, E=E, goal=goal, w=w, dt=np.float32(0.1))


# ---- emulation of the reference's update_act launch structure ---------------
# Test infrastructure: a literal restatement, block by block and tree level by tree level, of
# update_act / update_act_kernel / sum_red_adim / copy_act (reference src/point_mass.cu:384-480,
# 828-926, 668-741, 756-761) in float32, to pin which samples the reference's update covers for
# act_dim != 2.  Only shapes without a racy in-place middle pass (first-level grid < 256*A).

def _ref_tree(partial, A, B=256):
    """In-block tree of both kernels (src/point_mass.cu:893-909, 709-725): `s > 1`, not `s > 0`."""
    s = B * A // 2
    while s > 1:
        tids = np.arange(B)
        act = tids[tids * A < s]
        for j in range(A):      # threads run in lock step: read all, then write all
            add = partial[act * A + j + s].copy()
            partial[act * A + j] = (partial[act * A + j] + add).astype(np.float32)
        s >>= 1
    return partial


def _ref_update_act_kernel(w, e_flat, T, t, A, n, grid, B=256):
    v_r = np.zeros(grid * A, np.float32)
    max_size = n * A * T
    tids = np.arange(B)
    for b in range(grid):
        # shared memory holds B*A floats; the tree reads up to index B*A-1+... inside that range
        partial = np.zeros(B * A, np.float32)
        i = b * (B * 2) * A * T + t * A + tids * T * A
        shift = B * T * A
        k = b * B + tids + B * b
        for j in range(A):
            both = i + shift < max_size
            one = (~both) & (i < max_size)
            val = np.zeros(B, np.float32)
            ib, kb = i[both], k[both]
            val[both] = (w[kb] * e_flat[ib + j] + w[kb + B] * e_flat[ib + shift + j]).astype(np.float32)
            io, ko = i[one], k[one]
            val[one] = (w[ko] * e_flat[io + j]).astype(np.float32)
            partial[tids * A + j] = val
        partial = _ref_tree(partial, A, B)
        v_r[b * A:(b + 1) * A] = partial[:A]
    return v_r


def _ref_sum_red_adim(v, n, A, grid, B=256):
    out = v.copy()
    tids = np.arange(B)
    for b in range(grid):
        partial = np.zeros(B * A, np.float32)
        i = b * (B * 2) * A + tids * A
        shift = B * A
        for j in range(A):
            both = i + shift < n * A
            one = (~both) & (i < n * A)
            val = np.zeros(B, np.float32)
            val[both] = (v[i[both] + j] + v[i[both] + shift + j]).astype(np.float32)
            val[one] = v[i[one] + j]
            partial[tids * A + j] = val
        partial = _ref_tree(partial, A, B)
        out[b * A:(b + 1) * A] = partial[:A]
    return out


def ref_update_emulated(U, wts, E):
    """U + what the reference's update_act adds, by its own launch structure (float32)."""
    E = f32(E)
    K, T, A = E.shape
    B = 256
    w = np.concatenate([f32(wts), np.zeros(2 * B, np.float32)])        # tail reads stay in bounds
    e_flat = np.concatenate([E.reshape(-1), np.zeros(2 * B * T * A + A, np.float32)])
    Uw = f32(U).reshape(T, A).copy()
    grid1 = K // (B * A) + 1
    assert grid1 < B * A, "shape has a racy in-place middle pass in the reference: not emulated"
    for t in range(T):
        v_r = _ref_update_act_kernel(w, e_flat, T, t, A, K, grid1, B)
        if grid1 > 1:
            n = grid1
            v_pad = np.concatenate([v_r, np.zeros(2 * B * A, np.float32)])
            v_r = _ref_sum_red_adim(v_pad, n, A, 1, B)
        Uw[t] = (Uw[t] + v_r[:A]).astype(np.float32)
    return Uw


def ref_update_mask(K, A):
    """Samples the reference's update sums (what ref_update_emulated reproduces): all for
    act_dim 2; the first 512*(K/768+1) for act_dim 3; k even and -- with a second level, K >= 256
    -- (k/512) even for act_dim 1."""
    k = np.arange(K)
    if A == 3:
        return k < min(K, 512 * (K // 768 + 1))
    if A == 1:
        m = (k % 2) == 0
        return m & ((k // 512) % 2 == 0) if K >= 256 else m
    assert A == 2, "the reference has no 4-D system; its trees would also mix axes there"
    return np.ones(K, bool)
