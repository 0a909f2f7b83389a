#!/usr/bin/env python3
"""bench.py -- rollouts/s of one full MPPI solve (sample + rollout + cost + beta/nabla + weighted
update + shift) on synthetic point-mass data, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4] [--chunks C]

A "step" is one MPPI solve over one batch of rollouts.  Default workload = BASELINE configs[1]:
point_mass2d, K = 1e4 rollouts, T = 200 steps, per GPU (weak scaling: with N GPUs the global
batch is N*K, sharded by sample; the only exchange is T*A+2 floats per rank and solve, written by
the combine kernel straight into the peers' inboxes over xGMI, or an RCCL all-gather: --transport).
Inputs are resident on the device before the timed region; the timed region is bracketed by a
barrier + torch.cuda.synchronize() on both sides; value = N*K*steps / max-over-ranks time.

The JSON line also carries
  roofline      the rollout kernel (dominant): algorithmic HBM bytes per launch / its mean
                duration from HIP events recorded on the launch stream inside the timed region;
                `traffic` (PMC HBM bytes) and `alu` (VALU issue-slot utilisation from the PMC
                instruction counters) come from the committed rocprofv3 summaries of the same
                command, and say so in `traffic_source` / `alu.source`
  latency       the reference's timed unit (src/main.cu:329-332): blocking get_act + set_x,
                measured after the timed region in the same process
  cpu_baseline  the product's own serial CPU controller (ControllerBase through mppi_cpu_*, the
                repo's "serial CPU path") on this host, 1 thread, on a bounded sample of the same
                workload; `all_cores` the same with os.cpu_count() threads; `oracle` the CPU oracle
                (oracle/mppi_oracle.c + rocRAND-host sampler) as a cross-check
  extra.c4      with --gpus N > 1: config 4's strong-scaling leg (K = 1e6 global, K/N per rank)
                timed after the headline region

With --gpus N > 1 and no WORLD_SIZE in the environment, bench.py starts its N rank processes
itself (one per GPU) before anything touches a GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (A, K per GPU, T, description)
    "c2": (2, 10_000, 200, "point_mass2d K=1e4 T=200 (BASELINE configs[1])"),
    "c3": (3, 100_000, 200, "point_mass3d K=1e5 T=200 (BASELINE configs[2])"),
    "c4": (3, 125_000, 200, "point_mass3d K=1e6/8 per GPU T=200 (BASELINE configs[3] shard)"),
    "c1": (1, 100, 50, "point_mass1d K=100 T=50 (BASELINE configs[0] shape, on the GPU)"),
    "floor": (2, 10_000, 8, "launch-floor probe: 2-D K=1e4 T=8 (not a BASELINE config)"),
    # sweep points between config 3 and config 4 (where the noise no longer fits the 256 MB MALL)
    "c3x2": (3, 200_000, 200, "point_mass3d K=2e5 T=200 (sweep point, not a BASELINE config)"),
    "c3x4": (3, 400_000, 200, "point_mass3d K=4e5 T=200 (sweep point, not a BASELINE config)"),
    # the reference's SHIPPED configs (config/point_mass{1,2,3}d.yaml: samples 3000, horizon 50)
    "s1": (1, 3000, 50, "point_mass1d.yaml as shipped: K=3000 T=50"),
    "s2": (2, 3000, 50, "point_mass2d.yaml as shipped: K=3000 T=50"),
    "s3": (3, 3000, 50, "point_mass3d.yaml as shipped: K=3000 T=50"),
    "c4full": (3, 1_000_000, 200, "point_mass3d K=1e6 T=200 on ONE GPU (BASELINE configs[3] unsharded)"),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable
PROFILE_TRAFFIC = "traffic_r03.json"   # committed rocprofv3 --pmc summaries (profiles/README.md)
PROFILE_ALU = "alu_r03.json"


def algorithmic_bytes_rollout(K, T, A):
    """Bytes the rollout launch must move for K rollouts in this engine's dataflow:
    one E store (4*T*A) + one cost store (4) per rollout.  SURVEY section 8(d)'s per-rollout
    figure 2*4*T*A + 2R + 16 minus what the design removes: the E re-load (update fused into
    the rollout), the RNG state (R = 0, counter-based Philox) and the three cost re-loads."""
    return K * (4 * T * A + 4)


# goal / weights of the reference's YAML files (config/point_mass2d.yaml:6-16,
# config/point_mass3d.yaml:6-20); 1-D and 4-D follow the same pattern
PRESETS = {
    1: ([1, 0], [1, 5]),
    2: ([1, 0, 0, 0], [1, 1, 50, 50]),
    3: ([1, .5, .75, 0, 0, 0], [1, 1, 1, 5, 5, 5]),
    4: ([1, .5, .75, .25, 0, 0, 0, 0], [1, 1, 1, 1, 5, 5, 5, 5]),
}


def make_inputs(A, T):
    """Synthetic inputs of the benchmark: x0 ~ 0.1 N(0,1) (seed 0), U0 = 0, the YAML goal / w,
    dt = 0.1.  (The same values tests/oracle_lib.make_case(A, 1, T, seed=0, u_scale=0) produces;
    written out here so that the GPU leg imports nothing of the oracle.)"""
    import numpy as np
    rng = np.random.default_rng(0)
    x0 = (rng.standard_normal(2 * A) * 0.1).astype(np.float32)
    return {"x0": x0, "U": np.zeros((T, A), np.float32),
            "goal": np.array(PRESETS[A][0], np.float32), "w": np.array(PRESETS[A][1], np.float32),
            "dt": np.float32(0.1)}


def cpu_baseline_oracle(A, K, T, budget_s=6.0):
    """Serial oracle, full solve incl. its own sampler, same K/T; returns rollouts/s.  The ONLY
    place where bench.py touches the oracle (a cross-check of the product's CPU controller)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    c = make_inputs(A, T)
    sig = [0.025] * A
    U = c["U"].copy()
    n = 0
    t_tot = 0.0
    while t_tot < budget_s and n < 200:
        t0 = time.perf_counter()
        E = ol.noise(0, n, 0, K, T, A, sig)
        out = ol.solve(c["x0"], U, E, c["goal"], c["w"], c["dt"], f64_update=False)
        t_tot += time.perf_counter() - t0
        U = out["U"]
        n += 1
    assert np.isfinite(U).all()
    return K * n / t_tot, n


def cpu_baseline_controller(A, K, T, threads, budget_s):
    """The product's serial CPU path: ControllerBase (mppi_gpu_amd/csrc/controller_base.cpp: the
    reference's class of that name made real) through the C ABI mppi_cpu_*,
    full solves incl. its Philox + Box-Muller sampler on `threads` host threads."""
    import numpy as np
    from mppi_gpu_amd import ControllerBase
    c = make_inputs(A, T)
    ctl = ControllerBase(K, T, float(c["dt"]), 2 * A, A)
    ctl.setActions(c["U"])
    ctl.setCost(c["goal"], c["w"])
    ctl.setSeed(0)
    ctl.setThreads(threads)
    n = 0
    t_tot = 0.0
    act = None
    while t_tot < budget_s and n < 400:
        t0 = time.perf_counter()
        act = ctl.next(c["x0"])
        t_tot += time.perf_counter() - t0
        n += 1
    assert np.isfinite(act).all()
    ctl.close()
    return K * n / t_tot, n


def measured_hbm_peak(torch, n_bytes=1 << 30, reps=12):
    """the peak a plain device copy reaches on THIS box, measured in-run with
    torch events on torch's current stream (1 GiB read + 1 GiB written per copy, best of `reps`),
    and the same for a pure write (fill), since the rollout launch only stores."""
    src = torch.empty(n_bytes // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    best_c = best_f = 0.0
    for _ in range(3):
        dst.copy_(src); dst.zero_()
    for _ in range(reps):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); dst.copy_(src); b.record(); dst.zero_(); c.record()
        c.synchronize()
        best_c = max(best_c, 2 * n_bytes / (a.elapsed_time(b) * 1e-3) / 1e9)
        best_f = max(best_f, n_bytes / (b.elapsed_time(c) * 1e-3) / 1e9)
    del src, dst
    torch.cuda.empty_cache()
    return {"copy_GBs": round(best_c, 1), "fill_GBs": round(best_f, 1),
            "what": "torch device copy / zero-fill of 1 GiB, best of %d, bytes read + written" % reps}


def ess_of(weights):
    """Effective sample size 1 / sum(w^2) of normalised weights (float64)."""
    import numpy as np
    w = np.asarray(weights, np.float64)
    tot = w.sum()
    if not tot > 0:
        return 0.0
    w = w / tot
    return float(1.0 / np.sum(w * w))


def lambda_for_ess(cost, target):
    """lambda at which the softmax of -cost / lambda has the effective sample size `target`
    (bisection in log lambda on the path costs the engine reported; chooses a workload, checks
    nothing: the ESS a run ACHIEVED is read from its own weights afterwards)."""
    import numpy as np
    c = np.asarray(cost, np.float64)
    c = c - c.min()

    def ess(lam):
        w = np.exp(-c / lam)
        return float(w.sum() ** 2 / np.sum(w * w))

    lo, hi = 1e-2, 1e7
    for _ in range(60):
        mid = float(np.sqrt(lo * hi))
        lo, hi = (mid, hi) if ess(mid) < target else (lo, mid)
    return float(np.sqrt(lo * hi))


def spread_weights_leg(m, K, n_warm, n):
    """The same engine and workload with lambda chosen so that about a third of the batch carries
    weight (the reference hard-codes lambda = 1, src/point_mass.cu:53-54).  The kernels run the
    same instructions whatever the weights: the time must not depend on lambda."""
    cost = m.get_inf(x=False, u=False, e=False, beta=False, nabla=False, weight=False)["cost"]
    lam = lambda_for_ess(cost, K / 3.0)
    m.set_params(lam)
    dt_v, k_v, k_nv, c_v = timed_engine_run(m, n_warm, n)
    w = m.get_inf(x=False, u=False, e=False, cost=False, beta=False, nabla=False)["weight"]
    m.set_params(1.0)
    return {"what": "the same workload with lambda chosen for an effective sample size near K/3; "
                    "lambda = 1 is the reference's hard-coded value",
            "lambda": lam, "ess": round(ess_of(w), 1), "ms_per_step": dt_v * 1e3,
            "value": K / dt_v, "rollout_kernel_ms": round(k_v, 5)}


def pmc_child_passes(workload, kernel_sub, counter_sets, extra_args=()):
    """Run this file as a child under `rocprofv3 --kernel-trace --pmc <set>` once per counter set
    (counters in their own runs, --kernel-trace only) and return, for the kernel whose name contains
    `kernel_sub`: ({counter: mean value per dispatch}, mean dispatch duration in us, dispatches)
    or (None, reason, 0)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", 0
    vals, durs = {}, []
    tmp = tempfile.mkdtemp(prefix="mppi_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for n, cset in enumerate(counter_sets):
            d = os.path.join(tmp, f"pass{n}")
            cmd = [exe, "--kernel-trace", "--pmc", *cset, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", "20",
                   "--warmup", "5", "--ramp-ms", "0", "--no-cpu-baseline", "--no-events", "--no-pmc",
                   "--no-extra", "--no-latency", *extra_args]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=180)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {' '.join(cset)} failed: {r.stderr.strip()[-200:]}", 0
            for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    if kernel_sub in row["Kernel_Name"]:
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
                for row in csv.DictReader(open(f)):
                    if kernel_sub in row["Kernel_Name"]:
                        durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    except (subprocess.TimeoutExpired, OSError) as ex:
        return None, f"child profiling run: {ex}", 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    want = [c for cset in counter_sets for c in cset]
    missing = [c for c in want if c not in vals]
    if missing:
        return None, f"no rows for {missing} of a kernel named *{kernel_sub}*", 0
    n = min(len(v) for v in vals.values())
    return {c: sum(v) / len(v) for c, v in vals.items()}, (sum(durs) / len(durs) if durs else 0.0), n


def pmc_traffic_live(workload, kernel_sub, extra_args=()):
    """HBM bytes per launch, measured NOW (guides/MI355X_MICROARCH.md "HBM"): FETCH_SIZE and
    WRITE_SIZE are in KiB, one TCC counter per pass; gfx950 counts the bytes of a wide coalesced
    read at half -> FETCH_SIZE x 2; WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
    v, why, n = pmc_child_passes(workload, kernel_sub, [["FETCH_SIZE"], ["WRITE_SIZE"]], extra_args)
    if v is None:
        return None, why
    fetch, write = v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
    return {"hbm_bytes_per_launch": round(2 * fetch + write), "fetch_bytes_raw": round(fetch),
            "fetch_bytes_corrected_x2": round(2 * fetch), "write_bytes": round(write),
            "dispatches": n}, None


def rocprof_kernel_ms_live(workload, kernel_sub, extra_args=(), steps=2000):
    """Mean duration of the kernel by `rocprofv3 --kernel-trace --stats` over a child run of this
    command (no counters): what the committed profiles/*_kernel_stats.csv hold, measured NOW.
    `steps` is chosen by the caller so that the run lasts ~0.25 s: the mean of --stats covers EVERY
    dispatch of the process, and the first hundred of a process run at rising clocks (C3: 83 us
    against 65 at steady state, tools/kt_gaps.sh) -- a short profiled run reads 5-7 % slow.
    (ms, calls) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="mppi_kt_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "--",
               sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps),
               "--warmup", "20", "--no-cpu-baseline", "--no-events", "--no-pmc", "--no-extra",
               "--no-latency", *extra_args]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True,
                           text=True, timeout=180)
        if r.returncode != 0:
            return None, f"rocprofv3 --stats failed: {r.stderr.strip()[-200:]}"
        for f in glob.glob(os.path.join(tmp, "*", "*kernel_stats.csv")):
            for row in csv.DictReader(open(f)):
                if kernel_sub in row["Name"]:
                    return float(row["AverageNs"]) * 1e-6, int(row["Calls"])
        return None, f"no kernel named *{kernel_sub}* in the stats"
    except (subprocess.TimeoutExpired, OSError) as ex:
        return None, f"child profiling run: {ex}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# issue cost of a wave64 instruction in SIMD cycles at the nominal 2.4 GHz, measured on MI355X with
# tools/ubench_issue (profiles/r02_ubench_issue.txt: transcendentals, plain fp32) and
# tools/ubench_int (profiles/r03_ubench_int.txt: v_mad_u64_u32 alone 4.5-4.9 -- round 2's 6.3 had
# the move that widens its addend in it --, v_bitop3_b32 with its SGPR key 4.25)
ISSUE_COST = {"INT64": 4.7, "TRANS_F32": 8.4, "BITOP3": 4.25, "OTHER": 2.3}


def pmc_alu_live(workload, kernel_sub, extra_args=(), kernel_ms=None):
    """The VALU side of the roofline, measured NOW: instruction counts by class, the counter-based
    VALU busy (rocprof's VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE) and the
    issue-cycle model (count x measured issue cycles per class / SIMD cycles of the launch)."""
    sets = [["SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_FMA_F32",
             "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32"],
            ["SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"], ["GRBM_GUI_ACTIVE"]]
    v, dur_us, n = pmc_child_passes(workload, kernel_sub, sets, extra_args)
    if v is None:
        return None, dur_us
    # the issue-cycle model is priced against the STEADY-STATE duration of the launch (this run's
    # event stamps): the 25 dispatches of a counter pass run at the clocks of a cold process
    model_us = kernel_ms * 1e3 if kernel_ms else dur_us
    n_simd, f_nom = 1024, 2.4e9
    n_all, n64, ntr = v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU_INT64"], v["SQ_INSTS_VALU_TRANS_F32"]
    nb3 = n64                       # one v_bitop3_b32 per v_mad_u64_u32 (two of each per Philox round)
    cyc = (n64 * ISSUE_COST["INT64"] + ntr * ISSUE_COST["TRANS_F32"] + nb3 * ISSUE_COST["BITOP3"]
           + (n_all - n64 - ntr - nb3) * ISSUE_COST["OTHER"])
    gui = v["GRBM_GUI_ACTIVE"]
    inst = 8.0 if dur_us > 0 and gui / (dur_us * 1e-6) > 6e9 else 1.0    # summed over the 8 XCDs, or not
    gui_cycles = gui / inst
    return {"bound": "valu", "achieved": round(cyc / (n_simd * model_us * 1e-6 * f_nom), 4), "peak": 1.0,
            "unit": "fraction of the launch's VALU issue cycles (issue-cycle model)",
            "valu_busy_counter": round(v["SQ_ACTIVE_INST_VALU"] * 4.0 / n_simd / gui_cycles, 4),
            "valu_instructions_per_launch": round(n_all), "v_mad_u64_u32": round(n64),
            "transcendental": round(ntr), "fma_f32": round(v["SQ_INSTS_VALU_FMA_F32"]),
            "add_f32": round(v["SQ_INSTS_VALU_ADD_F32"]), "mul_f32": round(v["SQ_INSTS_VALU_MUL_F32"]),
            "wait_any_fraction_of_wave_cycles": round(v["SQ_WAIT_ANY"] / max(1.0, v["SQ_WAVE_CYCLES"]), 4),
            "kernel_us_in_these_passes": round(dur_us, 3), "kernel_us_model": round(model_us, 3),
            "dispatches": n,
            "source": "measured in this run: three child passes of this command under rocprofv3 "
                      "--kernel-trace --pmc (SQ instruction counters; SQ_ACTIVE_INST_VALU; "
                      "GRBM_GUI_ACTIVE), per-class issue costs from profiles/r02_ubench_issue.txt "
                      "and profiles/r03_ubench_int.txt"}, None


def kernel_symbol(geo, riding, A):
    """Name (up to the first template argument) of the rollout kernel a geometry launches."""
    if geo["strict"]:
        return f"k_rollout_stream<{A}"
    if geo["packed"]:
        return (f"k_rollout_packed_ride<{A}" if riding else f"k_rollout_packed<{A}")
    return f"k_rollout_ride<{A}" if riding else f"k_rollout_fused<{A}"


def under_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def attach_live_traffic(roof, workload, kernel_sub, extra_args=()):
    """Replace roof['traffic'] (from the committed summaries) with a live PMC measurement."""
    if roof is None:
        return
    t, why = pmc_traffic_live(workload, kernel_sub, extra_args)
    if t is None:
        roof["traffic_live_error"] = why
        return
    roof["traffic"] = t["hbm_bytes_per_launch"]
    roof["traffic_detail"] = t
    roof["traffic_over_algorithmic"] = round(t["hbm_bytes_per_launch"]
                                             / roof["algorithmic_bytes_per_launch"], 4)
    roof["traffic_source"] = ("measured in this run: two child passes of this command under rocprofv3 "
                              "--kernel-trace --pmc (FETCH_SIZE x2 + WRITE_SIZE, one counter per pass), "
                              "mean over %d dispatches of %s" % (t["dispatches"], kernel_sub))
    a, why = pmc_alu_live(workload, kernel_sub, extra_args, kernel_ms=roof.get("kernel_ms"))
    if a is None:
        roof["alu_live_error"] = why
    else:
        roof["alu"] = a
    k_ms_ev = roof.get("kernel_ms") or 0.05
    km, calls = rocprof_kernel_ms_live(workload, kernel_sub, extra_args,
                                       steps=int(min(20000, max(400, 250.0 / k_ms_ev))))
    if km is None:
        roof["kernel_ms_rocprof_error"] = calls
    else:
        roof["kernel_ms_rocprof"] = round(km, 6)
        roof["frac_rocprof"] = round(roof["algorithmic_bytes_per_launch"] / (km * 1e-3) / 1e9
                                     / roof["peak"], 4)
        roof["kernel_ms_rocprof_source"] = ("measured in this run: child pass under rocprofv3 "
                                            "--kernel-trace --stats, mean of %d dispatches" % calls)


def quantiles(xs):
    xs = sorted(xs)
    n = len(xs)
    return {"median": xs[n // 2], "p10": xs[n // 10], "p90": xs[(n * 9) // 10], "n": n}


def committed_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def roofline_entry(workload, K, T, A, geo, riding, k_ms, k_n, c_ms, solve_ms=None):
    """The `roofline` object of one workload from the live event timing of its rollout launches.
    PMC counters cannot be collected from inside the timed process: `traffic`, the rocprofv3
    kernel duration and the VALU figures come from the COMMITTED summaries of this workload and
    geometry (tools/traffic.sh, tools/kt.sh, tools/alu.sh -> profiles/), and are labelled so."""
    kind = "ride" if riding else "plain"
    shape = f"packed{geo['groups_per_lane']}" if geo["packed"] else f"chunks={geo['chunks']}"
    prof_key = f"{workload}:{shape}:{kind}"
    ab = algorithmic_bytes_rollout(K, T, A)
    ach = ab / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic = traffic_src = None
    tj = committed_profile(PROFILE_TRAFFIC)
    if tj and not geo["strict"]:
        ent = tj["entries"].get(prof_key)
        if ent:
            traffic = ent["hbm_bytes_per_launch"]
            traffic_src = (f"committed profile profiles/{PROFILE_TRAFFIC} (rocprofv3 --pmc "
                           "FETCH_SIZE x2 + WRITE_SIZE, separate passes, same command); not "
                           "measured in this run")
    base = "k_rollout_packed" if geo["packed"] else "k_rollout_fused"
    kname = ("k_rollout_stream" if geo["strict"] else
             (base + "_ride" if geo["packed"] else "k_rollout_ride")
             + " (rollout of solve j + combine of solve j-1 in one launch)" if riding
             else base)
    roof = {"bound": "hbm", "kernel": kname,
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": ab, "kernel_ms": round(k_ms, 5),
            "kernel_ms_source": "HIP events stamped on the dispatch (hipExtLaunchKernelGGL), "
                                "second launch of each stamped pair, inside the timed region",
            "launches_timed": k_n, "combine_kernel_ms": round(c_ms, 5)}
    if solve_ms:
        # the same algorithmic bytes over the WHOLE solve (wall time per step of the timed region:
        # rollout + combine + whatever lies between the launches), what north_star's ">= 40 % of
        # the HBM roofline" at rollouts/s of the solve is stated on
        roof["solve_frac"] = round(ab / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roof["solve_frac_what"] = ("algorithmic_bytes_per_launch / ms_per_step / peak: the solve, "
                                   "not the kernel")
    aj = committed_profile(PROFILE_ALU)
    if aj and not geo["strict"]:
        ent = aj["entries"].get(prof_key)
        if ent:
            # rocprofv3's own average duration of the same kernel (profiled runs clock a few per
            # cent lower than this un-profiled one) and the fraction it gives
            if ent.get("kernel_ms_rocprof"):
                roof["kernel_ms_rocprof"] = ent["kernel_ms_rocprof"]
                roof["frac_rocprof"] = round(ab / (ent["kernel_ms_rocprof"] * 1e-3) / 1e9
                                             / HBM_PEAK_GBS, 4)
            roof["alu"] = dict(ent["alu"], source=f"committed profile profiles/{PROFILE_ALU} "
                               "(rocprofv3 --pmc SQ instruction counters, same command); not "
                               "measured in this run")
    return roof


def timed_engine_run(m, n_warm, n, every=32, ramp_ms=30.0):
    """n solves enqueued back to back on engine m: (seconds per solve, rollout kernel ms, launches
    stamped, combine kernel ms).  Every leg first keeps the GPU busy for ramp_ms: whatever came
    before it (creating an engine, reading costs back, a host-side bisection) left the device idle
    for milliseconds, and a leg measured on the way back up from the idle clocks reads 10-20 %
    slow (C3: 79 against 66.7 us per solve, tools/lambda_speed.py)."""
    t_r = time.perf_counter()
    while time.perf_counter() - t_r < ramp_ms * 1e-3:
        for _ in range(20):
            m.solve_async()
        m.sync_act()
    for _ in range(n_warm):
        m.solve_async()
    m.sync_act()
    m.set_profiling(every)
    t0 = time.perf_counter()
    for _ in range(n):
        m.solve_async()
    m.sync_act()
    dt = (time.perf_counter() - t0) / n
    k_ms, k_n = m.kernel_ms(0)
    c_ms, _ = m.kernel_ms(1)
    m.set_profiling(0)
    return dt, k_ms, k_n, c_ms


def spawn_ranks(n_gpus):
    """--gpus N without a launcher: start the N rank processes (one per GPU) before anything here
    touches a GPU, pass their output through and exit with the worst of their codes."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll: a rank that dies would leave the others waiting in a collective until its time-out
    codes = [None] * n_gpus
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):           # exactly the processes started above
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    sys.exit(max(abs(c) for c in codes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--ramp-ms", type=float, default=30.0,
                    help="untimed solves for this long before the W warm-up steps, so that short "
                         "runs are not measured at idle clocks (0 = off)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--packing", type=int, default=0,
                    help="mppi_set_packing: 0 auto, -1 row-aligned kernel only, n packed with n groups per lane")
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--strict", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel events")
    ap.add_argument("--pipeline", type=int, default=0, choices=(0, 1),
                    help="mppi_set_pipeline mode: 0 deferred combine (default: back-to-back solves "
                         "are one launch each while launches are short), 1 eager (rollout + "
                         "combine launch per solve)")
    ap.add_argument("--blocking", action="store_true",
                    help="analysis: every step is a blocking get_act (the closed-loop call), not an "
                         "asynchronous solve")
    ap.add_argument("--transport", choices=("auto", "direct", "collective"), default="auto",
                    help="rank-partial exchange of the sharded solve: direct peer stores, RCCL "
                         "all-gather, or auto (direct if it validates against the collective)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="analysis: all ranks share cuda:0 and torch.distributed runs over gloo "
                         "(RCCL wants one GPU per rank); exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (local solve, RCCL all-gather, finish) even "
                         "with one rank: rehearsal of the N > 1 path on a one-GPU box")
    ap.add_argument("--inject", action="store_true",
                    help="ANALYSIS ONLY: injected-noise mode (no sampling); not a valid bench result")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic live (child rocprofv3 --pmc passes); the "
                         "committed summary under profiles/ is quoted instead")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only (no extra.* legs)")
    ap.add_argument("--no-latency", action="store_true", help="skip the blocking get_act latency leg")
    ap.add_argument("--event-every", type=int, default=32,
                    help="record HIP events around the kernels of every n-th timed solve")
    args = ap.parse_args()

    import numpy as np
    import torch
    from mppi_gpu_amd import PointMassModel

    N = args.gpus
    if N > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(N)                  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != N:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU fallback"
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if N > 1 or args.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    A, K, T, desc = WORKLOADS[args.workload]
    c = make_inputs(A, T)                               # x0 ~ 0.1 N(0,1), U0 = 0, yaml goal/w

    if N == 1 and not args.force_sharded:
        m = PointMassModel(K, T, float(c["dt"]), 2 * A, A)
        sharded = None
    else:
        from mppi_gpu_amd.sharded import ShardedPointMassModel
        sharded = ShardedPointMassModel(N * K, T, float(c["dt"]), 2 * A, A,
                                        transport=args.transport)
        m = sharded.engine          # this rank's shard: samples [rank*K, (rank+1)*K)
    m.set_pipeline(args.pipeline)
    m.set_tuning(chunks=args.chunks, strict=args.strict, max_blocks=args.max_blocks)
    if args.packing:
        m.set_packing(args.packing)
    m.set_seed(0)
    (sharded or m).memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
    geo = m.geometry()
    if args.inject:
        m.set_noise(np.zeros((K, T, A), np.float32))
    step = m.solve_async if sharded is None else sharded.solve_async
    if args.blocking:
        step = m.get_act if sharded is None else sharded.get_act

    def fence():
        m.flush_async()       # nothing held back: a deferred combine is on its stream before the waits
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        m.sync_act()          # the engine's own stream (same HIP runtime as torch's: see _capi)

    if args.ramp_ms > 0:                       # clock ramp; every rank takes the same decisions
        t_r = time.perf_counter()
        for _batch in range(2000):
            for _ in range(50):
                step()
            m.sync_act()
            go = time.perf_counter() - t_r < args.ramp_ms * 1e-3
            if dist is not None:
                gt = torch.tensor([1 if go else 0], dtype=torch.int32,
                                  device="cpu" if args.rehearse_one_gpu else "cuda")
                dist.all_reduce(gt, op=dist.ReduceOp.MIN)
                go = bool(gt.item())
            if not go:
                break
    for _ in range(args.warmup):
        step()
    fence()
    # stamped launches: every n-th solve and its successor.  Stamping costs ~7 us of host and
    # dispatch time per stamped launch (hipExtLaunchKernelGGL), so the timed region is stamped
    # sparsely whatever its length; a SHORT run (the driver's --steps 20 holds one stamped pair) is
    # followed by an untimed profiling batch, see below
    event_every = max(2, args.event_every)
    if not args.no_events:
        m.set_profiling(event_every)
    fence()
    cnt0 = m.launch_counts()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_s = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt_s], device="cpu" if args.rehearse_one_gpu else "cuda",
                          dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = float(tt.item())
    act = m.sync_act()
    assert np.isfinite(act).all(), "non-finite action"

    roof = None
    # mode 0 lets the combine ride in the next rollout launch while launches are short (at most two
    # tiles per block) and all their blocks fit the chip at once (engine.hip enqueue_rollout); the
    # engine's own launch counters say what happened in the timed region
    cnt1 = m.launch_counts()
    riding = cnt1["riding"] - cnt0["riding"] > (cnt1["rollout"] - cnt0["rollout"]) // 2
    if not args.no_events:
        k_ms, k_n = m.kernel_ms(0)
        c_ms, _ = m.kernel_ms(1)
        k_n_region = k_n
        if k_n < 16 and not args.blocking:
            # too few stamped launches inside the timed region for a mean: the SAME engine goes on
            # for an untimed batch of 160 solves with every 4th pair stamped (the stamps of the
            # timed region stay in the mean)
            m.set_profiling(0)
            before = (k_ms * k_n, k_n)
            m.set_profiling(4)
            for _ in range(160):
                step()
            fence()
            kb_ms, kb_n = m.kernel_ms(0)
            c_ms, _ = m.kernel_ms(1)
            k_n = before[1] + kb_n
            k_ms = (before[0] + kb_ms * kb_n) / max(1, k_n)
        roof = roofline_entry(args.workload, K, T, A, geo, riding, k_ms, k_n, c_ms,
                              solve_ms=dt_s / args.steps * 1e3)
        roof["event_every"] = event_every
        roof["launches_timed_in_region"] = k_n_region
        if k_n != k_n_region:
            roof["kernel_ms_source"] += ("; the timed region held %d stamped launch(es), the other "
                                         "%d come from an untimed batch of 160 solves run right "
                                         "after it on the same engine" % (k_n_region, k_n - k_n_region))
        if riding and sharded is None:
            # for reference, outside the timed region: the two kernels on their own (eager mode,
            # one rollout launch + one combine launch per solve)
            m.set_pipeline(1)
            m.set_profiling(8)
            for _ in range(512):
                m.solve_async()
            m.sync_act()
            r_ms, _ = m.kernel_ms(0)
            c2_ms, _ = m.kernel_ms(1)
            m.set_profiling(0)
            m.set_pipeline(0)
            ab = algorithmic_bytes_rollout(K, T, A)
            roof["solo"] = {"rollout_kernel_ms": round(r_ms, 5), "combine_kernel_ms": round(c2_ms, 5),
                            "rollout_frac": round(ab / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                            if r_ms > 0 else None}
    m.set_profiling(0)
    if roof is not None and not args.rehearse_one_gpu:      # (every rank, on its own GPU: the ranks stay in step)
        pk = measured_hbm_peak(torch)
        roof["peak_measured"] = pk
        if pk["copy_GBs"] > 0:
            roof["frac_of_measured_copy_peak"] = round(roof["achieved"] / pk["copy_GBs"], 4)
            roof["frac_of_measured_fill_peak"] = round(roof["achieved"] / pk["fill_GBs"], 4)

    # ---- the reference's timed unit: a blocking get_act (+ set_x: the closed loop) -----------
    latency = None
    if not args.blocking and not args.inject and not args.no_latency:
        eng = sharded or m
        n_lat = 300 if K <= 200_000 else 50
        x_now = c["x0"].copy()
        t_r = time.perf_counter()          # (the legs before this one left the device idle: ramp)
        n_r = 0
        # (ranks of a sharded solve must make the same calls: a fixed count there)
        while n_r < 20 or (dist is None and time.perf_counter() - t_r < 0.03 and n_r < 5000):
            eng.get_act()
            n_r += 1
        fence()
        per_call = []
        tl = time.perf_counter()
        for _ in range(n_lat):
            t1 = time.perf_counter()
            eng.get_act()
            per_call.append(time.perf_counter() - t1)
        t_get = (time.perf_counter() - tl) / n_lat
        tl = time.perf_counter()
        for _ in range(n_lat):
            eng.get_act()
            m.set_x(x_now)
        t_loop = (time.perf_counter() - tl) / n_lat
        # median-of->=100 statistic of the back-to-back mode: 100 batches of 20
        # solves, each batch timed to its own wait
        per_batch = []
        step_b = m.solve_async if sharded is None else sharded.solve_async
        for _ in range(100):
            t1 = time.perf_counter()
            for _ in range(20):
                step_b()
            m.sync_act()
            per_batch.append((time.perf_counter() - t1) / 20)
        # the closed loop with a plant in it (reference src/main.cu:326-374: get_act -> simulate ->
        # set_x): the host is away for 20 us between two calls; the time of the calls alone, with
        # the next solve's noise drawn behind the combine meanwhile (mppi_set_noise_prefetch, the
        # default) and without
        plant = None
        if sharded is None and hasattr(m, "set_noise_prefetch"):
            plant = {"plant_step_us": 20.0}
            for name, mode in (("prefetch_off_ms", 0), ("prefetch_auto_ms", 1)):
                m.set_noise_prefetch(mode)
                acc = 0.0
                for i in range(n_lat + 20):
                    t1 = time.perf_counter()
                    m.get_act()
                    t2 = time.perf_counter()
                    if i >= 20:
                        acc += t2 - t1
                    m.set_x(x_now)
                    while time.perf_counter() - t2 < 20e-6:
                        pass
                plant[name] = round(acc / n_lat * 1e3, 5)
            plant["prefetch_counts"] = m.prefetch_counts()
            plant["what"] = ("get_act alone, the host busy for plant_step_us between two calls; "
                             "tools/latency_probe measures the same through the C ABI without Python")
        if dist is not None:
            tt = torch.tensor([t_get, t_loop], device="cpu" if args.rehearse_one_gpu else "cuda",
                              dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_get, t_loop = (float(v) for v in tt.tolist())
        latency = {"blocking_get_act_ms": round(t_get * 1e3, 5),
                   "blocking_get_act_plus_set_x_ms": round(t_loop * 1e3, 5),
                   "rollouts_per_s_blocking": N * K / t_get, "calls": n_lat,
                   "blocking_get_act_ms_quantiles": {k: (round(v * 1e3, 5) if k != "n" else v)
                                                     for k, v in quantiles(per_call).items()},
                   "back_to_back_ms_per_solve_quantiles": dict(
                       {k: (round(v * 1e3, 5) if k != "n" else v)
                        for k, v in quantiles(per_batch).items()},
                       what="100 batches of 20 solve_async + one wait, per-solve time of each batch"),
                   "what": "PointMassModel.get_act() through the Python binding: launch, solve, "
                           "wait for the action in host memory (reference src/main.cu:329-332)"}
        if plant is not None:
            latency["closed_loop_with_plant_step"] = plant

    # effective sample size of the benchmark's last solve (lambda = 1, the reference's value): says
    # in which regime the exp-weighted update ran
    ess_line = None
    if not args.inject:
        try:
            ess_line = round(ess_of(m.get_inf(x=False, u=False, e=False, cost=False, beta=False,
                                              nabla=False)["weight"]), 1)
        except Exception:
            ess_line = None
    extra = {}
    if N == 1 and sharded is None and not (args.blocking or args.inject or args.strict
                                           or args.no_events or args.no_extra):
        extra["spread_weights"] = spread_weights_leg(m, K, 100, min(args.steps, 500))
        # ---- the same workload with the noise NOT materialised (mppi_set_noise_store(0)): a
        #      reported variant, never the headline (E is an observable of the reference) --------
        m.set_noise_store(False)
        dt_v, k_v, k_nv, _ = timed_engine_run(m, 100, min(args.steps, 500))
        m.set_noise_store(True)
        extra["noise_not_materialised"] = {
            "what": "mppi_set_noise_store(0): the rollout stores path costs and partial sums only; "
                    "get_inf regenerates the noise bit for bit from the Philox counters (tested)",
            "ms_per_step": dt_v * 1e3, "value": K / dt_v, "rollout_kernel_ms": round(k_v, 5),
            "hbm_bytes_per_launch_algorithmic": 4 * K,
            "bound": "valu (no HBM stream left: see roofline.alu of the storing kernel)"}
        # ---- BASELINE configs[2] (point_mass3d, K = 1e5) in the same run: the long-launch regime,
        #      where the rollout kernel's roofline fraction is the meaningful one ------------------
        if args.workload == "c2":
            A3, K3, T3, desc3 = WORKLOADS["c3"]
            c3 = make_inputs(A3, T3)
            m3 = PointMassModel(K3, T3, float(c3["dt"]), 2 * A3, A3)
            m3.set_seed(0)
            m3.memcpy_set_data(c3["x0"], c3["U"], c3["goal"], c3["w"])
            geo3 = m3.geometry()
            dt3, k3, kn3, cm3 = timed_engine_run(m3, 60, 400)
            c3c = m3.launch_counts()
            riding3 = c3c["riding"] > c3c["rollout"] // 2
            extra["c3"] = {"workload": desc3, "ms_per_step": dt3 * 1e3, "value": K3 / dt3,
                           "unit": "rollouts/s", "steps": 400, "geometry": geo3,
                           "roofline": roofline_entry("c3", K3, T3, A3, geo3, riding3, k3, kn3, cm3,
                                                      solve_ms=dt3 * 1e3),
                           "launches": c3c}
            extra["c3"]["ess"] = round(ess_of(m3.get_inf(x=False, u=False, e=False, cost=False,
                                                         beta=False, nabla=False)["weight"]), 1)
            extra["c3"]["spread_weights"] = spread_weights_leg(m3, K3, 30, 200)
            m3.set_noise_store(False)
            dt3n, k3n, _, _ = timed_engine_run(m3, 30, 200)
            extra["c3"]["noise_not_materialised"] = {"ms_per_step": dt3n * 1e3,
                                                     "rollout_kernel_ms": round(k3n, 5)}
            m3.close()
    # ---- config 4's strong-scaling leg (K = 1e6 global) when run on several GPUs ---------------
    if N > 1 and not args.rehearse_one_gpu and args.workload != "c4":
        from mppi_gpu_amd.sharded import ShardedPointMassModel
        A4, T4, K4 = 3, 200, 1_000_000
        c4 = make_inputs(A4, T4)
        s4 = ShardedPointMassModel(K4, T4, float(c4["dt"]), 2 * A4, A4, transport=args.transport)
        s4.engine.set_seed(0)
        s4.memcpy_set_data(c4["x0"], c4["U"], c4["goal"], c4["w"])
        for _ in range(20):
            s4.solve_async()
        dist.barrier(); torch.cuda.synchronize(); s4.engine.sync_act()
        n4 = 200
        t4 = time.perf_counter()
        for _ in range(n4):
            s4.solve_async()
        dist.barrier(); torch.cuda.synchronize(); s4.engine.sync_act()
        d4 = time.perf_counter() - t4
        tt = torch.tensor([d4], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        d4 = float(tt.item())
        extra["c4"] = {"workload": "point_mass3d K=1e6 T=200 sharded over the ranks (BASELINE "
                                   "configs[3]), strong scaling", "global_rollouts": K4,
                       "rollouts_per_gpu": s4.engine.K, "steps": n4, "ms_per_step": d4 / n4 * 1e3,
                       "value": K4 * n4 / d4, "unit": "rollouts/s", "scaling": "strong",
                       "exchange": s4.transport, "exchange_validated": getattr(s4, "validated", None)}
        s4.close()

    # ---- roofline.traffic, live: the engines of this process are idle from here on --------------
    if rank == 0 and N == 1 and sharded is None and roof is not None and not args.no_pmc \
            and not under_profiler() and not args.inject:
        fwd = []
        if args.packing:
            fwd += ["--packing", str(args.packing)]
        if args.chunks:
            fwd += ["--chunks", str(args.chunks)]
        if args.pipeline:
            fwd += ["--pipeline", str(args.pipeline)]
        attach_live_traffic(roof, args.workload, kernel_symbol(geo, riding, A), fwd)
        c3x = extra.get("c3")
        if c3x:
            attach_live_traffic(c3x["roofline"], "c3",
                                kernel_symbol(c3x["geometry"], c3x["launches"]["riding"]
                                              > c3x["launches"]["rollout"] // 2, 3))

    if rank == 0:
        value = N * K * args.steps / dt_s
        line = {
            "metric": "rollouts/sec (K x T steps) per MPPI solve",
            "value": value, "unit": "rollouts/s", "n_gpus": N, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_s / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("INVALID: injected zero noise (analysis run)" if args.inject else
                     "INVALID: all ranks on one GPU (rehearsal)" if args.rehearse_one_gpu else
                     "synthetic"),
            "config": {"workload": desc, "rollouts_per_gpu": K, "horizon": T, "act_dim": A,
                       "global_rollouts": N * K, "sharding": f"samples x{N}",
                       "lambda": 1.0, "ess": ess_line,
                       "ess_what": "effective sample size 1/sum(w^2) of the last solve (this rank's "
                                   "shard): ~1 = the softmax is one-hot at the reference's lambda = 1; "
                                   "extra.spread_weights times the same workload at ESS ~ K/3",
                       "mode": "blocking get_act per step" if args.blocking else
                               "solves enqueued back to back (mppi_solve_async), one wait at the end",
                       "exchange": None if sharded is None else
                       {"transport": sharded.transport,
                        "what": {"direct": "combine kernel -> peer inboxes over xGMI (hipIpc), no collective",
                                 "collective": "RCCL all-gather of T*A+2 floats"}[sharded.transport],
                        "validated_against_collective": getattr(sharded, "validated", None)},
                       "geometry": geo, "rollout_steps_per_s": value * T,
                       "launches_in_timed_region": {k: cnt1[k] - cnt0[k] for k in ("rollout", "riding", "combine")}},
            "roofline": roof,
            "latency": latency,
        }
        if extra:
            line["extra"] = extra
        if N == 1 and not args.no_cpu_baseline:
            ncores = os.cpu_count() or 1
            v1, n1 = cpu_baseline_controller(A, K, T, 1, 10.0)
            line["cpu_baseline"] = {
                "value": v1, "unit": "rollouts/s", "cores": 1, "kind": "port",
                "sample": f"{n1} full solves of the same workload (K={K}, T={T}, A={A}) by the "
                          "product's serial CPU controller (ControllerBase via mppi_cpu_*: Philox + "
                          "Box-Muller sampler, rollout, cost, beta, nabla, update, shift), 1 thread"}
            va, na = cpu_baseline_controller(A, K, T, ncores, 5.0)
            line["cpu_baseline"]["all_cores"] = {"value": va, "cores": ncores, "solves": na}
            vo, no = cpu_baseline_oracle(A, K, T)
            line["cpu_baseline"]["oracle"] = {
                "value": vo, "cores": 1, "solves": no,
                "what": "oracle/mppi_oracle.c + rocRAND-host Philox sampler (the checker), cross-check"}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    m.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
