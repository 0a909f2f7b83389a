#!/usr/bin/env python3
"""Turn what tools/r3_profiles.sh (r2_profiles.sh for round 2) left under gpurun_out/ into the
committed summaries under profiles/: kernel-trace stats, SQ counters per wave, traffic_rNN.json,
alu_rNN.json, bench lines, parity table, micro-benchmark tables.  Run in the authoring container
after the GPU call; no GPU needed.      python tools/mkprofiles.py [round, default 3]"""
import csv
import json
import os
import shutil
import sys

RND = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R, RR = f"r{RND}", f"r{RND:02d}"            # gpurun_out tags / committed file prefix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")

# (profile tag, bench workload key, kernel-name substring, roofline key suffix)
KEYS = [
    (f"{R}_c2", "c2:chunks=16:ride", "k_rollout_ride<2"),
    (f"{R}_c2e", "c2:chunks=16:plain", "k_rollout_fused<2"),
    (f"{R}_c2", "c2:combine_small", "k_combine_small"),
    (f"{R}_c3", "c3:packed4:plain", "k_rollout_packed<3"),
    (f"{R}_c3", "c3:combine_small", "k_combine_small"),
] if RND == 2 else [
    (f"{R}_c2", "c2:chunks=16:ride", "k_rollout_ride<2"),
    (f"{R}_c2e", "c2:chunks=16:plain", "k_rollout_fused<2"),
    (f"{R}_c2e", "c2:combine", "k_combine<"),
    (f"{R}_c3", "c3:packed4:ride", "k_rollout_packed_ride<3"),
    (f"{R}_c3e", "c3:packed4:plain", "k_rollout_packed<3"),
    (f"{R}_c3e", "c3:combine", "k_combine<"),
]
# issue cost of a wave64 instruction in SIMD cycles at the nominal 2.4 GHz, measured with
# tools/ubench_issue on MI355X at >= 4 waves per SIMD and ILP >= 2 (profiles/r02_ubench_issue.txt)
COST = {"INT64": 6.3, "TRANS_F32": 8.4, "BITOP3": 4.2, "OTHER": 2.3}
N_SIMD, F_NOMINAL = 1024, 2.4e9


def rocprof_avg_us(tag, sub):
    for r in csv.DictReader(open(os.path.join(SRC, "prof", f"{tag}_stats.csv"))):
        if sub in r["Name"]:
            return float(r["AverageNs"]) * 1e-3, int(r["Calls"])
    return None, 0


def main():
    os.makedirs(DST, exist_ok=True)
    traffic = {"note": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate "
               "passes with --kernel-trace only, tools/traffic.sh); FETCH_SIZE doubled per "
               "guides/MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half), WRITE_SIZE "
               "exact for dwordx4 streaming stores. ':ride' = the launch that also carries the "
               "previous solve's combine, ':plain' = the rollout alone", "entries": {}}
    alu = {"note": "VALU side of the roofline from rocprofv3 --pmc SQ instruction counters (tools/alu.sh) "
           "and the rocprofv3 --kernel-trace --stats duration of the same command (tools/kt.sh). "
           "alu.achieved = sum over instruction classes of count x measured issue cycles, divided by "
           "the SIMD cycles of the launch (1024 SIMDs x duration x 2.4 GHz nominal); classes and "
           "costs: v_mad_u64_u32 (SQ_INSTS_VALU_INT64) %.1f, transcendentals %.1f, v_bitop3_b32 %.1f "
           "(not counted separately by the PMC: taken as one per v_mad_u64_u32, two of each per "
           "Philox round), every other VALU instruction %.1f cycles (tools/ubench_issue.hip, "
           "profiles/r02_ubench_issue.txt)" % (COST["INT64"], COST["TRANS_F32"], COST["BITOP3"],
                                               COST["OTHER"]),
           "entries": {}}
    for tag, key, sub in KEYS:
        tj = json.load(open(os.path.join(SRC, "prof", f"{tag}_traffic.json")))
        for name, v in tj["kernels"].items():
            if sub in name:
                traffic["entries"][key] = {"kernel": name, "hbm_bytes_per_launch": round(v["hbm_bytes"]),
                                           "fetch_bytes_corrected": round(v["fetch_bytes_corrected_x2"]),
                                           "write_bytes": round(v["write_bytes"])}
        aj = json.load(open(os.path.join(SRC, "prof", f"{tag}_alu.json")))
        us, calls = rocprof_avg_us(tag, sub)
        for name, v in aj["kernels"].items():
            if sub not in name:
                continue
            n_all, n64, ntr = v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU_INT64"], v["SQ_INSTS_VALU_TRANS_F32"]
            nb3 = n64 if "rollout" in name else 0.0
            cyc = (n64 * COST["INT64"] + ntr * COST["TRANS_F32"] + nb3 * COST["BITOP3"]
                   + (n_all - n64 - ntr - nb3) * COST["OTHER"])
            simd_cycles = N_SIMD * us * 1e-6 * F_NOMINAL
            # counter-based: rocprof's VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE
            # (SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE is reported per XCD instance
            # or summed over the 8 of them: normalised by what the dispatch duration implies)
            busy = None
            gui = v.get("GRBM_GUI_ACTIVE")
            act = v.get("SQ_ACTIVE_INST_VALU")
            dur_p = v.get("dispatch_us_mean_in_these_passes") or us
            if gui and act:
                inst = 8.0 if gui / (dur_p * 1e-6) > 6e9 else 1.0
                gui_cycles = gui / inst
                busy = {"valu_busy": round(act * 4.0 / N_SIMD / gui_cycles, 4),
                        "gui_active_cycles_per_dispatch_ns": round(gui_cycles / (dur_p * 1e3), 3),
                        "SQ_ACTIVE_INST_VALU": round(act), "GRBM_GUI_ACTIVE_per_xcd": round(gui_cycles),
                        "formula": "SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / GRBM_GUI_ACTIVE (rocprof VALUBusy); "
                                   "GRBM_GUI_ACTIVE also covers the launch overhead around a dispatch "
                                   "(cycles per ns of dispatch well above the 2.4 GHz clock for the "
                                   "13 us launches): a lower bound there, within 5 % for a 70 us launch"}
            alu["entries"][key] = {
                "kernel": name, "kernel_ms_rocprof": round(us * 1e-3, 6), "rocprof_calls": calls,
                "alu": {"bound": "valu", "achieved": round(cyc / simd_cycles, 4), "peak": 1.0,
                        "unit": "fraction of the launch's VALU issue cycles",
                        "valu_instructions_per_launch": round(n_all),
                        "v_mad_u64_u32": round(n64), "transcendental": round(ntr),
                        "fma_f32": round(v["SQ_INSTS_VALU_FMA_F32"]), "add_f32": round(v["SQ_INSTS_VALU_ADD_F32"]),
                        "mul_f32": round(v["SQ_INSTS_VALU_MUL_F32"]),
                        "issue_cycles_model": round(cyc),
                        "counter": busy}}
    json.dump(traffic, open(os.path.join(DST, f"traffic_{RR}.json"), "w"), indent=1)
    json.dump(alu, open(os.path.join(DST, f"alu_{RR}.json"), "w"), indent=1)
    for tag in ("c2", "c2e", "c3", "c4") + (("c3e",) if RND >= 3 else ()):
        shutil.copy(os.path.join(SRC, "prof", f"{R}_{tag}_stats.csv"),
                    os.path.join(DST, f"{RR}_{tag}_kernel_stats.csv"))
    for tag in ("c2", "c2e", "c3") + (("c3e",) if RND >= 3 else ()):
        shutil.copy(os.path.join(SRC, "prof", f"{R}_{tag}_pmc.txt"), os.path.join(DST, f"{RR}_{tag}_pmc_sq.txt"))
    def refresh(roof, key):
        # bench.py filled the counter-derived fields from the summaries committed BEFORE this run:
        # put this run's own (same box, same call) in their place
        if not roof or key not in alu["entries"]:
            return
        a, t = alu["entries"][key], traffic["entries"].get(key)
        roof["kernel_ms_rocprof"] = a["kernel_ms_rocprof"]
        roof["frac_rocprof"] = round(roof["algorithmic_bytes_per_launch"] / (a["kernel_ms_rocprof"] * 1e-3)
                                     / 1e9 / roof["peak"], 4)
        if "alu" in roof:
            roof["alu"].update(a["alu"])
        if t:
            roof["traffic"] = t["hbm_bytes_per_launch"]

    benches = ["bench_c2", "bench_c3", "bench_c4shard", "bench_c4full", "bench_c1", "bench_c2_sharded_1rank"]
    if RND >= 3:
        benches += ["bench_c2_driverlike", "bench_c3x2"]
    for f in benches:
        line = open(os.path.join(SRC, R, f + ".json")).read().strip().splitlines()[-1]
        d = json.loads(line)
        if RND == 2:
            # round 2's bench.py quoted the counters of the summaries committed BEFORE the run
            if f == "bench_c2":
                refresh(d.get("roofline"), "c2:chunks=16:ride")
                refresh(((d.get("extra") or {}).get("c3") or {}).get("roofline"), "c3:packed4:plain")
            if f == "bench_c3":
                refresh(d.get("roofline"), "c3:packed4:plain")
        # (round 3's bench.py measures traffic / alu / rocprof duration itself, in child passes)
        json.dump(d, open(os.path.join(DST, f"{RR}_{f}.json"), "w"), indent=1)
    extra_txt = ["ubench_issue.txt", "ubench_noise.txt", "latency_probe.txt"] if RND == 2 else \
                ["latency_probe.txt", "sweep_cost_error.txt", "lambda_speed.txt", "closed_loop.txt", "closed_loop_rates.txt", "trace_regions.txt", "soak.txt", "store_mode.txt"]
    for f in extra_txt:
        txt = [l for l in open(os.path.join(SRC, R, f)).read().splitlines()
               if "warning" not in l and "amdgpu.ids" not in l]
        open(os.path.join(DST, f"{RR}_" + f), "w").write("\n".join(txt) + "\n")
    if os.path.exists(os.path.join(SRC, f"parity_{RR}.json")):
        shutil.copy(os.path.join(SRC, f"parity_{RR}.json"), os.path.join(DST, f"parity_{RR}.json"))
    for k, v in alu["entries"].items():
        print(k, v["kernel_ms_rocprof"], v["alu"]["achieved"])
    for k, v in traffic["entries"].items():
        print(k, v["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
