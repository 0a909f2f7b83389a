"""bench.py's bookkeeping that needs no GPU: the algorithmic-bytes figure, the roofline object built
from event timings and the committed profile summaries, the rank spawner's environment."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_follow_survey_8d_minus_what_the_design_removes():
    # K*(2*4*T*A + 2R + 16); here: no E re-load, R = 0, one cost store
    assert bench.algorithmic_bytes_rollout(10_000, 200, 2) == 10_000 * 1604 == 16_040_000
    assert bench.algorithmic_bytes_rollout(100_000, 200, 3) == 100_000 * 2404


def test_roofline_entry_uses_live_timing_and_labels_committed_counters():
    geo = {"packed": True, "groups_per_lane": 4, "chunks": 0, "strict": False}
    r = bench.roofline_entry("c3", 100_000, 200, 3, geo, False, 0.075, 50, 0.006)
    assert r["kernel"] == "k_rollout_packed" and r["bound"] == "hbm"
    assert abs(r["achieved"] - 240.4e6 / 75e-6 / 1e9) < 1 and abs(r["frac"] - r["achieved"] / 8000) < 1e-4
    prof = json.load(open(os.path.join(ROOT, "profiles", bench.PROFILE_TRAFFIC)))
    assert "c3:packed4:plain" in prof["entries"]
    assert r["traffic"] == prof["entries"]["c3:packed4:plain"]["hbm_bytes_per_launch"]
    assert "not measured in this run" in r["traffic_source"] and "not measured in this run" in r["alu"]["source"]
    assert 0.3 < r["alu"]["achieved"] < 1.0 and r["alu"]["bound"] == "valu"
    assert r["kernel_ms_rocprof"] > 0 and 0.2 < r["frac_rocprof"] < 0.6
    # a geometry no profile was taken for: counters stay null, the live figures remain
    geo2 = {"packed": False, "groups_per_lane": 2, "chunks": 32, "strict": False}
    r2 = bench.roofline_entry("c3", 100_000, 200, 3, geo2, False, 0.11, 50, 0.006)
    assert r2["traffic"] is None and r2["traffic_source"] is None and "alu" not in r2
    assert r2["kernel"] == "k_rollout_fused"
    ride = bench.roofline_entry("c2", 10_000, 200, 2, {"packed": False, "groups_per_lane": 7, "chunks": 16,
                                                       "strict": False}, True, 0.0135, 60, 0.0)
    assert ride["kernel"].startswith("k_rollout_ride") and ride["traffic"] is not None


def test_workloads_are_the_baseline_configs():
    assert bench.WORKLOADS["c2"][:3] == (2, 10_000, 200)
    assert bench.WORKLOADS["c3"][:3] == (3, 100_000, 200)
    assert bench.WORKLOADS["c4"][:3] == (3, 125_000, 200) and bench.WORKLOADS["c4full"][1] == 1_000_000
    x = bench.make_inputs(3, 200)
    assert x["U"].shape == (200, 3) and list(x["goal"]) == [1, .5, .75, 0, 0, 0]
