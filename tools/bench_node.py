#!/usr/bin/env python3
"""Throughput of the SINGLE-PROCESS multi-GPU host (include/point_mass_sharded.hpp,
libmppi_gpu_amd_sharded.so: one shard engine + host worker thread per GPU, native RCCL) -- the
arrangement the reference's own C++ host would use.  bench.py measures the other arrangement (one
process per GPU), because the driver launches it that way.

    python tools/bench_node.py [--gpus N | --same-device N] [--transport collective|direct|copy]
                               [--workload c2|c3|c4|c4full] [--steps 500]

--gpus N: shards on devices 0..N-1 (0 = all visible).  --same-device N: N shards on device 0
(rehearsal on a one-GPU box; transports direct and copy; with `direct` the shards' riding launches
wait for each other and must fit the chip together: keep K small).  Prints one JSON line: global rollouts/s
with solves enqueued back to back, and the blocking get_act latency."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # workloads + synthetic inputs
from mppi_gpu_amd.node import NodePointMassModel

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=0)
ap.add_argument("--same-device", type=int, default=0)
ap.add_argument("--transport", default="collective", choices=("collective", "direct", "copy"))
ap.add_argument("--workload", default="c4full", choices=sorted(bench.WORKLOADS))
ap.add_argument("--steps", type=int, default=500)
args = ap.parse_args()

A, K, T, desc = bench.WORKLOADS[args.workload]
c = bench.make_inputs(A, T)
devices = [0] * args.same_device if args.same_device else None
m = NodePointMassModel(K, T, float(c["dt"]), 2 * A, A, n_shards=args.gpus, devices=devices,
                       transport=args.transport)
m.set_seed(0)
m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
t_r = time.perf_counter()
while time.perf_counter() - t_r < 0.05:        # clock ramp
    for _ in range(20):
        m.solve_async()
    m.sync_act()
t0 = time.perf_counter()
for _ in range(args.steps):
    m.solve_async()
act = m.sync_act()
dt = (time.perf_counter() - t0) / args.steps
n_lat = 200
t0 = time.perf_counter()
for _ in range(n_lat):
    m.get_act()
lat = (time.perf_counter() - t0) / n_lat
print(json.dumps({"what": "single-process multi-GPU host (ShardedPointMassModel / mppi_sharded_*)",
                  "workload": desc, "global_rollouts": K, "shards": m.n_shards,
                  "devices": [m.shard_info(i)["device"] for i in range(m.n_shards)],
                  "transport": m.transport, "steps": args.steps, "ms_per_step": dt * 1e3,
                  "value": K / dt, "unit": "rollouts/s", "blocking_get_act_ms": lat * 1e3,
                  "riding_launches_shard0": m.engine_launch_counts(0)["riding"]}))
m.close()
