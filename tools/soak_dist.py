#!/usr/bin/env python3
"""Soak of the sharded exchange: ranks are PROCESSES sharing cuda:0 (gloo for the set-up), long
chains of solves with the direct exchange riding in the next rollout launch against the same
chains through the collective transport with a synchronisation after every solve.
launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/soak_dist.py [n] [K per rank]
(the ranks' riding launches wait for one another, so their grids must fit on the one GPU TOGETHER:
two ranks of K = 5000, or three of K = 2000; one GPU per rank has no such limit)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import torch.distributed as dist
import oracle_lib as ol
from mppi_gpu_amd.sharded import ShardedPointMassModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
k_rank = int(sys.argv[2]) if len(sys.argv) > 2 else (5000 if world <= 2 else 2000)
A, K, T = 2, k_rank * world, 200
c = ol.make_case(A, 1, T, seed=0, u_scale=0.0)
out = []
for transport, every in (("direct", 0), ("collective", 1)):
    m = ShardedPointMassModel(K, T, float(c["dt"]), 2 * A, A, transport=transport)
    m.engine.set_seed(3)
    m.memcpy_set_data(c["x0"], c["U"], c["goal"], c["w"])
    for i in range(n):
        m.solve_async()
        if every or i % 500 == 499:
            m.sync_act()
    act = m.sync_act()
    out.append((act.copy(), m.get_u().copy()))
    m.close()
same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
flag = torch.tensor([1 if same else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print(f"dist soak: world={world} solves={n} direct(riding)==collective: {bool(flag.item())} act {out[0][0]}")
dist.destroy_process_group()
sys.exit(0 if flag.item() else 1)
