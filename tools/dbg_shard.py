import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel
K, A, T, seed = 5001, 2, 200, 23
case = ol.make_case(A, 1, T, seed=5, u_scale=0.03)
m = PointMassModel(K, T, float(case["dt"]), 2*A, A); m.set_seed(seed)
m.memcpy_set_data(case["x0"], case["U"], case["goal"], case["w"])
ref=[]
for it in range(3):
    a=m.get_act(); ref.append((a.copy(), m.get_u().copy(), m.get_inf(x=False,u=False,cost=False,beta=False,nabla=False,weight=False)["e"].copy()))
shards=[]
for off,n in ((0,2501),(2501,2500)):
    s=PointMassModel(n,T,float(case["dt"]),2*A,A,k_offset=off); s.set_seed(seed)
    s.memcpy_set_data(case["x0"], case["U"], case["goal"], case["w"]); shards.append(s)
L=shards[0].partial_len()
for it in range(3):
    g=torch.zeros(2,L,device="cuda"); torch.cuda.synchronize()
    for i,s in enumerate(shards):
        s.solve_local_async(g[i].data_ptr()); s.sync_act()
    acts=[]
    for s in shards:
        s.solve_finish_async(g.data_ptr(),2); acts.append(s.sync_act())
    E=np.concatenate([s.get_inf(x=False,u=False,cost=False,beta=False,nabla=False,weight=False)["e"] for s in shards])
    print(it, "act diff", np.abs(acts[0]-ref[it][0]).max(), "U diff", np.abs(shards[0].get_u()-ref[it][1]).max(), "E equal", np.array_equal(E, ref[it][2]), "partials", g[:, :2].cpu().numpy().tolist())
