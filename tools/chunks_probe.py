import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as ol
from mppi_gpu_amd import PointMassModel, MppiError
def timed(A,K,T,chunks):
    c=ol.make_case(A,1,T,seed=5,u_scale=0.0)
    with PointMassModel(K,T,float(c["dt"]),2*A,A) as m:
        m.set_seed(0)
        try:
            m.set_packing(-1); m.set_tuning(chunks=chunks)
            m.memcpy_set_data(c["x0"],c["U"],c["goal"],c["w"])
            m.solve_async(); m.sync_act()
        except MppiError as ex:
            return None,None
        t0=time.perf_counter()
        while time.perf_counter()-t0<0.04:
            for _ in range(20): m.solve_async()
            m.sync_act()
        best=1e9
        for _ in range(3):
            t0=time.perf_counter()
            for _ in range(2000): m.solve_async()
            m.sync_act()
            best=min(best,(time.perf_counter()-t0)/2000)
        return best*1e6, m.geometry()
for A,T in ((1,200),(2,200),(3,200),(4,200),(2,50)):
    for K in (1000,2000,3000,5000,7000,10000):
        row="A %d T %d K %5d:"%(A,T,K)
        for ch in (0,4,8,16,32,64):
            t,g=timed(A,K,T,ch)
            row+="  c%s %s"%(("auto(%d)"%g["chunks"]) if ch==0 and g else ch, "%6.2f"%t if t else "  n/a ")
        print(row,flush=True)
