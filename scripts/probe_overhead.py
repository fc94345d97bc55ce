import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from se_snmf_nat_amd import Context, Plan
ctx = Context(0)
F,T,r = 257,100000,256
for rep in range(3):
    t=time.perf_counter(); pl = Plan(ctx,F,T,r,max_iter=50,sparsity=5.0,cost_check=True); ctx.sync(); t1=time.perf_counter()
    pl.close(); t2=time.perf_counter()
    print("plan create %.2f ms destroy %.2f ms" % ((t1-t)*1e3,(t2-t1)*1e3))
H = np.asfortranarray(np.random.rand(r,T))
t=time.perf_counter(); H2 = H.copy(order="F"); print("H copy 205MB %.2f ms" % ((time.perf_counter()-t)*1e3))
t=time.perf_counter(); H3 = np.empty_like(H); print("empty %.3f ms" % ((time.perf_counter()-t)*1e3))
t=time.perf_counter(); H3[:] = 0; print("first touch %.2f ms" % ((time.perf_counter()-t)*1e3))
t=time.perf_counter(); H3[:] = 0; print("second touch %.2f ms" % ((time.perf_counter()-t)*1e3))
