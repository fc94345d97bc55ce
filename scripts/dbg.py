import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle.sparse_nmf_oracle import sparse_nmf as onmf, synth_problem
from se_snmf_nat_amd import sparse_nmf
for (F,T,r,p) in [(40,50,1,dict(cf="kl", sparsity=0.1, max_iter=10)), (129,300,300,dict(cf="beta", beta=1.5, sparsity=1, max_iter=10))]:
    V,W0,H0=synth_problem(F,T,r); p=dict(p,init_w=W0,init_h=H0,cost_check=1)
    w,h,o=sparse_nmf(V,p); wr,hr,orf=onmf(V,p)
    print(V.sum(), o['div'], orf['div'], o['cost']/orf['cost']-1, np.linalg.norm(w-wr)/np.linalg.norm(wr), np.linalg.norm(h-hr)/np.linalg.norm(hr))
import torch; print(torch.cuda.is_available(), torch.zeros(3,device='cuda'))
