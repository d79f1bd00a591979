import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
import bench
s = syn.band_structure(10000,50); rows, cols = s.kkt_pattern()
B = int(sys.argv[1])
vh, rh = bench.band_batch(s, 512, 3000)
dev = torch.device("cuda",0)
vals = torch.from_numpy(np.tile(vh,(B//512,1))).to(dev); rhs = torch.from_numpy(np.tile(rh,(B//512,1))).to(dev)
d = torch.zeros((B,s.N),dtype=torch.float64,device=dev); ro=torch.zeros(B,dtype=torch.float64,device=dev); rho=torch.zeros_like(ro)
nf=torch.zeros(B,dtype=torch.int32,device=dev); su=torch.zeros_like(nf)
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
L.set_timing(True)
for fl in (0,1):
    p = hipldl.default_params(); p[8] = fl
    ms=[]
    for it in range(4):
        ro.zero_()
        hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p, 0)
        torch.cuda.synchronize(); ms.append(L.last_kernel_ms())
    print("B",B,"ablate",fl, "kernel ms", np.round(ms[1:],2))
