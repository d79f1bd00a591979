import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
import ctypes as C
s = syn.band_structure(10000, 50); rows, cols = s.kkt_pattern()
for B in (1, 256):
    vh = np.stack([syn.band_values(s, 3001 + b)[0] for b in range(min(B, 8))]); rh = np.stack([syn.band_values(s, 3001 + b)[1] for b in range(min(B, 8))])
    vh = np.tile(vh, (B // len(vh) + 1, 1))[:B]; rh = np.tile(rh, (B // len(rh) + 1, 1))[:B]
    dev = torch.device("cuda", 0)
    vals = torch.from_numpy(vh).to(dev); rhs = torch.from_numpy(rh).to(dev); d = torch.zeros_like(rhs); ok = torch.zeros(B, dtype=torch.int32, device=dev)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    lib = hipldl.lib()
    def fac(): hipldl._check(lib.cnl_factorize_dev(L._h, vals.data_ptr(), 2.220446049250313e-16, ok.data_ptr(), 0))
    def sol(): hipldl._check(lib.cnl_solve_dev(L._h, rhs.data_ptr(), d.data_ptr(), 0))
    for f, name in ((fac, "factorize"), (sol, "solve")):
        fac(); f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize()
        print(f"B={B} {name}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call ({L.config['kernel']})")
    L.close()
