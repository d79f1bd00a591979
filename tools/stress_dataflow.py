"""Repeats the dataflow execution (B <= 8) many times and checks that every run gives bit-identical results."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
import bench
s = syn.band_structure(10000, 50); rows, cols = s.kkt_pattern()
prm = hipldl.default_params()
for B in (1, 5, 8):
    vh, rh = bench.band_batch(s, B, 3000)
    dev = torch.device("cuda", 0)
    vals = torch.from_numpy(vh).to(dev); rhs = torch.from_numpy(rh).to(dev)
    d = torch.zeros((B, s.N), dtype=torch.float64, device=dev); ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
    nf = torch.zeros(B, dtype=torch.int32, device=dev); ok = torch.zeros_like(nf)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    ref = None; bad = 0
    for it in range(400):
        d.fill_(float("nan")); ro.zero_()
        hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), ok.data_ptr(), prm, 0)
        torch.cuda.synchronize()
        cur = d.clone()
        if ref is None: ref = cur
        elif not torch.equal(torch.nan_to_num(cur, nan=-7.0), torch.nan_to_num(ref, nan=-7.0)) or not bool((ok == 1).all()): bad += 1
    print("B", B, L.config["kernel"], "runs 400, mismatching runs:", bad, "finite", bool(torch.isfinite(ref).all()))
    L.close()
