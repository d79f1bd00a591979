"""Phase times of the band kernel (library built with -DCNL_EXPERIMENT=1 -DBAND_STAMPS=1, loaded through CANNOLES_HIP_LIB): s_memtime ticks
per phase, summed over the epochs of one wavefront, printed as shares.  usage: band_stamps.py [B]"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
s = syn.band_structure(10000, 50); rows, cols = s.kkt_pattern()
vh, rh = bench.band_batch(s, 64, 3000)
dev = torch.device("cuda", 0)
rep = (B + 63) // 64
vals = torch.from_numpy(np.tile(vh, (rep, 1))[:B]).to(dev); rhs = torch.from_numpy(np.tile(rh, (rep, 1))[:B]).to(dev)
d = torch.zeros((B, s.N), dtype=torch.float64, device=dev); ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
nf = torch.zeros(B, dtype=torch.int32, device=dev); su = torch.zeros_like(nf)
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
p = hipldl.default_params()
for _ in range(3):
    hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p, 0)
torch.cuda.synchronize()
t = d[0, :24].cpu().numpy()
names = ["f:loop-top", "f:commit", "f:issue", "f:steps", "f:Lstore+junction-entry", "b:loop-top", "b:commit", "b:issue", "b:steps+store", "tail", "-", "-"]
for part in range(2):
    tt = t[12 * part: 12 * part + 12]
    tot = tt.sum()
    print(f"part {part}: total ticks {tot:.0f} (= {tot / 100e6 * 1e3:.3f} ms at 100 MHz)")
    for k, nm in enumerate(names):
        if tt[k]:
            print(f"   {nm:28s} {tt[k]:10.0f}  {100 * tt[k] / tot:5.1f} %")
