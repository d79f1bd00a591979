"""residual_vectors / trial_point at the headline shapes (cfg3 pattern, B problems): ms and fraction of 8 TB/s on their algorithmic bytes."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
s = syn.band_structure(10000, 50); rows, cols = s.kkt_pattern()
dev = torch.device("cuda:0")
vh, rh = bench.band_batch(s, 512, 3000)
vals = torch.from_numpy(np.tile(vh, (B // 512, 1))).to(dev)
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
g = torch.Generator(device=dev); g.manual_seed(1)
rv = torch.randn((B, s.nequ), dtype=torch.float64, device=dev, generator=g); Fx = torch.randn_like(rv)
lam = torch.randn((B, s.ncon), dtype=torch.float64, device=dev, generator=g); cx = torch.randn_like(lam)
rhs = torch.zeros((B, s.N), dtype=torch.float64, device=dev); nrm = torch.zeros((B, 2), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: hipldl.residual_vectors_dev(L, vals.data_ptr(), rv.data_ptr(), lam.data_ptr(), Fx.data_ptr(), cx.data_ptr(), rhs.data_ptr(), nrm.data_ptr(), st))
by = 8 * (s.nnzjF + s.nnzjc + 2 * s.nequ + 2 * s.ncon + s.N)
print(json.dumps({"B": B, "residual_vectors_ms": ms, "bytes_per_system": by, "GBps": by * B / ms / 1e6, "frac": by * B / ms / 1e6 / 8000}))
