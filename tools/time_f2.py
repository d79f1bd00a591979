"""Row f2 (cnl_prepare_newton_system_dev) and the trial point of row f1 on BASELINE config 3's pattern: time and fraction of the HBM rate
on their own bytes."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
dev = torch.device("cuda:0"); stream = torch.cuda.Stream()
s = syn.band_structure(10000, 50)
rows, cols = s.kkt_pattern()
nnz = len(rows)
nhF, nhc, njF, njc = len(s.hF[0]), len(s.hc[0]), len(s.jF[0]), len(s.jc[0])
res = {}
for B in [int(a) for a in sys.argv[1:]] or [8192]:
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    g = torch.Generator(device=dev); g.manual_seed(1)
    mk = lambda n: torch.randn((B, max(n, 1)), dtype=torch.float64, device=dev, generator=g)
    hF, hc, Jx, Jcx = mk(nhF), mk(nhc), mk(njF), mk(njc)
    delta = torch.full((B,), 1e-8, dtype=torch.float64, device=dev)
    vals = torch.zeros((B, nnz), dtype=torch.float64, device=dev)
    x, r, lam, d = mk(s.nvar), mk(s.nequ), mk(s.ncon), mk(s.N)
    xt, rt, lt, dl = torch.empty_like(x), torch.empty_like(r), torch.empty_like(lam), torch.empty_like(lam)
    def timed(fn, reps=20):
        with torch.cuda.stream(stream):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps): fn()
            e1.record(stream); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms2 = timed(lambda: hipldl.prepare_newton_system_dev(L, nhF, nhc, njF, njc, hF.data_ptr(), hc.data_ptr(), Jx.data_ptr(), Jcx.data_ptr(), delta.data_ptr(), vals.data_ptr(), stream.cuda_stream))
    by2 = 8 * (nhF + nhc + njF + njc + (nnz - s.nequ))     # reads of the four value arrays + writes of every slot but the -I diagonal
    mst = timed(lambda: hipldl.trial_point_dev(L, x.data_ptr(), r.data_ptr(), lam.data_ptr(), d.data_ptr(), 1e4, xt.data_ptr(), rt.data_ptr(), lt.data_ptr(), dl.data_ptr(), stream.cuda_stream))
    byt = 8 * (s.nvar + s.nequ + s.ncon + s.N + s.nvar + s.nequ + 2 * s.ncon)
    res[f"B{B}"] = {"prepare": {"ms": ms2, "bytes_per_system": by2, "frac": by2 * B / ms2 / 1e6 / 8000}, "trial_point": {"ms": mst, "bytes_per_system": byt, "frac": byt * B / mst / 1e6 / 8000}}
    print(B, json.dumps(res[f"B{B}"]), flush=True)
    L.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/f2_timing.json", "w"), indent=1)
