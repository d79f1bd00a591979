"""Row f1 (cnl_residual_vectors_dev) on BASELINE config 3's pattern: column tiles through LDS against the gather kernel."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
dev = torch.device("cuda:0"); stream = torch.cuda.Stream()
s = syn.band_structure(10000, 50)
rows, cols = s.kkt_pattern()
nnz = len(rows)
nnzj = len(s.jF[0]) + len(s.jc[0])
by = 8 * (nnzj + 2 * s.nequ + 2 * s.ncon + s.N)
res = {"bytes_per_system": by}
for B in [int(a) for a in sys.argv[1:]] or [8192]:
    g = torch.Generator(device=dev); g.manual_seed(1)
    vals = torch.randn((B, nnz), dtype=torch.float64, device=dev, generator=g)
    r = torch.randn((B, s.nequ), dtype=torch.float64, device=dev, generator=g); Fx = torch.randn_like(r)
    lam = torch.randn((B, s.ncon), dtype=torch.float64, device=dev, generator=g); cx = torch.randn_like(lam)
    outs = {}
    for tiles in (1, 0):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(f1_tiles=tiles))
        rhs = torch.zeros((B, s.N), dtype=torch.float64, device=dev); nrm = torch.zeros((B, 2), dtype=torch.float64, device=dev)
        def step():
            hipldl.residual_vectors_dev(L, vals.data_ptr(), r.data_ptr(), lam.data_ptr(), Fx.data_ptr(), cx.data_ptr(), rhs.data_ptr(), nrm.data_ptr(), stream.cuda_stream)
        with torch.cuda.stream(stream):
            for _ in range(3): step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20): step()
            e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        outs[tiles] = (rhs.clone(), nrm.clone())
        res[f"B{B}_tiles{tiles}"] = {"ms": ms, "GBps": by * B / ms / 1e6, "frac": by * B / ms / 1e6 / 8000, "tiles": L.config["f1_tiles"]}
        print(B, tiles, res[f"B{B}_tiles{tiles}"], flush=True)
        L.close()
    same = bool(torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1]))
    res[f"B{B}_bit_identical"] = same
    print("bit-identical:", same)
    del vals, r, Fx, outs
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/f1_timing.json", "w"), indent=1)
