"""Batches above one that fills the machine — on the bidirectional chain (cfg3's size: 4096 problems) or on the single stream (8192):
the remainder on a handle of its own (cnl_options.split_tail = 1) against the two halves / the single stream's extra round (0),
cnl_newton_system_dev, same box, interleaved.
usage: time_tail.py [batches, comma separated]     writes gpurun_out/tail_timing.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa: F401,E402
from cannoles_jl_amd import hipldl, synthetic as syn  # noqa: E402
import bench as BM  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
batches = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 4100, 4352, 4608, 4864, 5120, 5376, 8448, 9216, 10240, 12288, 14336]
s = syn.band_structure(10000, 50)
rows, cols = s.kkt_pattern()
p = hipldl.default_params()
Bmax = max(batches)
vals = torch.empty((Bmax, s.nnzNS), dtype=torch.float64, device=dev)
rhs = torch.empty((Bmax, s.N), dtype=torch.float64, device=dev)
for b0 in range(0, Bmax, 512):
    vh, rh = BM.band_batch(s, 512, seed=9000 + b0)
    n = min(512, Bmax - b0)
    vals[b0:b0 + n].copy_(torch.from_numpy(vh[:n]))
    rhs[b0:b0 + n].copy_(torch.from_numpy(rh[:n]))
d = torch.zeros((Bmax, s.N), dtype=torch.float64, device=dev)
ro = torch.zeros(Bmax, dtype=torch.float64, device=dev)
rho = torch.zeros(Bmax, dtype=torch.float64, device=dev)
nf = torch.zeros(Bmax, dtype=torch.int32, device=dev)
su = torch.zeros(Bmax, dtype=torch.int32, device=dev)
res = {}
for B in batches:
    res[B] = {}
    Ls = {t: hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(split_tail=t)) for t in (1, 0)}
    ms = {1: [], 0: []}
    for rnd in range(3):
        for t in (1, 0):
            L = Ls[t]

            def step():
                hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(),
                                         su.data_ptr(), p, stream.cuda_stream)
            with torch.cuda.stream(stream):
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(8):
                    step()
                e1.record(stream)
                torch.cuda.synchronize()
            ms[t].append(e0.elapsed_time(e1) / 8)
            assert bool((su[:B] == 1).all())
    for t in (1, 0):
        m = float(np.median(ms[t]))
        res[B]["tail" if t else "halves"] = {"ms_per_call": m, "systems_per_s": B / m * 1e3, "has_tail": Ls[t].config["tail"], "order": Ls[t].info["order"]}
        Ls[t].close()
    print(B, json.dumps(res[B]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/tail_timing.json", "w"), indent=1)
