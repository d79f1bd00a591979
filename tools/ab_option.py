"""A/B of one cnl_options switch (value 1 against 0) on cfg3's pattern: cnl_newton_system_dev, same box, interleaved rounds.
usage: ab_option.py <option> [batches, comma separated]     writes gpurun_out/ab_<option>.json"""
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
OPT = sys.argv[1]
batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8192, 4096, 1024, 256, 1]
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
    REPS = 8 if B >= 1024 else 50
    Ls = {t: hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(**{OPT: t})) for t in (1, 0)}
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
                for _ in range(REPS):
                    step()
                e1.record(stream)
                torch.cuda.synchronize()
            ms[t].append(e0.elapsed_time(e1) / REPS)
            assert bool((su[:B] == 1).all())
    for t in (1, 0):
        m = float(np.median(ms[t]))
        res[B]["on" if t else "off"] = {"ms_per_call": m, "systems_per_s": B / m * 1e3, "order": Ls[t].info["order"], "rounds_ms": [round(x, 4) for x in ms[t]]}
        Ls[t].close()
    print(B, json.dumps(res[B]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/ab_{OPT}.json", "w"), indent=1)
