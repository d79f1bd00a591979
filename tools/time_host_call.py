"""Host-pointer cnl_newton_system (the literal drop-in call) at small batches: pageable against pinned caller arrays."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
import bench

def pinned_like(a):
    t = torch.empty(a.shape, dtype=torch.float64, pin_memory=True)
    n = t.numpy(); n[...] = a
    return n, t

s = syn.band_structure(10000, 50); rows, cols = s.kkt_pattern()
prm = hipldl.default_params()
out = {}
for B in (1, 16, 512):
    vh, rh = bench.band_batch(s, B, 3000)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            v, _kv = pinned_like(vh); r, _kr = pinned_like(rh); d, _kd = pinned_like(np.zeros((B, s.N)))
        else:
            v, r, d = vh.copy(), rh.copy(), np.zeros((B, s.N))
        ro = np.zeros(B)
        for _ in range(5):
            hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, r, v, L, ro, prm)
        n = 30 if B <= 16 else 8
        t0 = time.perf_counter()
        for _ in range(n):
            hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, r, v, L, ro, prm)
        ms = 1e3 * (time.perf_counter() - t0) / n
        out[f"B{B}_{kind}"] = {"ms_per_call": ms, "systems_per_s": B / ms * 1e3}
        print(B, kind, "%.3f ms/call" % ms, "%.0f systems/s" % (B / ms * 1e3), flush=True)
    L.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/host_call_timing.json", "w"), indent=1)
