"""The reference's own call pattern — one system per call, host pointers, pageable arrays (cfg3's size): newton_system!, and the
two-call sequence try_to_factorize + solve_ldl!.  CANNOLES_HIP_LIB selects the library (A/B of builds: run twice)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
import bench

s = syn.band_structure(10000, 50); rows, cols = s.kkt_pattern()
prm = hipldl.default_params()
vh, rh = bench.band_batch(s, 1, 3000)
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=1)
v, r, d = vh.copy(), rh.copy(), np.zeros((1, s.N))


def timed(f, n=300):
    for _ in range(10):
        f()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        best = min(best, 1e3 * (time.perf_counter() - t0) / n)
    return best


out = {"newton_system_ms": timed(lambda: hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, r, v, L, 0.0, prm)),
       "try_to_factorize_ms": timed(lambda: hipldl.try_to_factorize(L, v, s.nvar, s.nequ, s.ncon, prm[0])),
       "solve_ldl_ms": timed(lambda: hipldl.solve_ldl_(r, L.factor, d))}
print(os.environ.get("CANNOLES_HIP_LIB", "default"), json.dumps(out), flush=True)
