"""Irregular sparsity (fronts of order > 64) at mid-size batches: systems/s of whatever serves it (tests' random structures)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
dev = torch.device("cuda", 0)
out = {}
for seed in (1, 2, 3):
    s = syn.random_structure(90 + 10 * seed, 130, 6 if seed % 2 else 0, 0.03, seed)
    rows, cols = s.kkt_pattern()
    for B in (96, 640, 4096):
        for general_dense in (1, 0):
            v8, r8 = syn.batch_values(s, 8, cfg=seed, gen=syn.random_values)
            reps = (B + 7) // 8
            vals = torch.from_numpy(np.tile(v8, (reps, 1))[:B].copy()).to(dev); rhs = torch.from_numpy(np.tile(r8, (reps, 1))[:B].copy()).to(dev)
            L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(general_dense=general_dense))
            d = torch.zeros((B, s.N), dtype=torch.float64, device=dev); ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
            nf = torch.zeros(B, dtype=torch.int32, device=dev); su = torch.zeros_like(nf)
            p = hipldl.default_params()
            def step():
                ro.zero_()
                hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p, 0)
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            key = f"seed{seed}_B{B}_{'dense' if general_dense else 'nodense'}"
            out[key] = {"kernel": L.config["kernel"], "fmax": L.info["fmax"], "ms": dt * 1e3, "systems_per_s": B / dt, "ok": int((su == 1).sum().item())}
            print(key, out[key], flush=True)
            L.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "irregular_timing.json"), "w"), indent=1)
