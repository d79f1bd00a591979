"""What the restriction "plans with out-of-line front classes keep the single stream" (cnl_options.staged_large_fronts = 0) costs:
cnl_newton_system_dev on irregular patterns whose fronts reach order 17 .. 64, small batches, both ways (the staged path with the
in-kernel ladder off: the combination that ran the randomised cases clean)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
dev = torch.device("cuda:0"); stream = torch.cuda.Stream()
p = hipldl.default_params()
res = {}
def fuzz_structure(seed):   # the structure of case `seed` of tools/fuzz_parity.py (families 0 and 2)
    rng = np.random.default_rng(100000 + seed)
    fam = rng.integers(3)
    if fam == 0:
        n = int(rng.integers(6, 120)); m = int(rng.integers(max(2, n // 2), 2 * n)); pc = int(rng.integers(0, min(6, n // 2) + 1))
        return syn.random_structure(n, m, pc, float(rng.uniform(0.03, 0.3)), seed, hess=bool(rng.integers(4)))
    n = int(rng.integers(130, 400)); m = int(rng.integers(n, n + 60)); pc = int(rng.integers(0, 4))
    return syn.random_structure(n, m, pc, float(rng.uniform(0.01, 0.04)), seed)


for name, seed in {"case7": 7, "case15": 15, "case176": 176, "case225": 225}.items():
    s = fuzz_structure(seed)
    rows, cols = s.kkt_pattern()
    for B in (1, 16):
        vals = np.stack([syn.random_values(s, 100 + b)[0] for b in range(B)]); rhs = np.stack([syn.random_values(s, 100 + b)[1] for b in range(B)])
        print(name, "n", s.nvar, "m", s.nequ, "p", s.ncon, flush=True) if B == 1 else None
        out = {}
        for tag, opt in (("single_stream", {"staged": 0}), ("staged", {})):   # (round 5: the staged execution is the default again — without the fused ladder)
            L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(**opt))
            tv, tr = torch.tensor(vals, device=dev), torch.tensor(rhs, device=dev)
            td = torch.zeros((B, s.N), dtype=torch.float64, device=dev); ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
            nf = torch.zeros(B, dtype=torch.int32, device=dev); su = torch.zeros_like(nf)
            def step():
                hipldl.newton_system_dev(L, tv.data_ptr(), tr.data_ptr(), td.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p, stream.cuda_stream)
            with torch.cuda.stream(stream):
                for _ in range(5): step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(100): step()
                e1.record(stream); torch.cuda.synchronize()
            out[tag] = {"ms_per_call": e0.elapsed_time(e1) / 100, "kernel": L.config["kernel"], "order": L.info["order"], "fmax": L.info.get("fmax"), "ok": bool((su == 1).all())}
            L.close()
        res[f"{name}_B{B}"] = out
        print(name, B, json.dumps(out), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/restriction_timing.json", "w"), indent=1)
