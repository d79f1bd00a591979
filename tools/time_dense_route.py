"""Small batches of irregular patterns: the staged multifrontal execution against ONE dense LDL' of the condensed matrix
(cnl_options.general_dense = 2 forces the latter wherever it is possible), cnl_newton_system_dev, device-resident."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
dev = torch.device("cuda:0"); stream = torch.cuda.Stream()
p = hipldl.default_params()
res = {}
def fuzz_structure(seed):   # the structures of tools/fuzz_parity.py
    rng = np.random.default_rng(100000 + seed)
    fam = rng.integers(3)
    if fam == 0:
        n = int(rng.integers(6, 120)); m = int(rng.integers(max(2, n // 2), 2 * n)); pc = int(rng.integers(0, min(6, n // 2) + 1))
        return syn.random_structure(n, m, pc, float(rng.uniform(0.03, 0.3)), seed, hess=bool(rng.integers(4)))
    if fam == 1:
        return None
    n = int(rng.integers(130, 400)); m = int(rng.integers(n, n + 60)); pc = int(rng.integers(0, 4))
    return syn.random_structure(n, m, pc, float(rng.uniform(0.01, 0.04)), seed)


cases = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    s = fuzz_structure(seed)
    if s is not None and s.nvar + s.ncon >= 96: cases.append((seed, s))
for ci, (n, s) in enumerate(cases):
    dens = 0
    rows, cols = s.kkt_pattern()
    for B in (1,):
        vals = np.stack([syn.random_values(s, 100 + b)[0] for b in range(B)]); rhs = np.stack([syn.random_values(s, 100 + b)[1] for b in range(B)])
        out = {}
        for tag, opt in (("default", {}), ("dense", {"general_dense": 2})):
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
                for _ in range(50): step()
                e1.record(stream); torch.cuda.synchronize()
            info = L.info
            out[tag] = {"N2": s.nvar + s.ncon, "flops": info.get("flops"), "ms": round(e0.elapsed_time(e1) / 50, 4), "kernel": L.config["kernel"], "order": info["order"], "fmax": info.get("fmax"), "nsuper": info.get("nsuper"),
                        "cost": info.get("cost"), "stages": info.get("stages"), "ok": bool((su == 1).all())}
            L.close()
        res[f"n{n}_d{dens}_B{B}"] = out
        print(n, dens, B, json.dumps(out), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/dense_route_timing.json", "w"), indent=1)
