import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
import bench
dev = torch.device("cuda", 0)
for (n, p, B) in ((1000, 10, 256), (1000, 10, 32), (10000, 50, 256)):
    s = syn.band_structure(n, p); rows, cols = s.kkt_pattern()
    vh, rh = bench.band_batch(s, B, 4000)
    vals, rhs = torch.from_numpy(vh).to(dev), torch.from_numpy(rh).to(dev)
    d = torch.zeros((B, s.N), dtype=torch.float64, device=dev); ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
    nf = torch.zeros(B, dtype=torch.int32, device=dev); su = torch.zeros_like(nf)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    p_ = hipldl.default_params()
    st = torch.cuda.Stream()
    def step():
        hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p_, st.cuda_stream)
    with torch.cuda.stream(st):
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(50): step()
        e1.record(st); torch.cuda.synchronize()
    c0 = hipldl.launch_counts()
    with torch.cuda.stream(st):
        step(); torch.cuda.synchronize()
    c1 = hipldl.launch_counts()
    print(n, p, B, "ms/call", e0.elapsed_time(e1) / 50, L.config["kernel"], L.info["order"], "launches per call", {k: c1[k] - c0[k] for k in c1}, "stages", len(L.plan_array("stage_ptr")) - 1, flush=True)
    L.close()
