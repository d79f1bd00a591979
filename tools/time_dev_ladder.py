"""Device-pointer newton_system! with and without the in-kernel rho ladder (cnl_options.device_ladder / device_ladder_fused):
  * the common case (nothing fails): one system of cfg3's size, 8, 256 problems; cfg4's pattern at 256 — must not get slower
  * cfg5 (every problem climbs to nfact = 6) at 256 problems of cfg4's pattern
  * one system of cfg3's size that climbs to nfact = 6
Writes gpurun_out/dev_ladder_timing.json."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa: F401,E402
from cannoles_jl_amd import hipldl, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
MODES = {"fused": {"device_ladder_fused": 1}, "behind": {}, "sequential": {"device_ladder": 0}}
if len(sys.argv) > 1:   # e.g. `behind` = the library's default only (profiling runs)
    MODES = {k: MODES[k] for k in sys.argv[1].split(",")}


def run(s, vals, rhs, B, mode, reps=30, restore=False):
    rows, cols = s.kkt_pattern()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(**MODES[mode]))
    p = hipldl.default_params()
    tv0 = torch.tensor(vals, device=dev)
    tv, tr = tv0.clone(), torch.tensor(rhs, device=dev)
    td = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    ro = torch.zeros(B, dtype=torch.float64, device=dev)
    rho = torch.zeros(B, dtype=torch.float64, device=dev)
    nf = torch.zeros(B, dtype=torch.int32, device=dev)
    su = torch.zeros(B, dtype=torch.int32, device=dev)

    def step():
        if restore:
            tv.copy_(tv0)
            ro.zero_()
        hipldl.newton_system_dev(L, tv.data_ptr(), tr.data_ptr(), td.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p,
                                 stream.cuda_stream)
    with torch.cuda.stream(stream):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out = {"ms_per_call": ms, "systems_per_s": B / ms * 1e3, "nfact_mean": float(nf.float().mean()), "all_success": bool((su == 1).all()),
           "order": L.info["order"], "timeouts": L.dataflow_timeouts()}
    L.close()
    return out


res = {}
s3 = syn.band_structure(10000, 50)
s4 = syn.band_structure(1000, 10)
for name, s, B, cfg, stress, reps in (("cfg3_B1", s3, 1, 3, None, 100), ("cfg3_B8", s3, 8, 3, None, 60), ("cfg3_B256", s3, 256, 3, None, 20),
                                      ("cfg4_B32", s4, 32, 4, None, 100), ("cfg4_B256", s4, 256, 4, None, 60),
                                      ("cfg5_B256_nfact6", s4, 256, 5, "ladder", 10), ("cfg5_B32_nfact6", s4, 32, 5, "ladder", 10),
                                      ("cfg3_B1_nfact6", s3, 1, 5, "ladder", 10), ("cfg3_B256_one_climber", s3, 256, 3, "one", 10)):
    if stress == "one":
        vals, rhs = syn.batch_values(s, B, cfg=3)
        vl, rl = syn.batch_values(s, 1, cfg=5, stress="ladder")
        vals[7], rhs[7] = vl[0], rl[0]
    else:
        vals, rhs = syn.batch_values(s, B, cfg=cfg, stress=stress) if stress else syn.batch_values(s, B, cfg=cfg)
    res[name] = {}
    for mode in MODES:
        if mode == "sequential" and stress and s is s3 and B > 1:
            continue   # tens of milliseconds per rung: known
        res[name][mode] = run(s, vals, rhs, B, mode, reps=reps, restore=bool(stress))
        print(name, mode, json.dumps(res[name][mode]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/dev_ladder_timing.json", "w"), indent=1)
