"""Times the device-resident lockstep outer loop (row f3) against the single-problem host loop with the CPU oracle."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import device_loop as DL, synthetic as syn, outer_loop, hipldl

def main():
    out = []
    for (n, p, B) in [(300, 4, 256), (300, 4, 2048), (300, 4, 8192), (2000, 10, 1024), (1000, 10, 16384)]:
        s = syn.band_structure(n, p)
        fam = DL.BandQuadFamily(s, B, seed=7, torch=torch, device="cuda:0", curvature=1.5, start=1.0, noise=0.5)
        prm = hipldl.default_params()
        DL.solve_batch_device(fam, prm)
        if "--profile" in sys.argv:
            DL.PROFILE = True
            print("profile", (n, p, B), json.dumps(DL.solve_batch_device(fam, prm)["profile_ms_per_step"]), flush=True)
            DL.PROFILE = False
        t0 = time.perf_counter()
        got = DL.solve_batch_device(fam, prm)
        dt = time.perf_counter() - t0
        rec = {"n": n, "p": p, "B": B, "seconds": dt, "problems_per_s": B / dt, "steps": got["steps"], "ms_per_step": 1e3 * got["loop_seconds"] / got["steps"], "setup_seconds": dt - got["loop_seconds"],
               "newton_systems": int(got["nlinsolve"].sum()), "factorisations": int(got["nfact"].sum()),
               "first_order": sum(st == "first_order" for st in got["status"]), "kernel": got["kernel"], "vals_layout": got.get("vals_layout")}
        if os.path.isdir("oracle"):
            from tests.test_oracle_pinning import oracle_newton, oracle_solver
            t0 = time.perf_counter()
            k = min(B, 64)
            for b in range(k):
                outer_loop.solve(fam.host_model(b), oracle_solver, oracle_newton, prm)
            rec["host_loop_cpu_oracle_problems_per_s_1core"] = k / (time.perf_counter() - t0)
        print(json.dumps(rec), flush=True)
        out.append(rec)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/device_loop_timing.json", "w"), indent=1)

main()
