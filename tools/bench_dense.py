#!/usr/bin/env python3
"""Device-resident timing of BASELINE config 2 (dense Jacobian, n = 1000, nequ = 2000) through cnl_newton_system_dev."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--opt", default="", help="cnl_options fields, key=value[,key=value]")
    args = ap.parse_args()
    import torch
    import cannoles_jl_amd  # noqa
    from cannoles_jl_amd import hipldl, synthetic as syn
    s = syn.dense_structure(args.n, args.m)
    rows, cols = s.kkt_pattern()
    B = args.batch
    dev = torch.device("cuda", 0)
    vh = np.stack([syn.dense_values(s, 2002 + b)[0] for b in range(B)])
    rh = np.stack([syn.dense_values(s, 2002 + b)[1] for b in range(B)])
    vals = torch.from_numpy(vh).to(dev); rhs = torch.from_numpy(rh).to(dev)
    d = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
    nf = torch.zeros(B, dtype=torch.int32, device=dev); ok = torch.zeros(B, dtype=torch.int32, device=dev)
    p = hipldl.default_params()
    opts = hipldl.Options(**{k: int(v) for k, v in (kv.split("=") for kv in args.opt.split(","))}) if args.opt else None
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=opts)
    st = torch.cuda.Stream(device=dev)
    def step():
        hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), ok.data_ptr(), p, st.cuda_stream)
    with torch.cuda.stream(st):
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(args.steps): step()
        e1.record(st)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / args.steps
    flop = args.m * args.n ** 2 + args.n ** 3 / 3
    import scipy.sparse as sp
    dh = d[0].cpu().numpy()
    Kl = sp.coo_matrix((vh[0], (rows - 1, cols - 1)), shape=(s.N, s.N)).tocsr()
    K = Kl + sp.tril(Kl, -1).T
    berr = np.abs(K @ dh + rh[0]).max() / (abs(K).sum(axis=1).max() * np.abs(dh).max() + np.abs(rh[0]).max())
    print(json.dumps({"workload": f"cfg2 dense n={args.n} nequ={args.m}", "batch": B, "ms_per_step": ms, "ms_per_system": ms / B,
                      "wall_ms_per_step": (t1 - t0) / args.steps * 1e3, "TFLOPs": flop * B / (ms * 1e-3) / 1e12,
                      "frac_of_78.6": flop * B / (ms * 1e-3) / 78.6e12, "success": bool(ok.all().item()), "backward_error": berr,
                      "kernel": L.config["kernel"]}))
    L.close()

if __name__ == "__main__":
    main()
