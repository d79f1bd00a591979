"""Band kernels (cnl_options.band_kernel) against the register-front kernel on the headline pattern (cfg3: n = nequ = 1e4, ncon = 50),
device-resident newton_system!, same process, interleaved.  usage: time_band.py [B ...]   prints one JSON line per batch size."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa: F401,E402
from cannoles_jl_amd import hipldl, synthetic as syn  # noqa: E402
import bench  # noqa: E402

n, p = int(os.environ.get("BAND_N", 10000)), int(os.environ.get("BAND_P", 50))
s = syn.band_structure(n, p)
rows, cols = s.kkt_pattern()
par = hipldl.default_params()
dev = torch.device("cuda", 0)
variants = {"front": dict(band_kernel=0), "band8": dict(band_kernel=1, band_problems_per_group=8), "band16": dict(band_kernel=1, band_problems_per_group=16),
            "band32": dict(band_kernel=1, band_problems_per_group=32), "bandmw": dict(band_kernel=1, band_movers=1), "bandmw2": dict(band_kernel=1, band_movers=2),
            "bandmw3": dict(band_kernel=1, band_movers=3),
            # cnl_options.batch_layout = CNL_LAYOUT_INTERLEAVED: vals interleaved over groups of 32 problems; il2: rhs as well (tuning key)
            "band32il": dict(band_kernel=1, batch_layout=1, band_problems_per_group=32), "band16il": dict(band_kernel=1, batch_layout=1, band_problems_per_group=16), "band32il2": dict(band_kernel=1, batch_layout=1, band_rhs_interleaved=1),
            "bandmw2il": dict(band_kernel=1, band_movers=2, batch_layout=1), "bandmw2il2": dict(band_kernel=1, band_movers=2, batch_layout=1, band_rhs_interleaved=1)}
if os.environ.get("BAND_VARIANTS"):
    variants = {k: v for k, v in variants.items() if k in os.environ["BAND_VARIANTS"].split(",")}
for B in [int(a) for a in sys.argv[1:]] or [8192]:
    vh, rh = bench.band_batch(s, min(B, 256), 3000)
    rep = max(1, (B + len(vh) - 1) // len(vh))
    vals = torch.from_numpy(np.tile(vh, (rep, 1))[:B]).to(dev)
    rhs = torch.from_numpy(np.tile(rh, (rep, 1))[:B]).to(dev)
    d = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    ro = torch.zeros(B, dtype=torch.float64, device=dev)
    rho = torch.zeros_like(ro)
    nf = torch.zeros(B, dtype=torch.int32, device=dev)
    su = torch.zeros_like(nf)
    st = torch.cuda.Stream()
    out = {"B": B, "n": n, "ncon": p}
    dref = None
    for name, opt in variants.items():
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, **opt))
        d.zero_()

        vin, rin = vals, rhs
        if L.config.get("batch_layout"):
            vin = torch.empty(hipldl.layout_len(L, 0), dtype=torch.float64, device=dev)
            hipldl.interleave_dev(L, 0, vals.data_ptr(), vin.data_ptr(), 0)
            if L.config.get("rhs_interleaved"):
                rin = torch.empty(hipldl.layout_len(L, 1), dtype=torch.float64, device=dev)
                hipldl.interleave_dev(L, 1, rhs.data_ptr(), rin.data_ptr(), 0)
            torch.cuda.synchronize()

        def step():
            hipldl.newton_system_dev(L, vin.data_ptr(), rin.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), par, st.cuda_stream)
        with torch.cuda.stream(st):
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(10):
                    step()
                e1.record(st)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
        ok = bool((su == 1).all())
        if dref is None:
            dref = d.clone()
            dev_rel = 0.0
        else:
            dev_rel = float((d - dref).abs().max() / dref.abs().max())
        out[name] = {"ms": min(ts), "ksys_per_s": B / min(ts), "all_ok": ok, "band": L.config.get("band"), "max_rel_dev_from_first": dev_rel}
        L.close()
    print(json.dumps(out), flush=True)
