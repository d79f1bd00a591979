"""Row f2 (cnl_prepare_newton_system_dev) on the headline pattern: problem-major `vals` against CNL_LAYOUT_INTERLEAVED, same process.
usage: time_prepare.py [B ...]   one JSON line per batch size (ms, fraction of 8 TB/s on the bytes the pass has to move)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa: F401,E402
from cannoles_jl_amd import hipldl, synthetic as syn  # noqa: E402

s = syn.band_structure(int(os.environ.get("BAND_N", 10000)), int(os.environ.get("BAND_P", 50)))
rows, cols = s.kkt_pattern()
dev = torch.device("cuda", 0)
nhF, nhc, njF, njc = len(s.hF[0]), len(s.hc[0]), len(s.jF[0]), len(s.jc[0])
by = 8 * (nhF + nhc + njF + njc + (s.nnzNS - s.nequ))
for B in [int(a) for a in sys.argv[1:]] or [4096]:
    mk = lambda n_: torch.randn((B, max(n_, 1)), dtype=torch.float64, device=dev)   # noqa: E731
    a_hF, a_hc, a_Jx, a_Jc = mk(nhF), mk(nhc), mk(njF), mk(njc)
    a_de = torch.full((B,), 1e-8, dtype=torch.float64, device=dev)
    out = {"B": B, "bytes_per_system": by}
    res = {}
    for name, lay in (("problem_major", 0), ("interleaved", 1)):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, batch_layout=lay))
        v = torch.zeros(hipldl.layout_len(L, 0) if lay else B * s.nnzNS, dtype=torch.float64, device=dev)

        def go():
            hipldl.prepare_newton_system_dev(L, nhF, nhc, njF, njc, a_hF.data_ptr(), a_hc.data_ptr() if s.ncon else 0, a_Jx.data_ptr(),
                                             a_Jc.data_ptr() if s.ncon else 0, a_de.data_ptr() if s.ncon else 0, v.data_ptr(), 0)
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if lay:
            w = torch.zeros(B * s.nnzNS, dtype=torch.float64, device=dev)
            hipldl.deinterleave_dev(L, 0, v.data_ptr(), w.data_ptr(), 0)
            torch.cuda.synchronize()
            res[name] = w
        else:
            res[name] = v
        out[name] = {"ms": ms, "frac": by * B / (ms * 1e-3) / 8e12}
        L.close()
    out["equal"] = bool(torch.equal(res["problem_major"], res["interleaved"]))
    print(json.dumps(out), flush=True)
    del res, a_hF, a_hc, a_Jx, a_Jc
