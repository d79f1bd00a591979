"""Import-order probe (tests/test_gpu_parity.py::test_import_order_does_not_matter): `python tools/rt_probe.py torch_first|lib_first|no_torch`
loads PyTorch and libcannoles_hip.so in the given order, runs one device-pointer newton_system! on a torch stream (no_torch: a
host-pointer call) and prints `RUNTIMES <n>` (HIP runtimes mapped into the process) and `OK` on success."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == "torch_first":
    import torch
    torch.zeros(1, device="cuda")
import cannoles_jl_amd  # noqa: F401,E402
from cannoles_jl_amd import hipldl, synthetic as syn  # noqa: E402
hipldl.lib()
if order == "lib_first":
    import torch
    torch.zeros(1, device="cuda")
import numpy as np  # noqa: E402

s = syn.band_structure(200, 4)
rows, cols = s.kkt_pattern()
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=4)
vals, rhs = syn.batch_values(s, 4, cfg=4)
if order == "no_torch":
    assert "torch" not in sys.modules
    d = np.zeros((4, s.N))
    out = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(4), hipldl.default_params())
    ok, dmax = bool(np.all(out[1])), float(np.abs(d).max())
else:
    st = torch.cuda.Stream()
    tv, tr = torch.tensor(vals, device="cuda"), torch.tensor(rhs, device="cuda")
    td = torch.zeros((4, s.N), dtype=torch.float64, device="cuda")
    ro, rho = torch.zeros(4, dtype=torch.float64, device="cuda"), torch.zeros(4, dtype=torch.float64, device="cuda")
    nf, su = torch.zeros(4, dtype=torch.int32, device="cuda"), torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hipldl.newton_system_dev(L, tv.data_ptr(), tr.data_ptr(), td.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(),
                             hipldl.default_params(), st.cuda_stream)
    torch.cuda.synchronize()
    ok, dmax = bool((su == 1).all()), float(td.abs().max())
print("RUNTIMES", len(hipldl.loaded_hip_runtimes()), hipldl.loaded_hip_runtimes())
print("OK" if ok and 0.1 < dmax < 100 else "FAILED", dmax)
