"""Timing experiment (not part of the test-suite; run on the GPU box): kernel time of variants of the
register-front kernel built with -DCNL_ABL=<bits> (pieces of the hot path removed, results wrong).
Build the variants first (CPU):  python tests/support/ablate.py build 1 2 4 ...
Run on the GPU:                  python tests/support/ablate.py run 8192 1 2 4 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUTD = os.path.join(ROOT, "build_abl")
CSRC = os.path.join(ROOT, "cannoles.jl_amd", "csrc")


def libpath(bits):
    """a variant is a number (CNL_ABL bits) or NAME=VALUE[,NAME=VALUE...] (arbitrary -D flags)"""
    return os.path.join(OUTD, "libcnl_abl" + str(bits).replace("=", "-").replace(",", "_").replace(":", "_").replace("+", "_") + ".so")


def flags(bits):
    # every variant is an EXPERIMENT build (csrc/kernels2.hip refuses the probes otherwise; cnl_version() turns negative)
    if str(bits).startswith("RAW:"):   # raw compiler flags, '+' separated: RAW:-mllvm+-amdgpu-...=1
        return "-DCNL_EXPERIMENT=1 " + " ".join(str(bits)[4:].split("+"))
    if "=" in str(bits):
        return "-DCNL_EXPERIMENT=1 " + " ".join("-D" + f for f in str(bits).split(","))
    return f"-DCNL_EXPERIMENT=1 -DCNL_ABL={bits}"


def build(bits_list):
    os.makedirs(OUTD, exist_ok=True)
    procs = []
    for bits in bits_list:
        cmd = ["make", "-s", "-j4", "-C", CSRC, f"OUT={libpath(bits)}", f"OBJDIR={libpath(bits)[:-3]}_obj", f"CXXFLAGS=-O3 -std=c++17 -fPIC {flags(bits)}"]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        p.wait()


def run_one(B, bits):
    env = dict(os.environ, CANNOLES_HIP_LIB=libpath(bits), CANNOLES_HIP_ALLOW_EXPERIMENT="1")
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import numpy as np, torch
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
import bench
s = syn.band_structure(10000, 50); rows, cols = s.kkt_pattern()
B = {B}
vh, rh = bench.band_batch(s, 512, 3000)
dev = torch.device("cuda", 0)
vals = torch.from_numpy(np.tile(vh, (B // 512, 1))).to(dev); rhs = torch.from_numpy(np.tile(rh, (B // 512, 1))).to(dev)
d = torch.zeros((B, s.N), dtype=torch.float64, device=dev); ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
nf = torch.zeros(B, dtype=torch.int32, device=dev); su = torch.zeros_like(nf)
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
L.set_timing(True)
p = hipldl.default_params()
ms = []
for it in range(4):
    ro.zero_()
    hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p, 0)
    torch.cuda.synchronize(); ms.append(L.last_kernel_ms())
print("B", B, "ablate", {str(bits)!r}, "kernel ms", np.round(ms[1:], 2))
"""
    subprocess.run([sys.executable, "-c", code], env=env)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        B = int(sys.argv[2])
        for bits in sys.argv[3:]:
            run_one(B, bits)
