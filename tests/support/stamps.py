"""Diagnostic: build the library with -DCNL_STAMPS and print the per-phase cycle shares of the
register-front kernel (not part of the test-suite; run on the GPU box)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "cannoles.jl_amd", "csrc")
if not os.environ.get("CANNOLES_HIP_LIB"):  # otherwise: a variant built beforehand (tests/support/ablate.py build CNL_STAMPS=1,...)
    # an EXPERIMENT build of its own (never over the product library): build_abl/libcnl_stamps.so
    outd = os.path.join(ROOT, "build_abl")
    os.makedirs(outd, exist_ok=True)
    subprocess.check_call(["make", "-s", "-j4", "-C", csrc, f"OUT={outd}/libcnl_stamps.so", f"OBJDIR={outd}/stamps_obj",
                           "CXXFLAGS=-O3 -std=c++17 -fPIC -DCNL_STAMPS -DCNL_EXPERIMENT=1"])
    os.environ["CANNOLES_HIP_LIB"] = os.path.join(outd, "libcnl_stamps.so")
os.environ.setdefault("CANNOLES_HIP_ALLOW_EXPERIMENT", "1")
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn

n, p, B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 50, int(sys.argv[3]) if len(sys.argv) > 3 else 256
s = syn.band_structure(n, p)
rows, cols = s.kkt_pattern()
import bench
vals, rhs = bench.band_batch(s, 8, 3000)   # the bench's values: no rho ladder
vals = np.tile(vals, (B // 8, 1)); rhs = np.tile(rhs, (B // 8, 1))
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
d = np.zeros((B, s.N))
prm = hipldl.default_params()
L.set_timing(True)
for it in range(3):
    _, ok_, _, _, nf_ = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B), prm)
print("kernel ms", L.last_kernel_ms(), "success", ok_.all(), "nfact max", nf_.max())
lib = hipldl.lib()
nw = B // 4
out = np.zeros(nw * 8, np.int64)
lib.cnl_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.cnl_debug_stamps(L._h, out.ctypes.data, out.size) == 0
st = out.reshape(nw, 8).astype(float)
if os.environ.get("CNL_STAMP_NAMES") == "backward":   # library built with -DCNL_STAMPS=2
    names = ["b:wait vmcnt", "b:issue prefetch", "b:idx,x,row reads", "b:chain", "b:stores+rotate", "b:loop head", "(unused)", "forward+ladder"]
else:
  names = ["zero+asm", "rec+prefetch", "extend-add", "eliminate(all)", "sync", "el:pivots(+ladder)", "backward", "el:loads"]
tot = st.sum(axis=1).mean()
print(L.info)
print("cycles per wave (s_memtime ticks): total %.3e" % tot)
for k, nm in enumerate(names):
    print("  %-14s %10.3e  %5.1f%%  per front %8.0f" % (nm, st[:, k].mean(), 100 * st[:, k].mean() / tot, st[:, k].mean() / L.info["nsuper"]))
