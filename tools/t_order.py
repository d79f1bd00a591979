import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
mode = sys.argv[1]
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
s = syn.band_structure(200, 4); rows, cols = s.kkt_pattern()
def mk(B=2):
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B); L.close(); return True
import torch
if mode == "lib_first":
    print("lib", mk()); x = torch.zeros(10, device="cuda"); print("torch ok", x.sum().item()); print("lib again", mk())
elif mode == "torch_first":
    x = torch.zeros(10, device="cuda"); print("torch ok"); print("lib", mk())
elif mode == "torch_big_first":
    x = torch.empty((4608, 120041), dtype=torch.float64, device="cuda"); print("torch ok"); print("lib", mk(4608))
