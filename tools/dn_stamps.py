import os, sys, numpy as np, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
s = syn.dense_structure(1000, 2000); rows, cols = s.kkt_pattern()
v, r = syn.dense_values(s, 2002)
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=1)
p = hipldl.default_params()
for _ in range(3):
    ok = hipldl.try_to_factorize(L, v, s.nvar, s.nequ, s.ncon, 2.2e-16)
# read stamps: jxp pointer unknown from python -> the library prints them when CNL_DN_STAMPS is set
L.close()
