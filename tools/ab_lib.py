#!/usr/bin/env python3
"""A/B of whole-library builds on ONE box (boxes of the pool differ by +-4 %, so only same-box pairs mean anything).
  build (CPU):  python tools/ab_lib.py build <name> <git-ref>|WORK [extra compiler flags]   -> build_abl/ab_<name>/libcannoles_hip.so
  run (GPU):    python tools/ab_lib.py run <B> <name> [<name> ...]    interleaved rounds, kernel ms of the headline step (cfg3 pattern)
WORK = the working tree as it is.  Product builds only (no -D probes): the libraries are loaded through CANNOLES_HIP_LIB."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTD = os.path.join(ROOT, "build_abl")


def build(name, ref, flags):
    d = os.path.join(OUTD, "ab_" + name)
    src = os.path.join(d, "src")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(src)
    if ref == "WORK":
        shutil.copytree(os.path.join(ROOT, "cannoles.jl_amd", "csrc"), os.path.join(src, "cannoles.jl_amd", "csrc"), ignore=shutil.ignore_patterns("build"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(src, "include"))
    else:
        tar = subprocess.run(["git", "-C", ROOT, "archive", ref, "cannoles.jl_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", src], input=tar, check=True)
    csrc = os.path.join(src, "cannoles.jl_amd", "csrc")
    srcs = [f for f in os.listdir(csrc) if f.endswith((".cpp", ".hip"))]
    objs = []
    procs = []
    for f in srcs:
        o = os.path.join(d, f + ".o")
        objs.append(o)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", *flags, "-c", "-o", o, f], cwd=csrc))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", os.path.join(d, "libcannoles_hip.so"), *objs])
    shutil.rmtree(src)
    for o in objs:
        os.remove(o)
    print("built", os.path.join(d, "libcannoles_hip.so"))


RUN = r'''
import sys; sys.path.insert(0, %(root)r)
import numpy as np, torch
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
import bench
s = syn.band_structure(%(n)d, %(p)d); rows, cols = s.kkt_pattern()
B = %(B)d
vh, rh = bench.band_batch(s, min(B, 512), 3000)
dev = torch.device("cuda", 0)
rep = max(1, B // 512)
vals = torch.from_numpy(np.tile(vh, (rep, 1))[:B]).to(dev); rhs = torch.from_numpy(np.tile(rh, (rep, 1))[:B]).to(dev)
d = torch.zeros((B, s.N), dtype=torch.float64, device=dev); ro = torch.zeros(B, dtype=torch.float64, device=dev); rho = torch.zeros_like(ro)
nf = torch.zeros(B, dtype=torch.int32, device=dev); su = torch.zeros_like(nf)
L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
p = hipldl.default_params()
st = torch.cuda.Stream()
def step():
    hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), su.data_ptr(), p, st.cuda_stream)
with torch.cuda.stream(st):
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(%(reps)d): step()
    e1.record(st); torch.cuda.synchronize()
print("RESULT %%.5f %%d" %% (e0.elapsed_time(e1) / %(reps)d, int((su == 1).all())))
'''


def run(B, names, n=10000, p=50, rounds=3):
    res = {k: [] for k in names}
    reps = 20 if B >= 1024 else 100
    for r in range(rounds):
        for k in names:
            lib = os.path.join(OUTD, "ab_" + k, "libcannoles_hip.so")
            env = dict(os.environ, CANNOLES_HIP_LIB=lib, CANNOLES_HIP_ALLOW_EXPERIMENT="1")
            out = subprocess.run([sys.executable, "-c", RUN % {"root": ROOT, "B": B, "n": n, "p": p, "reps": reps}], env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
            if not line:
                print(k, "FAILED", out.stderr[-500:])
                continue
            ms, ok = line[0].split()[1:]
            res[k].append(float(ms))
            print(f"round {r} {k:12s} B={B} {float(ms):.4f} ms/step  {B / float(ms):.1f} k systems/s ok={ok}", flush=True)
    for k in names:
        if res[k]:
            print(f"{k:12s} min {min(res[k]):.4f} median {sorted(res[k])[len(res[k]) // 2]:.4f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        a = sys.argv[2:]
        n, p = 10000, 50
        if "--cfg4" in a:
            a.remove("--cfg4"); n, p = 1000, 10
        run(int(a[0]), a[1:], n=n, p=p)
