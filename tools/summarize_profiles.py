#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*) into the small tracked files under profiles/:
<tag>_kernel_stats.csv (per-kernel durations), <tag>_pmc.csv (per-kernel counter averages per dispatch) and, when --batch /
--workload are given, <tag>_traffic.json (HBM bytes per launch of the dominant kernel as bench.py reads it)."""
import argparse, csv, glob, json, os, sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("tag"); ap.add_argument("--batch", type=int); ap.add_argument("--n", type=int, default=10000); ap.add_argument("--ncon", type=int, default=50)
ap.add_argument("--kernel", default="newton2_kernel")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
short = lambda n: n.replace("cnl::(anonymous namespace)::", "").replace("cnl::", "")
st = glob.glob(os.path.join(go, a.tag + "_stats", "*kernel_stats.csv"))
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(pr, a.tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if "cnl::" in r["Name"] or float(r["Percentage"]) > 0.5:
                w.writerow([short(r["Name"])[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
acc = defaultdict(lambda: defaultdict(list))
meta = {}
for d in sorted(glob.glob(os.path.join(go, a.tag + "_pmc_*"))):
    if not os.path.isdir(d): continue
    for fn in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            if "cnl::" not in r["Kernel_Name"]: continue
            k = short(r["Kernel_Name"]).split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            # (rocprofv3's LDS_Block_Size / Scratch_Size / VGPR_Count / SGPR_Count columns are NOT carried over: for these kernels they
            #  read 0 / 0 / 96 / 112 whatever the code object holds — dynamic LDS is not in the descriptor, and the VGPR figure is not
            #  the allocation.  The registers, spills and scratch of every instantiation are in profiles/<tag>_kernel_resources.txt,
            #  written by tools/kernel_resources.sh from the compiler's own resource-usage remarks.)
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"])
if acc:
    with open(os.path.join(pr, a.tag + "_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Kernel", "Grid_Size", "Workgroup_Size", "Counter", "Dispatches", "Mean_per_dispatch"])
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]; w.writerow([k, *meta[k], c, len(v), sum(v) / len(v)])
    kk = [k for k in acc if a.kernel in k]
    if kk and a.batch:
        k = max(kk, key=lambda q: sum(acc[q].get("FETCH_SIZE", [0])))
        fs = acc[k].get("FETCH_SIZE", []); ws = acc[k].get("WRITE_SIZE", [])
        if fs and ws:
            fkb, wkb = sum(fs) / len(fs), sum(ws) / len(ws)
            json.dump({"round": a.tag, "batch": a.batch, "workload": [a.n, a.ncon], "kernel": k, "FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb,
                       "hbm_bytes_per_launch": int((2 * fkb + wkb) * 1024),
                       "correction": "hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM); the factor is "
                                     "calibrated for 16-byte-per-lane streams only, this kernel gathers 8 bytes per lane, so the read side is an upper estimate",
                       "raw_sum_bytes": int((fkb + wkb) * 1024), "source": f"profiles/{a.tag}_pmc.csv"}, open(os.path.join(pr, a.tag + "_traffic.json"), "w"), indent=1)
print("written:", [f for f in os.listdir(pr) if f.startswith(a.tag)])
