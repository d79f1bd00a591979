#!/usr/bin/env python3
"""profiles/<tag>_bench.json (+ _bench_under_rocprof.json) from the bench lines of a tools/profile_round5.sh call, and the consistency
check of VERDICT r4 item 5: the rocprofv3 average duration of the dominant kernel must agree with the SAME call's bench (kernel_ms of
the unprofiled line) within 3 %, else the profile is refused (exit 1).  usage: check_profile.py <tag> [kernel substring]"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "band_newton"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
lines = [ln for ln in open(os.path.join(go, tag + "_bench.txt")) if ln.startswith("{")]
long_line, compact = json.loads(lines[0]), json.loads(lines[-1])
json.dump(long_line, open(os.path.join(pr, tag + "_bench.json"), "w"), indent=1)
json.dump(compact, open(os.path.join(pr, tag + "_bench_compact.json"), "w"), indent=1)
prof = [json.loads(ln) for ln in open(os.path.join(go, tag + "_stats.log")) if ln.startswith("{")]
if prof:
    json.dump(prof[0], open(os.path.join(pr, tag + "_bench_under_rocprof.json"), "w"), indent=1)
shutil.copy(os.path.join(go, tag + "_clocks.txt"), os.path.join(pr, tag + "_clocks.txt"))
st = glob.glob(os.path.join(go, tag + "_stats", "*kernel_stats.csv"))
rows = [r for r in csv.DictReader(open(st[0])) if kern in r["Name"]]
r = max(rows, key=lambda q: float(q["TotalDurationNs"]))
avg_ms, min_ms = float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6
bench_ms = long_line["roofline"]["kernel_ms"]
ratio = avg_ms / bench_ms
b_alg, units = long_line["roofline"]["bytes_per_system"], long_line["roofline"]["units_per_launch"]
res = {"tag": tag, "kernel": r["Name"][:100], "calls": int(r["Calls"]), "rocprof_avg_ms": avg_ms, "rocprof_min_ms": min_ms, "bench_kernel_ms": bench_ms,
       "ratio_rocprof_over_bench": ratio, "within_3_percent": abs(ratio - 1.0) <= 0.03,
       "frac_from_rocprof_avg": b_alg * units / (avg_ms * 1e-3) / 1e9 / 8000.0, "frac_in_bench_line": long_line["roofline"]["frac"],
       "same_gpurun_call": True}
json.dump(res, open(os.path.join(pr, tag + "_profile_check.json"), "w"), indent=1)
print(json.dumps(res))
sys.exit(0 if res["within_3_percent"] else 1)
