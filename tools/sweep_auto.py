"""Throughput against batch size with the plan cnl_create chooses by itself (bench.py --batch B --no-extras per point, one box).
usage: sweep_auto.py [batches, comma separated]     writes gpurun_out/batch_sweep.json"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = (64, 128, 256, 512, 768, 1024, 1536, 2048, 2560, 3072, 4096, 4352, 4608, 5120, 6144, 7168, 8192, 8448, 10240, 12288, 16384)
points = []
for B in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else DEFAULT):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "10", "--cpu-sample", "0", "--no-extras"], capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        c = j["config"]
        points.append({"batch": B, "systems_per_s": round(j["value"]), "ms_per_step": round(j["ms_per_step"], 3), "ordering": c["ordering"], "fronts": c["fronts"],
                       "kernel": c["kernel"]["kernel"], "remainder_handle": bool(c["kernel"].get("tail")), "all_success": c["all_success"],
                       "backward_error": float("%.1e" % c["backward_error"])})
        print("B", B, "systems/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], c["ordering"], c["fronts"], c["kernel"]["kernel"], "tail" if c["kernel"].get("tail") else "", "ok", c["all_success"], "%.1e" % c["backward_error"], flush=True)
    except Exception:
        print(B, "ERR", out.stderr[-300:], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"workload": "cfg3 pattern n=nequ=1e4 ncon=50, plans as cnl_create picks them (tools/sweep_auto.py, one box, one gpurun call)", "points": points},
          open(os.path.join(ROOT, "gpurun_out", "batch_sweep.json"), "w"), indent=1)
