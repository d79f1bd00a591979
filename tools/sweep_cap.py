"""Task cap of the latency plans against batch size (dataflow execution): python tools/sweep_cap.py"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in (1, 16, 256, 1024):
    for cap in (1, 2, 4, 6, 8, 12, 16):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "40", "--cpu-sample", "0", "--no-extras", "--opt", f"task_cap={cap}"], capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print("B", B, "cap", cap, "systems/s %.0f" % j["value"], "ms/step %.4f" % j["ms_per_step"], j["config"]["fronts"], j["config"]["ordering"], flush=True)
        except Exception:
            print(B, cap, "ERR", out.stderr[-200:], flush=True)
