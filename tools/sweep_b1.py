"""One system: task cap and nested-dissection leaf size of the latency plan under the dataflow execution."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for order in ("nd16+early", "nd32+early", "nd64+early", "nd96+early"):
    for cap in (2, 3, 4, 6, 8, 12, 16):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "1", "--steps", "60", "--cpu-sample", "0", "--no-extras", "--opt", f"task_cap={cap},force_order={order}"], capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(order, "cap", cap, "ms/step %.4f" % j["ms_per_step"], j["config"]["fronts"], flush=True)
        except Exception:
            print(order, cap, "ERR", out.stderr[-200:], flush=True)
