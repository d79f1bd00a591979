"""Staged (latency plan, launch per stage) against the classic single stream around the switch-over batch size."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in (1024, 2048, 3072, 4096, 5120, 6144):
    for mode, opt in (("staged", "plan_kind=2"), ("classic", "plan_kind=1")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "10", "--cpu-sample", "0", "--no-extras", "--opt", opt], capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print("B", B, mode, "systems/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], j["config"]["ordering"], flush=True)
        except Exception:
            print(B, mode, "ERR", out.stderr[-200:], flush=True)
