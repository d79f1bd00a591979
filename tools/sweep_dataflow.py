"""Dataflow (one launch per phase, dependency counters) against one launch per stage: python tools/sweep_dataflow.py"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in (1, 4, 8, 16, 64):
    for nodf in ("0", "1"):
        opt = f"dataflow={1 - int(nodf)},dataflow_waves=100000000"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "40", "--cpu-sample", "0", "--no-extras", "--opt", opt], capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print("B", B, "dataflow" if nodf == "0" else "per stage", "systems/s %.0f" % j["value"], "ms/step %.4f" % j["ms_per_step"], flush=True)
        except Exception:
            print(B, nodf, "ERR", out.stderr[-200:], flush=True)
