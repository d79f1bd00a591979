"""Staged (latency plan, launch per stage) against the classic single stream around the switch-over batch size."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in (1024, 2048, 3072, 4096, 5120, 6144):
    for mode, env in (("staged", {"CNL_STAGED_MAX": "100000"}), ("classic", {"CNL_STAGED_MAX": "0"})):
        e = dict(os.environ, **env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "10", "--cpu-sample", "0", "--no-extras"], env=e, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print("B", B, mode, "systems/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], j["config"]["ordering"], flush=True)
        except Exception:
            print(B, mode, "ERR", out.stderr[-200:], flush=True)
