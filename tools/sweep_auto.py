"""Throughput against batch size with the plan cnl_create chooses by itself."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 4, 16, 64, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 5120, 8192)):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "10", "--cpu-sample", "0", "--no-extras"], capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("B", B, "systems/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], j["config"]["ordering"], j["config"]["fronts"], j["config"]["kernel"]["kernel"], "ok", j["config"]["all_success"], "%.1e" % j["config"]["backward_error"], flush=True)
    except Exception:
        print(B, "ERR", out.stderr[-300:], flush=True)
