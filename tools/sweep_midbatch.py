"""Mid-size batches: latency plans with a few LARGE canonical parts run as whole-subtree tasks (candidates ndc<K>+early)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [(B, f"ndc{k}+early") for B in (512, 768, 1024, 1536, 2048, 2560) for k in (2, 4, 8, 16, 32)] if len(sys.argv) < 2 else [(int(a.split(":")[0]), a.split(":")[1]) for a in sys.argv[1:]]
for B, name in cases:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "10", "--cpu-sample", "0", "--no-extras", "--opt", f"plan_kind=2,force_order={name}"], capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("B", B, name, "systems/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], j["config"]["ordering"], j["config"]["fronts"], j["config"]["kernel"]["kernel"], "ok", j["config"]["all_success"], "%.1e" % j["config"]["backward_error"], flush=True)
    except Exception:
        print(B, name, "ERR", out.stderr[-300:], flush=True)
