"""Mid-size batches: latency plans with a few LARGE canonical parts run as whole-subtree tasks (candidates ndc<K>+early)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [(512, "ndc16+early"), (512, "ndc32+early"), (768, "ndc16+early"), (1024, "ndc16+early"), (1536, "ndc8+early"), (5120, "ndc2+early"), (6144, "ndc2+early"), (256, "ndc16+early"), (256, "ndc32+early")]
for B, name in cases:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "10", "--cpu-sample", "0", "--no-extras", "--opt", f"plan_kind=2,force_order={name}"], capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("B", B, name, "systems/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], j["config"]["ordering"], j["config"]["fronts"], j["config"]["kernel"]["kernel"], "ok", j["config"]["all_success"], "%.1e" % j["config"]["backward_error"], flush=True)
    except Exception:
        print(B, name, "ERR", out.stderr[-300:], flush=True)
