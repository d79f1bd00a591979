import os, sys, json, subprocess
ROOT = os.getcwd()
for B in (256, 64, 32):
    for name in ("", "ndc4+early", "ndc8+early", "ndc16+early", "ndc32+early", "nd16+early", "nd64+early", "nd128+early"):
        opt = f"plan_kind=2,force_order={name}" if name else ""
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "40", "--warmup", "5", "--cpu-sample", "0", "--no-extras", "--nvar", "1000", "--ncon", "10"] + (["--opt", opt] if opt else [])
        out = subprocess.run(cmd, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print("B", B, name or "(auto)", "systems/s %.0f" % j["value"], "ms/step %.4f" % j["ms_per_step"], j["config"]["ordering"], j["config"]["fronts"], flush=True)
        except Exception:
            print(B, name, "ERR", out.stderr[-200:], flush=True)
