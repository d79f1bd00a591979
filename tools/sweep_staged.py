import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(B, opt):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "20", "--cpu-sample", "0", "--no-extras", "--opt", opt], capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        return j["value"], j["ms_per_step"], j["config"]["ordering"], j["config"]["fronts"], j["config"]["kernel"]["kernel"]
    except Exception as ex:
        return ("ERR", out.stderr[-300:])
for B in (1, 64, 256):
    for cap in (8, 12, 20, 32, 48):
        print(B, "cap", cap, run(B, f"task_cap={cap}"), flush=True)
for B in (512, 1024, 2048, 4096):
    print(B, "staged", run(B, "plan_kind=2"), flush=True)
    print(B, "classic", run(B, "plan_kind=1"), flush=True)
