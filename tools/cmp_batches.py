"""Headline rates of library variants at several batch sizes: python tools/cmp_batches.py B1,B2,... VARIANT..."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
import ablate
for v in sys.argv[2:]:
    for B in [int(x) for x in sys.argv[1].split(",")]:
        e = dict(os.environ, CANNOLES_HIP_LIB=ablate.libpath(v))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "10", "--cpu-sample", "0", "--no-extras", "--opt", "plan_kind=1"], env=e, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(v, "B", B, "systems/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], "kernel ms %.3f" % j["roofline"]["kernel_ms"], j["config"]["kernel"]["lds2_bytes"], flush=True)
        except Exception:
            print(v, B, "ERR", out.stderr[-300:], flush=True)
