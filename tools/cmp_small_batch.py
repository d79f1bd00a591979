"""Small-batch rates of library variants (tests/support/ablate.py build NAME=VALUE ...): python tools/cmp_small_batch.py VARIANT..."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
import ablate
for v in sys.argv[1:]:
    for B in (1, 256, 2048, 8192):
        e = dict(os.environ, CANNOLES_HIP_LIB=ablate.libpath(v))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "20", "--cpu-sample", "0", "--no-extras"], env=e, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(v, "B", B, "systems/s %.0f" % j["value"], "ms/step %.4f" % j["ms_per_step"], j["config"]["kernel"]["kernel"], flush=True)
        except Exception:
            print(v, B, "ERR", out.stderr[-300:], flush=True)
