#!/bin/bash
# Address-translation counters of the headline step (run through gpurun from the repository root):
#   gpurun --timeout 900 -- 'bash tools/profile_tlb.sh r02'
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum --output-format csv -d $OUT/tlb1 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-extras > $OUT/${R}_tlb1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --output-format csv -d $OUT/tlb2 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-extras > $OUT/${R}_tlb2.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ("tlb1", "tlb2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
        for k in acc:
            if "newton2" in k or "expand" in k:
                print(k, {c: v / n[(k, c)] for c, v in acc[k].items()})
PY
