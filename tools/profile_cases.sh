#!/bin/bash
# rocprofv3 --kernel-trace --stats of the secondary workloads of the bench line (gpurun from the repo root):
#   staged execution (B = 256, B = 1), split batch (B = 5120), BASELINE config 4 (B = 256, 4096), the dense backend (cfg2,
#   B = 1 and 8) and the irregular family — every fraction in the bench line has a tracked summary under profiles/.
# usage: tools/profile_cases.sh <round tag>   -> gpurun_out/<tag>_<case>_stats/ ; then tools/summarize_profiles.py <tag>_<case>
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, program, args...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${name}_stats -o p -- python3 "$@" > $out/${tag}_${name}_stats.log 2>&1
}
B="$root/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extras"
run staged_B256 $B --batch 256
run staged_B1 $B --batch 1
run split_B5120 $B --batch 5120
run cfg4_B256 $B --nvar 1000 --ncon 10 --batch 256
run cfg4_B4096 $B --nvar 1000 --ncon 10 --batch 4096
run dense_B1 $root/tools/bench_dense.py --batch 1
run dense_B8 $root/tools/bench_dense.py --batch 8
run irregular $root/tools/time_irregular.py
run cfg5_B256 $root/tools/time_dev_ladder.py behind
ls $out | grep ${tag}_ | head -40
