#!/bin/bash
# rocprofv3 passes of the bench command on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats, then PMC counters in passes of their own (gpurun refuses --pmc together with other traces).
# usage: tools/profile_round.sh <tag> [bench args...]   -> gpurun_out/<tag>_*/ ; summarise with tools/summarize_profiles.py
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 6 --warmup 2 --cpu-sample 0 --no-extras $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o p -- python3 $root/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extras $* > $out/${tag}_stats.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/${tag}_pmc_$name -o p -- python3 $root/bench.py $args > $out/${tag}_pmc_$name.log 2>&1
done
ls $out | grep $tag
