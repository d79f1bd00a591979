#!/bin/bash
# ONE gpurun call = one box: clocks / power before, the unprofiled bench (the tracked line), rocprofv3 kernel statistics of the same
# command, the PMC passes (separate runs: gpurun refuses --pmc with other traces), clocks / power after.  The summariser
# (tools/summarize_profiles.py <tag> --kernel band_newton --batch B, then tools/check_profile.py <tag>) refuses a profile whose kernel
# time differs from the same call's bench by more than 3 %.
# usage (through gpurun, from the repo root): tools/profile_round5.sh <tag> [bench args...]
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
smi() { rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -E "GPU\[0\]|sclk|mclk|fclk|socclk|Power|Temperature|Performance" | head -20; }
{ echo "== before"; date -u +%FT%TZ; smi; } > $out/${tag}_clocks.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $root/bench.py --steps 20 --warmup 3 $* > $out/${tag}_bench.txt 2> $out/${tag}_bench.err
{ echo "== after the unprofiled bench"; date -u +%FT%TZ; smi; } >> $out/${tag}_clocks.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o p -- python3 $root/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-extras $* > $out/${tag}_stats.log 2>&1
{ echo "== after the kernel-trace pass"; date -u +%FT%TZ; smi; } >> $out/${tag}_clocks.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/${tag}_pmc_$name -o p -- python3 $root/bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-extras $* > $out/${tag}_pmc_$name.log 2>&1
done
{ echo "== after the PMC passes"; date -u +%FT%TZ; smi; } >> $out/${tag}_clocks.txt
tail -n 1 $out/${tag}_bench.txt | cut -c1-400
cat $out/${tag}_clocks.txt | head -60
ls $out | grep $tag
