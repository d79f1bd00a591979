#!/bin/bash
# Vector-memory counters of the bench command's dominant kernel (run through gpurun from the repo root): instructions, L1 (TCP)
# line requests, L1 -> L2 requests, TA/TD busy.  PMC passes of their own (no other trace domains).
# usage: tools/profile_vmem.sh <tag> [bench args...]   -> gpurun_out/<tag>_vmem_*/
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 4 --warmup 2 --cpu-sample 0 --no-extras $*"
timeout 120 rocprofv3 -L > $out/${tag}_counters_avail.txt 2>&1
for pass in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_FLAT" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  # (a pass whose counters the profiler rejects can hang until the box limit: every pass under its own timeout.  The TA_FLAT_*
  #  counters abort rocprofv3 on this image and are left out)
  timeout 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/${tag}_vmem_$name -o p -- python3 $root/bench.py $args > $out/${tag}_vmem_$name.log 2>&1
  tail -2 $out/${tag}_vmem_$name.log
done
python3 - <<PY
import csv, glob, os, collections
out = "$out"
tot = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "${tag}_vmem_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "newton2_kernel_t<false, false, true, false, false>" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        # rows are per dispatch and per dimension instance; n / dispatches = instances
        tot[k] = (v, n)
disp = None
for k, (v, n) in tot.items():
    print(k, "sum", v, "rows", n)
PY
