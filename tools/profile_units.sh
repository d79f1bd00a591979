#!/bin/bash
# Which unit of a CU does the dominant kernel keep busy?  Busy cycles per instruction class (SQ_ACTIVE_INST_*), issue cycles of the
# memory classes, FIFO-full events of the vector-memory path; PMC passes of their own, each under a timeout.
# usage: tools/profile_units.sh <tag> [bench args...]   -> gpurun_out/<tag>_units_*/ and a table on stdout
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 4 --warmup 2 --cpu-sample 0 --no-extras $*"
i=0
for pass in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_CYCLES" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_ATOMIC"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/${tag}_units_$i -o p -- python3 $root/bench.py $args > $out/${tag}_units_$i.log 2>&1
  tail -1 $out/${tag}_units_$i.log | cut -c1-160
done
python3 - <<PY
import csv, glob, os, collections, json
res = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join("$out", "${tag}_units_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f)):
        if "newton2_kernel_t<false, false, true, false, false>" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    for k, v in acc.items(): res[k] = v / max(1, len(disp))
for k, v in res.items(): print("%-34s %.4g" % (k, v))
json.dump(res, open(os.path.join("$out", "${tag}_units.json"), "w"), indent=1)
PY
