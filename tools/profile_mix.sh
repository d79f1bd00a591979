#!/bin/bash
# Instruction mix of the headline step: where the issue cycles of the dominant kernel go (gpurun -- 'bash tools/profile_mix.sh')
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU" \
            "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/mix$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 > $OUT/mix$i.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/mix*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "newton2" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
waves, fronts = 2048, 1000
for c in sorted(acc):
    v = acc[c] / n[c]
    print("%-28s %14.4g per launch   %10.1f per front and wave" % (c, v, v / waves / fronts))
PY
