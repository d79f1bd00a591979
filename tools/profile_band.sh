#!/bin/bash
# PMC passes of the band kernel alone (tools/time_band.py, one variant): HBM traffic and unit counters.  usage (through gpurun): tools/profile_band.sh <tag> [B]
set -u
tag=$1; B=${2:-8192}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export BAND_VARIANTS=${BAND_VARIANTS:-band16}
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/${tag}_pmc_$name -o p -- python3 $root/tools/time_band.py $B > $out/${tag}_pmc_$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/${tag}_pmc_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "band_newton" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print(f"{k:32s} per launch {v / n:.6g}  ({n} launches)")
PY
