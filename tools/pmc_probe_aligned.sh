set -u
root=$PWD
cd /tmp && export TMPDIR=/tmp
export CANNOLES_HIP_ALLOW_EXPERIMENT=1 BAND_VARIANTS=band32il
for v in BAND_DBG-32 BAND_DBG-32_BAND_PROBE_ALIGNED-1; do
  export CANNOLES_HIP_LIB=$root/build_abl/libcnl_abl$v.so
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-20)
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/gpurun_out/probe_${v}_$n -o p -- python3 $root/tools/time_band.py 16384 > $root/gpurun_out/probe_${v}_$n.log 2>&1
  done
done
ls $root/gpurun_out | grep probe_ | head
