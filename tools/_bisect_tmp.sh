export CANNOLES_HIP_LIB=$GRAFT_REPO_ROOT/build_abl/ab_exp/libcannoles_hip.so CANNOLES_HIP_ALLOW_EXPERIMENT=1 CNL_DBG_SCRATCHFILL=1 CNL_DBG_LDSFILL=0xffffffff
for c in 9236 9472; do FUZZ_OPTS="staged_large_fronts=1" timeout 120 python tools/fuzz_parity.py 1 $c > /tmp/o.txt 2>&1; echo "case $c rc $?: $(tail -n 1 /tmp/o.txt | cut -c1-150)"; done
for s0 in 9000 9300 9600 12000 15000; do FUZZ_OPTS="staged_large_fronts=1" timeout 500 python tools/fuzz_parity.py 300 $s0 > gpurun_out/fl_$s0.txt 2>&1; echo "== $s0 rc $?: $(tail -n 1 gpurun_out/fl_$s0.txt | cut -c1-200)"; done
