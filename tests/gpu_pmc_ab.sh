#!/bin/bash
# developer helper: HBM traffic / L2 counters of the 1024 x 1080p launch for two library builds
# usage (on the GPU box): tests/gpu_pmc_ab.sh libA.so libB.so
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab; mkdir -p $O; cd /tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr"; do
    t=$(echo $c | cut -d' ' -f1)
    FIASCO_AMD_LIB=$R/$lib timeout 300 rocprofv3 --pmc $c -d $O/$n.$t -o pmc -- python3 $R/tests/gpu_perf_probe.py 1920 1080 1024 8 1 > $O/$n.$t.log 2>&1
    python3 $R/profiles/summarize_rocpd.py $O/$n.$t/*_results.db 2>&1 | grep fiasco_frame > $O/$n.$t.txt
    rm -rf $O/$n.$t
  done
done
grep -h "" $O/*.txt | cut -c1-150
