#!/bin/bash
# developer helper (GPU box, through gpurun): HBM traffic of the default kernel BY OPERATION.
# The plain library and two builds that run one idempotent table operation twice (frame_coder.hip FC_DUP_OP;
# build them first:  tests/build_variant.sh dupinit "-DFC_DUP_OP=OP_INIT_RANGE";  tests/build_variant.sh dupappend
# "-DFC_DUP_OP=OP_APPEND") run the 1024 x 1080p launch under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes
# (MI355X_MICROARCH.md, HBM section); the difference to the plain build is the operation's traffic.
# Output: gpurun_out/traffic_by_op/*.txt (summaries), raw counter values for profiles/r06_traffic_by_op.txt.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/traffic_by_op
mkdir -p $O
cd /tmp
for lib in libfiasco_amd libfiasco_amd_dupinit libfiasco_amd_dupappend; do
  [ -f $R/fiasco_amd/$lib.so ] || { echo "missing $lib.so"; continue; }
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/raw
    FIASCO_AMD_LIB=$R/fiasco_amd/$lib.so timeout 300 rocprofv3 --pmc $c -d $O/raw -o pmc -- python3 $R/tests/gpu_perf_probe.py ${TBO_W:-1920} ${TBO_H:-1080} ${TBO_N:-1024} 64 1 > $O/${lib}_$c.log 2>&1
    python3 $R/profiles/summarize_rocpd.py $O/raw/*_results.db > $O/${lib}_$c.txt 2>&1
    echo "== $lib $c"; grep -h "fiasco" $O/${lib}_$c.txt | head -4; grep -h "fps (kernel)\|md5 frame0" $O/${lib}_$c.log | head -2
  done
done
rm -rf $O/raw
