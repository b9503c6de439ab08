#!/bin/bash
# Instruction-level profile of the frame kernel: rocprofv3 PC sampling over one
# launch of N x WxH frames (default: the bench workload, 1024 x 1920x1080 gray).
# usage (GPU box): tests/gpu_pc_sampling.sh [W H N [tag]]
# The library profiled is the line-table twin of the product (tests/build_variant.sh pcs "-gline-tables-only":
# same ISA apart from two exec saves per out-of-line prologue), so that samples carry file:line.
# Output: gpurun_out/pcs/<tag>_{hosttrap,stochastic}_top.txt (aggregated by profiles/summarize_pcs.py) and
# the first lines of the raw csv for the record.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-1920}; H=${2:-1080}; N=${3:-1024}; TAG=${4:-r05_1080p}
O=$R/gpurun_out/pcs
mkdir -p $O
cd /tmp
for m in ${PCS_METHODS:-stochastic host_trap}; do
  if [ $m = stochastic ]; then unit=cycles; iv=${PCS_IV_CYCLES:-4194304}; else unit=time; iv=${PCS_IV_US:-20000}; fi
  rm -rf /tmp/pcs_$m
  timeout ${PCS_TIMEOUT:-600} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $m \
      --pc-sampling-interval $iv --kernel-trace --output-format csv -d /tmp/pcs_$m -o pcs -- \
      env FIASCO_AMD_LIB=${PCS_LIB:-$R/fiasco_amd/libfiasco_amd_pcs.so} python3 $R/tests/gpu_perf_probe.py $W $H $N ${PCS_DISTINCT:-64} 1 > $O/${TAG}_$m.run.txt 2> $O/${TAG}_$m.err
  echo "$m: rc $?"; tail -3 $O/${TAG}_$m.err
  find /tmp/pcs_$m -type f | head; du -sh /tmp/pcs_$m
  f=$(find /tmp/pcs_$m -name '*pc_sampling*csv' | head -1)
  if [ -n "$f" ]; then
    head -5 "$f" > $O/${TAG}_$m.head.csv
    wc -l "$f"
    python3 $R/profiles/summarize_pcs.py "$f" > $O/${TAG}_${m}_top.txt 2> $O/${TAG}_${m}_top.err
    head -60 $O/${TAG}_${m}_top.txt
  fi
done
