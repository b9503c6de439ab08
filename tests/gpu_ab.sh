#!/bin/bash
# developer helper (GPU box): throughput probe of several builds of the library, one after the other
# usage: tests/gpu_ab.sh "<lib.so | ->:<env assignments>" ...   ("-" = fiasco_amd/libfiasco_amd.so)
#   AB_W / AB_H / AB_N / AB_ND / AB_REPS choose the workload (default 1920 1080 1024 64 2)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
W=${AB_W:-1920}; H=${AB_H:-1080}; N=${AB_N:-1024}; ND=${AB_ND:-64}; R=${AB_REPS:-2}
i=0
for spec in "$@"; do
  lib=${spec%%:*}; envs=""
  [ "$spec" != "$lib" ] && envs=${spec#*:}
  [ "$lib" = "-" ] && lib=fiasco_amd/libfiasco_amd.so
  echo "=== $spec"
  env $envs FIASCO_AMD_LIB=$lib timeout ${AB_TIMEOUT:-600} python3 tests/gpu_perf_probe.py $W $H $N $ND $R 2>&1 | tee gpurun_out/ab/run$i.txt | grep -v "^  sizes"
  i=$((i+1))
done
