#!/bin/bash
# developer helper (GPU box): frames/s of small 1080p / 4K launches for every number of workgroups per frame the chip
# can hold -- the table spec_policy() (core_hip.cpp) is tuned by.  usage: tests/gpu_spec_policy_sweep.sh [W H] n ...
cd "$(dirname "$0")/.."
W=${SWEEP_W:-1920}; H=${SWEEP_H:-1080}
for n in "$@"; do
  echo "== $n frames of ${W}x${H}"
  python3 tests/gpu_spec_batch.py $W $H $n default 0 3 4 5 6 8 2>&1 | grep -E "mode=|MISMATCH|ERROR" | sed 's/ | spec frames.*tables/ tables/' | cut -c1-120
done
