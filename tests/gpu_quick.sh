#!/bin/bash
# developer helper: device-vs-oracle stream + per-call trace comparison
# usage: tests/gpu_quick.sh [name ...]   (names: g96 g256 n512 g720 g1080 c00 c128 c512 k720 k1080)
# extra cfiasco arguments for both coders: QUICK_ARGS="-z 1"
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/q
names="${*:-n512 g720}"
python3 - $names <<'PY'
import sys; sys.path.insert(0,'tests')
from synth import *
gen = {'g96': lambda: synth(96,64,5), 'g256': lambda: synth(256,256,1234), 'n512': noise,
       'g720': lambda: synth(1280,720,1234), 'g1080': lambda: synth(1920,1080,1234),
       'c00': lambda: synth_color_c(320,256,0), 'c128': lambda: synth_color_c(128,128,0),
       'c512': lambda: synth_color_c(512,384,1),
       'k720': lambda: synth_color_k(1280,720), 'k1080': lambda: synth_color_k(1920,1080)}
for n in sys.argv[1:]:
    a = gen[n]()
    (write_ppm if a.ndim == 3 else write_pgm)('gpurun_out/q/%s.pnm' % n, a)
PY
for f in $names; do
  s0=$(date +%s%N)
  FIASCO_ORACLE_TRACE=gpurun_out/q/$f.or.trace oracle/cfiasco_oracle --progress-meter 0 ${QUICK_ARGS:-} -o gpurun_out/q/$f.or.fco gpurun_out/q/$f.pnm
  s1=$(date +%s%N)
  FIASCO_AMD_DEBUG=1 FIASCO_AMD_TRACE=gpurun_out/q/$f.gpu.trace timeout 120 fiasco_amd/bin/cfiasco --progress-meter 0 ${QUICK_ARGS:-} -o gpurun_out/q/$f.gpu.fco gpurun_out/q/$f.pnm
  s2=$(date +%s%N)
  echo "$f: oracle $(( (s1-s0)/1000000 )) ms, device $(( (s2-s1)/1000000 )) ms (process start + hip init included)"
  echo "$f: oracle $(stat -c %s gpurun_out/q/$f.or.fco) $(md5sum < gpurun_out/q/$f.or.fco | cut -c1-12)  device $(stat -c %s gpurun_out/q/$f.gpu.fco) $(md5sum < gpurun_out/q/$f.gpu.fco | cut -c1-12)"
  python3 tests/trace_diff.py gpurun_out/q/$f.or.trace gpurun_out/q/$f.gpu.trace
done
[ -n "${KEEP_TRACE:-}" ] || rm -f gpurun_out/q/*.trace; rm -f gpurun_out/q/*.pnm
