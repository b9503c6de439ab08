#!/bin/bash
# developer helper: device vs oracle on a short colour/gray sequence, trace of the LAST frame compared
# usage: tests/gpu_video_quick.sh color|gray nframes [cfiasco args...]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/q
kind=${1:-color}; n=${2:-2}; shift 2
python3 - $kind $n <<'PY'
import sys; sys.path.insert(0,'tests')
from synth import *
kind, n = sys.argv[1], int(sys.argv[2])
for f in range(n):
    if kind == 'color': write_ppm('gpurun_out/q/v%d.pnm' % f, synth_color_c(256, 192, f))
    else: write_pgm('gpurun_out/q/v%d.pnm' % f, synth(128, 96, 31, 3 * f))
PY
files=$(ls gpurun_out/q/v?.pnm | head -$n)
FIASCO_ORACLE_TRACE=gpurun_out/q/v.or.trace oracle/cfiasco_oracle --progress-meter 0 "$@" -o gpurun_out/q/v.or.fco $files
FIASCO_AMD_DEBUG=1 FIASCO_AMD_TRACE=gpurun_out/q/v.gpu.trace timeout 300 fiasco_amd/bin/cfiasco --progress-meter 0 "$@" -o gpurun_out/q/v.gpu.fco $files
echo "oracle $(stat -c %s gpurun_out/q/v.or.fco) $(md5sum < gpurun_out/q/v.or.fco | cut -c1-12)  device $(stat -c %s gpurun_out/q/v.gpu.fco) $(md5sum < gpurun_out/q/v.gpu.fco | cut -c1-12)"
python3 tests/trace_diff.py gpurun_out/q/v.or.trace gpurun_out/q/v.gpu.trace
rm -f gpurun_out/q/v?.pnm
