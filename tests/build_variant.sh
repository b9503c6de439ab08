#!/bin/bash
# developer helper: build fiasco_amd/libfiasco_amd_<name>.so with extra flags for the frame kernels
# usage: tests/build_variant.sh name "-DFC_SERIAL_PROFILE=1 ..."   (test with FIASCO_AMD_LIB=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; extra=${2:-}
cd $R/fiasco_amd/csrc
make -s >/dev/null
B=build/var_$name; mkdir -p $B
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm -I../../include -Ihost -Ihip -I../../experiments/r04_left_spine"
/opt/rocm/bin/hipcc $HF ${NARROW_FLAGS--DFC_SERIAL_LOOP=1} $extra -c hip/frame_coder.hip -o $B/frame_coder.o &
/opt/rocm/bin/hipcc $HF ${NARROW_FLAGS--DFC_SERIAL_LOOP=1} ${WIDE_FLAGS--DFC_WIDE_B=1024} $extra -DFC_VARIANT_WIDE=1 -c hip/frame_coder.hip -o $B/frame_coder_wide.o &
/opt/rocm/bin/hipcc $HF ${NARROW_FLAGS--DFC_SERIAL_LOOP=1} ${WIDE_FLAGS--DFC_WIDE_B=1024} $extra -DFC_VARIANT_WIDE=1 -DFC_GRAM_TRI=1 -c hip/frame_coder.hip -o $B/frame_coder_wide_tri.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfiasco_amd_$name.so build/fa_*.o $B/frame_coder.o $B/frame_coder_wide.o $B/frame_coder_wide_tri.o \
   build/frame_coder_big.o build/frame_coder_big_wide.o build/frame_coder_big_hm.o build/frame_coder_big_gm.o build/frame_coder_spec.o build/frame_coder_spec_wide.o build/core_hip.o -lm -lpthread
echo built fiasco_amd/libfiasco_amd_$name.so
