#!/bin/bash
# Round-end measurement recipe (run on the GPU box through gpurun):
#   1. bench.py (the judged JSON line)                      -> gpurun_out/prof/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command  -> gpurun_out/prof/kernel_trace_stats.txt
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC / SQ, separate passes, no tracing flags
#      (MI355X_MICROARCH.md, HBM section)                    -> gpurun_out/prof/pmc_*.txt
#   4. the 4K workload (256 frames of 3840x2160, limits extension): bench line only -- rocprofv3
#      --kernel-trace around the 4K run did not return on this pool (round 2: killed after 39 min),
#      so bench.py's own HIP-event launch time is the 4K kernel figure
# Summaries are produced with profiles/summarize_rocpd.py and copied into profiles/ by hand
# (profiles/r02_*).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp
python3 $R/bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-loop --no-small-launches --k4-frames 0 --config5-frames 0 --config3-frames 0 --no-k4-small > $O/bench_traced.json 2> $O/trace.err
python3 $R/profiles/summarize_rocpd.py $O/trace/*_results.db > $O/kernel_trace_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-loop --no-small-launches --k4-frames 0 --config5-frames 0 --config3-frames 0 --no-k4-small > $O/bench_pmc_$c.json 2> $O/pmc_$c.err
  python3 $R/profiles/summarize_rocpd.py $O/pmc_$c/*_results.db > $O/pmc_$c.txt 2>&1
done
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc_tcc -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-loop --no-small-launches --k4-frames 0 --config5-frames 0 --config3-frames 0 --no-k4-small > $O/bench_pmc_tcc.json 2> $O/pmc_tcc.err
python3 $R/profiles/summarize_rocpd.py $O/pmc_tcc/*_results.db > $O/pmc_tcc.txt 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-loop --no-small-launches --k4-frames 0 --config5-frames 0 --config3-frames 0 --no-k4-small > $O/bench_pmc_sq.json 2> $O/pmc_sq.err
python3 $R/profiles/summarize_rocpd.py $O/pmc_sq/*_results.db > $O/pmc_sq.txt 2>&1
# 4K: 256 frames in one launch through the frame queue (HBM holds about 130 of the 2.2 GB slabs of the tight capacity guess)
timeout 900 python3 $R/bench.py --width 3840 --height 2160 --frames-per-gpu 256 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-loop --no-small-launches --k4-frames 0 --config5-frames 0 --config3-frames 0 --no-k4-small > $O/bench_4k.json 2> $O/bench_4k.err
# small launches: several workgroups per frame (block-level speculation): kernel trace of 16 x 1080p and 1 x 4K
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_spec -o kt -- python3 $R/tests/gpu_spec_batch.py 1920 1080 16 default > $O/spec_16x1080p.txt 2> $O/trace_spec.err
python3 $R/profiles/summarize_rocpd.py $O/trace_spec/*_results.db > $O/spec_kernel_trace_stats.txt 2>&1
timeout 300 python3 $R/tests/gpu_spec_batch.py 3840 2160 8 0 default > $O/spec_8x4k.txt 2>&1
rm -rf $O/trace_spec
# round 5: PMC traffic of the 4K launch (fiasco_frame_kernel_wide_tri), separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/pmc4k_$c -o pmc -- python3 $R/tests/gpu_perf_probe.py 3840 2160 256 32 1 > $O/pmc4k_$c.log 2>&1
  python3 $R/profiles/summarize_rocpd.py $O/pmc4k_$c/*_results.db > $O/pmc4k_$c.txt 2>&1
  rm -rf $O/pmc4k_$c
done
# ... and kernel trace + PMC traffic of BASELINE config 5 (fiasco_frame_kernel_big_wide: 11 launches)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_c5 -o kt -- python3 $R/tests/gpu_config5.py 300 > $O/config5.txt 2> $O/trace_c5.err
python3 $R/profiles/summarize_rocpd.py $O/trace_c5/*_results.db > $O/config5_kernel_trace.txt 2>&1
rm -rf $O/trace_c5
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/pmcc5_$c -o pmc -- python3 $R/tests/gpu_config5.py 300 > $O/pmcc5_$c.log 2>&1
  python3 $R/profiles/summarize_rocpd.py $O/pmcc5_$c/*_results.db > $O/pmcc5_$c.txt 2>&1
  rm -rf $O/pmcc5_$c
done
# unit utilisation of the final kernel
bash $R/tests/gpu_pmc_units.sh fiasco_amd/libfiasco_amd.so > $O/pmc_units.log 2>&1
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_tcc $O/pmc_sq   # keep the summaries only
grep -h fiasco $O/kernel_trace_stats.txt $O/pmc_*.txt | head -40
