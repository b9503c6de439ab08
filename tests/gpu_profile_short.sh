#!/bin/bash
# The short half of tests/gpu_profile.sh: the judged bench line, its kernel trace, and the small launches
# (several workgroups per frame).  No PMC passes, no 4K batch.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp
python3 $R/bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-loop --no-small-launches > $O/bench_traced.json 2> $O/trace.err
python3 $R/profiles/summarize_rocpd.py $O/trace/*_results.db > $O/kernel_trace_stats.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_spec -o kt -- python3 $R/tests/gpu_spec_batch.py 1920 1080 16 default > $O/spec_16x1080p.txt 2> $O/trace_spec.err
python3 $R/profiles/summarize_rocpd.py $O/trace_spec/*_results.db > $O/spec_kernel_trace_stats.txt 2>&1
timeout 300 python3 $R/tests/gpu_spec_batch.py 3840 2160 8 0 default > $O/spec_8x4k.txt 2>&1
rm -rf $O/trace_spec $O/trace
grep -h fiasco $O/kernel_trace_stats.txt | head -10
