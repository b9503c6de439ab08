# developer helper: issue counters of the frame kernel for one build (run through gpurun)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
L=${1:-libfiasco_amd}
rm -rf $R/gpurun_out/pmc_ic
FIASCO_AMD_LIB=$R/fiasco_amd/$L.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_ic -o ic -- python3 $R/tests/gpu_perf_probe.py 1920 1080 1024 8 1 > $R/gpurun_out/pmc_ic.log 2>&1
python3 $R/profiles/summarize_rocpd.py $R/gpurun_out/pmc_ic/*_results.db 2>&1 | grep -E "fiasco.*(SQ_|dur)" | sed 's/fiasco_frame_kernel(DevFrame\*) *//; s/dispatches *1 *sum *//; s/ avg.*//'
rm -rf $R/gpurun_out/pmc_ic
