#!/bin/bash
# developer helper (GPU box): which unit of the CU is busy during the 1024 x 1080p launch?
# usage: tests/gpu_pmc_units.sh lib.so [lib.so ...]   -> gpurun_out/units/<lib>.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/units; mkdir -p $O; cd /tmp
for lib in "$@"; do
  n=$(basename $lib .so); : > $O/$n.txt
  i=0
  for c in "GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    FIASCO_AMD_LIB=$R/$lib timeout 200 rocprofv3 --pmc $c -d $O/$n.p$i -o pmc -- python3 $R/tests/gpu_perf_probe.py 1920 1080 1024 16 1 > $O/$n.p$i.log 2>&1
    python3 $R/profiles/summarize_rocpd.py $O/$n.p$i/*_results.db 2>&1 | grep -E "^fiasco_frame|dur_ns" | sed 's/fiasco_frame_kernel([^)]*)//; s/dispatches *1 *sum *//; s/ avg.*//' | cut -c1-160 >> $O/$n.txt
    tail -2 $O/$n.p$i.log | grep -i error >> $O/$n.txt
    rm -rf $O/$n.p$i
  done
  echo "=== $n"; cat $O/$n.txt
done
