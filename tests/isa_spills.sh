#!/bin/bash
# developer helper: compile the device coder with line tables and attribute scratch
# (spill / private array) instructions to source lines; extra compiler flags: ISA_FLAGS="-DFC_WG_PER_CU=4"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/fiasco_isa; rm -rf $T; mkdir -p $T; cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -gline-tables-only \
  -I$R/fiasco_amd/csrc/hip -I$R/fiasco_amd/csrc/host -I$R/include -save-temps \
  ${ISA_FLAGS:-} -c $R/fiasco_amd/csrc/hip/frame_coder.hip -o fc.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "VGPRs:|ScratchSize|VGPRs Spill|LDS Size|Occupancy" | sed 's/\[-Rpass[^]]*\]//g; s/^.*remark: [^ ]* *//'
python3 - <<'EOF'
import re, collections
S='/tmp/fiasco_isa/frame_coder-hip-amdgcn-amd-amdhsa-gfx950.s'
cur=None; loc=None; fil=None
files={}
cnt=collections.Counter(); tot=0
for line in open(S):
    m=re.match(r'^(_Z\w+):',line)
    if m: cur=m.group(1)
    m=re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',line)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2)).split('/')[-1]
    m=re.match(r'\s+\.loc\s+(\d+)\s+(\d+)',line)
    if m: fil=int(m.group(1)); loc=int(m.group(2))
    if line.strip().startswith('scratch_'):
        cnt[((cur or '?')[:24],files.get(fil,'?'),loc)]+=1; tot+=1
print('scratch instructions:', tot)
for k,v in sorted(cnt.items(), key=lambda kv:-kv[1])[:16]: print(' ',v,k)
EOF
