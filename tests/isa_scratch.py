#!/usr/bin/env python3
"""Developer helper: scratch (spill / private array) instructions of a frame-kernel build by function and
source line, with the build's real flags (tests/isa_mix.py KFLAGS).  usage: tests/isa_scratch.py [flags]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/fiasco_isa_scr"
os.makedirs(OUT, exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-I" + ROOT + "/include", "-I" + ROOT + "/fiasco_amd/csrc/host", "-I" + ROOT + "/fiasco_amd/csrc/hip",
       "-mllvm", "-disable-machine-licm", "-DFC_SERIAL_LOOP=1", "-gline-tables-only", "-save-temps", "-c",
       ROOT + "/fiasco_amd/csrc/hip/frame_coder.hip", "-o", "fc.o"] + sys.argv[1:]
r = subprocess.run(cmd, cwd=OUT, capture_output=True, text=True)
if r.returncode:
    sys.stderr.write(r.stderr[-3000:]); sys.exit(1)
S = OUT + "/frame_coder-hip-amdgcn-amd-amdhsa-gfx950.s"
cur = None; loc = None; fil = None; files = {}
cnt = collections.Counter()
for line in open(S):
    m = re.match(r'^(_Z\w+):', line)
    if m: cur = m.group(1)
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', line)
    if m: fil = int(m.group(1)); loc = int(m.group(2))
    if line.strip().startswith('scratch_'):
        cnt[((cur or '?')[:28], files.get(fil, '?'), loc, line.strip().split()[0].replace('scratch_', ''))] += 1
only = os.environ.get("ISA_FUNC", "")
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0], kv[0][1], kv[0][2])):
    if only in k[0]: print(' ', v, k)
