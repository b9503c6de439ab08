#!/usr/bin/env python3
"""Developer helper: static instruction mix of a frame-kernel build, per function.

usage: tests/isa_mix.py [extra hipcc flags ...]      (default: the 256-thread default build's KFLAGS)

Compiles fiasco_amd/csrc/hip/frame_coder.hip with -save-temps into /tmp/fiasco_isa_mix and counts, per
function of the gfx950 assembly: VALU / SALU / LDS (ds_) / VMEM (global_, flat_, scratch_, buffer_)
instructions, and the classes the round-3 verdict asked to budget: scratch_, flat_ vs global_, v_div_*,
f64 arithmetic, 64-bit address arithmetic (v_lshlrev_b64, v_add_co / v_addc_co pairs, v_mad_u64_u32).
The committed copy of its output is profiles/r04_isa_mix.txt.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/fiasco_isa_mix"
KFLAGS = ["-mllvm", "-disable-machine-licm", "-DFC_SERIAL_LOOP=1"]


def build(extra):
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fno-fast-math", "-I" + ROOT + "/include", "-I" + ROOT + "/fiasco_amd/csrc/host",
           "-I" + ROOT + "/fiasco_amd/csrc/hip", "-save-temps", "-c",
           ROOT + "/fiasco_amd/csrc/hip/frame_coder.hip", "-o", "fc.o",
           "-Rpass-analysis=kernel-resource-usage"] + KFLAGS + extra
    r = subprocess.run(cmd, cwd=OUT, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        sys.exit(1)
    res = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^ ]+ +(Function Name|VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|"
                      r"LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if m:
            res.setdefault(m.group(1), m.group(2))
    return res


def classify(op):
    c = []
    if op.startswith("v_"):
        c.append("valu")
    elif op.startswith("s_"):
        c.append("salu")
    elif op.startswith("ds_"):
        c.append("lds")
    elif op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        c.append("vmem")
    if op.startswith("scratch_"):
        c.append("scratch")
    if op.startswith("flat_"):
        c.append("flat")
    if op.startswith("global_"):
        c.append("global")
    if op.startswith("v_div_") or op.startswith("v_rcp") :
        c.append("div")
    if re.search(r"_f64|f64_", op):
        c.append("f64")
    if op in ("v_lshlrev_b64", "v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32", "v_addc_co_u32", "v_add_co_u32",
              "v_ashrrev_i64", "v_lshl_add_u64"):
        c.append("addr64")
    if "dpp" in op:
        c.append("dpp")
    if op in ("s_barrier",):
        c.append("barrier")
    if op.startswith("s_waitcnt"):
        c.append("waitcnt")
    if op in ("s_swappc_b64", "s_setpc_b64"):
        c.append("call")
    return c


def main():
    extra = sys.argv[1:]
    res = build(extra)
    S = os.path.join(OUT, "frame_coder-hip-amdgcn-amd-amdhsa-gfx950.s")
    cur = None
    per = collections.OrderedDict()
    for line in open(S):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith((".L", "\t")):
            name = m.group(1)
            if not name.startswith(".L"):
                cur = name
                per.setdefault(cur, collections.Counter())
            continue
        if cur is None or not line.startswith("\t"):
            continue
        t = line.strip()
        if not t or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        if not re.match(r"^(v_|s_|ds_|global_|flat_|scratch_|buffer_)", op):
            continue
        per[cur]["total"] += 1
        for c in classify(op):
            per[cur][c] += 1
    try:
        names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + list(per.keys()), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        names = list(per.keys())
    cols = ["total", "valu", "salu", "lds", "vmem", "scratch", "flat", "global", "div", "f64", "addr64", "dpp", "barrier", "waitcnt"]
    print("# static instruction mix, flags: %s" % " ".join(KFLAGS + extra))
    print("# resources: " + ", ".join("%s %s" % kv for kv in res.items()))
    print("%-44s" % "function" + "".join("%8s" % c for c in cols))
    tot = collections.Counter()
    for (k, v), nm in zip(per.items(), names):
        if not v["total"]:
            continue
        nm = re.sub(r"\(.*", "", nm)[:43]
        print("%-44s" % nm + "".join("%8d" % v[c] for c in cols))
        tot.update(v)
    print("%-44s" % "ALL" + "".join("%8d" % tot[c] for c in cols))


if __name__ == "__main__":
    main()
