#!/usr/bin/env python3
"""Developer helper: static instructions of a frame-kernel build attributed to SOURCE functions (through the line
tables: an inlined callee's instructions count for the callee), per assembly function.

usage: tests/isa_lines.py [asm function substring] [extra hipcc flags ...]

Compiles fiasco_amd/csrc/hip/frame_coder.hip with -gline-tables-only -save-temps into /tmp/fiasco_isa_lines, then walks
the gfx950 assembly: every instruction belongs to the source line of the last `.loc`, a source line to the function
whose definition starts last before it (definitions found by a regular expression: good enough for a budget).
Output: per source function VALU / SALU / LDS / VMEM / waitcnt+barrier counts inside the chosen assembly function
(default: the kernel itself, where the matching pursuit, its set-up and the partition search's callers are inlined).
Used for profiles/r06_chain_budget.txt.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/fiasco_isa_lines"
HIP = os.path.join(ROOT, "fiasco_amd", "csrc", "hip")
KFLAGS = ["-mllvm", "-disable-machine-licm", "-DFC_SERIAL_LOOP=1"]


def build(extra):
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fno-fast-math", "-I" + ROOT + "/include", "-I" + ROOT + "/fiasco_amd/csrc/host", "-I" + HIP,
           "-gline-tables-only", "-save-temps", "-c", os.path.join(HIP, "frame_coder.hip"), "-o", "fc.o"] + KFLAGS + extra
    r = subprocess.run(cmd, cwd=OUT, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        sys.exit(1)
    return os.path.join(OUT, "frame_coder-hip-amdgcn-amd-amdhsa-gfx950.s")


def source_functions(path):
    """[(first line, name)] of the function definitions of a source file"""
    out = []
    rx = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:static\s+)?(?:__device__|__global__)[^;{]*?\b(\w+)\s*\(")
    lines = open(path, errors="ignore").read().split("\n")
    pend = ""
    for i, ln in enumerate(lines, 1):
        t = pend + ln
        m = rx.match(t.strip())
        if m and not t.strip().endswith(";"):
            out.append((i, m.group(1)))
        pend = (ln + " ") if ln.strip().startswith("template") and "(" not in ln else ""
    return out


def main():
    want = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "fiasco_frame_kernelP"
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    asm = build(extra)
    files, fn_of = {}, {}
    cur_fn, loc = None, (0, 0)
    table = collections.defaultdict(lambda: collections.Counter())
    for ln in open(asm, errors="ignore"):
        s = ln.strip()
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", s)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3))
            continue
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", s)
        if m and not s.startswith("."):
            cur_fn = m.group(1)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        if not cur_fn or want not in cur_fn or not s or s[0] in ".;" or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^(v_|s_|ds_|global_|flat_|scratch_|buffer_)", op):
            continue
        path = files.get(loc[0], "?")
        if path not in fn_of:
            fn_of[path] = source_functions(path) if os.path.exists(path) else []
        name = "?"
        for first, nm in fn_of[path]:
            if first <= loc[1]:
                name = nm
            else:
                break
        key = os.path.basename(path) + ":" + name
        c = table[key]
        c["total"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
        elif op in ("s_waitcnt", "s_barrier", "s_nop"):
            c["wait"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["vmem"] += 1
            if op.startswith("scratch_"):
                c["scratch"] += 1
    print("# static instructions inside assembly function *%s*, by SOURCE function (line tables)" % want)
    print("%-44s %7s %7s %7s %6s %6s %8s %6s" % ("source function", "total", "valu", "salu", "lds", "vmem", "scratch", "wait"))
    tot = collections.Counter()
    for key, c in sorted(table.items(), key=lambda kv: -kv[1]["total"]):
        print("%-44s %7d %7d %7d %6d %6d %8d %6d" % (key[:44], c["total"], c["valu"], c["salu"], c["lds"], c["vmem"], c["scratch"], c["wait"]))
        tot.update(c)
    print("%-44s %7d %7d %7d %6d %6d %8d %6d" % ("ALL", tot["total"], tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["scratch"], tot["wait"]))


if __name__ == "__main__":
    main()
