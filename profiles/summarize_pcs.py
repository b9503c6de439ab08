#!/usr/bin/env python3
"""Aggregate a rocprofv3 PC-sampling csv (host_trap or stochastic) into a hot-spot table.

usage: summarize_pcs.py pc_sampling.csv [top_n]

Prints: the csv's columns, the number of samples, the share of samples per source line (the
Instruction_Comment column carries file:line when the code object was built with -gline-tables-only:
tests/build_variant.sh pcs "-gline-tables-only"), the top instructions with their source line, and --
for the stochastic method -- the distribution of every low-cardinality column (issue / stall reasons)
overall and for each of the top source lines.
"""
import collections
import csv
import os
import re
import sys

path = sys.argv[1]
top_n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
csv.field_size_limit(1 << 30)
rd = csv.DictReader(open(path, newline=""))
cols = rd.fieldnames or []
print("# columns:", cols)
icol = next((c for c in cols if c.lower() == "instruction"), None)
ccol = next((c for c in cols if "comment" in c.lower()), None)
skip = {icol, ccol}
by_line = collections.Counter()
by_inst = collections.Counter()
cat = {c: collections.Counter() for c in cols if c not in skip}
cat_line = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))
n = 0
for row in rd:
    n += 1
    ins = (row.get(icol) or "").strip()
    com = (row.get(ccol) or "").strip()
    com = re.sub(r"^.*/(fiasco_amd/csrc/)?", "", com)
    by_line[com] += 1
    by_inst[(com, ins)] += 1
    for c in cat:
        v = row.get(c)
        if len(cat[c]) <= 64:
            cat[c][v] += 1
            if len(cat_line[com][c]) <= 64:
                cat_line[com][c][v] += 1
print("# samples: %d" % n)
low = [c for c in cat if 1 < len(cat[c]) <= 40]
print("\n== low-cardinality columns (all samples) ==")
for c in low:
    print("%-36s %s" % (c, ", ".join("%s: %.1f%%" % (k, 100.0 * v / max(n, 1)) for k, v in cat[c].most_common(12))))
print("\n== top source lines ==")
acc = 0
for com, v in by_line.most_common(top_n):
    acc += v
    extra = ""
    for c in low:
        d = cat_line[com][c]
        if d:
            k, kv = d.most_common(1)[0]
            extra += "  %s=%s %.0f%%" % (c[:18], k, 100.0 * kv / v)
    print("%6.2f%%  (cum %5.1f%%)  %-34s%s" % (100.0 * v / max(n, 1), 100.0 * acc / max(n, 1), com or "?", extra))
print("\n== top instructions ==")
for (com, ins), v in by_inst.most_common(top_n * 2):
    print("%6.2f%%  %-34s %s" % (100.0 * v / max(n, 1), com or "?", ins[:90]))
# per source file + hundred-line bucket: a coarse "which function" view without symbol information
print("\n== by file and 100-line bucket ==")
b = collections.Counter()
for com, v in by_line.items():
    m = re.match(r"(.*):(\d+)", com)
    b[(m.group(1), int(m.group(2)) // 100 * 100) if m else (com, 0)] += v
for (f, l), v in b.most_common(top_n):
    print("%6.2f%%  %s:%d.." % (100.0 * v / max(n, 1), f, l))
