#!/bin/bash
# developer helper (GPU box): sample shader clock / power while a command runs
# usage: tests/gpu_clock_watch.sh <tag> <command...>
tag=$1; shift
mkdir -p gpurun_out/clk
( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | python3 -c "
import json,sys,time
try:
    d=json.load(sys.stdin); c=d[sorted(d)[0]]
    print(time.time(), {k:v for k,v in c.items() if 'sclk' in k.lower() or 'ower' in k})
except Exception as e: print('err',e)
"; sleep 0.25; done ) > gpurun_out/clk/$tag.txt &
W=$!
"$@"
kill $W
python3 - <<PY
import re
v=[]; p=[]
for l in open('gpurun_out/clk/$tag.txt'):
    m=re.search(r"sclk[^:]*: '\(?(\d+)Mhz", l)
    if m: v.append(int(m.group(1)))
    m=re.search(r"ower[^:]*: '([\d.]+)'", l)
    if m: p.append(float(m.group(1)))
print('$tag sclk samples', len(v), 'min/avg/max', (min(v), sum(v)/len(v), max(v)) if v else None, 'power min/avg/max', (min(p), sum(p)/len(p), max(p)) if p else None)
PY
tail -3 gpurun_out/clk/$tag.txt
