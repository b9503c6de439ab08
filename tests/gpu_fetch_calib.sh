#!/bin/bash
# developer helper (GPU box): calibrate FETCH_SIZE / WRITE_SIZE for 4 B/lane rows, 16 B/lane rows and 4-byte
# gathers (tests/calib/fetch_calib.hip) -> gpurun_out/calib/fetch_calib.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/calib; mkdir -p $O; cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib $R/tests/calib/fetch_calib.hip || exit 1
: > $O/fetch_calib.txt
for k in read4 read16 gather4 write4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p; timeout 200 rocprofv3 --pmc $c -d $O/p -o pmc -- /tmp/fetch_calib $k > $O/run.log 2>&1
    v=$(python3 $R/profiles/summarize_rocpd.py $O/p/*_results.db 2>/dev/null | grep -E "^(read4|read16|gather4|write4).*$c" | sed 's/.*sum *//; s/ .*//')
    echo "$k $c counter_KB $v  $(grep useful_bytes $O/run.log)" | tee -a $O/fetch_calib.txt
  done
done
rm -rf $O/p
python3 - <<PY >> $O/fetch_calib.txt
import re
rows=[l.split() for l in open('$O/fetch_calib.txt') if 'counter_KB' in l]
print('# kernel counter counter_bytes/useful_bytes counter_bytes/line_bytes')
for r in rows:
    try:
        cb=float(r[3])*1024; useful=float(r[6]); lines=float(r[8])
        print('# %-8s %-10s %.3f %.3f' % (r[0], r[1], cb/useful, cb/lines))
    except Exception as e: print('# parse', r, e)
PY
tail -10 $O/fetch_calib.txt
