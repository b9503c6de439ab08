"""Developer helper: first differing approximate_range record between two trace files
(FIASCO_ORACLE_TRACE / FIASCO_AMD_TRACE)."""
import sys
import numpy as np
dt = np.dtype([('seq', 'i4'), ('level', 'i4'), ('image', 'i4'), ('D', 'i4'), ('states', 'i4'),
               ('nedges', 'i4'), ('cost', 'f4'), ('err', 'f4'), ('mbits', 'f4'), ('wbits', 'f4'),
               ('into', 'i2', 6), ('w', 'f4', 5)])
a = np.fromfile(sys.argv[1], dt)
b = np.fromfile(sys.argv[2], dt)
print(len(a), len(b), 'records')
n = min(len(a), len(b))
for i in range(n):
    if a[i].tobytes() != b[i].tobytes():
        print('first diff at record', i)
        for k in range(max(0, i - 2), min(n, i + 3)):
            print(' A', a[k]); print(' B', b[k])
        break
else:
    print('no diff in common prefix')
