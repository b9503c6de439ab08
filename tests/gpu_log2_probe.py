"""developer probe: device log2 vs host log2 over every float in (0, 1] (run through gpurun)"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiasco_amd
lib = fiasco_amd.library()
f = lib.L.fiasco_amd_selftest_log2
c = ctypes
f.argtypes = [c.c_uint, c.c_uint, c.POINTER(c.c_ulonglong), c.POINTER(c.c_ulonglong), c.POINTER(c.c_ulonglong), c.POINTER(c.c_float)]
f.restype = c.c_int
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 127)
n, dd, df, bad = c.c_ulonglong(), c.c_ulonglong(), c.c_ulonglong(), c.c_float()
t = time.time()
ok = f(lo, hi, n, dd, df, bad)
print("ok", ok, "checked", n.value, "double-mismatch", dd.value, "float-mismatch", df.value, "first bad", bad.value, "%.1f s" % (time.time() - t), lib.error_message())
