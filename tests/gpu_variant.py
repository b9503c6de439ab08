"""Developer helper: encode one golden input with an alternative build of the library.
usage: gpu_variant.py path/to/lib.so [input name]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiasco_amd
lib = fiasco_amd.Library(sys.argv[1])
lib.set_verbosity(0)
name = sys.argv[2] if len(sys.argv) > 2 else "g256"
data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".pgm"), "rb").read()
t0 = time.time()
out = lib.encode_batch([data], 20.0, lib.cli_options())[0]
print(sys.argv[1], name, "->", None if out is None else (len(out), hashlib.md5(out).hexdigest()), "%.2f s" % (time.time() - t0), lib.error_message() if out is None else "")
