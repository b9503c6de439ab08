"""Developer tool (GPU box): re-run ONE frame of a tests/fuzz_parity.py round on device and oracle
with per-call traces and show the first differing approximate_range record.
usage: fuzz_repro.py seed frame_index [frames_per_round=24]"""
import os
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiasco_amd
import fuzz_parity as fz

seed, idx = int(sys.argv[1]), int(sys.argv[2])
per = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rng = np.random.default_rng(seed)
spec = fz.random_options(rng)
q = float(rng.choice([2.0, 8.0, 20.0, 45.0, 90.0]))
frames = [fz.random_image(rng, bool(rng.integers(0, 3) == 0)) for _ in range(per)]
frame = frames[idx]
print("spec", spec, "q", q, "frame header", frame[:15].split(b"\n")[:2])
out = os.environ.get("FUZZ_OUT", "/tmp")
gpu = fiasco_amd.library()
ora = fiasco_amd.Library(os.path.join(os.path.dirname(fiasco_amd.LIB_PATH), "..", "oracle", "liboracle_fiasco.so"))
res = []
for lib, env, name in ((gpu, "FIASCO_AMD_TRACE", "gpu"), (ora, "FIASCO_ORACLE_TRACE", "or")):
    lib.set_verbosity(0)
    o = lib.cli_options()
    fz.apply(o, spec)
    os.environ[env] = os.path.join(out, "repro.%s.trace" % name)
    dump = os.path.join(out, "repro.%s.wfa" % name)
    if os.path.exists(dump):
        os.remove(dump)
    os.environ["FIASCO_DUMP_WFA"] = dump
    res.append(lib.encode_batch([frame], q, o)[0])
    del os.environ[env]
print("device", None if res[0] is None else len(res[0]), "oracle", None if res[1] is None else len(res[1]))
open(os.path.join(out, "repro.pnm"), "wb").write(frame)
subprocess.call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_diff.py"),
                 os.path.join(out, "repro.or.trace"), os.path.join(out, "repro.gpu.trace")])
