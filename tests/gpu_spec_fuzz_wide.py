"""Developer helper: the 1024-thread speculating build against one 1024-thread workgroup per frame on random
large frames (more than 3072 states; limits extension) -- device against device, the one-workgroup path is
pinned against the reference elsewhere.  usage: gpu_spec_fuzz_wide.py [cases] [seed]"""
import os, sys, time
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth, fiasco_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = fiasco_amd.library(); lib.set_verbosity(0)
lib.set_limits(30000, 26)
bad = 0
for c in range(cases):
    rng = np.random.default_rng(seed + c)
    w = int(rng.integers(1100, 1700)) * 2; h = int(rng.integers(700, 1100)) * 2
    colour = bool(rng.integers(0, 3) == 0)
    n = int(rng.integers(1, 4))
    q = float(rng.choice([8.0, 20.0, 45.0]))
    frames = []
    for i in range(n):
        if colour:
            frames.append(synth.ppm_bytes(synth.synth_color_k(w, h, int(rng.integers(1, 10000)))))
        else:
            frames.append(synth.pgm_bytes(synth.synth(w, h, int(rng.integers(1, 10000)))))
    o = lib.cli_options()
    os.environ["FIASCO_AMD_SPEC"] = "0"
    t0 = time.time(); ref = lib.encode_batch(frames, q, o); t_ref = time.time() - t0
    G = int(rng.integers(3, 9))
    os.environ["FIASCO_AMD_SPEC"] = str(G)
    os.environ["FIASCO_AMD_SPEC_T"] = str(int(rng.integers(0, max(1, G - 2))))
    lib.reset_stats()
    t0 = time.time(); got = lib.encode_batch(frames, q, o); t_got = time.time() - t0
    st = lib.get_stats()
    o.delete()
    ok = got == ref and None not in ref
    bad += not ok
    print("%dx%d %s n=%d q=%g G=%d T=%s: one workgroup %.2f s, several %.2f s, spec frames %d wrong %d -> %s"
          % (w, h, "colour" if colour else "gray", n, q, G, os.environ["FIASCO_AMD_SPEC_T"], t_ref, t_got,
             st.spec_frames, st.spec_wrong, "identical" if ok else "MISMATCH " + str(lib.error_message())), flush=True)
print("wide fuzz: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
