"""Developer helper (GPU box): n distinct frames through the 256-thread build (left-spine batching, block
pricing) and through one wide workgroup per frame / several workgroups per frame (neither): every stream
must be the same.  usage: gpu_cross_build.py W H n [reps]"""
import hashlib
import os
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth
import fiasco_amd

w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lib = fiasco_amd.Library(os.environ["FIASCO_AMD_LIB"]) if os.environ.get("FIASCO_AMD_LIB") else fiasco_amd.library()
lib.set_verbosity(0)
if max(w, h) > 2048:
    lib.set_limits(30000, 26)
opt = lib.cli_options()
frames = [synth.pgm_bytes(synth.synth(w, h, 5000 + i)) for i in range(n)]


def run(env, tag):
    for k, v in env.items():
        os.environ[k] = v
    try:
        b = fiasco_amd.Batch(lib, frames, 20.0, opt)
        lib.reset_stats()
        outs = [b.encode() for _ in range(reps)]
        st = lib.get_stats()
        b.free()
    finally:
        for k in env:
            del os.environ[k]
    print(tag, "frames by build", list(st.frames_by_build), "spec frames", st.spec_frames, "steps", st.n_steps, "fulleval", st.n_fulleval)
    return outs


a = run({"FIASCO_AMD_NO_WIDE": "1", "FIASCO_AMD_SPEC": "0"}, "narrow")
b = run({"FIASCO_AMD_SPEC": "0"}, "wide  ")
bad = 0
for r in range(reps):
    for i in range(n):
        if a[r][i] != b[0][i]:
            bad += 1
            print("MISMATCH rep %d frame %d: narrow %s wide %s" % (r, i, a[r][i] and hashlib.md5(a[r][i]).hexdigest()[:12], b[0][i] and hashlib.md5(b[0][i]).hexdigest()[:12]))
print("cross-build check: %d frames x %d reps, %d mismatches" % (n, reps, bad))
sys.exit(1 if bad else 0)
