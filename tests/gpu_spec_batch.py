"""Developer helper: frames/s of small batches with and without block-level speculation.
usage: gpu_spec_batch.py W H n [mode ...]    mode: 0 | default | G | G:T"""
import os, sys, time, hashlib
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth, fiasco_amd
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
modes = sys.argv[4:] or ["0", "default"]
lib = fiasco_amd.Library(os.environ["FIASCO_AMD_LIB"]) if os.environ.get("FIASCO_AMD_LIB") else fiasco_amd.library()
lib.set_verbosity(0); opt = lib.cli_options()
if max(w, h) > 2048:
    lib.set_limits(30000, 26)
if os.environ.get("PROBE_COLOR"):
    if max(w, h) > 1280: lib.set_limits(30000, 26)
    frames = [synth.ppm_bytes(synth.synth_color_k(w, h, 1234 if i == 0 else 1000 + i)) for i in range(n)]
else:
    frames = [synth.pgm_bytes(synth.synth(w, h, 1234 if i == 0 else 1000 + i)) for i in range(n)]
ref = None
for mode in modes:
    os.environ.pop("FIASCO_AMD_SPEC", None); os.environ.pop("FIASCO_AMD_SPEC_T", None)
    if mode != "default":
        g = mode.split(":")
        os.environ["FIASCO_AMD_SPEC"] = g[0]
        if len(g) > 1: os.environ["FIASCO_AMD_SPEC_T"] = g[1]
    b = fiasco_amd.Batch(lib, frames, 20.0, opt)
    for rep in range(2):
        lib.reset_stats(); out = b.encode(); st = lib.get_stats()
    print("%dx%d n=%d mode=%s: kernel %.3f s %.1f frames/s | spec frames %d tasks %d wrong %d (taken over %d) inline %d wait %.2f s tables %d/%d | append rows dealt %d, wait %.3f s (FIASCO_AMD_SPEC_APP=%s)"
          % (w, h, n, mode, st.kernel_ms / 1e3, n / (st.kernel_ms / 1e3), st.spec_frames, st.spec_tasks, st.spec_wrong, st.spec_adopted,
             st.spec_inline, st.spec_wait / 1e8, st.spec_tab_used, st.spec_tab_missed, st.spec_app_rows, st.spec_app_wait / 1e8,
             os.environ.get("FIASCO_AMD_SPEC_APP", "auto")), flush=True)
    if any(o is None for o in out): print("   ERROR", lib.error_message())
    if ref is None: ref = out
    else: print("   identical" if out == ref else "   MISMATCH")
    b.free()
