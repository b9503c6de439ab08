"""Developer helper: append helpers of speculating frames -- per-frame comparison against one workgroup per frame."""
import os, sys, hashlib
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth, fiasco_amd
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = fiasco_amd.library(); lib.set_verbosity(0); opt = lib.cli_options()
if max(w, h) > 2048: lib.set_limits(30000, 26)
frames = [synth.pgm_bytes(synth.synth(w, h, 1234 if i == 0 else 1000 + i)) for i in range(n)]
res = {}
for name, env in (("one_wg", {"FIASCO_AMD_SPEC": "0"}), ("spec_noapp", {"FIASCO_AMD_SPEC_APP": "0"}), ("spec_app", {})) + tuple(("spec_app%s" % a, dict([("FIASCO_AMD_SPEC_APP", a.split(":")[0])] + ([("FIASCO_AMD_SPEC_APPDBG", a.split(":")[1])] if ":" in a else []))) for a in sys.argv[4:]):
    for k in ("FIASCO_AMD_SPEC", "FIASCO_AMD_SPEC_APP", "FIASCO_AMD_SPEC_APPDBG"): os.environ.pop(k, None)
    os.environ.update(env)
    lib.reset_stats()
    out = lib.encode_batch(frames, 20.0, opt)
    st = lib.get_stats()
    res[name] = out
    print(name, "kernel %.3f s" % (st.kernel_ms / 1e3), "launches", st.launches, "rows dealt", st.spec_app_rows, "wait %.3f s" % (st.spec_app_wait / 1e8),
          [None if o is None else hashlib.md5(o).hexdigest()[:8] for o in out], lib.error_message() if None in out else "", flush=True)
print({k: v == res["one_wg"] for k, v in res.items()})
