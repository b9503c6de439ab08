"""Developer helper: throughput of the staged batch encoder for n frames of WxH.
usage: gpu_perf_probe.py W H n_frames n_distinct [repeats]"""
import hashlib
import os
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth
import fiasco_amd

w, h, n, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
lib = fiasco_amd.Library(os.environ["FIASCO_AMD_LIB"]) if os.environ.get("FIASCO_AMD_LIB") else fiasco_amd.library()
lib.set_verbosity(0)
if max(w, h) > 2048:
    lib.set_limits(30000, 26)
opt = lib.cli_options()
if os.environ.get("PROBE_PRED"):            # intra prediction: the big kernel builds (256 threads, two frames per CU, for full launches)
    opt.set_prediction(1, 6, 10)
if os.environ.get("PROBE_COLOR"):          # colour frames (Y, Cb, Cr bands): SURVEY Appendix C k-generator
    if max(w, h) > 1280:
        lib.set_limits(30000, 26)              # 1080p colour needs the limits extension
    uniq = [synth.ppm_bytes(synth.synth_color_k(w, h, 1234 if i == 0 else 1000 + i)) for i in range(nd)]
else:
    uniq = [synth.pgm_bytes(synth.synth(w, h, 1234 if i == 0 else 1000 + i)) for i in range(nd)]
frames = [uniq[i % nd] for i in range(n)]
t0 = time.time()
batch = fiasco_amd.Batch(lib, frames, 20.0, opt)
t_stage = time.time() - t0
for r in range(reps):
    lib.reset_stats()
    t0 = time.time()
    out = batch.encode()
    dt = time.time() - t0
    st = lib.get_stats()
    ok = sum(o is not None for o in out)
    tot = st.bytes_mp + st.bytes_img + st.bytes_gram
    ks = max(st.kernel_ms / 1e3, 1e-9)
    print("frames %d ok %d  stage %.2f s  encode wall %.3f s  kernel %.3f s  launches %d -> %.2f fps (wall) %.2f fps (kernel)"
          % (n, ok, t_stage, dt, ks, st.launches, n / dt, n / ks))
    print("  alg GB/frame mp %.3f img %.3f gram %.3f = %.3f;  achieved %.1f GB/s (%.2f%% of 8 TB/s)"
          % (st.bytes_mp / n / 1e9, st.bytes_img / n / 1e9, st.bytes_gram / n / 1e9, tot / n / 1e9,
             tot / ks / 1e9, tot / ks / 8e10))
    tt = max(st.t_total, 1)
    print("  phase %%: init %.1f approx %.1f ipis %.1f append %.1f serial %.1f | frame %.3f s avg | per frame: mp %d steps %d fulleval %d"
          % (100.0 * st.t_init / tt, 100.0 * st.t_approx / tt, 100.0 * st.t_ipis / tt,
             100.0 * st.t_append / tt, 100.0 * st.t_serial / tt, st.t_total / max(ok, 1) / 1e8,
             st.n_mp // max(ok, 1), st.n_steps // max(ok, 1), st.n_fulleval // max(ok, 1)))
    print("  mp: phaseA %.1f%% phaseB %.1f%% of frame; block evals/call %.2f, full evals/call %.1f"
          % (100.0 * st.t_mpA / tt, 100.0 * st.t_mpB / tt, st.n_blockevals / max(st.n_mp, 1), st.n_fulleval / max(st.n_mp, 1)))
    print("  per frame: blocks %d appends %d block-evals %d" % (st.n_blocks // max(ok, 1), st.n_appends // max(ok, 1), st.n_blockevals // max(ok, 1)))
    print("  states: avg %.0f max %d; re-encoded frames %d; frames per build (256/1024/big256/big512/1024tri): %s"
          % (st.states_sum / max(ok, 1), st.states_max, st.reencodes, list(st.frames_by_build)))
    if any(st.dbg):
        print("  dbg:", " ".join("%.3g" % (x / max(ok, 1)) for x in st.dbg))
    if not ok:
        print(lib.error_message())
    # replicas of one input must give one stream: anything else is a race on the device
    var = [len(set(out[i] for i in range(j, n, nd))) for j in range(nd)]
    if max(var) > 1:
        print("  NON-DETERMINISTIC: distinct streams per distinct input:", var)
print("  sizes per input:", [len(out[j]) if out[j] else None for j in range(nd)])
print("  md5 frame0:", hashlib.md5(out[0]).hexdigest() if out[0] else None, "sizes", sorted(set(len(o) for o in out if o))[:4])
batch.free()
