"""Developer helper: where a predicted 720p colour frame of BASELINE config 5 spends its time on the device
(the big kernel build).  usage: gpu_config5_phases.py [frames=10]   (pattern ippppppppp, --prediction)"""
import os, sys, tempfile, time
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, fiasco_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
td = tempfile.mkdtemp()
paths = []
for f in range(n):
    p = os.path.join(td, "v%03d.ppm" % f)
    synth.write_ppm(p, synth.synth_color_k(1280, 720, 1234, 3 * f)); paths.append(p)
lib = fiasco_amd.Library(os.environ["FIASCO_AMD_LIB"]) if os.environ.get("FIASCO_AMD_LIB") else fiasco_amd.library()
lib.set_verbosity(0)
o = lib.cli_options(); o.set_prediction(1, 6, 10)
out = os.path.join(td, "dev.fco")
for first, label in ((1, "I frame"), (n, "%d frames" % n)):
    lib.reset_stats()
    t0 = time.time()
    assert lib.fiasco_coder(paths[:first], out, 20.0, o) == 1, lib.error_message()
    dt = time.time() - t0
    st = lib.get_stats()
    tt = max(st.t_total, 1)
    print("%s: wall %.2f s, kernel %.2f s in %d launches, frames %d" % (label, dt, st.kernel_ms / 1e3, st.launches, st.frames))
    print("  phase %%: init %.1f approx %.1f (A %.1f B %.1f) ipis %.1f append %.1f serial %.1f" % (
        100 * st.t_init / tt, 100 * st.t_approx / tt, 100 * st.t_mpA / tt, 100 * st.t_mpB / tt, 100 * st.t_ipis / tt,
        100 * st.t_append / tt, 100 * st.t_serial / tt))
    fr = max(st.frames, 1)
    print("  per frame: mp calls %d steps %d full evals %d blocks %d appends %d; states avg %d max %d; builds %s" % (
        st.n_mp // fr, st.n_steps // fr, st.n_fulleval // fr, st.n_blocks // fr, st.n_appends // fr,
        st.states_sum // fr, st.states_max, list(st.frames_by_build)))
    d = list(st.dbg)
    print("  big-build ops %%: chroma %.1f pred_setup %.1f pred_finish %.1f norms %.1f mc_search %.1f" % tuple(100 * d[k] / tt for k in (2, 3, 4, 5, 6)))
