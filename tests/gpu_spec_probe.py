"""Developer helper: block-level speculation (several workgroups per frame) against the plain
one-workgroup-per-frame launch -- same bytes?  how much faster?
usage: gpu_spec_probe.py [W H n_frames [G ...]]      (G = 0: speculation off)"""
import hashlib
import os
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth
import fiasco_amd

w, h, n = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 1)
gs = [int(a) for a in sys.argv[4:]] or [0, 8]
lib = fiasco_amd.library()
lib.set_verbosity(0)
if max(w, h) > 2048:
    lib.set_limits(30000, 26)
opt = lib.cli_options()
frames = [synth.pgm_bytes(synth.synth(w, h, 1234 if i == 0 else 1000 + i)) for i in range(n)]
ref = None
for g in gs:
    os.environ["FIASCO_AMD_SPEC"] = str(g)
    batch = fiasco_amd.Batch(lib, frames, 20.0, opt)
    for rep in range(2):
        lib.reset_stats()
        t0 = time.time()
        out = batch.encode()
        dt = time.time() - t0
        st = lib.get_stats()
        md5 = [hashlib.md5(o).hexdigest()[:12] if o else None for o in out]
        print("G=%d rep %d: wall %.3f s kernel %.3f s -> %.2f frames/s | spec frames %d tasks %d confirmed %d wrong %d "
              "timeout %d inline %d wait %.3f s | builds %s | md5 %s"
              % (g, rep, dt, st.kernel_ms / 1e3, n / max(st.kernel_ms / 1e3, 1e-9), st.spec_frames, st.spec_tasks,
                 st.spec_confirmed, st.spec_wrong, st.spec_timeout, st.spec_inline, st.spec_wait / 1e8,
                 list(st.frames_by_build), md5[:3]), flush=True)
        tt = max(st.t_total, 1)
        print("      chain: frame %.3f s avg; init %.1f%% approx %.1f%% ipis %.1f%% append %.1f%% serial %.1f%% (of the chain's time); mp calls %d"
              % (st.t_total / n / 1e8, 100.0 * st.t_init / tt, 100.0 * st.t_approx / tt, 100.0 * st.t_ipis / tt,
                 100.0 * st.t_append / tt, 100.0 * st.t_serial / tt, st.n_mp // n), flush=True)
        print("      checkpoints %.1f%% returns %.1f%% of the chain's time; tables from workers %d, missed %d"
              % (100.0 * st.dbg[0] / tt, 100.0 * st.dbg[1] / tt, st.spec_tab_used, st.spec_tab_missed))
        if any(o is None for o in out):
            print("   ERROR:", lib.error_message())
    if ref is None:
        ref = out
    elif out != ref:
        bad = [i for i in range(n) if out[i] != ref[i]]
        print("   MISMATCH against G=%d in frames %s" % (gs[0], bad[:16]))
    else:
        print("   identical to G=%d" % gs[0])
    batch.free()
