"""Developer helper: throughput of encode_batch for n frames of WxH (distinct seeds)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth, fiasco_amd
w, h, n, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib = fiasco_amd.library()
lib.set_verbosity(0)
opt = lib.cli_options()
uniq = [synth.pgm_bytes(synth.synth(w, h, 1000 + i)) for i in range(nd)]
frames = [uniq[i % nd] for i in range(n)]
lib.encode_batch(frames[:1], 20.0, opt)          # warm-up (HIP init, code load)
lib.reset_stats()
t0 = time.time()
out = lib.encode_batch(frames, 20.0, opt)
dt = time.time() - t0
st = lib.get_stats()
ok = sum(o is not None for o in out)
tot = st.bytes_mp + st.bytes_img + st.bytes_gram
print("frames %d ok %d  wall %.3f s  kernel %.3f s  launches %d  -> %.2f fps (wall) %.2f fps (kernel)" %
      (n, ok, dt, st.kernel_ms / 1e3, st.launches, n / dt, n / (st.kernel_ms / 1e3)))
print("alg bytes/frame: mp %.3f GB img %.3f GB gram %.3f GB total %.3f GB;  achieved %.1f GB/s" %
      (st.bytes_mp / n / 1e9, st.bytes_img / n / 1e9, st.bytes_gram / n / 1e9, tot / n / 1e9,
       tot / (st.kernel_ms / 1e3) / 1e9))
print("per frame: mp calls %d steps %d blocks %d appends %d fulleval %d; sizes %s" %
      (st.n_mp // n, st.n_steps // n, st.n_blocks // n, st.n_appends // n, st.n_fulleval // n,
       sorted(set(len(o) for o in out if o))[:6]))
tt = max(st.t_total, 1)
print("phase share of frame time: init %.1f%% approx %.1f%% ipis %.1f%% append %.1f%% serial %.1f%%; frame %.3f s avg" %
      (100.0 * st.t_init / tt, 100.0 * st.t_approx / tt, 100.0 * st.t_ipis / tt, 100.0 * st.t_append / tt,
       100.0 * st.t_serial / tt, st.t_total / n / 1e8))
if not ok:
    print(lib.error_message())
