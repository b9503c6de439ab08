"""BASELINE config 5 on the device: N-frame 1280x720 colour sequence (SURVEY App. C k-generator,
x shifted by 3 px per frame), pattern ippppppppp, --prediction, block levels 6..10.
usage: gpu_config5.py [frames] [check_frames]   (check_frames: prefix also run through the oracle)
Prints one JSON line: frames/s of fiasco_coder() on the device (file I/O included)."""
import hashlib, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import synth

def make(args):
    f, path = args
    synth.write_ppm(path, synth.synth_color_k(1280, 720, 1234, 3 * f))
    return path

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ncheck = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    td = tempfile.mkdtemp()
    with mp.get_context("fork").Pool(min(16, len(os.sched_getaffinity(0)))) as pool:
        paths = pool.map(make, [(f, os.path.join(td, "v%03d.ppm" % f)) for f in range(n)])
    import fiasco_amd
    lib = fiasco_amd.Library(os.environ["FIASCO_AMD_LIB"]) if os.environ.get("FIASCO_AMD_LIB") else fiasco_amd.library()
    lib.set_verbosity(0)
    o = lib.cli_options(); o.set_prediction(1, 6, 10)
    out = os.path.join(td, "dev.fco")
    res = {}
    if n >= 3 and not os.environ.get("CONFIG5_NO_CHECK"):                                   # SURVEY App. C known answer of the reference
        assert lib.fiasco_coder(paths[:3], out, 20.0, o) == 1, lib.error_message()
        res["v0[0-2] md5 == reference"] = hashlib.md5(open(out, "rb").read()).hexdigest() == "2528889c0453c4590289ad07d9fb87e3"
    lib.reset_stats()
    t0 = time.time()
    assert lib.fiasco_coder(paths, out, 20.0, o) == 1, lib.error_message()
    dt = time.time() - t0
    st = lib.get_stats()
    # a step of the sweep lasts as long as its slowest frame: mean frame time x steps against the kernel time
    res["kernel_seconds"] = st.kernel_ms / 1e3
    res["launches"] = int(st.launches)
    res["mean_frame_seconds"] = st.t_total / 1e8 / max(st.frames, 1)
    res["frames_searched"] = int(st.frames)
    data = open(out, "rb").read()
    res.update({"workload": "%d frames 1280x720 colour, ippppppppp, --prediction" % n, "seconds": dt,
                "frames_per_s": n / dt, "bytes": len(data), "md5": hashlib.md5(data).hexdigest()})
    if n == 300:                                 # the real reference's stream for these 300 inputs (68 min on one core)
        res["300-frame md5 == reference"] = res["md5"] == "714a25c639d8daae598f84095f4856f7"
    if ncheck:
        ora = fiasco_amd.Library(os.path.join(ROOT, "oracle", "liboracle_fiasco.so")); ora.set_verbosity(0)
        oo = ora.cli_options(); oo.set_prediction(1, 6, 10)
        a, b = os.path.join(td, "a.fco"), os.path.join(td, "b.fco")
        t0 = time.time()
        assert ora.fiasco_coder(paths[:ncheck], a, 20.0, oo) == 1, ora.error_message()
        res["oracle_seconds_%d_frames" % ncheck] = time.time() - t0
        assert lib.fiasco_coder(paths[:ncheck], b, 20.0, o) == 1, lib.error_message()
        res["device == oracle on the first %d frames" % ncheck] = open(a, "rb").read() == open(b, "rb").read()
    print(json.dumps(res))

if __name__ == "__main__":
    main()
