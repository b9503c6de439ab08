#!/usr/bin/env python3
"""bench.py -- grayscale frames/s encoded by the HIP hot path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic frames per GPU: F
independent 1920x1080 grayscale frames (BASELINE config 2: CLI defaults, -q 20, 8x8 px
minimum range blocks, default dictionary; every frame has its own seed), all in flight at
once -- one persistent workgroup per frame, four frames per CU -- through the staged C-ABI
entries of include/libfiasco_amd.h.  Two timed loops over the same batch:

  value        inputs resident in HBM when the timed region starts (fiasco_amd_batch_stage
               outside it); timed: device partition search + matching pursuit, download of
               the automata, host-side .fco entropy writer, K passes pipelined
               (fiasco_amd_batch_submit / _collect).
  config.pcie_inclusive_frames_per_s
               the same K passes, but every pass encodes frames that arrive as raw PNM
               buffers in host memory INSIDE the timed region (fiasco_amd_batch_upload: parse,
               pinned staging, host->HBM copy), the transfer of pass i+1 overlapping the
               kernel of pass i.

Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL).  `--gpus N` without
a torchrun environment starts the N ranks itself.  Frames are independent units (SURVEY.md
§8e), so every rank encodes its own F frames (weak scaling) with no data-path collective; the
finished byte strings are gathered over RCCL after the timed region (fiasco_amd/sharding.py).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# reference streams of the survey image (seed 1234): stock reference at 1080p, the survey's
# patched reference (limits extension) at 4K
REF_MD5_SEED1234 = {(1920, 1080): "c7a9f8a7029644d5fed0dc0583e8b478",
                    (3840, 2160): "b121615161f96c0b541f29ef7228380e",
                    (1280, 720): "ffd04173f6956c80c711933c98c30d5b",
                    (256, 256): "5cdbb073486c509e54153d790b1674f5"}


def cpu_baseline(frame_pnm, w, h):
    """Time the CPU coder on the host cores of this box on a bounded sample (1 frame of the
    same workload, 1 thread: the reference algorithm is single threaded).  Uses the real
    reference binary when the prebuilt oracle/_ref travels with the repo, else the port."""
    tmp = "/tmp/fiasco_bench_cpu"
    os.makedirs(tmp, exist_ok=True)
    src = os.path.join(tmp, "frame.pgm")
    open(src, "wb").write(frame_pnm)
    out = os.path.join(tmp, "frame.fco")
    ref = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref")
    port = os.path.join(ROOT, "oracle", "cfiasco_oracle")
    env = dict(os.environ, FIASCO_DATA=os.path.join(ROOT, "fiasco_amd", "data"))
    if max(w, h) > 2048:           # the stock reference crashes above level 22 (SURVEY finding 2)
        return {"value": None, "unit": "frames/s", "cores": 1, "kind": "reference",
                "sample": "not run: the stock reference cannot encode %dx%d" % (w, h)}
    for kind, exe in (("reference", ref), ("port", port)):
        if not os.path.exists(exe):
            continue
        try:
            # one warm-up run, then the median of five
            times = []
            r = None
            for i in range(6):
                t0 = time.time()
                r = subprocess.run([exe, "--progress-meter", "0", "-o", out, src], env=env,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
                if r.returncode != 0:
                    break
                if i:
                    times.append(time.time() - t0)
            dt = sorted(times)[len(times) // 2] if times else None
        except Exception:
            continue
        if r is not None and r.returncode == 0 and dt and os.path.exists(out):
            md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
            res = {"value": 1.0 / dt, "unit": "frames/s", "cores": 1, "kind": kind,
                   "sample": "1 frame %dx%d gray, -q 20 -z 0: median of %d runs after a warm-up, %.2f s (min %.2f, max %.2f), "
                             "stream md5 %s" % (w, h, len(times), dt, min(times), max(times), md5[:12])}
            # SURVEY 8d also asks for "all cores, one frame per core": C concurrent processes of
            # the same coder, one frame each (C capped at 32 to bound host memory and time)
            try:
                ncores = min(len(os.sched_getaffinity(0)), 32)
                t0 = time.time()
                procs = [subprocess.Popen([exe, "--progress-meter", "0", "-o", "%s.%d" % (out, i), src], env=env,
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(ncores)]
                ok = all(p.wait(timeout=600) == 0 for p in procs)
                dta = time.time() - t0
                if ok:
                    res["all_cores"] = {"value": ncores / dta, "unit": "frames/s", "cores": ncores,
                                        "sample": "%d concurrent processes, one frame each, %.1f s" % (ncores, dta)}
                for i in range(ncores):
                    if os.path.exists("%s.%d" % (out, i)):
                        os.remove("%s.%d" % (out, i))
            except Exception:
                pass
            return res
    return {"value": None, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "cpu coder unavailable"}


def pmc_traffic(frames, w, h, colour=False):
    """HBM bytes per launch from the PMC passes of tests/gpu_profile.sh (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs of the same workload), committed under profiles/.
    Counters cannot be collected from inside the timed run: the figure is STATIC -- measured on
    another run of the same kernel build, reported only for the workload it was measured on -- and
    the JSON line says so (`traffic_source`).  Returns (bytes or None, source or None)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        for e in (j if isinstance(j, list) else [j]):
            if (frames, w, h, bool(colour)) == (e.get("frames_per_launch", 768), e.get("width", 1920), e.get("height", 1080), bool(e.get("colour", False))):
                return (e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"],
                        "profiles/%s (static: rocprofv3 --pmc FETCH_SIZE x %g + WRITE_SIZE, separate passes over this workload "
                        "and kernel build, not collected in this run)" % (name, e.get("fetch_correction", 1.0)))
    return None, None


def small_launches(lib, opt, uniq, w, h):
    """BASELINE config 2 taken literally -- ONE frame -- and a batch of 16: launches that leave most
    of the chip empty.  The launcher then gives every frame several workgroups (block-level
    speculation): chain, table workers, verifiers.  Timed next to the same launch with
    one (wide) workgroup per frame (FIASCO_AMD_SPEC=0); the streams must be the same bytes.  Outside
    the timed region of the headline figure."""
    import fiasco_amd
    import synth
    res = {}
    os.environ["FIASCO_AMD_DEBUG"] = "1"      # FIASCO_AMD_SPEC below is a developer switch of the library
    # BASELINE config 3 (colour 1080p: needs the declared limits extension) and one 4K frame
    colour = [synth.ppm_bytes(synth.synth_color_k(w, h))]
    gray4k = [synth.pgm_bytes(synth.synth(3840, 2160, 1234))]
    for name, frames in (("single_frame", uniq[:1]), ("batch_of_16", uniq[:16]), ("single_colour_frame", colour),
                         ("single_4k_frame", gray4k)):
        if len(frames) < (16 if name == "batch_of_16" else 1):
            continue
        big = name in ("single_colour_frame", "single_4k_frame")
        if big:
            lib.set_limits(30000, 26)
        o = lib.cli_options()
        ent, ref = {}, None
        for key, env in (("several_workgroups_per_frame", None), ("one_workgroup_per_frame", "0")):
            if env is None:
                os.environ.pop("FIASCO_AMD_SPEC", None)
            else:
                os.environ["FIASCO_AMD_SPEC"] = env
            try:
                b = fiasco_amd.Batch(lib, frames, 20.0, o)
                b.encode()
                best = None
                for _ in range(3):
                    lib.reset_stats()
                    t0 = time.perf_counter()
                    out = b.encode()
                    dt = time.perf_counter() - t0
                    best = dt if best is None or dt < best else best
                st = lib.get_stats()
                b.free()
            finally:
                os.environ.pop("FIASCO_AMD_SPEC", None)
            if ref is None:
                ref = out
            ent[key] = {"seconds": best, "frames_per_s": len(frames) / best,
                        "workgroups_per_frame_launches": int(st.spec_frames), "identical_streams": out == ref,
                        "blocks_confirmed": int(st.spec_confirmed), "blocks_sent_back": int(st.spec_wrong),
                        "blocks_taken_over_from_the_verifier": int(st.spec_adopted),
                        "blocks_searched_by_the_chain": int(st.spec_inline)}
        o.delete()
        if big:
            lib.set_limits(6000, 22)
        res[name] = ent
    return res


def k4_pass(lib, nframes, rank):
    """BASELINE config 4's workload on ONE GPU, driver-timed: `nframes` independent 3840x2160 grayscale
    frames in one launch (declared limits extension, SURVEY 8c: the stock reference cannot encode 4K),
    inputs resident in HBM, one pass (kernel + automaton download + .fco writer).  Frame 0 is the survey
    image, whose stream must be the patched reference's (md5 b1216151..., SURVEY App. C and
    tests/golden/MANIFEST_BIG.json)."""
    import fiasco_amd
    w, h = 3840, 2160
    seeds = [1234 if (rank == 0 and i == 0) else 200000 + i for i in range(nframes)]
    tg = time.perf_counter()
    frames = make_frames(w, h, seeds)
    t_gen = time.perf_counter() - tg
    lib.L.fiasco_amd_release_memory()          # the slabs of the 1080p batch
    lib.set_limits(30000, 26)
    o = lib.cli_options()
    try:
        ts = time.perf_counter()
        b = fiasco_amd.Batch(lib, frames, 20.0, o)
        t_stage = time.perf_counter() - ts
        import torch
        torch.cuda.synchronize()
        lib.reset_stats()
        t0 = time.perf_counter()
        out = b.encode()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = lib.get_stats()
        b.free()
    finally:
        o.delete()
        lib.set_limits(6000, 22)
        lib.L.fiasco_amd_release_memory()
    ok = out is not None and all(x is not None for x in out)
    alg = float(st.bytes_mp + st.bytes_img + st.bytes_gram)
    ks = st.kernel_ms / 1e3
    md5 = hashlib.md5(out[0]).hexdigest() if ok else None
    return {"frames": nframes, "all_encoded": ok, "frames_per_s": nframes / dt if ok else None, "seconds": dt,
            "kernel_only_frames_per_s": st.frames / ks if ks else None, "launches": int(st.launches),
            "reencoded_frames": int(st.reencodes), "frames_by_kernel_build": list(st.frames_by_build),
            "roofline": {"bound": "hbm", "achieved": alg / ks / 1e9 if ks else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / ks / 1e9 / HBM_PEAK_GBS if ks else None, "traffic": pmc_traffic(nframes, w, h)[0],
                         "traffic_source": pmc_traffic(nframes, w, h)[1],
                         "kernel": "fiasco_frame_kernel_wide_tri", "avg_launch_ms": st.kernel_ms / max(st.launches, 1),
                         "algorithmic_bytes_per_launch": alg / max(st.launches, 1)},
            "parity": ("stream md5 of survey frame == patched reference (%s)" % md5[:12])
                      if md5 == REF_MD5_SEED1234[(w, h)] else "MISMATCH: %s" % md5,
            "stage_seconds": t_stage, "generate_seconds": t_gen,
            "limits": "MAXSTATES 30000, MAXLEVEL 26 (declared extension, SURVEY 8c)"}


def launch_form(lib, frames, w, h, cus=256):
    """The launcher's choice for a launch of `frames` gray frames of w x h on one GPU (pure policy functions of the
    library, include/libfiasco_amd_hip.h): kernel build, workgroups per frame, append helpers per frame."""
    big = 1 if max(w, h) > 2048 else 0
    g = int(lib.L.fiasco_amd_spec_workgroups(frames, cus, big, 0 if big else 1, 1 if big else 4))
    if g:
        build = "fiasco_frame_kernel_spec_wide (1024 threads)" if big else "fiasco_frame_kernel_spec (256 threads)"
        return {"frames": frames, "kernel_build": build, "workgroups_per_frame": g,
                "append_helpers_per_frame": int(lib.L.fiasco_amd_spec_append_helpers(frames, cus, g, big))}
    if big:
        build = "fiasco_frame_kernel_wide_tri (1024 threads, triangular Gram tables, frame queue)"
    else:
        build = "fiasco_frame_kernel (256 threads, 4 frames per CU)" if frames > cus else "fiasco_frame_kernel_wide (1024 threads)"
    return {"frames": frames, "kernel_build": build, "workgroups_per_frame": 1, "append_helpers_per_frame": 0}


def k4_small_pass(lib, counts=(64, 8)):
    """BASELINE config 4 as written -- 64 independent 3840x2160 frames -- on ONE GPU, and 8 frames: one GPU's share of
    the job on an 8-GPU node.  Launches this small leave the chip empty: the launcher gives every frame several
    workgroups (block-level speculation, the 1024-thread build).  Reported: rate, kernel build, workgroups per frame."""
    import fiasco_amd
    w, h = 3840, 2160
    res = {}
    frames = make_frames(w, h, [1234] + [300000 + i for i in range(max(counts) - 1)])
    lib.L.fiasco_amd_release_memory()
    lib.set_limits(30000, 26)
    o = lib.cli_options()
    try:
        for n in counts:
            b = fiasco_amd.Batch(lib, frames[:n], 20.0, o)
            import torch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.encode()                       # first pass of a batch: the verifiers' buffers and control blocks are set up
            torch.cuda.synchronize()
            dt_first = time.perf_counter() - t0
            lib.reset_stats()
            t0 = time.perf_counter()
            out = b.encode()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = lib.get_stats()
            b.free()
            ok = out is not None and all(x is not None for x in out)
            md5 = hashlib.md5(out[0]).hexdigest() if ok else None
            cus = 256
            res["frames_%d" % n] = {
                "frames": n, "all_encoded": ok, "seconds": dt, "frames_per_s": n / dt if ok else None,
                "first_pass_seconds": dt_first,
                "kernel_seconds": st.kernel_ms / 1e3, "launches": int(st.launches),
                "frames_with_several_workgroups": int(st.spec_frames),
                "workgroups_per_frame": int(lib.L.fiasco_amd_spec_workgroups(n, cus, 1, 0, 1)) or 1,
                # round 6: further workgroups of a frame that build their shares of the Gram row of every appended state
                "append_helpers_per_frame": int(lib.L.fiasco_amd_spec_append_helpers(n, cus, int(lib.L.fiasco_amd_spec_workgroups(n, cus, 1, 0, 1)), 1)),
                "append_rows_dealt": int(st.spec_app_rows),
                "kernel_build": "fiasco_frame_kernel_spec_wide" if st.spec_frames else "fiasco_frame_kernel_wide / _wide_tri",
                "frames_by_kernel_build": list(st.frames_by_build),
                "parity": ("frame 0 == patched reference (%s)" % md5[:12]) if md5 == REF_MD5_SEED1234[(w, h)] else "MISMATCH: %s" % md5}
    finally:
        o.delete()
        lib.set_limits(6000, 22)
        lib.L.fiasco_amd_release_memory()
    return res


REF_MD5_K1080 = "ca81b603e331985870426d8257eaacbd"          # patched reference, k1080.ppm at -z 0 (tests/golden/MANIFEST_BIG.json)


def _colour_frame(args):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    w, h, seed = args
    return synth.ppm_bytes(synth.synth_color_k(w, h, seed))


def config3_pass(lib, nframes):
    """BASELINE config 3 as a throughput figure: `nframes` independent 1920x1080 colour frames (SURVEY App. C
    k-generator, one seed each; Y, Cb, Cr bands, chroma dictionary 40) at the CLI defaults, which need the declared
    limits extension (> 6000 states per frame, SURVEY finding 7).  Frame 0 is k1080.ppm (patched reference: ca81b603...)."""
    import multiprocessing as mp
    import fiasco_amd
    w, h = 1920, 1080
    tg = time.perf_counter()
    with mp.get_context("fork").Pool(min(16, len(os.sched_getaffinity(0)))) as pool:
        frames = pool.map(_colour_frame, [(w, h, 1234 if i == 0 else 400000 + i) for i in range(nframes)], chunksize=2)
    t_gen = time.perf_counter() - tg
    lib.L.fiasco_amd_release_memory()
    lib.set_limits(30000, 26)
    o = lib.cli_options()
    try:
        b = fiasco_amd.Batch(lib, frames, 20.0, o)
        import torch
        torch.cuda.synchronize()
        lib.reset_stats()
        t0 = time.perf_counter()
        out = b.encode()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = lib.get_stats()
        b.free()
    finally:
        o.delete()
        lib.set_limits(6000, 22)
        lib.L.fiasco_amd_release_memory()
    ok = out is not None and all(x is not None for x in out)
    md5 = hashlib.md5(out[0]).hexdigest() if ok else None
    alg = float(st.bytes_mp + st.bytes_img + st.bytes_gram)
    ks = st.kernel_ms / 1e3
    return {"workload": "%d independent 1920x1080 colour frames, cfiasco defaults, limits extension" % nframes,
            "all_encoded": ok, "seconds": dt, "frames_per_s": nframes / dt if ok else None,
            "kernel_seconds": ks, "launches": int(st.launches), "reencoded_frames": int(st.reencodes),
            "frames_by_kernel_build": list(st.frames_by_build), "states_max": int(st.states_max),
            "roofline": {"bound": "hbm", "achieved": alg / ks / 1e9 if ks else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / ks / 1e9 / HBM_PEAK_GBS if ks else None,
                         "traffic": pmc_traffic(nframes, 1920, 1080, True)[0], "traffic_source": pmc_traffic(nframes, 1920, 1080, True)[1]},
            "parity": ("frame 0 == patched reference (%s)" % md5[:12]) if md5 and md5.startswith(REF_MD5_K1080) else "MISMATCH: %s" % md5,
            "generate_seconds": t_gen}


REF_MD5_CONFIG5_300 = "714a25c639d8daae598f84095f4856f7"     # the real reference's stream of these 300 inputs (68 min on one core)


def config5_pass(lib, nframes):
    """BASELINE config 5 on ONE GPU, driver-timed: `nframes` 1280x720 colour frames (SURVEY App. C k-generator, shifted
    by 3 px per frame), pattern ippppppppp, --prediction, through fiasco_coder() -- PPM files read, 30 GOPs searched
    side by side per frame position (big kernel build, 8 workgroups per frame for the table passes), every coded frame
    decoded on the device for the next one, .fco written.  300 frames: the stream must be the real reference's."""
    import multiprocessing as mp
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_config5
    td = tempfile.mkdtemp()
    tg = time.perf_counter()
    with mp.get_context("fork").Pool(min(16, len(os.sched_getaffinity(0)))) as pool:
        paths = pool.map(gpu_config5.make, [(f, os.path.join(td, "v%03d.ppm" % f)) for f in range(nframes)])
    t_gen = time.perf_counter() - tg
    lib.L.fiasco_amd_release_memory()
    o = lib.cli_options()
    o.set_prediction(1, 6, 10)
    out = os.path.join(td, "dev.fco")
    try:
        lib.reset_stats()
        t0 = time.perf_counter()
        ok = lib.fiasco_coder(paths, out, 20.0, o) == 1
        dt = time.perf_counter() - t0
        st = lib.get_stats()
        data = open(out, "rb").read() if ok else b""
    finally:
        o.delete()
        lib.L.fiasco_amd_release_memory()
        import shutil
        shutil.rmtree(td, ignore_errors=True)
    md5 = hashlib.md5(data).hexdigest() if ok else None
    alg = float(st.bytes_mp + st.bytes_img + st.bytes_gram)
    ks5 = st.kernel_ms / 1e3
    return {"workload": "%d frames 1280x720 colour, ippppppppp, --prediction, fiasco_coder() file to file" % nframes,
            # the same byte formulas summed by the device coder over all launches of the sequence / the
            # HIP-event time of those launches; kernel fiasco_frame_kernel_big_wide (30 frames x 8 workgroups per launch)
            "roofline": {"bound": "hbm", "achieved": alg / ks5 / 1e9 if ks5 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / ks5 / 1e9 / HBM_PEAK_GBS if ks5 else None,
                         # the 14 launches of the sequence together (whole job), as `algorithmic_bytes` is
                         "traffic": pmc_traffic(nframes, 1280, 720, True)[0], "traffic_source": pmc_traffic(nframes, 1280, 720, True)[1],
                         "kernel": "fiasco_frame_kernel_big_wide", "algorithmic_bytes": alg},
            "all_encoded": ok, "seconds": dt, "frames_per_s": nframes / dt if ok else None, "bytes": len(data),
            "kernel_seconds": st.kernel_ms / 1e3, "launches": int(st.launches), "reencoded_frames": int(st.reencodes),
            "workgroups_per_frame_for_tables": int(st.coop_workgroups), "frames_decoded_on_device": int(st.decoder_frames),
            "parity": (("stream md5 == the real reference's (%s)" % md5[:12]) if md5 == REF_MD5_CONFIG5_300 else "MISMATCH: %s" % md5)
                      if nframes == 300 else "md5 %s (known answer exists for 300 frames only)" % md5,
            "generate_seconds": t_gen}


# ---- synthetic frames: one seed per frame (SURVEY Appendix C generator, tests/synth.py) ----

_BASE = {}


def _frame(args):
    """synth.synth(w, h, seed) with the seed-independent part of the image computed once per
    process.  Same float operations in the same order: bit-identical to tests/synth.py (the
    md5 of the seed-1234 frame's stream is checked against the reference's)."""
    import numpy as np
    w, h, seed = args
    if (w, h) not in _BASE:
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        base = 128 + 60 * np.sin(x / 17.0) * np.cos(y / 23.0) + 40 * (((x // 32) + (y // 32)) % 2)
        blob = 50 * np.exp(-((x - w * 0.3) ** 2 + (y - h * 0.6) ** 2) / (2 * (w / 10) ** 2))
        _BASE[(w, h)] = (base, blob)
    base, blob = _BASE[(w, h)]
    img = base + np.random.default_rng(seed).normal(0, 6, (h, w))
    img += blob
    return b'P5\n%d %d\n255\n' % (w, h) + np.clip(img, 0, 255).astype(np.uint8).tobytes()


def make_frames(w, h, seeds):
    import multiprocessing as mp
    nproc = max(1, min(len(os.sched_getaffinity(0)), 32, len(seeds) // 4))
    work = [(w, h, s) for s in seeds]
    if nproc == 1:
        return [_frame(a) for a in work]
    with mp.get_context("fork").Pool(nproc) as pool:
        return pool.map(_frame, work, chunksize=4)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-gpu", type=int, default=1024)   # 4 frames per CU x 256 CUs
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic frames per rank (0 = every frame its own seed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie-loop", action="store_true", help="skip the PCIe-inclusive timed loop")
    ap.add_argument("--no-small-launches", action="store_true", help="skip the single-frame / 16-frame timings")
    ap.add_argument("--no-decoder", action="store_true", help="skip the decoder pass over the frames (SURVEY 8f row F4)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank encodes --frames-per-gpu frames (default); strong: --frames-per-gpu is the "
                         "TOTAL of the job, dealt round robin to the ranks (BASELINE config 4: --scaling strong "
                         "--width 3840 --height 2160 --frames-per-gpu 64 on 1/2/4/8 GPUs)")
    ap.add_argument("--k4-frames", type=int, default=256,
                    help="frames of the extra 3840x2160 pass (BASELINE config 4 on one GPU; 0 = skip)")
    ap.add_argument("--config3-frames", type=int, default=1024,
                    help="frames of the extra 1920x1080 COLOUR pass (BASELINE config 3 as a batch; 0 = skip)")
    ap.add_argument("--no-k4-small", action="store_true", help="skip the 64- and 8-frame 4K launches (BASELINE config 4 as written)")
    ap.add_argument("--config5-frames", type=int, default=300,
                    help="frames of the extra 1280x720 colour video pass (BASELINE config 5 on one GPU; 0 = skip)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become `python -m torch.distributed.run` with one rank per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # developer checks of the multi-rank path: FIASCO_BENCH_SAME_GPU=1 puts all ranks on GPU 0 of a
    # 1-GPU box (gloo for the tiny collectives); FIASCO_BENCH_DRYRUN=1 runs launcher, rendezvous,
    # timing reductions and the gather of the streams WITHOUT the device coder (no GPU needed; the
    # line carries "dry_run": true and no rate) -- tests/test_sharding.py runs it with 2 ranks
    same_gpu = os.environ.get("FIASCO_BENCH_SAME_GPU") == "1"
    dry = os.environ.get("FIASCO_BENCH_DRYRUN") == "1"
    if same_gpu:
        local = 0

    F = a.frames_per_gpu
    total_job = world * F
    if a.scaling == "strong":
        # the job is F frames in all; rank r takes the frames r, r + W, ... (sharding.shard_indices)
        total_job = F
        F = len(range(rank, total_job, world))
        assert F > 0, "--scaling strong: fewer frames than ranks"
    ndist = a.distinct if a.distinct > 0 else F
    # frame 0 of rank 0 is the survey image (seed 1234) whose reference stream md5 is known
    seeds = [1234 if (rank == 0 and i == 0) else 100000 + rank * a.frames_per_gpu + i for i in range(ndist)]
    tg = time.perf_counter()
    uniq = make_frames(a.width, a.height, seeds) if not dry else [b"P5\n32 32\n255\n" + bytes(1024)] * ndist
    t_gen = time.perf_counter() - tg
    frames = [uniq[i % len(uniq)] for i in range(F)]

    import torch
    import torch.distributed as dist
    from fiasco_amd.sharding import gather_streams, shard_indices

    if dry:
        dev = cdev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        cdev = torch.device("cpu") if same_gpu else dev      # where the collectives' tensors live
    if world > 1:
        if same_gpu or dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    st = st2 = None
    dt2 = 0.0
    t_stage = 0.0
    root = None
    if dry:
        barrier()
        t0 = time.perf_counter()
        out = [b"FIASCO\n" + bytes([rank, i & 255]) for i in range(F)]
        barrier()
        dt = time.perf_counter() - t0
    else:
        import fiasco_amd
        lib = fiasco_amd.library()
        lib.set_verbosity(0)
        lib.set_device(local)
        if max(a.width, a.height) > 2048:
            lib.set_limits(30000, 26)      # declared limits extension: 4K needs level 24
        opt = lib.cli_options()

        ts = time.perf_counter()
        batch = fiasco_amd.Batch(lib, frames, 20.0, opt)       # inputs now resident in HBM
        t_stage = time.perf_counter() - ts
        out = None
        for _ in range(a.warmup):
            out = batch.encode()
        # A launch this small runs several workgroups per frame; the coder's byte counters then cover
        # the chain workgroup only.  The algorithmic bytes of a frame do not depend on who computes
        # them: taken from one pass with one workgroup per frame.
        alg_override = None
        if lib.get_stats().spec_frames:
            os.environ["FIASCO_AMD_DEBUG"] = "1"; os.environ["FIASCO_AMD_SPEC"] = "0"
            b1 = fiasco_amd.Batch(lib, frames, 20.0, opt)
            lib.reset_stats(); b1.encode(); s1 = lib.get_stats(); b1.free()
            os.environ.pop("FIASCO_AMD_SPEC"); os.environ.pop("FIASCO_AMD_DEBUG")
            alg_override = float(s1.bytes_mp + s1.bytes_img + s1.bytes_gram) / max(s1.launches, 1)
        # ---- loop A: inputs resident in HBM ----
        barrier()
        lib.reset_stats()
        t0 = time.perf_counter()
        # K passes, pipelined: the device search of pass i+1 is started before the host writes
        # the .fco streams of pass i (fiasco_amd_batch_submit / _collect); every pass is complete
        # -- kernel, automaton download, entropy writer -- when the timed region ends
        batch.submit()
        for i in range(a.steps):
            out = batch.collect(resubmit=i + 1 < a.steps)
        barrier()
        dt = time.perf_counter() - t0
        st = lib.get_stats()
        root = batch.stats(0)          # coder-side error of the survey frame
        decoder = None
        try:                           # ... and its decoded PSNR (dfiasco -s 0 + pnmpsnr)
            decoded_psnr = batch.decode_psnr(0)[0][0] if rank == 0 else None
            if rank == 0 and not a.no_decoder:
                # all frames of the pass through the device decoder (one launch per level and
                # frame, csrc/hip/frame_decoder.inc) + download + the host's pnmpsnr sum -- outside the timed region
                lib.reset_stats()
                t_dec = time.time()
                good, ps, _ = batch.decode_psnr_all()
                t_dec = time.time() - t_dec
                dst = lib.get_stats()
                decoder = {"workload": "decode_image of the %d automata of one pass + PSNR against the inputs" % len(ps),
                           "frames": good, "seconds": round(t_dec, 3), "frames_per_s": round(good / t_dec, 1) if t_dec > 0 else None,
                           "psnr_db_min_max": [round(min(p[0] for p in ps), 2), round(max(p[0] for p in ps), 2)],
                           # device side alone (HIP events around the flights: automaton upload + one kernel per level +
                           # assembly): algorithmic bytes = 2 per level-image pixel written and per (pixel, term) read
                           "device_seconds": round(dst.decoder_us / 1e6, 4),
                           "device_frames_per_s": round(dst.decoder_frames / (dst.decoder_us / 1e6), 1) if dst.decoder_us else None,
                           "device_GBps": round(dst.decoder_bytes / (dst.decoder_us / 1e6) / 1e9, 1) if dst.decoder_us else None,
                           "device_frac_of_hbm_peak": round(dst.decoder_bytes / (dst.decoder_us / 1e6) / 8e12, 4) if dst.decoder_us else None,
                           "frame0_equals_single_call": abs(ps[0][0] - decoded_psnr) < 1e-12}
        except Exception as e:
            decoded_psnr = None
            decoder = {"error": str(e)[:200]}
        assert out is not None and all(o is not None for o in out), lib.error_message()
        # ---- loop B: the frames of every pass cross PCIe inside the timed region ----
        if not a.no_pcie_loop:
            def rot(r):
                r %= F
                return frames[r:] + frames[:r]
            if a.warmup:                   # pinned staging + device buffers are allocated here
                batch.upload(rot(0)); batch.encode()
            barrier()
            lib.reset_stats()
            t0 = time.perf_counter()
            batch.upload(rot(1))
            batch.submit()
            out2 = None
            for i in range(a.steps):
                if i + 1 < a.steps:
                    batch.upload(rot(i + 2))           # overlaps the running pass
                out2 = batch.collect(resubmit=i + 1 < a.steps)
            barrier()
            dt2 = time.perf_counter() - t0
            st2 = lib.get_stats()
            assert out2 is not None and all(o is not None for o in out2), lib.error_message()
            # the last pass encoded the list rotated by K: slot j holds frame (j + K) % F
            k = a.steps % F
            assert all(out2[j] == out[(j + k) % F] for j in range(0, F, max(1, F // 64))), \
                "pipelined pass did not encode the uploaded frames"
        batch.free()
        small = k4 = c5 = c3 = k4s = None
        if rank == 0 and world == 1 and a.k4_frames > 0 and (a.width, a.height) == (1920, 1080):
            k4 = k4_pass(lib, a.k4_frames, rank)
        if rank == 0 and world == 1 and not a.no_k4_small and (a.width, a.height) == (1920, 1080):
            k4s = k4_small_pass(lib)
        if rank == 0 and world == 1 and a.config3_frames > 0 and (a.width, a.height) == (1920, 1080):
            c3 = config3_pass(lib, a.config3_frames)
        if rank == 0 and world == 1 and a.config5_frames > 0 and (a.width, a.height) == (1920, 1080):
            c5 = config5_pass(lib, a.config5_frames)
        if rank == 0 and world == 1 and not a.no_small_launches and (a.width, a.height) == (1920, 1080):
            small = small_launches(lib, opt, uniq, a.width, a.height)

    t = torch.tensor([dt, dt2], dtype=torch.float64, device=cdev)
    agg = torch.tensor([float(st.kernel_ms) if st else 0.0,
                        float(st.bytes_mp + st.bytes_img + st.bytes_gram) if st else 0.0,
                        float(st.launches) if st else 0.0, float(st.frames) if st else 0.0,
                        float(st2.kernel_ms) if st2 else 0.0, float(st2.frames) if st2 else 0.0],
                       dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        # trivial gather of the per-rank streams over RCCL (outside the timed region)
        # global item r + i*W is frame i of rank r (round-robin, sharding.shard_indices)
        ng = min(total_job // world if a.scaling == "strong" else F, 64)      # a sample of every rank's streams (same count on all)
        keys = shard_indices(world * ng, rank, world)
        local_streams = {k: out[i] for i, k in enumerate(keys)}
        alls = gather_streams(local_streams, world * ng, device=cdev)
        assert all(s and s[:7] == b"FIASCO\n" for s in alls)
    dt, dt2 = [float(x) for x in t.tolist()]
    kernel_ms, alg_bytes, launches, nframes, kernel_ms2, nframes2 = [float(x) for x in agg.tolist()]

    if rank == 0:
        md5_ref = REF_MD5_SEED1234.get((a.width, a.height))
        if md5_ref and not dry:
            assert hashlib.md5(out[0]).hexdigest() == md5_ref, "parity lost: stream differs from the reference"
        total_frames = total_job * a.steps
        value = total_frames / dt if not dry else None
        per_launch_bytes = alg_bytes / max(launches, 1)
        if not dry and alg_override is not None and world == 1:
            per_launch_bytes = alg_override
        avg_kernel_s = (kernel_ms / 1e3) / max(launches, 1)
        achieved = per_launch_bytes / avg_kernel_s / 1e9 if avg_kernel_s else None
        res = {
            "metric": "grayscale frames/sec encoded",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch of %d independent %dx%d grayscale PGM frames per GPU (%d distinct "
                                   "seeds per GPU), cfiasco defaults (-q 20, block levels 6..10, 3 elements, "
                                   "small.fco basis, rle/adaptive models), bit-identical .fco streams"
                                   % (a.frames_per_gpu if a.scaling == "weak" else F, a.width, a.height, ndist)
                                   + ("" if a.scaling == "weak" else "; strong scaling: %d frames in all, dealt round robin "
                                                                      "to %d ranks" % (total_job, world)),
                       "frames_per_gpu": F, "distinct_frames_per_gpu": ndist, "frames_total_per_step": total_job,
                       # what a rank's launch looks like (the SCALE record of a strong-scaling run is read with this):
                       # kernel build, workgroups per frame, append helpers -- the launcher's policy for F frames on this GPU
                       "launch_form_per_rank": launch_form(lib, F, a.width, a.height) if not dry else None,
                       "parallelism": "frames x%d" % world,
                       "kernel_only_frames_per_s": nframes / (kernel_ms / 1e3) * world if kernel_ms else None,
                       # raw PNM in host memory -> parse -> pinned -> HBM inside the timed region,
                       # overlapped with the previous pass (fiasco_amd_batch_upload)
                       "pcie_inclusive_frames_per_s": total_frames / dt2 if dt2 else None,
                       "pcie_inclusive_kernel_only_frames_per_s":
                           nframes2 / (kernel_ms2 / 1e3) * world if kernel_ms2 else None,
                       "stage_seconds_first_batch": t_stage, "generate_seconds": t_gen,
                       "parity": ("stream md5 of survey frame == reference (%s)" % md5_ref[:12]) if md5_ref else None,
                       "estimated_psnr_db": root["psnr_db"] if root else None,
                       # decoded like `dfiasco -s 0`, compared like bin/pnmpsnr.c (reference tools on the
                       # reference's stream of the survey frame: 34.02 dB, tests/golden/MANIFEST.json)
                       "decoded_psnr_db": decoded_psnr if not dry else None,
                       "decoder": decoder if not dry else None,
                       # what `value` times and what it does not: the reference's own timer (codec/coder.c:709,884)
                       # also covers reading the PNM file and writing the .fco file
                       "value_covers": "device search + automaton download + host .fco entropy writer into memory, "
                                       "inputs resident in HBM; NOT PNM file read / parse / upload (see "
                                       "pcie_inclusive_frames_per_s) and NOT writing the .fco files to disk",
                       # BASELINE config 4's workload on this GPU: 256 x 3840x2160 in one launch (north-star
                       # target: >= 50 frames/s on one GPU)
                       "k4_frames_per_s": (k4 or {}).get("frames_per_s") if not dry else None,
                       "k4": k4 if not dry else None,
                       # BASELINE config 4 as written: 64 frames on this GPU, and 8 (one GPU's share on an 8-GPU node)
                       "k4_64": k4s if not dry else None,
                       # BASELINE config 3 as a batch: 1080p colour at the CLI defaults (limits extension)
                       "config3": c3 if not dry else None,
                       # BASELINE config 5's workload on this GPU (north-star row F3/F4: motion search, prediction, decoder)
                       "config5": c5 if not dry else None,
                       # launches that leave the chip empty: several workgroups per frame (speculation)
                       "small_launches": small if not dry else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None,
                         "traffic": pmc_traffic(F, a.width, a.height)[0],
                         "traffic_source": pmc_traffic(F, a.width, a.height)[1],
                         "kernel": ("fiasco_frame_kernel_spec" + ("_wide" if max(a.width, a.height) > 2048 else "")
                                    if st is not None and st.spec_frames else "fiasco_frame_kernel"),
                         "avg_launch_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": per_launch_bytes},
        }
        if dry:
            res["dry_run"] = True
        if world == 1 and not a.no_cpu_baseline and not dry:
            res["cpu_baseline"] = cpu_baseline(uniq[0], a.width, a.height)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
