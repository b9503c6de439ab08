"""Host-side behaviour of the C API: options object, error convention, exports.

These mirror the rules of reference codec/options.c:139-708 and codec/coder.c:85-182
(return 1 = ok, 0 = failure + fiasco_get_error_message()).  No compute on a GPU here.
"""
import ctypes
import os
import subprocess

import pytest

import fiasco_amd


def test_library_exports_every_declared_symbol(product):
    for name in fiasco_amd.EXPORTED_SYMBOLS:
        assert hasattr(product.L, name), name


def test_headers_and_symbol_list_agree():
    """Every function declared in include/*.h is in EXPORTED_SYMBOLS (and so checked above)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    decl = set()
    for h in ("libfiasco_amd.h", "libfiasco_amd_hip.h"):
        src = open(os.path.join(root, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        decl |= set(re.findall(r"\b((?:fiasco_|fa_core_|open_file)\w*)\s*\(", src))
    decl -= {"fiasco_c_options"}
    assert decl <= set(fiasco_amd.EXPORTED_SYMBOLS), decl - set(fiasco_amd.EXPORTED_SYMBOLS)


def test_workgroups_per_frame_of_a_launch(product):
    """The launcher's policy for small launches (a pure function, no device): a CU per
    workgroup down to five per frame, then shared CUs -- 256-thread build only -- down to three per frame
    and three per CU; 4K frames never share a CU."""
    f = product.L.fiasco_amd_spec_workgroups
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    narrow = lambda n: f(n, 256, 0, 1, 4)
    # round 6 (hand-offs with one releasing lane): the 256-thread build fills the chip's four workgroups per CU, but
    # stays at five per frame once the launch passes 2.5 workgroups per CU (tests/gpu_spec_policy_sweep.sh)
    assert [narrow(n) for n in (1, 16, 32, 48, 64, 80)] == [8, 8, 8, 8, 8, 8]
    assert [narrow(n) for n in (81, 96, 106, 107, 128, 192, 204, 205, 256, 257, 341, 342, 1024)] == [7, 6, 6, 5, 5, 5, 5, 4, 4, 3, 3, 0, 0]
    # frames that may need the 1024-thread build, or a build a CU holds only one of: a CU per workgroup
    assert [f(n, 256, 0, 0, 5) for n in (51, 64, 85, 86)] == [5, 4, 3, 0]
    assert [f(n, 256, 0, 1, 1) for n in (64, 85, 86)] == [4, 3, 0]
    # 4K: a CU per workgroup, like the others (until round 5: at most half the CUs once a frame had four)
    assert [f(n, 256, 1, 0, 1) for n in (1, 8, 16, 32, 33, 48, 64, 85, 86)] == [8, 8, 8, 8, 7, 5, 4, 3, 0]
    # ... and append helpers for the chains where CUs are left (fiasco_amd_spec_append_helpers: frames, CUs, G, wide build)
    h = product.L.fiasco_amd_spec_append_helpers
    h.restype = ctypes.c_int
    h.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    assert [h(n, 256, 8, 1) for n in (1, 8, 16, 23, 24, 25, 26, 32)] == [3, 3, 3, 3, 2, 2, 0, 0]
    assert h(0, 256, 8, 1) == 0 and h(8, 256, 1, 1) == 0
    # the 256-thread build: three while the launch stays below 1.5 workgroups per CU
    assert [h(n, 256, g, 0) for n, g in ((1, 8), (16, 8), (32, 8), (34, 7), (35, 7), (64, 5), (128, 3))] == [3, 3, 3, 3, 3, 0, 0]
    assert f(0, 256, 0, 1, 5) == 0 and f(4, 0, 0, 1, 5) == 0


def test_workgroups_for_the_tables_of_big_frames(product):
    """The launcher's policy for big frames (prediction, P/B frames, -z 1/2) that leave the chip empty: several
    workgroups build the <sub-block, state> tables of a frame (csrc/hip/frame_coder.h FcCoop).  Frames are launched in
    groups of eight, every workgroup must be resident at one per CU: BASELINE config 5 (30 GOPs side by side) gets 8."""
    f = product.L.fiasco_amd_coop_workgroups
    f.restype = ctypes.c_uint
    f.argtypes = [ctypes.c_uint, ctypes.c_int]
    assert [f(n, 256) for n in (1, 8, 30, 32, 33, 64, 65, 128, 129, 256)] == [8, 8, 8, 8, 4, 4, 2, 2, 1, 1]
    assert [f(n, 64) for n in (1, 8, 9, 16, 17, 32, 33)] == [8, 8, 4, 4, 2, 2, 1]


def test_jobs_of_a_gop_are_dealt_by_the_gop_not_by_their_place_in_the_call(product):
    """Multi-device hygiene (a 1-GPU box cannot run it, a pure function can be checked
    anywhere): the device share of a job is fa_share_of(key, index, shares) (csrc/host/fa_host.h) for the search
    (fa_core_stage) and for the decoder (fa_core_decode_frames) alike.  Without a key: round robin by index.  With a
    key -- the sequence engine passes the GOP number for every frame of a video and for the decode of its reference
    frames (fa_sequence.c) --: a function of the key and the number of devices ALONE, so a GOP keeps its device when
    earlier GOPs end, when the probe's result stands in for job 0, and whatever the size of the step's batch."""
    f = product.L.fiasco_amd_share_of
    f.restype = ctypes.c_uint
    f.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
    for shares in (1, 2, 3, 8):
        assert [f(0, i, shares) for i in range(20)] == [i % shares for i in range(20)]
        for gop in range(40):
            assert {f(gop + 1, i, shares) for i in range(64)} == {gop % shares}      # the index does not matter
    # 30 GOPs on 8 devices: steps in which GOPs 0..4 have ended (shorter GOPs) -- the others stay where they were
    full = {g: f(g + 1, g, 8) for g in range(30)}
    later = {g: f(g + 1, i, 8) for i, g in enumerate(range(5, 30))}
    assert all(later[g] == full[g] for g in later)
    assert sorted(set(full.values())) == list(range(8))                              # and all devices are used
    assert f(5, 3, 0) == 0


def test_no_oracle_in_product():
    """The product library must not link or contain the CPU oracle."""
    out = subprocess.run(["nm", "-D", "--defined-only", fiasco_amd.LIB_PATH], capture_output=True, text=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if line.strip()]
    assert "fiasco_coder" in names and "fiasco_amd_core_name" in names
    # drop-in hygiene (csrc/exports.map): only the public names are dynamic symbols -- fiasco.h's API, the two
    # symbols the reference CLI objects import, the fiasco_amd_* extensions and the seam fa_core_* that INTEGRATION.md 3
    # documents; no other fa_*, no fc_*, no device stubs
    ok = lambda n: n.startswith("fiasco_") or n == "open_file" or n.startswith("fa_core_")
    assert all(ok(n) for n in names), [n for n in names if not ok(n)]
    # ... and the seam is the HIP core's
    assert "fa_core_encode_frames" in names
    dump = open(fiasco_amd.LIB_PATH, "rb").read()
    assert b"oracle-cpu" not in dump and b"oracle_core" not in dump
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "fiasco_amd")):
        for f in files:
            if f.endswith(".py") or f == "Makefile":       # build recipes and python plumbing
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt, f
            elif f.endswith((".c", ".cpp", ".hip")):        # sources: no include / link of it
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle_core" not in txt, f


def test_reference_side_code_links_against_the_seam(product, tmp_path):
    """a maintainer replaces subdivide() inside the reference tree by a call of the seam
    (fa_core_encode_frames, or the staged fa_core_stage / _run / _submit / _finish2 / _unstage, and
    fa_core_decode_frames for the reference frames of a video).  Such a build links libfiasco_amd.so: the seam must
    be among its dynamic symbols (csrc/exports.map).  Link only -- nothing is called without a GPU."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "seam_user.c"
    src.write_text("""
#include "fa_host.h"
/* what a patched codec/coder.c would reference */
void *seam[] = { (void *) fa_core_encode_frames, (void *) fa_core_stage, (void *) fa_core_run, (void *) fa_core_submit,
                 (void *) fa_core_finish2, (void *) fa_core_finish, (void *) fa_core_unstage, (void *) fa_core_upload_buffer,
                 (void *) fa_core_upload_commit, (void *) fa_core_decode_frames, (void *) fa_core_release_dev,
                 (void *) fa_core_name };
int main (void) { return seam[0] == 0; }
""")
    exe = str(tmp_path / "seam_user")
    r = subprocess.run(["gcc", "-std=gnu99", "-I" + os.path.join(root, "fiasco_amd", "csrc", "host"), "-I" + os.path.join(root, "include"),
                        "-o", exe, str(src), "-L" + os.path.dirname(fiasco_amd.LIB_PATH), "-lfiasco_amd",
                        "-Wl,-rpath," + os.path.dirname(fiasco_amd.LIB_PATH)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_option_validation_messages(product):
    o = product.c_options_new()
    cases = [
        ("set_optimizations", (6, 10, 3, 0, 0), "Size of dictionary has to be a positive number."),
        ("set_optimizations", (6, 10, 0, 100, 0), "At least one dictionary element has to be used in an approximation."),
        ("set_optimizations", (6, 3, 3, 100, 0), "Maximum image block size has to be at least level 4."),
        ("set_optimizations", (3, 10, 3, 100, 0), "Minimum image block size has to be at least level 4."),
        ("set_optimizations", (8, 6, 3, 100, 0), "Maximum block size has to be larger or equal minimum block size."),
        ("set_prediction", (0, 6, 5), "Maximum prediction block size has to be at least level 6"),
        ("set_quantization", (1, 1, 5, 1), "Number of RPF mantissa bits `1', `5' have to be in the interval [2,8]."),
        ("set_quantization", (3, 7, 5, 1), "Invalid RPF ranges `7', `1' specified."),
        ("set_chroma_quality", (2.0, 0), "Size of chroma compression dictionary has to be a positive number."),
        ("set_chroma_quality", (0.0, 40), "Quality of chroma channel compression has to be positive value."),
        ("set_frame_pattern", ("ipx",), "Frame type pattern contains invalid character `x' (choose I, B or P)."),
        ("set_frame_pattern", ("",), "Frame type pattern doesn't contain any character."),
        ("set_smoothing", (101,), "Smoothing percentage must be in the range [-1, 100]."),
        ("set_progress_meter", (7,), "Invalid progress meter `7' specified (valid values are 0, 1, or 2)."),
        ("set_tiling", (9, 4), "Invalid tiling method `9' specified (valid methods are 0, 1, 2, or 3)."),
        ("set_basisfile", ("/nonexistent/basis.fco",), None),
    ]
    for name, args, msg in cases:
        with pytest.raises(fiasco_amd.FiascoError) as e:
            getattr(o, name)(*args)
        if msg is not None:
            assert str(e.value) == msg
    # valid calls return 1
    assert o.set_optimizations(6, 10, 3, 10000, 0) == 1
    assert o.set_title("t") == 1 and o.set_comment("c") == 1
    o.delete()


def test_wrong_options_object_is_rejected(product):
    """cast_c_options(): the private pointer must carry the COFIASCO tag (options.c:682-708)."""
    class Fake(ctypes.Structure):
        _fields_ = [("fn", ctypes.c_void_p * 13), ("private", ctypes.c_void_p)]
    buf = ctypes.create_string_buffer(b"NOTFIASCO" + b"\0" * 64)
    fake = Fake()
    fake.private = ctypes.cast(buf, ctypes.c_void_p)
    rc = product.L.fiasco_c_options_set_smoothing(ctypes.byref(fake), 70)
    assert rc == 0
    assert product.error_message() == "Parameter `options' doesn't match required type."


def test_coder_parameter_errors(product, tmp_path):
    o = product.cli_options()
    assert product.fiasco_coder([str(tmp_path / "missing.pgm")], str(tmp_path / "o.fco"), 20.0, o) == 0
    assert "Can't open frame" in product.error_message()
    assert product.fiasco_coder(["x.pgm"], str(tmp_path / "o.fco"), 0.0, o) == 0
    assert product.error_message() == "Compression quality has to be positive."
    bad = tmp_path / "bad.pgm"
    bad.write_bytes(b"P3\n32 32\n255\n" + b"0" * 100)
    assert product.fiasco_coder([str(bad)], str(tmp_path / "o.fco"), 20.0, o) == 0
    assert "image format 'P3' not supported" in product.error_message()
    small = tmp_path / "small.pgm"
    small.write_bytes(b"P5\n16 16\n255\n" + bytes(256))
    assert product.fiasco_coder([str(small)], str(tmp_path / "o.fco"), 20.0, o) == 0
    assert "has to be at least 32 pixels" in product.error_message()
    odd = tmp_path / "odd.pgm"
    odd.write_bytes(b"P5\n33 32\n255\n" + bytes(33 * 32))
    assert product.fiasco_coder([str(odd)], str(tmp_path / "o.fco"), 20.0, o) == 0
    assert product.error_message() == "Width and height of images must be even numbers."
    o.delete()


def test_limits_extension_api(product):
    assert product.get_limits() == (6000, 22)          # stock MAXSTATES / MAXLEVEL
    product.set_limits(30000, 26)
    assert product.get_limits() == (30000, 26)
    with pytest.raises(fiasco_amd.FiascoError):
        product.set_limits(8, 22)
    product.set_limits(6000, 22)


def test_4k_needs_the_declared_limits_extension(product, tmp_path):
    """The stock reference cannot code level-24 images (SURVEY finding 2); this library
    reports it instead of crashing."""
    p = tmp_path / "k.pgm"
    p.write_bytes(b"P5\n3840 2160\n255\n" + bytes(3840 * 2160))
    o = product.cli_options()
    assert product.fiasco_coder([str(p)], str(tmp_path / "k.fco"), 20.0, o) == 0
    assert "exceeds MAXLEVEL 22" in product.error_message()
    o.delete()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_product_fails_loudly_without_gpu(product, inputs, tmp_path):
    """No CPU fallback: without a HIP device the hot path refuses to run."""
    o = product.cli_options()
    rc = product.fiasco_coder([inputs.path("g96x64")], str(tmp_path / "o.fco"), 20.0, o)
    assert rc == 0
    assert "no HIP device available" in product.error_message()
    out = product.encode_batch([inputs.data("g96x64")], 20.0, o)
    assert out == [None]
    o.delete()


def test_input_template_expansion(oracle, inputs, tmp_path):
    """prefix[start-end]suffix templates (codec/coder.c:390-488) == explicit name list."""
    import shutil
    shutil.copy(inputs.path("f0_96x64"), tmp_path / "fr0.pgm")
    shutil.copy(inputs.path("f1_96x64"), tmp_path / "fr1.pgm")
    o = oracle.cli_options(pattern="i")
    assert oracle.fiasco_coder([str(tmp_path / "fr[0-1].pgm")], str(tmp_path / "a.fco"), 20.0, o) == 1
    assert oracle.fiasco_coder([str(tmp_path / "fr0.pgm"), str(tmp_path / "fr1.pgm")],
                               str(tmp_path / "b.fco"), 20.0, o) == 1
    assert (tmp_path / "a.fco").read_bytes() == (tmp_path / "b.fco").read_bytes()
    o.delete()


def test_unsupported_modes_are_errors_not_silence(oracle, product, inputs, tmp_path):
    """Half-pixel vectors: the reference's own half-pixel path reads outside the reference frame
    (codec/motion.c:271 divides the vector after its conversion to unsigned), so there is nothing
    to be compatible with -- refused with a message, before the output file is touched.  The
    product without a GPU refuses everything loudly (no CPU fallback)."""
    for lib in (oracle, product):
        o = lib.cli_options()            # default pattern ippppppppp -> 2nd frame is a P frame
        o.set_video_param(25, 1, 0, 1)
        out = tmp_path / ("v_%s.fco" % lib.core_name())
        out.write_bytes(b"keep")
        rc = lib.fiasco_coder([inputs.path("f0_96x64"), inputs.path("f1_96x64")], str(out), 20.0, o)
        assert rc == 0 and "Half-pixel" in lib.error_message()
        o.delete()


def test_coding_order_of_b_frames(oracle, inputs, tmp_path):
    """video_coder (codec/coder.c:490-668) codes the future reference of a run of B frames
    first: with pattern ibbp the frame numbers in the stream are 0, 3, 1, 2."""
    names = [inputs.path("m%d_128x96" % i) for i in range(4)]
    o = oracle.cli_options(pattern="ibbp")
    out = str(tmp_path / "b.fco")
    assert oracle.fiasco_coder(names, out, 20.0, o) == 1, oracle.error_message()
    o.delete()
    golden = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seq4_gray_ibbp.fco"), "rb").read()
    assert open(out, "rb").read() == golden


def test_submit_collect_on_oracle_seam(oracle, inputs):
    """The pipelined batch entry points exist above every core (the oracle's submit is a
    no-op): submit + collect == encode."""
    import fiasco_amd
    frames = [inputs.data(n) for n in ("g96x64", "g160x120")]
    b = fiasco_amd.Batch(oracle, frames, 20.0, oracle.cli_options())
    ref = b.encode()
    b.submit()
    assert b.collect(resubmit=True) == ref
    assert b.collect() == ref
    b.free()


def check_upload_replaces_inputs(lib, inputs):
    """fiasco_amd_batch_upload: a staged batch takes new frames of the same geometry; what the
    next pass returns is what a fresh batch of those frames returns, pass after pass."""
    import fiasco_amd
    import synth
    o = lib.cli_options()
    sets = [[synth.pgm_bytes(synth.synth(96, 64, 10 * k + i)) for i in range(9)] for k in range(3)]
    want = [lib.encode_batch(s, 20.0, o) for s in sets]
    assert all(None not in w for w in want)
    b = fiasco_amd.Batch(lib, sets[0], 20.0, o)
    assert b.encode() == want[0]
    b.upload(sets[1])
    assert b.encode() == want[1]
    # pipelined: the frames of pass i + 1 are handed over while pass i is in flight
    b.submit()
    b.upload(sets[2])
    assert b.collect(resubmit=True) == want[1]
    b.upload(sets[0])
    assert b.collect(resubmit=True) == want[2]
    assert b.collect() == want[0]
    # a frame of another size is refused and the batch keeps its frames
    bad = list(sets[1]); bad[3] = synth.pgm_bytes(synth.synth(64, 64, 1))
    with pytest.raises(fiasco_amd.FiascoError):
        b.upload(bad)
    assert "size" in lib.error_message()
    assert b.encode() == want[0]
    b.free(); o.delete()


def test_upload_on_oracle_seam(oracle, inputs):
    check_upload_replaces_inputs(oracle, inputs)


# root-range figures printed by the real reference (`cfiasco -V 2`, codec/coder.c:918-923) for
# two committed inputs: (squared error, total costs) per band
REFERENCE_STATS = {
    "g256": [(3135488.50, 4090260.50)],
    "c00": [(2026837.50, 2934227.75), (2232672.50, 3808977.50), (2250933.50, 3659728.75)],
}


def check_reference_stats(lib, inputs):
    import fiasco_amd
    o = lib.cli_options()
    b = fiasco_amd.Batch(lib, [inputs.data("g256"), inputs.data("c00")], 20.0, o)
    assert None not in b.encode()
    for i, name in enumerate(("g256", "c00")):
        for band, (err, costs) in enumerate(REFERENCE_STATS[name]):
            st = b.stats(i, band)
            assert st is not None and (st["err"], st["costs"]) == (err, costs), (name, band, st)
    assert b.stats(0, 1) is None and b.stats(2) is None and b.stats(1, 3) is None
    assert abs(b.stats(0)["psnr_db"] - 31.33) < 0.005      # "PSNR: 31.33 dB" in the reference's log
    b.free(); o.delete()


def test_batch_stats_match_reference_log(oracle, inputs):
    check_reference_stats(oracle, inputs)


def test_stage1_pricing_restructuring(tmp_path):
    """The device prices the run-length part of a candidate from per-step uniform parts
    (mp_device.inc, StepCtx); tests/stage1_pricing_check.c replays that algebra on the CPU
    against the plain walk over the merged position list -- every float must be identical."""
    import subprocess
    exe = str(tmp_path / "s1check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage1_pricing_check.c")
    subprocess.check_call(["gcc", "-O1", "-ffp-contract=off", "-o", exe, src])
    out = subprocess.run([exe, "400000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " 0 mismatches" in out.stdout


def test_unmodified_reference_cli_links_against_this_library(product, oracle, inputs, tmp_path):
    """INTEGRATION.md 1 as a test: the object files of the UNMODIFIED reference CLI
    (bin/cwfa.c, params.c, binerror.c, getopt*.c, compiled by oracle/ref_build.sh) link against
    libfiasco_amd.so with no unresolved symbol, and -- linked against the oracle library, which
    shares every line of host code with the product -- produce the reference's golden stream."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = [os.path.join(root, "oracle", "_ref", "obj", "bin_%s.o" % n)
            for n in ("cwfa", "params", "binerror", "getopt", "getopt1")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("reference objects not built here (oracle/ref_build.sh needs /root/reference)")
    exe = str(tmp_path / "cfiasco_product")
    r = subprocess.run(["gcc", "-o", exe] + objs + ["-L" + os.path.join(root, "fiasco_amd"), "-lfiasco_amd",
                        "-Wl,-rpath," + os.path.join(root, "fiasco_amd"), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr                       # no unresolved symbol
    env = dict(os.environ, FIASCO_DATA=os.path.join(root, "fiasco_amd", "data"))
    if not os.path.exists("/dev/kfd"):
        out = subprocess.run([exe, "--progress-meter", "0", "-o", str(tmp_path / "p.fco"), inputs.path("g256")],
                             env=env, capture_output=True, text=True)
        assert out.returncode != 0 and "no HIP device" in out.stderr     # loud, no fallback
    exe2 = str(tmp_path / "cfiasco_oracle_cli")
    r = subprocess.run(["gcc", "-o", exe2] + objs + ["-L" + os.path.join(root, "oracle"), "-loracle_fiasco",
                        "-Wl,-rpath," + os.path.join(root, "oracle"), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe2, "--progress-meter", "0", "-o", str(tmp_path / "o.fco"), inputs.path("g256")],
                         env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    golden = open(os.path.join(root, "tests", "golden", "g256_q20.fco"), "rb").read()
    assert (tmp_path / "o.fco").read_bytes() == golden
