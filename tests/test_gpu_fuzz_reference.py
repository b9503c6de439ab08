"""Differential fuzz of the DEVICE coder against the REAL reference coder, on the GPU box (-m gpu).

Why this exists: the CPU oracle and the product share their host C (PNM
reader, option mapping, sequence engine, `.fco' writer -- oracle/Makefile links fiasco_amd/csrc/host/*.c),
so a device-vs-oracle comparison on fresh seeds cannot see an error there: both sides would write the
same wrong bytes.  The real reference binary travels to the GPU box as build output of
oracle/ref_build.sh (oracle/_ref/cfiasco_ref, git-ignored, not gpurun-ignored): here both command-line
coders -- the reference's, and the UNMODIFIED reference CLI objects (bin/cwfa.c, params.c ... as compiled
by oracle/ref_build.sh) linked against libfiasco_amd.so -- encode the same random inputs with the same
random arguments.  The option mapping is shared by construction (it IS the reference's,
/root/reference/bin/cwfa.c:252-393); everything below fiasco_coder() is the reference on one side and this
repository's host C + HIP kernels on the other.  The streams must be byte-identical, or both must fail.

Nothing under oracle/ but the reference's own binaries is involved; skipped cleanly when they did not travel.
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import synth
from conftest import GOLDEN, REF_SHARE, ROOT
from fuzz_parity import random_image

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref")
CLI_OBJS = [os.path.join(ROOT, "oracle", "_ref", "obj", "bin_%s.o" % n)
            for n in ("cwfa", "params", "binerror", "getopt", "getopt1")]
N_CASES = int(os.environ.get("FUZZ_REF_CASES", "40"))        # long runs of the round: FUZZ_REF_CASES=300 FUZZ_REF_SEED=...
SEED0 = int(os.environ.get("FUZZ_REF_SEED", "60600"))
# seeds of the long runs of round 6 (2 x 400 cases, seeds 71000.. and 82000..) on which the device differed from the
# reference -- both times a REFUSAL where the reference codes the stream (an unsigned prediction-window check; a missing
# reference frame that no motion search can reach): always run
PINNED_SEEDS = [71064, 71136]


@pytest.fixture(scope="module")
def product_cli(product, tmp_path_factory):
    """The reference's CLI objects linked against the product library (as in
    test_unmodified_reference_cli_encodes_on_the_gpu)."""
    assert os.path.exists("/dev/kfd"), "no GPU on this box"
    if not (os.path.exists(REF) and all(os.path.exists(o) for o in CLI_OBJS)):
        pytest.skip("the reference's binaries did not travel (oracle/ref_build.sh builds them in the build container)")
    exe = str(tmp_path_factory.mktemp("cli") / "cfiasco_reference_cli_on_product")
    r = subprocess.run(["gcc", "-o", exe] + CLI_OBJS + ["-L" + os.path.join(ROOT, "fiasco_amd"), "-lfiasco_amd",
                        "-Wl,-rpath," + os.path.join(ROOT, "fiasco_amd"), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def make_case(seed, td):
    """Random input files + cfiasco arguments: sizes 32 .. 400, gray / colour, 1..4-frame streams (all intra,
    P and B frames), -q, -z 0..2, --prediction and block-level windows, mantissas 2..8, ranges, chroma
    options incl. dictionaries past 63 states, the initial bases (small / medium / large + our long ones)."""
    rng = np.random.default_rng(seed)
    colour = bool(rng.integers(0, 3) == 0)
    video = bool(rng.integers(0, 2))
    if os.environ.get("FUZZ_REF_VIDEO") == "1":         # long runs: colour videos (the level ratchet, the reference frames)
        colour = bool(rng.integers(0, 3) != 0)
        video = True
    pattern, nfr, extra = "i", int(rng.choice([1, 1, 1, 2, 3])), []
    if video:
        pattern = str(rng.choice(["ip", "ipp", "ippp", "ibp", "ibbp", "ipb", "ipbbp"]))
        nfr = int(rng.integers(2, 5))
    if rng.integers(0, 2):
        extra.append("--prediction")
    if rng.integers(0, 3) == 0:
        lo = int(rng.integers(6, 11))             # prediction window (bin/cwfa.c:351-359: both at least level 6)
        extra += ["--min-level", str(lo), "--max-level", str(int(rng.integers(lo, 13)))]
    args = ["-q", str(rng.choice([1, 2, 5, 8, 20, 45, 60, 90, 99])), "-z", str(int(rng.integers(0, 3))),
            "--dictionary-size", str(rng.choice([1, 8, 40, 300, 10000])),
            "--rpf-mantissa", str(int(rng.integers(2, 9))), "--dc-rpf-mantissa", str(int(rng.integers(2, 9))),
            "--rpf-range", str(rng.choice([0.75, 1.0, 1.5, 2.0])), "--dc-rpf-range", str(rng.choice([0.75, 1.0, 1.5, 2.0])),
            "--chroma-qfactor", str(rng.choice([1.0, 2.0, 3.5])), "--chroma-dictionary", str(rng.choice([1, 5, 40, 63, 64, 100, 200])),
            "--tiling-exponent", str(int(rng.integers(0, 6))), "--pattern", pattern] + extra
    basis = str(rng.choice(["", "", "", "long_a.fco", "long_b.fco", "long_c.fco", "medium.fco", "large.fco"]))
    if basis:
        args += ["--basis-name", basis]
    first = random_image(rng, colour)                    # 32 .. 398 x 32 .. 318
    hdr = first.split(b"\n")[1].split()
    w, h = int(hdr[0]), int(hdr[1])
    names = []
    for f in range(nfr):
        p = os.path.join(td, "s%d_f%d.%s" % (seed, f, "ppm" if colour else "pgm"))
        if f == 0:
            open(p, "wb").write(first)
        elif rng.integers(0, 5):
            # a displaced, slightly noisy copy of the first frame: motion compensation can win
            hl = len(first) - w * h * (3 if colour else 1)
            a0 = np.frombuffer(first[hl:], np.uint8).reshape((h, w, 3) if colour else (h, w))
            a = np.roll(a0, (int(rng.integers(-5, 6)) * f, int(rng.integers(-5, 6)) * f), (0, 1)).astype(np.int32)
            a = np.clip(a + rng.integers(-3, 4, a.shape), 0, 255).astype(np.uint8)
            (synth.write_ppm if colour else synth.write_pgm)(p, a)
        else:
            a = rng.integers(0, 256, (h, w, 3) if colour else (h, w)).astype(np.uint8)
            (synth.write_ppm if colour else synth.write_pgm)(p, a)
        names.append(p)
    return args, names, (w, h, colour, nfr)


def run(exe, args, names, out, env):
    r = subprocess.run([exe, "--progress-meter", "0"] + args + ["-o", out] + names, env=env,
                       capture_output=True, text=True, timeout=600)
    data = open(out, "rb").read() if r.returncode == 0 and os.path.exists(out) else None
    return r, data


def test_device_equals_the_real_reference_on_random_cases(product_cli, tmp_path):
    env = dict(os.environ, FIASCO_DATA=GOLDEN + ":" + REF_SHARE)
    env.pop("FIASCO_AMD_DEBUG", None)                  # a user's environment: no developer switch
    td = str(tmp_path)
    tally = {"ok": 0, "bothfail": 0, "refcrash": 0}
    bad = []
    for seed in list(range(SEED0, SEED0 + N_CASES)) + PINNED_SEEDS:
        args, names, shape = make_case(seed, td)
        r, want = run(REF, args, names, os.path.join(td, "r%d.fco" % seed), env)
        if r.returncode < 0 or r.returncode >= 128:
            tally["refcrash"] += 1                      # the reference itself crashed on this input: nothing to compare
            continue
        p, got = run(product_cli, args, names, os.path.join(td, "p%d.fco" % seed), env)
        if want is None and got is None and p.returncode > 0:
            tally["bothfail"] += 1
        elif want is not None and got == want:
            tally["ok"] += 1
        else:
            bad.append((seed, shape, " ".join(args), r.returncode, p.returncode,
                        None if want is None else hashlib.md5(want).hexdigest(),
                        None if got is None else hashlib.md5(got).hexdigest(), p.stderr[-300:]))
    print("device vs REAL reference: %d cases from seed %d + %d pinned: %s" % (N_CASES, SEED0, len(PINNED_SEEDS), tally))
    if bad:                                             # the whole story, not pytest's shortened repr
        import json
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(bad, open(os.path.join(ROOT, "gpurun_out", "fuzz_reference_failures.json"), "w"), indent=1)
        for b in bad:
            print("MISMATCH seed %d %s: %s\n   reference rc %d md5 %s, device rc %d md5 %s: %s" % (b[0], b[1], b[2], b[3], b[5], b[4], b[6], b[7]))
    assert not bad, [b[0] for b in bad]
    assert tally["ok"] >= N_CASES * 3 // 4, tally          # the comparison must not be hollow
