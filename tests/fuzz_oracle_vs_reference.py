"""Developer tool (build container only): differential fuzzing of the CPU ORACLE against the REAL
reference coder (oracle/_ref/cfiasco_ref, built from /root/reference by oracle/ref_build.sh).

Random gray/colour images x random option sets the reference CLI can express (quality, -z 0..2,
dictionary size, RPF mantissas 2..8 and ranges, chroma options incl. dictionaries of 64..200 states, initial
bases, tiling options, 1..3-frame all-intra
streams; with FUZZ_VIDEO=1 also intra prediction, prediction levels and 2..5-frame streams with
P and B frames whose frames are displaced, noisy copies of the first) are encoded by both
command-line coders; the streams must be identical, or both must
fail.  Inputs on which the reference itself crashes are counted and skipped.  This widens the pin
of the oracle beyond the committed golden vectors; the device is tied to the oracle by
tests/fuzz_parity.py.

usage: fuzz_oracle_vs_reference.py [cases] [seed0] [workers]
"""
import os
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import synth  # noqa: E402
from fuzz_parity import random_image  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref")
# FUZZ_HUGE=1: images beyond 2048 pixels (up to 2600 x 1700) and the state budget of the declared limits
# extension -- against the limits-extension build of the reference (oracle/ref_build.sh:
# MAXSTATES 30000, MAXLEVEL 26), the oracle under the same limits
REF_BIG = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref_big")
ORA = os.path.join(ROOT, "oracle", "cfiasco_oracle")
HUGE = os.environ.get("FUZZ_HUGE") == "1"


def one(seed):
    rng = np.random.default_rng(seed)
    colour = bool(rng.integers(0, 4) == 0)
    nfr = int(rng.choice([1, 1, 1, 2, 3]))
    video = os.environ.get("FUZZ_VIDEO") == "1"
    pattern = "i"
    extra = []
    if video:
        pattern = str(rng.choice(["i", "ip", "ipp", "ippp", "ibp", "ibbp", "ipb", "ib", "ipbbp", "ibpbp"]))
        nfr = 1 if pattern == "i" else int(rng.integers(2, 6))
        if rng.integers(0, 2):
            extra.append("--prediction")
        if rng.integers(0, 3) == 0:
            lo = int(rng.integers(4, 11))
            extra += ["--min-level", str(lo), "--max-level", str(int(rng.integers(lo, 13)))]
    args = ["-q", str(rng.choice([1, 2, 5, 8, 20, 45, 60, 90, 99])), "-z", str(int(rng.integers(0, 3))),
            "--dictionary-size", str(rng.choice([1, 8, 40, 300, 10000])),
            "--rpf-mantissa", str(int(rng.integers(2, 9))), "--dc-rpf-mantissa", str(int(rng.integers(2, 9))),
            "--rpf-range", str(rng.choice([0.75, 1.0, 1.5, 2.0])), "--dc-rpf-range", str(rng.choice([0.75, 1.0, 1.5, 2.0])),
            "--chroma-qfactor", str(rng.choice([1.0, 2.0, 3.5])), "--chroma-dictionary", str(rng.choice([1, 5, 40, 63, 64, 100, 200])),
            "--tiling-exponent", str(int(rng.integers(0, 6))), "--pattern", pattern] + extra
    # round 5: initial bases whose edge lists run past MAXEDGES -- our own (tests/golden/long_*.fco) and the
    # reference's medium.fco / large.fco (oracle/_ref/share, installed by oracle/ref_build.sh)
    basis = str(rng.choice(["", "", "", "long_a.fco", "long_b.fco", "long_c.fco", "medium.fco", "large.fco"]))
    if basis:
        args += ["--basis-name", basis]
    ref, ora_extra = REF, []
    if HUGE:
        ref, ora_extra = REF_BIG, ["--limit-states", "30000", "--limit-level", "26"]
        colour = bool(rng.integers(0, 6) == 0)
        nfr = int(rng.choice([1, 1, 1, 2]))
    with tempfile.TemporaryDirectory() as td:
        names = []
        first = random_image(rng, colour)
        hdr = first.split(b"\n")[1].split()
        w, h = int(hdr[0]), int(hdr[1])
        for f in range(nfr):
            p = os.path.join(td, "f%d.%s" % (f, "ppm" if colour else "pgm"))
            if f == 0:
                open(p, "wb").write(first)
            elif video and rng.integers(0, 5):
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
        env = dict(os.environ, FIASCO_DATA=os.path.join(HERE, "golden") + ":" + os.path.join(ROOT, "oracle", "_ref", "share"))
        r = subprocess.run([ref, "--progress-meter", "0"] + args + ["-o", os.path.join(td, "r.fco")] + names,
                           env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode < 0 or r.returncode >= 128:
            return seed, "refcrash", args
        o = subprocess.run([ORA, "--progress-meter", "0"] + ora_extra + args + ["-o", os.path.join(td, "o.fco")] + names,
                           env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode != 0 or o.returncode != 0:
            return seed, ("bothfail" if r.returncode != 0 and o.returncode != 0 else "MISMATCH rc %d/%d" % (r.returncode, o.returncode)), args
        same = open(os.path.join(td, "r.fco"), "rb").read() == open(os.path.join(td, "o.fco"), "rb").read()
        return seed, ("ok" if same else "MISMATCH bytes (%dx%d colour %s frames %d)" % (w, h, colour, nfr)), args


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    tally = {}
    with ThreadPoolExecutor(workers) as ex:
        for seed, res, args in ex.map(one, range(seed0, seed0 + n)):
            key = res.split()[0]
            tally[key] = tally.get(key, 0) + 1
            if key not in ("ok", "bothfail"):
                print("seed %d: %s  %s" % (seed, res, " ".join(args)), flush=True)
    print("oracle vs reference: %d cases from seed %d: %s" % (n, seed0, tally))
    return 1 if "MISMATCH" in tally else 0


if __name__ == "__main__":
    sys.exit(main())
