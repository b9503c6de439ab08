"""Developer tool (GPU box): differential fuzzing of the device coder against the CPU oracle.

Random gray/colour images of random (even) sizes, qualities and option sets -- everything the
device scope covers: block-level windows 4..12, 1..5 vectors, second-domain retry, dictionary
sizes, RPF mantissas/ranges, chroma options -- encoded by both libraries through
fiasco_amd_encode_batch; any byte difference is reported with the seed that reproduces it.

usage: fuzz_parity.py [rounds] [frames_per_round] [seed0]
environment: FUZZ_SPEC=1 only option sets of the default geometry (block levels 6..10, <= 3 vectors, no
retries: the kernel builds that give a frame several workgroups), every round with a random split of
the workgroups (block-level speculation); FUZZ_PRED=1 also intra prediction with random level windows; FUZZ_BIG=1 sizes up to 1000 x 800; FUZZ_REPL=n every frame n times in one launch (the
replicas must agree: a full device exposes timing-dependent faults that single frames hide)
"""
import os
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiasco_amd
import synth

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FIASCO_DATA", os.path.join(_ROOT, "tests", "golden") + ":" + os.path.join(_ROOT, "oracle", "_ref", "share"))


def random_image(rng, colour):
    big = int(os.environ.get("FUZZ_BIG", "0"))          # FUZZ_BIG=1: sizes up to 1000 x 800
    w = int(rng.integers(16, 500 if big else 200)) * 2
    h = int(rng.integers(16, 400 if big else 160)) * 2
    if os.environ.get("FUZZ_HUGE") == "1":              # beyond the stock reference: one side > 2048 (level >= 23)
        w = int(rng.integers(1030, 1300)) * 2
        h = int(rng.integers(100, 850)) * 2
        if rng.integers(0, 3) == 0:
            w, h = h, w
    kind = int(rng.integers(0, 5))
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    def plane():
        if kind == 0:
            a = 128 + 60 * np.sin(x / rng.uniform(5, 40)) * np.cos(y / rng.uniform(5, 40)) + rng.normal(0, rng.uniform(0, 12), (h, w))
        elif kind == 1:
            a = rng.integers(0, 256, (h, w)).astype(np.float64)
        elif kind == 2:
            a = np.full((h, w), float(rng.integers(0, 256))) + rng.normal(0, 1.5, (h, w))
        elif kind == 3:
            a = 255.0 * (((x // rng.integers(2, 40)) + (y // rng.integers(2, 40))) % 2)
        else:
            a = 128 + 100 * np.tanh((x - w / 2) / rng.uniform(3, 60)) + 20 * np.sin(y / 7.0)
        return np.clip(a, 0, 255).astype(np.uint8)
    if colour:
        return synth.ppm_bytes(np.stack([plane(), plane(), plane()], -1))
    return synth.pgm_bytes(plane())


def random_options(rng, lib=None):
    lo = int(rng.integers(4, 9))
    hi = int(rng.integers(max(lo, 6), 13))
    el = int(rng.integers(1, 6))
    lvl = int(rng.integers(0, 3))                       # 2: retries + full_search (cfiasco -z 3)
    dic = int(rng.choice([8, 40, 300, 10000]))
    mant = int(rng.integers(2, 9)); dmant = int(rng.integers(2, 9))        # 6 .. 8: the FC_HM kernel build
    rr = int(rng.integers(0, 4)); dr = int(rng.integers(0, 4))
    cq = float(rng.choice([1.0, 2.0, 3.5])); cd = int(rng.choice([1, 5, 40, 63, 64, 100, 200]))
    pred = (0, 6, 10)
    if os.environ.get("FUZZ_PRED") == "1" and rng.integers(0, 4):     # intra prediction (ND)
        plo = int(rng.integers(6, 11))
        pred = (1, plo, int(rng.integers(plo, 13)))
    if os.environ.get("FUZZ_SPEC") == "1":
        lo, hi, el, lvl, pred = 6, 10, int(rng.integers(1, 4)), 0, (0, 6, 10)
        mant = int(rng.integers(2, 5))
    # initial basis: the built-in one, our own long ones (tests/golden/make_basis.py), the reference's where installed
    share = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "share", "medium.fco")
    names = ["", "", "", "long_a.fco", "long_b.fco", "long_c.fco"] + (["medium.fco", "large.fco"] if os.path.exists(share) else [])
    basis = str(rng.choice(names))
    if os.environ.get("FUZZ_SPEC") == "1":
        basis = ""
    # the other entries of the reference's model registries (fiasco_amd_c_options_set_models; the FC_GM kernel build)
    models = None
    if os.environ.get("FUZZ_SPEC") != "1" and rng.integers(0, 3) == 0:
        pools = ["adaptive", "basis", "uniform", "rle", "rle-no-chroma"]
        models = (str(rng.choice(pools)), str(rng.choice(pools)), str(rng.choice(["adaptive", "uniform"])),
                  str(rng.choice(["adaptive", "uniform"])))
    spec = (lo, hi, el, dic, lvl, mant, rr, dmant, dr, cq, cd, pred, basis, models)
    return spec


def apply(o, spec):
    lo, hi, el, dic, lvl, mant, rr, dmant, dr, cq, cd, pred, basis, models = spec
    if basis:
        o.set_basisfile(basis.encode())
    if models:
        assert o.lib.L.fiasco_amd_c_options_set_models(o.handle, *[m.encode() for m in models])
    o.set_prediction(*pred)
    o.set_optimizations(lo, hi, el, dic, lvl)
    o.set_quantization(mant, rr, dmant, dr)
    o.set_chroma_quality(cq, cd)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    gpu = fiasco_amd.Library(os.environ["FIASCO_AMD_LIB"]) if os.environ.get("FIASCO_AMD_LIB") else fiasco_amd.library()
    ora = fiasco_amd.Library(os.path.join(os.path.dirname(fiasco_amd.LIB_PATH), "..", "oracle", "liboracle_fiasco.so"))
    gpu.set_verbosity(0); ora.set_verbosity(0)
    bad = 0
    refused = 0
    reasons = set()
    n = 0
    t0 = time.time()
    for r in range(rounds):
        rng = np.random.default_rng(seed0 + r)
        spec = random_options(rng, gpu)
        q = float(rng.choice([2.0, 8.0, 20.0, 45.0, 90.0]))
        frames = [random_image(rng, bool(rng.integers(0, 3) == 0)) for _ in range(per)]
        og, oo = gpu.cli_options(), ora.cli_options()
        apply(og, spec); apply(oo, spec)
        if r % 2:
            os.environ["FIASCO_AMD_NO_WIDE"] = "1"
        else:
            os.environ.pop("FIASCO_AMD_NO_WIDE", None)
        if os.environ.get("FUZZ_SPEC") == "1":          # several workgroups per frame, random split
            os.environ.pop("FIASCO_AMD_NO_WIDE", None)
            G = int(rng.integers(2, 9))
            os.environ["FIASCO_AMD_SPEC"] = str(G)
            os.environ["FIASCO_AMD_SPEC_T"] = str(int(rng.integers(0, max(1, G - 1))))
            if rng.integers(0, 3) == 0:
                os.environ["FIASCO_AMD_SPEC_TABWAIT"] = "0"
            else:
                os.environ.pop("FIASCO_AMD_SPEC_TABWAIT", None)
        print("round seed %d spec %s q %s" % (seed0 + r, spec, q), flush=True)
        repl = int(os.environ.get("FUZZ_REPL", "1"))    # FUZZ_REPL=n: every frame n times in the launch
        got_all = gpu.encode_batch(frames * repl, q, og)
        gmsg = gpu.error_message()
        got = got_all[:len(frames)]
        for k in range(1, repl):                         # replicas must agree (timing-dependent faults)
            for i in range(len(frames)):
                if got_all[k * len(frames) + i] != got[i]:
                    bad += 1
                    print("MISMATCH seed %d frame %d: replica %d differs from replica 0 (non-deterministic)"
                          % (seed0 + r, i, k), flush=True)
        exp = ora.encode_batch(frames, q, oo)
        og.delete(); oo.delete()
        for i, (g, e) in enumerate(zip(got, exp)):
            n += 1
            if g is None and e is not None and "device coder" in gmsg:
                if refused == 0 or gmsg not in reasons:
                    reasons.add(gmsg)
                    print("refused:", gmsg, flush=True)
                refused += 1                      # outside the device scope, said so
                continue
            if g != e:
                bad += 1
                print("MISMATCH seed %d frame %d spec %s q %s: device %s oracle %s (%s)"
                      % (seed0 + r, i, spec, q, None if g is None else len(g), None if e is None else len(e),
                         gmsg if g is None else ""), flush=True)
    # a few multi-frame streams through fiasco_coder() (all-I; colour streams carry the ratcheted
    # minimum block level from frame to frame)
    import tempfile
    nseq = int(os.environ.get("FUZZ_NSEQ", max(1, rounds // 5)))
    for r in range(nseq):
        rng = np.random.default_rng(seed0 + 100000 + r)
        spec = random_options(rng)
        q = float(rng.choice([8.0, 20.0, 45.0]))
        colour = bool(rng.integers(0, 2))
        first = random_image(rng, colour)
        hdr = first.split(b"\n")[1].split()
        w, h = int(hdr[0]), int(hdr[1])
        nfr = int(rng.integers(2, 5))
        pattern = "i"
        if os.environ.get("FUZZ_PRED") == "1":          # P and B frames
            pattern = str(rng.choice(["i", "ip", "ipp", "ippp", "ipip", "ibp", "ibbp", "ipbp", "ipb"]))
        with tempfile.TemporaryDirectory() as td:
            names = []
            for f in range(nfr):
                a = rng.integers(0, 256, (h, w, 3 if colour else 1)).astype(np.float64)
                base = np.frombuffer(first[len(first) - w * h * (3 if colour else 1):], np.uint8).reshape(h, w, -1)
                if pattern != "i":                       # a moving scene: motion compensation can win
                    base = np.roll(base, (int(rng.integers(-4, 5)) * f, int(rng.integers(-4, 5)) * f), (0, 1))
                    a = 0.9 * base + 0.1 * a
                img = np.clip(0.85 * base + 0.15 * a + 3 * f, 0, 255).astype(np.uint8)
                pth = os.path.join(td, "f%02d.%s" % (f, "ppm" if colour else "pgm"))
                (synth.write_ppm if colour else synth.write_pgm)(pth, img if colour else img[:, :, 0])
                names.append(pth)
            og, oo = gpu.cli_options(pattern=pattern), ora.cli_options(pattern=pattern)
            apply(og, spec); apply(oo, spec)
            rg = gpu.fiasco_coder(names, os.path.join(td, "g.fco"), q, og)
            gmsg = gpu.error_message()
            ro = ora.fiasco_coder(names, os.path.join(td, "o.fco"), q, oo)
            n += 1
            if rg == 0 and ro == 1 and "device coder" in gmsg:
                refused += 1
            elif rg != ro or (rg == 1 and open(os.path.join(td, "g.fco"), "rb").read() != open(os.path.join(td, "o.fco"), "rb").read()):
                bad += 1
                print("MISMATCH sequence seed %d spec %s q %s colour %s frames %d pattern %s: rc %d/%d (%s)"
                      % (seed0 + 100000 + r, spec, q, colour, nfr, pattern, rg, ro, gmsg), flush=True)
    print("fuzz: %d frames in %d rounds, %d mismatches, %d refused by the device (with message), %.1f s"
          % (n, rounds, bad, refused, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
