#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/ with the REAL reference coder.

Runs only in the build container (needs oracle/_ref/cfiasco_ref built by
oracle/ref_build.sh from /root/reference).  For every case it synthesises the input
(tests/synth.py), runs the reference CLI and records:
    small cases : input PNM + reference .fco committed as files
    large cases : md5 + size of input and of the reference .fco (inputs are re-synthesised
                  by the tests and verified against the recorded md5 before use)
Output: tests/golden/MANIFEST.json + tests/golden/*.pgm|ppm|fco
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref")
TMP = "/tmp/fiasco_golden"

INPUTS = {
    # name: (kind, args, commit_file)
    "g96x64":   ("synth", dict(w=96, h=64, seed=5), True),
    "g64x32":   ("synth", dict(w=64, h=32, seed=6), True),
    "g32x32":   ("synth", dict(w=32, h=32, seed=8), True),
    "g160x120": ("synth", dict(w=160, h=120, seed=7), True),
    "g256":     ("synth", dict(w=256, h=256, seed=1234), True),
    "n128x96":  ("noise", dict(w=128, h=96, seed=7), True),
    "f0_96x64": ("synth", dict(w=96, h=64, seed=21), True),
    "f1_96x64": ("synth", dict(w=96, h=64, seed=22), True),
    "n512":     ("noise", dict(w=512, h=384, seed=7), False),
    "g720":     ("synth", dict(w=1280, h=720, seed=1234), False),
    "g1080":    ("synth", dict(w=1920, h=1080, seed=1234), False),
    "c00":      ("color_c", dict(w=320, h=256, f=0), False),
    "c256":     ("color_c", dict(w=256, h=192, f=0), False),   # (128 x 128 colour: the reference
    "c256b":    ("color_c", dict(w=256, h=192, f=1), False),   #  fails, "Can't write more than 121 weights")
    "g100x70":  ("synth", dict(w=100, h=70, seed=9), True),
    "flat64":   ("flat", dict(w=64, h=64, v=200), True),
    "check64":  ("checker", dict(w=64, h=64, cell=4), True),
    "ramp96":   ("ramp", dict(w=96, h=32), True),
    "k720":     ("color_k", dict(w=1280, h=720), False),     # SURVEY App. C smooth-chroma generator
    # colour frames (48 x 70 and 74 x 138) found by tests/fuzz_oracle_vs_reference.py (seeds 1431,
    # 1187): coded as one stream, y_column flags of a frame show through on the next one
    "carry48_a": ("file", dict(ext="ppm"), True), "carry48_b": ("file", dict(ext="ppm"), True),
    "carry48_c": ("file", dict(ext="ppm"), True),
    "carry74_a": ("file", dict(ext="ppm"), True), "carry74_b": ("file", dict(ext="ppm"), True),
    "carry74_c": ("file", dict(ext="ppm"), True),
    # video: four frames of a scene that moves 3 px per frame (SURVEY App. C generator with shift)
    "m0_128x96": ("synth", dict(w=128, h=96, seed=31, shift=0), True),
    "m1_128x96": ("synth", dict(w=128, h=96, seed=31, shift=3), True),
    "m2_128x96": ("synth", dict(w=128, h=96, seed=31, shift=6), True),
    "m3_128x96": ("synth", dict(w=128, h=96, seed=31, shift=9), True),
    "c256c":    ("color_c", dict(w=256, h=192, f=2), False),
    # v00/v01/v02.ppm, 1280x720 smooth-chroma colour, x shifted by 3 f
    "v00":      ("color_k", dict(w=1280, h=720, shift=0), False),
    "v01":      ("color_k", dict(w=1280, h=720, shift=3), False),
    "v02":      ("color_k", dict(w=1280, h=720, shift=6), False),
    "k1080":    ("color_k", dict(w=1920, h=1080), False),
}

CASES = [
    # (case name, [input names], extra CLI args)
    ("g32x32_q20", ["g32x32"], []),
    ("g64x32_q20", ["g64x32"], []),
    ("g96x64_q20", ["g96x64"], []),
    ("g160x120_q20", ["g160x120"], []),
    ("g256_q20", ["g256"], []),
    ("g256_q5", ["g256"], ["-q", "5"]),
    ("g256_q60", ["g256"], ["-q", "60"]),
    ("g256_z1", ["g256"], ["-z", "1"]),
    ("g256_z2", ["g256"], ["-z", "2"]),
    ("g256_dict64", ["g256"], ["--dictionary-size", "64"]),
    ("g256_rpf", ["g256"], ["--rpf-mantissa", "4", "--rpf-range", "2.0", "--dc-rpf-mantissa", "4"]),
    ("g256_title", ["g256"], ["-t", "a title", "-c", "a comment"]),
    ("n128x96_q20", ["n128x96"], []),
    ("n128x96_q60", ["n128x96"], ["-q", "60"]),
    ("n128x96_z1", ["n128x96"], ["-z", "1"]),
    ("n128x96_z2", ["n128x96"], ["-z", "2"]),
    ("seq2_gray_i", ["f0_96x64", "f1_96x64"], ["--pattern", "i"]),
    ("n512_q20", ["n512"], []),
    ("n512_z1", ["n512"], ["-z", "1"]),
    ("n512_z2", ["n512"], ["-z", "2"]),
    ("n512_q5_z1", ["n512"], ["-q", "5", "-z", "1"]),
    ("g720_q20", ["g720"], []),
    ("g1080_q20", ["g1080"], []),
    ("c00_q20", ["c00"], []),
    ("c00_z1", ["c00"], ["-z", "1"]),
    # colour with committed inputs, chroma options, the minimum-level carry between the frames
    # of a colour stream (codec/coder.c:785-797), options the CLI passes through unchanged
    ("c256_q20", ["c256"], []),
    ("c256_z2", ["c256"], ["-z", "2"]),
    ("c256_chroma", ["c256"], ["--chroma-qfactor", "3.5", "--chroma-dictionary", "5"]),
    ("seq2_color_i", ["c256", "c256b"], ["--pattern", "i"]),
    ("seq2_color_i_rev", ["c256b", "c256"], ["--pattern", "i"]),
    ("g256_dict8", ["g256"], ["--dictionary-size", "8"]),
    ("g256_tiling", ["g256"], ["--tiling-exponent", "2", "--tiling-method", "asc-variance"]),
    ("g256_ranges", ["g256"], ["--rpf-range", "1.0", "--dc-rpf-range", "2.0"]),
    ("g96x64_q90", ["g96x64"], ["-q", "90"]),
    ("g100x70_q20", ["g100x70"], []),
    # degenerate images: constant, full-swing checkerboard, horizontal ramp
    ("flat64_q20", ["flat64"], []),
    ("check64_q20", ["check64"], []),
    ("check64_z2", ["check64"], ["-z", "2"]),
    ("ramp96_q20", ["ramp96"], []),
    ("g256_q1", ["g256"], ["-q", "1"]),
    ("g256_q99", ["g256"], ["-q", "99"]),
    ("c256_z1", ["c256"], ["-z", "1"]),
    ("seq3_gray_i", ["f0_96x64", "f1_96x64", "g96x64"], ["--pattern", "i"]),
    ("k720_q20", ["k720"], []),
    ("seq3_color_carry48", ["carry48_a", "carry48_b", "carry48_c"],
     ["-q", "99", "-z", "2", "--dictionary-size", "300", "--rpf-mantissa", "5", "--dc-rpf-mantissa", "5",
      "--rpf-range", "0.75", "--dc-rpf-range", "2.0", "--chroma-qfactor", "3.5", "--chroma-dictionary", "1",
      "--pattern", "i"]),
    ("seq3_color_carry74", ["carry74_a", "carry74_b", "carry74_c"],
     ["--dictionary-size", "8", "--rpf-mantissa", "4", "--dc-rpf-mantissa", "2", "--rpf-range", "2.0",
      "--dc-rpf-range", "1.5", "--chroma-qfactor", "3.5", "--chroma-dictionary", "1", "--pattern", "i"]),
]

# Further pins of the ORACLE against the real reference (small inputs, option sweeps).  The GPU
# suite does not list them one by one: the device is tied to the oracle on these option sets by
# the differential fuzzer (tests/fuzz_parity.py).
ORACLE_CASES = [
    ("o_g160_z1", ["g160x120"], ["-z", "1"]),
    ("o_g160_z2", ["g160x120"], ["-z", "2"]),
    ("o_g100_z2", ["g100x70"], ["-z", "2"]),
    ("o_g256_m2", ["g256"], ["--rpf-mantissa", "2"]),
    ("o_g256_m5", ["g256"], ["--rpf-mantissa", "5"]),
    ("o_g256_dm3", ["g256"], ["--dc-rpf-mantissa", "3"]),
    ("o_g256_r075", ["g256"], ["--rpf-range", "0.75", "--dc-rpf-range", "0.75"]),
    ("o_g256_dict1", ["g256"], ["--dictionary-size", "1"]),
    ("o_g256_dict40_z2", ["g256"], ["--dictionary-size", "40", "-z", "2"]),
    ("o_n128_q2", ["n128x96"], ["-q", "2"]),
    ("o_n128_q45_z1", ["n128x96"], ["-q", "45", "-z", "1"]),
    ("o_g64x32_z2", ["g64x32"], ["-z", "2"]),
    ("o_g32x32_q60_z2", ["g32x32"], ["-q", "60", "-z", "2"]),
    ("o_ramp96_z1", ["ramp96"], ["-z", "1"]),
    # ("flat64", -z 2): the reference itself dies with a segmentation fault on the constant image
    ("o_c256_q45_chroma", ["c256"], ["-q", "45", "--chroma-qfactor", "1.0", "--chroma-dictionary", "63"]),
]

# intra prediction (codec/prediction.c), P and B frames (codec/mwfa.c,
# codec/motion.c, the decoder that rebuilds the reference frames) -- streams of the real reference.
VIDEO_CASES = [
    ("pred_g256", ["g256"], ["--prediction"]),
    ("pred_n128x96", ["n128x96"], ["--prediction"]),
    ("pred_g96x64_q60", ["g96x64"], ["--prediction", "-q", "60"]),
    ("pred_g256_z1", ["g256"], ["--prediction", "-z", "1"]),
    ("pred_g256_lv79", ["g256"], ["--prediction", "--min-level", "7", "--max-level", "9"]),
    ("pred_c256", ["c256"], ["--prediction"]),
    ("seq2_gray_ip", ["f0_96x64", "f1_96x64"], ["--pattern", "ip"]),
    ("seq2_gray_ip_pred", ["f0_96x64", "f1_96x64"], ["--pattern", "ip", "--prediction"]),
    # cfiasco parses --half-pixel and never passes it on (bin/cwfa.c): same stream as above
    ("seq2_gray_ip_halfpel", ["f0_96x64", "f1_96x64"], ["--pattern", "ip", "--prediction", "--half-pixel"]),
    ("seq4_gray_ippp", ["m0_128x96", "m1_128x96", "m2_128x96", "m3_128x96"], []),
    ("seq4_gray_ibbp", ["m0_128x96", "m1_128x96", "m2_128x96", "m3_128x96"], ["--pattern", "ibbp"]),
    ("seq4_gray_ibbp_pred_q60", ["m0_128x96", "m1_128x96", "m2_128x96", "m3_128x96"],
     ["--pattern", "ibbp", "--prediction", "-q", "60"]),
    ("seq4_gray_ipbb", ["m0_128x96", "m1_128x96", "m2_128x96", "m3_128x96"], ["--pattern", "ipbb"]),
    ("seq3_color_ipp", ["c256", "c256b", "c256c"], []),
    ("seq3_color_ibp", ["c256", "c256b", "c256c"], ["--pattern", "ibp", "--prediction"]),
    # SURVEY App. C known answer: 15 774 B, md5 2528889c0453c4590289ad07d9fb87e3
    ("v720_ipp_pred", ["v00", "v01", "v02"], ["--prediction"]),
]

# BASELINE config 3 on the stock reference: 1920x1080 colour at -z 1 (5639 B,
# md5 c678f8ef9e9613ba40f49d93882b8491); lands in "cases"
CASES.append(("k1080_z1", ["k1080"], ["-z", "1"]))


def make_input(name):
    kind, a, _ = INPUTS[name]
    if kind == "synth":
        return synth.pgm_bytes(synth.synth(a["w"], a["h"], a["seed"], a.get("shift", 0))), "pgm"
    if kind == "noise":
        return synth.pgm_bytes(synth.noise(a["w"], a["h"], a["seed"])), "pgm"
    if kind == "file":
        return open(os.path.join(HERE, name + "." + a["ext"]), "rb").read(), a["ext"]
    if kind == "flat":
        import numpy as np
        return synth.pgm_bytes(np.full((a["h"], a["w"]), a["v"], np.uint8)), "pgm"
    if kind == "checker":
        import numpy as np
        y, x = np.mgrid[0:a["h"], 0:a["w"]]
        return synth.pgm_bytes((255 * (((x // a["cell"]) + (y // a["cell"])) % 2)).astype(np.uint8)), "pgm"
    if kind == "ramp":
        import numpy as np
        y, x = np.mgrid[0:a["h"], 0:a["w"]]
        return synth.pgm_bytes((x * 255 // (a["w"] - 1)).astype(np.uint8)), "pgm"
    if kind == "color_k":
        return synth.ppm_bytes(synth.synth_color_k(a["w"], a["h"], 1234, a.get("shift", 0))), "ppm"
    if kind == "color_c":
        return synth.ppm_bytes(synth.synth_color_c(a["w"], a["h"], a["f"])), "ppm"
    raise ValueError(kind)


# Streams of the real reference that take too long to regenerate with every run of this script:
# produced once with the same binary (oracle/_ref/cfiasco_ref), command and generator recorded.
LONG_KNOWN_ANSWERS = {
    "config5_300": {
        "frames": 300,
        "generator": "synth.synth_color_k(1280, 720, 1234, 3 * f), f = 0..299 (tests/gpu_config5.py)",
        "command": "cfiasco_ref --progress-meter 0 --prediction -o out.fco 'v[000-299].ppm'  "
                   "(pattern ippppppppp, -q 20, block levels 6..10)",
        "bytes": 1886498, "md5": "714a25c639d8daae598f84095f4856f7",
        "produced_by": "oracle/_ref/cfiasco_ref, 68 min on one core; oracle/cfiasco_oracle gives the same stream",
    },
}


def main():
    if not os.path.exists(REF):
        sys.exit("reference CLI missing: run oracle/ref_build.sh first")
    os.makedirs(TMP, exist_ok=True)
    man = {"generator": "tests/golden/make_golden.py",
           "reference": "l-tamas/Fiasco (FIASCO 1.3) built by oracle/ref_build.sh: gcc -O2 -fcommon",
           # what the reference was built with when these streams were produced (oracle/_ref/BUILD_INFO)
           "reference_build": dict(l.strip().split(": ", 1) for l in
                                   open(os.path.join(ROOT, "oracle", "_ref", "BUILD_INFO")) if ": " in l),
           "inputs": {}, "cases": []}
    paths = {}
    for name, (kind, a, commit) in INPUTS.items():
        data, ext = make_input(name)
        p = os.path.join(TMP, name + "." + ext)
        open(p, "wb").write(data)
        paths[name] = p
        ent = {"kind": kind, "args": a, "md5": hashlib.md5(data).hexdigest(), "bytes": len(data),
               "ext": ext, "file": None}
        if commit:
            ent["file"] = name + "." + ext
            open(os.path.join(HERE, ent["file"]), "wb").write(data)
        man["inputs"][name] = ent
    env = dict(os.environ, FIASCO_DATA="/root/reference/data")
    for cname, ins, args in CASES:
        out = os.path.join(TMP, cname + ".fco")
        cmd = [REF, "--progress-meter", "0"] + args + ["-o", out] + [paths[i] for i in ins]
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
        if r.returncode != 0:
            sys.exit("reference failed on %s: %s" % (cname, r.stderr.decode()))
        data = open(out, "rb").read()
        ent = {"name": cname, "inputs": ins, "args": args, "md5": hashlib.md5(data).hexdigest(),
               "bytes": len(data), "file": None}
        if all(INPUTS[i][2] for i in ins):
            ent["file"] = cname + ".fco"
            open(os.path.join(HERE, ent["file"]), "wb").write(data)
        man["cases"].append(ent)
        print("%-16s %6d B  %s" % (cname, len(data), ent["md5"]))
    man["oracle_cases"] = []
    for cname, ins, args in ORACLE_CASES:
        out = os.path.join(TMP, cname + ".fco")
        cmd = [REF, "--progress-meter", "0"] + args + ["-o", out] + [paths[i] for i in ins]
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
        if r.returncode != 0:
            sys.exit("reference failed on %s: %s" % (cname, r.stderr.decode()))
        data = open(out, "rb").read()
        man["oracle_cases"].append({"name": cname, "inputs": ins, "args": args, "bytes": len(data),
                                    "md5": hashlib.md5(data).hexdigest(), "file": None})
        print("%-28s %6d B  %s" % (cname, len(data), man["oracle_cases"][-1]["md5"]))
    man["video_cases"] = []
    for cname, ins, args in VIDEO_CASES:
        out = os.path.join(TMP, cname + ".fco")
        cmd = [REF, "--progress-meter", "0"] + args + ["-o", out] + [paths[i] for i in ins]
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
        if r.returncode != 0:
            sys.exit("reference failed on %s: %s" % (cname, r.stderr.decode()))
        data = open(out, "rb").read()
        ent = {"name": cname, "inputs": ins, "args": args, "bytes": len(data),
               "md5": hashlib.md5(data).hexdigest(), "file": None}
        if all(INPUTS[i][2] for i in ins):
            ent["file"] = cname + ".fco"
            open(os.path.join(HERE, ent["file"]), "wb").write(data)
        man["video_cases"].append(ent)
        print("%-28s %6d B  %s" % (cname, len(data), ent["md5"]))
    man["long_known_answers"] = LONG_KNOWN_ANSWERS
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
