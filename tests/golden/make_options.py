#!/usr/bin/env python3
"""Known answers of the REAL reference for the rest of the option space `cfiasco` can reach (round 5):
initial bases other than the built-in one (--basis-name), RPF mantissas of 6 .. 8 bits, chroma dictionaries
of more than 63 states.  Results go into tests/golden/MANIFEST.json "option_cases".

Bases: tests/golden/long_{a,b,c}.fco are our own (make_basis.py); medium.fco / large.fco are the reference's
installed data files (oracle/_ref/share, put there by oracle/ref_build.sh -- git-ignored build output that
travels to the GPU box with the reference binaries).  A case that needs them says "needs": "share".
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402

SHARE = os.path.join(ROOT, "oracle", "_ref", "share")
TMP = "/tmp/fiasco_golden_options"
SEQ = ["m0_128x96", "m1_128x96", "m2_128x96", "m3_128x96"]
# inputs of these cases that tests/golden/make_golden.py does not list (320 x 256 colour, frames 1 and 2 of c00's generator)
make_golden.INPUTS.update({"c01": ("color_c", dict(w=320, h=256, f=1), False), "c02": ("color_c", dict(w=320, h=256, f=2), False)})

CASES = [
    # (name, inputs, args)
    ("b_longa_g256", ["g256"], ["--basis-name", "long_a.fco"]),
    ("b_longa_c256", ["c256"], ["--basis-name", "long_a.fco"]),
    ("b_longa_g160_z1", ["g160x120"], ["--basis-name", "long_a.fco", "-z", "1"]),
    ("b_longa_pred", ["g256"], ["--basis-name", "long_a.fco", "--prediction"]),
    ("b_longb_g256", ["g256"], ["--basis-name", "long_b.fco"]),
    ("b_longb_n128_z2", ["n128x96"], ["--basis-name", "long_b.fco", "-z", "2"]),
    ("b_longc_g256", ["g256"], ["--basis-name", "long_c.fco"]),
    ("b_longc_c256_z1", ["c256"], ["--basis-name", "long_c.fco", "-z", "1"]),
    ("b_longc_seq_ippp", SEQ, ["--basis-name", "long_c.fco"]),
    ("b_longa_seq_ibbp_pred", SEQ, ["--basis-name", "long_a.fco", "--pattern", "ibbp", "--prediction"]),
    ("b_medium_g256", ["g256"], ["--basis-name", "medium.fco"]),
    ("b_large_g256", ["g256"], ["--basis-name", "large.fco"]),
    ("b_medium_c256", ["c256"], ["--basis-name", "medium.fco"]),
    ("b_large_c256_z1", ["c256"], ["--basis-name", "large.fco", "-z", "1"]),
    ("b_medium_g160_z1", ["g160x120"], ["--basis-name", "medium.fco", "-z", "1"]),
    ("b_large_n128_z2", ["n128x96"], ["--basis-name", "large.fco", "-z", "2"]),
    ("b_medium_seq_ippp", SEQ, ["--basis-name", "medium.fco"]),
    ("b_large_g720", ["g720"], ["--basis-name", "large.fco"]),
    # RPF mantissas 6 .. 8 (codec/options.c:510-553; lib/rpf.c:187-198 maps > 8 to 2)
    ("m6_g256", ["g256"], ["--rpf-mantissa", "6"]),
    ("m7_g256", ["g256"], ["--rpf-mantissa", "7"]),
    ("m8_g256", ["g256"], ["--rpf-mantissa", "8"]),
    ("m8_dm8_n128", ["n128x96"], ["--rpf-mantissa", "8", "--dc-rpf-mantissa", "8"]),
    ("dm7_g256", ["g256"], ["--dc-rpf-mantissa", "7"]),
    ("m7_c256", ["c256"], ["--rpf-mantissa", "7", "--dc-rpf-mantissa", "6"]),
    ("m6_g160_z1", ["g160x120"], ["--rpf-mantissa", "6", "-z", "1"]),
    ("m8_g160_z2", ["g160x120"], ["--rpf-mantissa", "8", "--dc-rpf-mantissa", "8", "-z", "2"]),
    ("m6_pred_g256", ["g256"], ["--rpf-mantissa", "6", "--prediction"]),
    ("m8_seq_ippp", SEQ, ["--rpf-mantissa", "8", "--dc-rpf-mantissa", "7"]),
    ("m8_g720", ["g720"], ["--rpf-mantissa", "8"]),
    # chroma dictionaries of more than 63 states (codec/options.c:296-332, codec/domain-pool.c:854-879)
    ("cd64_c00", ["c00"], ["--chroma-dictionary", "64"]),
    ("cd100_c00", ["c00"], ["--chroma-dictionary", "100"]),
    ("cd100_c256", ["c256"], ["--chroma-dictionary", "100"]),
    ("cd150_c00_z1", ["c00"], ["--chroma-dictionary", "150", "-z", "1"]),
    ("cd200_k720", ["k720"], ["--chroma-dictionary", "200"]),
    ("cd1000_k720", ["k720"], ["--chroma-dictionary", "1000", "--chroma-qfactor", "1.0"]),
    ("cd100_seq2_color_i", ["c00", "c01"], ["--chroma-dictionary", "100", "--pattern", "i"]),
    ("cd80_seq3_color_ipp", ["c00", "c01", "c02"], ["--chroma-dictionary", "80"]),
    # all three at once
    ("b_medium_m7_cd100_c256", ["c256"], ["--basis-name", "medium.fco", "--rpf-mantissa", "7", "--chroma-dictionary", "100"]),
]


def main():
    os.makedirs(TMP, exist_ok=True)
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    env = dict(os.environ, FIASCO_DATA=HERE + ":" + SHARE)
    paths = {}
    out = []
    for name, ins, args in CASES:
        for i in ins:
            if i not in paths:
                data, ext = make_golden.make_input(i)
                paths[i] = os.path.join(TMP, i + "." + ext)
                open(paths[i], "wb").write(data)
                if i not in man["inputs"]:
                    kind, a, _ = make_golden.INPUTS[i]
                    man["inputs"][i] = {"kind": kind, "args": a, "md5": hashlib.md5(data).hexdigest(), "bytes": len(data),
                                        "ext": ext, "file": None}
        fco = os.path.join(TMP, name + ".fco")
        r = subprocess.run([make_golden.REF, "--progress-meter", "0"] + args + ["-o", fco] + [paths[i] for i in ins],
                           env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        if r.returncode < 0 or r.returncode >= 128:
            sys.exit("the reference crashed on %s" % name)
        ent = {"name": name, "inputs": ins, "args": args, "file": None}
        if r.returncode != 0:       # e.g. "Can't write more than N weights": the oracle and the device must fail too
            ent["fails"] = True
            ent["message"] = r.stderr.decode("latin-1").strip().split("\n")[-1]
            data = b""
        else:
            data = open(fco, "rb").read()
            ent["md5"] = hashlib.md5(data).hexdigest(); ent["bytes"] = len(data)
        if any(a in ("medium.fco", "large.fco") for a in args):
            ent["needs"] = "share"
        out.append(ent)
        print("%-26s %7d B  %s" % (name, len(data), ent.get("md5", "FAILS: " + ent.get("message", ""))))
    man["option_cases"] = out
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
