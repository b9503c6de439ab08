#!/usr/bin/env python3
"""Known answers for `cfiasco -z 3' (check_for_underflow / check_for_overflow / full_search,
codec/approx.c:119-206,420) from the ZERO-INITIALISED variant of the real reference (round 6).

The stock reference reads members of an automatic `mp_t' that nobody has written under -z 3 (codec/approx.c:88,
439-446): what it writes depends on what its stack held.  oracle/ref_build.sh builds a declared variant,
cfiasco_ref_z3 -- one sed line on a throw-away copy of codec/approx.c that zero-initialises that struct, all other
objects stock -- which defines those reads the way the oracle and the device define them.  This script
  * checks that the variant writes what the STOCK reference writes without -z 3 (the patch changes nothing else),
  * records the variant's -z 3 streams as tests/golden/MANIFEST.json "z3_cases",
  * and reports where the stock reference's -z 3 stream happens to equal the variant's ("stock_equal").
Inputs that make_golden.py does not commit as files are re-synthesised by the tests (md5 checked).
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
import make_options  # noqa: E402,F401  (adds c01 / c02 to make_golden.INPUTS)

REF_Z3 = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref_z3")
SHARE = os.path.join(ROOT, "oracle", "_ref", "share")
TMP = "/tmp/fiasco_golden_z3"
SEQ = ["m0_128x96", "m1_128x96", "m2_128x96", "m3_128x96"]

CASES = [
    # (name, inputs, args) -- every case runs with -z 3
    ("z3_g256_q20", ["g256"], []),
    ("z3_g256_q5", ["g256"], ["-q", "5"]),
    ("z3_g256_q60", ["g256"], ["-q", "60"]),
    ("z3_g160x120", ["g160x120"], []),
    ("z3_g100x70_q45", ["g100x70"], ["-q", "45"]),
    ("z3_n128x96", ["n128x96"], []),
    ("z3_n128x96_q60", ["n128x96"], ["-q", "60"]),
    ("z3_check64", ["check64"], []),
    ("z3_n512", ["n512"], []),
    ("z3_c256", ["c256"], []),
    ("z3_c00_q45", ["c00"], ["-q", "45"]),
    ("z3_c256_chroma", ["c256"], ["--chroma-qfactor", "1.0", "--chroma-dictionary", "100"]),
    ("z3_pred_g256", ["g256"], ["--prediction"]),
    ("z3_pred_n128x96_lv79", ["n128x96"], ["--prediction", "--min-level", "7", "--max-level", "9"]),
    ("z3_pred_c256", ["c256"], ["--prediction"]),
    ("z3_seq_ippp", SEQ, []),
    ("z3_seq_ibbp_pred", SEQ, ["--pattern", "ibbp", "--prediction"]),
    ("z3_m7_g256", ["g256"], ["--rpf-mantissa", "7", "--dc-rpf-mantissa", "6"]),
    ("z3_rpf2_g160", ["g160x120"], ["--rpf-mantissa", "2", "--rpf-range", "0.75", "--dc-rpf-mantissa", "2"]),
    ("z3_longa_g256", ["g256"], ["--basis-name", "long_a.fco"]),
]


def run(exe, args, files, out, env):
    r = subprocess.run([exe, "--progress-meter", "0"] + args + ["-o", out] + files, env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    if r.returncode < 0 or r.returncode >= 128:
        return "crash", None
    if r.returncode != 0:
        return r.stderr.decode("latin-1").strip().split("\n")[-1], None
    return None, open(out, "rb").read()


def main():
    os.makedirs(TMP, exist_ok=True)
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    env = dict(os.environ, FIASCO_DATA=HERE + ":" + SHARE)
    paths, out = {}, []
    for name, ins, args in CASES:
        for i in ins:
            if i not in paths:
                data, ext = make_golden.make_input(i)
                paths[i] = os.path.join(TMP, i + "." + ext)
                open(paths[i], "wb").write(data)
        files = [paths[i] for i in ins]
        # the patch must not change anything the stock reference defines: the same case without -z 3
        for z in ([], ["-z", "1"], ["-z", "2"]):
            e0, d0 = run(make_golden.REF, args + z, files, os.path.join(TMP, "s.fco"), env)
            e1, d1 = run(REF_Z3, args + z, files, os.path.join(TMP, "v.fco"), env)
            if (e0, d0) != (e1, d1):
                sys.exit("%s %s: the variant differs from the stock reference WITHOUT -z 3" % (name, z))
        err, data = run(REF_Z3, args + ["-z", "3"], files, os.path.join(TMP, name + ".fco"), env)
        if err == "crash":
            sys.exit("the variant crashed on %s" % name)
        es, ds = run(make_golden.REF, args + ["-z", "3"], files, os.path.join(TMP, "s3.fco"), env)
        ent = {"name": name, "inputs": ins, "args": args + ["-z", "3"], "file": None,
               "stock_equal": es is None and ds == data}
        if err:
            ent["fails"] = True; ent["message"] = err; data = b""
        else:
            ent["md5"] = hashlib.md5(data).hexdigest(); ent["bytes"] = len(data)
        out.append(ent)
        print("%-24s %7d B  %s  stock reference %s" % (name, len(data), ent.get("md5", "FAILS: " + ent.get("message", "")),
                                                       "equal" if ent["stock_equal"] else ("differs" if es is None else es)))
    man["z3_cases"] = out
    man["z3_reference"] = "oracle/_ref/cfiasco_ref_z3: codec/approx.c:88 `mp_t mp' zero-initialised (oracle/ref_build.sh), all other objects stock"
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
