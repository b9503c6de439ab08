#!/usr/bin/env python3
"""Golden answers of the LIMITS-EXTENSION reference (oracle/_ref/cfiasco_ref_big, built by
oracle/ref_build.sh from a patched throw-away copy of the reference: MAXSTATES 30000, MAXLEVEL 26,
SURVEY.md 8c) for what the stock reference cannot encode: images above 2048 pixels and 1080p colour
at the CLI defaults.  Build container only.  Inputs are synthesised (tests/synth.py) and verified
by md5 before use; only md5 + size of the streams are recorded: tests/golden/MANIFEST_BIG.json.
The cases run in parallel (a 4K frame takes the reference three minutes)."""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref_big")
STOCK = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref")
TMP = "/tmp/fiasco_golden_big"

INPUTS = {
    "g2160":       ("synth", dict(w=3840, h=2160, seed=1234)),     # b1216151..., 32 567 B
    "g2160_s1000": ("synth", dict(w=3840, h=2160, seed=1000)),
    "g2160_s1001": ("synth", dict(w=3840, h=2160, seed=1001)),
    "g2304":       ("synth", dict(w=2304, h=1296, seed=77)),
    "g2100x900":   ("synth", dict(w=2100, h=900, seed=78)),        # ragged, level 23
    "c1080":       ("color_c", dict(w=1920, h=1080, f=0)),         # b5bfe613..., 20 687 B
    "k1080":       ("color_k", dict(w=1920, h=1080)),              # ca81b603..., 10 892 B
    "g256":        ("synth", dict(w=256, h=256, seed=1234)),
}
CASES = [
    ("big_g2160", ["g2160"], []),
    ("big_g2160_s1000", ["g2160_s1000"], []),
    ("big_seq2_4k", ["g2160_s1000", "g2160_s1001"], ["--pattern", "i"]),
    ("big_g2304", ["g2304"], []),
    ("big_g2100x900", ["g2100x900"], []),
    ("big_c1080", ["c1080"], []),
    ("big_k1080", ["k1080"], []),
    # with the stock dictionary size the patched build must give the stock build's stream
    ("big_g256_dict6000", ["g256"], ["--dictionary-size", "6000"]),
]
SURVEY = {"big_g2160": ("b121615161f96c0b541f29ef7228380e", 32567),
          "big_c1080": ("b5bfe6132db7a198175d3a70653e9261", 20687),
          "big_k1080": ("ca81b603e331985870426d8257eaacbd", 10892)}


def make_input(name):
    kind, a = INPUTS[name]
    if kind == "synth":
        return synth.pgm_bytes(synth.synth(a["w"], a["h"], a["seed"])), "pgm"
    if kind == "color_c":
        return synth.ppm_bytes(synth.synth_color_c(a["w"], a["h"], a["f"])), "ppm"
    if kind == "color_k":
        return synth.ppm_bytes(synth.synth_color_k(a["w"], a["h"], 1234)), "ppm"
    raise ValueError(kind)


def main():
    if not os.path.exists(REF):
        sys.exit("limits-extension reference missing: run oracle/ref_build.sh first")
    os.makedirs(TMP, exist_ok=True)
    man = {"generator": "tests/golden/make_golden_big.py",
           "reference": "l-tamas/Fiasco (FIASCO 1.3), limits extension: MAXSTATES 30000, MAXLEVEL 26, "
                        "init_tree_model entry 21 for levels >= 22 (oracle/ref_build.sh), gcc -O2 -fcommon",
           "reference_build": dict(l.strip().split(": ", 1) for l in
                                   open(os.path.join(ROOT, "oracle", "_ref", "BUILD_INFO")) if ": " in l),
           "inputs": {}, "cases": []}
    paths = {}
    for name, (kind, a) in INPUTS.items():
        data, ext = make_input(name)
        p = os.path.join(TMP, name + "." + ext)
        open(p, "wb").write(data)
        paths[name] = p
        man["inputs"][name] = {"kind": kind, "args": a, "md5": hashlib.md5(data).hexdigest(), "bytes": len(data), "ext": ext}
    env = dict(os.environ, FIASCO_DATA="/root/reference/data")

    def run(case):
        cname, ins, args = case
        out = os.path.join(TMP, cname + ".fco")
        cmd = [REF, "--progress-meter", "0"] + args + ["-o", out] + [paths[i] for i in ins]
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError("reference failed on %s: %s" % (cname, r.stderr.decode()))
        data = open(out, "rb").read()
        return {"name": cname, "inputs": ins, "args": args, "md5": hashlib.md5(data).hexdigest(), "bytes": len(data)}

    with ThreadPoolExecutor(max_workers=min(len(CASES), len(os.sched_getaffinity(0)))) as ex:
        for ent in ex.map(run, CASES):
            if ent["name"] in SURVEY:
                ent["survey_md5"] = SURVEY[ent["name"]][0]
                assert (ent["md5"], ent["bytes"]) == SURVEY[ent["name"]], (ent, "differs from SURVEY App. C")
            man["cases"].append(ent)
            print("%-20s %7d B  %s" % (ent["name"], ent["bytes"], ent["md5"]), flush=True)
    # patched == stock where the stock build can run
    out = os.path.join(TMP, "stock_g256.fco")
    subprocess.check_call([STOCK, "--progress-meter", "0", "-o", out, paths["g256"]], env=env)
    stock = open(out, "rb").read()
    d6000 = [c for c in man["cases"] if c["name"] == "big_g256_dict6000"][0]
    assert hashlib.md5(stock).hexdigest() == d6000["md5"], "patched build with --dictionary-size 6000 != stock build"
    json.dump(man, open(os.path.join(HERE, "MANIFEST_BIG.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
