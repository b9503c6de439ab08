#!/usr/bin/env python3
"""Known answers for the entries of the reference's model registries that fiasco.h cannot select
(codec/domain-pool.c:188-236: adaptive, basis, uniform, rle-no-chroma; codec/coeff.c:97-131: uniform).
The reference is run as the variants oracle/ref_build.sh builds with other defaults in
codec/options.c:77-80 (cfiasco_ref_<variant>); results go into tests/golden/MANIFEST.json "model_cases".
A case on which the reference itself fails is recorded as such (the oracle must fail too)."""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402

VARIANTS = {"adaptive": ("adaptive", "adaptive", "adaptive", "adaptive"), "uniform": ("uniform", "uniform", "uniform", "uniform"),
            "basis": ("basis", "rle", "adaptive", "uniform"), "nochroma": ("rle-no-chroma", "rle", "adaptive", "adaptive"),
            "rleuni": ("rle", "adaptive", "uniform", "adaptive")}
RUNS = [(["g160x120"], []), (["n128x96"], []), (["c256"], []), (["g160x120"], ["-z", "1"]), (["c256"], ["-z", "2"]),
        (["g160x120"], ["--prediction"]), (["f0_96x64", "f1_96x64"], ["--prediction"]), (["n128x96"], ["-q", "60"]),
        (["c256"], ["--chroma-dictionary", "5"]), (["g256"], [])]
TMP = "/tmp/fiasco_golden_models"


def main():
    os.makedirs(TMP, exist_ok=True)
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    env = dict(os.environ, FIASCO_DATA="/root/reference/data")
    paths = {}
    for ins, _ in RUNS:
        for i in ins:
            if i not in paths:
                data, ext = make_golden.make_input(i)
                paths[i] = os.path.join(TMP, i + "." + ext)
                open(paths[i], "wb").write(data)
    out = []
    for v, models in VARIANTS.items():
        exe = os.path.join(ROOT, "oracle", "_ref", "cfiasco_ref_" + v)
        for k, (ins, args) in enumerate(RUNS):
            fco = os.path.join(TMP, "m.fco")
            r = subprocess.run([exe, "--progress-meter", "0"] + args + ["-o", fco] + [paths[i] for i in ins],
                               env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            ent = {"name": "model_%s_%d" % (v, k), "inputs": ins, "args": args, "models": list(models), "file": None}
            if r.returncode < 0 or r.returncode >= 128:
                continue                                   # the reference crashed: nothing to pin
            if r.returncode != 0:
                ent["fails"] = True
            else:
                data = open(fco, "rb").read()
                ent["md5"] = hashlib.md5(data).hexdigest(); ent["bytes"] = len(data)
            out.append(ent)
            print("%-18s %-28s %s" % (ent["name"], " ".join(ins + args), ent.get("md5", "FAILS")))
    man["model_cases"] = out
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
