#!/usr/bin/env python3
"""Decoded-PSNR known answers of the REAL reference: for the gray single-frame cases
below, `dfiasco_ref -s 0` decodes the reference's own stream (no smoothing: the frame the coder itself
would use as a reference) and `pnmpsnr_ref` compares it with the input.  The printed figure ("%.2f dB") goes
into tests/golden/MANIFEST.json under "decoded_psnr"; the product's fiasco_amd_batch_decode_psnr() must
print the same two decimals.  Build container only (oracle/_ref from oracle/ref_build.sh)."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402

CASES = ["g256_q20", "g256_q5", "g256_q60", "g96x64_q20", "n128x96_q20", "g100x70_q20", "n512_q20", "g720_q20", "g1080_q20"]
REFDIR = os.path.join(ROOT, "oracle", "_ref")
TMP = "/tmp/fiasco_golden_psnr"


def main():
    os.makedirs(TMP, exist_ok=True)
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    cases = {c["name"]: c for c in man["cases"]}
    env = dict(os.environ, FIASCO_DATA="/root/reference/data")
    out = {}
    for name in CASES:
        c = cases[name]
        assert len(c["inputs"]) == 1
        data, ext = make_golden.make_input(c["inputs"][0])
        src = os.path.join(TMP, name + "." + ext)
        open(src, "wb").write(data)
        fco = os.path.join(TMP, name + ".fco")
        subprocess.check_call([os.path.join(REFDIR, "cfiasco_ref"), "--progress-meter", "0"] + c["args"] + ["-o", fco, src],
                              env=env, stderr=subprocess.DEVNULL)
        dec = os.path.join(TMP, name + ".dec.pgm")
        subprocess.check_call([os.path.join(REFDIR, "dfiasco_ref"), "-s", "0", "-o", dec, fco], env=env, stderr=subprocess.DEVNULL)
        r = subprocess.run([os.path.join(REFDIR, "pnmpsnr_ref"), src, dec], env=env, stderr=subprocess.PIPE, text=True)
        m = re.search(r"([0-9.]+) dB", r.stderr)
        assert m, r.stderr
        raw = open(dec, "rb").read()
        w, h = [int(v) for v in raw.split(b"\n", 2)[1].split()]
        import hashlib
        out[name] = {"psnr_db": m.group(1), "decoded_md5": hashlib.md5(raw[len(raw) - w * h:]).hexdigest(),
                     "tools": "dfiasco_ref -s 0 -o dec.pgm ref.fco; pnmpsnr_ref in.pgm dec.pgm; decoded_md5 = md5 of dec.pgm's pixel bytes"}
        print("%-14s %s dB" % (name, m.group(1)))
    man["decoded_psnr"] = out
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
