"""Developer helper: a golden case through the FC_HM / FC_GM kernel builds against the CPU oracle.
usage: gpu_gm_probe.py"""
import os, sys, json, hashlib, tempfile
os.environ["FIASCO_AMD_DEBUG"] = "1"
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import conftest, fiasco_amd
man = json.load(open(R + "/tests/golden/MANIFEST.json"))
lib = fiasco_amd.library(); lib.set_verbosity(0)
ora = fiasco_amd.Library(R + "/oracle/liboracle_fiasco.so"); ora.set_verbosity(0)
td = tempfile.mkdtemp()
inp = conftest.Inputs(man, td)
for name, extra in (("pred_g96x64_q60", []), ("pred_g96x64_q60", ["--rpf-mantissa", "6"]), ("pred_g96x64_q60", ["--dc-rpf-mantissa", "6"]),
                    ("pred_g256", ["-q", "60"]), ("pred_g256", ["-q", "60", "--rpf-mantissa", "6"]), ("pred_n128x96", ["-q", "45"]),
                    ("pred_n128x96", ["-q", "90", "--rpf-mantissa", "7"])):
    c = dict([x for x in man["video_cases"] if x["name"] == name][0])
    c["args"] = c["args"] + extra
    exp = conftest.encode_case(ora, c, inp, td)
    for force in ("", "1"):
        if force: os.environ["FIASCO_AMD_FORCE_GM"] = "1"
        else: os.environ.pop("FIASCO_AMD_FORCE_GM", None)
        got = conftest.encode_case(lib, c, inp, td)
        print(name, extra, "GM" if force else "auto", "ok" if got == exp else "MISMATCH %s/%s" % (None if got is None else len(got), None if exp is None else len(exp)))
