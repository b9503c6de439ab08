"""developer probe: which earlier encodes make a later one differ (state leaking between calls)"""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def run(seq):
    import fiasco_amd, conftest
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST.json")))
    lib = fiasco_amd.library(); lib.set_verbosity(0)
    td = tempfile.mkdtemp()
    inputs = conftest.Inputs(man, td)
    res = []
    for name, narrow in seq:
        case = [c for c in man["video_cases"] if c["name"] == name][0]
        if narrow: os.environ["FIASCO_AMD_NO_WIDE"] = "1"
        else: os.environ.pop("FIASCO_AMD_NO_WIDE", None)
        got = conftest.encode_case(lib, case, inputs, td)
        res.append(got is not None and hashlib.md5(got).hexdigest() == case["md5"])
    return res

if __name__ == "__main__":
    if len(sys.argv) > 1:
        seq = json.loads(sys.argv[1])
        print("RESULT", json.dumps(run(seq)))
        sys.exit(0)
    full = [(n, w) for n in ["pred_g256", "pred_n128x96", "pred_g96x64_q60", "pred_g256_z1", "pred_g256_lv79", "pred_c256"]
            for w in (0, 1)] + [("seq2_gray_ip", 0), ("seq2_gray_ip_pred", 0), ("seq4_gray_ippp", 0), ("seq3_color_ipp", 0)]
    def ok(seq):
        out = subprocess.run([sys.executable, __file__, json.dumps(seq)], capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("RESULT")]
        return json.loads(line[0][7:])[-1] if line else None
    print("full:", ok(full))
    cur = full
    i = 0
    while i < len(cur) - 1:
        trial = cur[:i] + cur[i + 1:]
        if ok(trial) is False: cur = trial
        else: i += 1
    print("minimal failing sequence:", cur)
