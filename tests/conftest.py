"""pytest configuration: markers, library handles, golden-vector helpers.

Layout of the checks (see DESIGN.md "Parity"):
  -m "not gpu" : oracle (CPU restatement) vs the golden vectors produced by the real
                 reference; host-side API behaviour; C-ABI exports; 2-rank gloo sharding.
  -m gpu       : the HIP device coder, called through the C ABI, vs golden vectors and vs
                 the oracle on seeded inputs; batch API; full-size properties.
Only tests may touch oracle/ (the product never does).
"""
import hashlib
import json
import os
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle_fiasco.so")
# where --basis-name looks (open_file(): cwd, then the FIASCO_DATA list): our own long bases
# (tests/golden/long_*.fco) and the reference's installed data files (oracle/_ref/share, oracle/ref_build.sh)
REF_SHARE = os.path.join(ROOT, "oracle", "_ref", "share")
os.environ.setdefault("FIASCO_DATA", GOLDEN + ":" + REF_SHARE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(path):
    subprocess.check_call(["make", "-C", path], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle behind the same C API (test infrastructure)."""
    import fiasco_amd
    if not os.path.exists(ORACLE_LIB):
        _make(os.path.join(ROOT, "oracle"))
    lib = fiasco_amd.Library(ORACLE_LIB)
    assert lib.core_name() == "oracle-cpu"
    lib.set_verbosity(0)
    return lib


@pytest.fixture(scope="session")
def product():
    """The product library: host C + HIP device coder."""
    import fiasco_amd
    if not os.path.exists(fiasco_amd.LIB_PATH):
        fiasco_amd.build()
    lib = fiasco_amd.Library(fiasco_amd.LIB_PATH)
    assert lib.core_name() == "hip-gfx950"
    lib.set_verbosity(0)
    return lib


class Inputs:
    """Materialises golden inputs (committed file, or re-synthesised + md5 verified)."""

    def __init__(self, manifest, tmpdir):
        self.man = manifest
        self.tmp = str(tmpdir)
        self.cache = {}

    def data(self, name):
        if name in self.cache:
            return self.cache[name]
        import synth
        ent = self.man["inputs"][name]
        if ent.get("file"):
            d = open(os.path.join(GOLDEN, ent["file"]), "rb").read()
        else:
            a = ent["args"]
            if ent["kind"] == "synth":
                d = synth.pgm_bytes(synth.synth(a["w"], a["h"], a["seed"], a.get("shift", 0)))
            elif ent["kind"] == "noise":
                d = synth.pgm_bytes(synth.noise(a["w"], a["h"], a["seed"]))
            elif ent["kind"] == "color_k":
                d = synth.ppm_bytes(synth.synth_color_k(a["w"], a["h"], 1234, a.get("shift", 0)))
            else:
                d = synth.ppm_bytes(synth.synth_color_c(a["w"], a["h"], a["f"]))
        assert hashlib.md5(d).hexdigest() == ent["md5"], "input %s differs from the pinned md5" % name
        self.cache[name] = d
        return d

    def path(self, name):
        ent = self.man["inputs"][name]
        p = os.path.join(self.tmp, name + "." + ent["ext"])
        if not os.path.exists(p):
            open(p, "wb").write(self.data(name))
        return p


@pytest.fixture(scope="session")
def inputs(manifest, tmp_path_factory):
    return Inputs(manifest, tmp_path_factory.mktemp("inputs"))


@pytest.fixture(scope="session")
def manifest_big():
    """Known answers of the limits-extension reference build (oracle/ref_build.sh: MAXSTATES 30000,
    MAXLEVEL 26; tests/golden/make_golden_big.py): what the stock reference cannot encode."""
    return json.load(open(os.path.join(GOLDEN, "MANIFEST_BIG.json")))


@pytest.fixture(scope="session")
def inputs_big(manifest_big, tmp_path_factory):
    return Inputs(manifest_big, tmp_path_factory.mktemp("inputs_big"))


def options_from_args(lib, args):
    """Translate the cfiasco command line of a golden case into API calls the way
    reference bin/cwfa.c:252-393 does.  Returns (quality, options)."""
    import fiasco_amd
    kw = {}
    quality, optimize, dict_size = 20.0, 0, 10000
    title = comment = None
    rpf = dict(m=3, r=1.5, dm=5, dr=1.0)
    i = 0
    pred, pmin, pmax = 0, 6, 10
    while i < len(args):
        a = args[i]
        if a in ("--prediction", "--half-pixel"):      # flags; cfiasco drops --half-pixel
            pred = pred or a == "--prediction"
            i += 1
            continue
        v = args[i + 1]
        if a == "-q": quality = float(v)
        elif a == "--min-level": pmin = int(v)
        elif a == "--max-level": pmax = int(v)
        elif a == "-z": optimize = int(v)
        elif a == "--dictionary-size": dict_size = int(v)
        elif a == "--pattern": kw["pattern"] = v
        elif a == "-t": title = v
        elif a == "-c": comment = v
        elif a == "--rpf-mantissa": rpf["m"] = int(v)
        elif a == "--rpf-range": rpf["r"] = float(v)
        elif a == "--dc-rpf-mantissa": rpf["dm"] = int(v)
        elif a == "--dc-rpf-range": rpf["dr"] = float(v)
        elif a == "--chroma-qfactor": kw["chroma_qfactor"] = float(v)
        elif a == "--chroma-dictionary": kw["chroma_dictionary"] = int(v)
        elif a == "--tiling-exponent": kw["tiling_exponent"] = int(v)
        elif a == "--tiling-method": kw["tiling_method"] = v
        elif a == "--basis-name": kw["basis_name"] = v
        else: raise ValueError(a)
        i += 2
    o = lib.cli_options(optimize=optimize, dictionary_size=dict_size, **kw)
    o.set_prediction(1 if pred else 0, pmin, pmax)

    def rng(r):
        return 0 if r < 1 else 1 if r < 1.5 else 2 if r < 2.0 else 3
    o.set_quantization(rpf["m"], rng(rpf["r"]), rpf["dm"], rng(rpf["dr"]))
    if title: o.set_title(title)
    if comment: o.set_comment(comment)
    if "basis_name" in kw: o.set_basisfile(kw["basis_name"].encode())
    return quality, o


def option_cases(manifest):
    """The "option_cases" of the manifest (tests/golden/make_options.py) that can run here: the ones with the
    reference's own medium.fco / large.fco need oracle/_ref/share."""
    return [c for c in manifest["option_cases"] if c.get("needs") != "share" or os.path.exists(os.path.join(REF_SHARE, "medium.fco"))]


def encode_case(lib, case, inputs, outdir):
    """Run fiasco_coder() for one golden case; returns the stream bytes (or None)."""
    q, o = options_from_args(lib, case["args"])
    out = os.path.join(str(outdir), case["name"] + "." + lib.core_name() + ".fco")
    rc = lib.fiasco_coder([inputs.path(n) for n in case["inputs"]], out, q, o)
    o.delete()
    if rc != 1:
        return None
    return open(out, "rb").read()
