import os, sys, tempfile, subprocess
sys.path.insert(0, "tests")
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")
import test_gpu_fuzz_reference as T
env = dict(os.environ, FIASCO_DATA=T.GOLDEN + ":" + T.REF_SHARE)
env.pop("FIASCO_AMD_DEBUG", None)
exe = "/tmp/cli_on_product"
r = subprocess.run(["gcc", "-o", exe] + T.CLI_OBJS + ["-L" + os.path.join(T.ROOT, "fiasco_amd"), "-lfiasco_amd", "-Wl,-rpath," + os.path.join(T.ROOT, "fiasco_amd"), "-lm"], capture_output=True, text=True)
print(r.stderr)
td = "/tmp/fz4"; os.makedirs(td, exist_ok=True)
for seed in [int(a) for a in sys.argv[1:]]:
    args, names, shape = T.make_case(seed, td)
    print(seed, shape, " ".join(args))
    r, want = T.run(T.REF, args, names, td + "/r.fco", env)
    p, got = T.run(exe, args, names, td + "/p.fco", env)
    print("  ref rc", r.returncode, "device rc", p.returncode, "equal", got == want)
    print("  device stderr:", p.stderr[-600:])
    for extra in (["-V", "2"],):
        pass
