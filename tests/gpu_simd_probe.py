import os, sys
os.environ.setdefault("FIASCO_AMD_DEBUG", "1")     # the library honours its developer switches only with this
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import synth, fiasco_amd
lib = fiasco_amd.library(); lib.set_verbosity(0)
opt = lib.cli_options()
uniq = [synth.pgm_bytes(synth.synth(640, 480, 1000 + i)) for i in range(8)]
frames = [uniq[i % 8] for i in range(1024)]
b = fiasco_amd.Batch(lib, frames, 20.0, opt)
lib.reset_stats(); out = b.encode(); st = lib.get_stats()
v = st.dbg[7]
print("wave-0 SIMD histogram:", [(v >> (16 * k)) & 0xffff for k in range(4)], "frames by build", list(st.frames_by_build))
