"""fiasco_amd -- MI355X-native FIASCO encoder hot path behind the libfiasco C API.

Python here is plumbing only: a ctypes mirror of the C-ABI shared library
``libfiasco_amd.so`` (host C + hand-written HIP for gfx950).  The names follow the
reference interface (``fiasco_coder``, ``fiasco_c_options_set_*``; reference fiasco.h:303-398)
so that tests read like calls into the reference library.

There is no CPU fallback: ``fiasco_coder`` / ``encode_batch`` fail (return 0 / raise) when
the HIP device coder cannot run.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfiasco_amd.so")
CSRC = os.path.join(_HERE, "csrc")

# enums (values are ABI, include/libfiasco_amd.h)
FIASCO_NO_VERBOSITY, FIASCO_SOME_VERBOSITY, FIASCO_ULTIMATE_VERBOSITY = 0, 1, 2
FIASCO_TILING_SPIRAL_ASC, FIASCO_TILING_SPIRAL_DSC = 0, 1
FIASCO_TILING_VARIANCE_ASC, FIASCO_TILING_VARIANCE_DSC = 2, 3
FIASCO_RPF_RANGE_0_75, FIASCO_RPF_RANGE_1_00, FIASCO_RPF_RANGE_1_50, FIASCO_RPF_RANGE_2_00 = 0, 1, 2, 3
FIASCO_PROGRESS_NONE, FIASCO_PROGRESS_BAR, FIASCO_PROGRESS_PERCENT = 0, 1, 2

# every symbol include/libfiasco_amd.h and include/libfiasco_amd_hip.h declare
EXPORTED_SYMBOLS = [
    "fiasco_get_error_message", "fiasco_set_verbosity", "fiasco_get_verbosity", "fiasco_coder",
    "fiasco_c_options_new", "fiasco_c_options_delete", "fiasco_c_options_set_smoothing",
    "fiasco_c_options_set_frame_pattern", "fiasco_c_options_set_tiling",
    "fiasco_c_options_set_basisfile", "fiasco_c_options_set_chroma_quality",
    "fiasco_c_options_set_optimizations", "fiasco_c_options_set_prediction",
    "fiasco_c_options_set_video_param", "fiasco_c_options_set_quantization",
    "fiasco_c_options_set_progress_meter", "fiasco_c_options_set_comment",
    "fiasco_c_options_set_title", "fiasco_calloc", "open_file", "fiasco_amd_set_limits",
    "fiasco_amd_get_limits", "fiasco_amd_encode_batch", "fiasco_amd_free",
    "fiasco_amd_get_stats", "fiasco_amd_reset_stats", "fiasco_amd_spec_workgroups", "fiasco_amd_core_name", "fiasco_amd_rccl_gather", "fiasco_amd_set_device",
    "fiasco_amd_batch_stage", "fiasco_amd_batch_encode", "fiasco_amd_batch_free",
    "fiasco_amd_batch_submit", "fiasco_amd_batch_collect", "fiasco_amd_batch_stats", "fiasco_amd_batch_decode_psnr", "fiasco_amd_batch_decode_psnr_all", "fiasco_amd_coop_workgroups", "fiasco_amd_share_of", "fiasco_amd_spec_append_helpers", "fiasco_amd_batch_decode_plane", "fiasco_amd_c_options_set_models",
    "fiasco_amd_release_memory", "fiasco_amd_batch_upload", "fiasco_amd_set_devices", "fiasco_amd_device_count",
    "fiasco_amd_selftest_log2", "fiasco_amd_selftest_log2_patched",
    "fiasco_amd_selftest_log2_max_ulp",
    "fiasco_amd_seq_open", "fiasco_amd_seq_free", "fiasco_amd_seq_gops", "fiasco_amd_seq_frames",
    "fiasco_amd_seq_gop_of", "fiasco_amd_seq_ycol_size", "fiasco_amd_seq_initial_level",
    "fiasco_amd_seq_search", "fiasco_amd_seq_gop_result", "fiasco_amd_seq_ycol", "fiasco_amd_seq_write",
    "fiasco_amd_seq_probe",
]


class Stats(ctypes.Structure):
    """struct fiasco_amd_stats (include/libfiasco_amd_hip.h)."""
    _fields_ = [("kernel_ms", ctypes.c_double), ("launches", ctypes.c_ulonglong),
                ("frames", ctypes.c_ulonglong), ("bytes_mp", ctypes.c_ulonglong),
                ("bytes_img", ctypes.c_ulonglong), ("bytes_gram", ctypes.c_ulonglong),
                ("n_mp", ctypes.c_ulonglong), ("n_steps", ctypes.c_ulonglong),
                ("n_blocks", ctypes.c_ulonglong), ("n_appends", ctypes.c_ulonglong),
                ("n_fulleval", ctypes.c_ulonglong), ("t_init", ctypes.c_ulonglong),
                ("t_approx", ctypes.c_ulonglong), ("t_ipis", ctypes.c_ulonglong),
                ("t_append", ctypes.c_ulonglong), ("t_serial", ctypes.c_ulonglong),
                ("t_total", ctypes.c_ulonglong), ("t_mpA", ctypes.c_ulonglong),
                ("t_mpB", ctypes.c_ulonglong), ("n_blockevals", ctypes.c_ulonglong),
                ("dbg", ctypes.c_ulonglong * 8), ("states_sum", ctypes.c_ulonglong),
                ("states_max", ctypes.c_ulonglong), ("reencodes", ctypes.c_ulonglong),
                ("frames_by_build", ctypes.c_ulonglong * 5),
                ("spec_frames", ctypes.c_ulonglong), ("spec_tasks", ctypes.c_ulonglong),
                ("spec_confirmed", ctypes.c_ulonglong), ("spec_wrong", ctypes.c_ulonglong),
                ("spec_timeout", ctypes.c_ulonglong), ("spec_inline", ctypes.c_ulonglong),
                ("spec_wait", ctypes.c_ulonglong), ("spec_tab_used", ctypes.c_ulonglong),
                ("spec_tab_missed", ctypes.c_ulonglong), ("spec_adopted", ctypes.c_ulonglong),
                ("decoder_frames", ctypes.c_ulonglong), ("decoder_bytes", ctypes.c_ulonglong),
                ("decoder_us", ctypes.c_ulonglong), ("coop_frames", ctypes.c_ulonglong),
                ("coop_workgroups", ctypes.c_ulonglong),
                ("spec_app_rows", ctypes.c_ulonglong), ("spec_app_wait", ctypes.c_ulonglong)]


def build(verbose=False):
    """Compile libfiasco_amd.so (gcc for the host C, hipcc --offload-arch=gfx950 for the
    device coder).  Works without a GPU (cross compilation)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC], stdout=out)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce %s" % LIB_PATH)


class FiascoError(RuntimeError):
    pass


class Library:
    """ctypes binding of one libfiasco-compatible shared library."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise FiascoError("%s is missing: run fiasco_amd.build() (there is no fallback)" % path)
        self.path = path
        L = self.L = ctypes.CDLL(path)
        c = ctypes
        L.fiasco_get_error_message.restype = c.c_char_p
        L.fiasco_set_verbosity.argtypes = [c.c_int]
        L.fiasco_get_verbosity.restype = c.c_int
        L.fiasco_c_options_new.restype = c.c_void_p
        L.fiasco_c_options_delete.argtypes = [c.c_void_p]
        L.fiasco_coder.argtypes = [c.POINTER(c.c_char_p), c.c_char_p, c.c_float, c.c_void_p]
        L.fiasco_coder.restype = c.c_int
        L.fiasco_amd_c_options_set_models.argtypes = [c.c_void_p, c.c_char_p, c.c_char_p, c.c_char_p, c.c_char_p]
        L.fiasco_amd_c_options_set_models.restype = c.c_int
        for name, args in [
            ("set_smoothing", [c.c_int]), ("set_frame_pattern", [c.c_char_p]),
            ("set_tiling", [c.c_int, c.c_uint]), ("set_basisfile", [c.c_char_p]),
            ("set_chroma_quality", [c.c_float, c.c_uint]),
            ("set_optimizations", [c.c_uint] * 5), ("set_prediction", [c.c_int, c.c_uint, c.c_uint]),
            ("set_video_param", [c.c_uint, c.c_int, c.c_int, c.c_int]),
            ("set_quantization", [c.c_uint, c.c_int, c.c_uint, c.c_int]),
            ("set_progress_meter", [c.c_int]), ("set_comment", [c.c_char_p]),
            ("set_title", [c.c_char_p]),
        ]:
            fn = getattr(L, "fiasco_c_options_" + name)
            fn.argtypes = [c.c_void_p] + args
            fn.restype = c.c_int
        L.fiasco_amd_set_limits.argtypes = [c.c_uint, c.c_uint]
        L.fiasco_amd_set_limits.restype = c.c_int
        L.fiasco_amd_get_limits.argtypes = [c.POINTER(c.c_uint), c.POINTER(c.c_uint)]
        L.fiasco_amd_encode_batch.argtypes = [c.c_uint, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t),
                                              c.c_float, c.c_void_p, c.POINTER(c.c_void_p),
                                              c.POINTER(c.c_size_t)]
        L.fiasco_amd_encode_batch.restype = c.c_int
        L.fiasco_amd_free.argtypes = [c.c_void_p]
        L.fiasco_amd_core_name.restype = c.c_char_p
        if hasattr(L, "fiasco_amd_get_stats"):
            L.fiasco_amd_get_stats.argtypes = [c.POINTER(Stats)]

    # -- misc ------------------------------------------------------------------
    def error_message(self):
        return self.L.fiasco_get_error_message().decode("latin-1")

    def core_name(self):
        return self.L.fiasco_amd_core_name().decode()

    def set_verbosity(self, level):
        self.L.fiasco_set_verbosity(level)

    def set_limits(self, max_states, max_level):
        if not self.L.fiasco_amd_set_limits(max_states, max_level):
            raise FiascoError(self.error_message())

    def get_limits(self):
        a, b = ctypes.c_uint(), ctypes.c_uint()
        self.L.fiasco_amd_get_limits(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def set_devices(self, ids):
        """fiasco_amd_set_devices: the devices the batch entries spread their frames over ([] = automatic)."""
        arr = (ctypes.c_int * max(len(ids), 1))(*ids)
        self.L.fiasco_amd_set_devices.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        if not self.L.fiasco_amd_set_devices(arr, len(ids)):
            raise FiascoError(self.error_message())

    def device_count(self):
        return self.L.fiasco_amd_device_count()

    def set_device(self, device):
        self.L.fiasco_amd_set_device.argtypes = [ctypes.c_int]
        if not self.L.fiasco_amd_set_device(device):
            raise FiascoError(self.error_message())

    def get_stats(self):
        st = Stats()
        self.L.fiasco_amd_get_stats(ctypes.byref(st))
        return st

    def reset_stats(self):
        self.L.fiasco_amd_reset_stats()

    # -- options ---------------------------------------------------------------
    def c_options_new(self):
        return COptions(self)

    def cli_options(self, optimize=0, dictionary_size=10000, progress=FIASCO_PROGRESS_NONE, **kw):
        """Options object configured like reference bin/cwfa.c:252-393 does for the CLI
        defaults (block levels [6,10] and 3 elements at --optimize 0; [4,12] / 5 above)."""
        o = self.c_options_new()
        o.set_frame_pattern(kw.get("pattern", "ippppppppp"))
        o.set_chroma_quality(kw.get("chroma_qfactor", 2.0), kw.get("chroma_dictionary", 40))
        o.set_smoothing(kw.get("smooth", 70))
        o.set_progress_meter(progress)
        o.set_tiling({"desc-variance": FIASCO_TILING_VARIANCE_DSC, "asc-variance": FIASCO_TILING_VARIANCE_ASC,
                      "asc-spiral": FIASCO_TILING_SPIRAL_ASC, "desc-spiral": FIASCO_TILING_SPIRAL_DSC}
                     [kw.get("tiling_method", "desc-variance")], kw.get("tiling_exponent", 4))
        if optimize <= 0:
            o.set_optimizations(6, 10, 3, dictionary_size, 0)
        else:
            o.set_optimizations(4, 12, 5, dictionary_size, optimize - 1)
        o.set_prediction(0, kw.get("min_level", 6), kw.get("max_level", 10))
        o.set_quantization(3, FIASCO_RPF_RANGE_1_50, 5, FIASCO_RPF_RANGE_1_00)
        return o

    # -- coder -----------------------------------------------------------------
    def fiasco_coder(self, inputnames, outputname, quality=20.0, options=None):
        """int fiasco_coder(inputname[], outputname, quality, options): 1 ok / 0 failure."""
        arr = (ctypes.c_char_p * (len(inputnames) + 1))()
        for i, n in enumerate(inputnames):
            arr[i] = os.fsencode(n)
        arr[len(inputnames)] = None
        return self.L.fiasco_coder(arr, os.fsencode(outputname) if outputname else None,
                                   ctypes.c_float(quality), options.handle if options else None)

    def encode_batch(self, pnm_list, quality=20.0, options=None):
        """fiasco_amd_encode_batch: independent stills (raw PNM bytes) -> list of .fco bytes
        (None for a frame that failed)."""
        n = len(pnm_list)
        bufs = (ctypes.c_char_p * n)(*pnm_list)
        lens = (ctypes.c_size_t * n)(*[len(b) for b in pnm_list])
        outs = (ctypes.c_void_p * n)()
        olen = (ctypes.c_size_t * n)()
        self.L.fiasco_amd_encode_batch(n, bufs, lens, ctypes.c_float(quality),
                                       options.handle if options else None, outs, olen)
        res = []
        for i in range(n):
            if outs[i]:
                res.append(ctypes.string_at(outs[i], olen[i]))
                self.L.fiasco_amd_free(outs[i])
            else:
                res.append(None)
        return res


class Batch:
    """Staged batch (fiasco_amd_batch_stage / _encode / _free): inputs stay resident in HBM
    between encode() calls."""

    def __init__(self, lib, pnm_list, quality=20.0, options=None):
        c = ctypes
        L = lib.L
        L.fiasco_amd_batch_stage.argtypes = [c.c_uint, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t),
                                             c.c_float, c.c_void_p]
        L.fiasco_amd_batch_stage.restype = c.c_void_p
        L.fiasco_amd_batch_encode.argtypes = [c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_size_t)]
        L.fiasco_amd_batch_encode.restype = c.c_int
        L.fiasco_amd_batch_free.argtypes = [c.c_void_p]
        L.fiasco_amd_batch_submit.argtypes = [c.c_void_p]
        L.fiasco_amd_batch_submit.restype = c.c_int
        L.fiasco_amd_batch_collect.argtypes = [c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_size_t), c.c_int]
        L.fiasco_amd_batch_collect.restype = c.c_int
        self.lib = lib
        self.n = len(pnm_list)
        bufs = (c.c_char_p * self.n)(*pnm_list)
        lens = (c.c_size_t * self.n)(*[len(b) for b in pnm_list])
        self.handle = L.fiasco_amd_batch_stage(self.n, bufs, lens, c.c_float(quality),
                                               options.handle if options else None)
        if not self.handle:
            raise FiascoError(lib.error_message())

    def upload(self, pnm_list):
        """fiasco_amd_batch_upload: new frames for every slot (host PNM buffers -> pinned ->
        HBM, not waited for); the next submit / collect(resubmit=True) encodes them."""
        c = ctypes
        assert len(pnm_list) == self.n
        f = self.lib.L.fiasco_amd_batch_upload
        f.argtypes = [c.c_void_p, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t)]
        f.restype = c.c_int
        bufs = (c.c_char_p * self.n)(*pnm_list)
        lens = (c.c_size_t * self.n)(*[len(b) for b in pnm_list])
        if not f(self.handle, bufs, lens):
            raise FiascoError(self.lib.error_message())

    def submit(self):
        """fiasco_amd_batch_submit: start a pass over the resident inputs, do not wait."""
        return self.lib.L.fiasco_amd_batch_submit(self.handle)

    def collect(self, resubmit=False):
        """fiasco_amd_batch_collect: streams of the submitted pass; with resubmit the next pass
        is started before the host writes them (writer of pass i overlaps search of pass i+1)."""
        return self._run(lambda outs, olen: self.lib.L.fiasco_amd_batch_collect(
            self.handle, outs, olen, 1 if resubmit else 0))

    def encode(self):
        return self._run(lambda outs, olen: self.lib.L.fiasco_amd_batch_encode(self.handle, outs, olen))

    def _run(self, call):
        outs = (ctypes.c_void_p * self.n)()
        olen = (ctypes.c_size_t * self.n)()
        call(outs, olen)
        res = []
        for i in range(self.n):
            if outs[i]:
                res.append(ctypes.string_at(outs[i], olen[i]))
                self.lib.L.fiasco_amd_free(outs[i])
            else:
                res.append(None)
        return res

    def stats(self, i, band=0):
        """fiasco_amd_batch_stats: root-range costs / squared error of frame i (band 0..2) of the
        last finished pass and the coder-side PSNR the reference reports (codec/coder.c:918-923)."""
        import math
        c = ctypes
        f = self.lib.L.fiasco_amd_batch_stats
        f.argtypes = [c.c_void_p, c.c_uint, c.c_uint, c.POINTER(c.c_float), c.POINTER(c.c_float),
                      c.POINTER(c.c_uint), c.POINTER(c.c_uint)]
        f.restype = c.c_int
        costs, err, w, h = c.c_float(), c.c_float(), c.c_uint(), c.c_uint()
        if not f(self.handle, i, band, costs, err, w, h):
            return None
        mse = err.value / w.value / h.value
        return {"costs": costs.value, "err": err.value, "width": w.value, "height": h.value,
                "psnr_db": 10.0 * math.log10(255.0 * 255.0 / mse) if mse > 0 else float("inf")}

    def decode_plane(self, i, band, width, height):
        """fiasco_amd_batch_decode_plane: the decoded band as bytes (the payload of dfiasco -s 0's PGM for gray)."""
        f = self.lib.L.fiasco_amd_batch_decode_plane
        f.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_char_p]
        f.restype = ctypes.c_int
        buf = ctypes.create_string_buffer(width * height)
        if not f(self.handle, i, band, buf):
            raise FiascoError(self.lib.error_message())
        return buf.raw

    def decode_psnr_all(self):
        """fiasco_amd_batch_decode_psnr_all: (n decoded, [[psnr dB per band]], [[mse per band]]) of all frames,
        decoded by one call of the device decoder."""
        f = self.lib.L.fiasco_amd_batch_decode_psnr_all
        n = self.n
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        f.restype = ctypes.c_int
        p = (ctypes.c_double * (3 * n))(); m = (ctypes.c_double * (3 * n))()
        good = f(self.handle, p, m)
        return good, [list(p[3 * i:3 * i + 3]) for i in range(n)], [list(m[3 * i:3 * i + 3]) for i in range(n)]

    def decode_psnr(self, i):
        """fiasco_amd_batch_decode_psnr: decoded PSNR in dB per band of frame i (what `dfiasco -s 0` +
        `pnmpsnr` print for a gray frame), and the mean squared errors."""
        f = self.lib.L.fiasco_amd_batch_decode_psnr
        f.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        f.restype = ctypes.c_int
        ps, ms = (ctypes.c_double * 3)(), (ctypes.c_double * 3)()
        if not f(self.handle, i, ps, ms):
            raise FiascoError(self.lib.error_message())
        return list(ps), list(ms)

    def free(self):
        if self.handle:
            self.lib.L.fiasco_amd_batch_free(self.handle)
            self.handle = None


class Sequence:
    """fiasco_amd_seq_*: one video, this process searching and writing every world-th group of
    pictures (include/libfiasco_amd.h).  Frames: raw PNM bytes in display order."""

    def __init__(self, lib, pnm_list, quality=20.0, options=None, rank=0, world=1):
        c = ctypes
        L = self.L = lib.L
        self.lib = lib
        L.fiasco_amd_seq_open.argtypes = [c.c_uint, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t), c.c_float,
                                          c.c_void_p, c.c_uint, c.c_uint]
        L.fiasco_amd_seq_open.restype = c.c_void_p
        L.fiasco_amd_seq_free.argtypes = [c.c_void_p]
        for name in ("gops", "frames", "ycol_size", "initial_level"):
            f = getattr(L, "fiasco_amd_seq_" + name)
            f.argtypes = [c.c_void_p]; f.restype = c.c_uint
        L.fiasco_amd_seq_gop_of.argtypes = [c.c_void_p, c.c_uint]; L.fiasco_amd_seq_gop_of.restype = c.c_uint
        L.fiasco_amd_seq_search.argtypes = [c.c_void_p, c.POINTER(c.c_uint), c.POINTER(c.c_ubyte)]
        L.fiasco_amd_seq_search.restype = c.c_int
        L.fiasco_amd_seq_gop_result.argtypes = [c.c_void_p, c.c_uint, c.POINTER(c.c_uint), c.POINTER(c.c_int)]
        L.fiasco_amd_seq_gop_result.restype = c.c_int
        L.fiasco_amd_seq_ycol.argtypes = [c.c_void_p, c.c_uint]; L.fiasco_amd_seq_ycol.restype = c.c_void_p
        L.fiasco_amd_seq_write.argtypes = [c.c_void_p, c.c_uint, c.c_char_p, c.POINTER(c.c_void_p), c.POINTER(c.c_size_t)]
        L.fiasco_amd_seq_write.restype = c.c_int
        self._keep = (list(pnm_list), options)                      # borrowed by the C side
        n = len(pnm_list)
        self._bufs = (c.c_char_p * n)(*pnm_list)
        self._lens = (c.c_size_t * n)(*[len(b) for b in pnm_list])
        self.handle = L.fiasco_amd_seq_open(n, self._bufs, self._lens, c.c_float(quality),
                                            options.handle if options else None, rank, world)
        if not self.handle:
            raise FiascoError(lib.error_message())
        self.rank, self.world = rank, world
        self.gops = L.fiasco_amd_seq_gops(self.handle)
        self.frames = L.fiasco_amd_seq_frames(self.handle)
        self.ycol_size = L.fiasco_amd_seq_ycol_size(self.handle)
        self.initial_level = L.fiasco_amd_seq_initial_level(self.handle)

    def gop_of(self, frame):
        return self.L.fiasco_amd_seq_gop_of(self.handle, frame)

    def probe(self):
        """Level to speculate for the GOPs behind the first: what frame 0 alone leaves."""
        c = ctypes
        self.L.fiasco_amd_seq_probe.argtypes = [c.c_void_p, c.POINTER(c.c_uint)]
        self.L.fiasco_amd_seq_probe.restype = c.c_int
        lv = c.c_uint()
        if not self.L.fiasco_amd_seq_probe(self.handle, lv):
            raise FiascoError(self.lib.error_message())
        return lv.value

    def search(self, carry_in, todo):
        c = ctypes
        ci = (c.c_uint * self.gops)(*carry_in)
        td = (c.c_ubyte * self.gops)(*[1 if t else 0 for t in todo])
        if not self.L.fiasco_amd_seq_search(self.handle, ci, td):
            raise FiascoError(self.lib.error_message())

    def gop_result(self, gop):
        """(minimum level the GOP left, failed, message) -- None if the GOP was not searched here."""
        c = ctypes
        out, failed = c.c_uint(), c.c_int()
        if not self.L.fiasco_amd_seq_gop_result(self.handle, gop, out, failed):
            return None
        return out.value, bool(failed.value), self.lib.error_message() if failed.value else ""

    def ycol(self, frame):
        p = self.L.fiasco_amd_seq_ycol(self.handle, frame)
        return ctypes.string_at(p, self.ycol_size) if p and self.ycol_size else None

    def write(self, frame, ycol=None):
        c = ctypes
        out, n = c.c_void_p(), c.c_size_t()
        if not self.L.fiasco_amd_seq_write(self.handle, frame, ycol, out, n):
            raise FiascoError(self.lib.error_message())
        data = c.string_at(out, n.value)
        self.L.fiasco_amd_free(out)
        return data

    def free(self):
        if self.handle:
            self.L.fiasco_amd_seq_free(self.handle)
            self.handle = None


class COptions:
    """fiasco_c_options_t with the reference's setter names (fiasco.h:132-174)."""

    def __init__(self, lib):
        self.lib = lib
        self.handle = lib.L.fiasco_c_options_new()
        if not self.handle:
            raise FiascoError(lib.error_message())

    def _call(self, name, *args):
        conv = [a.encode() if isinstance(a, str) else a for a in args]
        ok = getattr(self.lib.L, "fiasco_c_options_" + name)(self.handle, *conv)
        if not ok:
            raise FiascoError(self.lib.error_message())
        return ok

    def delete(self):
        if self.handle:
            self.lib.L.fiasco_c_options_delete(self.handle)
            self.handle = None

    def __getattr__(self, name):
        if name.startswith("set_"):
            return lambda *a: self._call(name, *a)
        raise AttributeError(name)


_default = None


def library():
    """The product library (HIP hot path).  Raises if it has not been built."""
    global _default
    if _default is None:
        _default = Library(LIB_PATH)
        if _default.core_name() != "hip-gfx950":
            raise FiascoError("libfiasco_amd.so is not linked against the HIP device coder")
    return _default


def fiasco_coder(inputnames, outputname, quality=20.0, options=None):
    return library().fiasco_coder(inputnames, outputname, quality, options)


def encode_batch(pnm_list, quality=20.0, options=None):
    return library().encode_batch(pnm_list, quality, options)
