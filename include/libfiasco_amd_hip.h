/*
 *  libfiasco_amd_hip.h -- measurement hooks of the HIP hot path (C ABI, plain types).
 *
 *  The data-path boundary itself is fiasco_coder() / fiasco_amd_encode_batch() in
 *  libfiasco_amd.h; what is declared here only reports what the device coder did, so that
 *  bench.py can compute the roofline figures from the coder's OWN per-call counters
 *   and from HIP-event time measured on the stream the kernel ran on.
 */
#ifndef LIBFIASCO_AMD_HIP_H
#define LIBFIASCO_AMD_HIP_H 1
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fiasco_amd_stats {
    double             kernel_ms;   /* sum of HIP-event durations of fiasco_frame_kernel     */
    unsigned long long launches;    /* kernel launches                                       */
    unsigned long long frames;      /* frames encoded successfully                           */
    /* algorithmic bytes, summed per call from the coder's counters:
     *   bytes_mp   = sum over matching-pursuit calls of 4*D*(2+S) + 4*2^L
     *   bytes_img  = sum over init_range blocks of N*(128+4*NS) + 4*2^lc_max
     *   bytes_gram = sum over appended states of 4*(NL-1)*(s+1)*(1+E) + 4*(NL-1)*(s+1)     */
    unsigned long long bytes_mp, bytes_img, bytes_gram;
    unsigned long long n_mp, n_steps, n_blocks, n_appends, n_fulleval;
    /* per-phase time summed over frames, 100 MHz ticks of workgroup lane 0:
     * init_range, matching pursuit, incremental <block,state> tables, state append
     * (images + Gram rows), serial partition-search bookkeeping, whole frame */
    unsigned long long t_init, t_approx, t_ipis, t_append, t_serial, t_total;
    /* inside matching pursuit: candidate-parallel phase, ordered-replay phase (ticks); number
     * of 64-candidate blocks whose survivors were fully evaluated */
    unsigned long long t_mpA, t_mpB, n_blockevals;
    unsigned long long dbg[8];      /* free-form developer counters */
    /* states of the finished automata (sum / largest) and frames that were encoded a second time
     * because the capacity guess of their slab was too small */
    unsigned long long states_sum, states_max, reencodes;
    /* frames launched per kernel build: default 256 / 1024 threads, big 256 / 512 threads, default
     * 1024 threads with triangular Gram tables */
    unsigned long long frames_by_build[5];
    /* block-level speculation (several workgroups per frame, launches that leave workgroup slots of
     * the chip free): frames encoded that way, blocks whose subtree search was handed to a verifier
     * workgroup, verdicts that confirmed / contradicted the chain's guess, verifications given up
     * after a bounded wait, blocks the chain searched itself, ticks (100 MHz) it waited for verdicts */
    unsigned long long spec_frames, spec_tasks, spec_confirmed, spec_wrong, spec_timeout, spec_inline, spec_wait;
    /* blocks whose <sub-block, state> tables a table worker had ready for the chain / had not */
    unsigned long long spec_tab_used, spec_tab_missed;
    /* wrong guesses after which the chain took over the verifier's state instead of searching the block again */
    unsigned long long spec_adopted;
    /* the device decoder (csrc/hip/frame_decoder.inc): frames decoded, their algorithmic bytes
     * (2 bytes per pixel of every level image written and per (pixel, term) read, + the frame), device time of
     * the flights in microseconds (HIP events around uploads + kernels of a flight of <= 32 frames) */
    unsigned long long decoder_frames, decoder_bytes, decoder_us;
    /* big frames whose table passes were built by several workgroups (frame_coder.h FcCoop), and how many each */
    unsigned long long coop_frames, coop_workgroups;
    /* speculating frames with append helpers (frame_coder.h FcSpecCtl.app_*): Gram rows of appended states that were
     * dealt to the helpers, ticks (100 MHz) the chains waited for them */
    unsigned long long spec_app_rows, spec_app_wait;
} fiasco_amd_stats;

void fiasco_amd_get_stats(fiasco_amd_stats *out);
void fiasco_amd_reset_stats(void);

/* The launcher's choice for a launch of `frames` frames on a device with `cus` compute units: how many
 * workgroups a frame gets (chain + table workers + verifiers, block-level speculation; 0 = one workgroup per
 * frame).  `big_frames`: frames beyond 2048 pixels in a dimension (4K); `narrow_only`: every frame fits the
 * 256-thread build (at most 3072 states with the verifiers' ids); `occupancy`: workgroups of that build a CU
 * holds at once.  A pure function (no device needed); FIASCO_AMD_SPEC overrides it at run time. */
int fiasco_amd_spec_workgroups(unsigned frames, int cus, int big_frames, int narrow_only, int occupancy);

/* name of the hot-path backend linked into this library: "hip-gfx950" for the product,
 * "oracle-cpu" for the test-only oracle library (reference seam: codec/approx.h:24-27,
 * codec/ip.h:22-34, codec/subdivide.h -- the functions the backend replaces). */
const char *fiasco_amd_core_name(void);

/* One process per GPU (BASELINE config 4: frames dealt round robin to the ranks, item i on rank
 * i mod world): the finished streams of all ranks meet on `root` -- the job's only communication, three small
 * collectives over RCCL / xGMI (counts + failure flags, a status round once every rank has its buffers, then lengths +
 * payloads in one padded all-gather).  EVERY rank must call it; a failure of one rank (a device error, no memory for
 * the payload, a deal that is not round robin: rank r must hold ceil((total - r) / world) streams) travels in the
 * next message and all ranks return 0 together before the big all-gather -- nobody is left waiting in a collective.
 * (Exception: a rank that cannot allocate the first 24 (world + 1) bytes of device memory.)  `comm` is the caller's
 * ncclComm_t, `stream` a hipStream_t (or NULL); every rank passes its n_local streams.  On `root`: *all / *all_len hold
 * the *n_all streams of the job in item order (free each with fiasco_amd_free(), the two arrays with free()); the other
 * ranks get *n_all = 0.  RCCL is taken from the process at run time (no link-time dependency).  1 ok / 0 + message.
 * In the reference nothing corresponds: it is single threaded (codec/coder.c:490-668 is the loop that is sharded). */
int fiasco_amd_rccl_gather(void *comm, void *stream, int rank, int world, int root,
                           unsigned n_local, const unsigned char *const *data, const size_t *len,
                           unsigned char ***all, size_t **all_len, unsigned *n_all);

/* Devices.  Frames are independent units: every batch entry -- fiasco_amd_encode_batch(),
 * the staged batches, fiasco_coder() on an all-intra stream or a video (its groups of pictures) -- spreads
 * its frames round robin over the devices of the process, one host thread + stream + slab pool per
 * device, results in input order, no collective.  The devices are: FIASCO_AMD_DEVICES="0,1,..." from the
 * environment if set; else what fiasco_amd_set_devices() chose; else the ONE device of
 * fiasco_amd_set_device() (one process per GPU: the multi-process harness); else every visible device.
 * An id may be listed twice (two shares on one GPU).  Replacement of staged inputs (fiasco_amd_batch_upload) works
 * with several devices too: one pinned host buffer, every share copies the planes of ITS frames.  Threading: the
 * batch entries of ONE process may be called from several host threads, but calls that spread over more than one
 * device share run one after the other (the shares' worker threads belong to the process, core_hip.cpp
 * for_each_share).  All return 1 on success, 0 + error message. */
/* workgroups per frame the launcher gives the table passes of `frames` big frames (prediction, P/B frames, -z 1/2)
 * on a chip of `cus` CUs: 1, 2, 4 or 8 (csrc/hip/frame_coder.h FcCoop); pure function */
unsigned fiasco_amd_coop_workgroups(unsigned frames, int cus);
/* append helpers per frame the launcher adds to a launch of `frames` speculating frames with G workgroups each
 * (wide_build: the 1024-thread build, frames beyond 3072 states): workgroups that build their shares of the Gram row of
 * every state the chain appends (codec/control.c:48-131, codec/ip.c:184-260; csrc/hip/frame_coder.h FcSpecCtl.app_*).
 * A pure function of its arguments (no device). */
int fiasco_amd_spec_append_helpers(unsigned frames, int cus, int G, int wide_build);
/* which of `shares` device shares of the process takes a job (a pure function, no device): share_key == 0 -> the job's
 * index in the call, round robin (frames of a batch); share_key = key + 1 -> key mod shares whatever the
 * index and however many jobs the call holds.  The sequence engine keys the frames of a video and the decodes of their
 * reference frames by their group of pictures, so a GOP never changes its device (codec/coder.c:490-668 is the loop
 * that is sharded; the reference has one device: the host). */
unsigned fiasco_amd_share_of(unsigned share_key, unsigned index, unsigned shares);
int fiasco_amd_set_device(int device);
int fiasco_amd_set_devices(const int *ids, int n);      /* n = 0: back to the automatic choice */
int fiasco_amd_device_count(void);                      /* shares a batch is split into */

/* The launcher keeps the per-frame HBM slabs of finished calls in a process-wide pool
 * (hipMalloc of hundreds of MB per frame is slow); this returns the pool to the driver. */
void fiasco_amd_release_memory(void);

/* Self test of the one libm function the rate models need on both sides of the seam: double
 * log2 of a float probability (codec/coeff.c:232-237, codec/domain-pool.c:772,
 * codec/bintree.c:67).  Evaluates it on the device for EVERY float with a biased exponent in
 * [exp_lo, exp_hi] (1..127: all of (0, 1]) and compares with the host's libm bit for bit.
 * n_double: arguments whose double results differ; n_float: those whose (float) -log2 differ
 * too (first_bad = one of them).  Returns 1 when the run completed. */
int fiasco_amd_selftest_log2(unsigned exp_lo, unsigned exp_hi, unsigned long long *n_checked,
                             unsigned long long *n_double, unsigned long long *n_float,
                             float *first_bad);

/* The same comparison through the table the frame kernel consults for the arguments on which the
 * two libm's differ (built once per process and device, cached under $FIASCO_AMD_CACHE or
 * /tmp): n_double must be 0 -- the coefficient prices, which subtract these doubles from a float
 * sum (codec/coeff.c:228-236), are then the host's for every probability that can occur.
 * n_entries = size of the table. */
/* largest distance, in ulps, between a device and a host logarithm seen by the comparisons so far */
unsigned long long fiasco_amd_selftest_log2_max_ulp(void);
int fiasco_amd_selftest_log2_patched(unsigned exp_lo, unsigned exp_hi, unsigned long long *n_checked,
                                     unsigned long long *n_double, unsigned long long *n_entries);

#ifdef __cplusplus
}
#endif
#endif
