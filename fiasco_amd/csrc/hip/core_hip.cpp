/*
 *  core_hip.cpp -- host launcher of the device frame coder: implements the seam of
 *  fa_host.h (fa_core_stage / fa_core_run / fa_core_unstage / fa_core_encode_frames) on
 *  top of the HIP runtime.
 *
 *  stage : carve one HBM slab per frame (layout: frame_coder.h) from a process-wide slab
 *          pool, upload the int16 pixel plane; the basis automaton travels inside the
 *          DevFrame descriptor.  After stage the inputs are resident in HBM.
 *  run   : ONE persistent kernel launch with one workgroup per staged frame (all frames in
 *          flight at once); every frame packs its finished automaton into a per-launch
 *          buffer, which comes down with ONE device->host copy on a second stream (double
 *          buffered: the next launch does not wait for it).  Frames whose state capacity
 *          guess was too small are re-staged with a larger slab and relaunched.
 *  There is no CPU fallback: without a usable GPU every job fails with an error message.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <pthread.h>
#include <utility>
#include <vector>
#include "fa_host.h"
#include "frame_coder.h"
#include "libfiasco_amd_hip.h"

/* Developer and test switches (FIASCO_AMD_SPEC, _NO_WIDE, _FORCE_TRI, _CAP_GUESS, _QUEUE_SLABS, _TRACE, ...)
 * are honoured only when FIASCO_AMD_DEBUG is set to something other than 0: a drop-in library must not
 * change its behaviour because of a stray variable in a user's environment.  What a user may set without
 * it: FIASCO_AMD_CACHE (cache directory), FIASCO_AMD_NO_LOG2_TABLE (encode without the log2 correction
 * table), FIASCO_AMD_DEVICES (devices the batch entries spread their frames over, end of this file). */
extern "C" const char *fa_knob(const char *name)
{
    const char *d = getenv("FIASCO_AMD_DEBUG");
    return d && *d && strcmp(d, "0") != 0 ? getenv(name) : nullptr;
}


/* ------------------------------------------------------------------ per-device state
 *
 * Everything the launcher keeps between calls belongs to ONE device: the pool of slabs, the log2
 * correction table, the counters.  A process that encodes on one device (the default on a 1-GPU box, a
 * rank of the multi-process harness after fiasco_amd_set_device()) uses g_state0 from whatever thread
 * calls in.  The multi-device entries at the end of this file give every further device a DevState of its
 * own and run its share of a batch on a host thread whose t_dev points there. */
struct PoolEntry { char *base; size_t bytes; int device; };   /* device: where hipMalloc gave the slab out (checked on every acquire) */
struct Log2Patch { unsigned *d_keys = nullptr; double *d_vals = nullptr; unsigned mask = 0; int device = -1;
                   unsigned long long entries = 0; bool tried = false, ok = false; };
struct DevState {
    fiasco_amd_stats stats;
    std::vector<PoolEntry> free;
    Log2Patch l2;
    char l2_err[200];
    DevState() { memset(&stats, 0, sizeof stats); l2_err[0] = 0; }
};
static DevState g_state0;
static thread_local DevState *t_dev = &g_state0;
#define g_stats  (t_dev->stats)
#define g_free   (t_dev->free)
#define g_l2     (t_dev->l2)
#define g_l2_err (t_dev->l2_err)

extern "C" void fc_launch(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                       const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream);
extern "C" void fc_launch_big(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                       const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream);
extern "C" void fc_launch_wide(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                       const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream);
extern "C" void fc_launch_wide_tri(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                       const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream);
extern "C" void fc_launch_big_wide(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                       const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream);
/* the 512-thread big build with coefficient models of up to 512 symbols per context (frame_coder.h FC_HM) */
extern "C" void fc_launch_big_hm(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                       const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream);
/* ... and with the other models of the reference's registries (frame_coder.h FC_GM) */
extern "C" void fc_launch_big_gm(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                       const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream);

/* block-level speculation (frame_coder.h, FcSpecCtl): n frames with G workgroups each */
extern "C" void fc_launch_spec(DevFrame *d_frames, DevFrame *d_vframes, unsigned n, unsigned G, unsigned H, hipStream_t stream);
extern "C" unsigned fc_spec_slot_bytes(void);
extern "C" unsigned fc_spec_ctl_bytes(void);
extern "C" int fc_occupancy_spec(void);
extern "C" void fc_launch_spec_wide(DevFrame *d_frames, DevFrame *d_vframes, unsigned n, unsigned G, unsigned H, hipStream_t stream);
extern "C" unsigned fc_spec_slot_bytes_wide(void);

/* which of the two kernel builds (frame_coder.hip) encodes a job: the default build covers
 * the CLI's -z 0 geometry, the big one block levels 4..12, up to 5 vectors and the
 * second-domain retry */
/* A basis the rows inside DevFrame cannot hold: more than FC_MAXBASIS states, or a label with more than
 * MAXEDGES edges (data/medium.fco, large.fco: the reference's append_edge runs on into the next row,
 * fa_wfa_append_edge).  It travels as the memory image of its rows (DevFrame.bx); big kernel builds only. */
static bool long_basis(const fa_wfa *w)
{
    if (!w) return false;
    if (w->basis_states > FC_MAXBASIS) return true;
    for (unsigned s = 0; s < w->basis_states; s++)
        for (unsigned l = 0; l < 2; l++) {
            unsigned e = 0;
            while (e < 6 && FA_INTO(w, s, l, e) != FA_NO_EDGE) e++;
            if (e > FA_MAXEDGES) return true;
        }
    return false;
}
static size_t bx_bytes(const fa_wfa *w) { return 16 + (size_t) w->basis_states * 80 + 72; }

/* RPF mantissas of more than 5 bits (cfiasco --rpf-mantissa / --dc-rpf-mantissa 6 .. 8): the FC_HM build */
static bool needs_hm_variant(const fa_cparams *cp)
{
    return cp->rpf.mantissa_bits > 5 || cp->dc_rpf.mantissa_bits > 5 || cp->d_rpf.mantissa_bits > 5 || cp->d_dc_rpf.mantissa_bits > 5;
}

/* models other than the `rle' pools and the `adaptive' coefficients that fiasco.h can ask for
 * (fiasco_amd_c_options_set_models; the delta set counts when it is used: prediction, P / B frames): the FC_GM build */
static bool needs_gm_variant(const fa_job *job)
{
    const fa_cparams *cp = &job->cp;
    const bool delta_used = cp->prediction || job->frame_type != FA_I_FRAME;
    return cp->pool_kind != FA_POOL_RLE || cp->coeff_kind != FA_COEFF_ADAPTIVE
           || (delta_used && (cp->d_pool_kind != FA_POOL_RLE || cp->d_coeff_kind != FA_COEFF_ADAPTIVE));
}

static bool needs_big_variant(const fa_cparams *cp, const fa_wfa *basis)
{
    if (needs_hm_variant(cp)) return true;
    if (cp->prediction) return true;         /* second model set, residual search: big build only */
    if (long_basis(basis)) return true;
    /* the default build reads 3 edge slots per label (frame_coder.hip FC_MAXE): a basis file
     * whose states have more goes to the big build */
    if (basis)
        for (unsigned s = 0; s < basis->basis_states; s++)
            for (unsigned l = 0; l < 2; l++)
                for (unsigned e = 0; e < 6 && FA_INTO(basis, s, l, e) != FA_NO_EDGE; e++)
                    if (e >= 3) return true;
    unsigned dcs = 1u << (1 + cp->dc_rpf.mantissa_bits), sy = 1u << (1 + cp->rpf.mantissa_bits);
    return cp->lc_min_level <= cp->images_level || cp->lc_max_level > 10 || cp->max_elements > 3
           || cp->second_domain_block || cp->check_for_underflow || cp->check_for_overflow || cp->full_search
           || (cp->lc_max_level - cp->lc_min_level + 1) * sy + dcs > FC_MAXCOEFF
           /* aac snapshots beyond the default build's LDS pool (frame_coder.hip SNAP_POOL16; one per
            * depth + one per block level with children): the big build parks them in HBM */
           || (cp->level - cp->lc_min_level + 3 + cp->lc_max_level - cp->lc_min_level)
              * ((32 + 2 * ((cp->lc_max_level - cp->lc_min_level + 1) * sy + dcs) + 15) / 16) > FC_SNAP16_WIDE;
}

/* The 256-thread default build keeps a shorter stack and smaller snapshot pools in LDS than the
 * wide one (frame_coder.hip: FC_MAXDEPTH_NARROW, FC_SNAP16_NARROW, FC_SNAPTM_NARROW -- sized
 * for what the stock reference accepts, level <= 22); a frame beyond them is given to the
 * wide build whatever the size of the launch. */
static bool needs_wide_variant(const fa_cparams *cp)
{
    unsigned dcs = 1u << (1 + cp->dc_rpf.mantissa_bits), sy = 1u << (1 + cp->rpf.mantissa_bits);
    const unsigned n16 = (32 + 2 * ((cp->lc_max_level - cp->lc_min_level + 1) * sy + dcs) + 15) / 16;
    const unsigned depths = cp->level - cp->lc_min_level + 3;
    return depths - 1 > FC_MAXDEPTH_NARROW
           || (depths + cp->lc_max_level - cp->lc_min_level) * n16 > FC_SNAP16_NARROW
           || depths * 4 * ((2 * cp->limit_level + 3) / 4) > FC_SNAPTM_NARROW;
}

/* Capacity memory.  The first guess of a frame's state capacity (fa_core_stage) is a formula of the frame size;
 * a frame that needs more is searched again with 1.5 x the capacity -- and so would be every later frame of the
 * same kind: the P frames of a 720p colour sequence with --prediction need 1.5 x what their I frames need, and
 * each was searched twice (BASELINE config 5: 20 launches for 10 frames).  So the process remembers, per kind of
 * frame (size, colour, frame type class, block levels, price), the largest need it has seen, and the guess starts
 * there.  The capacity is memory layout only: streams do not depend on it. */
struct CapHint { unsigned long long key; int needP, needPA; };
static pthread_mutex_t g_hint_mu = PTHREAD_MUTEX_INITIALIZER;
static CapHint g_hints[64];
static unsigned g_hint_n, g_hint_next;
static unsigned long long cap_key(const fa_job *job)
{
    const fa_cparams *cp = &job->cp;
    unsigned pb;
    memcpy(&pb, &cp->price, 4);
    const unsigned v[] = { job->image->width, job->image->height, (unsigned) (job->image->color != 0),
                           (unsigned) (job->frame_type != FA_I_FRAME), (unsigned) (cp->prediction != 0), cp->lc_min_level,
                           cp->lc_max_level, cp->p_min_level, cp->p_max_level, cp->max_elements, pb, cp->limit_states };
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof v / sizeof v[0]; i++)
        for (int b = 0; b < 4; b++) { h ^= (v[i] >> (8 * b)) & 0xff; h *= 1099511628211ull; }
    return h ? h : 1;
}
static void cap_hint_get(const fa_job *job, int *needP, int *needPA)
{
    const unsigned long long k = cap_key(job);
    *needP = *needPA = 0;
    pthread_mutex_lock(&g_hint_mu);
    for (unsigned i = 0; i < g_hint_n; i++)
        if (g_hints[i].key == k) { *needP = g_hints[i].needP; *needPA = g_hints[i].needPA; break; }
    pthread_mutex_unlock(&g_hint_mu);
}
static void cap_hint_put(const fa_job *job, int needP, int needPA)
{
    const unsigned long long k = cap_key(job);
    pthread_mutex_lock(&g_hint_mu);
    unsigned i = 0;
    for (; i < g_hint_n; i++) if (g_hints[i].key == k) break;
    if (i == g_hint_n) {
        if (g_hint_n < sizeof g_hints / sizeof g_hints[0]) g_hint_n++;
        else i = g_hint_next++ % (sizeof g_hints / sizeof g_hints[0]);       /* full: round robin */
        g_hints[i].key = k; g_hints[i].needP = g_hints[i].needPA = 0;
    }
    if (needP > g_hints[i].needP) g_hints[i].needP = needP;
    if (needPA > g_hints[i].needPA) g_hints[i].needPA = needPA;
    pthread_mutex_unlock(&g_hint_mu);
}

/* (fiasco_amd_get_stats / _reset_stats: with the multi-device entries, end of file) */
static int spec_policy(size_t frames, int cus, bool big_frames, bool narrow_only, int occ);
/* workgroups per frame for the table passes of big frames (frame_coder.h FcCoop): the frames are launched in groups
 * of eight (XCD placement), all workgroups must be resident at one per CU */
static unsigned coop_policy(size_t frames, int cus)
{
    const size_t padded = (frames + 7) / 8 * 8;
    if (padded * 8 <= (size_t) cus) return 8;
    if (padded * 4 <= (size_t) cus) return 4;
    if (padded * 2 <= (size_t) cus) return 2;
    return 1;
}
extern "C" unsigned fiasco_amd_coop_workgroups(unsigned frames, int cus) { return coop_policy(frames, cus); }
/* the dealing function of the device shares (fa_host.h), for the tests: job -> share is a function of the key alone */
extern "C" unsigned fiasco_amd_share_of(unsigned share_key, unsigned index, unsigned shares) { return fa_share_of(share_key, index, shares); }
extern "C" int fiasco_amd_spec_workgroups(unsigned frames, int cus, int big_frames, int narrow_only, int occupancy)
{
    return spec_policy(frames, cus, big_frames != 0, narrow_only != 0, occupancy);
}
extern "C" const char *fa_core_name(void) { return "hip-gfx950"; }

/* workgroups (= frames) of a kernel build that one CU holds at once, as the runtime computes it
 * from the build's registers and LDS (frame_coder.hip FC_OCCUPANCY) */
extern "C" int fc_occupancy(void);
extern "C" int fc_occupancy_wide(void);
extern "C" int fc_occupancy_big(void);
extern "C" int fc_occupancy_big_wide(void);
static size_t frames_per_cu(bool big, bool wide)
{
    static int cache[4] = { 0, 0, 0, 0 };
    const int i = (big ? 2 : 0) + (wide ? 1 : 0);
    if (!cache[i]) {
        cache[i] = i == 0 ? fc_occupancy() : i == 1 ? fc_occupancy_wide() : i == 2 ? fc_occupancy_big() : fc_occupancy_big_wide();
        if (cache[i] < 1) cache[i] = 1;
    }
    return (size_t) cache[i];
}

extern "C" void fiasco_amd_release_memory(void);

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* ------------------------------------------------------------------ log2 self test
 *
 * The rate models price symbols with double log2 of a float probability p = count / (float)
 * total (codec/coeff.c:232-237, codec/domain-pool.c:772, codec/bintree.c:67).  The host side
 * of the reference evaluates it with glibc, the device with ROCm's ocml; bit parity of the
 * streams needs both to return the same DOUBLE for every argument that can occur.  Every such
 * argument is a float in (0, 1] (and 1 - p is one in [0, 1)), so the claim can be checked
 * exhaustively: this entry evaluates log2((double) p) on the device for all floats of an
 * exponent range and compares the doubles bit for bit with glibc's on the host. */
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

__global__ void selftest_log2_kernel(unsigned first_bits, unsigned n, double *out, const unsigned *keys,
                                     const double *vals, unsigned mask)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = first_bits + i;
    double v = log2((double) __uint_as_float(k));
    if (keys)                                   /* the frame kernel's log2_host(), mp_device.inc */
        for (unsigned h = (k * 2654435761u) & mask, kk; (kk = keys[h]) != 0; h = (h + 1) & mask)
            if (kk == k) { v = vals[h]; break; }
    out[i] = v;
}

struct L2Task { const double *dev; unsigned first_bits, n, t, nt; unsigned long long dd, df; unsigned bad;
                unsigned long long max_ulp;
                std::vector<std::pair<unsigned, double>> *collect; };

static void *l2_thread(void *arg)
{
    L2Task *k = (L2Task *) arg;
    for (unsigned i = k->t; i < k->n; i += k->nt) {
        unsigned bits = k->first_bits + i;
        float p; memcpy(&p, &bits, 4);
        double h = log2((double) p), d = k->dev[i];
        if (memcmp(&h, &d, 8) != 0) {
            long long hb, db;
            memcpy(&hb, &h, 8); memcpy(&db, &d, 8);
            unsigned long long dist = (unsigned long long) (hb > db ? hb - db : db - hb);
            if (dist > k->max_ulp) k->max_ulp = dist;
            k->dd++;
            if (k->collect) k->collect->push_back(std::make_pair(bits, h));
            if ((float) -h != (float) -d) { if (!k->df) k->bad = bits; k->df++; }
        }
    }
    return nullptr;
}

/* the table of host log2 values the kernels use (DevFrame.l2_*), per process */
static unsigned long long g_l2_max_ulp;      /* largest distance seen by the last comparisons, in ulps */
extern "C" unsigned long long fiasco_amd_selftest_log2_max_ulp(void) { return g_l2_max_ulp; }

/* compare over the floats with biased exponent in [exp_lo, exp_hi]; with `use_table` the device
 * side goes through the patch table like the frame kernel does; `collect` gathers the
 * arguments that differ together with the host's value */
static int log2_compare(unsigned exp_lo, unsigned exp_hi, bool use_table, unsigned long long *n_checked,
                        unsigned long long *n_double, unsigned long long *n_float, float *first_bad,
                        std::vector<std::pair<unsigned, double>> *collect)
{
    const unsigned CH = 1u << 23;                  /* one binade per launch */
    double *d_out = nullptr, *h_out = nullptr;
    unsigned long long checked = 0, dd = 0, df = 0;
    unsigned bad = 0;
    if (exp_lo < 1) exp_lo = 1;
    if (exp_hi > 127) exp_hi = 127;
    if (hipMalloc((void **) &d_out, (size_t) CH * 8) != hipSuccess
        || hipHostMalloc((void **) &h_out, (size_t) CH * 8, hipHostMallocDefault) != hipSuccess) {
        fa_set_error("selftest: HIP error: %s", hipGetErrorString(hipGetLastError()));
        if (d_out) (void) hipFree(d_out);
        return 0;
    }
    for (unsigned e = exp_lo; e <= exp_hi; e++) {
        const unsigned first = e << 23;
        const unsigned n = e == 127 ? 1u : CH;     /* 1.0 is the largest probability */
        hipLaunchKernelGGL(selftest_log2_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, first, n, d_out,
                           use_table ? g_l2.d_keys : nullptr, use_table ? g_l2.d_vals : nullptr, g_l2.mask);
        if (hipMemcpy(h_out, d_out, (size_t) n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
            fa_set_error("selftest: HIP error: %s", hipGetErrorString(hipGetLastError()));
            (void) hipFree(d_out); (void) hipHostFree(h_out);
            return 0;
        }
        enum { NT = 16 };
        pthread_t th[NT];
        L2Task task[NT];
        std::vector<std::pair<unsigned, double>> part[NT];
        int started[NT] = { 0 };
        for (unsigned t = 0; t < NT; t++) {
            task[t] = L2Task{ h_out, first, n, t, NT, 0, 0, 0, 0, collect ? &part[t] : nullptr };
            if (t) started[t] = pthread_create(&th[t], nullptr, l2_thread, &task[t]) == 0;
        }
        l2_thread(&task[0]);
        for (unsigned t = 1; t < NT; t++) { if (started[t]) pthread_join(th[t], nullptr); else l2_thread(&task[t]); }
        for (unsigned t = 0; t < NT; t++) {
            dd += task[t].dd;
            if (task[t].max_ulp > g_l2_max_ulp) g_l2_max_ulp = task[t].max_ulp;
            if (task[t].df && !df) bad = task[t].bad;
            df += task[t].df;
            if (collect) collect->insert(collect->end(), part[t].begin(), part[t].end());
        }
        checked += n;
    }
    (void) hipFree(d_out); (void) hipHostFree(h_out);
    if (n_checked) *n_checked = checked;
    if (n_double) *n_double = dd;
    if (n_float) *n_float = df;
    if (first_bad) memcpy(first_bad, &bad, 4);
    return 1;
}

/* floats with biased exponent in [exp_lo, exp_hi] (126 = [0.5, 1)); returns 1 when the run
 * completed.  n_double / n_float: arguments whose double result / whose (float) -log2 differ. */
extern "C" int fiasco_amd_selftest_log2(unsigned exp_lo, unsigned exp_hi, unsigned long long *n_checked,
                                        unsigned long long *n_double, unsigned long long *n_float,
                                        float *first_bad)
{
    return log2_compare(exp_lo, exp_hi, false, n_checked, n_double, n_float, first_bad, nullptr);
}

static bool log2_patch_build(void);

/* the same comparison THROUGH the table the frame kernel uses: n_double must come out 0 */
extern "C" int fiasco_amd_selftest_log2_patched(unsigned exp_lo, unsigned exp_hi, unsigned long long *n_checked,
                                                unsigned long long *n_double, unsigned long long *n_entries)
{
    if (!log2_patch_build()) { fa_set_error("selftest: no log2 table on this device: %s", g_l2_err); return 0; }
    if (n_entries) *n_entries = g_l2.entries;
    if (!g_l2.d_keys && g_l2.entries) { fa_set_error("selftest: no log2 table on this device"); return 0; }
    return log2_compare(exp_lo, exp_hi, g_l2.d_keys != nullptr, n_checked, n_double, nullptr, nullptr, nullptr);
}

/* Build (or load from the cache file) the table of this process: every float in (0, 1] whose
 * device log2 differs from the host's, with the host's value.  About a second of work the first
 * time on a box; the list (some 12 MB) is then kept in a per-user cache directory -- $FIASCO_AMD_CACHE,
 * else $XDG_CACHE_HOME/fiasco_amd, else $HOME/.cache/fiasco_amd, /tmp only as the last resort -- under a
 * name that carries the host libm's and the device's answers to a few probe arguments.  A cache file is
 * taken only if its checksum fits and EVERY stored value is what this host's log2 computes now.
 *
 * Returns false -- with the reason in g_l2_err -- when the table is needed but could not be built or
 * brought onto the device: 1 018 853 arguments differ between ocml and glibc, frames coded without
 * the table could differ from the reference's streams, so fa_core_stage() fails them instead
 * (FIASCO_AMD_NO_LOG2_TABLE=1 runs without the table on purpose). */

static unsigned long long fnv64(const void *p, size_t n, unsigned long long h = 1469598103934665603ull)
{
    const unsigned char *c = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 1099511628211ull;
    return h;
}

static void mkdir_p(const char *dir)
{
    char tmp[512];
    snprintf(tmp, sizeof tmp, "%s", dir);
    for (char *q = tmp + 1; *q; q++)
        if (*q == '/') { *q = 0; (void) mkdir(tmp, 0700); *q = '/'; }
    (void) mkdir(tmp, 0700);
}

/* cache directory of this user; created if need be */
static void l2_cache_dir(char *out, size_t n)
{
    const char *e;
    if ((e = getenv("FIASCO_AMD_CACHE")) && *e) snprintf(out, n, "%s", e);
    else if ((e = getenv("XDG_CACHE_HOME")) && *e) snprintf(out, n, "%s/fiasco_amd", e);
    else if ((e = getenv("HOME")) && *e) snprintf(out, n, "%s/.cache/fiasco_amd", e);
    else snprintf(out, n, "/tmp");
    mkdir_p(out);
    if (access(out, W_OK) != 0) snprintf(out, n, "/tmp");
}

static bool log2_patch_build(void)
{
    int dev = 0;
    g_l2_err[0] = 0;
    if (hipGetDevice(&dev) != hipSuccess) { snprintf(g_l2_err, sizeof g_l2_err, "no current HIP device"); return false; }
    if (g_l2.tried && g_l2.device == dev) {
        if (!g_l2.ok) snprintf(g_l2_err, sizeof g_l2_err, "an earlier attempt on this device failed");
        return g_l2.ok;
    }
    if (g_l2.d_keys) { (void) hipFree(g_l2.d_keys); (void) hipFree(g_l2.d_vals); g_l2 = Log2Patch(); }
    g_l2.tried = true; g_l2.device = dev; g_l2.ok = false;
    if (getenv("FIASCO_AMD_NO_LOG2_TABLE")) { g_l2.ok = true; return true; }
    if (fa_knob("FIASCO_AMD_FAIL_LOG2_TABLE")) {                    /* tests: what a failed build looks like */
        snprintf(g_l2_err, sizeof g_l2_err, "failure requested by FIASCO_AMD_FAIL_LOG2_TABLE");
        return false;
    }
    std::vector<std::pair<unsigned, double>> list;
    char path[600];
    {
        /* fingerprint: host libm on a few awkward arguments + device name */
        hipDeviceProp_t prop;
        unsigned long long fp = 1469598103934665603ull;
        const float probe[] = { 0.3f, 1.0f / 3, 0.7f, 5.0f / 7, 0.0123f, 0.999f, 1e-3f, 0.57f };
        for (unsigned i = 0; i < sizeof probe / sizeof probe[0]; i++) {
            double v = log2((double) probe[i]);
            unsigned long long b; memcpy(&b, &v, 8);
            fp = (fp ^ b) * 1099511628211ull;
        }
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess)
            for (const char *c = prop.gcnArchName; *c; c++) fp = (fp ^ (unsigned char) *c) * 1099511628211ull;
        int rt = 0; (void) hipRuntimeGetVersion(&rt);
        fp = (fp ^ (unsigned) rt) * 1099511628211ull;
        char dir[512];
        l2_cache_dir(dir, sizeof dir);
        snprintf(path, sizeof path, "%s/fiasco_amd_log2_%016llx.bin", dir, fp);
    }
    bool loaded = false;
    if (FILE *f = fopen(path, "rb")) {
        /* magic, entries, FNV-1a of the payload */
        unsigned long long hdr[3] = { 0, 0, 0 };
        if (fread(hdr, 8, 3, f) == 3 && hdr[0] == 0x33474f4c41464full && hdr[1] < (1ull << 26)) {
            list.resize((size_t) hdr[1]);
            loaded = fread(list.data(), sizeof list[0], list.size(), f) == list.size()
                     && fgetc(f) == EOF
                     && fnv64(list.data(), list.size() * sizeof list[0]) == hdr[2];
            /* every stored value must still be what this host computes (a second of log2 calls
             * at most: cheap next to trusting a file somebody else could have written) */
            for (size_t i = 0; loaded && i < list.size(); i++) {
                float pf; memcpy(&pf, &list[i].first, 4);
                double v = log2((double) pf);
                if (!(pf > 0.0f && pf <= 1.0f) || memcmp(&v, &list[i].second, 8) != 0) loaded = false;
            }
        }
        fclose(f);
        if (!loaded) list.clear();
    }
    if (!loaded) {
        unsigned long long nd = 0;
        if (!log2_compare(1, 127, false, nullptr, &nd, nullptr, nullptr, &list)) {
            snprintf(g_l2_err, sizeof g_l2_err, "the comparison of the device's log2 with the host's did not run (%s)",
                     hipGetErrorString(hipGetLastError()));
            return false;
        }
        char tmp[640];
        snprintf(tmp, sizeof tmp, "%s.%d", path, (int) getpid());
        int fd = open(tmp, O_WRONLY | O_CREAT | O_EXCL, 0600);
        if (FILE *f = fd >= 0 ? fdopen(fd, "wb") : nullptr) {
            unsigned long long hdr[3] = { 0x33474f4c41464full, (unsigned long long) list.size(),
                                          fnv64(list.data(), list.size() * sizeof list[0]) };
            bool ok = fwrite(hdr, 8, 3, f) == 3 && fwrite(list.data(), sizeof list[0], list.size(), f) == list.size();
            ok = fclose(f) == 0 && ok;
            if (!ok || rename(tmp, path) != 0) (void) remove(tmp);      /* no cache: built again next time */
        } else if (fd >= 0) close(fd);
    }
    g_l2.entries = list.size();
    if (list.empty()) { g_l2.ok = true; return true; }       /* the two logarithms agree everywhere: nothing to correct */
    unsigned slots = 1024;
    while (slots < 4 * list.size()) slots <<= 1;
    std::vector<unsigned> keys(slots, 0u);
    std::vector<double> vals(slots, 0.0);
    for (size_t i = 0; i < list.size(); i++) {
        unsigned h = (list[i].first * 2654435761u) & (slots - 1);
        while (keys[h]) h = (h + 1) & (slots - 1);
        keys[h] = list[i].first; vals[h] = list[i].second;
    }
    hipError_t e;
    if ((e = hipMalloc((void **) &g_l2.d_keys, (size_t) slots * 4)) != hipSuccess
        || (e = hipMalloc((void **) &g_l2.d_vals, (size_t) slots * 8)) != hipSuccess
        || (e = hipMemcpy(g_l2.d_keys, keys.data(), (size_t) slots * 4, hipMemcpyHostToDevice)) != hipSuccess
        || (e = hipMemcpy(g_l2.d_vals, vals.data(), (size_t) slots * 8, hipMemcpyHostToDevice)) != hipSuccess) {
        (void) hipGetLastError();
        if (g_l2.d_keys) (void) hipFree(g_l2.d_keys);
        if (g_l2.d_vals) (void) hipFree(g_l2.d_vals);
        g_l2.d_keys = nullptr; g_l2.d_vals = nullptr;
        snprintf(g_l2_err, sizeof g_l2_err, "%llu corrections could not be brought onto the device (%s)",
                 (unsigned long long) list.size(), hipGetErrorString(e));
        return false;
    }
    g_l2.mask = slots - 1;
    g_l2.ok = true;
    return true;
}

/* ------------------------------------------------------------------ slab pool */


/* the current device of the calling thread (the share's, bind_share); -1 when HIP cannot tell */
static int pool_device(void)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { (void) hipGetLastError(); d = -1; }
    return d;
}

static char *slab_acquire(size_t bytes, size_t *got)
{
    /* A slab never meets another device: the pool belongs to a share (t_dev), a share to a device.  An entry whose
     * tag says otherwise (a device list changed under a pool, a share bound to the wrong device) is a bug of the
     * launcher -- such an entry is not handed out, the call allocates afresh and says so. */
    const int here = pool_device();
    size_t best = g_free.size();
    for (size_t i = 0; i < g_free.size(); i++)
        if (g_free[i].device != here && g_free[i].device >= 0 && here >= 0) {
            static bool told = false;
            if (!told) { told = true; fprintf(stderr, "libfiasco_amd: slab pool entry of device %d met device %d (not used)\n", g_free[i].device, here); }
        }
    for (size_t i = 0; i < g_free.size(); i++)
        if ((g_free[i].device == here || g_free[i].device < 0 || here < 0)
            && g_free[i].bytes >= bytes && g_free[i].bytes <= bytes + bytes / 4
            && (best == g_free.size() || g_free[i].bytes < g_free[best].bytes))
            best = i;
    if (best != g_free.size()) {
        char *p = g_free[best].base;
        *got = g_free[best].bytes;
        g_free.erase(g_free.begin() + (long) best);
        return p;
    }
    char *p = nullptr;
    size_t free_b = 0, total_b = 0;
    /* leave a reserve for the launch's own buffers (descriptors, packed automata, uploads) */
    const size_t reserve = (size_t) 768 << 20;
    bool fits = hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= bytes + reserve;
    if (!fits || hipMalloc((void **) &p, bytes) != hipSuccess) {
        /* pool may be holding slabs of other sizes: drop them and try once more */
        (void) hipGetLastError();
        if (g_free.empty()) return nullptr;
        for (size_t i = 0; i < g_free.size(); i++) (void) hipFree(g_free[i].base);
        g_free.clear();
        fits = hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= bytes + reserve;
        if (!fits || hipMalloc((void **) &p, bytes) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    }
    *got = bytes;
    return p;
}

static void slab_release(char *p, size_t bytes)
{
    if (p) g_free.push_back(PoolEntry{p, bytes, pool_device()});
}

/* ------------------------------------------------------------------ layout */

struct Layout {
    size_t gram, gcol, diag, ipis, d5, d4, img, imgT, imgT4, norms, num, den, est, ipdo, used, tree, into, weight,
           final_d, level_of_state, domain_type, x, y, ycol, pool_states, pos, hits, ycol0, snap, pix16, total;
    size_t ipis_alt, d5_alt, d4_alt, pix_save, sv_gram, sv_img, sv_auto;   /* prediction only */
    size_t mv, past, future, mc_fwd, mc_bwd, pix_chroma;                    /* P frames only */
    size_t coop;                                                            /* FcCoop: header + the block's pixels */
    size_t bx;                                                              /* DevFrame.bx: rows of a long basis */
    size_t gq, lginv;                                                       /* FC_GM build: DevFrame.gq, DevFrame.lginv */
    int    max_save;
};

/* P: capacity for states with tables; PA >= P: capacity of the automaton arrays (chroma
 * states of a colour frame never own tables) */
static Layout make_layout(int P, int PA, int NL, int NS, int NA, int NI, int il, int low, size_t npix,
                          int max_save, int inter, int plevels, int color, bool tri, bool hm, int gm_states = 0)
{
    Layout L;
    memset(&L, 0, sizeof L);             /* compared with memcmp (frame queue) */
    size_t o = 0;
    L.max_save = max_save;
#define CARVE(field, bytes) do { L.field = o; o = align_up(o + (bytes), 256); } while (0)
    /* Gram tables: full symmetric, or (tri) the lower triangle with packed rows + a row of slack */
    CARVE(gram, tri ? (size_t) NL * ((size_t) P * (P + 1) / 2 + P) * 4 : (size_t) NL * P * P * 4);
    CARVE(gcol, tri ? (size_t) NL * FC_TRI_HOT * P * 4 : 0);      /* columns of the first states as rows (frame_coder.h) */
    CARVE(diag, (size_t) NL * P * 4);
    CARVE(ipis, (size_t) NS * P * 4);
    CARVE(d5, (size_t) NA * P * 4);
    CARVE(d4, low ? (size_t) 2 * NA * P * 4 : 0);
    CARVE(img, (size_t) P * NI * 4);
    CARVE(imgT, ((size_t) 1 << il) * P * 4);
    CARVE(imgT4, low ? ((size_t) 1 << (il - 1)) * P * 4 : 0);
    CARVE(norms, (size_t) NS * 4);
    CARVE(num, (size_t) P * 4);
    CARVE(den, (size_t) P * 4);
    CARVE(est, (size_t) P * 4);
    CARVE(ipdo, (size_t) FC_MAXED * P * 4);
    CARVE(used, (size_t) P);
    /* tree .. y are downloaded with ONE copy: keep them adjacent */
    CARVE(tree, (size_t) 2 * PA * 2);
    CARVE(into, (size_t) 12 * PA * 2);
    CARVE(weight, (size_t) 12 * PA * 4);
    CARVE(final_d, (size_t) PA * 4);
    CARVE(level_of_state, (size_t) PA);
    CARVE(domain_type, (size_t) PA);
    CARVE(x, (size_t) 2 * PA * 2);
    CARVE(y, (size_t) 2 * PA * 2);
    CARVE(ycol, (size_t) 2 * PA);
    CARVE(mv, inter ? (size_t) 10 * PA * 2 : 0);     /* downloaded with the automaton */
    CARVE(pool_states, (size_t) (P + 8) * 2);
    CARVE(pos, (size_t) (PA + 8) * 2);
    CARVE(hits, (size_t) (PA + 8) * 4);
    CARVE(ycol0, (size_t) 2 * PA);                   /* initial y_column flags (colour streams) */
    /* model snapshots of the big build: aac [depth][slots][n16] x 16 bytes, with prediction 5
     * slots per depth and the tree-model snapshots [depth][2][28] behind them */
    const size_t n16max = FC_N16(hm ? FC_MAXCOEFF_HM : FC_MAXCOEFF_BIG_STD);        /* the kernel build's FC_N16MAX */
    CARVE(snap, max_save ? (size_t) (FC_MAXDEPTH_BIG * 5 * n16max + FC_MAXDEPTH_BIG * 2 * 28) * 16
                         : (size_t) 26 * 2 * n16max * 16);
    /* prediction: second table set for residual blocks, block pixels + norms, displaced rows */
    CARVE(ipis_alt, max_save ? (size_t) NS * P * 4 : 0);
    CARVE(d5_alt, max_save ? (size_t) NA * P * 4 : 0);
    CARVE(d4_alt, max_save && low ? (size_t) 2 * NA * P * 4 : 0);
    CARVE(pix_save, max_save ? (size_t) (4096 + 128) * 4 : 0);
    CARVE(sv_gram, (size_t) max_save * NL * P * 4);
    CARVE(sv_img, (size_t) max_save * (NI + 48 + NL) * 4);
    CARVE(sv_auto, (size_t) max_save * sizeof(FcSavedRow));
    /* P frames: reference frame planes, displacement cost tables, private chroma planes */
    CARVE(past, inter ? npix * 2 : 0);
    CARVE(future, inter == 2 ? npix * 2 : 0);
    CARVE(mc_fwd, inter ? (size_t) plevels * 1024 * 4 : 0);
    CARVE(mc_bwd, inter ? (size_t) plevels * 1024 * 4 : 0);
    CARVE(pix_chroma, inter && color ? npix / 3 * 2 * 2 : 0);
    CARVE(coop, FC_COOP_HDR + ((size_t) (NS + 1) << il) * 4);
    CARVE(bx, FC_BX_BYTES);
    CARVE(gq, gm_states ? (size_t) FC_GQ_SLOTS * P * 2 : 0);
    CARVE(lginv, gm_states ? ((size_t) gm_states + 2) * 8 : 0);
    CARVE(pix16, npix * 2);
#undef CARVE
    L.total = o;
    return L;
}

static int device_supported(const fa_job *job, char *why, size_t n)
{
    const fa_cparams *cp = &job->cp;
    /* a P / B frame whose reference frame is missing (e.g. an I frame coded between a B frame and its past reference:
     * the reference coder drops both references at an I frame, codec/coder.c:580-628, and dereferences a null frame at
     * its first motion search) -- IF a motion search can happen: the searches run on ranges of the prediction window
     * (codec/prediction.c:96-208), and a minimum block level that a colour frame has ratcheted above the window's top
     * (codec/coder.c:785-797) leaves no such range.  The reference then codes the frame without ever looking at the
     * missing frame, and so does the device (tests/test_gpu_fuzz_reference.py, seed 71136: pattern `ipb', 4 frames). */
    const bool window_reachable = (int) cp->p_max_level >= (int) cp->lc_min_level;
    if (job->frame_type != FA_I_FRAME && window_reachable
        && (!job->past || (job->frame_type == FA_B_FRAME && !job->future) || cp->search_range != 16)) {
        snprintf(why, n, "Motion search without a reference frame (frame pattern).");
        return 0;
    }
    /* (signed: after a colour frame the minimum block level may have been ratcheted ABOVE the prediction window --
     * codec/coder.c:785-797 -- which then holds no level at all; found by tests/test_gpu_fuzz_reference.py, seed 71064:
     * the unsigned difference refused a P frame the reference codes) */
    if ((cp->prediction || job->frame_type != FA_I_FRAME) && (int) cp->p_max_level - (int) cp->lc_min_level + 1 > 9) {
        snprintf(why, n, "prediction over more than 9 block levels is not supported by the device coder (levels %u .. %u, frame type %d)",
                 cp->lc_min_level, cp->p_max_level, job->frame_type);
        return 0;
    }
    if (cp->images_level != 5 || cp->lc_min_level < 4) {
        snprintf(why, n, "device coder needs images_level 5 and min block level >= 4");
        return 0;
    }
    if (cp->lc_max_level > 12) { snprintf(why, n, "max block level > 12 is not supported by the device coder"); return 0; }
    if (cp->max_elements > 5) { snprintf(why, n, "more than 5 vectors per block are not supported by the device coder"); return 0; }
    {
        unsigned dcs = 1u << (1 + cp->dc_rpf.mantissa_bits), sy = 1u << (1 + cp->rpf.mantissa_bits);
        if ((cp->lc_max_level - cp->lc_min_level + 1) * sy + dcs > FC_MAXCOEFF_HM) {
            snprintf(why, n, "coefficient model too large for the device coder "
                             "(block levels x mantissa symbols > %d)", FC_MAXCOEFF_HM);
            return 0;
        }
    }
    if (cp->rpf.mantissa_bits > 8 || cp->dc_rpf.mantissa_bits > 8 || cp->d_rpf.mantissa_bits > 8 || cp->d_dc_rpf.mantissa_bits > 8) {
        snprintf(why, n, "RPF mantissa > 8 bits is not supported by the device coder");      /* (alloc_rpf never makes one) */
        return 0;
    }
    if (long_basis(job->wfa)) {
        const fa_wfa *w = job->wfa;
        if (bx_bytes(w) > FC_BX_BYTES) { snprintf(why, n, "initial basis too large for the device coder"); return 0; }
        /* every edge list must end inside the rows of the basis (the device takes a copy of those rows; a list that
         * ran on into the rows of the coder's own states would change while the frame is coded) */
        for (unsigned r = 0; r < w->basis_states * 2; r++) {
            unsigned e = r * 6;
            while (e < w->basis_states * 12 && w->into[e] != FA_NO_EDGE) e++;
            if (e >= w->basis_states * 12) { snprintf(why, n, "edge lists of the initial basis run on into the coder's states"); return 0; }
        }
    }
    /* (every entry of the reference's model registries runs on the device since round 5: the `rle' pools and the
     * `adaptive' coefficients of fiasco.h in the fast builds, the others in the FC_GM build) */
    return 1;
}

/* ------------------------------------------------------------------ staging */

struct FrameSlot {
    int      job;            /* index into jobs[] */
    char    *base = nullptr;
    size_t   bytes = 0;
    int      P = 0, PA = 0;
    int      floorP = 0, floorPA = 0;   /* what frames of this kind needed before (capacity memory) */
    Layout   L;
    DevFrame F;
    bool     staged = false, done = false, big = false, rejected = false;
    bool     hm = false;         /* coefficient models of more than 64 symbols per context: the FC_HM kernel build */
    bool     gm = false;         /* models beyond rle / adaptive: the FC_GM kernel build */
    std::vector<double> lginv_host;      /* upload source of DevFrame.lginv */
    bool     wide_only = false;  /* default geometry, but beyond the 256-thread build's LDS pools */
    bool     tri = false;        /* triangular Gram tables (half the slab; the wide_tri build of the kernel) */
    bool     borrow = false;     /* no slab of its own: encoded in the slab of a queue workgroup */
    bool     spec = false;       /* several workgroups per frame (FC_SPEC build): the slab's capacity holds the
                                  * verifiers' state-id ranges */
    std::vector<uint8_t> ycol_host;      /* upload source of ycol0, alive until the slot goes */
    std::vector<int32_t> bx_host;        /* ... of DevFrame.bx */
    const int16_t *ext_pix = nullptr;    /* pixel planes outside the slab (fa_core_upload_commit) */
    const int16_t *ext_next = nullptr;   /* ... of the frames the NEXT pass encodes */
    /* the host image this pass encodes, as of its submit: fiasco_amd_batch_upload() may point
     * job->image at the NEXT pass's frames while this one is still running, and a re-stage of the
     * running pass (capacity regrow, later wave) must not read those */
    const fa_image *src = nullptr;
};

struct Staged;
static inline const fa_image *slot_image(const Staged *S, const FrameSlot &fs);

struct Staged {
    unsigned n = 0;
    fa_job  *jobs = nullptr;
    std::vector<FrameSlot> slots;
    DevFrame *d_frames = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ok = false;
    char err[200] = "";
    int  ncu = 256;                /* CUs of the device the batch was staged for */
    /* launch in flight (fa_core_submit .. fa_core_finish) */
    std::vector<size_t>   batch;
    std::vector<DevFrame> hf;
    FcTrace *d_trace = nullptr;
    bool inflight = false, launch_failed = false, broken = false;
    int  good = 0;
    std::vector<std::pair<size_t, size_t>> to_unpack;   /* (slot, offset in pinned) */
    char  *pinned = nullptr;       /* host staging buffer for the automaton downloads */
    size_t pinned_bytes = 0;
    /* packed automata of a launch (DevFrame.pack_dst), double buffered: launch i + 1 writes the
     * other buffer while the copy of launch i is still on its way to the host */
    char  *d_pack[2] = { nullptr, nullptr };
    size_t d_pack_bytes[2] = { 0, 0 }, pack_need = 0;
    int    parity = 0;
    bool   packed = false, copy_pending = false;
    std::vector<size_t> pack_off;
    hipStream_t cstream = nullptr;
    /* replacement inputs (fa_core_upload_buffer / _commit): pinned host staging memory, two
     * device buffers used alternately (the running pass reads one, the upload fills the
     * other), a stream of their own and the event the next launch waits for */
    char  *up_host = nullptr;
    size_t up_host_bytes = 0;
    bool   up_host_shared = false;   /* the buffer belongs to the batch of several shares (MultiStaged): not freed here */
    char  *up_dev[2] = { nullptr, nullptr };
    size_t up_dev_bytes[2] = { 0, 0 };
    int    up_parity = 0;
    bool   up_pending = false;
    hipStream_t ustream = nullptr;
    hipEvent_t  ev_up = nullptr;
    /* frame queue (frame_coder.hip, FC_KERNEL): frames beyond the number of resident workgroups
     * (or beyond what HBM holds in slabs) borrow the slab of whichever workgroup takes them */
    int       lender0 = -1;        /* first slot with a slab of the queue's layout */
    Layout    qL;                  /* that layout and capacity (the slot itself may be re-staged larger) */
    int       qP = 0, qPA = 0;
    bool      qbig = false, qtri = false;   /* kernel build of the queue's frames */
    size_t    lenders = 0, borrowers = 0, lender_cap = 0;
    char     *qpix = nullptr;      /* pixel planes of the borrowers */
    size_t    qpix_bytes = 0, qpix_used = 0;
    unsigned long long *d_ring = nullptr;   /* free slabs in the order they were handed back, per kernel build */
    size_t    ring_n = 0;
    unsigned *d_queue = nullptr;   /* two counters per kernel build: tickets taken, slabs handed back */
    unsigned *d_ptrmask = nullptr;
    bool      ptrmask_ready = false;
    bool      no_coop = false, no_coop_done = false;
    FcCoop    coop_hdr;               /* what a launch writes over the control blocks of its frames (source of async copies) */        /* a frame's helper workgroups did not answer (FC_ERR_COOP): one workgroup per frame from here on */
    /* block-level speculation: workgroups per frame (0 = off), the descriptors of the verifier
     * workgroups, and one buffer with -- per frame -- control block + checkpoint slots, then the
     * verifiers' private tables */
    int       specG = 0;
    int       specH[2] = { 0, 0 };    /* append helpers per frame of the launch in flight, per workgroup width (FcSpecCtl.app_*) */
    bool      no_app = false;         /* the append helpers of a frame did not answer (FC_ERR_COOP): none from here on */
    DevFrame *d_vframes = nullptr;
    size_t    vframes_n = 0;
    char     *d_spec = nullptr;
    size_t    d_spec_bytes = 0, spec_ctl_span = 0;
    std::vector<size_t> spec_frames;       /* batch positions of the speculating frames of the launch in flight */
    size_t    spec_first[2] = { 0, 0 }, spec_n[2] = { 0, 0 };    /* ... per workgroup width (256, 1024 threads) */
};

/* A launch that leaves workgroup slots of the chip free gives its frames several workgroups each
 * (frame_coder.h, FcSpecCtl).  Which frames: gray intra frames of the default geometry whose state
 * capacity -- with room for the verifiers' id ranges -- still fits the 256-thread build.
 * FIASCO_AMD_SPEC=0 switches it off, FIASCO_AMD_SPEC=<G> asks for G workgroups per frame. */
/* of the G workgroups of a frame: the chain, T table workers, G - 1 - T verifiers */
static int spec_workers(int G)
{
    const char *e = fa_knob("FIASCO_AMD_SPEC_T");           /* experiments */
    if (e && atoi(e) >= 0 && atoi(e) < G - 1) return atoi(e);
    return G >= 6 ? 2 : G >= 4 ? 1 : 0;
}

/* the blocks of the largest block level in the order the partition search visits them
 * (codec/subdivide.c:277-290: the children of a node, first label first; invisible ranges are skipped, :118-120) */
static void spec_block_list(const DevFrame &F, std::vector<uint16_t> &out)
{
    struct Node { int level, x, y; };
    std::vector<Node> stack;
    stack.push_back(Node{F.level, 0, 0});
    out.clear();
    while (!stack.empty()) {
        const Node n = stack.back();
        stack.pop_back();
        if (n.x >= F.width || n.y >= F.height) continue;
        if (n.level == F.lc_max) { out.push_back((uint16_t) n.x); out.push_back((uint16_t) n.y); continue; }
        if (n.level < F.lc_max) continue;
        const int l1 = n.level - 1;
        const int w1 = 1 << (l1 >> 1), h1 = 1 << ((l1 + 1) >> 1);
        /* second child first onto the stack: the first is visited first */
        if (n.level & 1) { stack.push_back(Node{l1, n.x, n.y + h1}); stack.push_back(Node{l1, n.x, n.y}); }
        else             { stack.push_back(Node{l1, n.x + w1, n.y}); stack.push_back(Node{l1, n.x, n.y}); }
    }
}

static int spec_policy(size_t frames, int cus, bool big_frames, bool narrow_only, int occ);

/* Append helpers per frame (frame_coder.h FcSpecCtl.app_*): further workgroups of a speculating frame that build their
 * shares of every Gram row the chain appends.  For the 1024-thread speculating build (frames beyond 3072 states: 4K),
 * whose launches give a frame a CU per workgroup and leave the rest of the chip empty -- BASELINE config 4 as written
 * puts 8 frames on a GPU: 8 x 8 workgroups on 256 CUs -- and whose rows are long (up to 10 passes of the 1024 lanes).
 * Three where the chip has CUs left for them; fewer than 2 are not worth the hand-off.  A function of its arguments
 * alone (fiasco_amd_spec_append_helpers); FIASCO_AMD_SPEC_APP=<H> (tests, experiments) asks for H. */
static int spec_app_policy(size_t frames, int cus, int G, bool wide_build, int occ)
{
    if (!frames || G < 2 || cus < 1) return 0;
    if (occ < 1) occ = 1;
    const size_t room = (size_t) cus * (size_t) occ / frames;       /* workgroups per frame that can be resident */
    size_t H = room > (size_t) G ? room - (size_t) G : 0;
    const char *e = fa_knob("FIASCO_AMD_SPEC_APP");
    if (e) { const size_t want = (size_t) (atoi(e) > 0 ? atoi(e) : 0); return (int) (want < H ? want : H); }
    if (!wide_build) {
        /* the 256-thread build (rows of up to 3072 entries, 12 passes of the lanes): three helpers while the launch
         * stays below 1.5 workgroups per CU -- 1080p: 1 frame 0.367 -> 0.343 s, 16 frames 39.4 -> 42.7, 32 frames 75 -> 80
         * frames/s; 64 and 128 frames (CUs shared by three and more workgroups): nothing, not given */
        return H >= 3 && 2 * frames * ((size_t) G + 3) <= 3 * (size_t) cus ? 3 : 0;
    }
    /* measured (8 x 4K, round 6): 2, 3, 5 and 7 helpers give the same 1.58 .. 1.62 s against 1.87 without -- the hand-off
     * (two releases, two acquires per row) is what a dealt row costs, not the shares; 16 frames 8.5 -> 9.6 frames/s with 3
     * (helpers are light: they may use the half of the chip the frames' own workgroups leave alone, spec_policy) */
    if (H > 3) H = 3;
    return H >= 2 ? (int) H : 0;
}
extern "C" int fiasco_amd_spec_append_helpers(unsigned frames, int cus, int G, int wide_build)
{
    return spec_app_policy(frames, cus, G, wide_build != 0, wide_build ? 1 : 4);      /* the builds' workgroups per CU */
}

static int spec_groups(size_t frames, int cus, bool big_frames, bool narrow_only)
{
    const char *e = fa_knob("FIASCO_AMD_SPEC");
    if (e && atoi(e) <= 1) return 0;
    if (fa_knob("FIASCO_AMD_TRACE") || fa_knob("FIASCO_AMD_NO_WIDE") || fa_knob("FIASCO_AMD_FORCE_TRI")) return 0;
    int occ = fc_occupancy_spec();
    if (occ < 1) occ = 1;
    if (!frames) return 0;
    if (e) {                                  /* as asked, if the chip holds that many workgroups at once */
        size_t G = (size_t) cus * (size_t) occ / frames;
        if ((size_t) atoi(e) < G) G = (size_t) atoi(e);
        if (G > FC_SPEC_MAXG) G = FC_SPEC_MAXG;
        return G >= 2 ? (int) G : 0;
    }
    return spec_policy(frames, cus, big_frames, narrow_only, occ);
}

/* workgroups per frame of a launch (0: one, no speculation): a function of its arguments alone
 * (fiasco_amd_spec_workgroups, include/libfiasco_amd_hip.h) */
static int spec_policy(size_t frames, int cus, bool big_frames, bool narrow_only, int occ)
{
    if (!frames || cus < 1) return 0;
    /* a CU per workgroup while the frames leave that many, at most FC_SPEC_MAXG; at least two verifiers per chain (one
     * keeps it waiting: slower than no speculation at all) */
    size_t G = (size_t) cus / frames;
    if (G > FC_SPEC_MAXG) G = FC_SPEC_MAXG;
    /* The 256-thread build shares CUs (four workgroups each): as many workgroups per frame as are resident, but not
     * more than five once the launch passes 2.5 workgroups per CU.  Round 6 (hand-offs with one releasing lane;
     * tests/gpu_spec_policy_sweep.sh, 1080p frames/s by workgroups per frame):
     *   frames      3      4      5      6      8     one workgroup each
     *     48       61     76     99     98    104      39
     *     64       81    101    128    127    132      52
     *     96      114    144    186    181    177      78
     *    128      148    189    228    231    203     103
     *    192      195    250    299      (5 is what fits)  154
     *    256      245    275      (4 is what fits)         206
     * (until round 5, when every lane fenced at every hand-off: 5 / 4 / 3 for 64 / 96 / 256 frames -- 113, 132, 213.) */
    if (narrow_only && !big_frames && occ >= 2) {
        G = (size_t) cus * (size_t) occ / frames;
        if (G > FC_SPEC_MAXG) G = FC_SPEC_MAXG;
        if (G > 5 && 2 * frames * G > 5 * (size_t) cus) {
            G = 5 * (size_t) cus / (2 * frames);
            if (G < 5) G = 5;
        }
    }
    /* (Until round 5 4K frames were kept to half the CUs -- 32 frames: 7.2 frames/s with 8 workgroups each, 9.1 with 4:
     * every lane of every workgroup fenced at each hand-off and the L2 write-backs slowed everybody down.  With one
     * releasing lane per hand-off, round 6: 32 frames 10.6 with 4, 15.2 with 6, 15.4 with 8; 24 frames 9.4 -> 13.4.) */
    (void) big_frames;
    return G >= 3 ? (int) G : 0;
}

static inline const fa_image *slot_image(const Staged *S, const FrameSlot &fs)
{
    return fs.src ? fs.src : S->jobs[fs.job].image;
}

static void fill_frame(FrameSlot &fs, const fa_job *job)
{
    const fa_cparams *cp = &job->cp;
    const fa_wfa *w = job->wfa;
    DevFrame &F = fs.F;
    const Layout &L = fs.L;
    char *base = fs.base;
    memset(&F, 0, sizeof F);
    F.price = cp->price;
    F.lc_min = (int) cp->lc_min_level; F.lc_max = (int) cp->lc_max_level;
    F.images_level = (int) cp->images_level; F.max_elements = (int) cp->max_elements;
    const bool bxl = long_basis(w);
    {
        unsigned live = cp->max_elements;
        for (unsigned st = 0; st < (bxl ? 0u : w->basis_states); st++)
            for (unsigned l = 0; l < 2; l++) {
                unsigned e = 0;
                while (e < 6 && FA_INTO(w, st, l, e) != FA_NO_EDGE) e++;
                if (e > live) live = e;
            }
        F.maxe_live = (int) live;
    }
    F.level = (int) cp->level; F.width = (int) job->image->width; F.height = (int) job->image->height;
    F.pool_max = (int) cp->pool_max_states; F.limit_states = (int) cp->limit_states;
    F.ML = (int) cp->limit_level;
    F.rpf_mant = (int) cp->rpf.mantissa_bits; F.dc_mant = (int) cp->dc_rpf.mantissa_bits;
    F.rpf_range = cp->rpf.range; F.dc_range = cp->dc_rpf.range;
    F.P = fs.P; F.PA = fs.PA;
    F.gram_ls = fs.tri ? (unsigned) ((size_t) fs.P * (fs.P + 1) / 2 + fs.P) : (unsigned) fs.P * (unsigned) fs.P;
    F.color = job->image->color ? 1 : 0;
    /* pools whose chroma list is not cut down (uniform, rle-no-chroma ...: FC_GM build) search every state: full tables */
    F.chroma_cl_cap = fa_knob("FIASCO_AMD_CLMAX") ? atoi(fa_knob("FIASCO_AMD_CLMAX")) : 0;     /* tests: the overflow path of Sh::cl */
    F.chroma_sparse = !fa_knob("FIASCO_AMD_CHROMA_FULL") && (cp->pool_kind == FA_POOL_RLE || cp->pool_kind == FA_POOL_ADAPTIVE || cp->pool_kind == FA_POOL_BASIS);
    F.chroma_max = (int) cp->chroma_max_states;
    F.chroma_decrease = cp->chroma_decrease;
    F.plane = (unsigned long long) job->image->width * job->image->height;
    F.gl0 = (int) (cp->lc_min_level < cp->images_level ? cp->lc_min_level : cp->images_level);
    F.NL = (int) cp->lc_max_level - F.gl0 + 1;
    F.second_domain_block = cp->second_domain_block ? 1 : 0;
    F.check_underflow = cp->check_for_underflow ? 1 : 0;
    F.check_overflow = cp->check_for_overflow ? 1 : 0;
    F.full_search = cp->full_search ? 1 : 0;
    F.NS = (int) fa_size_of_tree(cp->products_level);
    F.NA = 1 << (cp->lc_max_level - cp->images_level);
    F.NI = (int) fa_size_of_tree(cp->images_level);
    F.dcs = 1 << (1 + F.dc_mant); F.sy = 1 << (1 + F.rpf_mant);
    F.coeff_size = (F.lc_max - F.lc_min + 1) * F.sy + F.dcs;
    F.coeff_nt = F.lc_max - F.lc_min + 2;
    F.basis_states = (int) w->basis_states;
    F.bx = bxl ? (const int *) (base + L.bx) : nullptr;
    F.gm_pool[0] = (int) cp->pool_kind; F.gm_pool[1] = (int) cp->d_pool_kind;
    F.gm_coeff[0] = (int) cp->coeff_kind; F.gm_coeff[1] = (int) cp->d_coeff_kind;
    F.gq = fs.gm ? (int16_t *) (base + L.gq) : nullptr;
    F.lginv = fs.gm ? (const double *) (base + L.lginv) : nullptr;
    for (unsigned s = 0; s < (bxl ? 0u : w->basis_states); s++) {
        F.b_final[s] = w->final_distribution[s];
        F.b_dtype[s] = w->domain_type[s];
        for (int l = 0; l < 2; l++) {
            F.b_tree[s][l] = FA_TREE(w, s, l);
            for (int e = 0; e < 6; e++) {
                F.b_into[s][l][e] = FA_INTO(w, s, l, e);
                F.b_weight[s][l][e] = FA_WEIGHT(w, s, l, e);
                if (FA_INTO(w, s, l, e) == FA_NO_EDGE) break;
            }
        }
    }
    F.pix16 = (const int16_t *) (base + L.pix16);
    F.gram = (float *) (base + L.gram); F.diag = (float *) (base + L.diag);
    F.ipis = (float *) (base + L.ipis); F.d5 = (float *) (base + L.d5);
    F.gcol = (float *) (base + L.gcol);
    F.d4 = (float *) (base + L.d4); F.imgT4 = (float *) (base + L.imgT4);
    F.img = (float *) (base + L.img); F.imgT = (float *) (base + L.imgT);
    F.norms = (float *) (base + L.norms);
    F.num = (float *) (base + L.num); F.den = (float *) (base + L.den);
    F.est = (float *) (base + L.est); F.ipdo = (float *) (base + L.ipdo);
    F.used = (uint8_t *) (base + L.used);
    F.tree = (int16_t *) (base + L.tree); F.into = (int16_t *) (base + L.into);
    F.weight = (float *) (base + L.weight); F.final_d = (float *) (base + L.final_d);
    F.level_of_state = (uint8_t *) (base + L.level_of_state);
    F.domain_type = (uint8_t *) (base + L.domain_type);
    F.x = (uint16_t *) (base + L.x); F.y = (uint16_t *) (base + L.y);
    F.ycol = (uint8_t *) (base + L.ycol);
    F.ycol0 = job->ycol_carry ? (const uint8_t *) (base + L.ycol0) : nullptr;
    F.pool_states = (int16_t *) (base + L.pool_states);
    F.pos = (int16_t *) (base + L.pos);
    F.hits = (int *) (base + L.hits);
    F.snap_hbm = fs.big ? (void *) (base + L.snap) : nullptr;
    F.l2_keys = g_l2.d_keys; F.l2_vals = g_l2.d_vals; F.l2_mask = g_l2.mask;
    /* prediction (codec/coder.c:716-745): gray frames try it from the root; a colour frame only
     * gets the second rle pool (intra prediction is never asked for its bands, :805-806) */
    const int inter = job->frame_type != FA_I_FRAME;
    F.pred_on = cp->prediction || inter ? 1 : 0;
    F.pred_root = job->image->color ? inter : (cp->prediction || inter ? 1 : 0);
    F.search_range = (int) cp->search_range;
    F.mv = (int16_t *) (base + L.mv);
    F.past = (const int16_t *) (base + L.past); F.future = (const int16_t *) (base + L.future);
    F.coop = (FcCoop *) (base + L.coop);
    F.mc_fwd = (float *) (base + L.mc_fwd); F.mc_bwd = inter ? (float *) (base + L.mc_bwd) : nullptr;
    F.pix_chroma = (int16_t *) (base + L.pix_chroma);
    F.frame_type = job->frame_type;
    F.p_min = (int) cp->p_min_level; F.p_max = (int) cp->p_max_level;
    F.d_rpf_mant = (int) cp->d_rpf.mantissa_bits; F.d_dc_mant = (int) cp->d_dc_rpf.mantissa_bits;
    F.d_rpf_range = cp->d_rpf.range; F.d_dc_range = cp->d_dc_rpf.range;
    F.d_dcs = 1 << (1 + F.d_dc_mant); F.d_sy = 1 << (1 + F.d_rpf_mant);
    F.d_coeff_size = (F.lc_max - F.lc_min + 1) * F.d_sy + F.d_dcs;
    F.ipis_alt = (float *) (base + L.ipis_alt); F.d5_alt = (float *) (base + L.d5_alt);
    F.d4_alt = (float *) (base + L.d4_alt); F.pix_save = (float *) (base + L.pix_save);
    F.sv_gram = (float *) (base + L.sv_gram); F.sv_img = (float *) (base + L.sv_img);
    F.sv_auto = (FcSavedRow *) (base + L.sv_auto);
    F.max_save = L.max_save;
    F.slab_base = base; F.slab_bytes = L.total;
}

/* allocate the slab of one frame for capacity fs.P and upload its pixel plane */
static void slot_layout(Staged *S, FrameSlot &fs)
{
    const fa_job *job = &S->jobs[fs.job];
    const fa_cparams *cp = &job->cp;
    int il = (int) cp->images_level;
    int low = cp->lc_min_level < cp->images_level;
    int NL = (int) (cp->lc_max_level - (low ? cp->lc_min_level : cp->images_level) + 1);
    int NS = (int) fa_size_of_tree(cp->products_level);
    int NA = 1 << (cp->lc_max_level - cp->images_level);
    int NI = (int) fa_size_of_tree(cp->images_level);
    size_t npix = (size_t) job->image->width * job->image->height;
    const int bands = job->image->color ? 3 : 1;
    int max_save = 0;
    const int inter = job->frame_type;
    if (cp->prediction || inter) {
        int span = (int) cp->p_max_level - (int) cp->lc_min_level + 1;
        max_save = 1 << (span < 1 ? 1 : span > 9 ? 9 : span);
    }
    fs.L = make_layout(fs.P, fs.PA, NL, NS, NA, NI, il, low, npix * bands, max_save, inter,
                       (int) cp->p_max_level - (int) cp->p_min_level + 1, job->image->color ? 1 : 0, fs.tri, fs.hm || fs.gm,
                       fs.gm ? (int) cp->limit_states : 0);
}

/* ---- frame queue: which frames may share slabs ---- */

static bool queue_eligible(const Staged *S, const FrameSlot &fs)
{
    const fa_job *job = &S->jobs[fs.job];
    /* inputs of P/B frames and the carried y_column of a colour stream live inside the slab */
    return job->frame_type == FA_I_FRAME && !job->ycol_carry && !fs.spec && !long_basis(job->wfa) && !fs.hm && !fs.gm && !fa_knob("FIASCO_AMD_NO_QUEUE");
}

/* same geometry, capacity and coder parameters as the queue's first frame: any of its slabs fits */
static bool queue_layout(const Staged *S, const FrameSlot &fs)
{
    if (S->lender0 < 0) return false;
    return fs.P == S->qP && fs.PA == S->qPA && fs.big == S->qbig && fs.tri == S->qtri
           && memcmp(&fs.L, &S->qL, sizeof(Layout)) == 0;
}

static void fill_frame(FrameSlot &fs, const fa_job *job);

/* a frame without a slab: descriptor laid out for the slab of the queue's first frame (the
 * workgroup that takes it re-bases the pointers), pixel planes in the queue's pixel buffer */
static int stage_borrower(Staged *S, FrameSlot &fs, size_t frames_left)
{
    fa_job *job = &S->jobs[fs.job];
    const FrameSlot &ref = S->slots[S->lender0];
    const size_t npix = (size_t) job->image->width * job->image->height;
    const int bands = job->image->color ? 3 : 1;
    const size_t need = align_up(npix * bands * 2, 256);
    if (!S->qpix) {
        size_t bytes = need * frames_left;
        if (hipMalloc((void **) &S->qpix, bytes) != hipSuccess) { S->qpix = nullptr; (void) hipGetLastError(); return 0; }
        S->qpix_bytes = bytes; S->qpix_used = 0;
    }
    if (S->qpix_used + need > S->qpix_bytes) return 0;
    fs.base = ref.base;                       /* the layout reference, not an owned slab */
    fill_frame(fs, job);
    fs.base = nullptr; fs.bytes = 0;
    fs.borrow = true;
    fs.ext_pix = (const int16_t *) (S->qpix + S->qpix_used);
    fs.F.pix16 = fs.ext_pix;
    for (int b = 0; b < bands; b++)
        if (hipMemcpyAsync(S->qpix + S->qpix_used + (size_t) b * npix * 2, slot_image(S, fs)->pixels[b], npix * 2,
                           hipMemcpyHostToDevice, S->stream) != hipSuccess) {
            snprintf(job->errmsg, sizeof job->errmsg, "HIP error: pixel upload failed");
            fs.borrow = false; fs.ext_pix = nullptr;
            return 0;
        }
    S->qpix_used += need;
    S->borrowers++;
    fs.staged = true;
    return 1;
}

static int stage_slot(Staged *S, FrameSlot &fs)
{
    fa_job *job = &S->jobs[fs.job];
    const fa_cparams *cp = &job->cp;
    int il = (int) cp->images_level;
    int low = cp->lc_min_level < cp->images_level;
    int NL = (int) (cp->lc_max_level - (low ? cp->lc_min_level : cp->images_level) + 1);
    int NS = (int) fa_size_of_tree(cp->products_level);
    int NA = 1 << (cp->lc_max_level - cp->images_level);
    int NI = (int) fa_size_of_tree(cp->images_level);
    size_t npix = (size_t) job->image->width * job->image->height;
    const int bands = job->image->color ? 3 : 1;
    /* states a prediction attempt can displace: the nodes of a subtree from the largest
     * predicted level down to the smallest block level */
    int max_save = 0;
    const int inter = job->frame_type;                 /* 0 I, 1 P, 2 B */
    if (cp->prediction || inter) {
        int span = (int) cp->p_max_level - (int) cp->lc_min_level + 1;
        max_save = 1 << (span < 1 ? 1 : span > 9 ? 9 : span);
    }
    fs.L = make_layout(fs.P, fs.PA, NL, NS, NA, NI, il, low, npix * bands, max_save, inter,
                       (int) cp->p_max_level - (int) cp->p_min_level + 1, job->image->color ? 1 : 0, fs.tri, fs.hm || fs.gm,
                       fs.gm ? (int) cp->limit_states : 0);
    fs.base = slab_acquire(fs.L.total, &fs.bytes);
    /* developer aid: FIASCO_AMD_POISON=<byte> fills the slab first -- the kernel must write every
     * cell before it reads it, whatever an earlier frame left there */
    if (fs.base && fa_knob("FIASCO_AMD_POISON"))
        (void) hipMemsetAsync(fs.base, atoi(fa_knob("FIASCO_AMD_POISON")), fs.L.total, S->stream);
    if (!fs.base) {
        snprintf(job->errmsg, sizeof job->errmsg, "out of HBM: frame needs %.2f GiB", fs.L.total / 1073741824.0);
        return 0;
    }
    fill_frame(fs, job);
    const bool hmx = fs.hm || fs.gm;              /* the FC_GM build has the FC_HM build's model sizes */
    const int maxsym = hmx ? FC_MAXSYM_HM : FC_MAXSYM_STD;
    if (fs.F.coeff_size > (hmx ? FC_MAXCOEFF_HM : fs.big ? FC_MAXCOEFF_BIG_STD : FC_MAXCOEFF) || fs.F.dcs > maxsym || fs.F.sy > maxsym
        || (fs.F.pred_on && (fs.F.d_coeff_size > (hmx ? FC_MAXCOEFF_HM : FC_MAXCOEFF_BIG_STD) || fs.F.d_dcs > maxsym || fs.F.d_sy > maxsym))
        || fs.F.ML > 26) {
        snprintf(job->errmsg, sizeof job->errmsg,
                 "coefficient model too large for the device coder (levels x mantissa symbols > %d)", hmx ? FC_MAXCOEFF_HM : FC_MAXCOEFF_BIG_STD);
        slab_release(fs.base, fs.bytes); fs.base = nullptr;
        fs.done = true; fs.rejected = true;      /* permanent: not a matter of free HBM */
        return 0;
    }
    if (fs.ext_pix) fs.F.pix16 = fs.ext_pix;       /* the planes live outside the slab already */
    for (int b = 0; b < bands && !fs.ext_pix; b++)
        if (hipMemcpyAsync(fs.base + fs.L.pix16 + (size_t) b * npix * 2, slot_image(S, fs)->pixels[b], npix * 2,
                           hipMemcpyHostToDevice, S->stream) != hipSuccess) {
            snprintf(job->errmsg, sizeof job->errmsg, "HIP error: pixel upload failed");
            slab_release(fs.base, fs.bytes); fs.base = nullptr;
            return 0;
        }
    /* reference frames the device decoder left on this device (fa_image.dev, frame_decoder.inc) are taken from
     * there; anything else comes from the host planes */
    int here = -1;
    if (hipGetDevice(&here) != hipSuccess) { (void) hipGetLastError(); here = -1; }
    if (job->frame_type != FA_I_FRAME && job->past)
        for (int b = 0; b < bands; b++)
            if ((job->past->dev && job->past->dev_id == here
                 ? hipMemcpyAsync(fs.base + fs.L.past + (size_t) b * npix * 2, (const int16_t *) job->past->dev + (size_t) b * npix, npix * 2,
                                  hipMemcpyDeviceToDevice, S->stream)
                 : hipMemcpyAsync(fs.base + fs.L.past + (size_t) b * npix * 2, job->past->pixels[b], npix * 2,
                                  hipMemcpyHostToDevice, S->stream)) != hipSuccess) {
                snprintf(job->errmsg, sizeof job->errmsg, "HIP error: reference frame upload failed");
                slab_release(fs.base, fs.bytes); fs.base = nullptr;
                return 0;
            }
    if (job->frame_type == FA_B_FRAME && job->future)
        for (int b = 0; b < bands; b++)
            if ((job->future->dev && job->future->dev_id == here
                 ? hipMemcpyAsync(fs.base + fs.L.future + (size_t) b * npix * 2, (const int16_t *) job->future->dev + (size_t) b * npix, npix * 2,
                                  hipMemcpyDeviceToDevice, S->stream)
                 : hipMemcpyAsync(fs.base + fs.L.future + (size_t) b * npix * 2, job->future->pixels[b], npix * 2,
                                  hipMemcpyHostToDevice, S->stream)) != hipSuccess) {
                snprintf(job->errmsg, sizeof job->errmsg, "HIP error: reference frame upload failed");
                slab_release(fs.base, fs.bytes); fs.base = nullptr;
                return 0;
            }
    if (job->ycol_carry) {                 /* [cap][2] on the host, [2][PA] on the device */
        const fa_wfa *w = job->wfa;
        fs.ycol_host.assign((size_t) 2 * fs.PA, 0);
        for (unsigned s = 0; s < w->cap && s < (unsigned) fs.PA; s++)
            for (int l = 0; l < 2; l++) fs.ycol_host[(size_t) l * fs.PA + s] = w->y_column[s * 2 + l];
        if (hipMemcpyAsync(fs.base + fs.L.ycol0, fs.ycol_host.data(), fs.ycol_host.size(),
                           hipMemcpyHostToDevice, S->stream) != hipSuccess) {
            snprintf(job->errmsg, sizeof job->errmsg, "HIP error: y_column upload failed");
            slab_release(fs.base, fs.bytes); fs.base = nullptr;
            return 0;
        }
    }
    if (fs.F.lginv) {                      /* log2 (1.0 / n) as THIS host's libm gives it: uniform_bits, codec/domain-pool.c:592-615 */
        const unsigned nmax = cp->limit_states + 1;
        fs.lginv_host.assign(nmax + 1, 0.0);
        for (unsigned k = 1; k <= nmax; k++) fs.lginv_host[k] = log2(1.0 / k);
        if (hipMemcpyAsync(fs.base + fs.L.lginv, fs.lginv_host.data(), fs.lginv_host.size() * 8, hipMemcpyHostToDevice, S->stream) != hipSuccess) {
            snprintf(job->errmsg, sizeof job->errmsg, "HIP error: table upload failed");
            slab_release(fs.base, fs.bytes); fs.base = nullptr;
            return 0;
        }
    }
    if (fs.F.bx) {                         /* the rows of a long basis, as they lie in the host's memory (DevFrame.bx) */
        const fa_wfa *w = job->wfa;
        const unsigned nb = w->basis_states, nr = 12 * nb + 12;
        fs.bx_host.assign((bx_bytes(w) + 3) / 4, 0);
        int32_t *b = fs.bx_host.data();
        b[0] = (int32_t) nb; b[1] = (int32_t) nr;
        memcpy(b + 4, w->final_distribution, (size_t) nb * 4);
        for (unsigned s = 0; s < nb; s++) b[4 + nb + s] = w->domain_type[s];
        float *bw = (float *) (b + 4 + 2 * nb);
        int16_t *bi = (int16_t *) (b + 4 + 2 * nb + nr);
        for (unsigned k = 0; k < nr; k++) { bi[k] = k < 12 * nb ? w->into[k] : (int16_t) FA_NO_EDGE; bw[k] = k < 12 * nb ? w->weight[k] : 0.0f; }
        if (hipMemcpyAsync(fs.base + fs.L.bx, b, fs.bx_host.size() * 4, hipMemcpyHostToDevice, S->stream) != hipSuccess) {
            snprintf(job->errmsg, sizeof job->errmsg, "HIP error: basis upload failed");
            slab_release(fs.base, fs.bytes); fs.base = nullptr;
            return 0;
        }
    }
    fs.staged = true;
    return 1;
}

static void core1_unstage(void *h)
{
    Staged *S = (Staged *) h;
    if (!S) return;
    if (S->inflight) {                       /* a submitted launch nobody collected */
        (void) hipStreamSynchronize(S->stream);
        if (S->d_trace) (void) hipFree(S->d_trace);
    }
    for (size_t k = 0; k < S->slots.size(); k++)
        if (S->slots[k].base) slab_release(S->slots[k].base, S->slots[k].bytes);
    if (S->d_frames) (void) hipFree(S->d_frames);
    if (S->qpix) (void) hipFree(S->qpix);
    if (S->d_ring) (void) hipFree(S->d_ring);
    if (S->d_queue) (void) hipFree(S->d_queue);
    if (S->d_ptrmask) (void) hipFree(S->d_ptrmask);
    if (S->d_vframes) (void) hipFree(S->d_vframes);
    if (S->d_spec) (void) hipFree(S->d_spec);
    if (S->cstream) { (void) hipStreamSynchronize(S->cstream); (void) hipStreamDestroy(S->cstream); }
    for (int i = 0; i < 2; i++) if (S->d_pack[i]) (void) hipFree(S->d_pack[i]);
    if (S->pinned) (void) hipHostFree(S->pinned);
    if (S->ustream) { (void) hipStreamSynchronize(S->ustream); (void) hipStreamDestroy(S->ustream); }
    if (S->ev_up) (void) hipEventDestroy(S->ev_up);
    if (S->up_host && !S->up_host_shared) (void) hipHostFree(S->up_host);
    for (int i = 0; i < 2; i++) if (S->up_dev[i]) (void) hipFree(S->up_dev[i]);
    if (S->ev0) (void) hipEventDestroy(S->ev0);
    if (S->ev1) (void) hipEventDestroy(S->ev1);
    if (S->stream) (void) hipStreamDestroy(S->stream);
    delete S;
}

static void *core1_stage(unsigned n, fa_job *jobs)
{
    Staged *S = new Staged;
    int ndev = 0;
    S->n = n; S->jobs = jobs;
    for (unsigned i = 0; i < n; i++) { jobs[i].status = 0; jobs[i].errmsg[0] = 0; }
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        for (unsigned i = 0; i < n; i++)
            snprintf(jobs[i].errmsg, sizeof jobs[i].errmsg,
                     "libfiasco_amd: no HIP device available (the hot path has no CPU fallback)");
        return S;
    }
    if (!log2_patch_build()) {               /* once per process and device */
        /* frames coded without it could differ from the reference's: fail them, loudly */
        for (unsigned i = 0; i < n; i++)
            snprintf(jobs[i].errmsg, sizeof jobs[i].errmsg,
                     "libfiasco_amd: no log2 correction table, streams could differ from the reference's "
                     "(FIASCO_AMD_NO_LOG2_TABLE=1 encodes without it): %s", g_l2_err);
        return S;                             /* S->ok stays false: nothing of this batch runs */
    }
    int specG = 0;
    {
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        /* every frame for the 256-thread build of the speculating kernel?  (several of its workgroups fit a CU) */
        bool narrow_only = n > 0;
        for (unsigned i = 0; i < n && narrow_only; i++) {
            const fa_cparams *cp = &jobs[i].cp;
            if (!jobs[i].image) { narrow_only = false; break; }
            const unsigned bw = fa_width_of_level(cp->lc_max_level), bh = fa_height_of_level(cp->lc_max_level);
            const size_t blocks = (size_t) ((jobs[i].image->width + bw - 1) / bw) * ((jobs[i].image->height + bh - 1) / bh);
            if (needs_wide_variant(cp) || blocks + blocks * 3 / 8 + 64 + FC_SPEC_MAXG * FC_SPEC_TEMPS > 3072) narrow_only = false;
        }
        specG = spec_groups(n, ncu, n > 0 && jobs[0].image && (jobs[0].image->width > 2048 || jobs[0].image->height > 2048), narrow_only);
        S->specG = specG; S->ncu = ncu;
    }
    if (hipStreamCreate(&S->stream) != hipSuccess || hipEventCreate(&S->ev0) != hipSuccess
        || hipEventCreate(&S->ev1) != hipSuccess
        || hipMalloc((void **) &S->d_frames, sizeof(DevFrame) * (n ? n : 1)) != hipSuccess) {
        for (unsigned i = 0; i < n; i++)
            snprintf(jobs[i].errmsg, sizeof jobs[i].errmsg, "HIP error: cannot create stream/events");
        return S;
    }
    for (unsigned i = 0; i < n; i++) {
        if (!device_supported(&jobs[i], jobs[i].errmsg, sizeof jobs[i].errmsg)) continue;
        /* first guess of the state capacity: one state per bintree node above the largest
         * block level (2 x #blocks) ... measured need at -q 20 is ~1.3 x #blocks */
        const fa_cparams *cp = &jobs[i].cp;
        unsigned bw = fa_width_of_level(cp->lc_max_level), bh = fa_height_of_level(cp->lc_max_level);
        size_t blocks = (size_t) ((jobs[i].image->width + bw - 1) / bw) * ((jobs[i].image->height + bh - 1) / bh);
        size_t guess = blocks + blocks * 3 / 8 + 64;
        /* predicted frames: the residual of a predicted block subdivides where the block itself would not --
         * 720p colour P frames with --prediction end with 2.0 .. 2.3 table states per block (config 5) */
        if (jobs[i].frame_type != FA_I_FRAME) guess = blocks * 5 / 2 + 64;
        /* tests / experiments: FIASCO_AMD_CAP_GUESS=<states> forces the first guess (a frame that
         * outgrows it is encoded again with 1.5 x the capacity, complete_wave) */
        if (fa_knob("FIASCO_AMD_CAP_GUESS") && atoi(fa_knob("FIASCO_AMD_CAP_GUESS")) > 0)
            guess = (size_t) atoi(fa_knob("FIASCO_AMD_CAP_GUESS"));
        /* what frames of this kind needed before (cap_hint_put): 1/16 on top, frames of a sequence drift */
        int hintP = 0, hintPA = 0;
        const bool forced = fa_knob("FIASCO_AMD_CAP_GUESS") && atoi(fa_knob("FIASCO_AMD_CAP_GUESS")) > 0;
        if (!forced && !fa_knob("FIASCO_AMD_NO_CAP_HINT")) cap_hint_get(&jobs[i], &hintP, &hintPA);
        if ((size_t) hintP + hintP / 16 + 32 > guess) guess = (size_t) hintP + hintP / 16 + 32;
        if (guess > cp->limit_states) guess = cp->limit_states;
        FrameSlot fs;
        fs.job = (int) i;
        fs.P = (int) align_up(guess, 64);
        fs.big = needs_big_variant(cp, jobs[i].wfa) || jobs[i].frame_type != FA_I_FRAME
                 /* a chroma dictionary of more than 63 states: the list scan of the big builds (mp_steps_list_global) */
                 || (jobs[i].image->color && cp->chroma_max_states > 63);
        fs.hm = needs_hm_variant(cp);
        fs.gm = needs_gm_variant(&jobs[i]) || fa_knob("FIASCO_AMD_FORCE_GM") != nullptr;     /* (tests: every frame through the FC_GM build) */
        if (fs.gm) fs.big = true;
        fs.wide_only = !fs.big && needs_wide_variant(cp);
                /* (experiments: FIASCO_AMD_SPEC_BUILD1 runs the speculating kernel build with ONE workgroup per frame) */
        const bool build1 = !specG && fa_knob("FIASCO_AMD_SPEC_BUILD1") != nullptr;
        if ((specG || build1) && !fs.big) {
            /* the 256-thread build up to 3072 states, the 1024-thread one (4K; frames beyond the narrow
             * build's LDS pools) up to 12288 */
            const size_t withids = align_up(guess + (build1 ? 0 : (size_t) (specG - 1 - spec_workers(specG)) * FC_SPEC_TEMPS), 64);
            if (withids <= 12 * 1024 && withids <= align_up(cp->limit_states, 64)) { fs.spec = true; fs.P = (int) withids; }
        }
        /* tests: the triangular layout (chosen below for HBM-bound batches) for every default-geometry frame */
        if (!fs.big && fa_knob("FIASCO_AMD_FORCE_TRI")) fs.tri = true;
        /* colour: the two chroma bands add auxiliary states (no tables) */
        size_t cap = align_up(cp->limit_states, 64);
        fs.PA = jobs[i].image->color ? (int) (3 * (size_t) fs.P > cap ? cap : 3 * (size_t) fs.P) : fs.P;
        if ((size_t) hintPA + hintPA / 16 + 32 > (size_t) fs.PA) {
            const size_t want = align_up((size_t) hintPA + hintPA / 16 + 32, 64);
            fs.PA = (int) (want > cap ? cap : want);
        }
        if (fs.PA < fs.P) fs.PA = fs.P;
        if (hintP) { fs.floorP = hintP + hintP / 16 + 32; fs.floorPA = hintPA + hintPA / 16 + 32; }
        S->slots.push_back(fs);
    }
    /* Stage the frames.  Every frame gets a slab of its own until the device is full -- as many
     * frames of one layout as the chip runs workgroups at once, or as HBM holds; the frames
     * after that join the FRAME QUEUE of that layout (no slab: whichever workgroup finishes its
     * frame takes the next one into its slab).  What can neither have a slab nor join the queue
     * is staged by run() as slabs free up. */
    int cus = 0;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    if (!S->slots.empty() && !fa_knob("FIASCO_AMD_NO_TIGHT")) {
        /* HBM-bound batches (4K: a slab is 3 GB, 97 % of it the Gram tables, quadratic in the state
         * capacity): when the slabs the chip could keep busy do not fit, the capacity guess drops
         * from 1.375 to 1.15 states per block of the largest block level -- a third more frames in
         * flight; a frame that outgrows it is encoded again with 1.5 x the capacity (complete_wave) */
        FrameSlot probe = S->slots[0];
        slot_layout(S, probe);
        size_t free_b = 0, total_b = 0, pooled = 0;
        for (size_t i = 0; i < g_free.size(); i++) pooled += g_free[i].bytes;
        size_t want = S->slots.size();
        const size_t resident = (size_t) cus * frames_per_cu(probe.big, probe.P > 12 * 256 || probe.wide_only || probe.hm || probe.gm);
        if (want > resident) want = resident;
        const bool hbm_bound = hipMemGetInfo(&free_b, &total_b) == hipSuccess && probe.L.total * want > free_b + pooled;
        if ((hbm_bound || S->slots.size() > resident) && queue_eligible(S, probe)) {
            /* the pixel planes of the frames that will queue for a slab: set aside before the slabs
             * take what HBM has (when HBM is the limit nobody knows yet how many slabs will fit) */
            const fa_image *im = jobs[probe.job].image;
            const size_t need = align_up((size_t) im->width * im->height * (im->color ? 3 : 1) * 2, 256);
            const size_t frames = hbm_bound ? S->slots.size() : S->slots.size() - resident;
            if (hipMalloc((void **) &S->qpix, need * frames) == hipSuccess) { S->qpix_bytes = need * frames; S->qpix_used = 0; }
            else { S->qpix = nullptr; (void) hipGetLastError(); }
        }
        bool still_bound = hbm_bound;
        if (hbm_bound) {
            /* first remedy: the triangular Gram tables -- half the slab; the kernel build that reads
             * them exists for the default geometry at the wide workgroup (frames with more than 3072
             * states: 4K), where memory is what keeps CUs idle.  FIASCO_AMD_NO_TRI keeps the full tables. */
            for (size_t k = 0; k < S->slots.size(); k++) {
                FrameSlot &fs = S->slots[k];
                if (!fs.big && fs.P > 12 * 256 && !fa_knob("FIASCO_AMD_NO_TRI")) fs.tri = true;
            }
            FrameSlot probe2 = S->slots[0];
            slot_layout(S, probe2);
            still_bound = probe2.L.total * want > free_b + pooled;
        }
        if (still_bound)
            for (size_t k = 0; k < S->slots.size(); k++) {
                FrameSlot &fs = S->slots[k];
                const fa_job *job = &jobs[fs.job];
                const fa_cparams *cp = &job->cp;
                unsigned bw = fa_width_of_level(cp->lc_max_level), bh = fa_height_of_level(cp->lc_max_level);
                size_t blocks = (size_t) ((job->image->width + bw - 1) / bw) * ((job->image->height + bh - 1) / bh);
                size_t tight = align_up(blocks + blocks * 3 / 20 + 64, 64);
                if ((size_t) fs.floorP > tight) tight = align_up((size_t) fs.floorP, 64);   /* never below a known need */
                if (tight > cp->limit_states) tight = align_up(cp->limit_states, 64);
                if ((size_t) fs.P <= tight || fs.spec) continue;
                const size_t cap = align_up(cp->limit_states, 64);
                fs.P = (int) tight;
                fs.PA = job->image->color ? (int) (3 * tight > cap ? cap : 3 * tight) : fs.P;
                if ((size_t) fs.floorPA > (size_t) fs.PA) fs.PA = (int) (align_up((size_t) fs.floorPA, 64) > cap ? cap : align_up((size_t) fs.floorPA, 64));
                if (fs.PA < fs.P) fs.PA = fs.P;
            }
    }
    for (size_t k = 0; k < S->slots.size(); k++) {
        FrameSlot &fs = S->slots[k];
        const bool elig = queue_eligible(S, fs);
        slot_layout(S, fs);
        if (elig && queue_layout(S, fs) && S->lenders >= S->lender_cap
            && stage_borrower(S, fs, S->slots.size() - k))
            continue;
        if (stage_slot(S, fs)) {
            if (elig && S->lender0 < 0) {
                S->lender0 = (int) k; S->lenders = 1;
                S->qL = fs.L; S->qP = fs.P; S->qPA = fs.PA; S->qbig = fs.big; S->qtri = fs.tri;
                /* workgroups the chip holds at once: frame_coder.hip FC_WG_PER_CU of the build the
                 * launch will use (wide build for P > 3072: one per CU) */
                S->lender_cap = (size_t) cus * frames_per_cu(fs.big, fs.P > 12 * 256 || fs.wide_only || fs.hm || fs.gm);
                if (fa_knob("FIASCO_AMD_QUEUE_SLABS") && atoi(fa_knob("FIASCO_AMD_QUEUE_SLABS")) > 0)
                    S->lender_cap = (size_t) atoi(fa_knob("FIASCO_AMD_QUEUE_SLABS"));     /* tests: a short queue on small batches */
            } else if (elig && queue_layout(S, fs)) S->lenders++;
            continue;
        }
        if (fs.rejected) continue;         /* outside the device scope: message recorded */
        if (elig && queue_layout(S, fs) && S->lenders >= 1) {     /* HBM is full: queue */
            S->jobs[fs.job].errmsg[0] = 0;
            if (stage_borrower(S, fs, S->slots.size() - k)) { S->lender_cap = S->lenders; continue; }
        }
        if (k == 0) continue;              /* does not fit even alone: error already recorded */
        S->jobs[fs.job].errmsg[0] = 0;     /* later wave */
        break;
    }
    (void) hipStreamSynchronize(S->stream);
    S->ok = true;
    return S;
}

/* ---- replacement inputs for a staged batch (a stream of batches) ---- */

static int16_t *core1_upload_buffer(void *h, size_t bytes)
{
    Staged *S = (Staged *) h;
    if (!S || !S->ok || !bytes) return nullptr;
    /* the previous upload has left this memory long ago (a whole pass lies in between) */
    if (S->ustream) (void) hipStreamSynchronize(S->ustream);
    if (bytes > S->up_host_bytes || S->up_host_shared) {
        if (S->up_host && !S->up_host_shared) (void) hipHostFree(S->up_host);
        S->up_host_shared = false;
        S->up_host = nullptr; S->up_host_bytes = 0;
        if (hipHostMalloc((void **) &S->up_host, bytes, hipHostMallocDefault) != hipSuccess) {
            S->up_host = nullptr; (void) hipGetLastError();
            return nullptr;
        }
        S->up_host_bytes = bytes;
    }
    return (int16_t *) S->up_host;
}

static int core1_upload_commit(void *h)
{
    Staged *S = (Staged *) h;
    if (!S || !S->ok || !S->up_host) return 0;
    if (!S->ustream && hipStreamCreateWithFlags(&S->ustream, hipStreamNonBlocking) != hipSuccess) {
        S->ustream = nullptr; (void) hipGetLastError(); return 0;
    }
    if (!S->ev_up && hipEventCreateWithFlags(&S->ev_up, hipEventDisableTiming) != hipSuccess) {
        S->ev_up = nullptr; (void) hipGetLastError(); return 0;
    }
    /* the buffer the RUNNING pass does not read; it holds the planes of THIS share's frames back to back (with
     * several shares the frames of a share are every D-th of the caller's buffer): one copy per frame */
    const int p = S->up_parity ^ 1;
    size_t need = 0;
    for (size_t k = 0; k < S->slots.size(); k++) {
        const fa_image *im = S->jobs[S->slots[k].job].image;
        const size_t npix = (size_t) im->width * im->height * (im->color ? 3 : 1);
        const size_t o = (size_t) ((const char *) im->pixels[0] - S->up_host);
        if ((const char *) im->pixels[0] < S->up_host || o + npix * 2 > S->up_host_bytes) {
            fa_set_error("upload: frame planes lie outside the upload buffer");
            return 0;
        }
        need += align_up(npix * 2, 256);
    }
    if (!need) return 1;                         /* nothing the device can encode */
    if (need > S->up_dev_bytes[p]) {
        if (S->up_dev[p]) (void) hipFree(S->up_dev[p]);
        S->up_dev[p] = nullptr; S->up_dev_bytes[p] = 0;
        if (hipMalloc((void **) &S->up_dev[p], need) != hipSuccess) {
            S->up_dev[p] = nullptr; (void) hipGetLastError();
            fa_set_error("out of HBM: no room for %.1f MiB of replacement frames", need / 1048576.0);
            return 0;
        }
        S->up_dev_bytes[p] = need;
    }
    {
        size_t at = 0;
        bool fail = false;
        for (size_t k = 0; k < S->slots.size() && !fail; k++) {
            const fa_image *im = S->jobs[S->slots[k].job].image;
            const size_t len = (size_t) im->width * im->height * (im->color ? 3 : 1) * 2;
            fail = hipMemcpyAsync(S->up_dev[p] + at, im->pixels[0], len, hipMemcpyHostToDevice, S->ustream) != hipSuccess;
            /* taken over by the next submit: a re-encode of the RUNNING pass (capacity guess too
             * small) still reads that pass's frames */
            S->slots[k].ext_next = (const int16_t *) (S->up_dev[p] + at);
            at += align_up(len, 256);
        }
        if (fail || hipEventRecord(S->ev_up, S->ustream) != hipSuccess) {
            fa_set_error("HIP error: %s", hipGetErrorString(hipGetLastError()));
            return 0;
        }
    }
    S->up_pending = true;
    return 1;
}

/* copy the finished automaton of one frame back into the job's fa_wfa */
static int collect(Staged *S, FrameSlot &fs, const char *pinned)
{
    fa_job *job = &S->jobs[fs.job];
    const DevFrame &F = fs.F;
    const Layout &L = fs.L;
    const int P = fs.PA;                 /* pitch of the automaton arrays */
    fa_wfa *w = job->wfa;
    unsigned ns = (unsigned) F.states;
    size_t span = L.pool_states - L.tree;
    std::vector<char> own;
    if (!pinned) {                       /* no staging buffer: plain synchronous copy */
        own.resize(span);
        if (hipMemcpy(own.data(), fs.base + L.tree, span, hipMemcpyDeviceToHost) != hipSuccess) {
            snprintf(job->errmsg, sizeof job->errmsg, "HIP error: automaton download failed");
            return 0;
        }
        pinned = own.data();
    }
    struct { const char *p; const char *data() const { return p; } } host = { pinned };
    const int16_t *tree = (const int16_t *) (host.data());
    const int16_t *into = (const int16_t *) (host.data() + (L.into - L.tree));
    const float *weight = (const float *) (host.data() + (L.weight - L.tree));
    const float *fin = (const float *) (host.data() + (L.final_d - L.tree));
    const uint8_t *los = (const uint8_t *) (host.data() + (L.level_of_state - L.tree));
    const uint8_t *dt = (const uint8_t *) (host.data() + (L.domain_type - L.tree));
    const uint16_t *xs = (const uint16_t *) (host.data() + (L.x - L.tree));
    const uint16_t *ys = (const uint16_t *) (host.data() + (L.y - L.tree));
    const uint8_t *ycol = (const uint8_t *) (host.data() + (L.ycol - L.tree));
    const int16_t *mv = (const int16_t *) (host.data() + (L.mv - L.tree));
    const bool inter = job->frame_type != FA_I_FRAME;
    for (unsigned s = 0; s < w->basis_states; s++) w->level_of_state[s] = 0xff;   /* codec/control.c:133-173 */
    fa_wfa_remove_states(w, w->basis_states);
    for (unsigned s = w->basis_states; s < ns; s++) {
        w->final_distribution[s] = fin[s];
        w->domain_type[s] = dt[s];
        w->level_of_state[s] = los[s];
        w->delta_state[s] = 0;
        for (int l = 0; l < 2; l++) {
            FA_TREE(w, s, l) = tree[(size_t) l * P + s];
            w->x[s * 2 + l] = xs[(size_t) l * P + s];
            w->y[s * 2 + l] = ys[(size_t) l * P + s];
            w->y_state[s * 2 + l] = FA_RANGE;
            w->y_column[s * 2 + l] = F.color ? ycol[(size_t) l * P + s] : 0;
            w->prediction[s * 2 + l] = 0;
            if (inter) {
                fa_mv *m = &w->mv[s * 2 + l];
                m->type = mv[(size_t) (0 * 2 + l) * P + s]; m->fx = mv[(size_t) (1 * 2 + l) * P + s];
                m->fy = mv[(size_t) (2 * 2 + l) * P + s]; m->bx = mv[(size_t) (3 * 2 + l) * P + s];
                m->by = mv[(size_t) (4 * 2 + l) * P + s];
            }
            for (int e = 0; e < 6; e++) {
                FA_INTO(w, s, l, e) = into[(size_t) (l * 6 + e) * P + s];
                FA_WEIGHT(w, s, l, e) = weight[(size_t) (l * 6 + e) * P + s];
                if (FA_INTO(w, s, l, e) == FA_NO_EDGE) break;
            }
        }
    }
    if (F.color)                         /* the flags of EVERY state id: the next frame of a stream
                                          * starts from them (fa_job.ycol_carry) */
        for (unsigned s = 0; s < w->cap && s < (unsigned) P; s++)
            for (int l = 0; l < 2; l++) w->y_column[s * 2 + l] = ycol[(size_t) l * P + s];
    w->states = ns;
    w->root_state = (unsigned) F.root_state;
    job->stats[0].costs = F.costs; job->stats[0].err = F.err;
    job->stats[0].tree_bits = F.tree_bits; job->stats[0].matrix_bits = F.matrix_bits;
    job->stats[0].weights_bits = F.weights_bits;
    if (F.color) {
        for (int b = 0; b < 2; b++) {
            job->stats[b + 1].costs = F.c_costs[b]; job->stats[b + 1].err = F.c_err[b];
            job->stats[b + 1].tree_bits = F.c_tree_bits[b];
            job->stats[b + 1].matrix_bits = F.c_matrix_bits[b];
            job->stats[b + 1].weights_bits = F.c_weights_bits[b];
        }
        /* co-located luminance states (codec/subdivide.c:167-173,560-567): a pure function of
         * the finished trees -- walk each chroma tree next to the luminance tree.  The root is
         * {{Y, Cb}, {Cr, -}} (codec/coder.c:803-833). */
        int ycb = FA_TREE(w, ns - 1, 0), crs = FA_TREE(w, ns - 1, 1);
        int roots[2] = { FA_TREE(w, ycb, 1), FA_TREE(w, crs, 0) };
        int yroot = FA_TREE(w, ycb, 0);
        std::vector<std::pair<int, int>> stack;
        for (int b = 0; b < 2; b++) {
            stack.push_back(std::make_pair(roots[b], yroot));
            while (!stack.empty()) {
                std::pair<int, int> t = stack.back();
                stack.pop_back();
                int s = t.first, y = t.second;
                if (s == FA_RANGE || (unsigned) s < w->basis_states) continue;
                for (int l = 0; l < 2; l++) {
                    int ny = y != FA_RANGE ? FA_TREE(w, y, l) : FA_RANGE;
                    w->y_state[s * 2 + l] = (int16_t) ny;
                    stack.push_back(std::make_pair((int) FA_TREE(w, s, l), ny));
                }
            }
        }
    }
    job->lc_min_level_out = (unsigned) F.lc_min_out;
    job->status = 1;
    g_stats.frames += 1;
    g_stats.bytes_mp += F.bytes_mp; g_stats.bytes_img += F.bytes_img; g_stats.bytes_gram += F.bytes_gram;
    g_stats.n_mp += F.n_mp; g_stats.n_steps += F.n_steps; g_stats.n_blocks += F.n_blocks;
    g_stats.n_appends += F.n_appends; g_stats.n_fulleval += F.n_fulleval;
    g_stats.t_init += F.t_init; g_stats.t_approx += F.t_approx; g_stats.t_ipis += F.t_ipis;
    g_stats.t_append += F.t_append; g_stats.t_serial += F.t_serial; g_stats.t_total += F.t_total;
    g_stats.t_mpA += F.t_mpA; g_stats.t_mpB += F.t_mpB; g_stats.n_blockevals += F.n_blockevals;
    for (int k = 0; k < 8; k++) g_stats.dbg[k] += F.dbg[k];
    g_stats.states_sum += ns;
    if (ns > g_stats.states_max) g_stats.states_max = ns;
    cap_hint_put(job, F.ystates_out, F.states);
    if (fa_knob("FIASCO_AMD_CAP_TRACE"))
        fprintf(stderr, "capacity: frame type %d used %d table states of %d, %d states of %d; %.3f s on the device\n", job->frame_type,
                F.ystates_out, fs.P, F.states, fs.PA, (double) F.t_total / 1e8);
    return 1;
}

/* device-side state of the frame queue: the ring of free slabs, the counters, and the map
 * of the descriptor's slab pointers (one bit per 8-byte word): the words that move with the base
 * when the same frame is laid out for two different slabs, plus pack_src */
static bool queue_resources(Staged *S, size_t frames)
{
    if (5 * frames > S->ring_n) {
        if (S->d_ring) (void) hipFree(S->d_ring);
        S->d_ring = nullptr; S->ring_n = 0;
        if (hipMalloc((void **) &S->d_ring, sizeof(unsigned long long) * 5 * frames) != hipSuccess) { (void) hipGetLastError(); return false; }
        S->ring_n = 5 * frames;
    }
    if (!S->d_queue && hipMalloc((void **) &S->d_queue, 10 * sizeof(unsigned)) != hipSuccess) {
        S->d_queue = nullptr; (void) hipGetLastError(); return false;
    }
    if (!S->ptrmask_ready) {
        const size_t words = FC_DESC_WORDS, mwords = (words + 31) / 32;
        std::vector<unsigned> mask(mwords, 0u);
        FrameSlot a = S->slots[S->lender0], b = S->slots[S->lender0];
        const fa_job *job = &S->jobs[a.job];
        b.base = a.base + (1u << 24);
        fill_frame(a, job); fill_frame(b, job);
        const unsigned long long *wa = (const unsigned long long *) &a.F, *wb = (const unsigned long long *) &b.F;
        for (size_t w = 0; w < sizeof(DevFrame) / 8; w++)
            if (wa[w] != wb[w]) mask[w >> 5] |= 1u << (w & 31);
        const size_t wp = offsetof(DevFrame, pack_src) / 8;
        mask[wp >> 5] |= 1u << (wp & 31);
        if (!S->d_ptrmask && hipMalloc((void **) &S->d_ptrmask, mwords * sizeof(unsigned)) != hipSuccess) {
            S->d_ptrmask = nullptr; (void) hipGetLastError(); return false;
        }
        if (hipMemcpy(S->d_ptrmask, mask.data(), mwords * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) return false;
        S->ptrmask_ready = true;
    }
    return true;
}

/* build the next launch from the frames that are staged and not yet encoded, upload their
 * descriptors and start the kernel(s); nothing is waited for.  Returns false when there is
 * nothing to launch. */
static bool launch_wave(Staged *S)
{
    std::vector<size_t> &batch = S->batch;
    batch.clear();
    for (size_t k = 0; k < S->slots.size(); k++)
        if (S->slots[k].staged && !S->slots[k].done) batch.push_back(k);
    if (batch.empty()) return false;
    /* one launch per kernel build (frame_coder.hip): geometry {default, big} x workgroup width
     * {256, 512 or 1024 threads}.  The wide builds take launches with no more frames than CUs (the
     * chip cannot be filled with frames anyway: give each frame twice the lanes) and frames
     * whose state capacity exceeds the 256-thread build's register-resident scan (4K). */
    size_t group_n[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, group_lend[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, group_borrow[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    {
        int cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        const bool few = batch.size() <= (size_t) cus && !fa_knob("FIASCO_AMD_NO_WIDE");
        /* per build: first the frames of the queue's layout -- those with a slab (the queue's
         * workgroups), then those without --, then every other frame (one workgroup each) */
        std::vector<size_t> ordered;
        for (int g = 0; g < 9; g++)
            for (int part = 0; part < 3; part++)
                for (size_t b = 0; b < batch.size(); b++) {
                    const FrameSlot &fs = S->slots[batch[b]];
                    const bool wide = few || fs.P > 12 * 256 || fs.wide_only;
                    /* group 5: the FC_HM build, 6: the FC_GM build; groups 7, 8: several workgroups per frame (FC_SPEC builds, 256 / 1024 threads) */
                    const bool spec = fs.spec && (S->specG >= 2 || fa_knob("FIASCO_AMD_SPEC_BUILD1")) && !fs.borrow && !fs.tri && !fs.big && fs.P <= 12 * 1024;
                    /* (FIASCO_AMD_SPEC_WIDE=0 / 1: experiments with the width of the workgroups) */
                    const char *sw = fa_knob("FIASCO_AMD_SPEC_WIDE");
                    const bool spec_wide = spec && (fs.P > 12 * 256 || fs.wide_only || (sw && atoi(sw) == 1));
                    if ((spec ? (spec_wide ? 8 : 7) : fs.gm ? 6 : fs.hm ? 5 : fs.tri ? 4 : (int) fs.big * 2 + (int) wide) != g) continue;
                    const bool q = S->borrowers && queue_eligible(S, fs) && queue_layout(S, fs);
                    const int where = fs.borrow ? 1 : q ? 0 : 2;
                    if (where != part) continue;
                    ordered.push_back(batch[b]);
                    group_n[g]++;
                    if (g < 5) g_stats.frames_by_build[g]++; else if (g <= 6) g_stats.frames_by_build[3]++; else g_stats.spec_frames++;
                    if (part == 0) group_lend[g]++; else if (part == 1) group_borrow[g]++;
                }
        batch.swap(ordered);
    }
    std::vector<DevFrame> &hf = S->hf;
    hf.resize(batch.size());
    S->d_trace = nullptr;
    const char *trace_path = fa_knob("FIASCO_AMD_TRACE");
    const int trace_cap = 400000;
    for (size_t b = 0; b < batch.size(); b++) hf[b] = S->slots[batch[b]].F;
    {   /* where every frame packs its finished automaton */
        S->parity ^= 1;
        char *&pack = S->d_pack[S->parity];
        S->pack_off.assign(batch.size(), 0);
        size_t need = 0;
        for (size_t b = 0; b < batch.size(); b++) {
            const Layout &L = S->slots[batch[b]].L;
            S->pack_off[b] = need;
            need += align_up(L.pool_states - L.tree, 256);
        }
        S->pack_need = need;
        if (need > S->d_pack_bytes[S->parity]) {
            if (pack) (void) hipFree(pack);
            pack = nullptr; S->d_pack_bytes[S->parity] = 0;
            if (hipMalloc((void **) &pack, need) == hipSuccess) S->d_pack_bytes[S->parity] = need;
            else { pack = nullptr; (void) hipGetLastError(); }
        }
        if (!S->cstream && hipStreamCreateWithFlags(&S->cstream, hipStreamNonBlocking) != hipSuccess) {
            S->cstream = nullptr; (void) hipGetLastError();
        }
        S->packed = pack != nullptr && S->cstream != nullptr;
        for (size_t b = 0; b < batch.size(); b++) {
            const FrameSlot &fs = S->slots[batch[b]];
            hf[b].pack_src = fs.F.slab_base + fs.L.tree;     /* a borrower's is re-based by the kernel */
            hf[b].pack_bytes = (unsigned) (fs.L.pool_states - fs.L.tree);
            hf[b].pack_dst = S->packed ? pack + S->pack_off[b] : nullptr;
        }
    }
    if (trace_path && hipMalloc((void **) &S->d_trace, sizeof(FcTrace) * trace_cap) == hipSuccess) {
        hf[0].trace = S->d_trace; hf[0].trace_cap = trace_cap;
    }
    bool fail = false;
    S->spec_frames.clear();
    S->spec_first[0] = S->spec_first[1] = 0; S->spec_n[0] = S->spec_n[1] = 0;
    if (group_n[7] + group_n[8] && S->specG < 2) {       /* FIASCO_AMD_SPEC_BUILD1: the build alone */
        const size_t nall = group_n[7] + group_n[8], first_all = batch.size() - nall;
        S->spec_first[0] = first_all; S->spec_n[0] = group_n[7];
        S->spec_first[1] = first_all + group_n[7]; S->spec_n[1] = group_n[8];
        for (size_t i = 0; i < nall; i++) hf[first_all + i].spec = nullptr;
    } else if (group_n[7] + group_n[8]) {
        /* the speculating frames (groups 5 and 6: 256 / 1024 threads per workgroup; they are the last of
         * the batch): control block + checkpoint slots + block list + table ring per frame, then per
         * verifier workgroup its private <sub-block, state> tables, scan scratch and pool list; verifier
         * v of a frame owns the state ids [P - 16 v, P - 16 (v - 1)) */
        const int G = S->specG;
        const int T = spec_workers(G), NV = G - 1 - T;           /* table workers, verifiers */
        const size_t nall = group_n[7] + group_n[8], first_all = batch.size() - nall;
        /* append helpers: only a launch of ONE width (the residency sum below is per build) */
        S->specH[0] = S->specH[1] = 0;
        if (!S->no_app) {
            if (group_n[8] && !group_n[7]) S->specH[1] = spec_app_policy(group_n[8], S->ncu, G, true, 1);
            else if (group_n[7] && !group_n[8]) S->specH[0] = spec_app_policy(group_n[7], S->ncu, G, false, fc_occupancy_spec());
        }
        S->spec_first[0] = first_all; S->spec_n[0] = group_n[7];
        S->spec_first[1] = first_all + group_n[7]; S->spec_n[1] = group_n[8];
        /* one span for every frame of the launch (sized for the largest) */
        size_t max_blocks = 0, max_tab = 0, max_slot = 0;
        std::vector<std::vector<uint16_t>> lists(nall);
        for (size_t i = 0; i < nall; i++) {
            const DevFrame &F = hf[first_all + i];
            spec_block_list(F, lists[i]);
            if (lists[i].size() / 2 > max_blocks) max_blocks = lists[i].size() / 2;
            const size_t tab = align_up(((size_t) F.NS + (size_t) F.NA) * (size_t) F.P * 4, 256);
            if (tab > max_tab) max_tab = tab;
            const size_t slot = i < group_n[7] ? fc_spec_slot_bytes() : fc_spec_slot_bytes_wide();
            if (slot > max_slot) max_slot = slot;
        }
        const size_t off_blocks = align_up((size_t) fc_spec_ctl_bytes() + (size_t) 2 * FC_SPEC_W * max_slot, 256);      /* checkpoint + result slots */
        const size_t off_tabs = align_up(off_blocks + max_blocks * 4, 256);
        const size_t span = align_up(off_tabs + (size_t) FC_SPEC_R * max_tab, 256);
        std::vector<size_t> priv(nall);
        size_t need = span * nall;
        for (size_t i = 0; i < nall; i++) {
            const DevFrame &F = hf[first_all + i];
            const size_t P = (size_t) F.P;
            priv[i] = align_up((size_t) F.NS * P * 4, 256)
                      + align_up((size_t) F.NA * P * 4, 256) + 3 * align_up(P * 4, 256) + align_up((size_t) FC_MAXED * P * 4, 256)
                      + align_up(P, 256) + align_up((P + 8) * 2, 256) + align_up(((size_t) F.PA + 8) * 4, 256);
            need += priv[i] * (size_t) NV;
        }
        if (need > S->d_spec_bytes) {
            if (S->d_spec) (void) hipFree(S->d_spec);
            S->d_spec = nullptr; S->d_spec_bytes = 0;
            if (hipMalloc((void **) &S->d_spec, need) == hipSuccess) S->d_spec_bytes = need; else (void) hipGetLastError();
        }
        if (nall * (size_t) (G - 1) > S->vframes_n) {
            if (S->d_vframes) (void) hipFree(S->d_vframes);
            S->d_vframes = nullptr; S->vframes_n = 0;
            if (hipMalloc((void **) &S->d_vframes, sizeof(DevFrame) * nall * (size_t) (G - 1)) == hipSuccess) S->vframes_n = nall * (size_t) (G - 1);
            else (void) hipGetLastError();
        }
        S->spec_ctl_span = span;
        if (S->d_spec && S->d_vframes) {
            std::vector<DevFrame> vf(nall * (size_t) (G - 1));
            size_t o = span * nall;
            for (size_t i = 0; i < nall; i++) {
                DevFrame &C = hf[first_all + i];
                C.spec = (FcSpecCtl *) (S->d_spec + span * i);
                C.spec_role = 0; C.spec_G = G; C.spec_T = T;
                C.spec_cap = C.P - NV * FC_SPEC_TEMPS;
                C.spec_tb = C.P;
                for (int r = 1; r <= T; r++) {                   /* table workers: the chain's descriptor */
                    DevFrame &V = vf[i * (size_t) (G - 1) + (size_t) (r - 1)];
                    V = C;
                    V.spec_role = r; V.trace = nullptr; V.trace_cap = 0; V.pack_dst = nullptr;
                }
                for (int r = T + 1; r < G; r++) {
                    DevFrame &V = vf[i * (size_t) (G - 1) + (size_t) (r - 1)];
                    V = C;
                    V.spec_role = r; V.spec_tb = C.P - (r - T) * FC_SPEC_TEMPS;
                    const size_t P = (size_t) C.P;
                    char *q = S->d_spec + o;
                    V.ipis = (float *) q;  q += align_up((size_t) C.NS * P * 4, 256);
                    V.d5 = (float *) q;    q += align_up((size_t) C.NA * P * 4, 256);
                    V.num = (float *) q;   q += align_up(P * 4, 256);
                    V.den = (float *) q;   q += align_up(P * 4, 256);
                    V.est = (float *) q;   q += align_up(P * 4, 256);
                    V.ipdo = (float *) q;  q += align_up((size_t) FC_MAXED * P * 4, 256);
                    V.used = (uint8_t *) q; q += align_up(P, 256);
                    V.pool_states = (int16_t *) q; q += align_up((P + 8) * 2, 256);
                    V.hits = (int *) q;    q += align_up(((size_t) C.PA + 8) * 4, 256);
                    V.trace = nullptr; V.trace_cap = 0; V.pack_dst = nullptr;
                    o += priv[i];
                }
                S->spec_frames.push_back(first_all + i);
            }
            /* control blocks: zero, then what the host knows (sizes, offsets, the block list) */
            for (size_t i = 0; i < nall && !fail; i++)
                fail = hipMemsetAsync(S->d_spec + span * i, 0, off_tabs, S->stream) != hipSuccess;
            fail = fail || hipStreamSynchronize(S->stream) != hipSuccess;
            for (size_t i = 0; i < nall && !fail; i++) {
                FcSpecCtl h;
                memset(&h, 0, sizeof h);
                h.slot_bytes = (unsigned) max_slot;
                /* a colour frame: the chroma bands' tables too, from every workgroup but the chain (even
                 * without table workers for the luminance band) */
                h.n_blocks = (unsigned) (lists[i].size() / 2);
                h.n_tabs = hf[first_all + i].color ? 3u * h.n_blocks : (T ? h.n_blocks : 0u);
                h.tab_stride = (unsigned) max_tab;
                /* 120 us: about what the chain needs to build the tables itself (tests: FIASCO_AMD_SPEC_TABWAIT=0
                 * makes it take the worker's tables only when they are there already) */
                h.tab_wait = fa_knob("FIASCO_AMD_SPEC_TABWAIT") ? (unsigned) atoi(fa_knob("FIASCO_AMD_SPEC_TABWAIT")) : 12000u;
                h.off_blocks = off_blocks; h.off_tabs = off_tabs;
                {   /* append helpers of the frame's width group (frames of the 256-thread build come first) */
                    const int wk = i < group_n[7] ? 0 : 1;
                    h.app_H = (unsigned) S->specH[wk];
                    h.app_min = wk ? 2048u : 512u;               /* two passes of the workgroup's lanes */
                    if (fa_knob("FIASCO_AMD_SPEC_APPMIN")) h.app_min = (unsigned) atoi(fa_knob("FIASCO_AMD_SPEC_APPMIN"));
                    h.app_dbg = fa_knob("FIASCO_AMD_SPEC_APPDBG") ? (unsigned) atoi(fa_knob("FIASCO_AMD_SPEC_APPDBG")) : 0u;
                    h.app_wait = fa_knob("FIASCO_AMD_SPEC_APPWAIT_MS") ? 100000u * (unsigned) atoi(fa_knob("FIASCO_AMD_SPEC_APPWAIT_MS")) : 200000000u;   /* 2 s */
                }
                fail = hipMemcpy(S->d_spec + span * i, &h, sizeof h, hipMemcpyHostToDevice) != hipSuccess;
                if (!fail && !lists[i].empty())
                    fail = hipMemcpy(S->d_spec + span * i + off_blocks, lists[i].data(), lists[i].size() * 2, hipMemcpyHostToDevice) != hipSuccess;
            }
            fail = fail || hipMemcpy(S->d_vframes, vf.data(), sizeof(DevFrame) * vf.size(), hipMemcpyHostToDevice) != hipSuccess;
        } else
            for (size_t i = 0; i < nall; i++) hf[first_all + i].spec = nullptr;      /* no memory: one workgroup per frame */
    }
    fail = fail || hipMemcpyAsync(S->d_frames, hf.data(), sizeof(DevFrame) * batch.size(),
                               hipMemcpyHostToDevice, S->stream) != hipSuccess;
    /* ---- one persistent launch per kernel build: one workgroup per frame ---- */
    fail = fail || hipEventRecord(S->ev0, S->stream) != hipSuccess;
    {
        typedef void (*launch_fn)(DevFrame *, unsigned, unsigned, unsigned long long *, unsigned *, const unsigned *,
                                  unsigned long long, unsigned, hipStream_t);
        /* bound of a queued frame's wait for a slab (frame_coder.hip); tests shorten it */
        unsigned long long qwait = FC_QUEUE_WAIT_TICKS;
        if (fa_knob("FIASCO_AMD_QUEUE_WAIT_MS")) qwait = 100000ull * (unsigned long long) atoll(fa_knob("FIASCO_AMD_QUEUE_WAIT_MS"));
        static const launch_fn launch[7] = { fc_launch, fc_launch_wide, fc_launch_big, fc_launch_big_wide, fc_launch_wide_tri, fc_launch_big_hm, fc_launch_big_gm };
        size_t first = 0;
        for (int k = 0; k < 2 && !fail; k++) {
            if (!S->spec_n[k]) continue;
            /* without the verifiers' buffers: G = 1, the chain alone */
            const bool on = S->d_spec && S->d_vframes && !S->spec_frames.empty();
            const size_t all_first = S->spec_first[0];
            DevFrame *vfr = S->d_vframes ? S->d_vframes + (S->spec_first[k] - all_first) * (size_t) (S->specG - 1) : nullptr;
            (k ? fc_launch_spec_wide : fc_launch_spec)(S->d_frames + S->spec_first[k], vfr, (unsigned) S->spec_n[k],
                                                       on ? (unsigned) S->specG : 1u, on ? (unsigned) S->specH[k] : 0u, S->stream);
        }
        for (int g = 0; g < 7 && !fail; g++) {
            size_t plain = group_n[g], at = first;
            if (group_borrow[g]) {
                /* the queue: group_lend[g] frames with slabs first, then the frames that borrow one */
                const size_t nq = group_lend[g] + group_borrow[g];
                if (!group_lend[g] || !S->packed || !queue_resources(S, batch.size())) {
                    for (size_t b = at + group_lend[g]; b < at + nq; b++) hf[b].status = FC_ERR_INTERNAL;
                    fail = fail || hipMemcpyAsync(S->d_frames + at, hf.data() + at, sizeof(DevFrame) * nq,
                                                  hipMemcpyHostToDevice, S->stream) != hipSuccess;
                    if (group_lend[g])
                        launch[g](S->d_frames + at, (unsigned) group_lend[g], (unsigned) group_lend[g], nullptr, nullptr, nullptr, qwait, 1u, S->stream);
                } else {
                    unsigned long long *ring = S->d_ring + (size_t) g * batch.size();
                    fail = fail || hipMemsetAsync(S->d_queue + 2 * g, 0, 2 * sizeof(unsigned), S->stream) != hipSuccess;
                    fail = fail || hipMemsetAsync(ring, 0, nq * sizeof(unsigned long long), S->stream) != hipSuccess;
                    launch[g](S->d_frames + at, (unsigned) nq, (unsigned) group_lend[g], ring, S->d_queue + 2 * g,
                              S->d_ptrmask, qwait, 1u, S->stream);
                }
                at += nq; plain -= nq;
            }
            if (plain) {
                /* big frames that leave the chip empty (a step of a video: 30 GOPs): several workgroups build the
                 * tables of a frame (frame_coder.h FcCoop).  One workgroup of 512 threads per CU; every workgroup
                 * of the launch must be resident: W x frames <= CUs, and nothing else launched beside it */
                unsigned W = 1;
                bool any_bx = false;                /* a long basis: its table rows are built by the frame's own workgroup */
                for (size_t b = at; b < at + plain; b++) any_bx = any_bx || hf[b].bx != nullptr;
                if (g == 3 && batch.size() == plain && !S->no_coop && !any_bx) {
                    W = coop_policy(plain, S->ncu ? S->ncu : 256);
                    if (fa_knob("FIASCO_AMD_COOP") && atoi(fa_knob("FIASCO_AMD_COOP")) >= 1) {
                        W = (unsigned) atoi(fa_knob("FIASCO_AMD_COOP"));
                        if (W > 8) W = 8;
                        if ((size_t) W * ((plain + 7) / 8 * 8) > (size_t) (S->ncu ? S->ncu : 256)) W = 1;
                    }
                }
                if (S->no_coop) S->no_coop_done = true;
                if (W > 1) {
                    unsigned D = 1;
                    while ((1u << D) < W) D++;
                    if (fa_knob("FIASCO_AMD_COOP_DEPTH") && atoi(fa_knob("FIASCO_AMD_COOP_DEPTH")) >= 1 && atoi(fa_knob("FIASCO_AMD_COOP_DEPTH")) <= 3
                        && (1 << atoi(fa_knob("FIASCO_AMD_COOP_DEPTH"))) >= (int) W)
                        D = (unsigned) atoi(fa_knob("FIASCO_AMD_COOP_DEPTH"));
                    FcCoop &hdr = S->coop_hdr;
                    memset(&hdr, 0, sizeof hdr);
                    hdr.depth = D;
                    /* tests: FIASCO_AMD_COOP_WAIT_MS shortens the frame's wait, FIASCO_AMD_COOP_DEAF=1 sends the helpers
                     * home at once (the frame then fails with FC_ERR_COOP and is searched again by one workgroup) */
                    hdr.done_ticks = fa_knob("FIASCO_AMD_COOP_WAIT_MS") ? 100000ull * (unsigned long long) atoll(fa_knob("FIASCO_AMD_COOP_WAIT_MS"))
                                                                            : FC_COOP_DONE_TICKS;
                    hdr.quit = fa_knob("FIASCO_AMD_COOP_DEAF") ? 1u : 0u;
                    hdr.minsub = fa_knob("FIASCO_AMD_COOP_MINSUB") ? atoi(fa_knob("FIASCO_AMD_COOP_MINSUB")) : 1;
                    for (size_t b = at; b < at + plain && !fail; b++)
                        fail = hipMemcpyAsync(hf[b].coop, &hdr, sizeof(FcCoop), hipMemcpyHostToDevice, S->stream) != hipSuccess;
                    g_stats.coop_frames += plain; g_stats.coop_workgroups = W;
                }
                launch[g](S->d_frames + at, (unsigned) plain, (unsigned) plain, nullptr, nullptr, nullptr, qwait, W, S->stream);
            }
            first += group_n[g];
        }
    }
    fail = fail || hipGetLastError() != hipSuccess;
    fail = fail || hipEventRecord(S->ev1, S->stream) != hipSuccess;
    S->launch_failed = fail;
    return true;
}

/* wait for the launch, download the descriptors, collect every finished frame; a frame whose
 * capacity guess was too small gets a bigger slab and stays "not done" for the next launch */
static void complete_wave(Staged *S)
{
    std::vector<size_t> &batch = S->batch;
    std::vector<DevFrame> &hf = S->hf;
    bool fail = S->launch_failed;
    fail = fail || hipStreamSynchronize(S->stream) != hipSuccess;
    if (!fail) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, S->ev0, S->ev1) == hipSuccess) {
            g_stats.kernel_ms += ms;
            g_stats.launches += 1;
        }
        fail = hipMemcpy(hf.data(), S->d_frames, sizeof(DevFrame) * batch.size(),
                         hipMemcpyDeviceToHost) != hipSuccess;
    }
    if (!fail && !S->spec_frames.empty() && S->d_spec) {
        std::vector<FcSpecCtl> ctl(S->spec_frames.size());
        if (hipMemcpy2D(ctl.data(), sizeof(FcSpecCtl), S->d_spec, S->spec_ctl_span, sizeof(FcSpecCtl), ctl.size(),
                        hipMemcpyDeviceToHost) == hipSuccess)
            for (size_t i = 0; i < ctl.size(); i++) {
                g_stats.spec_tasks += ctl[i].n_tasks; g_stats.spec_confirmed += ctl[i].n_confirmed;
                g_stats.spec_wrong += ctl[i].n_wrong; g_stats.spec_timeout += ctl[i].n_timeout;
                g_stats.spec_inline += ctl[i].n_inline; g_stats.spec_wait += ctl[i].t_wait;
                g_stats.spec_tab_used += ctl[i].n_tab_used; g_stats.spec_tab_missed += ctl[i].n_tab_missed;
                g_stats.spec_adopted += ctl[i].n_adopted;
                g_stats.spec_app_rows += ctl[i].n_app_dealt; g_stats.spec_app_wait += ctl[i].t_app_wait;
            }
        else (void) hipGetLastError();
    }
    const char *trace_path = fa_knob("FIASCO_AMD_TRACE");
    if (S->d_trace && !fail && trace_path) {
        std::vector<FcTrace> tr((size_t) hf[0].trace_n);
        if (hipMemcpy(tr.data(), S->d_trace, sizeof(FcTrace) * tr.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            FILE *tf = fopen(trace_path, "wb");
            if (tf) { fwrite(tr.data(), sizeof(FcTrace), tr.size(), tf); fclose(tf); }
        }
    }
    if (S->d_trace) { (void) hipFree(S->d_trace); S->d_trace = nullptr; }
    if (fail) {
        const char *why = hipGetErrorString(hipGetLastError());   /* reading it clears it: once */
        for (size_t b = 0; b < batch.size(); b++) {
            FrameSlot &fs = S->slots[batch[b]];
            snprintf(S->jobs[fs.job].errmsg, sizeof S->jobs[fs.job].errmsg, "HIP error: %s", why);
            fs.done = true;
        }
        S->broken = true;
        return;
    }
    /* all automata of the launch come down into one pinned buffer: one copy of the packed
     * buffer on the copy stream (not waited for here: the next launch may start first), or --
     * without a packed buffer -- one async copy per frame */
    std::vector<size_t> off(batch.size(), (size_t) -1);
    {
        size_t need = 0;
        if (S->packed) {
            need = S->pack_need;
            for (size_t b = 0; b < batch.size(); b++) if (hf[b].status == FC_OK) off[b] = S->pack_off[b];
        } else
        for (size_t b = 0; b < batch.size(); b++)
            if (hf[b].status == FC_OK) {
                const Layout &L = S->slots[batch[b]].L;
                off[b] = need;
                need += align_up(L.pool_states - L.tree, 256);
            }
        if (S->copy_pending) { (void) hipStreamSynchronize(S->cstream); S->copy_pending = false; }
        if (need > S->pinned_bytes) {
            if (S->pinned) (void) hipHostFree(S->pinned);
            S->pinned = nullptr; S->pinned_bytes = 0;
            if (hipHostMalloc((void **) &S->pinned, need, hipHostMallocDefault) == hipSuccess) S->pinned_bytes = need;
            else { S->pinned = nullptr; (void) hipGetLastError(); }
        }
        if (S->pinned && S->packed) {
            if (hipMemcpyAsync(S->pinned, S->d_pack[S->parity], need, hipMemcpyDeviceToHost, S->cstream) == hipSuccess)
                S->copy_pending = true;
            else {
                (void) hipGetLastError();
                for (size_t b = 0; b < batch.size(); b++) off[b] = (size_t) -1;
            }
        } else if (S->pinned) {
            for (size_t b = 0; b < batch.size(); b++)
                if (off[b] != (size_t) -1) {
                    const FrameSlot &fs = S->slots[batch[b]];
                    if (hipMemcpyAsync(S->pinned + off[b], fs.base + fs.L.tree, fs.L.pool_states - fs.L.tree,
                                       hipMemcpyDeviceToHost, S->stream) != hipSuccess)
                        off[b] = (size_t) -1;
                }
            (void) hipStreamSynchronize(S->stream);
        }
    }
    for (size_t b = 0; b < batch.size(); b++) {
        FrameSlot &fs = S->slots[batch[b]];
        fa_job *job = &S->jobs[fs.job];
        int st = hf[b].status;
        void *tr_keep = fs.F.trace;
        fs.F = hf[b];
        fs.F.trace = (FcTrace *) tr_keep; fs.F.trace_cap = 0;
        if (fs.ext_pix) fs.F.pix16 = fs.ext_pix;
        size_t cap = align_up(job->cp.limit_states, 64);
        /* (a frame that shares its slab with verifiers has less than fs.P for itself -- their private
         * state ids lie at the top of the capacity --: at the state limit it is encoded once more by one
         * workgroup with all of it, like the reference would, before "Maximum number of states" is said) */
        if (st == FC_ERR_CAPACITY && ((size_t) fs.P < cap || (size_t) fs.PA < cap || fs.spec)) {
            /* capacity guess too small: bigger slab, same inputs, encode again */
            g_stats.reencodes += 1;
            size_t np = align_up((size_t) fs.P + (size_t) fs.P / 2, 64);
            size_t npa = align_up((size_t) fs.PA + (size_t) fs.PA / 2, 64);
            if ((size_t) fs.P >= cap || np >= cap) fs.spec = false;
            /* a frame of a launch with more workgroups than CUs that outgrows the 256-thread build would come
             * back in the 1024-thread speculating build, one workgroup per CU: its verifiers might not be
             * resident (the chain's waits are bounded, but slow) -- one workgroup for such a frame */
            if (np > 3072 && S->specG > 1 && (size_t) S->specG * S->n > (size_t) S->ncu) fs.spec = false;
            if (fs.base) slab_release(fs.base, fs.bytes);
            /* a borrower gets a slab of its own; its pixel planes stay where they are (the queue's
             * pixel buffer or an upload buffer): the host copy may belong to the next pass by now */
            if (fs.borrow) { fs.borrow = false; S->borrowers--; }
            fs.base = nullptr; fs.staged = false;
            fs.P = (int) (np > cap ? cap : np);
            fs.PA = (int) (npa > cap ? cap : npa);
            if (fs.PA < fs.P) fs.PA = fs.P;
            if (fs.P > 12 * 1024) fs.spec = false;     /* beyond the speculating builds: one (wide) workgroup */
            if (!stage_slot(S, fs)) fs.done = true;
            continue;
        }
        if (st == FC_ERR_COOP && fs.spec && !S->no_app) {
            /* the append helpers of a speculating frame did not answer in time: again without helpers */
            S->no_app = true;
            continue;
        }
        if (st == FC_ERR_COOP && !S->no_coop_done) {
            /* the helper workgroups of the frame were not there in time (not resident: masked CUs, a busy device):
             * the frame keeps its slab and is searched again by one workgroup -- a retry instead of a failure */
            S->no_coop = true;
            continue;
        }
        if (st == FC_ERR_QUEUE && fs.borrow) {
            /* the frame never got a slab from the queue (bounded wait in the kernel): a slab of its
             * own in the next launch; its pixel planes stay where they are */
            fs.borrow = false; S->borrowers--;
            fs.base = nullptr; fs.staged = false;
            if (!stage_slot(S, fs)) fs.done = true;
            continue;
        }
        fs.done = true;
        if (st == FC_OK) {
            /* unpacking into the job's fa_wfa is host work on host memory: deferred so that
             * a following submit can start the device first (flush_unpack) */
            if (S->pinned && off[b] != (size_t) -1) S->to_unpack.push_back(std::make_pair(batch[b], off[b]));
            else S->good += collect(S, fs, nullptr);
        } else {
            const char *msg = "device coder failed";
            if (st == FC_ERR_STATES || st == FC_ERR_CAPACITY) msg = "Maximum number of states reached!";
            else if (st == FC_ERR_NOROOT) msg = "No root state generated!";
            else if (st == FC_ERR_QUEUE) msg = "device coder: frame queue gave no slab";
            else if (st == FC_ERR_COOP) msg = "device coder: the helper workgroups of the frame did not answer";
            else if (st == FC_ERR_INTERNAL) msg = "device coder: frame exceeds a built-in capacity (recursion depth, snapshot stack or 16384 states)";
            if (st > FC_ERR_QUEUE) snprintf(job->errmsg, sizeof job->errmsg, "%s (status %d)", msg, st);
            else snprintf(job->errmsg, sizeof job->errmsg, "%s", msg);
        }
    }
    (void) hipStreamSynchronize(S->stream);
}

static void flush_unpack(Staged *S)
{
    if (S->copy_pending) { (void) hipStreamSynchronize(S->cstream); S->copy_pending = false; }
    for (size_t i = 0; i < S->to_unpack.size(); i++)
        S->good += collect(S, S->slots[S->to_unpack[i].first], S->pinned + S->to_unpack[i].second);
    S->to_unpack.clear();
}

/* start encoding every staged frame; returns immediately (the kernel runs) */
static int core1_submit(void *h)
{
    Staged *S = (Staged *) h;
    if (!S || !S->ok) return 0;
    if (S->inflight) return 1;
    /* job status / automata of the previous pass stay readable until fa_core_finish() */
    for (size_t k = 0; k < S->slots.size(); k++) {
        S->slots[k].done = false;
        S->slots[k].src = S->jobs[S->slots[k].job].image;      /* see FrameSlot::src */
    }
    S->good = 0; S->broken = false;
    if (S->up_pending) {
        /* a new pass takes over the replacement inputs: the launch waits for their transfer,
         * descriptors are uploaded by every launch anyway */
        (void) hipStreamWaitEvent(S->stream, S->ev_up, 0);
        for (size_t k = 0; k < S->slots.size(); k++) {
            FrameSlot &fs = S->slots[k];
            if (fs.ext_next) { fs.ext_pix = fs.ext_next; fs.F.pix16 = fs.ext_pix; }
        }
        S->up_parity ^= 1;
        S->up_pending = false;
    }
    S->inflight = launch_wave(S);
    return 1;
}

/* wait for the submitted launch and bring every frame to completion (re-encodes with larger
 * slabs, later waves of a batch that did not fit into HBM at once).  After it returns the
 * jobs' automata are in host memory and the device is free for the next submit. */
static int core1_finish2(void *h, int resubmit)
{
    Staged *S = (Staged *) h;
    if (!S || !S->ok) return 0;
    if (!S->inflight) { core1_submit(h); }
    for (size_t k = 0; k < S->slots.size(); k++) S->jobs[S->slots[k].job].status = 0;
    for (;;) {
        if (S->inflight) { complete_wave(S); S->inflight = false; if (S->broken) break; }
        {   /* anything left to encode (bigger slabs, later waves)?  then the staging buffer
             * is needed again: unpack first */
            bool more = false;
            for (size_t k = 0; k < S->slots.size(); k++) if (!S->slots[k].done) more = true;
            if (more) flush_unpack(S);
        }
        if (launch_wave(S)) { S->inflight = true; continue; }
        /* stage a later wave (frames that did not fit while others held their slabs):
         * finished frames give their slabs back first (they are re-staged by the next
         * run if the batch is encoded again) */
        bool any = false, pending = false;
        for (size_t k = 0; k < S->slots.size(); k++) {
            FrameSlot &fs = S->slots[k];
            if (!fs.staged && !fs.done && !S->jobs[fs.job].errmsg[0]) pending = true;
        }
        if (!pending) break;
        for (size_t k = 0; k < S->slots.size(); k++) {
            FrameSlot &fs = S->slots[k];
            if (fs.done && fs.base) { slab_release(fs.base, fs.bytes); fs.base = nullptr; fs.staged = false; }
        }
        for (size_t k = 0; k < S->slots.size(); k++) {
            FrameSlot &fs = S->slots[k];
            if (fs.staged || fs.done || S->jobs[fs.job].errmsg[0]) continue;
            if (stage_slot(S, fs)) any = true; else if (!fs.rejected) break;
        }
        if (!any) break;
    }
    int good_before = S->good;
    if (resubmit && !S->broken) {
        /* next pass on the device first, then the host-side unpacking of this one */
        std::vector<std::pair<size_t, size_t>> keep;
        keep.swap(S->to_unpack);
        core1_submit(h);                           /* resets S->good */
        S->to_unpack.swap(keep);
        int g = S->good;
        S->good = good_before;
        flush_unpack(S);
        good_before = S->good;
        S->good = g;
        return good_before;
    }
    flush_unpack(S);
    return S->good;
}


/* ------------------------------------------------------------------ several devices in one process
 *
 * Frames (separate fiasco_coder() calls, frames of a gray all-intra stream, the groups of pictures a
 * sequence is coded in) are independent units (tiles are not: codec/tiling.c is dead code in
 * this reference).  The seam fa_core_*() therefore spreads the jobs of a batch round robin over the
 * devices of the process -- job i goes to device i mod D -- and runs every share on a host thread of its
 * own with its own stream, slab pool and log2 table (DevState); results come back in job order.  No
 * collective is involved: what crosses between devices is nothing, what comes back per frame is its
 * automaton (kilobytes) over PCIe as before.  (The reference call site this serves: video_coder()'s
 * frame loop, codec/coder.c:490-668.)
 *
 * Which devices: FIASCO_AMD_DEVICES="0,1,4" if set (an id may repeat -- two shares on one GPU: the test of
 * this path on a 1-GPU box); else, once fiasco_amd_set_device(d) has been called -- one process per GPU,
 * the multi-process harness -- just d; else every visible device.  With one device nothing below starts a
 * thread; every share of a call -- also the only one -- runs bound to its device (bind_share) and the caller's
 * current device is restored afterwards (for_each_share).
 *
 * Threading contract: the batch entries may be called from several host threads.  Calls with ONE share run
 * concurrently as before (each on its calling thread).  The worker threads of the shares k >= 1 belong to the
 * process, not to a batch: calls that spread over several shares are serialised by g_share_lock, one phase
 * (stage / submit / finish / upload) at a time.  The workers are detached and parked on a condition variable;
 * they are never joined (a dlclose of the library with several devices in use is not supported). */
static std::vector<int> g_devices;              /* empty = not resolved yet */
static int  g_device_explicit = -1;             /* fiasco_amd_set_device() */
static std::vector<DevState *> g_dev_state;     /* [k] for share k (k >= 1; share 0 uses g_state0) */
static pthread_mutex_t g_dev_lock = PTHREAD_MUTEX_INITIALIZER;

static void resolve_devices(void)
{
    pthread_mutex_lock(&g_dev_lock);
    if (g_devices.empty()) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) { (void) hipGetLastError(); ndev = 0; }
        const char *e = getenv("FIASCO_AMD_DEVICES");
        if (e && *e) {
            for (const char *q = e; *q; ) {
                char *end;
                long v = strtol(q, &end, 10);
                if (end == q) break;
                if (v >= 0 && v < ndev) g_devices.push_back((int) v);
                q = *end ? end + 1 : end;
            }
        } else if (g_device_explicit >= 0) g_devices.push_back(g_device_explicit);
        else for (int d = 0; d < ndev; d++) g_devices.push_back(d);
        if (g_devices.empty()) g_devices.push_back(-1);      /* -1: whatever the current device is (or none) */
        while (g_dev_state.size() < g_devices.size()) g_dev_state.push_back(g_dev_state.empty() ? &g_state0 : new DevState);
    }
    pthread_mutex_unlock(&g_dev_lock);
}

extern "C" int fiasco_amd_device_count(void)
{
    resolve_devices();
    return (int) g_devices.size();
}

/* the devices of this process, chosen by the caller: n ids (an id may repeat), or n = 0 for the rule above */
extern "C" int fiasco_amd_set_devices(const int *ids, int n)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void) hipGetLastError(); ndev = 0; }
    for (int i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= ndev) { fa_set_error("libfiasco_amd: no HIP device %d", ids[i]); return 0; }
    fiasco_amd_release_memory();
    pthread_mutex_lock(&g_dev_lock);
    g_devices.clear();
    g_device_explicit = -1;
    for (int i = 0; i < n; i++) g_devices.push_back(ids[i]);
    while (g_dev_state.size() < g_devices.size()) g_dev_state.push_back(g_dev_state.empty() ? &g_state0 : new DevState);
    pthread_mutex_unlock(&g_dev_lock);
    return 1;
}

/* one process per GPU: bind this process's coder to a device of the node */
extern "C" int fiasco_amd_set_device(int device)
{
    int cur = -1;
    /* the slab pools hold memory of the device they were allocated on: never carry them over */
    if (hipGetDevice(&cur) != hipSuccess || cur != device || g_devices.size() != 1) fiasco_amd_release_memory();
    if (hipSetDevice(device) != hipSuccess) {
        fa_set_error("libfiasco_amd: cannot select HIP device %d", device);
        return 0;
    }
    pthread_mutex_lock(&g_dev_lock);
    g_device_explicit = device;
    g_devices.clear();                              /* resolved again by the next call */
    pthread_mutex_unlock(&g_dev_lock);
    return 1;
}

extern "C" void fiasco_amd_release_memory(void)
{
    int cur = -1;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (size_t k = 0; k < g_dev_state.size(); k++) {
        DevState *st = g_dev_state[k];
        if (st->free.empty()) continue;
        if (k < g_devices.size() && g_devices[k] >= 0 && g_devices.size() > 1) (void) hipSetDevice(g_devices[k]);
        for (size_t i = 0; i < st->free.size(); i++) (void) hipFree(st->free[i].base);
        st->free.clear();
    }
    if (g_dev_state.empty()) {
        for (size_t i = 0; i < g_state0.free.size(); i++) (void) hipFree(g_state0.free[i].base);
        g_state0.free.clear();
    }
    if (have_cur && g_devices.size() > 1) (void) hipSetDevice(cur);
}

/* counters: the sum over the shares; kernel time and the largest automaton: the maximum (the shares run
 * side by side, frames / kernel_ms stays the rate of the whole job) */
extern "C" void fiasco_amd_get_stats(fiasco_amd_stats *out)
{
    *out = g_state0.stats;
    for (size_t k = 1; k < g_dev_state.size(); k++) {
        const fiasco_amd_stats &b = g_dev_state[k]->stats;
        unsigned long long *o = (unsigned long long *) ((char *) out + sizeof(double));
        const unsigned long long *v = (const unsigned long long *) ((const char *) &b + sizeof(double));
        const size_t nw = (sizeof(fiasco_amd_stats) - sizeof(double)) / sizeof(unsigned long long);
        const size_t imax = (offsetof(fiasco_amd_stats, states_max) - sizeof(double)) / sizeof(unsigned long long);
        /* workgroups per frame of the table passes: a setting, the same on every share -- not a sum */
        const size_t icoop = (offsetof(fiasco_amd_stats, coop_workgroups) - sizeof(double)) / sizeof(unsigned long long);
        for (size_t i = 0; i < nw; i++) o[i] = i == imax || i == icoop ? (o[i] > v[i] ? o[i] : v[i]) : o[i] + v[i];
        if (b.kernel_ms > out->kernel_ms) out->kernel_ms = b.kernel_ms;
    }
}
extern "C" void fiasco_amd_reset_stats(void)
{
    memset(&g_state0.stats, 0, sizeof g_state0.stats);
    for (size_t k = 1; k < g_dev_state.size(); k++) memset(&g_dev_state[k]->stats, 0, sizeof(fiasco_amd_stats));
}

struct MultiStaged {
    unsigned n = 0;
    fa_job  *jobs = nullptr;
    struct Part { std::vector<unsigned> idx; std::vector<fa_job> sub; void *staged = nullptr; int good = 0;
                  size_t share = 0;   /* the device share (g_devices / g_dev_state index) this part runs on */ };
    std::vector<Part> parts;        /* one part: parts[0].staged works on jobs[] itself, nothing is copied */
    char  *up_host = nullptr;       /* several shares: the pinned buffer of fa_core_upload_buffer (the shares borrow it) */
    size_t up_host_bytes = 0;
};

/* One persistent host thread per share k >= 1 (created at its first use, parked on a condition variable between
 * calls): a phase of a batch -- stage, submit, finish, upload -- posts its share of the work there instead of
 * creating and joining a thread each time.  A worker binds itself to the device of its share, g_devices[k], at
 * the start of every task (the list may have changed since) and works on the share's DevState. */
struct ShareWorker {
    pthread_t th;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    void (*call)(void *, size_t) = nullptr;
    void *ctx = nullptr;
    size_t k = 0;
    int state = 0;                  /* 0 idle, 1 task posted, 2 task done */
};
static std::vector<ShareWorker *> g_workers;       /* [k], k >= 1; [0] unused */
static pthread_mutex_t g_share_lock = PTHREAD_MUTEX_INITIALIZER;    /* one multi-share phase at a time (see above) */

static void bind_share(size_t k)
{
    t_dev = k < g_dev_state.size() ? g_dev_state[k] : &g_state0;
    if (k < g_devices.size() && g_devices[k] >= 0) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != g_devices[k]) { (void) hipGetLastError(); (void) hipSetDevice(g_devices[k]); }
    }
}

static void *share_worker_main(void *p)
{
    ShareWorker *w = (ShareWorker *) p;
    for (;;) {
        pthread_mutex_lock(&w->mu);
        while (w->state != 1) pthread_cond_wait(&w->cv, &w->mu);
        pthread_mutex_unlock(&w->mu);
        bind_share(w->k);
        w->call(w->ctx, w->k);                 /* ctx names the part of the batch (for_shares) */
        pthread_mutex_lock(&w->mu);
        w->state = 2;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
    }
    return nullptr;
}

static ShareWorker *share_worker(size_t k)
{
    pthread_mutex_lock(&g_dev_lock);
    while (g_workers.size() <= k) g_workers.push_back(nullptr);
    ShareWorker *w = g_workers[k];
    if (!w) {
        w = new ShareWorker;
        w->k = k;
        if (pthread_create(&w->th, nullptr, share_worker_main, w) != 0) { delete w; w = nullptr; }
        else { (void) pthread_detach(w->th); g_workers[k] = w; }
    }
    pthread_mutex_unlock(&g_dev_lock);
    return w;
}

/* run fn(share) for every share: share 0 on the calling thread, the others on their workers; EVERY share --
 * also the only one of a call -- runs bound to its device g_devices[k] with the DevState of that share (the slab
 * pool of a share never sees another device), and the caller's current device is what it was afterwards */
template <typename Fn> static void for_shares(const std::vector<size_t> &share, Fn fn)
{
    const size_t D = share.size();
    int cur = -1;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (!have_cur) (void) hipGetLastError();
    /* the workers' call / ctx / state slots are per process: two host threads driving two multi-share batches
     * would overwrite each other's task (a lost task, or a wait for `state == 2' that never ends) */
    if (D > 1) pthread_mutex_lock(&g_share_lock);
    struct Task { Fn *fn; size_t part; };
    std::vector<Task> task(D);
    auto tramp = [](void *p, size_t) { Task *t = (Task *) p; (*t->fn)(t->part); };
    std::vector<ShareWorker *> posted(D, nullptr);
    for (size_t k = 1; k < D; k++) {
        /* part k on the worker of ITS share; two parts of one share (never dealt that way) would run one after the other */
        bool dup = false;
        for (size_t j = 0; j < k; j++) dup = dup || share[j] == share[k];
        ShareWorker *w = dup ? nullptr : share_worker(share[k]);
        if (!w) continue;
        task[k].fn = &fn; task[k].part = k;
        pthread_mutex_lock(&w->mu);
        w->call = tramp; w->ctx = &task[k]; w->state = 1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        posted[k] = w;
    }
    if (D) { bind_share(share[0]); fn(0); }
    for (size_t k = 1; k < D; k++) {
        if (posted[k]) {
            ShareWorker *w = posted[k];
            pthread_mutex_lock(&w->mu);
            while (w->state != 2) pthread_cond_wait(&w->cv, &w->mu);
            w->state = 0;
            pthread_mutex_unlock(&w->mu);
        } else { bind_share(share[k]); fn(k); }      /* no thread: one after the other */
    }
    if (D > 1) pthread_mutex_unlock(&g_share_lock);
    t_dev = &g_state0;
    if (have_cur) {
        int now = -1;
        if (hipGetDevice(&now) != hipSuccess || now != cur) { (void) hipGetLastError(); (void) hipSetDevice(cur); }
    }
}

/* fn(part) for every part of a staged batch, each on the share it was dealt to */
template <typename Fn> static void for_each_share(MultiStaged *M, Fn fn)
{
    std::vector<size_t> share(M->parts.size());
    for (size_t k = 0; k < share.size(); k++) share[k] = M->parts[k].share;
    for_shares(share, fn);
}

extern "C" void *fa_core_stage(unsigned n, fa_job *jobs)
{
    resolve_devices();
    MultiStaged *M = new MultiStaged;
    M->n = n; M->jobs = jobs;
    size_t D = g_devices.size();
    bool keyed = false;
    for (unsigned i = 0; i < n; i++) keyed = keyed || jobs[i].share_key != 0;
    /* jobs without a key: round robin over as many shares as there are jobs.  Jobs with a key (the GOP
     * of a video, fa_host.h fa_share_of): the share is a function of the key and of the number of devices ALONE -- not
     * of how many jobs this call happens to hold --, shares without a job get no part */
    if (!keyed && D > n) D = n ? n : 1;
    if (D == 1) { M->parts.resize(1); for_each_share(M, [&](size_t) { M->parts[0].staged = core1_stage(n, jobs); }); return M; }
    {
        std::vector<MultiStaged::Part> all(D);
        for (unsigned i = 0; i < n; i++) all[fa_share_of(jobs[i].share_key, i, (unsigned) D)].idx.push_back(i);
        size_t used = 0, only = 0;
        for (size_t k = 0; k < D; k++) if (!all[k].idx.empty()) { used++; only = k; }
        if (used <= 1) {
            /* every job on one share (the last GOPs of a video): ONE part that works on jobs[] itself, on that share */
            M->parts.resize(1);
            M->parts[0].share = used ? only : 0;
            for_each_share(M, [&](size_t) { M->parts[0].staged = core1_stage(n, jobs); });
            return M;
        }
        for (size_t k = 0; k < D; k++)
            if (!all[k].idx.empty()) { all[k].share = k; M->parts.push_back(all[k]); }
    }
    D = M->parts.size();
    for (size_t k = 0; k < D; k++) {
        MultiStaged::Part &P = M->parts[k];
        P.sub.resize(P.idx.size());
        for (size_t j = 0; j < P.idx.size(); j++) P.sub[j] = jobs[P.idx[j]];
    }
    for_each_share(M, [&](size_t k) { MultiStaged::Part &P = M->parts[k]; P.staged = core1_stage((unsigned) P.sub.size(), P.sub.data()); });
    for (size_t k = 0; k < D; k++)                                              /* what staging said about a job */
        for (size_t j = 0; j < M->parts[k].idx.size(); j++) jobs[M->parts[k].idx[j]] = M->parts[k].sub[j];
    return M;
}

extern "C" void fa_core_unstage(void *h)
{
    MultiStaged *M = (MultiStaged *) h;
    if (!M) return;
    for_each_share(M, [&](size_t k) { core1_unstage(M->parts[k].staged); });
    if (M->up_host) (void) hipHostFree(M->up_host);
    delete M;
}

extern "C" int fa_core_submit(void *h)
{
    MultiStaged *M = (MultiStaged *) h;
    if (!M) return 0;
    if (M->parts.size() == 1) { int r = 0; for_each_share(M, [&](size_t) { r = core1_submit(M->parts[0].staged); }); return r; }
    int ok = 1;
    for (size_t k = 0; k < M->parts.size(); k++)                               /* inputs as the caller has them now */
        for (size_t j = 0; j < M->parts[k].idx.size(); j++) {
            fa_job &dst = M->parts[k].sub[j];
            const fa_job &src = M->jobs[M->parts[k].idx[j]];
            dst.image = src.image; dst.frame_type = src.frame_type; dst.past = src.past; dst.future = src.future;
            dst.cp = src.cp; dst.wfa = src.wfa; dst.ycol_carry = src.ycol_carry;
        }
    for_each_share(M, [&](size_t k) { M->parts[k].good = core1_submit(M->parts[k].staged); });
    for (size_t k = 0; k < M->parts.size(); k++) ok = ok && M->parts[k].good;
    return ok;
}

extern "C" int fa_core_finish2(void *h, int resubmit)
{
    MultiStaged *M = (MultiStaged *) h;
    if (!M) return 0;
    if (M->parts.size() == 1) { int r = 0; for_each_share(M, [&](size_t) { r = core1_finish2(M->parts[0].staged, resubmit); }); return r; }
    for_each_share(M, [&](size_t k) { M->parts[k].good = core1_finish2(M->parts[k].staged, resubmit); });
    int good = 0;
    for (size_t k = 0; k < M->parts.size(); k++) {
        good += M->parts[k].good;
        for (size_t j = 0; j < M->parts[k].idx.size(); j++) M->jobs[M->parts[k].idx[j]] = M->parts[k].sub[j];
    }
    return good;
}

extern "C" int fa_core_finish(void *h) { return fa_core_finish2(h, 0); }

extern "C" int fa_core_run(void *h)
{
    if (!fa_core_submit(h)) return 0;
    return fa_core_finish(h);
}

/* replacement inputs for a staged batch (a stream of batches over PCIe).  The caller fills ONE pinned buffer with
 * the planes of all frames; every share then copies the planes of ITS frames to its device (core1_upload_commit).
 * With several shares the buffer belongs to the batch (portable pinned memory: every device reads it). */
extern "C" int16_t *fa_core_upload_buffer(void *h, size_t bytes)
{
    MultiStaged *M = (MultiStaged *) h;
    if (!M) return nullptr;
    if (M->parts.size() == 1) {
        int16_t *r = nullptr;
        for_each_share(M, [&](size_t) { r = core1_upload_buffer(M->parts[0].staged, bytes); });
        return r;
    }
    /* the previous uploads have left the buffer long ago (a whole pass lies in between); make sure */
    for_each_share(M, [&](size_t k) { Staged *S = (Staged *) M->parts[k].staged; if (S && S->ustream) (void) hipStreamSynchronize(S->ustream); });
    if (bytes > M->up_host_bytes) {
        if (M->up_host) (void) hipHostFree(M->up_host);
        M->up_host = nullptr; M->up_host_bytes = 0;
        if (hipHostMalloc((void **) &M->up_host, bytes, hipHostMallocPortable) != hipSuccess) {
            M->up_host = nullptr; (void) hipGetLastError();
            /* the shares still point into the buffer that was just freed: a later commit must not bounds-check
             * against it or copy from it */
            for (size_t k = 0; k < M->parts.size(); k++) {
                Staged *S = (Staged *) M->parts[k].staged;
                if (S && S->up_host_shared) { S->up_host = nullptr; S->up_host_bytes = 0; S->up_host_shared = false; }
            }
            return nullptr;
        }
        M->up_host_bytes = bytes;
    }
    for (size_t k = 0; k < M->parts.size(); k++) {
        Staged *S = (Staged *) M->parts[k].staged;
        if (!S || !S->ok) return nullptr;
        if (S->up_host && !S->up_host_shared) (void) hipHostFree(S->up_host);
        S->up_host = M->up_host; S->up_host_bytes = M->up_host_bytes; S->up_host_shared = true;
    }
    return (int16_t *) M->up_host;
}

extern "C" int fa_core_upload_commit(void *h)
{
    MultiStaged *M = (MultiStaged *) h;
    if (!M) return 0;
    if (M->parts.size() == 1) { int r = 0; for_each_share(M, [&](size_t) { r = core1_upload_commit(M->parts[0].staged); }); return r; }
    for (size_t k = 0; k < M->parts.size(); k++)                               /* the new images of the caller's jobs */
        for (size_t j = 0; j < M->parts[k].idx.size(); j++) M->parts[k].sub[j].image = M->jobs[M->parts[k].idx[j]].image;
    for_each_share(M, [&](size_t k) { M->parts[k].good = core1_upload_commit(M->parts[k].staged); });
    int ok = 1;
    for (size_t k = 0; k < M->parts.size(); k++) ok = ok && M->parts[k].good;
    return ok;
}

extern "C" int fa_core_encode_frames(unsigned n, fa_job *jobs)
{
    void *h = fa_core_stage(n, jobs);
    int good = fa_core_run(h);
    fa_core_unstage(h);
    return good;
}

/* ------------------------------------------------------------------ gather of the streams over RCCL
 *
 * One process per GPU (BASELINE config 4): every rank encodes its share of the frames -- frame i of the
 * job on rank i mod W -- and the finished byte strings (kilobytes per frame) meet on one rank: two small all-gathers for
 * the counts and lengths, one padded all-gather for the payloads, over xGMI.  No data-path collective exists; this is
 * the only communication of the job.  RCCL is NOT a link-time dependency of the library: the entry points are taken
 * from the copy the process already has (the one that made the caller's communicator), else from librccl.so. */
#include <dlfcn.h>
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, void *);
typedef const char *(*nccl_errstr_fn)(int);
enum { FA_NCCL_UINT8 = 1, FA_NCCL_UINT64 = 5 };          /* ncclDataType_t, rccl.h */

extern "C" int fiasco_amd_rccl_gather(void *comm, void *stream_, int rank, int world, int root,
                                      unsigned n_local, const unsigned char *const *data, const size_t *len,
                                      unsigned char ***all, size_t **all_len, unsigned *n_all)
{
    hipStream_t stream = (hipStream_t) stream_;
    if (all) *all = nullptr;
    if (all_len) *all_len = nullptr;
    if (n_all) *n_all = 0;
    if (!comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || (n_local && (!data || !len))) {
        fa_set_error("fiasco_amd_rccl_gather: bad arguments");
        return 0;
    }
    static nccl_allgather_fn allgather = nullptr;
    static nccl_errstr_fn errstr = nullptr;
    if (!allgather) {
        allgather = (nccl_allgather_fn) dlsym(RTLD_DEFAULT, "ncclAllGather");
        errstr = (nccl_errstr_fn) dlsym(RTLD_DEFAULT, "ncclGetErrorString");
        if (!allgather) {
            void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (h) { allgather = (nccl_allgather_fn) dlsym(h, "ncclAllGather"); errstr = (nccl_errstr_fn) dlsym(h, "ncclGetErrorString"); }
        }
        if (!allgather) { fa_set_error("fiasco_amd_rccl_gather: no RCCL in this process (librccl.so)"); return 0; }
    }
    /* Failure discipline: a collective that one rank skips hangs every other rank.  So every rank takes part in
     * every collective the OTHERS will enter: a rank-local failure travels as a flag in the next message -- word 2 of
     * the counts, then a status round after the payload buffers have been allocated (whose size no rank knows before
     * the counts are in) -- and all ranks, looking at the same gathered words, fail TOGETHER before the all-gather of
     * the streams.  The one exception is the first allocation (24 (W + 1) bytes): a rank that cannot get that cannot
     * signal anything.  The root's return value is the job's; a rank that is not the root returns 1 once its part is
     * delivered. */
    unsigned long long *d_u64 = nullptr;
    unsigned char *d_pay = nullptr;
    int ok = 1, rc = 0;
    const size_t W = (size_t) world;
    std::vector<unsigned long long> h_cnt(W * 3), h_st(W * 3);
#define GCHECK(call, what) do { if ((call) != hipSuccess) { if (ok) fa_set_error("fiasco_amd_rccl_gather: %s: %s", what, hipGetErrorString(hipGetLastError())); ok = 0; } } while (0)
#define NCHECK(call, what) do { if ((rc = (call)) != 0) { if (ok) fa_set_error("fiasco_amd_rccl_gather: %s: %s", what, errstr ? errstr(rc) : "RCCL error"); ok = 0; } } while (0)
    /* 1. counts, total bytes and failure flag of every rank */
    unsigned long long mine[3] = { n_local, 0, 0 };
    for (unsigned i = 0; i < n_local; i++) mine[1] += len[i];
    if (hipMalloc((void **) &d_u64, sizeof(unsigned long long) * 3 * (W + 1)) != hipSuccess) {
        fa_set_error("fiasco_amd_rccl_gather: hipMalloc: %s", hipGetErrorString(hipGetLastError()));
        return 0;
    }
    GCHECK(hipMemcpyAsync(d_u64 + 3 * W, mine, sizeof mine, hipMemcpyHostToDevice, stream), "upload");
    NCHECK(allgather(d_u64 + 3 * W, d_u64, 3, FA_NCCL_UINT64, comm, stream), "all-gather of the counts");
    GCHECK(hipMemcpyAsync(h_cnt.data(), d_u64, sizeof(unsigned long long) * 3 * W, hipMemcpyDeviceToHost, stream), "download");
    GCHECK(hipStreamSynchronize(stream), "synchronize");
    size_t maxn = 0, maxb = 0, total = 0;
    if (ok) {
        for (size_t r = 0; r < W; r++) {
            if (h_cnt[3 * r] > maxn) maxn = (size_t) h_cnt[3 * r];
            if (h_cnt[3 * r + 1] > maxb) maxb = (size_t) h_cnt[3 * r + 1];
            total += (size_t) h_cnt[3 * r];
        }
        /* the deal must be round robin (item i on rank i mod W): rank r holds ceil((total - r) / W) streams -- anything
         * else (or garbage from a rank whose upload failed) would be put in the wrong places below; every rank sees the
         * same words and fails alike */
        for (size_t r = 0; r < W; r++) {
            const size_t want = total > r ? (total - r + W - 1) / W : 0;
            if ((size_t) h_cnt[3 * r] != want || h_cnt[3 * r + 2] != 0) {
                fa_set_error(h_cnt[3 * r + 2] ? "fiasco_amd_rccl_gather: rank %d reported a failure"
                                              : "fiasco_amd_rccl_gather: rank %d holds %llu of %llu streams: the frames were not dealt round robin",
                             (int) r, (unsigned long long) h_cnt[3 * r], (unsigned long long) total);
                ok = 0;
                break;
            }
        }
    }
    /* 2. per rank: maxn lengths + maxb payload bytes (padded).  Buffers first, then a status round: nobody enters the
     * big all-gather unless everybody can (a rank whose step 1 failed locally reports that here too) */
    const size_t slot = ok ? align_up(maxn * 8 + maxb, 16) : 0;
    std::vector<unsigned char> h_send, h_recv;
    unsigned long long st[3] = { ok ? 0ull : 1ull, 0, 0 };
    if (ok && slot) {
        try { h_send.assign(slot, 0); if (rank == root) h_recv.resize(slot * W); } catch (...) { st[0] = 1; }
        if (!st[0] && hipMalloc((void **) &d_pay, slot * (W + 1)) != hipSuccess) { (void) hipGetLastError(); d_pay = nullptr; st[0] = 1; }
        if (st[0]) { fa_set_error("fiasco_amd_rccl_gather: out of memory for %zu bytes per rank", slot); ok = 0; }
    }
    {
        int ok2 = 1;                                   /* the status round itself; `ok' keeps the first message */
        if (hipMemcpyAsync(d_u64 + 3 * W, st, sizeof st, hipMemcpyHostToDevice, stream) != hipSuccess) ok2 = 0;
        if (allgather(d_u64 + 3 * W, d_u64, 3, FA_NCCL_UINT64, comm, stream) != 0) ok2 = 0;
        if (hipMemcpyAsync(h_st.data(), d_u64, sizeof(unsigned long long) * 3 * W, hipMemcpyDeviceToHost, stream) != hipSuccess) ok2 = 0;
        if (hipStreamSynchronize(stream) != hipSuccess) ok2 = 0;
        if (!ok2) { (void) hipGetLastError(); if (ok) fa_set_error("fiasco_amd_rccl_gather: the status round failed"); ok = 0; }
        for (size_t r = 0; ok2 && r < W; r++)
            if (h_st[3 * r]) { if (ok) fa_set_error("fiasco_amd_rccl_gather: rank %d cannot take part (see its message)", (int) r); ok = 0; break; }
    }
    if (ok && slot) {
        size_t o = maxn * 8;
        for (unsigned i = 0; i < n_local; i++) {
            const unsigned long long l = len[i];
            memcpy(h_send.data() + (size_t) i * 8, &l, 8);
            memcpy(h_send.data() + o, data[i], len[i]);
            o += len[i];
        }
        GCHECK(hipMemcpyAsync(d_pay + slot * W, h_send.data(), slot, hipMemcpyHostToDevice, stream), "upload");
        NCHECK(allgather(d_pay + slot * W, d_pay, slot, FA_NCCL_UINT8, comm, stream), "all-gather of the streams");
        if (rank == root) GCHECK(hipMemcpyAsync(h_recv.data(), d_pay, slot * W, hipMemcpyDeviceToHost, stream), "download");
        GCHECK(hipStreamSynchronize(stream), "synchronize");
    }
    if (d_pay) (void) hipFree(d_pay);
    if (d_u64) (void) hipFree(d_u64);
#undef GCHECK
#undef NCHECK
    if (!ok) return 0;
    if (rank != root || !all || !all_len || !n_all) return 1;
    /* 3. the root: stream k of rank r is item r + k * world of the job (the round-robin deal, checked above) */
    unsigned char **out = (unsigned char **) calloc(total ? total : 1, sizeof *out);
    size_t *olen = (size_t *) calloc(total ? total : 1, sizeof *olen);
    int oom = !out || !olen;
    for (size_t r = 0; !oom && r < W; r++) {
        const unsigned char *base = h_recv.data() + slot * r;
        size_t o = maxn * 8;
        for (size_t k = 0; !oom && k < (size_t) h_cnt[3 * r]; k++) {
            unsigned long long l;
            memcpy(&l, base + k * 8, 8);
            const size_t item = r + k * W;               /* < total: the deal was checked */
            out[item] = (unsigned char *) malloc(l ? (size_t) l : 1);
            if (!out[item]) { oom = 1; break; }
            memcpy(out[item], base + o, (size_t) l);
            olen[item] = (size_t) l;
            o += (size_t) l;
        }
    }
    if (oom) {
        if (out) for (size_t i = 0; i < total; i++) free(out[i]);
        free(out); free(olen);
        fa_set_error("fiasco_amd_rccl_gather: out of memory");
        return 0;
    }
    *all = out; *all_len = olen; *n_all = (unsigned) total;
    return 1;
}

#include "frame_decoder.inc"
