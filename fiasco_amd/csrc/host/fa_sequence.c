/*
 *  fa_sequence.c -- the frames of a video in coding order, groups of pictures in parallel.
 *
 *  video_coder (reference codec/coder.c:490-668) codes the frames of a sequence one after the
 *  other: a P/B frame is predicted from the reconstruction of the frames coded before it
 *  (:580-651).  An I frame drops every reference frame (:581-592), so the frames from one I frame
 *  up to the next -- a group of pictures, GOP -- depend on nothing before them, with two
 *  exceptions that exist for colour streams only:
 *    * options.lc_min_level is raised while the Cb band of a frame is coded and never lowered
 *      (:785-797): every colour frame hands one integer to the next one;
 *    * wfa->y_column survives remove_states (codec/wfalib.c:277-310): the three states that join
 *      the bands of a colour frame show the flags an earlier state with the same id left behind,
 *      in this frame or in any frame before it.  Only the stream writer reads them.
 *  So the partition search of all GOPs runs side by side -- frame j of every GOP is one batch for
 *  the core (one persistent workgroup each on the device) -- with the minimum level at the start
 *  of a GOP SPECULATED (it only ever goes up and usually settles in the first frame) and verified
 *  against what the GOP before it left; GOPs that started from a wrong value are searched again.
 *  The y_column chain is resolved afterwards, frame by frame, when the streams are written: the
 *  core starts every frame from a sentinel and only the entries it did not touch are inherited.
 *
 *  With rank / world a process searches and writes only every world-th GOP; what the ranks have to
 *  exchange is small (one integer per GOP, one flag array per frame, the finished byte strings)
 *  and is left to the caller (fiasco_amd/sharding.py does it with torch.distributed).
 */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
#include "fa_host.h"

/* developer aid: FIASCO_AMD_SEQ_TIMING=1 prints where a sweep spends its time (stderr) */
static double seq_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

#define YCOL_UNSET 2          /* y_column entry the frame did not write */

struct fa_seq {
    const fa_options *op;
    fa_info    wi;
    fa_cparams cp;
    unsigned   nframes;
    const unsigned char *const *bufs;     /* raw PNM of every frame (display order), borrowed */
    const size_t *lens;
    char const *const *names;
    int        color;
    unsigned   ncoded;                    /* frames in coding order */
    unsigned  *order;                     /* display number */
    int       *type, *isfut;
    unsigned   ngop, *gfirst;             /* GOP g = coded frames [gfirst[g], gfirst[g+1]) */
    unsigned   rank, world;
    unsigned   cap2;                      /* entries of a y_column array */
    fa_wfa   **wfa;                       /* per coded frame: finished automaton (own GOPs) */
    fa_stats (*stats)[3];
    unsigned  *carry_used, *carry_out;    /* per GOP: minimum level it started from / left */
    uint8_t   *gdone, *gfail;
    char     (*gerr)[160];
    /* the probe frame (fa_seq_probe) is the first frame of GOP 0 searched from the stream's own starting level:
     * the sweep takes it over instead of searching it again (the slowest frame of its step: the other GOPs start
     * from the ratcheted level) */
    fa_wfa    *probe_wfa;
    fa_stats   probe_stats[3];
    unsigned   probe_out;
};

static int frame_type_of(unsigned display, const char *pattern, int *ok)
{
    int c = toupper((unsigned char) pattern[display % strlen(pattern)]);
    *ok = 1;
    if (display == 0 || c == 'I') return FA_I_FRAME;    /* the first frame is always intra (:522-523) */
    if (c == 'P') return FA_P_FRAME;
    if (c == 'B') return FA_B_FRAME;
    fa_set_error("Frame type %c not valid. Choose one of I,B or P.", c);
    *ok = 0;
    return FA_I_FRAME;
}

/* The order in which video_coder codes the frames and the type it gives each: a run of B frames
 * is preceded by its future reference, which is the next non-B frame of the pattern or -- at the
 * end of the sequence -- the last frame, coded as a P frame (codec/coder.c:535-571). */
static int coding_order(unsigned nframes, const char *pattern, unsigned *order, int *type, int *is_future)
{
    unsigned display = 0, k = 0;
    int future_display = -1, ok;
    while (display < nframes) {
        int t = frame_type_of(display, pattern, &ok);
        unsigned frame;
        if (!ok) return 0;
        if ((int) display == future_display) { display++; continue; }
        if (t == FA_B_FRAME && (int) display > future_display) {
            unsigned i = display;
            while (t == FA_B_FRAME) {
                i++;
                if (i >= nframes) { future_display = (int) i - 1; t = FA_P_FRAME; }
                else {
                    future_display = (int) i;
                    t = frame_type_of(i, pattern, &ok);
                    if (!ok) return 0;
                }
            }
            frame = (unsigned) future_display;
        } else {
            frame = display;
            display++;
        }
        order[k] = frame; type[k] = t; is_future[k] = (int) frame == future_display;
        k++;
    }
    return (int) k;
}

void fa_seq_free(fa_seq *s)
{
    unsigned k;
    if (!s) return;
    for (k = 0; s->wfa && k < s->ncoded; k++) fa_wfa_free(s->wfa[k]);
    fa_wfa_free(s->probe_wfa);
    free(s->wfa); free(s->stats); free(s->order); free(s->type); free(s->isfut); free(s->gfirst);
    free(s->carry_used); free(s->carry_out); free(s->gdone); free(s->gfail); free(s->gerr);
    fa_info_free(&s->wi);
    free(s);
}

fa_seq *fa_seq_open(const fa_options *op, float quality, unsigned nframes, const unsigned char *const *bufs,
                    const size_t *lens, char const *const *names, unsigned rank, unsigned world)
{
    fa_seq *s = (fa_seq *) calloc(1, sizeof *s);
    unsigned i, w = 0, h = 0;
    int n;
    if (!s) { fa_set_error("Out of memory!"); return NULL; }
    s->op = op; s->nframes = nframes; s->bufs = bufs; s->lens = lens; s->names = names;
    s->rank = rank; s->world = world ? world : 1;
    for (i = 0; i < nframes; i++) {
        unsigned fw, fh; int fc; size_t off;
        if (!fa_pnm_header(bufs[i], lens[i], names ? names[i] : "<memory>", &fw, &fh, &fc, &off)) goto bad;
        if (i == 0) { w = fw; h = fh; s->color = fc; }
        else if (fw != w || fh != h) {
            fa_set_error("`%s': all images of a sequence have to be of the same size.", names ? names[i] : "<memory>");
            goto bad;
        } else if (fc != s->color) {
            fa_set_error("`%s': all images a sequence have to use the same color model.", names ? names[i] : "<memory>");
            goto bad;
        }
    }
    if (!nframes) { fa_set_error("Can't open frame `%s'.", "<none>"); goto bad; }
    if (!fa_setup_params(op, quality, w, h, s->color, nframes, &s->wi, &s->cp)) goto bad;
    s->order = (unsigned *) calloc(nframes, sizeof *s->order);
    s->type = (int *) calloc(nframes, sizeof *s->type);
    s->isfut = (int *) calloc(nframes, sizeof *s->isfut);
    s->gfirst = (unsigned *) calloc(nframes + 1, sizeof *s->gfirst);
    if (!s->order || !s->type || !s->isfut || !s->gfirst) { fa_set_error("Out of memory!"); goto bad; }
    n = coding_order(nframes, op->pattern, s->order, s->type, s->isfut);
    if (n <= 0) goto bad;
    s->ncoded = (unsigned) n;
    for (i = 0; i < s->ncoded; i++)
        if (s->type[i] == FA_I_FRAME) s->gfirst[s->ngop++] = i;
    s->gfirst[s->ngop] = s->ncoded;
    s->cap2 = s->cp.limit_states * 2;
    s->wfa = (fa_wfa **) calloc(s->ncoded, sizeof *s->wfa);
    s->stats = (fa_stats (*)[3]) calloc(s->ncoded, sizeof *s->stats);
    s->carry_used = (unsigned *) calloc(s->ngop, sizeof *s->carry_used);
    s->carry_out = (unsigned *) calloc(s->ngop, sizeof *s->carry_out);
    s->gdone = (uint8_t *) calloc(s->ngop, 1);
    s->gfail = (uint8_t *) calloc(s->ngop, 1);
    s->gerr = (char (*)[160]) calloc(s->ngop, sizeof *s->gerr);
    if (!s->wfa || !s->stats || !s->carry_used || !s->carry_out || !s->gdone || !s->gfail || !s->gerr) {
        fa_set_error("Out of memory!");
        goto bad;
    }
    return s;
bad:
    fa_seq_free(s);
    return NULL;
}

unsigned fa_seq_gops(const fa_seq *s) { return s->ngop; }
unsigned fa_seq_frames(const fa_seq *s) { return s->ncoded; }
unsigned fa_seq_ycol_size(const fa_seq *s) { return s->cap2; }
unsigned fa_seq_initial_level(const fa_seq *s) { return s->cp.lc_min_level; }
int fa_seq_is_mine(const fa_seq *s, unsigned gop) { return gop % s->world == s->rank; }
const fa_info *fa_seq_info(const fa_seq *s) { return &s->wi; }
unsigned fa_seq_gop_of(const fa_seq *s, unsigned k)
{
    unsigned g = 0;
    while (g + 1 < s->ngop && s->gfirst[g + 1] <= k) g++;
    return g;
}
void fa_seq_gop_result(const fa_seq *s, unsigned gop, unsigned *carry_out, int *failed)
{
    if (carry_out) *carry_out = s->carry_out[gop];
    if (failed) *failed = s->gfail[gop];
}
const char *fa_seq_gop_error(const fa_seq *s, unsigned gop) { return s->gerr[gop]; }
const fa_stats *fa_seq_stats(const fa_seq *s, unsigned k) { return s->stats[k]; }

/* per-GOP running state of a sweep */
typedef struct gop_run {
    unsigned g, pos, n;                    /* GOP, next frame inside it, frames */
    fa_image *reconst, *past, *future;
    int last_was_future, dead;
    unsigned carry;
} gop_run;

/* parsing the inputs of one step: share `first, first + stride, ...' of the running GOPs */
#define PARSE_THREADS 16
typedef struct parsed { fa_image *im; char err[160]; } parsed;
typedef struct parse_share {
    fa_seq *s;
    const gop_run *run;
    parsed *pre;
    unsigned nrun, step, first, stride;
} parse_share;
static void *parse_thread(void *arg)
{
    parse_share *p = (parse_share *) arg;
    fa_seq *s = p->s;
    unsigned r;
    for (r = p->first; r < p->nrun; r += p->stride) {
        const gop_run *q = &p->run[r];
        unsigned k;
        p->pre[r].im = NULL; p->pre[r].err[0] = 0;
        if (q->dead || p->step >= q->n) continue;
        k = s->gfirst[q->g] + p->step;
        p->pre[r].im = fa_image_from_pnm(s->bufs[s->order[k]], s->lens[s->order[k]], s->names ? s->names[s->order[k]] : "<memory>");
        if (!p->pre[r].im) snprintf(p->pre[r].err, sizeof p->pre[r].err, "%s", fiasco_get_error_message());   /* this thread's message */
    }
    return NULL;
}

/* Partition search of the GOPs of this rank marked in todo[], each starting from carry_in[g].
 * Frame j of all of them is one batch for the core.  Returns 0 only on an internal error (out of
 * memory); a GOP whose search fails is recorded (fa_seq_gop_result) -- whether that is an error
 * of the stream is known once its starting value has been verified. */
int fa_seq_search(fa_seq *s, const unsigned *carry_in, const uint8_t *todo)
{
    const int video = s->ncoded > s->ngop;            /* some frame is not intra */
    gop_run *run = (gop_run *) calloc(s->ngop ? s->ngop : 1, sizeof *run);
    fa_job *jobs = (fa_job *) calloc(s->ngop ? s->ngop : 1, sizeof *jobs);
    fa_image **ims = (fa_image **) calloc(s->ngop ? s->ngop : 1, sizeof *ims);
    unsigned *who = (unsigned *) calloc(s->ngop ? s->ngop : 1, sizeof *who);
    uint8_t *need = (uint8_t *) calloc(s->ngop ? s->ngop : 1, 1);
    fa_dec_job *djobs = (fa_dec_job *) calloc(s->ngop ? s->ngop : 1, sizeof *djobs);
    parsed *pre = (parsed *) calloc(s->ngop ? s->ngop : 1, sizeof *pre);
    unsigned nrun = 0, g, r, step, maxlen = 0;
    int rc = 0;
    double t_prep = 0, t_core = 0, t_dec = 0, t0;
    if (!run || !jobs || !ims || !who || !need || !djobs || !pre) { fa_set_error("Out of memory!"); goto out; }
    for (g = 0; g < s->ngop; g++) {
        unsigned k;
        if (!fa_seq_is_mine(s, g) || !todo[g]) continue;
        for (k = s->gfirst[g]; k < s->gfirst[g + 1]; k++) { fa_wfa_free(s->wfa[k]); s->wfa[k] = NULL; }
        run[nrun].g = g; run[nrun].n = s->gfirst[g + 1] - s->gfirst[g]; run[nrun].carry = carry_in[g];
        s->carry_used[g] = carry_in[g]; s->gdone[g] = 0; s->gfail[g] = 0; s->gerr[g][0] = 0;
        if (run[nrun].n > maxlen) maxlen = run[nrun].n;
        nrun++;
    }
    for (step = 0; step < maxlen; step++) {
        t0 = seq_now();
        unsigned nb = 0, b, predone = 0;
        /* the inputs of this step: PNM -> planes of every GOP's frame side by side on the host's cores */
        {
            parse_share sh[PARSE_THREADS];
            pthread_t th[PARSE_THREADS];
            int started[PARSE_THREADS] = { 0 };
            long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
            unsigned nt = ncpu > PARSE_THREADS ? PARSE_THREADS : ncpu < 1 ? 1 : (unsigned) ncpu, t;
            if (nt > nrun) nt = nrun ? nrun : 1;
            for (t = 0; t < nt; t++) { sh[t].s = s; sh[t].run = run; sh[t].pre = pre; sh[t].nrun = nrun; sh[t].step = step; sh[t].first = t; sh[t].stride = nt; }
            for (t = 1; t < nt; t++) started[t] = pthread_create(&th[t], NULL, parse_thread, &sh[t]) == 0;
            parse_thread(&sh[0]);
            for (t = 1; t < nt; t++) { if (started[t]) pthread_join(th[t], NULL); else parse_thread(&sh[t]); }
        }
        for (r = 0; r < nrun; r++) {
            gop_run *q = &run[r];
            unsigned k;
            int type;
            if (q->dead || step >= q->n) continue;
            k = s->gfirst[q->g] + step;
            type = s->type[k];
            /* reference frames, codec/coder.c:580-628 */
            if (type == FA_I_FRAME) {
                fa_image_free(q->past); fa_image_free(q->future); fa_image_free(q->reconst);
                q->past = q->future = q->reconst = NULL;
            } else if (type == FA_P_FRAME) {
                fa_image_free(q->past); q->past = q->reconst; q->reconst = NULL;
                fa_image_free(q->future); q->future = NULL;
            } else if (q->last_was_future) {
                fa_image_free(q->future); q->future = q->reconst; q->reconst = NULL;
            } else if (s->wi.B_as_past_ref) {
                fa_image_free(q->past); q->past = q->reconst; q->reconst = NULL;
            } else {
                fa_image_free(q->reconst); q->reconst = NULL;
            }
            q->last_was_future = s->isfut[k];
            ims[nb] = pre[r].im; pre[r].im = NULL;             /* parsed side by side above */
            if (!ims[nb]) { snprintf(s->gerr[q->g], 160, "%s", pre[r].err); s->gfail[q->g] = 1; q->dead = 1; continue; }
            memset(&jobs[nb], 0, sizeof jobs[nb]);
            jobs[nb].image = ims[nb];
            jobs[nb].cp = s->cp;
            jobs[nb].cp.lc_min_level = q->carry;
            jobs[nb].share_key = q->g + 1;         /* a GOP stays on one device share whatever its index in this step's batch */
            if (nb == 0 && step == 0 && q->g == 0 && s->probe_wfa && q->carry == s->cp.lc_min_level && type == FA_I_FRAME) {
                /* the probe was this very search: its result stands in for the job (entry 0, not sent to the core) */
                jobs[nb].frame_type = type;
                jobs[nb].wfa = s->probe_wfa; s->probe_wfa = NULL;
                memcpy(jobs[nb].stats, s->probe_stats, sizeof jobs[nb].stats);
                jobs[nb].lc_min_level_out = s->probe_out;
                jobs[nb].status = 1;
                predone = 1;
                who[nb++] = r;
                continue;
            }
            jobs[nb].frame_type = type; jobs[nb].past = q->past; jobs[nb].future = q->future;
            jobs[nb].wfa = fa_wfa_alloc(s->cp.limit_states);
            if (!jobs[nb].wfa) {
                /* give back what this step has prepared so far (the `out' path only knows run[]) */
                unsigned j;
                fa_set_error("Out of memory!");
                fa_image_free(ims[nb]);
                for (j = 0; j < nb; j++) { fa_wfa_free(jobs[j].wfa); fa_image_free(ims[j]); }
                goto out;
            }
            if (!fa_load_basis(s->op->basis_name, jobs[nb].wfa) || jobs[nb].wfa->states >= s->cp.limit_states) {
                if (jobs[nb].wfa->states >= s->cp.limit_states) fa_set_error("Maximum number of states reached!");
                snprintf(s->gerr[q->g], 160, "%s", fiasco_get_error_message());
                s->gfail[q->g] = 1; q->dead = 1;
                fa_wfa_free(jobs[nb].wfa); fa_image_free(ims[nb]);
                continue;
            }
            if (s->color) {                /* every entry "not written by this frame" */
                memset(jobs[nb].wfa->y_column, YCOL_UNSET, (size_t) jobs[nb].wfa->cap * 2);
                jobs[nb].ycol_carry = 1;
            }
            who[nb++] = r;
        }
        if (!nb) continue;
        t_prep += seq_now() - t0; t0 = seq_now();
        if (nb > predone) (void) fa_core_encode_frames(nb - predone, jobs + predone);
        t_core += seq_now() - t0; t0 = seq_now();
        for (b = 0; b < nb; b++) {
            gop_run *q = &run[who[b]];
            const unsigned k = s->gfirst[q->g] + step;
            need[b] = 0;
            if (!jobs[b].status) {
                snprintf(s->gerr[q->g], 160, "%s", jobs[b].errmsg);
                s->gfail[q->g] = 1; q->dead = 1;
                fa_wfa_free(jobs[b].wfa);
            } else {
                s->wfa[k] = jobs[b].wfa;
                s->wfa[k]->frame_type = s->type[k];
                memcpy(s->stats[k], jobs[b].stats, sizeof jobs[b].stats);
                q->carry = jobs[b].lc_min_level_out;
                need[b] = video && step + 1 < q->n;       /* reference for the frames to come (:647-651) */
            }
            fa_image_free(ims[b]); ims[b] = NULL;
        }
        /* the references of the frames to come (codec/coder.c:647-651): decode_image + restore_mc, one batch for
         * the core -- the device, which keeps the planes for the next search (fa_image.dev) */
        {
            unsigned any = 0;
            for (b = 0; b < nb; b++) {
                gop_run *q = &run[who[b]];
                const unsigned k = s->gfirst[q->g] + step;
                memset(&djobs[b], 0, sizeof djobs[b]);
                djobs[b].skip = !need[b]; djobs[b].keep_dev = 1;
                djobs[b].share_key = q->g + 1;     /* ... and its reference frames are decoded there */
                if (!need[b]) continue;
                any = 1;
                djobs[b].wfa = s->wfa[k]; djobs[b].width = s->wi.width; djobs[b].height = s->wi.height;
                djobs[b].color = s->color; djobs[b].frame_type = s->type[k];
                djobs[b].past = q->past; djobs[b].future = q->future; djobs[b].p_max_level = s->wi.p_max_level;
            }
            if (any) (void) fa_core_decode_frames(nb, djobs);
            for (b = 0; b < nb; b++) {
                gop_run *q = &run[who[b]];
                if (!need[b]) continue;
                q->reconst = djobs[b].out;
                if (q->reconst && fa_knob("FIASCO_AMD_SEQ_RECONST_LOG")) {
                    /* tests: one line per reconstructed frame (GOP, step, FNV-1a of its planes) -- the device decoder
                     * against the host restatement on P and B frames */
                    FILE *lf = fopen(fa_knob("FIASCO_AMD_SEQ_RECONST_LOG"), "a");
                    if (lf) {
                        unsigned long long h = 1469598103934665603ull;
                        int band;
                        for (band = 0; band < (s->color ? 3 : 1); band++) {
                            const unsigned char *p8 = (const unsigned char *) q->reconst->pixels[band];
                            size_t i, n8 = (size_t) q->reconst->width * q->reconst->height * 2;
                            for (i = 0; i < n8; i++) { h ^= p8[i]; h *= 1099511628211ull; }
                        }
                        fprintf(lf, "%u %u %d %016llx\n", q->g, step, (int) s->type[s->gfirst[q->g] + step], h);
                        fclose(lf);
                    }
                }
                if (!q->reconst) {
                    snprintf(s->gerr[q->g], 160, "%s", djobs[b].errmsg[0] ? djobs[b].errmsg : "decoder failed");
                    s->gfail[q->g] = 1; q->dead = 1;
                }
            }
        }
        t_dec += seq_now() - t0;
    }
    if (fa_knob("FIASCO_AMD_SEQ_TIMING"))
        fprintf(stderr, "fa_seq_search: %u GOPs, %u steps: prepare %.2f s, core %.2f s, decode %.2f s\n",
                nrun, maxlen, t_prep, t_core, t_dec);
    for (r = 0; r < nrun; r++) {
        s->carry_out[run[r].g] = run[r].carry;
        s->gdone[run[r].g] = 1;
    }
    rc = 1;
out:
    for (r = 0; run && r < nrun; r++) { fa_image_free(run[r].reconst); fa_image_free(run[r].past); fa_image_free(run[r].future); }
    free(run); free(jobs); free(ims); free(who); free(need); free(djobs);
    for (r = 0; pre && r < nrun; r++) fa_image_free(pre[r].im);
    free(pre);
    return rc;
}

/* The level to speculate for the GOPs behind the first one: what the very first frame (always an
 * I frame) leaves -- the ratchet usually settles there.  One extra frame through the core
 * instead of a second sweep over all GOPs; gray streams never change the level. */
int fa_seq_probe(fa_seq *s, unsigned *level)
{
    fa_job job;
    fa_image *im;
    int ok;
    *level = s->cp.lc_min_level;
    if (!s->color || s->ngop < 2) return 1;
    im = fa_image_from_pnm(s->bufs[s->order[0]], s->lens[s->order[0]], s->names ? s->names[s->order[0]] : "<memory>");
    if (!im) return 0;
    memset(&job, 0, sizeof job);
    job.image = im; job.cp = s->cp; job.frame_type = FA_I_FRAME;
    job.wfa = fa_wfa_alloc(s->cp.limit_states);
    ok = job.wfa && fa_load_basis(s->op->basis_name, job.wfa) && job.wfa->states < s->cp.limit_states;
    if (ok) {                                  /* exactly the job the sweep makes of this frame (fa_seq_search) */
        memset(job.wfa->y_column, YCOL_UNSET, (size_t) job.wfa->cap * 2);
        job.ycol_carry = 1;
        ok = fa_core_encode_frames(1, &job) == 1;
    }
    fa_wfa_free(s->probe_wfa); s->probe_wfa = NULL;
    if (ok) {
        *level = job.lc_min_level_out;
        s->probe_wfa = job.wfa; job.wfa = NULL;
        memcpy(s->probe_stats, job.stats, sizeof job.stats);
        s->probe_out = job.lc_min_level_out;
    }
    fa_wfa_free(job.wfa); fa_image_free(im);
    return 1;                                  /* a failing first frame shows up in the sweep */
}

/* y_column of coded frame k as the core left it: YCOL_UNSET where the frame wrote nothing */
const uint8_t *fa_seq_ycol_raw(const fa_seq *s, unsigned k)
{
    return s->wfa[k] ? s->wfa[k]->y_column : NULL;
}

/* resolve raw flags of a frame against what the frames before it left: chain[] is updated in place */
void fa_seq_ycol_resolve(uint8_t *chain, const uint8_t *raw, unsigned n)
{
    unsigned i;
    for (i = 0; i < n; i++) if (raw[i] != YCOL_UNSET) chain[i] = raw[i];
}

/* stream of coded frame k (header included if it is frame number 0); ycol: the resolved flags of
 * this frame (gray: NULL) */
int fa_seq_write(fa_seq *s, unsigned k, const uint8_t *ycol, fa_bitw *out)
{
    fa_wfa *w = s->wfa[k];
    if (!w) { fa_set_error("frame %d was not searched by this process", (int) k); return 0; }
    if (s->color && ycol) memcpy(w->y_column, ycol, (size_t) w->cap * 2);
    return fa_write_frame(w, &s->wi, s->type[k], s->order[k], s->op->prediction, s->op->normal_domains,
                          s->op->delta_domains, out);
}

/* the streams of the frames `first, first + stride, ...', each into a writer of its own */
#define WR_THREADS 16
typedef struct wr_share {
    fa_seq *s;
    fa_bitw *fb;
    uint8_t *ok;
    char (*err)[160];
    unsigned first, stride;
} wr_share;
static void *wr_frames_thread(void *arg)
{
    wr_share *w = (wr_share *) arg;
    fa_seq *s = w->s;
    unsigned k;
    for (k = w->first; k < s->ncoded; k += w->stride) {
        fa_bw_init(&w->fb[k]);
        w->ok[k] = (uint8_t) (fa_seq_write(s, k, NULL, &w->fb[k]) != 0);
        if (!w->ok[k]) snprintf(w->err[k], 160, "%s", fiasco_get_error_message());
    }
    return NULL;
}

/* Everything in one process: search with speculation until every GOP started from what its
 * predecessor left, then the streams in coding order.  Used by fiasco_coder(). */
int fa_seq_encode_all(fa_seq *s, fa_bitw *out, void (*report)(const fa_wfa *, const fa_stats *, const fa_info *))
{
    unsigned *carry = (unsigned *) calloc(s->ngop, sizeof *carry);
    uint8_t *todo = (uint8_t *) calloc(s->ngop, 1), *chain = NULL;
    unsigned g, k, first_invalid = 0;
    int rc = 0, pass;
    if (!carry || !todo) { fa_set_error("Out of memory!"); goto out; }
    {
        unsigned guess;
        double tp = seq_now();
        if (!fa_seq_probe(s, &guess)) goto out;
        if (fa_knob("FIASCO_AMD_SEQ_TIMING")) fprintf(stderr, "fa_seq_encode_all: probe %.2f s (level %u -> %u)\n", seq_now() - tp, s->cp.lc_min_level, guess);
        for (g = 0; g < s->ngop; g++) { carry[g] = g ? guess : s->cp.lc_min_level; todo[g] = 1; }
    }
    for (pass = 0; ; pass++) {
        unsigned t = s->cp.lc_min_level;
        if (!fa_seq_search(s, carry, todo)) goto out;
        /* verify the chain (gray streams never change the level: one pass) */
        for (g = 0; g < s->ngop; g++) {
            if (s->carry_used[g] != t) break;
            if (s->gfail[g]) { fa_set_error("%s", s->gerr[g]); goto out; }
            t = s->carry_out[g];
        }
        first_invalid = g;
        if (g == s->ngop) break;
        /* search the rest again, all of it starting from the value that is now known for the
         * first of them (the level only goes up and usually stays) */
        for (g = 0; g < s->ngop; g++) { todo[g] = g >= first_invalid; if (todo[g]) carry[g] = t; }
        if (pass > 64) { fa_set_error("internal error: minimum level chain does not settle"); goto out; }
    }
    if (s->color) {
        chain = (uint8_t *) calloc(s->cap2, 1);       /* calloc'ed in the reference (codec/wfalib.c:107) */
        if (!chain) { fa_set_error("Out of memory!"); goto out; }
    }
    double tw = seq_now();
    {
        /* The streams of the frames side by side (the writer is a pure function of a finished automaton and the
         * flags resolved above it), joined in coding order.  A frame ends byte-aligned -- the arithmetic coder of its
         * last section is flushed (lib/arith.c:86-115) --, and then what a fresh writer produced for the next frame
         * IS what the shared one would have appended.  Should a frame end inside a byte (no edges at all), the rest
         * is written the sequential way. */
        wr_share sh[WR_THREADS];
        pthread_t th[WR_THREADS];
        int started[WR_THREADS] = { 0 };
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        unsigned nt = ncpu > WR_THREADS ? WR_THREADS : ncpu < 1 ? 1 : (unsigned) ncpu, t;
        fa_bitw *fb = (fa_bitw *) calloc(s->ncoded ? s->ncoded : 1, sizeof *fb);
        uint8_t *okv = (uint8_t *) calloc(s->ncoded ? s->ncoded : 1, 1);
        char (*errv)[160] = (char (*)[160]) calloc(s->ncoded ? s->ncoded : 1, 160);
        int failed = 0;
        if (!fb || !okv || !errv) { free(fb); free(okv); free(errv); fa_set_error("Out of memory!"); goto out; }
        for (k = 0; k < s->ncoded; k++) {
            if (!s->wfa[k]) { free(fb); free(okv); free(errv); fa_set_error("frame %d was not searched by this process", (int) k); goto out; }
            if (s->color) {                     /* the flags of frame k: resolved in coding order, kept in the automaton */
                fa_seq_ycol_resolve(chain, fa_seq_ycol_raw(s, k), s->cap2);
                memcpy(s->wfa[k]->y_column, chain, (size_t) s->wfa[k]->cap * 2);
            }
            if (report) report(s->wfa[k], s->stats[k], &s->wi);
        }
        if (nt > s->ncoded) nt = s->ncoded ? s->ncoded : 1;
        for (t = 0; t < nt; t++) { sh[t].s = s; sh[t].fb = fb; sh[t].ok = okv; sh[t].err = errv; sh[t].first = t; sh[t].stride = nt; }
        for (t = 1; t < nt; t++) started[t] = pthread_create(&th[t], NULL, wr_frames_thread, &sh[t]) == 0;
        wr_frames_thread(&sh[0]);
        for (t = 1; t < nt; t++) { if (started[t]) pthread_join(th[t], NULL); else wr_frames_thread(&sh[t]); }
        for (k = 0; k < s->ncoded && !failed; k++) {
            const int aligned = out->bitpos == 0 || out->nbits == 0;
            if (!okv[k]) { fa_set_error("%s", errv[k]); failed = 1; break; }
            if (aligned) fa_bw_append(out, &fb[k]);
            else if (!fa_seq_write(s, k, NULL, out)) failed = 1;          /* (the flags are in the automaton already) */
        }
        for (k = 0; k < s->ncoded; k++) fa_bw_free(&fb[k]);
        free(fb); free(okv); free(errv);
        if (failed) goto out;
    }
    if (fa_knob("FIASCO_AMD_SEQ_TIMING")) fprintf(stderr, "fa_seq_encode_all: writer %.2f s\n", seq_now() - tw);
    rc = 1;
out:
    free(carry); free(todo); free(chain);
    return rc;
}

/* ---------------------------------------------------------------- C ABI for multi-rank callers */

struct fiasco_amd_seq { fa_seq *s; fiasco_c_options_t *defaults; };

fiasco_amd_seq_t *fiasco_amd_seq_open(unsigned n, const unsigned char *const *pnm, const size_t *pnm_len,
                                      float quality, const fiasco_c_options_t *options, unsigned rank, unsigned world)
{
    fiasco_amd_seq_t *q;
    const fa_options *op;
    if (quality <= 0) { fa_set_error("Compression quality has to be positive."); return NULL; }
    if (world == 0 || rank >= world) { fa_set_error("rank %d of %d?", (int) rank, (int) world); return NULL; }
    q = (fiasco_amd_seq_t *) calloc(1, sizeof *q);
    if (!q) { fa_set_error("Out of memory!"); return NULL; }
    if (options) op = fa_cast_options(options);
    else { q->defaults = fiasco_c_options_new(); op = q->defaults ? fa_cast_options(q->defaults) : NULL; }
    if (op) q->s = fa_seq_open(op, quality, n, pnm, pnm_len, NULL, rank, world);
    if (!q->s) { if (q->defaults) fiasco_c_options_delete(q->defaults); free(q); return NULL; }
    return q;
}

void fiasco_amd_seq_free(fiasco_amd_seq_t *q)
{
    if (!q) return;
    fa_seq_free(q->s);
    if (q->defaults) fiasco_c_options_delete(q->defaults);
    free(q);
}

int fiasco_amd_seq_probe(fiasco_amd_seq_t *q, unsigned *level) { return fa_seq_probe(q->s, level); }
unsigned fiasco_amd_seq_gops(const fiasco_amd_seq_t *q) { return q->s->ngop; }
unsigned fiasco_amd_seq_frames(const fiasco_amd_seq_t *q) { return q->s->ncoded; }
unsigned fiasco_amd_seq_ycol_size(const fiasco_amd_seq_t *q) { return q->s->color ? q->s->cap2 : 0; }
unsigned fiasco_amd_seq_initial_level(const fiasco_amd_seq_t *q) { return q->s->cp.lc_min_level; }
unsigned fiasco_amd_seq_gop_of(const fiasco_amd_seq_t *q, unsigned frame) { return fa_seq_gop_of(q->s, frame); }
int fiasco_amd_seq_search(fiasco_amd_seq_t *q, const unsigned *carry_in, const unsigned char *todo)
{
    return fa_seq_search(q->s, carry_in, todo);
}
int fiasco_amd_seq_gop_result(const fiasco_amd_seq_t *q, unsigned gop, unsigned *carry_out, int *failed)
{
    if (gop >= q->s->ngop || !q->s->gdone[gop]) return 0;
    {
        int f = 0;
        fa_seq_gop_result(q->s, gop, carry_out, &f);
        if (failed) *failed = f;
        if (f) fa_set_error("%s", q->s->gerr[gop]);
    }
    return 1;
}
const unsigned char *fiasco_amd_seq_ycol(const fiasco_amd_seq_t *q, unsigned frame)
{
    return frame < q->s->ncoded && q->s->color ? fa_seq_ycol_raw(q->s, frame) : NULL;
}
int fiasco_amd_seq_write(fiasco_amd_seq_t *q, unsigned frame, const unsigned char *ycol,
                         unsigned char **out, size_t *out_len)
{
    fa_bitw bw;
    int ok;
    *out = NULL; *out_len = 0;
    if (frame >= q->s->ncoded) { fa_set_error("no such frame"); return 0; }
    fa_bw_init(&bw);
    ok = fa_seq_write(q->s, frame, ycol, &bw);
    if (ok && bw.nbits % 8) {
        /* byte strings of different processes are put together: a frame has to end on a byte
         * boundary (it does unless it has no edge at all, output/matrices.c:177-254) */
        fa_set_error("frame %d does not end on a byte boundary", (int) frame);
        ok = 0;
    }
    if (ok) {
        *out_len = fa_bw_finish(&bw);
        *out = (unsigned char *) malloc(*out_len ? *out_len : 1);
        if (*out) memcpy(*out, bw.buf, *out_len); else { fa_set_error("Out of memory!"); ok = 0; }
    }
    fa_bw_free(&bw);
    return ok;
}
