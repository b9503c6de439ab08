/*
 *  fa_coder.c -- fiasco_coder() and the batch entry point: the host-side frame driver.
 *
 *  Mirrors the control flow of reference codec/coder.c:
 *    fiasco_coder   :85-182   parameter checks, output stream, basis, price
 *    alloc_coder    :190-366  level / limit derivation (fa_setup_params below)
 *    get_input_image_name :390-488  "prefix[start-end{+,-}step]suffix" templates
 *    video_coder    :490-668  frame loop (I frames only in this library build)
 *    frame_coder    :692-892  per-frame models + partition search + write_next_wfa
 *  Everything from subdivide() downwards runs behind fa_core_encode_frames().
 */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>
#include <errno.h>
#include <pthread.h>
#include <unistd.h>
#include "fa_host.h"

static unsigned g_limit_states = FA_STOCK_STATES;
static unsigned g_limit_level  = FA_STOCK_LEVEL;

int fiasco_amd_set_limits(unsigned max_states, unsigned max_level)
{
    if (max_states < 16 || max_states > FA_CAP_STATES
        || max_level < FA_STOCK_LEVEL || max_level > FA_CAP_LEVEL) {
        fa_set_error("Limits out of range (states 16..%d, level %d..%d).",
                     FA_CAP_STATES, FA_STOCK_LEVEL, FA_CAP_LEVEL);
        return 0;
    }
    g_limit_states = max_states;
    g_limit_level  = max_level;
    return 1;
}

void fiasco_amd_get_limits(unsigned *max_states, unsigned *max_level)
{
    if (max_states) *max_states = g_limit_states;
    if (max_level)  *max_level  = g_limit_level;
}

void fa_limits(unsigned *max_states, unsigned *max_level)
{
    fiasco_amd_get_limits(max_states, max_level);
}

unsigned fa_image_level(unsigned width, unsigned height)
{
    unsigned lx = (unsigned) (log2((double) (width - 1)) + 1);
    unsigned ly = (unsigned) (log2((double) (height - 1)) + 1);
    unsigned m = lx > ly ? lx : ly;
    return m * 2 - ((ly == lx + 1) ? 1 : 0);
}

static unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
static unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }

void fa_info_free(fa_info *wi)
{
    free(wi->basis_name); free(wi->title); free(wi->comment);
    wi->basis_name = wi->title = wi->comment = NULL;
}

int fa_setup_params(const fa_options *op, float quality, unsigned width, unsigned height,
                    int color, unsigned frames, fa_info *wi, fa_cparams *cp)
{
    memset(wi, 0, sizeof *wi);
    memset(cp, 0, sizeof *cp);
    wi->frames = frames; wi->width = width; wi->height = height; wi->color = color;
    wi->level = fa_image_level(width, height);
    if (wi->level > g_limit_level) {
        /* the stock reference overruns tree_t.total[MAXLEVEL] here and crashes
         * (SURVEY finding 2); this library reports the condition instead */
        fa_set_error("Image level %d exceeds MAXLEVEL %d (use fiasco_amd_set_limits).",
                     (int) wi->level, (int) g_limit_level);
        return 0;
    }
    cp->level        = wi->level;
    cp->lc_min_level = umax(op->lc_min_level, 3);
    cp->lc_max_level = umin(op->lc_max_level, wi->level - 1);
    if ((int) wi->level - (int) op->tiling_exponent < 6)
        fa_warning("Image tiles must be at least 8x8 pixels large.\n"
                   "Setting tiling size to 8x8 pixels.");
    if (cp->lc_min_level > cp->lc_max_level) cp->lc_min_level = cp->lc_max_level;
    wi->p_min_level = umax(op->p_min_level, cp->lc_min_level);
    wi->p_max_level = umin(op->p_max_level, cp->lc_max_level);
    if (wi->p_min_level > wi->p_max_level) wi->p_min_level = wi->p_max_level;
    cp->images_level   = umin(op->images_level, cp->lc_max_level - 1);
    cp->products_level = (unsigned) ((int) cp->lc_max_level - (int) cp->images_level - 1 > 0
                                     ? cp->lc_max_level - cp->images_level - 1 : 0);
    wi->max_states   = umax(umin(op->max_states, g_limit_states), 1);
    cp->max_elements = umax(umin(op->max_elements, FA_MAXEDGES), 1);
    cp->pool_max_states = wi->max_states;
    wi->title   = strdup(op->title);
    wi->comment = strdup(op->comment);
    wi->basis_name = strdup(op->basis_name);
    fa_rpf_init(&wi->rpf,      op->rpf_mantissa,      op->rpf_range);
    fa_rpf_init(&wi->dc_rpf,   op->dc_rpf_mantissa,   op->dc_rpf_range);
    fa_rpf_init(&wi->d_rpf,    op->d_rpf_mantissa,    op->d_rpf_range);
    fa_rpf_init(&wi->d_dc_rpf, op->d_dc_rpf_mantissa, op->d_dc_rpf_range);
    cp->rpf = wi->rpf; cp->dc_rpf = wi->dc_rpf; cp->d_rpf = wi->d_rpf; cp->d_dc_rpf = wi->d_dc_rpf;
    wi->chroma_max_states = umax(1, op->chroma_max_states);
    cp->chroma_max_states = wi->chroma_max_states;
    cp->chroma_decrease   = op->chroma_decrease;
    wi->search_range   = op->search_range;
    wi->fps            = op->fps;
    wi->half_pixel     = op->half_pixel_prediction;
    wi->cross_B_search = op->half_pixel_prediction;   /* sic, codec/coder.c:359 */
    wi->B_as_past_ref  = op->B_as_past_ref;
    wi->smoothing      = op->smoothing;
    cp->second_domain_block = op->second_domain_block;
    cp->check_for_underflow = op->check_for_underflow;
    cp->check_for_overflow  = op->check_for_overflow;
    cp->full_search         = op->full_search;
    cp->prediction     = op->prediction;
    cp->p_min_level    = wi->p_min_level;
    cp->p_max_level    = wi->p_max_level;
    cp->delta_domains  = op->delta_domains;
    cp->normal_domains = op->normal_domains;
    cp->search_range   = wi->search_range;
    cp->half_pixel     = wi->half_pixel;
    cp->cross_B_search = wi->cross_B_search;
    cp->price        = 128 * 64 / quality;
    cp->limit_states = g_limit_states;
    cp->limit_level  = g_limit_level;
    if (frames > 1 && op->half_pixel_prediction) {
        /* codec/motion.c:271 divides the vector after its conversion to unsigned: every negative
         * half-pixel component addresses memory far outside the frame.  Nothing to be bit
         * compatible with; cfiasco cannot switch it on (bin/cwfa.c never sets video parameters) */
        fa_set_error("Half-pixel motion vectors are not supported (the reference coder's half-pixel "
                     "path reads outside the reference frame).");
        return 0;
    }
    {   /* alloc_domain_pool / alloc_coeff_model (codec/domain-pool.c:203-236, codec/coeff.c:107-131): an
         * unknown name is a warning and the first entry of the table */
        int known;
        cp->pool_kind = fa_pool_kind_of(op->id_domain_pool, &known);
        if (!known) fa_warning("Can't initialize domain pool '%s'. Using default value '%s'.", op->id_domain_pool, fa_pool_name(cp->pool_kind));
        cp->d_pool_kind = fa_pool_kind_of(op->id_d_domain_pool, &known);
        if (!known) fa_warning("Can't initialize domain pool '%s'. Using default value '%s'.", op->id_d_domain_pool, fa_pool_name(cp->d_pool_kind));
        cp->coeff_kind = fa_coeff_kind_of(op->id_rpf_model, &known);
        if (!known) fa_warning("Can't initialize coefficients model '%s'. Using default value '%s'.", op->id_rpf_model, fa_coeff_name(cp->coeff_kind));
        cp->d_coeff_kind = fa_coeff_kind_of(op->id_d_rpf_model, &known);
        if (!known) fa_warning("Can't initialize coefficients model '%s'. Using default value '%s'.", op->id_d_rpf_model, fa_coeff_name(cp->d_coeff_kind));
    }
    return 1;
}

/* ---------------------------------------------------------------- input names */

/* i-th frame name of a NULL-terminated template list; NULL past the end; *err set on a
 * malformed template (codec/coder.c:390-488) */
static char *input_name(char const *const *templ, unsigned ith, int *err)
{
    *err = 0;
    for (; *templ; templ++) {
        const char *t = *templ, *open = strchr(t, '[');
        if (!open) {
            if (ith == 0) return strdup(t);
            ith--;
            continue;
        }
        {
            const char *s = open + 1, *s2;
            unsigned ndig = 0;
            int first, last, inc = 1, num;
            for (s2 = s; isdigit((unsigned char) *s2); s2++) ndig++;
            if (sscanf(s, "%d", &first) != 1 || first < 0 || *s2++ != '-') goto bad;
            s = s2;
            while (isdigit((unsigned char) *s2)) s2++;
            if (sscanf(s, "%d", &last) != 1 || last < 0) goto bad;
            if (*s2 == '+' || *s2 == '-') {
                s = s2++;
                while (isdigit((unsigned char) *s2)) s2++;
                if (sscanf(s, "%d", &inc) != 1) goto bad;
            }
            if (*s2 != ']') goto bad;
            num = first + inc * (int) ith;
            if (num < 0) goto bad;
            if ((inc > 0 && (unsigned) num > (unsigned) last)
                || (inc <= 0 && (unsigned) num < (unsigned) last)) {
                if (inc == 0) goto bad;
                ith -= (unsigned) ((last - first) / inc + 1);
            } else {
                size_t plen = (size_t) (open - t);
                char *name = (char *) malloc(plen + 32 + strlen(s2 + 1));
                if (!name) { *err = 1; return NULL; }
                memcpy(name, t, plen);
                sprintf(name + plen, "%0*d%s", (int) ndig, num, s2 + 1);
                return name;
            }
        }
    }
    return NULL;
bad:
    fa_set_error("Input name template conversion failure.\nCheck spelling of template.");
    *err = 1;
    return NULL;
}

/* ---------------------------------------------------------------- one still -> bytes */

static int prepare_job(fa_job *job, const fa_image *im, const fa_cparams *cp, const char *basis)
{
    memset(job, 0, sizeof *job);
    job->image = im;
    job->cp    = *cp;
    job->wfa   = fa_wfa_alloc(cp->limit_states);
    if (!job->wfa) { fa_set_error("Out of memory!"); return 0; }
    if (!fa_load_basis(basis, job->wfa)) return 0;
    if (job->wfa->states >= cp->limit_states) {
        fa_set_error("Maximum number of states reached!");
        return 0;
    }
    return 1;
}

static void report(const fa_wfa *wfa, const fa_stats *stats, const fa_info *wi)
{
    int b, nb = wi->color ? 3 : 1;
    for (b = 0; b < nb; b++) {
        const fa_stats *s = &stats[b];
        double mse = s->err / wi->width / wi->height;
        fa_debug("WFA contains %d states (%d basis states).", (int) wfa->states,
                 (int) wfa->basis_states);
        fa_debug("Estimated error: %.2f (RMSE: %.2f, PSNR: %.2f dB).", (double) s->err,
                 sqrt(mse), 10 * log(255.0 * 255.0 / mse) / log(10.0));
        fa_debug("(T: %.0f, M: %.0f, W: %.0f)", (double) s->tree_bits, (double) s->matrix_bits,
                 (double) s->weights_bits);
        fa_debug("Total costs : %.2f", (double) s->costs);
    }
}

/* ---------------------------------------------------------------- public: fiasco_coder */

int fiasco_coder(char const *const *inputname, const char *outputname, float quality,
                 const fiasco_c_options_t *options)
{
    static char const *const default_input[] = { "-", NULL };
    char const *const *templ;
    fiasco_c_options_t *defaults = NULL;
    const fa_options *op;
    fa_bitw out;
    FILE *fout = NULL;
    unsigned nframes = 0, i;
    int rc = 0, have_out = 0, err;
    char **names = NULL;
    unsigned char **bufs = NULL;
    size_t *lens = NULL;
    fa_seq *seq = NULL;

    templ = (!inputname || !inputname[0] || strcmp(inputname[0], "-") == 0) ? default_input : inputname;
    if (quality <= 0) { fa_set_error("Compression quality has to be positive."); return 0; }
    if (quality >= 100)
        fa_warning("Quality typically is 1 (worst) to 100 (best).\nBe prepared for a long running time.");
    if (options) {
        op = fa_cast_options(options);
        if (!op) return 0;
    } else {
        defaults = fiasco_c_options_new();
        if (!defaults) return 0;
        op = fa_cast_options(defaults);
    }

    /* the frame names (no file is touched yet) */
    for (;; nframes++) {
        char *nm = input_name(templ, nframes, &err);
        if (!nm) { if (err) goto done; break; }
        {
            char **n2 = (char **) realloc(names, (nframes + 1) * sizeof *names);
            unsigned char **b2 = (unsigned char **) realloc(bufs, (nframes + 1) * sizeof *bufs);
            size_t *l2 = (size_t *) realloc(lens, (nframes + 1) * sizeof *lens);
            if (n2) names = n2;
            if (b2) bufs = b2;
            if (l2) lens = l2;
            if (!n2 || !b2 || !l2) { free(nm); fa_set_error("Out of memory!"); goto done; }
        }
        names[nframes] = nm; bufs[nframes] = NULL; lens[nframes] = 0;
    }
    /* what this library refuses is refused before the output file is truncated */
    if (nframes > 1 && op->half_pixel_prediction) {
        fa_set_error("Half-pixel motion vectors are not supported (the reference coder's half-pixel "
                     "path reads outside the reference frame).");
        goto done;
    }

    /* the reference opens (and truncates) the output stream before anything else can fail */
    fout = open_file(outputname, "FIASCO_DATA", WRITE_ACCESS);
    if (!fout) {
        fa_set_error("Can't write outputfile `%s'.\n%s", outputname ? outputname : "<stdout>",
                     strerror(errno));
        goto done;
    }
    for (i = 0; i < nframes; i++) {
        const char *nm = strcmp(names[i], "-") == 0 ? NULL : names[i];
        bufs[i] = fa_read_whole_file(nm, "FIASCO_IMAGES", &lens[i]);
        if (!bufs[i]) {
            fa_set_error("Can't open frame `%s'.\n%s", names[i], strerror(errno));
            goto done;
        }
    }
    if (nframes == 0) { fa_set_error("Can't open frame `%s'.", "<none>"); goto done; }

    /* video_coder (codec/coder.c:490-668): the groups of pictures side by side, fa_sequence.c */
    seq = fa_seq_open(op, quality, nframes, (const unsigned char *const *) bufs, lens,
                      (char const *const *) names, 0, 1);
    if (!seq) goto done;
    fa_bw_init(&out); have_out = 1;
    if (!fa_seq_encode_all(seq, &out, report)) goto done;
    {
        size_t nbytes = fa_bw_finish(&out);
        if (fwrite(out.buf, 1, nbytes, fout) != nbytes) {
            fa_set_error("Can't write remaining %d bytes of bitfile!", (int) nbytes);
            goto done;
        }
    }
    rc = 1;
done:
    fa_seq_free(seq);
    if (have_out) fa_bw_free(&out);
    if (fout && fout != stdout) fclose(fout);
    else if (fout) fflush(fout);
    for (i = 0; i < nframes; i++) { free(names[i]); free(bufs ? bufs[i] : NULL); }
    free(names); free(bufs); free(lens);
    if (defaults) fiasco_c_options_delete(defaults);
    return rc;
}

/* ---------------------------------------------------------------- public: batch */

struct fiasco_amd_batch {
    unsigned   n;
    fa_job    *jobs;
    fa_image **ims;
    fa_image **prev_ims;      /* frames of the pass before the last upload: a pass that was
                               * submitted with them may still be in flight */
    fa_info   *infos;
    int        normal_domains, delta_domains, prediction;
    void      *staged;        /* core handle: inputs resident where the core computes */
};

int fiasco_amd_batch_stats(const fiasco_amd_batch_t *b, unsigned i, unsigned band,
                           float *costs, float *err, unsigned *width, unsigned *height)
{
    if (!b || i >= b->n || band > 2 || !b->jobs[i].status) return 0;
    if (band && !b->jobs[i].image->color) return 0;
    if (costs)  *costs  = b->jobs[i].stats[band].costs;
    if (err)    *err    = b->jobs[i].stats[band].err;
    if (width)  *width  = b->jobs[i].image->width;
    if (height) *height = b->jobs[i].image->height;
    return 1;
}

/* Decoded PSNR of frame i of a batch whose last pass succeeded: the finished automaton
 * is decoded like `dfiasco -s 0` does (decode_image, codec/decoder.c:411-536, no smoothing -- the frame the
 * coder itself would use as a reference) and compared with the input the way bin/pnmpsnr.c:36-163 compares
 * two PNM files: both sides as bytes, clip((pixel >> 4) + 128) (lib/image.c gray_write), squared differences
 * summed sequentially in float, 10 log10(255^2 / mean).  psnr_db[band] = +inf when the planes do not
 * differ (pnmpsnr: "don't differ" below 1e-4).  For square gray frames that is pnmpsnr's figure to the
 * last digit (tests/golden/MANIFEST.json "decoded_psnr").  For w x h frames the reference tool divides by
 * w x w -- fiasco_image_get_height() returns the width, lib/image.c:134 -- so its mean is ours x h / w for
 * h <= w (the tests convert); here the mean is over the pixels of the image.  For colour frames the three
 * planes Y, Cb, Cr are compared as they are, without pnmpsnr's detour through RGB. */
/* the frame of a finished intra job through the core's decoder (the device; the host decoder in the test oracle) */
static fa_image *decode_job(const fa_job *job)
{
    fa_dec_job d;
    memset(&d, 0, sizeof d);
    d.wfa = job->wfa; d.width = job->image->width; d.height = job->image->height; d.color = job->image->color;
    d.frame_type = FA_I_FRAME;
    if (fa_core_decode_frames(1, &d) != 1 || !d.out) {
        fa_set_error("%s", d.errmsg[0] ? d.errmsg : "decoder failed");
        return NULL;
    }
    return d.out;
}

/* mean squared error and PSNR of a decoded frame against its original, as bin/pnmpsnr.c:92-101 sums them */
static void psnr_of(const fa_image *orig, const fa_image *dec, double psnr_db[3], double mse[3])
{
    const unsigned nb = orig->color ? 3 : 1;
    unsigned band;
    for (band = 0; band < 3; band++) { if (psnr_db) psnr_db[band] = 0; if (mse) mse[band] = 0; }
    for (band = 0; band < nb; band++) {
        const int16_t *p = orig->pixels[band], *q = dec->pixels[band];
        const size_t n = (size_t) orig->width * orig->height;
        size_t k;
        float norm = 0;                                 /* real_t, summed in file order (bin/pnmpsnr.c:92-101) */
        for (k = 0; k < n; k++) {
            int a = (p[k] >> 4) + 128, c = (q[k] >> 4) + 128;
            a = a < 0 ? 0 : a > 255 ? 255 : a;
            c = c < 0 ? 0 : c > 255 ? 255 : c;
            norm += (float) ((a - c) * (a - c));
        }
        norm /= (float) n;
        if (mse) mse[band] = norm;
        if (psnr_db) psnr_db[band] = norm > 1e-4 ? 10 * log(255.0 * 255.0 / norm) / log(10.0) : INFINITY;
    }
}

int fiasco_amd_batch_decode_psnr(const fiasco_amd_batch_t *b, unsigned i, double psnr_db[3], double mse[3])
{
    fa_image *dec;
    if (!b || i >= b->n || !b->jobs[i].status || !b->jobs[i].wfa) {
        fa_set_error("fiasco_amd_batch_decode_psnr: frame %u has no finished automaton", i);
        return 0;
    }
    if (b->jobs[i].frame_type != FA_I_FRAME) {
        fa_set_error("fiasco_amd_batch_decode_psnr: intra frames only (a P/B frame needs its reference frames)");
        return 0;
    }
    dec = decode_job(&b->jobs[i]);
    if (!dec) return 0;
    psnr_of(b->jobs[i].image, dec, psnr_db, mse);
    fa_image_free(dec);
    return 1;
}

/* All frames of the batch in ONE call of the core's decoder (the device runs them back to back on a stream, every
 * device of the process its share); psnr_db / mse: [n][3], either may be NULL.  Frames without a finished intra
 * automaton get zeros.  Returns the number of frames decoded. */
static void *psnr_thread(void *arg);
typedef struct psnr_share { const fiasco_amd_batch_t *b; fa_dec_job *d; double *psnr_db, *mse; unsigned first, stride; } psnr_share;
int fiasco_amd_batch_decode_psnr_all(const fiasco_amd_batch_t *b, double *psnr_db, double *mse)
{
    enum { MAXT = 16 };
    fa_dec_job *d;
    psnr_share sh[MAXT];
    pthread_t th[MAXT];
    int started[MAXT] = { 0 }, good;
    unsigned i, nt, t;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (!b || !b->n) { fa_set_error("fiasco_amd_batch_decode_psnr_all: empty batch"); return 0; }
    d = (fa_dec_job *) calloc(b->n, sizeof *d);
    if (!d) { fa_set_error("Out of memory!"); return 0; }
    for (i = 0; i < b->n; i++) {
        const fa_job *job = &b->jobs[i];
        if (!job->status || !job->wfa || job->frame_type != FA_I_FRAME) { d[i].skip = 1; continue; }
        d[i].wfa = job->wfa; d[i].width = job->image->width; d[i].height = job->image->height; d[i].color = job->image->color;
        d[i].frame_type = FA_I_FRAME;
    }
    good = fa_core_decode_frames(b->n, d);
    nt = ncpu > MAXT ? MAXT : ncpu < 1 ? 1 : (unsigned) ncpu;
    if (nt > b->n) nt = b->n;
    for (t = 0; t < nt; t++) { sh[t].b = b; sh[t].d = d; sh[t].psnr_db = psnr_db; sh[t].mse = mse; sh[t].first = t; sh[t].stride = nt; }
    for (t = 1; t < nt; t++) started[t] = pthread_create(&th[t], NULL, psnr_thread, &sh[t]) == 0;
    psnr_thread(&sh[0]);
    for (t = 1; t < nt; t++) { if (started[t]) pthread_join(th[t], NULL); else psnr_thread(&sh[t]); }
    for (i = 0; i < b->n; i++) {
        if (!d[i].skip && !d[i].out && d[i].errmsg[0]) fa_set_error("%s", d[i].errmsg);
        fa_image_free(d[i].out);
    }
    free(d);
    return good;
}
static void *psnr_thread(void *arg)
{
    psnr_share *s = (psnr_share *) arg;
    unsigned i, k;
    for (i = s->first; i < s->b->n; i += s->stride) {
        double p[3] = { 0, 0, 0 }, m[3] = { 0, 0, 0 };
        if (s->d[i].out) psnr_of(s->b->jobs[i].image, s->d[i].out, p, m);
        for (k = 0; k < 3; k++) { if (s->psnr_db) s->psnr_db[i * 3 + k] = p[k]; if (s->mse) s->mse[i * 3 + k] = m[k]; }
    }
    return NULL;
}

/* The decoded frame itself: band `band` of frame i as bytes, clip((pixel >> 4) + 128) in raster order
 * (width x height of the input) -- for a gray frame exactly the payload of the PGM that `dfiasco -s 0 -o`
 * writes (lib/image.c gray_write :449-483).  out must hold width * height bytes. */
int fiasco_amd_batch_decode_plane(const fiasco_amd_batch_t *b, unsigned i, unsigned band, unsigned char *out)
{
    const fa_image *orig;
    fa_image *dec;
    size_t k, n;
    if (!b || i >= b->n || !b->jobs[i].status || !b->jobs[i].wfa || !out) {
        fa_set_error("fiasco_amd_batch_decode_plane: frame %u has no finished automaton", i);
        return 0;
    }
    orig = b->jobs[i].image;
    if (b->jobs[i].frame_type != FA_I_FRAME || band >= (orig->color ? 3u : 1u)) {
        fa_set_error("fiasco_amd_batch_decode_plane: intra frames only, band < %u", orig->color ? 3u : 1u);
        return 0;
    }
    dec = decode_job(&b->jobs[i]);
    if (!dec) return 0;
    n = (size_t) orig->width * orig->height;
    for (k = 0; k < n; k++) {
        int v = (dec->pixels[band][k] >> 4) + 128;
        out[k] = (unsigned char) (v < 0 ? 0 : v > 255 ? 255 : v);
    }
    fa_image_free(dec);
    return 1;
}

void fiasco_amd_batch_free(fiasco_amd_batch_t *b)
{
    unsigned i;
    if (!b) return;
    if (b->staged) fa_core_unstage(b->staged);
    for (i = 0; i < b->n; i++) {
        if (b->jobs && b->jobs[i].wfa) fa_wfa_free(b->jobs[i].wfa);
        if (b->ims) fa_image_free(b->ims[i]);
        if (b->prev_ims) fa_image_free(b->prev_ims[i]);
        if (b->infos) fa_info_free(&b->infos[i]);
    }
    free(b->jobs); free(b->ims); free(b->prev_ims); free(b->infos);
    free(b);
}

fiasco_amd_batch_t *fiasco_amd_batch_stage(unsigned n, const unsigned char *const *pnm,
                                           const size_t *pnm_len, float quality,
                                           const fiasco_c_options_t *options)
{
    fiasco_c_options_t *defaults = NULL;
    const fa_options *op;
    fiasco_amd_batch_t *b;
    unsigned i;

    if (quality <= 0) { fa_set_error("Compression quality has to be positive."); return NULL; }
    if (options) { op = fa_cast_options(options); if (!op) return NULL; }
    else { defaults = fiasco_c_options_new(); if (!defaults) return NULL; op = fa_cast_options(defaults); }
    b = (fiasco_amd_batch_t *) calloc(1, sizeof *b);
    if (b) {
        b->jobs  = (fa_job *) calloc(n ? n : 1, sizeof *b->jobs);
        b->ims   = (fa_image **) calloc(n ? n : 1, sizeof *b->ims);
        b->infos = (fa_info *) calloc(n ? n : 1, sizeof *b->infos);
    }
    if (!b || !b->jobs || !b->ims || !b->infos) {
        fa_set_error("Out of memory!");
        if (defaults) fiasco_c_options_delete(defaults);
        fiasco_amd_batch_free(b);
        return NULL;
    }
    b->n = n;
    b->normal_domains = op->normal_domains;
    b->delta_domains  = op->delta_domains;
    b->prediction     = op->prediction;
    for (i = 0; i < n; i++) {
        fa_cparams cp;
        b->ims[i] = fa_image_from_pnm(pnm[i], pnm_len[i], "<memory>");
        if (!b->ims[i] || !fa_setup_params(op, quality, b->ims[i]->width, b->ims[i]->height,
                                          b->ims[i]->color, 1, &b->infos[i], &cp)
            || !prepare_job(&b->jobs[i], b->ims[i], &cp, op->basis_name)) {
            if (defaults) fiasco_c_options_delete(defaults);
            fiasco_amd_batch_free(b);
            return NULL;
        }
    }
    if (defaults) fiasco_c_options_delete(defaults);
    b->staged = fa_core_stage(n, b->jobs);
    return b;
}

/* the entropy writer of every finished frame (output/write.c:53-119 in the reference): a pure
 * function of the frame's automaton, so the frames of a batch are written by a few host
 * threads */
typedef struct { fiasco_amd_batch_t *b; unsigned char **outv; size_t *out_len; unsigned t, nt, good;
                 char err[256]; } wr_task;

/* developer aid: FIASCO_DUMP_WFA=<file> appends a text dump of every automaton handed to the
 * writer (diff the dumps of two cores to find what the per-call traces cannot show) */
static void dump_wfa(const fa_wfa *w)
{
    const char *path = fa_knob("FIASCO_DUMP_WFA");
    FILE *f;
    unsigned s, l, e;
    if (!path || !(f = fopen(path, "a"))) return;
    fprintf(f, "wfa states %u basis %u root %u\n", w->states, w->basis_states, w->root_state);
    for (s = 0; s < w->states; s++) {
        fprintf(f, "%u: fd %.9g lvl %u dt %u", s, w->final_distribution[s], w->level_of_state[s], w->domain_type[s]);
        for (l = 0; l < 2; l++) {
            fprintf(f, " | t %d xy %u,%u ys %d yc %u :", FA_TREE(w, s, l), w->x[s * 2 + l], w->y[s * 2 + l],
                    w->y_state[s * 2 + l], w->y_column[s * 2 + l]);
            for (e = 0; e < 6 && FA_INTO(w, s, l, e) != FA_NO_EDGE; e++)
                fprintf(f, " %d*%.9g", FA_INTO(w, s, l, e), FA_WEIGHT(w, s, l, e));
        }
        fprintf(f, "\n");
    }
    fclose(f);
}

static void *wr_thread(void *arg)
{
    wr_task *w = (wr_task *) arg;
    fiasco_amd_batch_t *b = w->b;
    unsigned i;
    for (i = w->t; i < b->n; i += w->nt) {
        fa_bitw out;
        if (!b->jobs[i].status) continue;
        if (b->n == 1) dump_wfa(b->jobs[i].wfa);
        fa_bw_init(&out);
        if (fa_write_frame(b->jobs[i].wfa, &b->infos[i], FA_I_FRAME, 0, b->prediction, b->normal_domains,
                           b->delta_domains, &out)) {
            w->out_len[i] = fa_bw_finish(&out);
            w->outv[i] = (unsigned char *) malloc(w->out_len[i]);
            if (w->outv[i]) {
                memcpy(w->outv[i], out.buf, w->out_len[i]);
                w->good++;
            } else {
                w->out_len[i] = 0;
                snprintf(w->err, sizeof w->err, "Out of memory!");
            }
        } else if (!w->err[0])       /* the last-error string is per thread: hand it to the caller */
            snprintf(w->err, sizeof w->err, "%s", fiasco_get_error_message());
        fa_bw_free(&out);
    }
    return NULL;
}

static unsigned write_streams(fiasco_amd_batch_t *b, unsigned char **outv, size_t *out_len)
{
    enum { MAXT = 16 };
    pthread_t th[MAXT];
    wr_task task[MAXT];
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    unsigned nt = b->n / 16, t, good = 0, i;
    if (nt > MAXT) nt = MAXT;
    if (ncpu > 0 && nt > (unsigned) ncpu) nt = (unsigned) ncpu;
    if (nt < 1) nt = 1;
    for (t = 0; t < nt; t++) {
        task[t].b = b; task[t].outv = outv; task[t].out_len = out_len;
        task[t].t = t; task[t].nt = nt; task[t].good = 0; task[t].err[0] = 0;
    }
    {
        int started[MAXT] = { 0 };
        for (t = 1; t < nt; t++)
            started[t] = pthread_create(&th[t], NULL, wr_thread, &task[t]) == 0;
        wr_thread(&task[0]);
        for (t = 1; t < nt; t++) {
            if (started[t]) pthread_join(th[t], NULL);
            else wr_thread(&task[t]);        /* no thread: the caller does that share */
        }
    }
    for (t = 0; t < nt; t++) {
        good += task[t].good;
        if (task[t].err[0]) fa_set_error("%s", task[t].err);     /* published after the join */
    }
    for (i = 0; i < b->n; i++)
        if (!b->jobs[i].status) fa_set_error("%s", b->jobs[i].errmsg);
    return good;
}

/* ---- replacing the inputs of a staged batch while a pass runs (a stream of batches) ---- */

typedef struct { fiasco_amd_batch_t *b; const unsigned char *const *pnm; const size_t *len;
                 int16_t *buf; const size_t *off; fa_image **out; unsigned t, nt, bad; char err[256]; } up_task;

static void *up_thread(void *arg)
{
    up_task *u = (up_task *) arg;
    fiasco_amd_batch_t *b = u->b;
    unsigned i;
    for (i = u->t; i < b->n; i += u->nt) {
        const fa_image *old = b->ims[i];
        u->out[i] = fa_image_from_pnm_into(u->pnm[i], u->len[i], "<memory>", old->width, old->height,
                                           old->color, u->buf + u->off[i]);
        if (!u->out[i]) {
            if (!u->bad) snprintf(u->err, sizeof u->err, "%s", fiasco_get_error_message());
            u->bad++;
        }
    }
    return NULL;
}

/* New frames for every slot of a staged batch (same sizes, colour model and options): parsed
 * by a few host threads straight into the core's upload staging memory and copied to the
 * device without waiting -- call it between submit and collect and the transfer overlaps the
 * pass that is running; the NEXT submit (or collect with resubmit) encodes the new frames. */
int fiasco_amd_batch_upload(fiasco_amd_batch_t *b, const unsigned char *const *pnm, const size_t *pnm_len)
{
    enum { MAXT = 32 };
    pthread_t th[MAXT];
    up_task task[MAXT];
    int started[MAXT] = { 0 };
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    unsigned nt, t, i, bad = 0;
    size_t total = 0, *off;
    int16_t *buf;
    fa_image **nims;

    if (!b || !b->staged || !b->n) { fa_set_error("Batch is not staged."); return 0; }
    off = (size_t *) malloc(b->n * sizeof *off);
    nims = (fa_image **) calloc(b->n, sizeof *nims);
    if (!off || !nims) { free(off); free(nims); fa_set_error("Out of memory!"); return 0; }
    for (i = 0; i < b->n; i++) {
        off[i] = total;
        total += (size_t) b->ims[i]->width * b->ims[i]->height * (b->ims[i]->color ? 3 : 1);
    }
    buf = fa_core_upload_buffer(b->staged, total * sizeof(int16_t));
    if (!buf) {
        free(off); free(nims);
        fa_set_error("No staging memory for %.1f MiB of frames.",
                     total * 2 / 1048576.0);
        return 0;
    }
    nt = b->n / 8;
    if (nt > MAXT) nt = MAXT;
    if (ncpu > 1 && nt > (unsigned) ncpu - 1) nt = (unsigned) ncpu - 1;
    if (nt < 1) nt = 1;
    for (t = 0; t < nt; t++) {
        task[t].b = b; task[t].pnm = pnm; task[t].len = pnm_len; task[t].buf = buf; task[t].off = off;
        task[t].out = nims; task[t].t = t; task[t].nt = nt; task[t].bad = 0; task[t].err[0] = 0;
    }
    for (t = 1; t < nt; t++) started[t] = pthread_create(&th[t], NULL, up_thread, &task[t]) == 0;
    up_thread(&task[0]);
    for (t = 1; t < nt; t++) {
        if (started[t]) pthread_join(th[t], NULL);
        else up_thread(&task[t]);
    }
    for (t = 0; t < nt; t++) {
        bad += task[t].bad;
        if (task[t].err[0]) fa_set_error("%s", task[t].err);
    }
    free(off);
    if (bad) {                                 /* nothing was replaced */
        for (i = 0; i < b->n; i++) fa_image_free(nims[i]);
        free(nims);
        return 0;
    }
    if (b->prev_ims) {
        for (i = 0; i < b->n; i++) fa_image_free(b->prev_ims[i]);
        free(b->prev_ims);
    }
    b->prev_ims = b->ims;                      /* alive until the next upload */
    b->ims = nims;
    for (i = 0; i < b->n; i++) b->jobs[i].image = nims[i];
    return fa_core_upload_commit(b->staged);
}

int fiasco_amd_batch_submit(fiasco_amd_batch_t *b)
{
    return b ? fa_core_submit(b->staged) : 0;
}

/* finish the submitted pass; with `resubmit` the next pass over the same resident inputs is
 * started before the host writes the streams of this one, so that the entropy writer of pass
 * i overlaps the device search of pass i+1 */
int fiasco_amd_batch_collect(fiasco_amd_batch_t *b, unsigned char **outv, size_t *out_len, int resubmit)
{
    unsigned i, good = 0;
    if (!b) return 0;
    for (i = 0; i < b->n; i++) { outv[i] = NULL; out_len[i] = 0; }
    fa_core_finish2(b->staged, resubmit);          /* the next pass touches device memory only */
    good = write_streams(b, outv, out_len);
    return (int) good;
}

int fiasco_amd_batch_encode(fiasco_amd_batch_t *b, unsigned char **outv, size_t *out_len)
{
    unsigned i, good = 0;
    if (!b) return 0;
    for (i = 0; i < b->n; i++) { outv[i] = NULL; out_len[i] = 0; }
    fa_core_run(b->staged);
    good = write_streams(b, outv, out_len);
    return (int) good;
}

int fiasco_amd_encode_batch(unsigned n, const unsigned char *const *pnm, const size_t *pnm_len,
                            float quality, const fiasco_c_options_t *options,
                            unsigned char **outv, size_t *out_len)
{
    unsigned i;
    int good;
    fiasco_amd_batch_t *b = fiasco_amd_batch_stage(n, pnm, pnm_len, quality, options);
    if (!b) {
        for (i = 0; i < n; i++) { outv[i] = NULL; out_len[i] = 0; }
        return 0;
    }
    good = fiasco_amd_batch_encode(b, outv, out_len);
    fiasco_amd_batch_free(b);
    return good;
}

/* include/libfiasco_amd_hip.h: which hot-path backend this library was linked with (the seam fa_core_*()) */
const char *fiasco_amd_core_name(void) { return fa_core_name(); }
