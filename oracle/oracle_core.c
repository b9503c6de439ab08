/*
 *  oracle_core.c -- TEST INFRASTRUCTURE: CPU restatement of the FIASCO encode-side hot
 *  path (partition search + matching pursuit + inner-product tables + rate models).
 *
 *  This file is the parity oracle for the HIP device coder.  It is NOT part of the
 *  product: only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may
 *  link or call it (it is built into oracle/liboracle_fiasco.so together with the host
 *  stream writer, never into libfiasco_amd.so).
 *
 *  Pinning: the .fco streams produced through this core are checked byte-for-byte
 *  against (a) the md5 / size known answers recorded in SURVEY.md §8c / Appendix C and
 *  (b) the committed golden streams under tests/golden/ that were generated with the
 *  real reference built by oracle/ref_build.sh (tests/test_oracle_pins.py).
 *
 *  Each function cites the reference file:line whose arithmetic it restates; float
 *  evaluation order is kept (sequential sums, no FMA: build with -ffp-contract=off).
 *  Single threaded, one frame at a time, plain C.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include "fa_host.h"

/* developer switches of the shared host code (fa_host.h): the oracle is test infrastructure, always on */
const char *fa_knob(const char *name) { return getenv(name); }

#define MAXED FA_MAXEDGES

/* ------------------------------------------------------------------ models */

/* "rle" domain-pool model, codec/domain-pool.c:621-630 (+ nested 1-domain qac model).
 * The states[] list is append-only along any live lineage of snapshots, so snapshots
 * share one array and differ only in n (see DESIGN.md "model snapshots"). */
typedef struct rle_model {
    int16_t  count[MAXED + 1];
    uint16_t total, n, max_domains, y_index;
    int16_t  d0_index;      /* qac model of domain 0: probability index */
    uint16_t d0_yindex;
    uint16_t d0_n;          /* 0 or 1 */
} rle_model;

typedef struct range {                  /* codec/cwfa.h:46-75 */
    unsigned x, y, image, address, level, global_address;
    float    weight[MAXED + 1];
    int16_t  into[MAXED + 1];
    int      tree;
    float    err, tree_bits, matrix_bits, weights_bits;
    fa_mv    mv;
    float    mv_tree_bits, mv_coord_bits, nd_tree_bits, nd_weights_bits;
    int      prediction;
} range;

/* a domain pool of the reference's registry (codec/domain-pool.c:188-236): "rle" (:621-879) and
 * "rle-no-chroma" (:886-899), the model-free "constant" pool {state 0} (:504-544), "adaptive" (qac: one
 * probability index per domain, :259-498), "basis" (the adaptive pool over the basis states, :546-560),
 * "uniform" (every usable state at one price, :562-615).  m.n / m.max_domains / m.y_index and states[] serve
 * the rle and the qac pools alike; qidx[] is qac_model_t.index. */
typedef struct pool {
    fa_pool_kind kind;
    int       constant;     /* kind == FA_POOL_CONSTANT */
    int       qac;          /* kind is FA_POOL_ADAPTIVE or FA_POOL_BASIS */
    rle_model m;
    int16_t  *states;       /* states[] of the lineage (see rle_model) */
    int16_t  *qidx;         /* qac: probability index per domain */
} pool;

/* "adaptive" coefficient model (codec/coeff.c:190-326): counts per context, int16 */
typedef struct cmodel {
    int           uniform;      /* "uniform" model (codec/coeff.c:133-188): mantissa bits + 1 per weight, no counts */
    int16_t      *cnt, *tot;
    const fa_rpf *rpf, *dc_rpf;
    unsigned      size, nt;
} cmodel;

/* motion_t (codec/cwfa.h:26-44): reference frames and the per-level displacement cost tables */
typedef struct motion {
    int             frame_type;
    const fa_image *past, *future;
    float          *fwd[FA_CAP_LEVEL + 2], *bwd[FA_CAP_LEVEL + 2];
    float           xbits[64];        /* == ybits: code lengths of the MPEG vector code */
    unsigned        range_size;
} motion;

typedef struct mpres {
    int16_t exclude[MAXED];
    int16_t indices[MAXED + 1];
    int16_t into[MAXED + 1];
    float   weight[MAXED];
    float   matrix_bits, weights_bits, err, costs;
} mpres;

typedef struct oc {
    const fa_cparams *cp;
    const fa_image   *im;
    fa_wfa           *w;
    unsigned lc_min, lc_max, images_level, products_level, max_elements;
    unsigned nimg, nprod, nlev;         /* table sizes */
    float    price;
    float   *images;                    /* [cap][nimg]  state images, levels 0..images_level */
    float   *ipis;                      /* [cap][nprod] <range sub-block, state> heap table  */
    float  **gram;                      /* [cap] -> nlev rows of (s+1) floats                */
    float   *pixels;                    /* 2^lc_max pixels of the current block, tree order  */
    unsigned ML;                        /* MAXLEVEL in force                                 */
    unsigned *tm;                       /* tree models: counts,total,p_counts,p_total [4*ML] */
    pool      pl[2];                    /* domain_pool, d_domain_pool */
    cmodel    cm[2];                    /* coeff, d_coeff */
    int16_t  *cbuf;                     /* counts + totals of both models, one block */
    size_t    cbuf_n;
    unsigned  coeff_min, coeff_max;
    float    *ipis_alt;                 /* second <sub-block, state> table: prediction residuals */
    int16_t  *planes[3];                /* pixels of the frame being coded (chroma: private copy
                                         * once the luminance motion is subtracted) */
    motion    mt;
    unsigned  p_min, p_max;
    float     m0[1024], m1[1024];       /* qac bit tables */
    /* matching-pursuit scratch, indexed by POSITION in the domain list */
    float    *rem_num, *rem_den, *ipdo; /* ipdo[d*MAXED + k] */
    uint8_t  *used;
    int16_t  *dlist;
    float     norm_ov[MAXED + 1], ipio[MAXED + 1];
    char      err[160];
    int       failed;
    FILE     *trace;                    /* FIASCO_ORACLE_TRACE: one record per approximate_range */
    int       trace_n;
} oc;

static float *img_of(oc *c, unsigned s) { return c->images + (size_t) s * c->nimg; }
static float *ipis_of(oc *c, unsigned s) { return c->ipis + (size_t) s * c->nprod; }
static int need_image(const fa_wfa *w, unsigned s) { return w->domain_type[s] != 0; }
static int usedomain(const fa_wfa *w, int s) { return w->domain_type[s] & FA_USE_DOMAIN; }

/* ------------------------------------------------------------------ rpf  (lib/rpf.c:59-169) */

static float btor(int b, const fa_rpf *r)
{
    uint32_t mant, bits;
    int sign, expo = 0;
    float v;
    if (b == -1) return 0;
    sign = b & 1;
    mant = ((uint32_t) b & ((1u << (r->mantissa_bits + 1)) - 1)) >> 1;
    mant <<= (23 - r->mantissa_bits);
    if (mant == 0)
        v = sign ? -1.0f : 1.0f;
    else {
        while (!(mant & (1u << 22))) { expo--; mant <<= 1; }
        mant <<= 1;
        bits = ((uint32_t) sign << 31) | ((uint32_t) (expo + 126) << 23) | (mant & 0x7fffffu);
        memcpy(&v, &bits, 4);
    }
    return v * r->range;
}

static float quant(float f, const fa_rpf *r) { return btor(fa_rtob(f, r), r); }

/* ------------------------------------------------------------------ tree model (codec/bintree.c) */

static void tree_init(unsigned *counts, unsigned *total, unsigned ML)
{
    static const unsigned c0[22] = {20,17,15,10,5,4,3,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1};
    static const unsigned c1[22] = {1,1,1,1,1,1,1,1,1,2,3,5,10,15,20,25,30,35,60,60,60,60};
    unsigned l;
    for (l = 0; l < ML; l++) {
        unsigned k = l < 22 ? l : 21;          /* limits extension: clamp */
        counts[l] = c1[k];
        total[l]  = c0[k] + c1[k];
    }
}

/* which: 0 = tree model, 1 = prediction tree model.  Index `level` may equal ML at the
 * image root of a level-22 image: it then aliases the next array exactly like the
 * reference's out-of-bounds read (codec/subdivide.c:253). */
static float tree_bits(const oc *c, int child, unsigned level, int which)
{
    const unsigned *counts = c->tm + (size_t) which * 2 * c->ML;
    float prob = counts[level] / (float) counts[c->ML + level];
    return child ? (float) -log2(prob) : (float) -log2(1 - prob);
}

static void tree_update(oc *c, int child, unsigned level, int which)
{
    unsigned *counts = c->tm + (size_t) which * 2 * c->ML;
    if (child) counts[level]++;
    counts[c->ML + level]++;
}

/* ------------------------------------------------------------------ qac tables (domain-pool.c:970-999) */

static void init_matrix_tables(oc *c)
{
    unsigned idx = 0, n, e;
    for (n = 1; n <= 9; n++)
        for (e = 0; e < (1u << n); e++, idx++) {
            c->m1[idx] = (float) -log2(1 / (float) (1 << n));
            c->m0[idx] = (float) -log2(1 - 1 / (float) (1 << n));
        }
    for (; idx < 1024; idx++) c->m0[idx] = c->m1[idx] = 0;
}

/* ------------------------------------------------------------------ domain pools (domain-pool.c) */

/* qac_append :448-464 */
static int qac_append(pool *pl, unsigned state)
{
    if (pl->m.n >= pl->m.max_domains) return 0;
    pl->qidx[pl->m.n] = pl->m.n > 0 ? pl->qidx[pl->m.n - 1] : 0;
    pl->states[pl->m.n++] = (int16_t) state;
    return 1;
}

/* alloc_domain_pool :203-236 and the six allocators behind it */
static void pool_init(oc *c, pool *pl, fa_pool_kind kind)
{
    unsigned m, s;
    unsigned max_domains = c->cp->pool_max_states;
    memset(&pl->m, 0, sizeof pl->m);
    pl->kind = kind;
    pl->constant = kind == FA_POOL_CONSTANT;
    pl->qac = kind == FA_POOL_ADAPTIVE || kind == FA_POOL_BASIS;
    if (!max_domains) max_domains = 1;              /* "Using at least DC component." :221-226 */
    if (pl->constant || kind == FA_POOL_UNIFORM) return;
    if (pl->qac) {
        pl->m.max_domains = (uint16_t) (kind == FA_POOL_BASIS ? c->w->basis_states : max_domains);
        for (s = 0; s < c->w->basis_states; s++)
            if (usedomain(c->w, (int) s)) qac_append(pl, s);
        return;
    }
    for (m = 0; m < MAXED + 1; m++) { pl->m.count[m] = 1; pl->m.total++; }
    pl->m.max_domains = (uint16_t) max_domains;
    for (s = 0; s < c->w->basis_states; s++)
        if (usedomain(c->w, (int) s)) {
            if (pl->m.n < pl->m.max_domains) {
                pl->states[pl->m.n++] = (int16_t) s;
                if (s == 0) { pl->m.d0_index = 0; pl->m.d0_n = 1; }
            }
        }
}

/* ->append: rle_append :832-852, qac_append :448-464, default_append :957-962 (the constant and the uniform
 * pool take everything) */
static int pool_append(pool *pl, unsigned state)
{
    if (pl->constant || pl->kind == FA_POOL_UNIFORM) return 1;
    if (pl->qac) return qac_append(pl, state);
    if (pl->m.n >= pl->m.max_domains) return 0;
    pl->states[pl->m.n++] = (int16_t) state;
    return 1;
}

/* -1 terminated candidate list: the pool states plus the co-located Y state if usable
 * and not already present (rle_generate :707-735, qac_generate :333-362); the constant pool is {0}
 * (:518-528), the uniform pool every usable state (:578-590).  Returns the list length. */
static unsigned pool_generate(oc *c, const pool *pl, int y_state, int16_t *out)
{
    unsigned n, len = pl->m.n;
    int present = 0;
    if (pl->constant) { out[0] = 0; out[1] = -1; return 1; }
    if (pl->kind == FA_POOL_UNIFORM) {
        unsigned state;
        for (state = 0, len = 0; state < c->w->states; state++)
            if (usedomain(c->w, (int) state)) out[len++] = (int16_t) state;
        out[len] = -1;
        return len;
    }
    if (y_state >= 0 && !usedomain(c->w, y_state)) y_state = -1;
    memcpy(out, pl->states, len * sizeof(int16_t));
    for (n = 0; n < len; n++) if (out[n] == y_state) present = 1;
    if (!present && y_state >= 0) out[len++] = (int16_t) y_state;
    out[len] = -1;
    return len;
}

static unsigned bits_bin_code(unsigned value, unsigned maxval)
{
    unsigned k = 0, v = maxval + 1, r;
    while (v >>= 1) k++;
    r = (maxval + 1) % (1u << k);
    return value < maxval + 1 - 2 * r ? k : k + 1;
}

/* price of the nested domain-0 model: qac_bits(:367-402) on a 1-domain list */
static float d0_bits(const oc *c, const rle_model *m, int y_state, int uses0)
{
    float b = 0;
    if (m->d0_n && 0 != y_state) b += c->m0[m->d0_index];
    if (y_state >= 0) b += c->m0[m->d0_yindex];
    if (uses0) {
        if (0 == y_state) { b -= c->m0[m->d0_yindex]; b += c->m1[m->d0_yindex]; }
        else              { b -= c->m0[m->d0_index];  b += c->m1[m->d0_index]; }
    }
    return b;
}

/* rle_bits :737-793, const_bits :530-536.  `used` = -1 terminated list of POSITIONS in
 * `domains`, or NULL. */
static float pool_bits(const oc *c, const pool *pl, const int16_t *domains, const int16_t *used, int y_state)
{
    const rle_model *m = &pl->m;
    int16_t sorted[MAXED + 1];
    unsigned n = 0, e, last;
    float bits;
    if (pl->constant) return 0;
    if (pl->kind == FA_POOL_UNIFORM) {
        /* uniform_bits :592-615.  `- n * log2 (..)` negates the UNSIGNED n first: the price of the empty
         * list is (2^32 - n) x log2((n - 1) / n), a large negative number -- reproduced, not repaired */
        unsigned state, nn = 0;
        for (state = 0; state < c->w->states; state++)
            if (usedomain(c->w, (int) state)) nn++;
        bits = (float) ((double) (0u - nn) * log2((double) ((float) (nn - 1) / (float) nn)));
        if (used)
            for (e = 0; used[e] != FA_NO_EDGE; e++) bits = (float) ((double) bits - log2(1.0 / nn));
        return bits;
    }
    if (y_state >= 0 && !usedomain(c->w, y_state)) y_state = -1;
    if (pl->qac) {
        /* qac_bits :364-402 */
        bits = 0;
        for (e = 0; e < m->n; e++)
            if (pl->states[e] != y_state) bits += c->m0[pl->qidx[e]];
        if (y_state >= 0) bits += c->m0[m->y_index];
        if (used)
            for (e = 0; used[e] != FA_NO_EDGE; e++) {
                if (domains[used[e]] == y_state) { bits -= c->m0[m->y_index]; bits += c->m1[m->y_index]; }
                else { bits -= c->m0[pl->qidx[used[e]]]; bits += c->m1[pl->qidx[used[e]]]; }
            }
        return bits;
    }
    if (used) {
        for (e = 0; used[e] != FA_NO_EDGE; e++)
            if (domains[used[e]] != y_state) sorted[n++] = used[e];
        /* the y-state terms the reference accumulates here are overwritten at :772 */
        for (e = 1; e < n; e++) {                     /* insertion sort, ascending */
            int16_t v = sorted[e]; unsigned j = e;
            while (j > 0 && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; j--; }
            sorted[j] = v;
        }
    }
    bits = (float) -log2(m->count[n] / (float) m->total);
    bits += d0_bits(c, m, y_state, used && n && sorted[0] == 0);
    last = 1;
    for (e = 0; e < n; e++) {
        int into = sorted[e];
        if (into && (unsigned) m->n - 1 - last) {
            bits += bits_bin_code((unsigned) into - last, (unsigned) m->n - 1 - last);
            last = (unsigned) into + 1;
        }
    }
    return bits;
}

/* rle_update :795-830 with the nested qac_update :404-446; default_update for the constant pool */
static void pool_update(oc *c, pool *pl, const int16_t *domains, const int16_t *used, int y_state)
{
    rle_model *m = &pl->m;
    int state_0 = 0, state_y = 0;
    unsigned edge = 0;
    if (pl->constant || pl->kind == FA_POOL_UNIFORM) return;      /* default_update */
    if (y_state >= 0 && !usedomain(c->w, y_state)) y_state = -1;
    if (pl->qac) {
        /* qac_update :404-446 */
        unsigned d;
        int used_y = 0, y_is_domain = 0;
        for (d = 0; d < m->n; d++) {
            pl->qidx[d]++;
            if (pl->states[d] == y_state) y_is_domain = 1;
        }
        if (used)
            for (edge = 0; used[edge] != FA_NO_EDGE; edge++) {
                if (domains[used[edge]] == y_state) {
                    if (y_is_domain) pl->qidx[used[edge]]--;
                    m->y_index >>= 1;
                    used_y = 1;
                } else {
                    pl->qidx[used[edge]]--;
                    pl->qidx[used[edge]] >>= 1;
                }
            }
        if (y_state >= 0 && !used_y) m->y_index++;
        for (d = 0; d < m->n; d++) if (pl->qidx[d] > 1020) pl->qidx[d] = 1020;
        if (m->y_index > 1020) m->y_index = 1020;
        return;
    }
    if (used)
        for (edge = 0; used[edge] != FA_NO_EDGE; edge++) {
            if (domains[used[edge]] == 0) state_0 = 1;
            else if (domains[used[edge]] == y_state) state_y = 1;
        }
    m->count[edge]++;
    m->total++;
    {   /* domain-0 model */
        int used_y = 0, y_is_domain = 0;
        if (m->d0_n) { m->d0_index++; if (0 == y_state) y_is_domain = 1; }
        if (state_0) {
            if (0 == y_state) { if (y_is_domain) m->d0_index--; m->d0_yindex >>= 1; used_y = 1; }
            else { m->d0_index--; m->d0_index >>= 1; }
        }
        if (y_state >= 0 && !used_y) m->d0_yindex++;
        if (m->d0_n && m->d0_index > 1020) m->d0_index = 1020;
        if (m->d0_yindex > 1020) m->d0_yindex = 1020;
    }
    if (state_y) m->y_index >>= 1; else m->y_index++;
    if (m->y_index > 1020) m->y_index = 1020;
}

/* rle_chroma :854-879 (only the normal pool is asked, codec/coder.c:779) */
static void pool_chroma(oc *c)
{
    pool *pl = &c->pl[0];
    rle_model *m = &pl->m;
    unsigned maxd = c->cp->chroma_max_states;
    /* default_chroma: the constant, uniform and rle-no-chroma pools stay as they are */
    if (pl->constant || pl->kind == FA_POOL_UNIFORM || pl->kind == FA_POOL_RLE_NO_CHROMA) return;
    if (pl->qac) {
        /* qac_chroma :466-498: the most referenced states keep the index they had */
        if (maxd < m->n) {
            int16_t *dom = fa_compute_hits(c->w->basis_states, c->w->states - 1, maxd, c->w);
            int16_t *st = (int16_t *) calloc(maxd + 1, sizeof(int16_t)), *ix = (int16_t *) calloc(maxd + 1, sizeof(int16_t));
            unsigned n, nw, old;
            for (n = 0; n < maxd && dom[n] >= 0; n++) st[n] = dom[n];
            if (n < maxd) maxd = n;
            free(dom);
            for (old = 0, nw = 0; nw < maxd && old < m->n; old++)
                if (pl->states[old] == st[nw]) ix[nw++] = pl->qidx[old];
            memcpy(pl->states, st, maxd * sizeof(int16_t));
            memcpy(pl->qidx, ix, maxd * sizeof(int16_t));
            free(st); free(ix);
            m->n = (uint16_t) maxd;
        }
        m->y_index = 0;
        m->max_domains = m->n;
        return;
    }
    if (maxd < m->n) {
        int16_t *dom = fa_compute_hits(c->w->basis_states, c->w->states - 1, maxd, c->w);
        unsigned n;
        for (n = 0; n < maxd && dom[n] >= 0; n++) pl->states[n] = dom[n];
        if (n < maxd) maxd = n;
        free(dom);
        m->n = (uint16_t) maxd;
    }
    m->y_index = 0;
    m->max_domains = m->n;
}

/* ------------------------------------------------------------------ aac coefficient model (coeff.c:190-326) */

static void coeff_init(oc *c)
{
    const fa_rpf *rp[2][2] = { { &c->cp->rpf, &c->cp->dc_rpf }, { &c->cp->d_rpf, &c->cp->d_dc_rpf } };
    unsigned k, i, levels;
    size_t off = 0;
    c->coeff_min = c->lc_min; c->coeff_max = c->lc_max;
    levels = c->coeff_max - c->coeff_min + 1;
    for (k = 0; k < 2; k++) {
        c->cm[k].uniform = (k ? c->cp->d_coeff_kind : c->cp->coeff_kind) == FA_COEFF_UNIFORM;
        c->cm[k].rpf = rp[k][0]; c->cm[k].dc_rpf = rp[k][1];
        c->cm[k].size = levels * (1u << (1 + rp[k][0]->mantissa_bits)) + (1u << (1 + rp[k][1]->mantissa_bits));
        c->cm[k].nt = levels + 1;
        off += c->cm[k].size + c->cm[k].nt;
    }
    c->cbuf_n = off;
    c->cbuf = (int16_t *) malloc(off * sizeof(int16_t));
    off = 0;
    for (k = 0; k < 2; k++) {
        cmodel *m = &c->cm[k];
        m->cnt = c->cbuf + off; off += m->size;
        m->tot = c->cbuf + off; off += m->nt;
        for (i = 0; i < m->size; i++) m->cnt[i] = 1;
        m->tot[0] = (int16_t) (1u << (1 + m->dc_rpf->mantissa_bits));
        for (i = 1; i < m->nt; i++) m->tot[i] = (int16_t) (1u << (1 + m->rpf->mantissa_bits));
    }
}

static int16_t *coeff_ctx(const oc *c, const cmodel *m, unsigned level)
{
    return m->cnt + (1u << (1 + m->dc_rpf->mantissa_bits))
           + (level - c->coeff_min) * (1u << (1 + m->rpf->mantissa_bits));
}

/* aac_bits :215-240.  A weight that quantises to RPF_ZERO indexes counts[-1]: in a level
 * context that is the last symbol of the context in front; in the DC context it lies in front
 * of the reference's array -- with glibc the upper half of the chunk header, 0, i.e. +inf bits */
static float coeff_bits(const oc *c, const cmodel *m, const float *wt, const int16_t *states, unsigned level)
{
    float bits = 0;
    const int16_t *ctx = coeff_ctx(c, m, level);
    unsigned e;
    if (m->uniform) {                                    /* uniform_bits, codec/coeff.c:155-170 */
        for (e = 0; states[e] != FA_NO_EDGE; e++)
            bits += (states[e] ? m->rpf : m->dc_rpf)->mantissa_bits + 1;
        return bits;
    }
    for (e = 0; states[e] != FA_NO_EDGE; e++)
        if (states[e])
            bits -= log2(ctx[fa_rtob(wt[e], m->rpf)] / (float) m->tot[level - c->coeff_min + 1]);
        else {
            int sym = fa_rtob(wt[e], m->dc_rpf);
            bits -= log2((sym < 0 ? 0 : m->cnt[sym]) / (float) m->tot[0]);
        }
    return bits;
}

static void coeff_update(oc *c, cmodel *m, const float *wt, const int16_t *states, unsigned level)
{
    int16_t *ctx = coeff_ctx(c, m, level);
    unsigned e;
    if (m->uniform) return;                              /* uniform_update */
    for (e = 0; states[e] != FA_NO_EDGE; e++)
        if (states[e]) {
            ctx[fa_rtob(wt[e], m->rpf)]++;
            m->tot[level - c->coeff_min + 1]++;
        } else {
            int sym = fa_rtob(wt[e], m->dc_rpf);
            if (sym >= 0) m->cnt[sym]++;
            m->tot[0]++;
        }
}

/* ------------------------------------------------------------------ inner products (codec/ip.c) */

static float dot_image_state(oc *c, unsigned address, unsigned level, unsigned dom)
{
    const float *p = c->pixels + (size_t) address * fa_size_of_level(level);
    const float *s = img_of(c, dom) + fa_address_of_level(level);
    float ip = 0;
    unsigned i;
    for (i = fa_size_of_level(level); i; i--) ip += *p++ * *s++;
    return ip;
}

static float dot_state_state(oc *c, unsigned d1, unsigned d2, unsigned level)
{
    const float *a = img_of(c, d1) + fa_address_of_level(level);
    const float *b = img_of(c, d2) + fa_address_of_level(level);
    float ip = 0;
    unsigned i;
    for (i = fa_size_of_level(level); i; i--) ip += *a++ * *b++;
    return ip;
}

static float ip_image_state(oc *c, unsigned image, unsigned address, unsigned level, unsigned dom)
{
    if (level <= c->images_level) return dot_image_state(c, address, level, dom);
    return ipis_of(c, dom)[image];
}

static float ip_state_state(oc *c, unsigned d1, unsigned d2, unsigned level)
{
    unsigned hi, lo;
    if (level <= c->images_level) return dot_state_state(c, d1, d2, level);
    hi = d1 > d2 ? d1 : d2; lo = d1 > d2 ? d2 : d1;
    return c->gram[hi][(size_t) (level - c->images_level - 1) * (hi + 1) + lo];
}

/* ip.c:72-154: tables <sub-block, state> for states from..states-1, all sub-blocks of
 * `image` down to images_level+1; per slot the additions run label 0 {child, edges},
 * label 1 {child, edges} */
static void compute_ip_images_state(oc *c, unsigned image, unsigned address, unsigned level,
                                    unsigned n, unsigned from)
{
    const fa_wfa *w = c->w;
    unsigned label, state;
    if (level <= c->images_level) return;
    if (level > c->images_level + 1)
        compute_ip_images_state(c, 2 * image + 1, address * 2, level - 1, 2 * n, from);
    for (label = 0; label < 2; label++)
        for (state = from; state < w->states; state++) {
            unsigned edge, cnt;
            int dom;
            float *dst;
            if (!need_image(w, state)) continue;
            if ((dom = FA_TREE(w, state, label)) != FA_RANGE) {
                dst = ipis_of(c, state) + image;
                if (level > c->images_level + 1) {
                    const float *src = ipis_of(c, (unsigned) dom) + image * 2 + label + 1;
                    for (cnt = n; cnt; cnt--, src += 2) *dst++ += *src;
                } else {
                    unsigned adr = address * 2 + label;
                    for (cnt = n; cnt; cnt--, adr += 2)
                        *dst++ += dot_image_state(c, adr, level - 1, (unsigned) dom);
                }
            }
            for (edge = 0; (dom = FA_INTO(w, state, label, edge)) != FA_NO_EDGE; edge++) {
                float wt = FA_WEIGHT(w, state, label, edge);
                dst = ipis_of(c, state) + image;
                if (level > c->images_level + 1) {
                    const float *src = ipis_of(c, (unsigned) dom) + image * 2 + label + 1;
                    for (cnt = n; cnt; cnt--, src += 2) *dst++ += *src * wt;
                } else {
                    unsigned adr = address * 2 + label;
                    for (cnt = n; cnt; cnt--, adr += 2)
                        *dst++ += wt * dot_image_state(c, adr, level - 1, (unsigned) dom);
                }
            }
        }
}

/* ip.c:184-260: Gram rows of states from..to at levels images_level+1 .. lc_max */
static void compute_ip_states_state(oc *c, unsigned from, unsigned to)
{
    const fa_wfa *w = c->w;
    unsigned level, s1, s2;
    for (level = c->images_level + 1; level <= c->lc_max; level++)
        for (s1 = from; s1 <= to; s1++)
            for (s2 = 0; s2 <= s1; s2++) {
                unsigned label;
                float ip = 0;
                if (!need_image(w, s2)) continue;
                for (label = 0; label < 2; label++) {
                    int d1, d2;
                    unsigned e1, e2;
                    float sum;
                    if ((d1 = FA_TREE(w, s1, label)) != FA_RANGE) {
                        sum = 0;
                        if ((d2 = FA_TREE(w, s2, label)) != FA_RANGE)
                            sum = ip_state_state(c, (unsigned) d1, (unsigned) d2, level - 1);
                        for (e2 = 0; (d2 = FA_INTO(w, s2, label, e2)) != FA_NO_EDGE; e2++)
                            sum += FA_WEIGHT(w, s2, label, e2)
                                   * ip_state_state(c, (unsigned) d1, (unsigned) d2, level - 1);
                        ip += sum;
                    }
                    for (e1 = 0; (d1 = FA_INTO(w, s1, label, e1)) != FA_NO_EDGE; e1++) {
                        float w1 = FA_WEIGHT(w, s1, label, e1);
                        sum = 0;
                        if ((d2 = FA_TREE(w, s2, label)) != FA_RANGE)
                            sum = ip_state_state(c, (unsigned) d1, (unsigned) d2, level - 1);
                        for (e2 = 0; (d2 = FA_INTO(w, s2, label, e2)) != FA_NO_EDGE; e2++)
                            sum += FA_WEIGHT(w, s2, label, e2)
                                   * ip_state_state(c, (unsigned) d1, (unsigned) d2, level - 1);
                        ip += w1 * sum;
                    }
                }
                c->gram[s1][(size_t) (level - c->images_level - 1) * (s1 + 1) + s2] = ip;
            }
}

/* ------------------------------------------------------------------ states (codec/control.c) */

/* control.c:205-258: state images at levels 1..images_level from the children's */
static void compute_images(oc *c, unsigned from, unsigned to)
{
    const fa_wfa *w = c->w;
    unsigned level, state, label;
    for (level = 1; level <= c->images_level; level++)
        for (state = from; state <= to; state++)
            for (label = 0; label < 2; label++) {
                unsigned half = fa_size_of_level(level - 1), edge, i;
                float *dst = img_of(c, state) + fa_address_of_level(level) + label * half;
                int dom;
                if ((dom = FA_TREE(w, state, label)) != FA_RANGE)
                    memcpy(dst, img_of(c, (unsigned) dom) + fa_address_of_level(level - 1),
                           half * sizeof(float));
                for (edge = 0; (dom = FA_INTO(w, state, label, edge)) != FA_NO_EDGE; edge++) {
                    const float *src = img_of(c, (unsigned) dom) + fa_address_of_level(level - 1);
                    float wt = FA_WEIGHT(w, state, label, edge);
                    for (i = 0; i < half; i++) dst[i] += src[i] * wt;
                }
            }
}

static void alloc_tables(oc *c, unsigned s)
{
    memset(img_of(c, s), 0, c->nimg * sizeof(float));
    memset(ipis_of(c, s), 0, c->nprod * sizeof(float));
    free(c->gram[s]);
    c->gram[s] = (float *) calloc((size_t) c->nlev * (s + 1) + 1, sizeof(float));
}

static void fail(oc *c, const char *msg)
{
    if (!c->failed) { c->failed = 1; snprintf(c->err, sizeof c->err, "%s", msg); }
}

/* control.c:48-131 */
static void append_state(oc *c, int auxiliary, float final, unsigned level_of_state)
{
    fa_wfa *w = c->w;
    unsigned s = w->states;
    w->final_distribution[s] = final;
    w->level_of_state[s] = (uint8_t) level_of_state;
    if (!auxiliary) {
        w->domain_type[s] = FA_USE_DOMAIN;
        alloc_tables(c, s);
        img_of(c, s)[0] = final;
        compute_images(c, s, s);
        compute_ip_states_state(c, s, s);
    } else {
        w->domain_type[s] = 0;
        free(c->gram[s]); c->gram[s] = NULL;
    }
    w->states++;
    if (w->states >= c->cp->limit_states) fail(c, "Maximum number of states reached!");
}

/* control.c:133-173 */
static void append_basis_states(oc *c)
{
    fa_wfa *w = c->w;
    unsigned s, n = w->basis_states;
    for (s = 0; s < n; s++) {
        alloc_tables(c, s);
        img_of(c, s)[0] = w->final_distribution[s];
        w->level_of_state[s] = (uint8_t) -1;
    }
    compute_images(c, 0, n - 1);
    compute_ip_states_state(c, 0, n - 1);
    w->states = n;
}

/* wfalib.c:152-180 */
static float final_distribution(const fa_wfa *w, unsigned state)
{
    unsigned label, edge;
    float f = 0;
    int dom;
    for (label = 0; label < 2; label++) {
        if ((dom = FA_TREE(w, state, label)) != FA_RANGE) f += w->final_distribution[dom];
        for (edge = 0; (dom = FA_INTO(w, state, label, edge)) != FA_NO_EDGE; edge++)
            f += FA_WEIGHT(w, state, label, edge) * w->final_distribution[dom];
    }
    return f / 2;
}

static void remove_states(oc *c, unsigned from)
{
    unsigned s;
    for (s = from; s < c->w->states; s++) { free(c->gram[s]); c->gram[s] = NULL; }
    fa_wfa_remove_states(c->w, from);
}

/* ------------------------------------------------------------------ matching pursuit (codec/approx.c) */

/* approx.c:644-699 */
static void orthogonalize(oc *c, unsigned index, unsigned n, unsigned level, const int16_t *dl)
{
    const float min_norm = 2e-3f;
    unsigned d, k;
    c->ipio[n]    = c->rem_num[index];
    c->norm_ov[n] = c->rem_den[index];
    for (d = 0; dl[d] >= 0; d++) {
        float t;
        if (c->used[d]) continue;
        t = ip_state_state(c, (unsigned) dl[index], (unsigned) dl[d], level);
        for (k = 0; k < n; k++)
            t -= c->ipdo[d * MAXED + k] / c->norm_ov[k] * c->ipdo[index * MAXED + k];
        c->ipdo[d * MAXED + n] = t;
        c->rem_den[d] -= t * t / c->norm_ov[n];
        c->rem_num[d] -= c->ipio[n] / c->norm_ov[n] * c->ipdo[d * MAXED + n];
        if (c->rem_den[d] / fa_size_of_level(level) < min_norm) c->used[d] = 1;
    }
}

/* approx.c:317-642 (the <v_l,o_n> refresh :554-569 feeds only itself and is omitted,
 * SURVEY §7.3 "dead code that looks live"; :570-571 are kept) */
static void matching_pursuit(oc *c, mpres *mp, int full_search, float price, unsigned max_edges,
                             int y_state, const range *rg, const pool *pl, const cmodel *cm)
{
    const float min_norm = 2e-3f;
    int16_t *dl = c->dlist;
    unsigned n, d, best_n = 0, size = fa_size_of_level(rg->level);
    int index;
    float norm, additional_bits;

    pool_generate(c, pl, y_state, dl);
    for (d = 0; dl[d] >= 0; d++) {
        c->used[d] = 0;
        c->rem_den[d] = ip_state_state(c, (unsigned) dl[d], (unsigned) dl[d], rg->level);
        if (c->rem_den[d] / size < min_norm)
            c->used[d] = 1;
        else
            c->rem_num[d] = ip_image_state(c, rg->image, rg->address, rg->level, (unsigned) dl[d]);
        if (!c->used[d] && fabs(c->rem_num[d]) < min_norm) c->used[d] = 1;
    }
    for (n = 0; mp->exclude[n] != FA_NO_EDGE; n++) c->used[mp->exclude[n]] = 1;

    for (norm = 0, n = 0; n < size; n++) {
        float p = c->pixels[(size_t) rg->address * size + n];
        norm += p * p;
    }
    additional_bits = rg->tree_bits + rg->mv_tree_bits + rg->mv_coord_bits + rg->nd_tree_bits
                      + rg->nd_weights_bits;                          /* approx.c:391-393 */

    mp->err = norm;
    mp->weights_bits = 0;
    mp->matrix_bits = pool_bits(c, pl, dl, NULL, y_state);
    mp->costs = (mp->matrix_bits + mp->weights_bits + additional_bits) * price + mp->err;

    n = 0;
    do {
        float min_matrix_bits = 0, min_weights_bits = 0, min_error = 0, min_weight[MAXED];
        float min_costs = full_search ? FA_MAXCOSTS : mp->costs;

        for (index = -1, d = 0; dl[d] >= 0; d++) {
            float matrix_bits, weights_bits;
            if (c->used[d]) continue;
            {   /* stage 1: price the candidate with placeholder weight 0.5 (:433-458) */
                int16_t vectors[MAXED + 1], states[MAXED + 1];
                float weights[MAXED + 1];
                unsigned i = 0, k;
                for (k = 0; k < n; k++)
                    if (mp->weight[k] != 0) {
                        vectors[i] = mp->indices[k];
                        states[i]  = dl[vectors[i]];
                        weights[i] = mp->weight[k];
                        i++;
                    }
                vectors[i] = (int16_t) d; states[i] = dl[d]; weights[i] = 0.5f;
                vectors[i + 1] = -1; states[i + 1] = -1;
                weights_bits = coeff_bits(c, cm, weights, states, rg->level);
                matrix_bits  = pool_bits(c, pl, dl, vectors, y_state);
            }
            if (((matrix_bits + weights_bits + additional_bits) * price + mp->err
                 - c->rem_num[d] * c->rem_num[d] / c->rem_den[d]) < min_costs) {
                unsigned k;
                int l;
                float m_bits, w_bits, r[MAXED], f[MAXED], costs, m_err;
                int v[MAXED];

                f[n] = c->rem_num[d] / c->rem_den[d];
                v[n] = (int) d;
                for (k = 0; k < n; k++) {
                    f[k] = c->ipio[k] / c->norm_ov[k];
                    v[k] = mp->indices[k];
                }
                for (l = (int) n; l >= 0; l--) {      /* back substitution, rounding each step */
                    const fa_rpf *rpf = dl[v[l]] ? cm->rpf : cm->dc_rpf;
                    r[l] = f[l] = quant(f[l], rpf);
                    for (k = 0; k < (unsigned) l; k++)
                        f[k] -= f[l] * c->ipdo[v[l] * MAXED + k] / c->norm_ov[k];
                }
                {
                    int16_t vectors[MAXED + 1], states[MAXED + 1];
                    float weights[MAXED + 1];
                    unsigned i = 0;
                    for (k = 0; k <= n; k++)
                        if (f[k] != 0) {
                            vectors[i] = (int16_t) v[k];
                            states[i]  = dl[v[k]];
                            weights[i] = f[k];
                            i++;
                        }
                    vectors[i] = -1; states[i] = -1;
                    w_bits = coeff_bits(c, cm, weights, states, rg->level);
                    m_bits = pool_bits(c, pl, dl, vectors, y_state);
                }
                c->norm_ov[n] = c->rem_den[d];
                c->ipio[n]    = c->rem_num[d];
                for (k = 0; k <= n; k++)
                    for (l = (int) k + 1; (unsigned) l <= n; l++)
                        r[k] += c->ipdo[v[l] * MAXED + k] * r[l] / c->norm_ov[k];
                m_err = norm;
                for (k = 0; k <= n; k++)
                    m_err += r[k] * r[k] * c->norm_ov[k] - 2 * r[k] * c->ipio[k];
                costs = (m_bits + w_bits + additional_bits) * price + m_err;
                if (costs < min_costs) {
                    index = (int) d;
                    min_costs = costs;
                    min_matrix_bits = m_bits;
                    min_weights_bits = w_bits;
                    min_error = m_err;
                    for (k = 0; k <= n; k++) min_weight[k] = f[k];
                }
            }
        }
        if (index >= 0) {
            if (min_costs < mp->costs) {
                unsigned k;
                mp->costs = min_costs;
                mp->err = min_error;
                mp->matrix_bits = min_matrix_bits;
                mp->weights_bits = min_weights_bits;
                for (k = 0; k <= n; k++) mp->weight[k] = min_weight[k];
                best_n = n + 1;
            }
            mp->indices[n] = (int16_t) index;
            mp->into[n] = dl[index];
            c->used[index] = 1;
            orthogonalize(c, (unsigned) index, n, rg->level, dl);
            n++;
        }
    } while (n < max_edges && index >= 0);

    mp->indices[best_n] = FA_NO_EDGE;
    mp->costs = (mp->matrix_bits + mp->weights_bits + additional_bits) * price + mp->err;
}

static int is_extreme(float wgt, const fa_rpf *rpf)
{
    return wgt == btor(fa_rtob(200, rpf), rpf) || wgt == btor(fa_rtob(-200, rpf), rpf);
}

/* approx.c:74-271 */
static float approximate_range(oc *c, float max_costs, float price, unsigned max_edges,
                               int y_state, range *rg, pool *pl, cmodel *cm)
{
    mpres mp;
    memset(&mp, 0, sizeof mp);
    mp.exclude[0] = FA_NO_EDGE;
    matching_pursuit(c, &mp, c->cp->full_search, price, max_edges, y_state, rg, pl, cm);

    if (c->cp->second_domain_block) {
        mpres t = mp;
        t.exclude[0] = t.indices[0];
        t.exclude[1] = FA_NO_EDGE;
        matching_pursuit(c, &t, c->cp->full_search, price, max_edges, y_state, rg, pl, cm);
        if (t.costs < mp.costs) mp = t;
    }
    if (c->cp->check_for_underflow) {
        int it = -1;
        mpres t = mp;
        do {
            int i;
            it++;
            t.exclude[it] = FA_NO_EDGE;
            for (i = 0; t.indices[i] != FA_NO_EDGE; i++)
                if (t.weight[i] == 0) { t.exclude[it] = t.indices[i]; break; }
            if (t.exclude[it] != FA_NO_EDGE) {
                t.exclude[it + 1] = FA_NO_EDGE;
                matching_pursuit(c, &t, c->cp->full_search, price, max_edges, y_state, rg, pl, cm);
                if (t.costs < mp.costs) mp = t;
            }
        } while (t.exclude[it] != FA_NO_EDGE && it < MAXED - 1);
    }
    if (c->cp->check_for_overflow) {
        int it = -1;
        mpres t = mp;
        do {
            int i;
            it++;
            t.exclude[it] = FA_NO_EDGE;
            for (i = 0; t.indices[i] != FA_NO_EDGE; i++) {
                const fa_rpf *rpf = t.indices[i] ? cm->rpf : cm->dc_rpf;
                if (is_extreme(t.weight[i], rpf)) { t.exclude[it] = t.indices[i]; break; }
            }
            if (t.exclude[it] != FA_NO_EDGE) {
                t.exclude[it + 1] = FA_NO_EDGE;
                matching_pursuit(c, &t, c->cp->full_search, price, max_edges, y_state, rg, pl, cm);
                if (t.costs < mp.costs) mp = t;
            }
        } while (t.exclude[it] != FA_NO_EDGE && it < MAXED - 1);
    }

    if (mp.costs < max_costs) {
        int ni = 0, oi, e;
        for (oi = 0; mp.indices[oi] != FA_NO_EDGE; oi++)
            if (mp.weight[oi] != 0) {
                mp.indices[ni] = mp.indices[oi];
                mp.into[ni]    = mp.into[oi];
                mp.weight[ni]  = mp.weight[oi];
                ni++;
            }
        mp.indices[ni] = FA_NO_EDGE;
        mp.into[ni]    = FA_NO_EDGE;
        pool_generate(c, pl, y_state, c->dlist);
        pool_update(c, pl, c->dlist, mp.indices, y_state);
        coeff_update(c, cm, mp.weight, mp.into, rg->level);
        for (e = 0; mp.indices[e] != FA_NO_EDGE; e++) {
            rg->into[e]   = mp.into[e];
            rg->weight[e] = mp.weight[e];
        }
        rg->into[e] = FA_NO_EDGE;
        rg->matrix_bits  = mp.matrix_bits;
        rg->weights_bits = mp.weights_bits;
        rg->err          = mp.err;
    } else {
        rg->into[0] = FA_NO_EDGE;
        mp.costs = FA_MAXCOSTS;
    }
    if (c->trace) {
        struct { int seq, level, image, D, states, nedges; float cost, err, mbits, wbits;
                 short into[6]; float w[5]; } t;
        int e;
        memset(&t, 0, sizeof t);
        t.seq = c->trace_n++; t.level = (int) rg->level; t.image = (int) rg->image;
        t.D = (int) (pl->constant ? 1 : pl->m.n); t.states = (int) c->w->states;
        t.cost = mp.costs; t.err = mp.err; t.mbits = mp.matrix_bits; t.wbits = mp.weights_bits;
        for (e = 0; e < 6; e++) t.into[e] = -1;
        for (e = 0; mp.costs < FA_MAXCOSTS && rg->into[e] != FA_NO_EDGE; e++) { t.into[e] = rg->into[e]; t.w[e] = rg->weight[e]; }
        t.nedges = e;
        fwrite(&t, sizeof t, 1, c->trace);
    }
    return mp.costs;
}

/* ------------------------------------------------------------------ partition search (codec/subdivide.c) */

/* subdivide.c:504-541: pixel/16 (C truncation), bintree (bit-interleaved) order */
static void cut_to_bintree(float *dst, const int16_t *src, unsigned sw, unsigned sh,
                           unsigned x0, unsigned y0, unsigned width, unsigned height)
{
    const unsigned mask01 = 0x555555, mask10 = 0xaaaaaa;
    unsigned x, y, xm, ym = 0;
    for (y = y0; y < y0 + height; y++, ym = (ym + mask10 + 1) & mask01) {
        xm = 0;
        for (x = x0; x < x0 + width; x++, xm = (xm + mask01 + 1) & mask10)
            dst[xm | ym] = (y >= sh || x >= sw) ? 0 : (float) (src[y * sw + x] / 16);
    }
}

/* subdivide.c:612-644 */
static void init_range(oc *c, range *rg, unsigned band)
{
    unsigned s;
    for (s = 0; s < c->w->states; s++)
        if (need_image(c->w, s)) memset(ipis_of(c, s), 0, c->nprod * sizeof(float));
    cut_to_bintree(c->pixels, c->planes[band], c->im->width, c->im->height, rg->x, rg->y,
                   fa_width_of_level(rg->level), fa_height_of_level(rg->level));
    rg->address = rg->image = 0;
    compute_ip_images_state(c, 0, 0, rg->level, 1, 0);
}

/* subdivide.c:549-610 */
static void init_new_state(oc *c, int auxiliary, int delta, range *rg, const range *child, const int *y_state)
{
    fa_wfa *w = c->w;
    unsigned label, s = w->states;
    int is_domain = 0;
    /* :571-581: a state that is not auxiliary is offered to the normal pool (unless it is a
     * delta state and delta_domains is off) and to the delta pool (if it is a delta state or
     * normal_domains is on); it keeps its image tables when either pool takes it */
    if (!auxiliary) {
        if (!delta || c->cp->delta_domains) is_domain = pool_append(&c->pl[0], s);
        if (delta || c->cp->normal_domains) is_domain = pool_append(&c->pl[1], s) || is_domain;
    }
    rg->into[0] = FA_NO_EDGE;
    rg->tree = (int) s;
    for (label = 0; label < 2; label++) {
        unsigned e;
        FA_TREE(w, s, label) = (int16_t) child[label].tree;
        w->y_state[s * 2 + label] = (int16_t) y_state[label];
        w->mv[s * 2 + label] = child[label].mv;
        w->x[s * 2 + label] = (uint16_t) child[label].x;
        w->y[s * 2 + label] = (uint16_t) child[label].y;
        w->prediction[s * 2 + label] = (uint8_t) child[label].prediction;
        w->y_column[s * 2 + label] = 0;
        for (e = 0; child[label].into[e] != FA_NO_EDGE; e++) {
            fa_wfa_append_edge(w, s, (unsigned) child[label].into[e], child[label].weight[e], label);
            if (child[label].into[e] == w->y_state[s * 2 + label]) w->y_column[s * 2 + label] = 1;
        }
    }
    w->delta_state[s] = (uint8_t) delta;
    append_state(c, !is_domain, final_distribution(w, s), rg->level);
}

static float fminf2(float a, float b) { return a > b ? b : a; }

/* ------------------------------------------------------------------ motion search (codec/mwfa.c) */

/* MPEG's vector-component code lengths (mv_code_table[][1], codec/mwfa.c:40-50) */
static const uint8_t mv_code_bits[33] = { 11, 11, 11, 11, 11, 11, 10, 10, 10, 8, 8, 8, 7, 5, 4, 3, 1,
                                          3, 4, 5, 7, 8, 8, 8, 10, 10, 10, 11, 11, 11, 11, 11, 11 };

static unsigned mt_sr(const oc *c) { return c->cp->search_range; }      /* full-pixel vectors only */

/* squared norm of original block - reference block (/16 per pixel, sequential float sum):
 * get_mcpe + mcpe_norm, codec/mwfa.c:610-684.  mc2 != NULL: interpolated prediction. */
static float mcpe_norm(const oc *c, unsigned x0, unsigned y0, unsigned width, unsigned height,
                       const int16_t *mc1, const int16_t *mc2)
{
    const int16_t *o = c->planes[0] + (size_t) y0 * c->im->width + x0;
    float norm = 0;
    unsigned x, y;
    for (y = 0; y < height; y++, o += c->im->width)
        for (x = 0; x < width; x++) {
            int16_t d = mc2 ? (int16_t) (o[x] - (mc1[y * width + x] + mc2[y * width + x]) / 2)
                            : (int16_t) (o[x] - mc1[y * width + x]);
            int q = d / 16;
            norm += (float) (q * q);
        }
    return norm;
}

static void get_mcpe(const oc *c, int16_t *mcpe, unsigned x0, unsigned y0, unsigned width, unsigned height,
                     const int16_t *mc1, const int16_t *mc2)
{
    const int16_t *o = c->planes[0] + (size_t) y0 * c->im->width + x0;
    unsigned x, y;
    for (y = 0; y < height; y++, o += c->im->width)
        for (x = 0; x < width; x++)
            *mcpe++ = mc2 ? (int16_t) (o[x] - (mc1[y * width + x] + mc2[y * width + x]) / 2)
                          : (int16_t) (o[x] - mc1[y * width + x]);
}

/* fill_norms_table, codec/mwfa.c:545-602 */
static void fill_norms_table(oc *c, unsigned x0, unsigned y0, unsigned level)
{
    const int sr = (int) mt_sr(c);
    const unsigned width = fa_width_of_level(level), height = fa_height_of_level(level);
    int16_t *blk = (int16_t *) malloc((size_t) width * height * sizeof(int16_t));
    unsigned index = 0;
    int mx, my;
    if (!c->mt.past || (c->mt.frame_type == FA_B_FRAME && !c->mt.future)) {
        /* e.g. an I frame as the future reference of a B frame: the reference coder dereferences
         * a null frame at the first motion search */
        fail(c, "Motion search without a reference frame (frame pattern).");
        free(blk);
        return;
    }
    for (my = -sr; my < sr; my++)
        for (mx = -sr; mx < sr; mx++, index++) {
            if ((int) x0 + mx < 0 || x0 + mx + width > c->im->width
                || (int) y0 + my < 0 || y0 + my + height > c->im->height) {
                c->mt.fwd[level][index] = 0.0f;
                c->mt.bwd[level][index] = 0.0f;
            } else {
                fa_extract_mc_block(blk, width, height, c->mt.past->pixels[0], c->mt.past->width, x0, y0, mx, my);
                c->mt.fwd[level][index] = mcpe_norm(c, x0, y0, width, height, blk, NULL);
                if (c->mt.frame_type == FA_B_FRAME) {
                    fa_extract_mc_block(blk, width, height, c->mt.future->pixels[0], c->mt.future->width,
                                        x0, y0, mx, my);
                    c->mt.bwd[level][index] = mcpe_norm(c, x0, y0, width, height, blk, NULL);
                }
            }
        }
    free(blk);
}

/* clear_norms_table / update_norms_table, codec/prediction.c:210-254 */
static void clear_norms_table(oc *c, unsigned level)
{
    if (level > c->p_min) {
        memset(c->mt.fwd[level], 0, c->mt.range_size * sizeof(float));
        memset(c->mt.bwd[level], 0, c->mt.range_size * sizeof(float));
    }
}

static void update_norms_table(oc *c, unsigned level)
{
    unsigned i;
    if (level > c->p_min) {
        for (i = 0; i < c->mt.range_size; i++) c->mt.fwd[level][i] += c->mt.fwd[level - 1][i];
        if (c->mt.frame_type == FA_B_FRAME)
            for (i = 0; i < c->mt.range_size; i++) c->mt.bwd[level][i] += c->mt.bwd[level - 1][i];
    }
}

/* find_best_mv, codec/mwfa.c:686-798 (full pixel) */
static float find_best_mv(const oc *c, float price, unsigned x0, unsigned y0, unsigned width, unsigned height,
                          float *bits, int *mx, int *my, const float *norms)
{
    const int sr = (int) mt_sr(c);
    float mincosts = FA_MAXCOSTS;
    unsigned index = 0;
    int x, y;
    *mx = *my = 0;
    for (y = -sr; y < sr; y++)
        for (x = -sr; x < sr; x++, index++)
            if ((int) x0 + x >= 0 && (int) y0 + y >= 0 && x0 + x + width <= c->im->width
                && y0 + y + height <= c->im->height) {
                float costs = norms[index] + (c->mt.xbits[x + sr] + c->mt.xbits[y + sr]) * price;
                if (costs < mincosts) { mincosts = costs; *mx = x; *my = y; }
            }
    *bits = c->mt.xbits[*mx + sr] + c->mt.xbits[*my + sr];
    return mincosts;
}

/* find_P_frame_mc, codec/mwfa.c:302-340 */
static void find_P_frame_mc(oc *c, int16_t *mcpe, float price, range *rg)
{
    const unsigned width = fa_width_of_level(rg->level), height = fa_height_of_level(rg->level);
    int16_t *blk = (int16_t *) malloc((size_t) width * height * sizeof(int16_t));
    int fx, fy;
    rg->mv_tree_bits = 1;
    rg->mv.type = FA_MV_FORWARD;
    find_best_mv(c, price, rg->x, rg->y, width, height, &rg->mv_coord_bits, &fx, &fy, c->mt.fwd[rg->level]);
    rg->mv.fx = (int16_t) fx; rg->mv.fy = (int16_t) fy;
    fa_extract_mc_block(blk, width, height, c->mt.past->pixels[0], c->mt.past->width, rg->x, rg->y, fx, fy);
    get_mcpe(c, mcpe, rg->x, rg->y, width, height, blk, NULL);
    free(blk);
}

/* find_B_frame_mc, codec/mwfa.c:342-543; cross_B_search is always off here: the reference sets
 * it from half_pixel_prediction (codec/coder.c:359) and half-pixel vectors are refused */
static void find_B_frame_mc(oc *c, int16_t *mcpe, float price, range *rg)
{
    const unsigned width = fa_width_of_level(rg->level), height = fa_height_of_level(rg->level);
    int16_t *b1 = (int16_t *) malloc((size_t) width * height * sizeof(int16_t));
    int16_t *b2 = (int16_t *) malloc((size_t) width * height * sizeof(int16_t));
    float forward_costs, backward_costs, interp_costs, forward_bits, backward_bits, interp_bits;
    int fx, fy, bx, by, type;
    forward_costs = find_best_mv(c, price, rg->x, rg->y, width, height, &forward_bits, &fx, &fy,
                                 c->mt.fwd[rg->level]) + 3 * price;
    backward_costs = find_best_mv(c, price, rg->x, rg->y, width, height, &backward_bits, &bx, &by,
                                  c->mt.bwd[rg->level]) + 3 * price;
    interp_bits = forward_bits + backward_bits;
    fa_extract_mc_block(b1, width, height, c->mt.past->pixels[0], c->mt.past->width, rg->x, rg->y, fx, fy);
    fa_extract_mc_block(b2, width, height, c->mt.future->pixels[0], c->mt.future->width, rg->x, rg->y, bx, by);
    interp_costs = mcpe_norm(c, rg->x, rg->y, width, height, b1, b2) + (interp_bits + 2) * price;
    if (forward_costs <= interp_costs) type = forward_costs <= backward_costs ? FA_MV_FORWARD : FA_MV_BACKWARD;
    else type = backward_costs <= interp_costs ? FA_MV_BACKWARD : FA_MV_INTERPOLATED;
    rg->mv.type = (int16_t) type;
    if (type == FA_MV_FORWARD) {
        rg->mv_tree_bits = 3; rg->mv_coord_bits = forward_bits;
        rg->mv.fx = (int16_t) fx; rg->mv.fy = (int16_t) fy;
        get_mcpe(c, mcpe, rg->x, rg->y, width, height, b1, NULL);
    } else if (type == FA_MV_BACKWARD) {
        rg->mv_tree_bits = 3; rg->mv_coord_bits = backward_bits;
        rg->mv.bx = (int16_t) bx; rg->mv.by = (int16_t) by;
        get_mcpe(c, mcpe, rg->x, rg->y, width, height, b2, NULL);
    } else {
        rg->mv_tree_bits = 2; rg->mv_coord_bits = interp_bits;
        rg->mv.fx = (int16_t) fx; rg->mv.fy = (int16_t) fy;
        rg->mv.bx = (int16_t) bx; rg->mv.by = (int16_t) by;
        get_mcpe(c, mcpe, rg->x, rg->y, width, height, b1, b2);
    }
    free(b1); free(b2);
}

/* subtract_mc, codec/mwfa.c:156-300: the luminance band's motion compensation, with the vector
 * components rounded to even, is taken off the chroma planes before they are coded */
static void subtract_mc(oc *c)
{
    const fa_wfa *w = c->w;
    unsigned state, label, band;
    int16_t *b1 = (int16_t *) malloc(fa_size_of_level(c->p_max) * sizeof(int16_t));
    int16_t *b2 = (int16_t *) malloc(fa_size_of_level(c->p_max) * sizeof(int16_t));
    for (state = w->basis_states; state < w->states; state++)
        for (label = 0; label < 2; label++) {
            const fa_mv *mv = &w->mv[state * 2 + label];
            const unsigned lv = (unsigned) w->level_of_state[state] - 1;
            const unsigned width = fa_width_of_level(lv), height = fa_height_of_level(lv);
            const unsigned x0 = w->x[state * 2 + label], y0 = w->y[state * 2 + label];
            if (mv->type == FA_MV_NONE) continue;
            for (band = 1; band < 3; band++) {
                int16_t *o = c->planes[band] + (size_t) y0 * c->im->width + x0;
                unsigned x, y;
                if (mv->type != FA_MV_BACKWARD)
                    fa_extract_mc_block(b1, width, height, c->mt.past->pixels[band], c->mt.past->width, x0, y0,
                                        (mv->fx / 2) * 2, (mv->fy / 2) * 2);
                if (mv->type != FA_MV_FORWARD)
                    fa_extract_mc_block(mv->type == FA_MV_BACKWARD ? b1 : b2, width, height,
                                        c->mt.future->pixels[band], c->mt.future->width, x0, y0,
                                        (mv->bx / 2) * 2, (mv->by / 2) * 2);
                for (y = 0; y < height; y++, o += c->im->width)
                    for (x = 0; x < width; x++)
                        o[x] = (int16_t) (o[x] - (mv->type == FA_MV_INTERPOLATED
                                                  ? (b1[y * width + x] + b2[y * width + x]) / 2 : b1[y * width + x]));
            }
        }
    free(b1); free(b2);
}

/* ------------------------------------------------------------------ prediction (codec/prediction.c) */

static float subdivide(oc *c, float max_costs, unsigned band, int y_state, range *rg, int prediction, int delta);

/* what store_state_data keeps of a state (codec/prediction.c:47-69, 502-565) */
typedef struct state_data {
    float   final_distribution;
    uint8_t level_of_state, domain_type;
    float  *images, *ipis, *gram;
    int16_t tree[2], y_state[2], into[2][MAXED + 1];
    uint8_t y_column[2], prediction[2];
    fa_mv   mv[2];
    uint16_t x[2], y[2];
    float   weight[2][MAXED + 1];
} state_data;

static state_data *store_state_data(oc *c, unsigned from, unsigned to)
{
    fa_wfa *w = c->w;
    state_data *data;
    unsigned s, l;
    if (to + 1 <= from) return NULL;
    data = (state_data *) calloc(to - from + 1, sizeof *data);
    for (s = from; s <= to; s++) {
        state_data *sd = &data[s - from];
        sd->final_distribution = w->final_distribution[s];
        sd->level_of_state = w->level_of_state[s];
        sd->domain_type = w->domain_type[s];
        /* the reference moves the table pointers; the tables here are rows of flat arrays */
        sd->images = (float *) malloc(c->nimg * sizeof(float));
        sd->ipis = (float *) malloc(c->nprod * sizeof(float));
        memcpy(sd->images, img_of(c, s), c->nimg * sizeof(float));
        memcpy(sd->ipis, ipis_of(c, s), c->nprod * sizeof(float));
        sd->gram = c->gram[s]; c->gram[s] = NULL;
        w->domain_type[s] = 0;
        for (l = 0; l < 2; l++) {
            sd->tree[l] = FA_TREE(w, s, l);
            sd->y_state[l] = w->y_state[s * 2 + l];
            sd->y_column[l] = w->y_column[s * 2 + l];
            sd->mv[l] = w->mv[s * 2 + l];
            sd->x[l] = w->x[s * 2 + l]; sd->y[l] = w->y[s * 2 + l];
            sd->prediction[l] = w->prediction[s * 2 + l];
            memcpy(sd->weight[l], &FA_WEIGHT(w, s, l, 0), sizeof sd->weight[l]);
            memcpy(sd->into[l], &FA_INTO(w, s, l, 0), sizeof sd->into[l]);
            FA_INTO(w, s, l, 0) = FA_NO_EDGE;
            FA_TREE(w, s, l) = FA_RANGE;
            w->y_state[s * 2 + l] = FA_RANGE;
        }
    }
    return data;
}

static void free_state_data(state_data *data, unsigned n)
{
    unsigned i;
    for (i = 0; i < n; i++) { free(data[i].images); free(data[i].ipis); free(data[i].gram); }
    free(data);
}

static void restore_state_data(oc *c, unsigned from, unsigned to, state_data *data)
{
    fa_wfa *w = c->w;
    unsigned s, l;
    if (to + 1 <= from) return;
    for (s = from; s <= to; s++) {
        state_data *sd = &data[s - from];
        w->final_distribution[s] = sd->final_distribution;
        w->level_of_state[s] = sd->level_of_state;
        w->domain_type[s] = sd->domain_type;
        memcpy(img_of(c, s), sd->images, c->nimg * sizeof(float));
        memcpy(ipis_of(c, s), sd->ipis, c->nprod * sizeof(float));
        free(c->gram[s]); c->gram[s] = sd->gram; sd->gram = NULL;
        for (l = 0; l < 2; l++) {
            FA_TREE(w, s, l) = sd->tree[l];
            w->y_state[s * 2 + l] = sd->y_state[l];
            w->y_column[s * 2 + l] = sd->y_column[l];
            w->mv[s * 2 + l] = sd->mv[l];
            w->x[s * 2 + l] = sd->x[l]; w->y[s * 2 + l] = sd->y[l];
            w->prediction[s * 2 + l] = sd->prediction[l];
            memcpy(&FA_WEIGHT(w, s, l, 0), sd->weight[l], sizeof sd->weight[l]);
            memcpy(&FA_INTO(w, s, l, 0), sd->into[l], sizeof sd->into[l]);
        }
    }
    free_state_data(data, to - from + 1);
    w->states = to + 1;
}

/* the residual of a predicted range is searched with fresh <sub-block, state> tables: the
 * reference swaps the per-state table pointers of the states that exist (prediction.c:302-309,
 * 443-450), here the second flat table takes their place.  Returns the table to go back to. */
static float *residual_tables_begin(oc *c, unsigned last_state)
{
    float *keep = c->ipis;
    c->ipis = c->ipis_alt;
    c->ipis_alt = keep;
    memset(c->ipis, 0, (size_t) (last_state + 1) * c->nprod * sizeof(float));
    return keep;
}

/* back to the tables of the block; `used`: the states appended meanwhile stay (their tables
 * are zeroed, :342-345, 481-484) */
static void residual_tables_end(oc *c, float *keep, unsigned last_state, int used)
{
    unsigned s;
    c->ipis_alt = c->ipis;
    c->ipis = keep;
    if (used)
        for (s = last_state + 1; s < c->w->states; s++)
            if (need_image(c->w, s)) memset(ipis_of(c, s), 0, c->nprod * sizeof(float));
}

static float range_costs(const range *rg, float price)
{
    return (rg->tree_bits + rg->matrix_bits + rg->weights_bits + rg->mv_tree_bits + rg->mv_coord_bits
            + rg->nd_tree_bits + rg->nd_weights_bits) * price + rg->err;
}

/* mc_prediction, codec/prediction.c:262-369 */
static float mc_prediction(oc *c, float max_costs, float price, unsigned band, int y_state, range *rg)
{
    range prange = *rg;
    const unsigned width = fa_width_of_level(rg->level), height = fa_height_of_level(rg->level);
    int16_t *mcpe = (int16_t *) calloc((size_t) width * height, sizeof(int16_t));
    float costs;

    if (prange.level == c->p_min) fill_norms_table(c, prange.x, prange.y, prange.level);
    if (!c->mt.past || (c->mt.frame_type == FA_B_FRAME && !c->mt.future))
        fail(c, "Motion search without a reference frame (frame pattern).");
    if (c->failed) { free(mcpe); return FA_MAXCOSTS; }
    if (c->mt.frame_type == FA_P_FRAME) find_P_frame_mc(c, mcpe, price, &prange);
    else find_B_frame_mc(c, mcpe, price, &prange);
    costs = (prange.mv_tree_bits + prange.mv_coord_bits) * price;
    if (costs < max_costs) {
        float *block_pixels = c->pixels, *tables;
        const unsigned last_state = c->w->states - 1;
        const float mvc = prange.mv_coord_bits, mvt = prange.mv_tree_bits;
        c->pixels = (float *) calloc(fa_size_of_level(c->lc_max), sizeof(float));
        cut_to_bintree(c->pixels, mcpe, width, height, 0, 0, width, height);
        tables = residual_tables_begin(c, last_state);
        prange.image = 0; prange.address = 0;
        prange.tree_bits = prange.matrix_bits = prange.weights_bits = 0;
        prange.mv_coord_bits = prange.mv_tree_bits = 0;
        prange.nd_weights_bits = prange.nd_tree_bits = 0;
        compute_ip_images_state(c, prange.image, prange.address, prange.level, 1, 0);
        costs += subdivide(c, max_costs - costs, band, y_state, &prange, 0, 1);
        if (costs < max_costs) {
            const unsigned img = rg->image, adr = rg->address;
            *rg = prange;
            rg->image = img; rg->address = adr;
            rg->mv_coord_bits = mvc; rg->mv_tree_bits = mvt;
            rg->prediction = 1;
            residual_tables_end(c, tables, last_state, 1);
            costs = range_costs(rg, price);
        } else {
            residual_tables_end(c, tables, last_state, 0);
            costs = FA_MAXCOSTS;
        }
        free(c->pixels);
        c->pixels = block_pixels;
    } else
        costs = FA_MAXCOSTS;
    free(mcpe);
    return costs;
}

/* nd_prediction, codec/prediction.c:371-500 */
static float nd_prediction(oc *c, float max_costs, float price, unsigned band, int y_state, range *rg)
{
    range lrange = *rg;
    float costs;
    {   /* the range's DC part, weight in the DC format of the normal model (:381-398) */
        const float x = ip_image_state(c, rg->image, rg->address, rg->level, 0);
        const float y = ip_state_state(c, 0, 0, rg->level);
        const float wgt = quant(x / y, c->cm[0].dc_rpf);
        const int16_t s[2] = { 0, -1 };
        lrange.into[0] = 0; lrange.into[1] = FA_NO_EDGE;
        lrange.weight[0] = wgt;
        lrange.mv_coord_bits = 0; lrange.mv_tree_bits = 0;
        lrange.nd_tree_bits = tree_bits(c, 0, lrange.level, 1);
        lrange.nd_weights_bits = 0;
        lrange.tree_bits = 0; lrange.matrix_bits = 0;
        lrange.weights_bits = coeff_bits(c, &c->cm[0], &wgt, s, rg->level);
    }
    costs = price * (lrange.weights_bits + lrange.nd_tree_bits);
    if (costs < max_costs) {
        float *block_pixels = c->pixels, *tables, *pixels;
        const unsigned last_state = c->w->states - 1, size = fa_size_of_level(rg->level);
        range rrange;
        unsigned n;
        {   /* original - approximation (:417-427) */
            const float wgt = -lrange.weight[0] * img_of(c, 0)[0];
            const float *src = c->pixels + (size_t) rg->address * size;
            pixels = (float *) calloc(fa_size_of_level(c->lc_max), sizeof(float));
            for (n = 0; n < size; n++) pixels[n] = src[n] + wgt;
            c->pixels = pixels;
        }
        rrange = *rg;
        rrange.tree_bits = rrange.matrix_bits = rrange.weights_bits = 0;
        rrange.mv_coord_bits = rrange.mv_tree_bits = 0;
        rrange.nd_tree_bits = rrange.nd_weights_bits = 0;
        rrange.image = 0; rrange.address = 0;
        tables = residual_tables_begin(c, last_state);
        compute_ip_images_state(c, rrange.image, rrange.address, rrange.level, 1, 0);
        costs += subdivide(c, max_costs - costs, band, y_state, &rrange, 0, 1);
        if (costs < max_costs && rrange.tree != FA_RANGE) {
            const unsigned img = rg->image, adr = rg->address;
            unsigned e;
            *rg = rrange;
            rg->image = img; rg->address = adr;
            rg->nd_tree_bits += lrange.nd_tree_bits;
            rg->nd_weights_bits += lrange.weights_bits;
            for (e = 0; lrange.into[e] != FA_NO_EDGE; e++) {
                rg->into[e] = lrange.into[e];
                rg->weight[e] = lrange.weight[e];
            }
            rg->into[e] = FA_NO_EDGE;
            rg->prediction = (int) e;
            residual_tables_end(c, tables, last_state, 1);
        } else {
            residual_tables_end(c, tables, last_state, 0);
            costs = FA_MAXCOSTS;
        }
        free(pixels);
        c->pixels = block_pixels;
    } else
        costs = FA_MAXCOSTS;
    return costs;
}

/* the models a subdivide() call may have to go back to */
typedef struct snapshot {
    rle_model pm[2];
    int16_t  *qi[2];        /* qac pools: the probability indices (qac_model_duplicate, :318-331) */
    int16_t  *cbuf;
    unsigned *tm;
    unsigned  states;
} snapshot;

static void snap_take(oc *c, snapshot *sn)
{
    unsigned k;
    sn->pm[0] = c->pl[0].m; sn->pm[1] = c->pl[1].m;
    for (k = 0; k < 2; k++) {
        sn->qi[k] = NULL;
        if (c->pl[k].qac) {
            sn->qi[k] = (int16_t *) malloc((c->pl[k].m.n + 1) * sizeof(int16_t));
            memcpy(sn->qi[k], c->pl[k].qidx, c->pl[k].m.n * sizeof(int16_t));
        }
    }
    sn->cbuf = (int16_t *) malloc(c->cbuf_n * sizeof(int16_t));
    memcpy(sn->cbuf, c->cbuf, c->cbuf_n * sizeof(int16_t));
    sn->tm = NULL;
}

static void snap_take_tm(oc *c, snapshot *sn)
{
    sn->tm = (unsigned *) malloc((size_t) 4 * c->ML * sizeof(unsigned));
    memcpy(sn->tm, c->tm, (size_t) 4 * c->ML * sizeof(unsigned));
}

static void snap_models(oc *c, const snapshot *sn)        /* pools + coefficient models */
{
    unsigned k;
    c->pl[0].m = sn->pm[0]; c->pl[1].m = sn->pm[1];
    for (k = 0; k < 2; k++)
        if (sn->qi[k]) memcpy(c->pl[k].qidx, sn->qi[k], sn->pm[k].n * sizeof(int16_t));
    memcpy(c->cbuf, sn->cbuf, c->cbuf_n * sizeof(int16_t));
}

static void snap_tm(oc *c, const snapshot *sn)
{
    memcpy(c->tm, sn->tm, (size_t) 4 * c->ML * sizeof(unsigned));
}

static void snap_free(snapshot *sn) { free(sn->cbuf); free(sn->tm); free(sn->qi[0]); free(sn->qi[1]); }

/* predict_range, codec/prediction.c:96-208.  `entry`: models at the entry of the enclosing
 * subdivide() call, `states`: wfa->states at that time. */
static float predict_range(oc *c, float max_costs, float price, range *rg, unsigned band, int y_state,
                           unsigned states, const snapshot *entry)
{
    fa_wfa *w = c->w;
    snapshot rec;
    const unsigned rec_states = w->states;
    state_data *rec_data;
    int16_t *tail[2] = { NULL, NULL };
    unsigned k;
    float costs;

    snap_take(c, &rec);
    snap_take_tm(c, &rec);
    rec_data = store_state_data(c, states, rec_states - 1);
    /* the pool snapshots share one states[] array per pool (see rle_model): what the recursion
     * appended behind the entry length is kept aside, the prediction appends there too */
    for (k = 0; k < 2; k++)
        if (!c->pl[k].constant && rec.pm[k].n > entry->pm[k].n) {
            tail[k] = (int16_t *) malloc((rec.pm[k].n - entry->pm[k].n) * sizeof(int16_t));
            memcpy(tail[k], c->pl[k].states + entry->pm[k].n, (rec.pm[k].n - entry->pm[k].n) * sizeof(int16_t));
        }
    w->states = states;
    snap_tm(c, entry);
    snap_models(c, entry);

    if (c->mt.frame_type == FA_I_FRAME) costs = nd_prediction(c, max_costs, price, band, y_state, rg);
    else costs = mc_prediction(c, max_costs, price, band, y_state, rg);

    if (costs < FA_MAXCOSTS) {
        if (rec_data) free_state_data(rec_data, rec_states - states);
        costs = range_costs(rg, price);
    } else {
        snap_models(c, &rec);
        snap_tm(c, &rec);
        for (k = 0; k < 2; k++)
            if (tail[k])
                memcpy(c->pl[k].states + entry->pm[k].n, tail[k], (rec.pm[k].n - entry->pm[k].n) * sizeof(int16_t));
        rg->prediction = 0;
        if (w->states != states) remove_states(c, states);
        restore_state_data(c, states, rec_states - 1, rec_data);
        costs = FA_MAXCOSTS;
    }
    free(tail[0]); free(tail[1]);
    snap_free(&rec);
    return costs;
}

/* ------------------------------------------------------------------ subdivide (codec/subdivide.c:60-502) */

static float subdivide(oc *c, float max_costs, unsigned band, int y_state, range *rg, int prediction, int delta)
{
    fa_wfa *w = c->w;
    float subdivide_costs, lincomb_costs, price;
    int new_y_state[2], try_mc, try_nd;
    unsigned label;
    snapshot entry, lc;
    range lrange, rrange, child[2];

    if (c->failed) return FA_MAXCOSTS;
    rg->into[0] = FA_NO_EDGE;
    rg->tree = FA_RANGE;
    if (rg->level < 3) return FA_MAXCOSTS;
    if (rg->x >= c->im->width || rg->y >= c->im->height) return 0;   /* not visible */

    try_mc = prediction && c->mt.frame_type != FA_I_FRAME && rg->level >= c->p_min && rg->level <= c->p_max
             && rg->x + fa_width_of_level(rg->level) <= c->im->width
             && rg->y + fa_height_of_level(rg->level) <= c->im->height;
    try_nd = prediction && c->mt.frame_type == FA_I_FRAME && rg->level >= c->p_min && rg->level <= c->p_max;
    if (try_mc) clear_norms_table(c, rg->level);

    if (rg->level == c->lc_max) init_range(c, rg, band);

    price = c->price;
    if (band != FA_Y) price *= c->cp->chroma_decrease;
    if (band != FA_Y)
        for (label = 0; label < 2; label++)
            new_y_state[label] = y_state != FA_RANGE ? FA_TREE(w, y_state, label) : FA_RANGE;
    else
        new_y_state[0] = new_y_state[1] = FA_RANGE;

    /* snapshot every model the recursion may touch */
    snap_take(c, &entry);
    snap_take_tm(c, &entry);
    entry.states = w->states;
    memset(&lrange, 0, sizeof lrange);

    if (rg->level <= c->lc_max) {
        lrange = *rg;
        lrange.tree = FA_RANGE;
        lrange.tree_bits = tree_bits(c, 0, lrange.level, 0);
        lrange.matrix_bits = 0;
        lrange.weights_bits = 0;
        lrange.mv_tree_bits = try_mc ? 1 : 0;       /* mc allowed but not used */
        lrange.mv_coord_bits = 0;
        lrange.nd_tree_bits = 0;
        lrange.nd_weights_bits = 0;
        lrange.prediction = 0;
        lincomb_costs = approximate_range(c, max_costs, price, c->max_elements, y_state, &lrange,
                                          &c->pl[delta ? 1 : 0], &c->cm[delta ? 1 : 0]);
    } else
        lincomb_costs = FA_MAXCOSTS;

    /* keep the models as modified by the linear combination, continue from the snapshot */
    snap_take(c, &lc);
    snap_models(c, &entry);

    if (rg->level > c->lc_min) {
        memset(child, 0, sizeof child);
        rrange = *rg;
        rrange.tree_bits = tree_bits(c, 1, rrange.level, 0);
        rrange.matrix_bits = 0;
        rrange.weights_bits = 0;
        rrange.err = 0;
        rrange.mv_tree_bits = try_mc ? 1 : 0;
        rrange.mv_coord_bits = 0;
        rrange.nd_tree_bits = try_nd ? tree_bits(c, 1, lrange.level, 1) : 0;
        rrange.nd_weights_bits = 0;
        rrange.prediction = 0;
        subdivide_costs = (rrange.tree_bits + rrange.weights_bits + rrange.matrix_bits + rrange.mv_tree_bits
                           + rrange.mv_coord_bits + rrange.nd_tree_bits + rrange.nd_weights_bits) * price;

        for (label = 0; label < 2; label++) {
            float remaining;
            child[label].image = rrange.image * 2 + label + 1;
            child[label].address = rrange.address * 2 + label;
            child[label].global_address = rrange.global_address * 2 + label;
            child[label].level = rrange.level - 1;
            child[label].x = (rrange.level & 1) ? rrange.x
                             : rrange.x + label * fa_width_of_level(rrange.level - 1);
            child[label].y = (rrange.level & 1)
                             ? rrange.y + label * fa_height_of_level(rrange.level - 1) : rrange.y;
            if (label && rrange.level <= c->lc_max)
                compute_ip_images_state(c, child[label].image, child[label].address,
                                        child[label].level, 1, entry.states);
            remaining = fminf2(lincomb_costs, max_costs) - subdivide_costs;
            if (remaining > 0)
                subdivide_costs += subdivide(c, remaining, band, new_y_state[label], &child[label],
                                             prediction, delta);
            else if (try_mc && child[label].level >= c->p_min)
                fill_norms_table(c, child[label].x, child[label].y, child[label].level);
            if (try_mc) update_norms_table(c, rrange.level);
            if (subdivide_costs >= fminf2(lincomb_costs, max_costs)) {
                subdivide_costs = FA_MAXCOSTS;
                break;
            }
            rrange.err             += child[label].err;
            rrange.tree_bits       += child[label].tree_bits;
            rrange.matrix_bits     += child[label].matrix_bits;
            rrange.weights_bits    += child[label].weights_bits;
            rrange.mv_tree_bits    += child[label].mv_tree_bits;
            rrange.mv_coord_bits   += child[label].mv_coord_bits;
            rrange.nd_weights_bits += child[label].nd_weights_bits;
            rrange.nd_tree_bits    += child[label].nd_tree_bits;
            tree_update(c, child[label].tree != FA_RANGE, child[label].level, 0);
            tree_update(c, !child[label].prediction, child[label].level, 1);
        }
    } else
        subdivide_costs = FA_MAXCOSTS;

    /* third alternative: predict the range, approximate the residual (:378-407) */
    if ((try_mc || try_nd) && !c->failed) {
        float pc = predict_range(c, fminf2(fminf2(lincomb_costs, subdivide_costs), max_costs), price, rg,
                                 band, y_state, entry.states, &entry);
        if (pc < FA_MAXCOSTS) {
            snap_free(&entry); snap_free(&lc);
            return pc;
        }
    }

    if (lincomb_costs >= FA_MAXCOSTS && subdivide_costs >= FA_MAXCOSTS) {
        snap_models(c, &entry);
        snap_tm(c, &entry);
        if (w->states != entry.states) remove_states(c, entry.states);
        subdivide_costs = FA_MAXCOSTS;
    } else if (lincomb_costs < subdivide_costs) {
        snap_models(c, &lc);
        snap_tm(c, &entry);
        *rg = lrange;
        if (w->states != entry.states) remove_states(c, entry.states);
        subdivide_costs = lincomb_costs;
    } else {
        int aux = band > FA_Y
                  || rg->x + fa_width_of_level(rg->level) > c->im->width
                  || rg->y + fa_height_of_level(rg->level) > c->im->height;
        /* the reference leaves through longjmp at the first "Maximum number of states
         * reached!" (codec/control.c:129-130); this restatement unwinds normally, so nothing
         * may be appended once the frame has failed */
        if (c->failed || w->states >= c->cp->limit_states) {
            fail(c, "Maximum number of states reached!");
            subdivide_costs = FA_MAXCOSTS;
        } else {
            init_new_state(c, aux, delta, &rrange, child, new_y_state);
            *rg = rrange;
        }
    }
    snap_free(&entry); snap_free(&lc);
    return subdivide_costs;
}

/* ------------------------------------------------------------------ frame (codec/coder.c:692-892) */

static void root_range(range *rg, unsigned level)
{
    memset(rg, 0, sizeof *rg);
    rg->level = level;
}

static void put_stats(fa_stats *st, float costs, const range *rg)
{
    st->costs = costs; st->err = rg->err; st->tree_bits = rg->tree_bits;
    st->matrix_bits = rg->matrix_bits; st->weights_bits = rg->weights_bits;
}

static int encode_one(fa_job *job)
{
    oc c;
    fa_wfa *w = job->wfa;
    const fa_cparams *cp = &job->cp;
    unsigned cap = cp->limit_states, s, l;
    const int inter = job->frame_type != FA_I_FRAME;
    range rg;
    float costs;

    memset(&c, 0, sizeof c);
    c.cp = cp; c.im = job->image; c.w = w;
    c.lc_min = cp->lc_min_level; c.lc_max = cp->lc_max_level;
    c.images_level = cp->images_level; c.products_level = cp->products_level;
    c.max_elements = cp->max_elements;
    c.price = cp->price;
    c.p_min = cp->p_min_level; c.p_max = cp->p_max_level;
    c.nimg = fa_size_of_tree(c.images_level);
    c.nprod = fa_size_of_tree(c.products_level);
    c.nlev = c.lc_max - c.images_level;
    c.ML = cp->limit_level;
    c.images = (float *) calloc((size_t) cap * c.nimg, sizeof(float));
    c.ipis = (float *) calloc((size_t) cap * c.nprod, sizeof(float));
    c.ipis_alt = (cp->prediction || inter) ? (float *) calloc((size_t) cap * c.nprod, sizeof(float)) : NULL;
    c.gram = (float **) calloc(cap, sizeof(float *));
    c.pixels = (float *) calloc(fa_size_of_level(c.lc_max), sizeof(float));
    c.tm = (unsigned *) calloc((size_t) 4 * c.ML + 4, sizeof(unsigned));
    c.pl[0].states = (int16_t *) calloc(cap + 2, sizeof(int16_t));
    c.pl[1].states = (int16_t *) calloc(cap + 2, sizeof(int16_t));
    c.pl[0].qidx = (int16_t *) calloc(cap + 2, sizeof(int16_t));
    c.pl[1].qidx = (int16_t *) calloc(cap + 2, sizeof(int16_t));
    c.rem_num = (float *) calloc(cap + 2, sizeof(float));
    c.rem_den = (float *) calloc(cap + 2, sizeof(float));
    c.ipdo = (float *) calloc((size_t) (cap + 2) * MAXED, sizeof(float));
    c.used = (uint8_t *) calloc(cap + 2, 1);
    c.dlist = (int16_t *) calloc(cap + 4, sizeof(int16_t));
    init_matrix_tables(&c);
    if (getenv("FIASCO_ORACLE_TRACE")) c.trace = fopen(getenv("FIASCO_ORACLE_TRACE"), "wb");
    for (s = 0; s < 3; s++) c.planes[s] = job->image->pixels[s];
    /* motion_t (alloc_motion, codec/mwfa.c:85-126) */
    c.mt.frame_type = job->frame_type;
    c.mt.past = job->past; c.mt.future = job->future;
    c.mt.range_size = (2 * cp->search_range) * (2 * cp->search_range);
    for (s = 0; s < 2 * cp->search_range && s < 33; s++) c.mt.xbits[s] = mv_code_bits[s];
    if (inter)
        for (l = c.p_min; l <= c.p_max; l++) {
            c.mt.fwd[l] = (float *) calloc(c.mt.range_size, sizeof(float));
            c.mt.bwd[l] = (float *) calloc(c.mt.range_size, sizeof(float));
        }

    append_basis_states(&c);
    tree_init(c.tm, c.tm + c.ML, c.ML);
    tree_init(c.tm + 2 * c.ML, c.tm + 3 * c.ML, c.ML);
    /* codec/coder.c:716-736: the delta pool is a second `rle' pool when residuals can occur,
     * else the constant pool */
    pool_init(&c, &c.pl[0], cp->pool_kind);
    pool_init(&c, &c.pl[1], (cp->prediction || inter) ? cp->d_pool_kind : FA_POOL_CONSTANT);
    coeff_init(&c);

    if (!job->image->color) {
        root_range(&rg, cp->level);
        costs = subdivide(&c, FA_MAXCOSTS, FA_GRAY, FA_RANGE, &rg, cp->prediction || inter, 0);
        put_stats(&job->stats[0], costs, &rg);
        if (!c.failed && rg.tree == FA_RANGE) fail(&c, "No root state generated!");
        else w->root_state = (unsigned) rg.tree;
    } else {
        int tree[3] = { FA_RANGE, FA_RANGE, FA_RANGE }, ycb = -1;
        int16_t *own[3] = { NULL, NULL, NULL };
        unsigned band;
        for (band = FA_Y; band <= FA_CR && !c.failed; band++) {
            if (band == FA_CB) {
                unsigned min_level = c.ML;    /* MAXLEVEL, codec/coder.c:785 */
                pool_chroma(&c);
                for (s = w->basis_states; s < w->states; s++) {
                    unsigned lin = (FA_TREE(w, s, 0) == FA_RANGE) + (FA_TREE(w, s, 1) == FA_RANGE);
                    if (lin && (unsigned) (w->level_of_state[s] - 1) < min_level)
                        min_level = (unsigned) (w->level_of_state[s] - 1);
                }
                c.lc_min = min_level;
                if (inter) {                  /* subtract mc of luminance (:798-799) */
                    size_t n = (size_t) job->image->width * job->image->height;
                    for (s = 1; s < 3; s++) {
                        own[s] = (int16_t *) malloc(n * sizeof(int16_t));
                        memcpy(own[s], job->image->pixels[s], n * sizeof(int16_t));
                        c.planes[s] = own[s];
                    }
                    subtract_mc(&c);
                }
            }
            root_range(&rg, cp->level);
            costs = subdivide(&c, FA_MAXCOSTS, band, tree[FA_Y], &rg, inter && band == FA_Y, 0);
            put_stats(&job->stats[band], costs, &rg);
            if (c.failed) break;
            if (rg.tree == FA_RANGE) { fail(&c, "No root state generated for color component!"); break; }
            tree[band] = rg.tree;
            if (band == FA_CB) {
                FA_TREE(w, w->states, 0) = (int16_t) tree[FA_Y];
                FA_TREE(w, w->states, 1) = (int16_t) tree[FA_CB];
                ycb = (int) w->states;
                append_state(&c, 1, final_distribution(w, w->states), cp->level + 1);
            }
        }
        if (!c.failed) {
            FA_TREE(w, w->states, 0) = (int16_t) tree[FA_CR];
            FA_TREE(w, w->states, 1) = FA_RANGE;
            append_state(&c, 1, final_distribution(w, w->states), cp->level + 1);
            FA_TREE(w, w->states, 0) = (int16_t) ycb;
            FA_TREE(w, w->states, 1) = (int16_t) (w->states - 1);
            append_state(&c, 1, final_distribution(w, w->states), cp->level + 2);
            w->root_state = w->states - 1;
        }
        free(own[1]); free(own[2]);
    }
    job->lc_min_level_out = c.lc_min;
    job->status = !c.failed;
    if (c.failed) snprintf(job->errmsg, sizeof job->errmsg, "%s", c.err);

    if (c.trace) fclose(c.trace);
    for (s = 0; s < cap; s++) free(c.gram[s]);
    for (l = 0; l < FA_CAP_LEVEL + 2; l++) { free(c.mt.fwd[l]); free(c.mt.bwd[l]); }
    free(c.images); free(c.ipis); free(c.ipis_alt); free(c.gram); free(c.pixels); free(c.tm);
    free(c.pl[0].states); free(c.pl[1].states); free(c.pl[0].qidx); free(c.pl[1].qidx);
    free(c.rem_num); free(c.rem_den); free(c.ipdo); free(c.used); free(c.dlist);
    free(c.cbuf);
    return job->status;
}

int fa_core_encode_frames(unsigned n, fa_job *jobs)
{
    unsigned i;
    int good = 0;
    for (i = 0; i < n; i++) good += encode_one(&jobs[i]);
    return good;
}

const char *fa_core_name(void) { return "oracle-cpu"; }

/* the decoder of the oracle build: the host restatement of codec/decoder.c / codec/motion.c (oracle_decoder.c) */
int fa_core_decode_frames(unsigned n, fa_dec_job *jobs)
{
    unsigned i;
    int good = 0;
    for (i = 0; i < n; i++) {
        fa_dec_job *j = &jobs[i];
        j->out = NULL; j->errmsg[0] = 0;
        if (j->skip) continue;
        j->out = fa_decode_image(j->width, j->height, j->wfa, j->color);
        if (j->out && j->frame_type != FA_I_FRAME && !fa_restore_mc(j->out, j->past, j->future, j->wfa, j->p_max_level)) {
            fa_image_free(j->out); j->out = NULL;
        }
        if (!j->out) snprintf(j->errmsg, sizeof j->errmsg, "%s", fiasco_get_error_message());
        else good++;
    }
    return good;
}
void fa_core_release_dev(void *dev, int dev_id) { (void) dev; (void) dev_id; }

/* staged form of the seam: the CPU oracle has nothing to make resident.  To behave like a
 * device that works while the host goes on, a submitted pass remembers WHICH frames it was
 * submitted with (the host may hand over the next ones before it collects, fa_core_upload_*)
 * and is computed when it is collected. */
struct oracle_staged {
    unsigned n; fa_job *jobs;
    const fa_image **snap;          /* frames of the submitted pass */
    int submitted;
    int16_t *up[2]; size_t up_bytes[2]; int up_cur;
};

void *fa_core_stage(unsigned n, fa_job *jobs)
{
    struct oracle_staged *s = (struct oracle_staged *) calloc(1, sizeof *s);
    s->n = n; s->jobs = jobs;
    s->snap = (const fa_image **) calloc(n ? n : 1, sizeof *s->snap);
    return s;
}

static int run_snapshot(struct oracle_staged *s)
{
    unsigned i;
    int good;
    for (i = 0; i < s->n; i++) {
        const fa_image *cur = s->jobs[i].image;
        if (s->jobs[i].wfa->states > s->jobs[i].wfa->basis_states)
            fa_wfa_remove_states(s->jobs[i].wfa, s->jobs[i].wfa->basis_states);
        if (s->submitted) { s->jobs[i].image = s->snap[i]; s->snap[i] = cur; }
    }
    good = fa_core_encode_frames(s->n, s->jobs);
    if (s->submitted)
        for (i = 0; i < s->n; i++) s->jobs[i].image = s->snap[i];
    s->submitted = 0;
    return good;
}

int fa_core_submit(void *h)
{
    struct oracle_staged *s = (struct oracle_staged *) h;
    unsigned i;
    if (!s) return 0;
    if (s->submitted) return 1;
    for (i = 0; i < s->n; i++) s->snap[i] = s->jobs[i].image;
    s->submitted = 1;
    return 1;
}

int fa_core_finish2(void *h, int resubmit)
{
    struct oracle_staged *s = (struct oracle_staged *) h;
    int good;
    if (!s) return 0;
    if (!s->submitted) fa_core_submit(h);
    good = run_snapshot(s);
    if (resubmit) fa_core_submit(h);
    return good;
}

int fa_core_finish(void *h) { return fa_core_finish2(h, 0); }
int fa_core_run(void *h) { return fa_core_submit(h) ? fa_core_finish(h) : 0; }

void fa_core_unstage(void *h)
{
    struct oracle_staged *s = (struct oracle_staged *) h;
    if (s) { free(s->up[0]); free(s->up[1]); free(s->snap); }
    free(s);
}

/* replacement inputs: plain memory, two buffers used alternately like the device's (the
 * frames of the pass in flight must stay intact) */
int16_t *fa_core_upload_buffer(void *h, size_t bytes)
{
    struct oracle_staged *s = (struct oracle_staged *) h;
    int p;
    if (!s) return NULL;
    p = s->up_cur ^= 1;
    if (bytes > s->up_bytes[p]) {
        free(s->up[p]);
        s->up[p] = (int16_t *) malloc(bytes);
        s->up_bytes[p] = s->up[p] ? bytes : 0;
    }
    return s->up[p];
}

int fa_core_upload_commit(void *h) { return h != NULL; }
