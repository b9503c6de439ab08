/*
 *  fa_writer.c -- .fco stream writer: a pure function of the finished automaton.
 *
 *  Stream layout and every coder below follow the reference writer:
 *    header           output/write.c:121-213
 *    frame            output/write.c:53-119
 *    tree             output/tree.c:46-176      (binary adaptive arithmetic coder)
 *    matrices         output/matrices.c:55-536  (column 0 QAC, #edges AC, index bin codes,
 *                                                chroma QAC columns)
 *    weights          output/weights.c:37-200   + lib/arith.c:196-306 (encode_array)
 *    interval rescale lib/arith.h:93-119
 *    nd               output/nd.c:49-242        (prediction tree + DC weights of predicted ranges)
 *    mc               output/mc.c:76-251        (motion tree VLC + MPEG vector codes)
 */
#include <stdlib.h>
#include <string.h>
#include "fa_host.h"

/* ------------------------------------------------------------ 16-bit interval coder */

typedef struct ac16 {
    uint16_t low, high;
    unsigned underflow;
    fa_bitw *out;
} ac16;

static void ac_init(ac16 *a, fa_bitw *out) { a->low = 0; a->high = 0xffff; a->underflow = 0; a->out = out; }

/* emit settled leading bits, count pending underflow bits (lib/arith.h:93-119) */
static void ac_rescale(ac16 *a)
{
    for (;;) {
        if (a->high < 0x8000) {
            fa_bw_put_bit(a->out, 0);
            for (; a->underflow; a->underflow--) fa_bw_put_bit(a->out, 1);
        } else if (a->low >= 0x8000) {
            fa_bw_put_bit(a->out, 1);
            for (; a->underflow; a->underflow--) fa_bw_put_bit(a->out, 0);
        } else if (a->high < 0xc000 && a->low >= 0x4000) {
            a->underflow++;
            a->high |= 0x4000;
            a->low  &= 0x3fff;
        } else
            break;
        a->high = (uint16_t) ((a->high << 1) | 1);
        a->low  = (uint16_t) (a->low << 1);
    }
}

static void ac_flush(ac16 *a)
{
    a->low = a->high;
    ac_rescale(a);
    fa_bw_align(a->out);
}

/* narrow [low,high] to the sub-interval [lo_c, hi_c) / scale */
static void ac_encode(ac16 *a, unsigned lo_c, unsigned hi_c, unsigned scale)
{
    unsigned range = (unsigned) (a->high - a->low) + 1;
    uint16_t base = a->low;
    a->high = (uint16_t) (base + (uint16_t) ((range * hi_c) / scale - 1));
    a->low  = (uint16_t) (base + (uint16_t) ((range * lo_c) / scale));
    ac_rescale(a);
}

/* quasi-arithmetic binary coder with shift probabilities (output/matrices.c:264-360) */
typedef struct qac {
    ac16 ac;
    unsigned index;
} qac;

static unsigned qac_shift(unsigned index)
{
    /* index blocks of size 2^n for n = 1..9 hold probability 2^-n */
    unsigned n = 1, start = 0;
    while (index >= start + (1u << n)) { start += 1u << n; n++; }
    return n;
}

static void qac_put(qac *q, int lps)
{
    ac16 *a = &q->ac;
    unsigned sh = qac_shift(q->index);
    if (!lps) {
        a->high = (uint16_t) (a->high - ((a->high - a->low) >> sh) - 1);
        ac_rescale(a);
        if (q->index < 1020) q->index++;
    } else {
        a->low = (uint16_t) (a->high - ((a->high - a->low) >> sh));
        ac_rescale(a);
        q->index >>= 1;
    }
}

/* ------------------------------------------------------------ header */

static void put_string(fa_bitw *out, const char *s, int limit)
{
    const char *p;
    for (p = s; p && *p && (!limit || p - s < 1024 - 2); p++)
        fa_bw_put_bits(out, (unsigned char) *p, 8);
    fa_bw_put_bits(out, 0, 8);
}

static void put_rpf(fa_bitw *out, const fa_rpf *r)
{
    fa_bw_put_bits(out, r->mantissa_bits - 2, 3);
    fa_bw_put_bits(out, (unsigned) r->range_e, 2);
}

static int rpf_differs(const fa_rpf *a, const fa_rpf *b)
{
    return a->mantissa_bits != b->mantissa_bits || a->range != b->range;
}

void fa_write_header(const fa_info *wi, fa_bitw *out)
{
    const unsigned k = 8;
    const char *p;
    for (p = "FIASCO"; *p; p++) fa_bw_put_bits(out, (unsigned char) *p, 8);
    fa_bw_put_bits(out, '\n', 8);
    put_string(out, wi->basis_name, 0);
    fa_bw_rice(out, 2, k);                    /* FIASCO_BINFILE_RELEASE, codec/wfa.h:25 */
    fa_bw_rice(out, 1, k);                    /* HEADER_TITLE   */
    put_string(out, wi->title, 1);
    fa_bw_rice(out, 2, k);                    /* HEADER_COMMENT */
    put_string(out, wi->comment, 1);
    fa_bw_rice(out, 0, k);                    /* HEADER_END     */
    fa_bw_rice(out, wi->max_states, k);
    fa_bw_put_bit(out, wi->color ? 1 : 0);
    fa_bw_rice(out, wi->width, k);
    fa_bw_rice(out, wi->height, k);
    if (wi->color) fa_bw_rice(out, wi->chroma_max_states, k);
    fa_bw_rice(out, wi->p_min_level, k);
    fa_bw_rice(out, wi->p_max_level, k);
    fa_bw_rice(out, wi->frames, k);
    fa_bw_rice(out, wi->smoothing, k);
    put_rpf(out, &wi->rpf);
    if (rpf_differs(&wi->rpf, &wi->dc_rpf)) { fa_bw_put_bit(out, 1); put_rpf(out, &wi->dc_rpf); }
    else fa_bw_put_bit(out, 0);
    if (rpf_differs(&wi->rpf, &wi->d_rpf)) { fa_bw_put_bit(out, 1); put_rpf(out, &wi->d_rpf); }
    else fa_bw_put_bit(out, 0);
    if (rpf_differs(&wi->dc_rpf, &wi->d_dc_rpf)) { fa_bw_put_bit(out, 1); put_rpf(out, &wi->d_dc_rpf); }
    else fa_bw_put_bit(out, 0);
    if (wi->frames > 1) {
        fa_bw_rice(out, wi->fps, k);
        fa_bw_rice(out, wi->search_range, k);
        fa_bw_put_bit(out, wi->half_pixel ? 1 : 0);
        fa_bw_put_bit(out, wi->B_as_past_ref ? 1 : 0);
    }
    fa_bw_align(out);
}

/* ------------------------------------------------------------ tree */

static int write_tree(const fa_wfa *wfa, fa_bitw *out)
{
    unsigned *queue = (unsigned *) malloc(sizeof(unsigned) * (wfa->states + 1));
    unsigned char *sym = (unsigned char *) malloc((size_t) wfa->states * 2 + 2);
    unsigned last = 1, cur, label, total = 0, n;
    unsigned sum0 = 1, sum1 = 11, scaling;
    ac16 a;
    if (!queue || !sym) { free(queue); free(sym); fa_set_error("Out of memory!"); return 0; }
    queue[0] = wfa->root_state;
    for (cur = 0; cur < last; cur++)           /* breadth first */
        for (label = 0; label < 2; label++) {
            int into = FA_TREE(wfa, queue[cur], label);
            if (into != FA_RANGE) { queue[last++] = (unsigned) into; sym[total++] = 1; }
            else sym[total++] = 0;
        }
    if (total != (wfa->states - wfa->basis_states) * 2) {
        fa_set_error("total [%d] != (states - basis_states) * 2 [%d]", (int) total,
                     (int) ((wfa->states - wfa->basis_states) * 2));
        free(queue); free(sym);
        return 0;
    }
    scaling = total / 20;
    ac_init(&a, out);
    for (n = 0; n < total; n++) {
        unsigned range = (unsigned) (a.high - a.low) + 1;
        if (!sym[n]) {
            a.high = (uint16_t) (a.low + (uint16_t) ((range * sum0) / sum1 - 1));
            ac_rescale(&a);
            sum0 = (uint16_t) (sum0 + 1);
        } else {
            a.low = (uint16_t) (a.low + (uint16_t) ((range * sum0) / sum1));
            ac_rescale(&a);
        }
        sum1 = (uint16_t) (sum1 + 1);
        if (sum1 > scaling) {
            sum0 >>= 1; sum1 >>= 1;
            if (!sum0) sum0 = 1;
            if (sum0 >= sum1) sum1 = sum0 + 1;
        }
    }
    ac_flush(&a);
    free(queue); free(sym);
    return 1;
}

/* ------------------------------------------------------------ matrices */

typedef struct rsort {
    uint16_t *state; uint8_t *label; uint16_t *max_domain; uint8_t *subdivided;
    unsigned n;
} rsort;

/* ranges in the order the coder visited them (codec/wfalib.c:658-696) */
static void sort_ranges(unsigned state, unsigned *domain, rsort *rs, const fa_wfa *wfa)
{
    unsigned label;
    for (label = 0; label < 2; label++) {
        if (FA_TREE(wfa, state, label) == FA_RANGE)
            rs->subdivided[rs->n] = 0;
        else {
            sort_ranges((unsigned) FA_TREE(wfa, state, label), domain, rs, wfa);
            rs->subdivided[rs->n] = 1;
        }
        rs->state[rs->n] = (uint16_t) state;
        rs->label[rs->n] = (uint8_t) label;
        rs->max_domain[rs->n] = (uint16_t) *domain;
        while (!(wfa->domain_type[rs->max_domain[rs->n]] & FA_USE_DOMAIN))
            rs->max_domain[rs->n]--;
        if (label == 1 || !rs->subdivided[rs->n]) rs->n++;
    }
    (*domain)++;
}

static unsigned edge_count(const fa_wfa *wfa, unsigned s, unsigned l)
{
    unsigned e = 0;
    while (FA_INTO(wfa, s, l, e) != FA_NO_EDGE) e++;
    return e;
}

static unsigned ilog2u(unsigned v) { unsigned k = 0; while (v >>= 1) k++; return k; }

static unsigned column0_encoding(const fa_wfa *wfa, unsigned last_row, fa_bitw *out)
{
    qac q;
    unsigned row, label, total = 0;
    ac_init(&q.ac, out); q.index = 0;
    for (row = wfa->basis_states; row <= last_row; row++)
        for (label = 0; label < 2; label++)
            if (FA_TREE(wfa, row, label) == FA_RANGE) {
                int uses0 = FA_INTO(wfa, row, label, 0) == 0;
                qac_put(&q, uses0);
                total += uses0;
            }
    ac_flush(&q.ac);
    return total;
}

static unsigned delta_encoding(int use_normal, int use_delta, const fa_wfa *wfa,
                               unsigned last_domain, fa_bitw *out)
{
    rsort rs;
    unsigned max_domain, total = 0, nslots = (last_domain + 1) * 2;
    rs.state      = (uint16_t *) calloc(nslots, sizeof(uint16_t));
    rs.label      = (uint8_t *)  calloc(nslots, 1);
    rs.max_domain = (uint16_t *) calloc(nslots, sizeof(uint16_t));
    rs.subdivided = (uint8_t *)  calloc(nslots, 1);
    rs.n = 0;
    max_domain = wfa->basis_states - 1;
    sort_ranges(last_domain, &max_domain, &rs, wfa);

    {   /* distribution of #edges, then the counts themselves through a static model */
        unsigned count[FA_MAXEDGES + 1] = {0}, cum[FA_MAXEDGES + 2];
        unsigned state, label, n, M = 0, r;
        ac16 a;
        for (state = wfa->basis_states; state <= last_domain; state++)
            for (label = 0; label < 2; label++)
                if (FA_TREE(wfa, state, label) == FA_RANGE) {
                    unsigned e = edge_count(wfa, state, label);
                    count[e]++;
                    if (e > M) M = e;
                }
        fa_bw_rice(out, M, 3);
        for (n = 0; n <= M; n++)
            fa_bw_rice(out, count[n], (unsigned) ((int) ilog2u(last_domain) - 2));
        cum[0] = 0;
        for (n = 1; n <= M + 1; n++) cum[n] = cum[n - 1] + count[n - 1];
        ac_init(&a, out);
        for (r = 0; r < rs.n; r++)
            if (!rs.subdivided[r]) {
                unsigned e = edge_count(wfa, rs.state[r], rs.label[r]);
                ac_encode(&a, (uint16_t) cum[e], (uint16_t) cum[e + 1], (uint16_t) cum[M + 1]);
            }
        ac_flush(&a);
    }
    {   /* domain indices */
        uint16_t *map1 = (uint16_t *) calloc(wfa->states, sizeof(uint16_t));
        uint16_t *map2 = (uint16_t *) calloc(wfa->states, sizeof(uint16_t));
        unsigned n1 = 0, n2 = 0, state, r;
        fa_bw_put_bit(out, (unsigned) use_normal);
        fa_bw_put_bit(out, (unsigned) use_delta);
        for (state = 0; state < wfa->states; state++) {
            int usable = wfa->domain_type[state] & FA_USE_DOMAIN;
            map1[state] = (uint16_t) n1;
            if (usable && (state < wfa->basis_states || use_delta || !wfa->delta_state[state])) n1++;
            map2[state] = (uint16_t) n2;
            if (usable && (state < wfa->basis_states || use_normal || wfa->delta_state[state])) n2++;
        }
        for (r = 0; r < rs.n; r++)
            if (!rs.subdivided[r]) {
                unsigned s = rs.state[r], l = rs.label[r], last = 1, e;
                const uint16_t *map = (wfa->delta_state[s] || wfa->mv[s * 2 + l].type != FA_MV_NONE)
                                      ? map2 : map1;               /* output/matrices.c:226-230 */
                unsigned max_value = map[rs.max_domain[r]];
                int dom;
                for (e = 0; (dom = FA_INTO(wfa, s, l, e)) != FA_NO_EDGE; e++)
                    if (dom > 0) {
                        total++;
                        if (max_value - last) {
                            fa_bw_bincode(out, map[dom] - last, max_value - last);
                            last = map[dom] + 1u;
                        }
                    }
            }
        free(map1); free(map2);
    }
    free(rs.state); free(rs.label); free(rs.max_domain); free(rs.subdivided);
    return total;
}

static unsigned chroma_encoding_n(const fa_wfa *wfa, unsigned chroma_max_states, fa_bitw *out)
{
    unsigned y_root = (unsigned) FA_TREE(wfa, (unsigned) FA_TREE(wfa, wfa->root_state, 0), 0);
    int16_t *y_domains = fa_compute_hits(wfa->basis_states, y_root, chroma_max_states, wfa);
    qac q;
    unsigned d, row, label, total = 0, next_index = 0;
    ac_init(&q.ac, out); q.index = 0;
    for (d = 0; y_domains[d] != -1; d++) {
        int save = 1;
        q.index = next_index;
        for (row = y_root + 1; row < wfa->states; row++) {
            for (label = 0; label < 2; label++)
                if (FA_TREE(wfa, row, label) == FA_RANGE) {
                    unsigned e;
                    int into, match = 0;
                    for (e = 0; (into = FA_INTO(wfa, row, label, e)) != FA_NO_EDGE
                                && (unsigned) into < row; e++)
                        if (into == y_domains[d] && into != wfa->y_state[row * 2 + label])
                            match = 1;
                    qac_put(&q, match);
                    total += (unsigned) match;
                }
            if (save) { next_index = q.index; save = 0; }
        }
    }
    q.index = 0;
    for (row = y_root + 1; row < wfa->states; row++)
        for (label = 0; label < 2; label++) {
            int yc = wfa->y_column[row * 2 + label] != 0;
            qac_put(&q, yc);
            total += (unsigned) yc;
        }
    ac_flush(&q.ac);
    free(y_domains);
    return total;
}

/* ------------------------------------------------------------ weights */

static void encode_array(fa_bitw *out, const unsigned *data, const unsigned *context,
                         const unsigned *c_symbols, unsigned n_context, unsigned n_data,
                         unsigned scaling)
{
    uint16_t **totals;
    unsigned c, n, i;
    ac16 a;
    if (!n_context) n_context = 1;
    totals = (uint16_t **) calloc(n_context, sizeof *totals);
    for (c = 0; c < n_context; c++) {
        /* one element in front of the table: the symbol may be RPF_ZERO = -1 (a leaf range that
         * is motion compensated but does not belong to a delta state is written in the NORMAL
         * format although the coder quantised it in the delta format, output/weights.c:140-170;
         * a weight too small for the normal format then has no symbol).  The reference reads
         * totals[-1] -- with glibc the upper half of the chunk header, 0 -- and goes on
         * (lib/arith.c:252-260, `int d`); so does this. */
        totals[c] = (uint16_t *) calloc(c_symbols[c] + 2, sizeof(uint16_t)) + 1;
        for (i = 0; i < c_symbols[c]; i++) totals[c][i + 1] = (uint16_t) (totals[c][i] + 1);
    }
    ac_init(&a, out);
    for (n = 0; n < n_data; n++) {
        int d = (int) data[n];
        uint16_t *t;
        c = n_context > 1 ? context[n] : 0;
        t = totals[c];
        ac_encode(&a, t[d], t[d + 1], t[c_symbols[c]]);
        for (i = (unsigned) (d + 1); i < c_symbols[c] + 1; i++) t[i]++;
        if (t[c_symbols[c]] > scaling)
            for (i = 1; i < c_symbols[c] + 1; i++) {
                t[i] >>= 1;
                if (t[i] <= t[i - 1]) t[i] = (uint16_t) (t[i - 1] + 1);
            }
    }
    ac_flush(&a);
    for (c = 0; c < n_context; c++) free(totals[c] - 1);
    free(totals);
}

static int write_weights(unsigned total, const fa_wfa *wfa, const fa_info *wi, fa_bitw *out)
{
    unsigned state, label, off1, off2, off3, off4, nw = 0, i;
    int min_level = FA_CAP_LEVEL + 8, max_level = 0, d_min = FA_CAP_LEVEL + 8, d_max = 0;
    int dc = 0, d_dc = 0, delta_approx = 0;
    unsigned *w_arr, *l_arr, *c_symbols;

    for (state = wfa->basis_states; state < wfa->states; state++)
        if (wfa->delta_state[state]) { delta_approx = 1; break; }
    /* the reference starts the minima at MAXLEVEL; any value above every real level works */
    for (state = wfa->basis_states; state < wfa->states; state++)
        for (label = 0; label < 2; label++)
            if (FA_TREE(wfa, state, label) == FA_RANGE) {
                int lv = (int) wfa->level_of_state[state] - 1;
                if (delta_approx && wfa->delta_state[state]) {
                    if (lv < d_min) d_min = lv;
                    if (lv > d_max) d_max = lv;
                    if (FA_INTO(wfa, state, label, 0) == 0) d_dc = 1;
                } else {
                    if (lv < min_level) min_level = lv;
                    if (lv > max_level) max_level = lv;
                    if (FA_INTO(wfa, state, label, 0) == 0) dc = 1;
                }
            }
    if (min_level > max_level) max_level = min_level - 1;
    if (d_min > d_max) d_max = d_min - 1;
    off1 = dc ? 1 : 0;
    off2 = off1 + (d_dc ? 1 : 0);
    off3 = off2 + (unsigned) (max_level - min_level + 1);
    off4 = off3 + (unsigned) (d_max - d_min + 1);

    w_arr = (unsigned *) calloc(total, sizeof(unsigned));
    l_arr = (unsigned *) calloc(total, sizeof(unsigned));
    for (state = wfa->basis_states; state < wfa->states; state++)
        for (label = 0; label < 2; label++)
            if (FA_TREE(wfa, state, label) == FA_RANGE) {
                unsigned e;
                int dom;
                for (e = 0; (dom = FA_INTO(wfa, state, label, e)) != FA_NO_EDGE; e++) {
                    int is_delta = delta_approx && wfa->delta_state[state];
                    float wt = FA_WEIGHT(wfa, state, label, e);
                    if (nw >= total) {
                        fa_set_error("Can't write more than %d weights.", (int) total);
                        free(w_arr); free(l_arr);
                        return 0;
                    }
                    if (dom) {
                        if (is_delta) {
                            w_arr[nw] = (unsigned) fa_rtob(wt, &wi->d_rpf);
                            l_arr[nw] = off3 + wfa->level_of_state[state] - 1 - (unsigned) d_min;
                        } else {
                            w_arr[nw] = (unsigned) fa_rtob(wt, &wi->rpf);
                            l_arr[nw] = off2 + wfa->level_of_state[state] - 1 - (unsigned) min_level;
                        }
                    } else {
                        w_arr[nw] = (unsigned) fa_rtob(wt, is_delta ? &wi->d_dc_rpf : &wi->dc_rpf);
                        l_arr[nw] = is_delta ? off1 : 0;
                    }
                    nw++;
                }
            }
    c_symbols = (unsigned *) calloc(off4 ? off4 : 1, sizeof(unsigned));
    c_symbols[0] = 1u << (wi->dc_rpf.mantissa_bits + 1);
    if (off1 != off2) c_symbols[off1] = 1u << (wi->d_dc_rpf.mantissa_bits + 1);
    for (i = off2; i < off3; i++) c_symbols[i] = 1u << (wi->rpf.mantissa_bits + 1);
    for (; i < off4; i++) c_symbols[i] = 1u << (wi->d_rpf.mantissa_bits + 1);
    encode_array(out, w_arr, l_arr, c_symbols, off4, total, 500);
    free(c_symbols); free(w_arr); free(l_arr);
    return 1;
}

/* ------------------------------------------------------------ frame */

/* codec/wfalib.c:698-732: a child reached through a (state,label) that also carries
 * edges (or lies below a delta state) holds a prediction residual */
static void locate_delta_images(fa_wfa *wfa)
{
    int state;
    unsigned label;
    for (state = (int) wfa->root_state; state >= (int) wfa->basis_states; state--)
        wfa->delta_state[state] = 0;
    for (state = (int) wfa->root_state; state >= (int) wfa->basis_states; state--)
        for (label = 0; label < 2; label++)
            if (FA_TREE(wfa, state, label) != FA_RANGE)
                if (wfa->mv[state * 2 + label].type != FA_MV_NONE
                    || FA_INTO(wfa, state, label, 0) != FA_NO_EDGE || wfa->delta_state[state])
                    wfa->delta_state[FA_TREE(wfa, state, label)] = 1;
}

/* ------------------------------------------------------------ prediction (output/nd.c) */

/* encode_nd_tree :65-177: breadth first, one binary decision per child between p_min_level and
 * p_max_level ("the child's range is predicted": the (state,label) carries edges as well);
 * adaptive binary coder, counts halved above 50.  Stores the number of predicted ranges in *nused;
 * returns 0 when the queue cannot be allocated (the reference aborts there). */
static int write_nd_tree(const fa_wfa *wfa, const fa_info *wi, fa_bitw *out, unsigned *nused)
{
    unsigned *queue = (unsigned *) malloc(sizeof(unsigned) * (wfa->states + 1));
    unsigned head = 0, tail = 0, used = 0, label;
    unsigned sum0 = 1, sum1 = 11;
    ac16 a;
    *nused = 0;
    if (!queue) { fa_set_error("Out of memory!"); return 0; }
    ac_init(&a, out);
    queue[tail++] = wfa->root_state;
    while (head < tail) {
        const unsigned next = queue[head++];
        if (wfa->level_of_state[next] > wi->p_max_level + 1) {
            for (label = 0; label < 2; label++)
                if (FA_TREE(wfa, next, label) != FA_RANGE) queue[tail++] = (unsigned) FA_TREE(wfa, next, label);
        } else if (wfa->level_of_state[next] > wi->p_min_level) {
            for (label = 0; label < 2; label++) {
                const int child = FA_TREE(wfa, next, label);
                unsigned range;
                if (child == FA_RANGE) continue;
                range = (unsigned) (a.high - a.low) + 1;
                if (FA_INTO(wfa, next, label, 0) != FA_NO_EDGE) {       /* prediction used: '1' */
                    used++;
                    a.low = (uint16_t) (a.low + (uint16_t) ((range * sum0) / sum1));
                    ac_rescale(&a);
                } else {                                                 /* '0': go on below */
                    if (wfa->level_of_state[child] > wi->p_min_level) queue[tail++] = (unsigned) child;
                    a.high = (uint16_t) (a.low + (uint16_t) ((range * sum0) / sum1 - 1));
                    ac_rescale(&a);
                    sum0 = (uint16_t) (sum0 + 1);
                }
                sum1 = (uint16_t) (sum1 + 1);
                if (sum1 > 50) {
                    sum0 >>= 1; sum1 >>= 1;
                    if (!sum0) sum0 = 1;
                    if (sum0 >= sum1) sum1 = sum0 + 1;
                }
            }
        }
    }
    ac_flush(&a);
    free(queue);
    *nused = used;
    return 1;
}

/* encode_nd_coefficients :179-242: the weights of every edge on a (state,label) that has a tree
 * child too, DC format, one adaptive context (scale 50) */
static int write_nd(const fa_wfa *wfa, const fa_info *wi, fa_bitw *out)
{
    unsigned total = 0, n = 0, state, label, e;
    unsigned *coeff, c_symbols = 1u << (wi->dc_rpf.mantissa_bits + 1);
    if (!write_nd_tree(wfa, wi, out, &total)) return 0;
    if (!total) return 1;
    coeff = (unsigned *) calloc(total, sizeof(unsigned));
    if (!coeff) { fa_set_error("Out of memory!"); return 0; }
    for (state = wfa->basis_states; state < wfa->states; state++)
        for (label = 0; label < 2; label++)
            if (FA_TREE(wfa, state, label) != FA_RANGE && FA_INTO(wfa, state, label, 0) != FA_NO_EDGE)
                for (e = 0; FA_INTO(wfa, state, label, e) != FA_NO_EDGE; e++) {
                    if (n >= total) {
                        fa_set_error("Can't write more than %d coefficients.", (int) total);
                        free(coeff);
                        return 0;
                    }
                    coeff[n++] = (unsigned) fa_rtob(FA_WEIGHT(wfa, state, label, e), &wi->dc_rpf);
                }
    encode_array(out, coeff, NULL, &c_symbols, 1, total, 50);
    free(coeff);
    return 1;
}

/* ------------------------------------------------------------ motion compensation (output/mc.c) */

/* MPEG's Huffman code of a vector component: {code, length}, index = component + search range
 * (codec/mwfa.c:40-50) */
static const unsigned short mv_code[33][2] = {
    {0x19, 11}, {0x1b, 11}, {0x1d, 11}, {0x1f, 11}, {0x21, 11}, {0x23, 11}, {0x13, 10}, {0x15, 10},
    {0x17, 10}, {0x7, 8}, {0x9, 8}, {0xb, 8}, {0x7, 7}, {0x3, 5}, {0x3, 4}, {0x3, 3}, {0x1, 1},
    {0x2, 3}, {0x2, 4}, {0x2, 5}, {0x6, 7}, {0xa, 8}, {0x8, 8}, {0x6, 8}, {0x16, 10}, {0x14, 10},
    {0x12, 10}, {0x22, 11}, {0x20, 11}, {0x1e, 11}, {0x1c, 11}, {0x1a, 11}, {0x18, 11} };

static void put_mv(fa_bitw *out, int v, unsigned sr)
{
    fa_bw_put_bits(out, mv_code[v + (int) sr][0], mv_code[v + (int) sr][1]);
}

static int write_mc(int frame_type, const fa_wfa *wfa, const fa_info *wi, fa_bitw *out)
{
    /* NONE, FORWARD, BACKWARD, INTERPOLATED: P frames 1 / 0, B frames 1 / 000 / 001 / 01 (:36-53) */
    static const unsigned char pcode[4][2] = { {1, 1}, {0, 1}, {0, 0}, {0, 0} };
    static const unsigned char bcode[4][2] = { {1, 1}, {0, 3}, {1, 3}, {1, 2} };
    const unsigned char (*code)[2] = frame_type == FA_P_FRAME ? pcode : bcode;
    const unsigned max_state = wi->color ? (unsigned) FA_TREE(wfa, (unsigned) FA_TREE(wfa, wfa->root_state, 0), 0)
                                         : wfa->states;
    unsigned *queue = (unsigned *) malloc(sizeof(unsigned) * (wfa->states + 1));
    unsigned last = 0, cur, state, label;
    if (!queue) { fa_set_error("Out of memory!"); return 0; }
    /* motion tree, breadth first from the states of level p_max_level (:76-146) */
    for (state = wfa->basis_states; state < max_state; state++)
        if ((int) wfa->level_of_state[state] - 1 == (int) wi->p_max_level) queue[last++] = state;
    for (cur = 0; cur < last; cur++)
        for (label = 0; label < 2; label++) {
            const unsigned lv = (unsigned) wfa->level_of_state[queue[cur]] - 1;
            unsigned type;
            state = queue[cur];
            type = (unsigned) wfa->mv[state * 2 + label].type;
            if (wfa->x[state * 2 + label] + fa_width_of_level(lv) <= wi->width
                && wfa->y[state * 2 + label] + fa_height_of_level(lv) <= wi->height)
                fa_bw_put_bits(out, code[type][0], code[type][1]);
            if (type == FA_MV_NONE && FA_TREE(wfa, state, label) != FA_RANGE && lv >= wi->p_min_level)
                queue[last++] = (unsigned) FA_TREE(wfa, state, label);
        }
    fa_bw_align(out);
    free(queue);
    /* vector components in state order (:148-251) */
    for (state = wfa->basis_states; state < max_state; state++)
        for (label = 0; label < 2; label++) {
            const fa_mv *mv = &wfa->mv[state * 2 + label];
            if (mv->type == FA_MV_FORWARD || mv->type == FA_MV_INTERPOLATED) {
                put_mv(out, mv->fx, wi->search_range); put_mv(out, mv->fy, wi->search_range);
            }
            if (mv->type == FA_MV_BACKWARD || mv->type == FA_MV_INTERPOLATED) {
                put_mv(out, mv->bx, wi->search_range); put_mv(out, mv->by, wi->search_range);
            }
        }
    fa_bw_align(out);
    return 1;
}

int fa_write_frame(const fa_wfa *wfa_in, const fa_info *wi, int frame_type, unsigned number,
                   int prediction, int normal_domains, int delta_domains, fa_bitw *out)
{
    fa_wfa *wfa = (fa_wfa *) wfa_in;     /* delta_state is (re)derived in place */
    unsigned edges, root_state;

    locate_delta_images(wfa);
    if (number == 0) fa_write_header(wi, out);
    fa_bw_rice(out, wfa->states, 8);
    fa_bw_rice(out, (unsigned) frame_type, 8);
    fa_bw_rice(out, number, 8);
    fa_bw_align(out);
    fa_bw_put_bit(out, 0);               /* tiling flag: always 0 (SURVEY finding 1) */
    fa_bw_align(out);
    if (!write_tree(wfa, out)) return 0;
    fa_bw_put_bit(out, prediction ? 1u : 0u);
    if (prediction && !write_nd(wfa, wi, out)) return 0;
    if (frame_type != FA_I_FRAME && !write_mc(frame_type, wfa, wi, out)) return 0;
    root_state = wi->color
        ? (unsigned) FA_TREE(wfa, (unsigned) FA_TREE(wfa, wfa->root_state, 0), 0)
        : wfa->root_state;
    edges  = column0_encoding(wfa, root_state, out);
    edges += delta_encoding(normal_domains, delta_domains, wfa, root_state, out);
    if (wi->color) edges += chroma_encoding_n(wfa, wi->chroma_max_states, out);
    if (edges && !write_weights(edges, wfa, wi, out)) return 0;
    return 1;
}
