/*
 *  fa_wfa.c -- host automaton container, initial basis, edge-hit statistics.
 *
 *  Container semantics follow reference codec/wfalib.c:45-116 (alloc), :233-275
 *  (append_edge keeps edges sorted by target), :277-312 (remove_states) and :182-231
 *  (compute_hits).  Basis: compiled-in "small.fco"/"small.wfa" (input/basis.c:51-139) or an
 *  ASCII basis file (input/read.c:219-340).
 */
#include <stdlib.h>
#include <string.h>
#include "fa_host.h"

fa_wfa *fa_wfa_alloc(unsigned cap)
{
    fa_wfa *w = (fa_wfa *) calloc(1, sizeof *w);
    unsigned s;
    if (!w) return NULL;
    w->cap = cap;
    w->final_distribution = (float *)   calloc(cap, sizeof(float));
    w->level_of_state     = (uint8_t *) calloc(cap, 1);
    w->domain_type        = (uint8_t *) calloc(cap, 1);
    w->delta_state        = (uint8_t *) calloc(cap, 1);
    w->tree       = (int16_t *)  calloc((size_t) cap * 2, sizeof(int16_t));
    w->x          = (uint16_t *) calloc((size_t) cap * 2, sizeof(uint16_t));
    w->y          = (uint16_t *) calloc((size_t) cap * 2, sizeof(uint16_t));
    w->into       = (int16_t *)  calloc((size_t) cap * 12, sizeof(int16_t));
    w->weight     = (float *)    calloc((size_t) cap * 12, sizeof(float));
    w->y_state    = (int16_t *)  calloc((size_t) cap * 2, sizeof(int16_t));
    w->y_column   = (uint8_t *)  calloc((size_t) cap * 2, 1);
    w->prediction = (uint8_t *)  calloc((size_t) cap * 2, 1);
    w->mv         = (fa_mv *)    calloc((size_t) cap * 2, sizeof(fa_mv));
    if (!w->mv || !w->final_distribution || !w->level_of_state || !w->domain_type || !w->delta_state
        || !w->tree || !w->x || !w->y || !w->into || !w->weight || !w->y_state
        || !w->y_column || !w->prediction) {
        fa_wfa_free(w);
        return NULL;
    }
    for (s = 0; s < cap; s++) {
        unsigned l;
        for (l = 0; l < 2; l++) {
            FA_INTO(w, s, l, 0) = FA_NO_EDGE;
            FA_TREE(w, s, l)    = FA_RANGE;
            w->y_state[s * 2 + l] = FA_RANGE;
        }
    }
    return w;
}

void fa_wfa_free(fa_wfa *w)
{
    if (!w) return;
    free(w->final_distribution); free(w->level_of_state); free(w->domain_type);
    free(w->delta_state); free(w->tree); free(w->x); free(w->y); free(w->into);
    free(w->weight); free(w->y_state); free(w->y_column); free(w->prediction); free(w->mv);
    free(w);
}

void fa_wfa_remove_states(fa_wfa *w, unsigned from)
{
    unsigned s, l;
    for (s = from; s < w->states; s++) {
        for (l = 0; l < 2; l++) {
            FA_INTO(w, s, l, 0)      = FA_NO_EDGE;
            FA_TREE(w, s, l)         = FA_RANGE;
            w->prediction[s * 2 + l] = 0;
            w->y_state[s * 2 + l]    = FA_RANGE;
            memset(&w->mv[s * 2 + l], 0, sizeof(fa_mv));      /* type NONE, zero vectors */
        }
        w->domain_type[s] = 0;
        w->delta_state[s] = 0;
    }
    w->states = from;
}

int fa_wfa_append_edge(fa_wfa *w, unsigned from, unsigned into, float weight, unsigned label)
{
    int pos = 0, last, e;
    while (FA_INTO(w, from, label, pos) != FA_NO_EDGE && FA_INTO(w, from, label, pos) < (int) into)
        pos++;
    for (last = pos; FA_INTO(w, from, label, last) != FA_NO_EDGE; last++)
        ;
    /* The reference does not check MAXEDGES (codec/wfalib.c:253-273): rows lie back to back in one block,
     * into[state][label][MAXEDGES + 1] (codec/wfa.h:131-133, codec/wfalib.c:70-75), so the sixth and later
     * edges of a label run on into the row of the next label (or state), whose own edges are then sorted in
     * among them.  The bases medium.fco / large.fco rely on it (up to 8 edges per label in the file, lists of
     * up to 33 entries in memory); every reader walks a list up to the first NO_EDGE, so the result is well
     * defined.  Only the end of the block is a limit. */
    if ((from * 2 + label) * 6 + (unsigned) last + 2 > w->cap * 12) return 0;
    for (e = last + 1; e > pos; e--) {          /* shift tail incl. the terminator */
        FA_INTO(w, from, label, e)   = FA_INTO(w, from, label, e - 1);
        FA_WEIGHT(w, from, label, e) = FA_WEIGHT(w, from, label, e - 1);
    }
    FA_INTO(w, from, label, pos)   = (int16_t) into;
    FA_WEIGHT(w, from, label, pos) = weight;
    return 1;
}

/* ------------------------------------------------------------------ basis */

static void basis_state0(fa_wfa *w)
{
    w->domain_type[0]        = FA_USE_DOMAIN;
    w->final_distribution[0] = 128;
    fa_wfa_append_edge(w, 0, 0, 1.0f, 0);
    fa_wfa_append_edge(w, 0, 0, 1.0f, 1);
}

static int builtin_small(fa_wfa *w)
{
    /* states {1, x, y}: values of input/basis.c:126-131 */
    static const float trans[][4] = { {1, 2, 0.5f, 0}, {1, 2, 0.5f, 1}, {1, 0, 0.5f, 1},
                                      {2, 1, 1.0f, 0}, {2, 1, 1.0f, 1} };
    unsigned i;
    w->basis_states = w->states = 3;
    basis_state0(w);
    for (i = 1; i < 3; i++) {
        w->final_distribution[i] = 64;
        w->domain_type[i]        = FA_USE_DOMAIN;
    }
    for (i = 0; i < sizeof trans / sizeof trans[0]; i++)
        fa_wfa_append_edge(w, (unsigned) trans[i][0], (unsigned) trans[i][1], trans[i][2],
                           (unsigned) trans[i][3]);
    return 1;
}

/* token reader for the ASCII basis: skips whitespace and '#' comments */
static int next_token(FILE *f, char *tok, size_t n)
{
    int c;
    size_t i = 0;
    for (;;) {
        while ((c = getc(f)) != EOF && (c == ' ' || c == '\t' || c == '\n' || c == '\r'))
            ;
        if (c == EOF) return 0;
        if (c != '#') break;
        while ((c = getc(f)) != EOF && c != '\n')
            ;
        if (c == EOF) return 0;
    }
    do {
        if (i + 1 < n) tok[i++] = (char) c;
        c = getc(f);
    } while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r');
    tok[i] = 0;
    return 1;
}

static int ascii_basis(const char *name, fa_wfa *w)
{
    FILE *f = open_file(name, "FIASCO_DATA", READ_ACCESS);
    char tok[64];
    unsigned s, n;
    long nl;
    if (!f) { fa_set_error("File `%s': I/O Error - %s.", name, "No such file or directory"); return 0; }
#define NEED_TOKEN() do { if (!next_token(f, tok, sizeof tok)) goto bad; } while (0)
    NEED_TOKEN();
    if (strcmp(tok, "Fiasco") != 0) {
        fclose(f);
        fa_set_error("Input file %s is not an ASCII FIASCO initial basis!", name);
        return 0;
    }
    NEED_TOKEN();
    nl = strtol(tok, NULL, 10);                 /* states besides state 0 */
    if (nl <= 0 || nl + 1 >= (long) w->cap) goto bad;
    n = (unsigned) nl;
    w->basis_states = w->states = n + 1;
    basis_state0(w);
    for (s = 1; s <= n; s++) { NEED_TOKEN(); w->domain_type[s] = atoi(tok) ? FA_USE_DOMAIN : FA_AUXILIARY; }
    for (s = 1; s <= n; s++) { NEED_TOKEN(); w->final_distribution[s] = strtof(tok, NULL); }
    for (s = 1; s <= n; s++) {
        NEED_TOKEN();
        if (atoi(tok) != (int) s) goto bad;
        for (;;) {
            int label, dom;
            float wt;
            NEED_TOKEN(); label = atoi(tok);
            if (label == -1) break;
            NEED_TOKEN(); dom = atoi(tok);
            NEED_TOKEN(); wt = strtof(tok, NULL);
            if (label < 0 || label > 1 || dom < 0 || dom > (int) n) goto bad;
            if (!fa_wfa_append_edge(w, s, (unsigned) dom, wt, (unsigned) label)) goto bad;
        }
    }
    fclose(f);
    return 1;
bad:
    fclose(f);
    fa_set_error("Format error: ASCII FIASCO initial basis file %s", name);
    return 0;
#undef NEED_TOKEN
}

int fa_load_basis(const char *name, fa_wfa *w)
{
    if (strcmp(name, "small.fco") == 0 || strcmp(name, "small.wfa") == 0)
        return builtin_small(w);
    fa_warning("WFA initial basis '%s' isn't linked with the excecutable yet."
               "\nLoading basis from disk instead.", name);
    return ascii_basis(name, w);
}

/* ------------------------------------------------------------------ hits */

typedef struct hit { int16_t key, value; } hit;

/* glibc's qsort is a stable merge sort for small arrays, so equal counts keep ascending
 * state order in the reference; made explicit here. */
static int cmp_hit(const void *a, const void *b)
{
    const hit *x = (const hit *) a, *y = (const hit *) b;
    if (x->key != y->key) return x->key > y->key ? -1 : 1;
    return x->value < y->value ? -1 : (x->value > y->value);
}

static int cmp_i16(const void *a, const void *b)
{
    int16_t x = *(const int16_t *) a, y = *(const int16_t *) b;
    return x < y ? -1 : x > y;
}

int16_t *fa_compute_hits(unsigned from, unsigned to, unsigned n, const fa_wfa *wfa)
{
    hit *hits = (hit *) calloc(to ? to : 1, sizeof *hits);
    int16_t *domains;
    unsigned s, l, e, d;
    if (!hits) return NULL;
    for (d = 0; d < to; d++) { hits[d].value = (int16_t) d; hits[d].key = 0; }
    for (s = from; s <= to; s++)
        for (l = 0; l < 2; l++)
            for (e = 0; FA_INTO(wfa, s, l, e) != FA_NO_EDGE; e++)
                hits[FA_INTO(wfa, s, l, e)].key++;
    if (to > 1) qsort(hits + 1, to - 1, sizeof *hits, cmp_hit);
    if (n > to) n = to;
    domains = (int16_t *) calloc(n + 1, sizeof *domains);
    if (!domains) { free(hits); return NULL; }
    for (d = 0; d < n && (!d || hits[d].key); d++)
        domains[d] = hits[d].value;
    n = d;
    qsort(domains, n, sizeof *domains, cmp_i16);
    domains[n] = -1;
    free(hits);
    return domains;
}
