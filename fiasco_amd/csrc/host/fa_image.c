/*
 *  fa_image.c -- raw PGM/PPM input into 12.4 fixed-point int16 planes.
 *
 *  Follows reference lib/image.c:283-391 (read_pnmheader, read_image):
 *    gray  : (g - 128) * 16
 *    colour: Y/Cb/Cr from double-precision coefficients, * 16, truncated to int16
 *  and lib/image.c:194-195 (width and height must be even), :316-323 (>= 32).
 *  maxval is read and ignored exactly as the reference does.
 */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "fa_host.h"

typedef struct cursor { const unsigned char *p, *end; } cursor;

/* whitespace and '#'-comments, reference lib/misc.c remove_comments() */
static int skip_ws_comments(cursor *c)
{
    for (;;) {
        while (c->p < c->end && isspace(*c->p)) c->p++;
        if (c->p >= c->end) return 0;
        if (*c->p != '#') return 1;
        while (c->p < c->end && *c->p != '\n') c->p++;
        if (c->p >= c->end) return 0;
    }
}

static int read_int(cursor *c, int *out)
{
    long v = 0;
    int neg = 0, digits = 0;
    if (!skip_ws_comments(c)) return 0;
    if (c->p < c->end && (*c->p == '-' || *c->p == '+')) { neg = *c->p == '-'; c->p++; }
    while (c->p < c->end && isdigit(*c->p)) {
        v = v * 10 + (*c->p - '0');
        if (v > 0x7fffffffL) v = 0x7fffffffL;
        c->p++; digits++;
    }
    if (!digits) return 0;
    *out = (int) (neg ? -v : v);
    return 1;
}

int fa_pnm_header(const unsigned char *buf, size_t len, const char *name,
                  unsigned *w, unsigned *h, int *color, size_t *data_off)
{
    cursor c;
    int v;
    if (!name) name = "stdin";
    if (len < 2) { fa_set_error("%s: EOF reached, input seems to be truncated!", name); return 0; }
    if (buf[0] == 'P' && buf[1] == '5') *color = 0;
    else if (buf[0] == 'P' && buf[1] == '6') *color = 1;
    else {
        fa_set_error("%s: image format '%c%c' not supported.", name, buf[0], buf[1]);
        return 0;
    }
    c.p = buf + 2; c.end = buf + len;
    if (!read_int(&c, &v)) { fa_set_error("Can't read integer value!"); return 0; }
    if (v < 32) { fa_set_error("Width of image `%s' has to be at least 32 pixels.", name); return 0; }
    *w = (unsigned) v;
    if (!read_int(&c, &v)) { fa_set_error("Can't read integer value!"); return 0; }
    if (v < 32) { fa_set_error("Height of image `%s' has to be at least 32 pixels.", name); return 0; }
    *h = (unsigned) v;
    if (!read_int(&c, &v)) { fa_set_error("Can't read integer value!"); return 0; }
    if (c.p >= c.end) { fa_set_error("%s: EOF reached, input seems to be truncated!", name); return 0; }
    c.p++;                                  /* the single separator byte */
    *data_off = (size_t) (c.p - buf);
    return 1;
}

void fa_image_free(fa_image *im)
{
    int b;
    if (!im) return;
    for (b = 0; b < 3; b++) free(im->pixels[b]);
    free(im);
}

fa_image *fa_image_from_pnm(const unsigned char *buf, size_t len, const char *name)
{
    unsigned w, h, n, i;
    int color, b;
    size_t off;
    fa_image *im;
    const unsigned char *px;

    if (!fa_pnm_header(buf, len, name, &w, &h, &color, &off)) return NULL;
    if ((w & 1) || (h & 1)) {
        fa_set_error("Width and height of images must be even numbers.");
        return NULL;
    }
    n = w * h;
    if (len - off < (size_t) n * (color ? 3 : 1)) {
        fa_set_error("File `%s': I/O Error - %s.", name ? name : "stdin", "truncated pixel data");
        return NULL;
    }
    im = (fa_image *) calloc(1, sizeof *im);
    if (!im) { fa_set_error("Out of memory!"); return NULL; }
    im->width = w; im->height = h; im->color = color;
    for (b = 0; b < (color ? 3 : 1); b++) {
        im->pixels[b] = (int16_t *) malloc((size_t) n * sizeof(int16_t));
        if (!im->pixels[b]) { fa_image_free(im); fa_set_error("Out of memory!"); return NULL; }
    }
    px = buf + off;
    if (!color) {
        for (i = 0; i < n; i++)
            im->pixels[0][i] = (int16_t) (((int) px[i] - 128) * 16);
    } else {
        for (i = 0; i < n; i++) {
            int r = px[3 * i], g = px[3 * i + 1], bl = px[3 * i + 2];
            /* double arithmetic, left to right, then C truncation toward zero */
            im->pixels[FA_Y][i]  = (int16_t) ((+0.2989 * r + 0.5866 * g + 0.1145 * bl - 128) * 16);
            im->pixels[FA_CB][i] = (int16_t) ((-0.1687 * r - 0.3312 * g + 0.5000 * bl) * 16);
            im->pixels[FA_CR][i] = (int16_t) ((+0.5000 * r - 0.4183 * g - 0.0816 * bl) * 16);
        }
    }
    return im;
}
