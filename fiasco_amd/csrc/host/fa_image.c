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

/* zeroed planes of the coder's 12.4 fixed point format */
fa_image *fa_image_alloc(unsigned width, unsigned height, int color)
{
    fa_image *im = (fa_image *) calloc(1, sizeof *im);
    int b;
    if (!im) return NULL;
    im->width = width; im->height = height; im->color = color;
    for (b = 0; b < (color ? 3 : 1); b++) {
        im->pixels[b] = (int16_t *) calloc((size_t) width * height, sizeof(int16_t));
        if (!im->pixels[b]) { fa_image_free(im); return NULL; }
    }
    return im;
}

void fa_image_free(fa_image *im)
{
    int b;
    if (!im) return;
    if (im->dev) fa_core_release_dev(im->dev, im->dev_id);
    if (!im->borrowed)
        for (b = 0; b < 3; b++) free(im->pixels[b]);
    free(im);
}

/* the pixel conversion of read_image (lib/image.c:365-389) into caller-provided planes */
static void convert_planes(const unsigned char *px, size_t n, int color, int16_t *const plane[3])
{
    size_t i;
    if (!color) {
        int16_t *g = plane[0];
        for (i = 0; i < n; i++)
            g[i] = (int16_t) (((int) px[i] - 128) * 16);
    } else {
        for (i = 0; i < n; i++) {
            int r = px[3 * i], g = px[3 * i + 1], bl = px[3 * i + 2];
            /* double arithmetic, left to right, then C truncation toward zero */
            plane[FA_Y][i]  = (int16_t) ((+0.2989 * r + 0.5866 * g + 0.1145 * bl - 128) * 16);
            plane[FA_CB][i] = (int16_t) ((-0.1687 * r - 0.3312 * g + 0.5000 * bl) * 16);
            plane[FA_CR][i] = (int16_t) ((+0.5000 * r - 0.4183 * g - 0.0816 * bl) * 16);
        }
    }
}

/* header checks shared by both entry points; sizes are formed in size_t and the dimensions
 * are bounded before anything is allocated (2^13 = the largest side of a level-26 image) */
static int checked_header(const unsigned char *buf, size_t len, const char *name,
                          unsigned *w, unsigned *h, int *color, size_t *off, size_t *n)
{
    if (!fa_pnm_header(buf, len, name, w, h, color, off)) return 0;
    if ((*w & 1) || (*h & 1)) {
        fa_set_error("Width and height of images must be even numbers.");
        return 0;
    }
    if (*w > (1u << (FA_CAP_LEVEL / 2)) || *h > (1u << (FA_CAP_LEVEL / 2))) {
        fa_set_error("Image `%s' is too large (at most %u x %u pixels).", name ? name : "stdin",
                     1u << (FA_CAP_LEVEL / 2), 1u << (FA_CAP_LEVEL / 2));
        return 0;
    }
    *n = (size_t) *w * (size_t) *h;
    if (len - *off < *n * (*color ? 3 : 1)) {
        fa_set_error("File `%s': I/O Error - %s.", name ? name : "stdin", "truncated pixel data");
        return 0;
    }
    return 1;
}

fa_image *fa_image_from_pnm(const unsigned char *buf, size_t len, const char *name)
{
    unsigned w, h;
    int color, b;
    size_t off, n;
    fa_image *im;

    if (!checked_header(buf, len, name, &w, &h, &color, &off, &n)) return NULL;
    im = (fa_image *) calloc(1, sizeof *im);
    if (!im) { fa_set_error("Out of memory!"); return NULL; }
    im->width = w; im->height = h; im->color = color;
    for (b = 0; b < (color ? 3 : 1); b++) {
        im->pixels[b] = (int16_t *) malloc(n * sizeof(int16_t));
        if (!im->pixels[b]) { fa_image_free(im); fa_set_error("Out of memory!"); return NULL; }
    }
    convert_planes(buf + off, n, color, im->pixels);
    return im;
}

/* the same into planes the caller owns (band planes back to back at `planes`): used when the
 * frames of a batch are replaced while the previous pass still runs (fiasco_amd_batch_upload) */
fa_image *fa_image_from_pnm_into(const unsigned char *buf, size_t len, const char *name,
                                 unsigned want_w, unsigned want_h, int want_color, int16_t *planes)
{
    unsigned w, h;
    int color, b;
    size_t off, n;
    fa_image *im;

    if (!checked_header(buf, len, name, &w, &h, &color, &off, &n)) return NULL;
    if (w != want_w || h != want_h || color != want_color) {
        fa_set_error("`%s': replacement frames must keep the size and colour model of the batch.",
                     name ? name : "stdin");
        return NULL;
    }
    im = (fa_image *) calloc(1, sizeof *im);
    if (!im) { fa_set_error("Out of memory!"); return NULL; }
    im->width = w; im->height = h; im->color = color; im->borrowed = 1;
    for (b = 0; b < (color ? 3 : 1); b++) im->pixels[b] = planes + (size_t) b * n;
    convert_planes(buf + off, n, color, im->pixels);
    return im;
}
