/*
 *  oracle_decoder.c -- the frame an automaton describes, in the encoder's 12.4 fixed point planes.
 *
 *  TEST INFRASTRUCTURE (oracle library only): the CPU restatement of the reference's decoder.  The product
 *  decodes on the device (fiasco_amd/csrc/hip/frame_decoder.inc) and does not contain this file.
 *
 *  The coder needs it after every frame of a video: the next P/B frame is predicted from the
 *  RECONSTRUCTED frame, not from the original (reference codec/coder.c:647-651).  Decoded
 *  pixels must therefore be the reference decoder's to the last bit; its arithmetic is integer
 *  only (codec/decoder.c:1106-1498):
 *    level 0      pixel = (int) (final_distribution * 8 + .5) * 2
 *    level l      every half of a state image = image of the tree child (copy) + for each edge
 *                   domain 0 : the constant (int) (weight * final_distribution[0] * 8 + .5) * 2
 *                   domain d : ((int_weight * pixel of d at level l-1) >> 10) << 1,
 *                              int_weight = (short) (weight * 512 + 0.5)  (codec/wfalib.c:273)
 *                 summed in 16 bits (the reference adds pixel pairs in 32-bit words and masks
 *                 the carry out of the lower one: wrap-around per pixel)
 *  The reference walks the levels bottom-up over explicitly allocated buffers
 *  (alloc_state_images :877-1015); the same values are produced here by a memoised recursion
 *  over (state, level).  Frame assembly and cropping: decode_image, codec/decoder.c:411-536.
 *  Motion compensation: restore_mc / extract_mc_block, codec/motion.c:37-334 (full-pixel
 *  vectors; see fa_extract_mc_block for half-pixel).
 */
#include <stdlib.h>
#include <string.h>
#include "fa_host.h"

typedef struct dec {
    const fa_wfa *w;
    unsigned      states, nlev;
    int16_t     **img;              /* [nlev][states], NULL = not computed yet */
    int           oom;
} dec;

static int16_t fixed_of(float v)
{
    return (int16_t) ((int) ((double) (v * 8) + .5) * 2);
}

static const int16_t *state_image(dec *d, unsigned state, unsigned level)
{
    const fa_wfa *w = d->w;
    int16_t **slot = &d->img[(size_t) level * d->states + state];
    int16_t *buf;
    unsigned label;

    if (*slot) return *slot;
    buf = (int16_t *) calloc(fa_size_of_level(level), sizeof(int16_t));
    if (!buf) { d->oom = 1; return NULL; }
    *slot = buf;
    if (level == 0) {
        buf[0] = fixed_of(w->final_distribution[state]);
        return buf;
    }
    {
        const unsigned cw = fa_width_of_level(level - 1), ch = fa_height_of_level(level - 1);
        const unsigned pw = fa_width_of_level(level);     /* row pitch of this image */
        for (label = 0; label < 2; label++) {
            /* odd level: the halves lie above each other, even level: side by side */
            int16_t *dst = buf + ((level & 1) ? label * ch * pw : label * cw);
            unsigned e, x, y;
            int dom;
            if (FA_TREE(w, state, label) != FA_RANGE) {
                const int16_t *src = state_image(d, (unsigned) FA_TREE(w, state, label), level - 1);
                if (!src) return NULL;
                for (y = 0; y < ch; y++) memcpy(dst + y * pw, src + y * cw, cw * sizeof(int16_t));
            }
            for (e = 0; (dom = FA_INTO(w, state, label, e)) != FA_NO_EDGE; e++) {
                if (dom == 0) {
                    const int16_t c = (int16_t) ((int) ((double) (FA_WEIGHT(w, state, label, e)
                                                                  * w->final_distribution[0] * 8) + .5) * 2);
                    for (y = 0; y < ch; y++)
                        for (x = 0; x < cw; x++)
                            dst[y * pw + x] = (int16_t) (uint16_t) ((unsigned) dst[y * pw + x] + (unsigned) c);
                } else {
                    const int iw = (int16_t) ((double) (FA_WEIGHT(w, state, label, e) * 512) + 0.5);
                    const int16_t *src = state_image(d, (unsigned) dom, level - 1);
                    if (!src) return NULL;
                    for (y = 0; y < ch; y++)
                        for (x = 0; x < cw; x++) {
                            const int t = ((iw * (int) src[y * cw + x]) >> 10) * 2;
                            dst[y * pw + x] = (int16_t) (uint16_t) ((unsigned) dst[y * pw + x] + (unsigned) t);
                        }
                }
            }
        }
    }
    return buf;
}

/* decode_image (codec/decoder.c:411-536) for the 4:4:4 format the coder asks for */
fa_image *fa_decode_image(unsigned orig_width, unsigned orig_height, const fa_wfa *w, int color)
{
    unsigned root[3] = { 0, 0, 0 }, max_level = 0, state, width = 0, height = 0, l, b;
    fa_image *frame;
    dec d;

    if (color) {
        unsigned r0 = (unsigned) FA_TREE(w, w->root_state, 0), r1 = (unsigned) FA_TREE(w, w->root_state, 1);
        root[FA_Y] = (unsigned) FA_TREE(w, r0, 0); root[FA_CB] = (unsigned) FA_TREE(w, r0, 1);
        root[FA_CR] = (unsigned) FA_TREE(w, r1, 0);
    } else
        root[0] = w->root_state;
    for (state = w->basis_states; state < w->states; state++)
        if (FA_INTO(w, state, 0, 0) != FA_NO_EDGE || FA_INTO(w, state, 1, 0) != FA_NO_EDGE) {
            /* largest level with a linear combination; actual frame size (:842-875) */
            unsigned lv = w->level_of_state[state];
            unsigned x = w->x[state * 2] + fa_width_of_level(lv), y = w->y[state * 2] + fa_height_of_level(lv);
            if (lv > max_level) max_level = lv;
            if (x > width) width = x;
            if (y > height) height = y;
        }
    width += width & 1; height += height & 1;
    if (width < orig_width) width = orig_width;
    if (height < orig_height) height = orig_height;
    frame = fa_image_alloc(width, height, color);
    if (!frame) { fa_set_error("Out of memory!"); return NULL; }

    memset(&d, 0, sizeof d);
    d.w = w; d.states = w->states; d.nlev = max_level + 1;
    d.img = (int16_t **) calloc((size_t) d.nlev * d.states, sizeof *d.img);
    if (!d.img) { fa_image_free(frame); fa_set_error("Out of memory!"); return NULL; }
    for (state = w->basis_states; state < w->states && !d.oom; state++) {
        /* the three states that join the bands of a colour frame lie above every band */
        if (color && (state == w->root_state || state == (unsigned) FA_TREE(w, w->root_state, 0)
                      || state == (unsigned) FA_TREE(w, w->root_state, 1)))
            continue;
        if (w->level_of_state[state] == max_level) {
            const unsigned band = !color || state <= root[FA_Y] ? 0 : state > root[FA_CB] ? FA_CR : FA_CB;
            const unsigned x0 = w->x[state * 2], y0 = w->y[state * 2];
            const unsigned bw = fa_width_of_level(max_level), bh = fa_height_of_level(max_level);
            const int16_t *src = state_image(&d, state, max_level);
            unsigned y;
            if (!src) break;
            for (y = 0; y < bh && y0 + y < height; y++) {
                unsigned n = x0 >= width ? 0 : (x0 + bw <= width ? bw : width - x0);
                memcpy(frame->pixels[band] + (size_t) (y0 + y) * width + x0, src + (size_t) y * bw,
                       n * sizeof(int16_t));
            }
        }
    }
    for (l = 0; l < d.nlev; l++)
        for (state = 0; state < d.states; state++) free(d.img[(size_t) l * d.states + state]);
    free(d.img);
    if (d.oom) { fa_image_free(frame); fa_set_error("Out of memory!"); return NULL; }

    if (orig_width != width || orig_height != height) {         /* crop (:502-530) */
        for (b = 0; b < (color ? 3u : 1u); b++) {
            int16_t *p = frame->pixels[b];
            unsigned y;
            if (orig_width != width)
                for (y = 0; y < orig_height; y++)
                    memmove(p + (size_t) y * orig_width, p + (size_t) y * width, orig_width * sizeof(int16_t));
        }
        frame->width = orig_width; frame->height = orig_height;
    }
    return frame;
}

/* extract_mc_block (codec/motion.c:231-261), full-pixel vectors.  The reference's half-pixel
 * branch divides the vector AFTER its conversion to unsigned (:271), which sends every
 * negative component far outside the frame -- it cannot be reproduced, only refused (the
 * option is unreachable from cfiasco, bin/cwfa.c never calls set_video_param). */
void fa_extract_mc_block(int16_t *mcblock, unsigned width, unsigned height, const int16_t *reference,
                         unsigned ref_width, unsigned xo, unsigned yo, int mx, int my)
{
    const int16_t *r = reference + (size_t) ((int) yo + my) * ref_width + (size_t) ((int) xo + mx);
    unsigned y;
    for (y = 0; y < height; y++, mcblock += width, r += ref_width)
        memcpy(mcblock, r, width * sizeof(int16_t));
}

/* restore_mc (codec/motion.c:37-229), enlarge factor 0, 4:4:4 */
int fa_restore_mc(fa_image *image, const fa_image *past, const fa_image *future, const fa_wfa *w,
                  unsigned p_max_level)
{
    const unsigned nb = image->color ? 3 : 1;
    unsigned state, label, root, b;
    int16_t *mc1 = (int16_t *) calloc(fa_size_of_level(p_max_level), sizeof(int16_t));
    int16_t *mc2 = (int16_t *) calloc(fa_size_of_level(p_max_level), sizeof(int16_t));
    if (!mc1 || !mc2) { free(mc1); free(mc2); fa_set_error("Out of memory!"); return 0; }
    root = image->color ? (unsigned) FA_TREE(w, (unsigned) FA_TREE(w, w->root_state, 0), 0) : w->root_state;
    for (state = w->basis_states; state <= root; state++)
        for (label = 0; label < 2; label++) {
            const fa_mv *mv = &w->mv[state * 2 + label];
            const unsigned level = (unsigned) w->level_of_state[state] - 1;
            const unsigned bw = fa_width_of_level(level), bh = fa_height_of_level(level);
            const unsigned x0 = w->x[state * 2 + label], y0 = w->y[state * 2 + label];
            if (mv->type == FA_MV_NONE || level > p_max_level) continue;
            for (b = 0; b < nb; b++) {
                int16_t *dst = image->pixels[b] + (size_t) y0 * image->width + x0;
                unsigned x, y;
                if (mv->type != FA_MV_BACKWARD)
                    fa_extract_mc_block(mc1, bw, bh, past->pixels[b], past->width, x0, y0, mv->fx, mv->fy);
                if (mv->type != FA_MV_FORWARD)
                    fa_extract_mc_block(mv->type == FA_MV_BACKWARD ? mc1 : mc2, bw, bh, future->pixels[b],
                                        future->width, x0, y0, mv->bx, mv->by);
                for (y = 0; y < bh; y++)
                    for (x = 0; x < bw; x++) {
                        const int add = mv->type == FA_MV_INTERPOLATED
                                        ? ((int) mc1[y * bw + x] + (int) mc2[y * bw + x]) >> 1
                                        : (int) mc1[y * bw + x];
                        dst[(size_t) y * image->width + x] =
                            (int16_t) (uint16_t) ((unsigned) dst[(size_t) y * image->width + x] + (unsigned) add);
                    }
            }
        }
    free(mc1); free(mc2);
    if (image->color) {                       /* chroma is clipped to [-128, 127] * 16 (:190-225) */
        for (b = 1; b < 3; b++) {
            int16_t *p = image->pixels[b];
            size_t n = (size_t) image->width * image->height, i;
            for (i = 0; i < n; i++) {
                int v = p[i] >> 4;
                v = v < -128 ? -128 : v > 127 ? 127 : v;
                p[i] = (int16_t) (v * 16);
            }
        }
    }
    return 1;
}

/* PSNR of a reconstructed plane against the original, both in 12.4 fixed point, as bin/pnmpsnr
 * measures it on 8-bit samples: pixel value = clamp(p / 16 + 128) */
double fa_plane_mse(const int16_t *a, const int16_t *b, size_t n)
{
    double sum = 0;
    size_t i;
    for (i = 0; i < n; i++) {
        int va = a[i] / 16 + 128, vb = b[i] / 16 + 128;
        va = va < 0 ? 0 : va > 255 ? 255 : va;
        vb = vb < 0 ? 0 : vb > 255 ? 255 : vb;
        sum += (double) (va - vb) * (va - vb);
    }
    return n ? sum / (double) n : 0;
}
