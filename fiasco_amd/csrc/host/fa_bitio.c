/*
 *  fa_bitio.c -- MSB-first bit writer, memory backed.
 *
 *  Bit-exact with reference lib/bit-io.c:148-329 (write side): the write cursor starts
 *  with bitpos = 8 (so an align before the first bit emits a whole zero byte), align pads
 *  with zero bits while bitpos != 0, and close_bitfile() writes every byte touched, at
 *  least one.  Rice / adjusted-binary codes follow lib/misc.c:186-244.
 */
#include <stdlib.h>
#include <string.h>
#include "fa_host.h"

void fa_bw_init(fa_bitw *b)
{
    memset(b, 0, sizeof *b);
    b->cap = 1u << 16;
    b->buf = (unsigned char *) calloc(b->cap, 1);
    b->bytes = 1;          /* ptr sits on byte 0 */
    b->bitpos = 8;
}

void fa_bw_free(fa_bitw *b) { free(b->buf); b->buf = NULL; }

void fa_bw_put_bit(fa_bitw *b, unsigned v)
{
    if (b->bitpos == 0) {                      /* advance to the next byte */
        if (b->bytes == b->cap) {
            size_t nc = b->cap * 2;
            unsigned char *nb = (unsigned char *) realloc(b->buf, nc);
            if (!nb) abort();
            memset(nb + b->cap, 0, nc - b->cap);
            b->buf = nb; b->cap = nc;
        }
        b->bytes++;
        b->bitpos = 8;
    }
    b->bitpos--;
    if (v) b->buf[b->bytes - 1] |= (unsigned char) (1u << b->bitpos);
    b->nbits++;
}

void fa_bw_put_bits(fa_bitw *b, unsigned v, unsigned n)
{
    while (n--) fa_bw_put_bit(b, (v >> n) & 1u);
}

void fa_bw_align(fa_bitw *b)
{
    while (b->bitpos) fa_bw_put_bit(b, 0);
}

size_t fa_bw_finish(fa_bitw *b) { return b->bytes; }

/* What `src` holds behind a byte-aligned `b` (bitpos 0, or nothing written yet): the cursor ends where the
 * writes into src would have left it had they gone to b. */
void fa_bw_append(fa_bitw *b, const fa_bitw *src)
{
    const int fresh = b->nbits == 0;                 /* byte 0 is touched but empty */
    const size_t at = fresh ? 0 : b->bytes, need = at + src->bytes;
    if (src->nbits == 0) return;
    if (need > b->cap) {
        size_t nc = b->cap;
        unsigned char *nb;
        while (nc < need) nc *= 2;
        nb = (unsigned char *) realloc(b->buf, nc);
        if (!nb) abort();
        memset(nb + b->cap, 0, nc - b->cap);
        b->buf = nb; b->cap = nc;
    }
    memcpy(b->buf + at, src->buf, src->bytes);
    b->bytes = need; b->bitpos = src->bitpos; b->nbits += src->nbits;
}

void fa_bw_rice(fa_bitw *b, unsigned value, unsigned k)
{
    unsigned u;
    for (u = value >> k; u; u--) fa_bw_put_bit(b, 1);
    fa_bw_put_bit(b, 0);
    fa_bw_put_bits(b, value & ((1u << k) - 1), k);
}

static unsigned ilog2(unsigned v) { unsigned k = 0; while (v >>= 1) k++; return k; }

void fa_bw_bincode(fa_bitw *b, unsigned value, unsigned maxval)
{
    unsigned k = ilog2(maxval + 1);
    unsigned r = (maxval + 1) % (1u << k);
    if (value < maxval + 1 - 2 * r) fa_bw_put_bits(b, value, k);
    else fa_bw_put_bits(b, value + maxval + 1 - 2 * r, k + 1);
}
