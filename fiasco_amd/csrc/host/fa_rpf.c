/*
 *  fa_rpf.c -- reduced precision format, host side (the stream writer quantises every
 *  stored weight once more, reference output/weights.c:137-165).
 *
 *  rtob follows reference lib/rpf.c:59-112: scale by 1/range, take the IEEE-754 single
 *  fields, build a 23-bit fixed point mantissa with the hidden one, shift by the
 *  exponent, keep mantissa_bits+1 bits, round half up.  The reference shifts a 32-bit
 *  unsigned by -exponent with counts up to 126; x86 masks the count to 5 bits and the
 *  oracle build depends on that, so the mask is written out.
 */
#include <string.h>
#include "fa_host.h"

void fa_rpf_init(fa_rpf *r, unsigned mantissa, int range_e)
{
    /* lib/rpf.c:186-199: out-of-interval mantissas BOTH fall back to 2 */
    if (mantissa < 2) {
        fa_warning("Size of RPF mantissa has to be in the interval [2,8]. Using minimum value 2.\n");
        mantissa = 2;
    } else if (mantissa > 8) {
        fa_warning("Size of RPF mantissa has to be in the interval [2,8]. Using maximum value 8.\n");
        mantissa = 2;
    }
    r->mantissa_bits = mantissa;
    r->range_e = range_e;
    switch (range_e) {
    case FIASCO_RPF_RANGE_0_75: r->range = 0.75f; break;
    case FIASCO_RPF_RANGE_1_50: r->range = 1.50f; break;
    case FIASCO_RPF_RANGE_2_00: r->range = 2.00f; break;
    case FIASCO_RPF_RANGE_1_00: r->range = 1.00f; break;
    default:
        fa_warning("Invalid RPF range specified. Using default value 1.0.");
        r->range = 1.00f; r->range_e = FIASCO_RPF_RANGE_1_00;
        break;
    }
}

int fa_rtob(float f, const fa_rpf *r)
{
    uint32_t bits, mant;
    int expo, sign;
    f /= r->range;
    memcpy(&bits, &f, 4);
    mant = bits & 0x7fffffu;
    expo = (int) ((bits >> 23) & 0xffu) - 126;
    sign = (int) (bits >> 31);
    mant = (mant >> 1) | (1u << 22);
    if (expo > 0) mant <<= ((unsigned) expo & 31u);
    else          mant >>= ((unsigned) (-expo) & 31u);
    mant >>= (23 - r->mantissa_bits - 1);
    mant += 1;
    mant >>= 1;
    if (mant == 0) return -1;                               /* RPF_ZERO */
    if (mant >= (1u << r->mantissa_bits)) return sign;      /* overflow: +-1.0 */
    return (int) (((mant & ((1u << r->mantissa_bits) - 1)) << 1) | (unsigned) sign);
}
