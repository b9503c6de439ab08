#!/bin/bash
# TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).  Evidence for DESIGN.md 7: why half-pixel motion
# vectors are refused by the device coder instead of reproduced.
#
# codec/motion.c:231-334 (extract_mc_block with half_pixel) converts the motion vector to `unsigned' before `/ 2'
# (`:234,271'): every NEGATIVE vector becomes a huge offset and the block is read far outside the reference frame --
# undefined behaviour whose result depends on what lies there.  This script compiles the reference's sources WHERE THEY
# LIE with gcc -fsanitize=address (same recipe as ref_build.sh, objects into oracle/_ref/obj_asan, git-ignored), links a
# 30-line driver that codes a 2-frame `IP' sequence with fiasco_c_options_set_video_param(.., half_pixel = 1, ..) --
# the only way to switch it on; `cfiasco' has no option for it -- and keeps AddressSanitizer's report:
#     oracle/_ref/halfpixel_asan.txt     (copied to profiles/r06_halfpixel_asan.txt for the record)
set -uo pipefail
REF=${FIASCO_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
[ -d "$REF/codec" ] || { echo "ref_halfpixel_asan: $REF not present" >&2; exit 0; }
[ -f "$OUT/cfg/config.h" ] || bash "$HERE/ref_build.sh" >/dev/null
mkdir -p "$OUT/obj_asan"
CFLAGS="-O1 -g -fsanitize=address -fno-omit-frame-pointer -fcommon -w -DHAVE_CONFIG_H -I$OUT/cfg -I$REF -I$REF/lib -I$REF/input -I$REF/output -I$REF/codec -DFIASCO_SHARE=\"$REF/data\""
objs=()
for f in "$REF"/lib/*.c "$REF"/input/*.c "$REF"/output/*.c "$REF"/codec/*.c; do
    o="$OUT/obj_asan/$(basename "$(dirname "$f")")_$(basename "${f%.c}").o"
    [ -f "$o" ] || gcc $CFLAGS -c "$f" -o "$o" || exit 1
    objs+=("$o")
done
DRV=$(mktemp /tmp/halfpixel_XXXXXX.c)
cat > "$DRV" <<'C'
#include <stdio.h>
#include "fiasco.h"
int main (int argc, char **argv)
{
   const char *names [3];
   fiasco_c_options_t *o = fiasco_c_options_new ();
   names [0] = argv [1]; names [1] = argv [2]; names [2] = NULL;
   fiasco_c_options_set_frame_pattern (o, "ip");
   fiasco_c_options_set_progress_meter (o, FIASCO_PROGRESS_NONE);
   /* fps, half_pixel_prediction, cross_B_search, B_as_past_ref (fiasco.h:375-380) */
   if (!fiasco_c_options_set_video_param (o, 25, 1, 0, 0)) { fprintf (stderr, "%s\n", fiasco_get_error_message ()); return 2; }
   if (!fiasco_coder (names, argv [3], 20.0, o)) { fprintf (stderr, "%s\n", fiasco_get_error_message ()); return 1; }
   return 0;
}
C
gcc $CFLAGS -I"$REF" "$DRV" "${objs[@]}" -o "$OUT/halfpixel_asan" -lm || { rm -f "$DRV"; exit 1; }
rm -f "$DRV"
# two 96 x 64 frames of tests/golden, the second displaced: negative vectors are certain
G=$HERE/../tests/golden
ASAN_OPTIONS=detect_leaks=0 FIASCO_DATA="$REF/data" "$OUT/halfpixel_asan" "$G/f0_96x64.pgm" "$G/f1_96x64.pgm" /tmp/halfpixel.fco > "$OUT/halfpixel_asan.txt" 2>&1
rc=$?
{ echo "# oracle/ref_halfpixel_asan.sh: the reference (gcc -O1 -fsanitize=address) coding f0_96x64.pgm f1_96x64.pgm, pattern ip, half_pixel = 1"; echo "# exit code $rc"; } | cat - "$OUT/halfpixel_asan.txt" | head -60 > "$OUT/halfpixel_asan_head.txt"
echo "ref_halfpixel_asan: exit code $rc, report in $OUT/halfpixel_asan.txt"
head -12 "$OUT/halfpixel_asan.txt"
