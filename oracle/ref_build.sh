#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the *real* reference coder as a checker.
#
# Compiles the FIASCO reference from the sources WHERE THEY LIE under
# /root/reference (nothing is copied into this repo) into oracle/_ref/:
#     oracle/_ref/libfiasco_ref.so   reference library (lib/ input/ output/ codec/)
#     oracle/_ref/cfiasco_ref        reference CLI     (bin/cwfa.c + params + getopt)
#     oracle/_ref/libfiasco_ref_big.so / cfiasco_ref_big
#                                    "limits extension" variant:
#                                    -DMAXSTATES/-DMAXLEVEL cannot be overridden from the
#                                    command line (plain #define in codec/wfa.h), so the
#                                    big variant is NOT built here; 4K parity is declared
#                                    "unpinned by a local reference build" in DESIGN.md.
#
# The sources include "config.h" unconditionally.  That file is autoconf output; it is
# produced here by running the reference's own pre-generated `configure` script OUT OF
# TREE (cwd = oracle/_ref/cfg), i.e. it is the genuine generated header, not a stand-in.
# The reference Makefiles are not used: the objects are compiled by the gcc lines below.
# Pinned build flags: -O2 -fcommon, baseline x86-64 (no FMA) -- SURVEY.md §8c.
#
# Only tests/, bench.py's cpu_baseline leg and __graft_entry__ may use oracle/_ref.
set -euo pipefail
REF=${FIASCO_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$REF/codec" ]; then
    echo "ref_build: $REF not present (GPU box?) -- keeping prebuilt $OUT" >&2
    exit 0
fi
mkdir -p "$OUT/cfg" "$OUT/obj"
if [ ! -f "$OUT/cfg/config.h" ]; then
    (cd "$OUT/cfg" && "$REF/configure" --quiet >configure.log 2>&1) || {
        echo "ref_build: configure failed, see $OUT/cfg/configure.log" >&2; exit 1; }
fi
# only the generated header is needed; the Makefiles/libtool that configure also emits are not
# used by this recipe and are not kept
find "$OUT/cfg" -mindepth 1 ! -name config.h -delete 2>/dev/null || true
CFLAGS="-O2 -g -fcommon -fPIC -w -DHAVE_CONFIG_H -I$OUT/cfg -I$REF -I$REF/lib -I$REF/input -I$REF/output -I$REF/codec -DFIASCO_SHARE=\"$REF/data\""
objs=()
for f in "$REF"/lib/*.c "$REF"/input/*.c "$REF"/output/*.c "$REF"/codec/*.c; do
    o="$OUT/obj/$(basename "$(dirname "$f")")_$(basename "${f%.c}").o"
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then
        gcc $CFLAGS -c "$f" -o "$o"
    fi
    objs+=("$o")
done
gcc -shared -fcommon -o "$OUT/libfiasco_ref.so" "${objs[@]}" -lm
cli=()
for f in cwfa params binerror getopt getopt1; do
    o="$OUT/obj/bin_$f.o"
    gcc $CFLAGS -I"$REF/bin" -c "$REF/bin/$f.c" -o "$o"
    cli+=("$o")
done
gcc -fcommon -o "$OUT/cfiasco_ref" "${cli[@]}" -L"$OUT" -lfiasco_ref -Wl,-rpath,'$ORIGIN' -lm
echo "ref_build: built $OUT/libfiasco_ref.so and $OUT/cfiasco_ref"
