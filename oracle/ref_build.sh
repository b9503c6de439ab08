#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the *real* reference coder as a checker.
#
# Compiles the FIASCO reference from the sources WHERE THEY LIE under
# /root/reference (nothing is copied into this repo) into oracle/_ref/:
#     oracle/_ref/libfiasco_ref.so   reference library (lib/ input/ output/ codec/)
#     oracle/_ref/cfiasco_ref        reference CLI     (bin/cwfa.c + params + getopt)
#     oracle/_ref/dfiasco_ref        reference decoder CLI (bin/dwfa.c) and
#     oracle/_ref/pnmpsnr_ref        PSNR tool (bin/pnmpsnr.c): decoded-PSNR known answers
#     oracle/_ref/share/             the reference's installed data files (data/*.fco, the initial bases)
#     oracle/_ref/libfiasco_ref_big.so / cfiasco_ref_big
#                                    "limits extension" variant: MAXSTATES 6000 -> 30000,
#                                    MAXLEVEL 22 -> 26, init_tree_model() uses entry 21 of its two count
#                                    tables for levels >= 22.  MAXSTATES / MAXLEVEL are plain #defines in
#                                    codec/wfa.h and cannot be overridden from the command line, so this
#                                    variant is compiled from a THROW-AWAY copy of the sources under
#                                    /tmp that is patched by the three sed lines below and deleted again;
#                                    only the objects and the two binaries land in oracle/_ref/.  It pins
#                                    what the stock reference cannot run at all: 4K (level 24) and 1080p
#                                    colour at the CLI defaults (> 6000 states), tests/golden/MANIFEST_BIG.json.
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
if [ -n "${FIASCO_SKIP_REF_BUILD:-}" ]; then
    echo "ref_build: FIASCO_SKIP_REF_BUILD set -- not touching $OUT" >&2
    exit 0
fi
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
# The generated header must say what the pinned streams were produced with: every macro that
# changes results, nothing else that does (oracle/ref_config.expect).
python3 - "$OUT/cfg/config.h" "$HERE/ref_config.expect" <<'PY' || exit 1
import re, sys
have = dict(re.findall(r'^#define\s+(\w+)\s*(.*)$', open(sys.argv[1]).read(), re.M))
bad = []
for line in open(sys.argv[2]):
    line = line.split('#')[0].split()
    if not line:
        continue
    if line[0] == 'absent':
        if line[1] in have: bad.append('%s must not be defined' % line[1])
    elif have.get(line[0], None) is None or have[line[0]].strip() != line[1]:
        bad.append('%s: expected %s, config.h has %r' % (line[0], line[1], have.get(line[0])))
if bad:
    sys.exit('ref_build: config.h differs from oracle/ref_config.expect:\n  ' + '\n  '.join(bad))
PY
{   # what the checker was built with (tests/golden/MANIFEST.json records the same at golden time)
    echo "gcc: $(gcc --version | head -1)"
    echo "config_h_sha256: $(sha256sum "$OUT/cfg/config.h" | cut -d' ' -f1)"
    echo "flags: -O2 -g -fcommon (baseline x86-64, no FMA contraction of the sources)"
} > "$OUT/BUILD_INFO"
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
# the reference's own decoder front end and PSNR tool (bin/dwfa.c, bin/pnmpsnr.c): they pin the decoded-PSNR
# figures of tests/golden/MANIFEST.json ("decoded_psnr": dfiasco -s 0, then pnmpsnr)
gcc $CFLAGS -I"$REF/bin" -c "$REF/bin/dwfa.c" -o "$OUT/obj/bin_dwfa.o"
gcc $CFLAGS -I"$REF/bin" -c "$REF/bin/pnmpsnr.c" -o "$OUT/obj/bin_pnmpsnr.o"
gcc -fcommon -o "$OUT/dfiasco_ref" "$OUT/obj/bin_dwfa.o" "$OUT/obj/bin_params.o" "$OUT/obj/bin_binerror.o" \
    "$OUT/obj/bin_getopt.o" "$OUT/obj/bin_getopt1.o" -L"$OUT" -lfiasco_ref -Wl,-rpath,'$ORIGIN' -lm
gcc -fcommon -o "$OUT/pnmpsnr_ref" "$OUT/obj/bin_pnmpsnr.o" "$OUT/obj/bin_binerror.o" -L"$OUT" -lfiasco_ref -Wl,-rpath,'$ORIGIN' -lm
# the reference's run-time data (what `make install' puts into $(pkgdatadir): the ASCII initial bases small /
# medium / large .fco that --basis-name names): installed beside the binaries so that the checker can be run
# with them where /root/reference does not exist (FIASCO_DATA=oracle/_ref/share); like everything under
# oracle/_ref/ they are build output, git-ignored, never part of the product
mkdir -p "$OUT/share"
cp "$REF"/data/*.fco "$OUT/share/"
echo "ref_build: built $OUT/libfiasco_ref.so, $OUT/cfiasco_ref, dfiasco_ref and pnmpsnr_ref"

# ---- the other entries of the reference's model registries (codec/domain-pool.c:188-236, codec/coeff.c:97-131)
# fiasco.h has no setter for c_options_t.id_domain_pool / id_d_domain_pool / id_rpf_model / id_d_rpf_model: the
# only way to make the reference run its "adaptive", "basis", "uniform" and "rle-no-chroma" pools or its
# "uniform" coefficient model is another default in codec/options.c:77-80.  One throw-away copy of THAT FILE
# per variant, four sed substitutions, linked with the stock objects: cfiasco_ref_<variant>.  They pin the
# oracle's restatement of those models (tests/test_oracle_pins.py::test_model_registry_*).
if [ -z "${FIASCO_SKIP_REF_MODELS:-}" ]; then
    mkdir -p "$OUT/obj_models"
    for v in "adaptive:adaptive:adaptive:adaptive:adaptive" "uniform:uniform:uniform:uniform:uniform" \
             "basis:basis:rle:adaptive:uniform" "nochroma:rle-no-chroma:rle:adaptive:adaptive" "rleuni:rle:adaptive:uniform:adaptive"; do
        IFS=: read -r name pool dpool rpf drpf <<< "$v"
        [ -x "$OUT/cfiasco_ref_$name" ] && [ "$OUT/cfiasco_ref_$name" -nt "$0" ] && continue
        TMPO=$(mktemp /tmp/fiasco_options_XXXXXX.c)
        sed -e "s/id_domain_pool 	  = strdup (\"rle\")/id_domain_pool = strdup (\"$pool\")/" \
            -e "s/id_d_domain_pool 	  = strdup (\"rle\")/id_d_domain_pool = strdup (\"$dpool\")/" \
            -e "s/id_rpf_model 	  = strdup (\"adaptive\")/id_rpf_model = strdup (\"$rpf\")/" \
            -e "s/id_d_rpf_model 	  = strdup (\"adaptive\")/id_d_rpf_model = strdup (\"$drpf\")/" \
            "$REF/codec/options.c" > "$TMPO"
        [ "$(grep -c "id_domain_pool = strdup (\"$pool\")\|id_d_domain_pool = strdup (\"$dpool\")\|id_rpf_model = strdup (\"$rpf\")\|id_d_rpf_model = strdup (\"$drpf\")" "$TMPO")" = 4 ] \
            || { echo "ref_build: the model-name patch did not apply ($name)" >&2; rm -f "$TMPO"; exit 1; }
        gcc $CFLAGS -c "$TMPO" -o "$OUT/obj_models/options_$name.o"
        rm -f "$TMPO"
        mobjs=()
        for o in "${objs[@]}"; do
            case "$o" in */codec_options.o) mobjs+=("$OUT/obj_models/options_$name.o");; *) mobjs+=("$o");; esac
        done
        gcc -shared -fcommon -o "$OUT/libfiasco_ref_$name.so" "${mobjs[@]}" -lm
        gcc -fcommon -o "$OUT/cfiasco_ref_$name" "${cli[@]}" -L"$OUT" -lfiasco_ref_$name -Wl,-rpath,'$ORIGIN' -lm
    done
    echo "ref_build: built the model-registry variants cfiasco_ref_{adaptive,uniform,basis,nochroma,rleuni}"
fi

# ---- `cfiasco -z 3' pinned (round 6): the zero-initialised variant ----
# With -z 3 (check_for_underflow / check_for_overflow / full_search, codec/approx.c:119-206,420) the reference reads
# members of the automatic `mp_t mp' of approximate_range() (codec/approx.c:88) that nobody has written: a step of
# a full_search run is kept without `mp->weight[]' being set (:439-446 read it), and the retry loops walk
# `indices[]' of a run that found nothing.  What the stock binary writes then depends on what the stack held.  This
# DECLARED variant defines those reads: ONE throw-away copy of codec/approx.c under /tmp with the declaration turned
# into `mp_t mp = mp_zero_;' (all members zero), compiled with the same flags and linked with the stock objects:
# cfiasco_ref_z3.  It pins the oracle's and the device's -z 3 streams (tests/golden/make_z3.py, "z3_cases");
# without -z 3 it writes what the stock reference writes (checked when the goldens are made).
if [ -z "${FIASCO_SKIP_REF_Z3:-}" ] && { [ ! -x "$OUT/cfiasco_ref_z3" ] || [ "$0" -nt "$OUT/cfiasco_ref_z3" ]; }; then
    mkdir -p "$OUT/obj_z3"
    TMPA=$(mktemp /tmp/fiasco_approx_XXXXXX.c)
    sed -e 's/^   mp_t\t  mp;$/   static const mp_t mp_zero_; mp_t mp = mp_zero_;/' "$REF/codec/approx.c" > "$TMPA"
    [ "$(grep -c 'static const mp_t mp_zero_; mp_t mp = mp_zero_;' "$TMPA")" = 1 ] \
        || { echo "ref_build: the -z 3 zero-initialisation patch did not apply" >&2; rm -f "$TMPA"; exit 1; }
    [ "$(diff "$REF/codec/approx.c" "$TMPA" | grep -c '^[<>]')" = 2 ] \
        || { echo "ref_build: the -z 3 patch changed more than one line" >&2; rm -f "$TMPA"; exit 1; }
    gcc $CFLAGS -c "$TMPA" -o "$OUT/obj_z3/codec_approx.o"
    rm -f "$TMPA"
    zobjs=()
    for o in "${objs[@]}"; do
        case "$o" in */codec_approx.o) zobjs+=("$OUT/obj_z3/codec_approx.o");; *) zobjs+=("$o");; esac
    done
    gcc -shared -fcommon -o "$OUT/libfiasco_ref_z3.so" "${zobjs[@]}" -lm
    gcc -fcommon -o "$OUT/cfiasco_ref_z3" "${cli[@]}" -L"$OUT" -lfiasco_ref_z3 -Wl,-rpath,'$ORIGIN' -lm
    echo "z3_variant: codec/approx.c:88 \`mp_t mp' zero-initialised (one sed line on a /tmp copy of that file)" >> "$OUT/BUILD_INFO"
    echo "ref_build: built $OUT/cfiasco_ref_z3 (-z 3 with defined reads)"
fi

# ---- limits extension: patched throw-away copy, same flags ----
if [ -z "${FIASCO_SKIP_REF_BIG:-}" ] && { [ ! -x "$OUT/cfiasco_ref_big" ] || [ "$0" -nt "$OUT/cfiasco_ref_big" ]; }; then
    TMPSRC=$(mktemp -d /tmp/fiasco_ref_big.XXXXXX)
    trap 'rm -rf "$TMPSRC"' EXIT
    cp -r "$REF/lib" "$REF/input" "$REF/output" "$REF/codec" "$REF/bin" "$REF/fiasco.h" "$TMPSRC/"
    sed -i -e 's/^#define MAXSTATES 6000$/#define MAXSTATES 30000/' \
           -e 's/^#define MAXLEVEL  22 *$/#define MAXLEVEL  26/' "$TMPSRC/codec/wfa.h"
    sed -i -e 's/counts_1 \[level\];/counts_1 [level < 22 ? level : 21];/' \
           -e 's/counts_0 \[level\] + counts_1 \[level\];/counts_0 [level < 22 ? level : 21] + counts_1 [level < 22 ? level : 21];/' \
           "$TMPSRC/codec/bintree.c"
    # the patch must have taken (a changed upstream file would otherwise give a silently stock build)
    grep -q '^#define MAXSTATES 30000$' "$TMPSRC/codec/wfa.h" && grep -q '^#define MAXLEVEL  26$' "$TMPSRC/codec/wfa.h" \
        && [ "$(grep -c 'level < 22 ? level : 21' "$TMPSRC/codec/bintree.c")" = 2 ] \
        || { echo "ref_build: the limits-extension patch did not apply" >&2; exit 1; }
    mkdir -p "$OUT/obj_big"
    BFLAGS="-O2 -g -fcommon -fPIC -w -DHAVE_CONFIG_H -I$OUT/cfg -I$TMPSRC -I$TMPSRC/lib -I$TMPSRC/input -I$TMPSRC/output -I$TMPSRC/codec -DFIASCO_SHARE=\"$REF/data\""
    bobjs=()
    for f in "$TMPSRC"/lib/*.c "$TMPSRC"/input/*.c "$TMPSRC"/output/*.c "$TMPSRC"/codec/*.c; do
        o="$OUT/obj_big/$(basename "$(dirname "$f")")_$(basename "${f%.c}").o"
        gcc $BFLAGS -c "$f" -o "$o"
        bobjs+=("$o")
    done
    gcc -shared -fcommon -o "$OUT/libfiasco_ref_big.so" "${bobjs[@]}" -lm
    bcli=()
    for f in cwfa params binerror getopt getopt1; do
        o="$OUT/obj_big/bin_$f.o"
        gcc $BFLAGS -I"$TMPSRC/bin" -c "$TMPSRC/bin/$f.c" -o "$o"
        bcli+=("$o")
    done
    gcc -fcommon -o "$OUT/cfiasco_ref_big" "${bcli[@]}" -L"$OUT" -lfiasco_ref_big -Wl,-rpath,'$ORIGIN' -lm
    rm -rf "$TMPSRC"; trap - EXIT
    echo "limits_extension: MAXSTATES 30000, MAXLEVEL 26, init_tree_model entry 21 for levels >= 22 (sed patch of a /tmp copy)" >> "$OUT/BUILD_INFO"
    echo "ref_build: built $OUT/libfiasco_ref_big.so and $OUT/cfiasco_ref_big (limits extension)"
fi
