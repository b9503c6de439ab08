/*
 *  frame_coder.hip -- the FIASCO encode-side hot path as ONE persistent gfx950 kernel:
 *  one 256-thread workgroup owns one frame and runs its complete partition search.
 *
 *  What runs here (reference file:line it is bit-compatible with):
 *    partition search        codec/subdivide.c:60-502   (serial state machine, lane 0,
 *                                                         explicit LDS stack)
 *    init_range              codec/subdivide.c:504-541,612-644
 *    matching pursuit        codec/approx.c:74-271,317-699 (domain-parallel, see below)
 *    inner-product tables    codec/ip.c:46-323
 *    state tables            codec/control.c:48-131,205-258
 *    rle pool / aac / tree   codec/domain-pool.c:621-852, codec/coeff.c:215-267,
 *    rate models             codec/bintree.c:35-73, lib/rpf.c:59-169, lib/misc.c:223-244
 *
 *  Parallel decomposition of one matching-pursuit call (D candidate domains):
 *    phase A (all waves)  per candidate d, fused: Gram-Schmidt update of rem_num/rem_den
 *                         against the vector chosen in the previous step + stage-1 cost
 *                         estimate e_d; wave-wide min of e_d per 64-candidate block.
 *    phase B (rounds)     exact replay of the reference's index-ordered scan with its
 *                         running `min_costs`: blocks whose min e_d cannot beat the
 *                         running minimum are skipped; the wave that owns the next block
 *                         evaluates its survivors' true costs lane-parallel and accepts in
 *                         index order by ballot/ffs (strict '<', codec/approx.c:459-462,592).
 *  Per-candidate scratch lives in registers (mp_reg.inc); mp_device.inc holds the rate
 *  terms, the general (HBM scratch) and the chroma (explicit list) variants of the scan.
 *  All float arithmetic keeps the reference's operation order; this file MUST be built
 *  with -ffp-contract=off (no FMA).  double log2() is evaluated once per call into small
 *  LDS tables (the rate models only ever need log2 of count/total ratios).
 *
 *  Device scope: gray and colour I frames (bands: codec/coder.c:738-833), `rle` pool,
 *  `adaptive` coefficients; the default build of this file covers the CLI defaults (block
 *  levels 6..10, <= 3 vectors), FC_VARIANT_BIG the other option sets (see below).
 *
 *  Register budget: the default build is compiled for 4 workgroups per CU, i.e. <= 128 VGPRs.
 *  The persistent loop makes EVERYTHING loop invariant in the compiler's eyes; what must not be
 *  computed once at kernel entry and kept for the kernel's lifetime is hidden from the hoisting
 *  (opaque thread index in the scan, out-of-line log2 tables, descriptor fields through
 *  sh.par in LDS).  tests/isa_spills.sh shows what still goes to scratch and from which line.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "frame_coder.h"

/* FC_VARIANT_WIDE: FC_WIDE_B (512, or 1024 for the default geometry: csrc/Makefile) threads per
 * frame instead of 256 -- for launches with no more frames than CUs and for frames with more
 * states than 12 x 256 (4K): more lanes per frame, 9216 states in the register slots, one
 * workgroup per CU */
#ifndef FC_VARIANT_WIDE
#define FC_VARIANT_WIDE 0
#endif
#if FC_VARIANT_WIDE
#ifndef FC_WIDE_B
#define FC_WIDE_B 512
#endif
#define B       FC_WIDE_B
#if defined(FC_VARIANT_BIG) && FC_VARIANT_BIG
#define FC_KREG 12               /* 6144 states with 4 orthogonal vectors each in registers */
#elif !defined(FC_KREG)
#define FC_KREG (9216 / FC_WIDE_B)
#endif
#else
#define B       FC_BLOCK
#define FC_KREG 12
#endif
/* Two builds of this file (csrc/Makefile): the default one for the CLI's -z 0 geometry (block
 * levels 6..10, <= 3 vectors: 4 frames per CU) and FC_VARIANT_BIG for everything else the
 * device supports (block levels 4..12, <= 5 vectors, second-domain retry: 2 frames per CU). */
#ifndef FC_VARIANT_BIG
#define FC_VARIANT_BIG 0
#endif
#if FC_HM && !(FC_VARIANT_BIG && FC_VARIANT_WIDE)
#error "FC_HM is a variant of the 512-thread big build"
#endif
#if FC_GM && !FC_HM
#error "FC_GM is a variant of the FC_HM build"
#endif
#if FC_VARIANT_BIG
#if FC_GM
#define FC_KERNEL    fiasco_frame_kernel_big_gm
#define FC_LAUNCH    fc_launch_big_gm
#define FC_OCCUPANCY fc_occupancy_big_gm
#elif FC_HM
#define FC_KERNEL    fiasco_frame_kernel_big_hm
#define FC_LAUNCH    fc_launch_big_hm
#define FC_OCCUPANCY fc_occupancy_big_hm
#elif FC_VARIANT_WIDE
#define FC_KERNEL    fiasco_frame_kernel_big_wide
#define FC_LAUNCH    fc_launch_big_wide
#define FC_OCCUPANCY fc_occupancy_big_wide
#else
#define FC_KERNEL    fiasco_frame_kernel_big
#define FC_LAUNCH    fc_launch_big
#define FC_OCCUPANCY fc_occupancy_big
#endif
#define FC_PIXELS    4096        /* 2^lc_max, lc_max <= 12 */
#define FC_NIP       4           /* orthogonal vectors kept per candidate: max_elements - 1 */
#define FC_CLMAX     2048
#define FC_WG_PER_CU (FC_VARIANT_WIDE ? 1 : 2)
#else
#if FC_VARIANT_WIDE
#if defined(FC_SPEC) && FC_SPEC
#define FC_KERNEL    fiasco_frame_kernel_spec_wide
#define FC_LAUNCH    fc_launch_spec_wide
#define FC_OCCUPANCY fc_occupancy_spec_wide
#define FC_SPEC_SLOT_BYTES fc_spec_slot_bytes_wide
#elif defined(FC_GRAM_TRI) && FC_GRAM_TRI
#define FC_KERNEL    fiasco_frame_kernel_wide_tri
#define FC_LAUNCH    fc_launch_wide_tri
#define FC_OCCUPANCY fc_occupancy_wide_tri
#else
#define FC_KERNEL    fiasco_frame_kernel_wide
#define FC_LAUNCH    fc_launch_wide
#define FC_OCCUPANCY fc_occupancy_wide
#endif
#elif defined(FC_SPEC) && FC_SPEC
#define FC_KERNEL    fiasco_frame_kernel_spec
#define FC_LAUNCH    fc_launch_spec
#define FC_OCCUPANCY fc_occupancy_spec
#define FC_SPEC_SLOT_BYTES fc_spec_slot_bytes
#else
#define FC_KERNEL    fiasco_frame_kernel
#define FC_LAUNCH    fc_launch
#define FC_OCCUPANCY fc_occupancy
#endif
#define FC_PIXELS    1024
#define FC_NIP       2
#define FC_CLMAX     768         /* Sh::cl: states of a chroma block with table entries somebody reads */
#ifndef FC_WG_PER_CU
/* workgroups (frames) per CU the kernel is built for: four 256-thread frames = 4 waves per SIMD,
 * i.e. at most 128 VGPRs and 40 KB of LDS per frame */
#define FC_WG_PER_CU (FC_VARIANT_WIDE ? 1 : 4)
#endif
#endif
/* FC_SPEC: block-level speculation (frame_coder.h, FcSpecCtl): a frame is served by several
 * workgroups that share its slab -- the 256-thread default build with the chain / verifier roles.
 * A build of its own so that the code of the launches that fill the chip with frames (one
 * workgroup per frame, no spare workgroup slots to speculate with) stays what it is. */
#ifndef FC_SPEC
#define FC_SPEC 0
#endif
#ifndef SPEC_THR
#define SPEC_THR 1.2f          /* FC_SPEC: see SpecLocal.mlc */
#endif
/* FC_SPINE: left-spine batching (mp_wave.inc) -- the linear-combination searches of a node and of its
 * chain of first children run at once, one per wave.  Bit-exact (the parked results are what the
 * searches would have found: same models, same dictionary), but it does not pay: a wave on its own
 * needs 3.4 x the time of the four waves together for one search, so a spine of three ranges costs what
 * three searches cost (519 .. 572 frames/s against 549 on the bench batch).
 * Kept as a build switch of the 256-thread default build for whoever wants to re-measure; off. */
/* FC_D5T: the table of level-images_level dots is kept state-major, d5T[state][label][NA / 2] (address a ->
 * label a & 1, column a >> 1), so that the first pass of op_ipis reads four consecutive slots of one term with
 * ONE 16-byte load instead of four 4-byte gathers from four rows.  Same values, same sums. */
#ifndef FC_D5T
#define FC_D5T (!FC_VARIANT_BIG)
#endif
#if FC_D5T && FC_VARIANT_BIG
#error "FC_D5T: the big build reads d5 rows as matching pursuit numerators"
#endif
#if FC_D5T
#define D5_AT(P, NA, a, s) ((unsigned) (s) * (unsigned) (NA) + (unsigned) ((a) & 1) * ((unsigned) (NA) >> 1) + ((unsigned) (a) >> 1))
#else
#define D5_AT(P, NA, a, s) ((unsigned) (a) * (unsigned) (P) + (unsigned) (s))
#endif
/* FC_PRIO_ROTATE: rotating instruction priority of the frames that share a CU (kernel loop); the 256-thread default
 * build, whose launches put four workgroups on a CU */
#ifndef FC_PRIO_ROTATE
#define FC_PRIO_ROTATE (!FC_VARIANT_BIG && !FC_VARIANT_WIDE && !FC_SPEC)
#endif
#ifndef FC_PRIO_SHIFT
#define FC_PRIO_SHIFT 20            /* 2^20 ticks of the 100 MHz wall clock: 10 ms per turn (82 us .. 42 ms measured: 580 .. 589 frames/s) */
#endif
#ifndef FC_SPINE
#define FC_SPINE 0
#endif
#if FC_SPINE && (FC_VARIANT_BIG || FC_VARIANT_WIDE || FC_SPEC)
#error "FC_SPINE is a variant of the 256-thread default build"
#endif
#define FC_SPINE_W 4            /* ranges of a spine searched at once: one per wave of the 256-thread build */
/* FC_BLKEST: the sweep of a matching-pursuit pass prices whole 64-state blocks at once where the position
 * pricing is the same for every candidate of the block (mp_device.inc, stage1_block_price).  Exact as
 * well, and also no gain (520 against 549 frames/s): from the second pass on a fifth of the blocks hold a
 * breakpoint of the pricing and go the long way round anyway.  Off. */
#ifndef FC_BLKEST
#define FC_BLKEST 0
#endif
#if FC_BLKEST && (FC_VARIANT_BIG || FC_SPEC)
#error "FC_BLKEST needs the default geometry without speculation (Sh::cum)"
#endif
/* FC_EST_RCP: the sweep's block minima are taken over a tight lower bound of the estimates
 * (reciprocal instead of division, stage1<.., LBQ> in mp_device.inc) */
#ifndef FC_EST_RCP
#define FC_EST_RCP 1
#endif
/* FC_MIN4: block minima of the register scan four slots at a time (interleaved DPP chains) */
#ifndef FC_MIN4
#define FC_MIN4 1
#endif
#define MAXED   FC_MAXED
/* edges per label a state of this build can have (= max_elements the build accepts): the table
 * ops read and gather exactly that many term slots (+ the tree child), not the format's 5 */
#define FC_MAXE (FC_NIP + 1)
#define NOEDGE  (-1)
#define RANGE_  (-1)
#define MAXCOSTS 1e20f
#define BIGF    3.0e38f
#define MIN_NORM 2e-3f

enum { OP_DONE = 0, OP_INIT_RANGE, OP_APPROX, OP_IPIS_INCR, OP_APPEND, OP_NOP, OP_CHROMA,
       OP_PRED_SETUP, OP_PRED_FINISH, OP_NORMS, OP_MC_SEARCH, OP_SPEC_CKPT };
enum { PH_ENTER = 0, PH_AFTER_INIT, PH_AFTER_LC, PH_CHILD, PH_CHILD2, PH_CHILD_RET, PH_DECIDE,
       PH_AFTER_APPEND, PH_PRED_BEGIN, PH_PRED_RECURSE, PH_PRED_RET, PH_PRED_DONE, PH_PRED_MC2, PH_PRED_GO,
       PH_SPEC_END };
enum { MV_NONE = 0, MV_FORWARD = 1, MV_BACKWARD = 2, MV_INTERPOLATED = 3 };
/* FC_DUP_OP=<op>: developer build that runs one of the idempotent table operations (OP_INIT_RANGE, OP_APPEND,
 * OP_IPIS_INCR: they write a function of what they read, the second run writes the same values) TWICE: the difference
 * of the PMC traffic counters to the plain build is that operation's HBM traffic (tests/gpu_traffic_by_op.sh,
 * profiles/r06_traffic_by_op.txt).  Same streams; the roofline counters of the op count double. */
#ifdef FC_DUP_OP
#define FC_DUP(o, call) do { if ((o) == FC_DUP_OP) { __syncthreads(); call; } } while (0)
#else
#define FC_DUP(o, call) do { } while (0)
#endif
#if FC_VARIANT_BIG
#define FC_DEPTH FC_MAXDEPTH_BIG
#elif FC_VARIANT_WIDE
#define FC_DEPTH FC_MAXDEPTH
#else
#define FC_DEPTH FC_MAXDEPTH_NARROW   /* deeper frames go to the 512-thread build (core_hip.cpp) */
#endif

/* edge slots of a range record: the vectors a build can keep plus the terminator (the stack of
 * range records is a third of the default build's LDS) */
#define RANGE_E (FC_MAXE + 1)
struct __attribute__((aligned(16))) Range {     /* copied as 128-bit LDS words by the serial lane */
    int   x, y, image, address, level, tree;
    float weight[RANGE_E];
    short into[RANGE_E];
    float err, tree_bits, matrix_bits, weights_bits;
#if FC_VARIANT_BIG
    float nd_tree_bits, nd_weights_bits, mv_tree_bits, mv_coord_bits;   /* codec/cwfa.h:68-73 */
    int   prediction;
    short mv[5];                           /* type, fx, fy, bx, by (mv_t, codec/wfa.h:58-72) */
#endif
};

struct Pool {                    /* rle model, codec/domain-pool.c:621-630 */
    short count[MAXED + 1];
    unsigned short total, n, max_domains, y_index;
    short d0_index;
    unsigned short d0_yindex, d0_n;
};

struct __attribute__((aligned(16))) SFrame {
    Range rg, lrange, rrange, child[2];
    Pool  pool0, pool_lc;
    float max_costs, lincomb, subdiv, ret, price;
    int   label, states, phase, leaf, coop;
    int   y_state, ny[2];        /* co-located luminance state of the range / of its children */
#if FC_GM
    int   rn0;                   /* Pool.n of the RESTING pool at the entry of the node (see PH_AFTER_INIT) */
#endif
#if FC_SPEC
    int   ckpt;                  /* a checkpoint of the workgroup was taken at the entry of this node */
#endif
#if FC_VARIANT_BIG
    /* prediction (codec/prediction.c:96-208): `pred` / `delta` are the arguments of the same name
     * of subdivide(); the rec_* members are what predict_range keeps of the subdivision result */
    int   pred, delta, try_pred, pred_done, rec_states;     /* try_pred: 1 nd, 2 mc */
    int   norm_first, norm_done;
    Pool  dpool0, pool_rec, dpool_rec;
    Range prange;                /* range of the residual search */
    float pred_max, pred_costs, nd_w, nd_wbits, nd_tbits;
#endif
};

struct MPState {
    int   n, best_n, index, D, N, level, image, address, row_state;
    short indices[MAXED + 1], into[MAXED + 1];
    float weight[MAXED];
    /* the RPF symbols of weight[0..2] as full_eval quantised them (SYMP_*; rtob(btor(sym)) == sym: what mp_step_prepare
     * and models_update would compute from the weights again, ~45 instructions of the serial lane apiece); an entry
     * that is not known is 0 (the scans with scratch in HBM do not carry them) */
    unsigned symp;
    float matrix_bits, weights_bits, err, costs, min_costs;
    float sel_ipdo[MAXED][MAXED];
    float norm_ov[MAXED + 1], ipio[MAXED + 1];
    short psorted[MAXED + 1];
    int   np;
    float wb_dc, wb_nd, norm, ab, price;
    int   y_state, ypos;         /* usable co-located luminance state / its list position, or -1 */
#if FC_VARIANT_BIG
    const float *numrow;         /* <range, state> row of the call: ipis slot, d5 or d4 address */
    short excl[MAXED + 1];       /* list positions excluded from this run, NOEDGE terminated */
#endif
#if FC_GM
    short kq[MAXED + 1];         /* quasi-arithmetic pools: probability index of the kept vectors' positions */
#endif
    /* per-step uniform parts of the stage-1 position pricing (mp_device.inc, StepCtx) */
    float s1_pre[MAXED], s1_sfx[MAXED], s1_z0, s1_zy;
    int   s1_last[MAXED], s1_k[MAXED], s1_thr[MAXED];
    unsigned s1_cd, s1_has;
};

/* aac model (coeff.c:190-208): totals first, then the counts, one 16-byte aligned block so
 * that a snapshot is a short run of 128-bit LDS copies */
struct __attribute__((aligned(16))) CoeffBuf {
    short tot[16];                 /* coeff_nt <= 16 contexts */
    short cnt[FC_VARIANT_BIG ? FC_MAXCOEFF_BIG : FC_MAXCOEFF];
};
#if FC_VARIANT_BIG
#define SNAP_POOL16 880            /* uint4 slots for aac snapshots: depth x 2 x n16 (what outgrows it lives in HBM) */
#define SNAP_TM_WORDS 2392         /* tree-model snapshots: depth x 4 x MAXLEVEL words */
#else
/* aac snapshots of the default build: one slot per depth (the models at the entry of the node) and
 * one more for each block level that has both a linear combination and children (the models
 * after the combination): (depths + levels) x n16 uint4.  The 256-thread build is sized for the
 * frames the stock reference accepts (level <= 22) at the CLI's models; what needs more goes to
 * the 512-thread build (one frame per CU, LDS to spare) -- core_hip.cpp routes by these numbers. */
#ifndef SNAP_POOL16
#define SNAP_POOL16 (FC_VARIANT_WIDE ? FC_SNAP16_WIDE : FC_SNAP16_NARROW)
#endif
/* the default build never prices with the second tree model (prediction, big build only): a
 * snapshot holds the first one alone, 2 x MAXLEVEL words rounded to 16 bytes (21 depths x 13 uint4) */
#define SNAP_TM_WORDS (FC_VARIANT_WIDE ? FC_SNAPTM_WIDE : FC_SNAPTM_NARROW)
#endif
#if FC_VARIANT_BIG || FC_VARIANT_WIDE
#define NBLOCKMIN   256            /* 64-candidate blocks: D <= 16384 */
#else
#define NBLOCKMIN   64             /* the 256-thread default build is given P <= 3072 (core_hip.cpp) */
#endif
#define TM_WORDS    (4 * 26 + 8)   /* 112 words = 28 uint4 */

struct RoundBox {                    /* mp_reg.inc: winner of the running step, in LDS */
    /* running min_costs, one slot per round parity: the owner of round r publishes into
     * m2[r & 1] and everybody reads it after the round's barrier.  With a single slot a fast
     * owner of round r + 1 could overwrite the value before a slow wave has read round r's
     * (seen as rare non-deterministic streams with four frames per CU) */
    float m2[2];
    int   state;                     /* winning state or -1 */
    int   idx;                       /* its list position (list-based scan only) */
    float cost, mbits, wbits, err, f[MAXED];
    float num, den, ip[MAXED - 1];
    unsigned evals, blockevals;
    unsigned symp;                   /* MPState::symp of the winner's weights */
};
/* sym + 2 in 10 bits per weight (sym = -1 .. 511); 0 = not known: that entry is quantised again (rtob) by whoever needs it
 * -- e.g. a weight left over from another run of the same call under full_search (codec/approx.c:439-446) */
#define SYMP_NONE 0u
#define SYMP_HAS(p, k) ((k) < 3 && (((p) >> (10 * (k))) & 1023u) != 0u)
#define SYMP_GET(p, k) ((int) (((p) >> (10 * (k))) & 1023u) - 2)
#define SYMP_PUT(sym, k) ((unsigned) ((sym) + 2) << (10 * (k)))

struct Sh {
    RoundBox rb;
    SFrame   st[FC_DEPTH];
    int      sp;
    int      op, a0, a1, a2, a3;
#if FC_VARIANT_BIG
    unsigned coopW, coop_seq;      /* workgroups of this frame (FcCoop), table builds published so far */
    int      coopD, coop_minsub;   /* FcCoop.depth / .minsub */
    unsigned long long coop_ticks; /* FcCoop.done_ticks */
#endif
    Pool     pool;
    CoeffBuf cb;
#if FC_VARIANT_BIG
    /* the second set of models (d_domain_pool, d_coeff; codec/coder.c:716-736).  The two `rle'
     * pools hold the same state list at all times (every state is offered to both,
     * codec/subdivide.c:571-581), only the counters differ: pool_states / pos are shared.
     * sh.pool / sh.cb / the quantiser in sh.par are the ACTIVE set: the normal models, or the
     * delta models while the residual of a predicted range is searched (swapped in and out by
     * OP_PRED_SETUP / OP_PRED_FINISH); the other set rests in dpool / dcb / dq. */
    Pool     dpool;
    CoeffBuf dcb;
    struct { int rpf_mant, dc_mant, sy, dcs; float rpf_range, dc_range; int half_nd, half_dc; } dq;
    int      nslot;                /* aac snapshot slots per depth: 2, or 5 with prediction */
    uint4   *snap_tm_p;            /* tree-model snapshots: snap_tm, or HBM with prediction */
    int      pred_active, pred_lo, pred_rec;   /* a residual search is running; displaced ids */
    struct { int type, fx, fy, bx, by; float bits, tree_bits; } mc;      /* result of OP_MC_SEARCH */
    unsigned long long mcred[B / 64];
    unsigned pred_saved[FC_MAXSAVE / 32];      /* their table rows are in the save area */
#endif
    uint4    snap_pool[SNAP_POOL16];
    uint4   *snap;                 /* snapshots live here: snap_pool, or HBM when they outgrow it */
    int      n16;                  /* uint4 per aac snapshot */
    __attribute__((aligned(16))) unsigned tm[TM_WORDS];
    __attribute__((aligned(16))) unsigned snap_tm[SNAP_TM_WORDS];
    float    m0tab[12];
    double   lgdc[FC_MAXSYM], lglv[FC_MAXSYM], lglv_m1;
    float    Ltab[MAXED + 1];
    float    Q0, Q1;
    float    tb[2];                /* default build: tree_bits (LEAF, CHILD) of the level being approximated (mp_tables) */
    MPState  mp;
#if FC_SPINE
    /* left-spine batching (mp_wave.inc): search state and result of the range w levels below the top node
     * of the current spine, with the log2 table of its level's coefficient context and its tree prices;
     * slot 0 is the top node itself.  spine_n ranges were searched when the node at stack depth spine_top
     * was entered; spine_next is the depth of the next one the partition search may pick up; spine_use
     * (lane 0, per OP_APPROX) is the slot of the node being approximated, 0 = search it now. */
    struct SpineSlot { MPState mp; double lglv[16], lglv_m1; float tb[2]; } spine[FC_SPINE_W];
    int      spine_top, spine_n, spine_next, spine_use;
    /* ticks (100 MHz) and calls of OP_APPROX by kind: 0 a spine of K >= 2 ranges, 1 a parked result picked
     * up, 2 one range searched by the whole workgroup; [3] = ranges searched in spines (DevFrame.dbg) */
    unsigned long long spine_t[3];
    unsigned spine_c[4];
#endif
#if FC_VARIANT_BIG
    MPState  mp_keep;              /* best result so far of a call with retries */
    int      apx_stage, apx_it, apx_more;   /* retry plan of approximate_range (lane 0) */
#endif
    float    blockmin[NBLOCKMIN];
#if FC_BLKEST
    /* cum[b] = pool position of the first pool state with id >= 64 b (states enter the rle pool in id
     * order, codec/domain-pool.c:832-852): the positions of block b are the run [cum[b], cum[b + 1]), the
     * last block's ends at pool.n.  Written when state 64 b is stored (store_new_state); an entry is
     * rewritten whenever that id is created again after a removal, so it always fits the dictionary. */
    unsigned short cum[NBLOCKMIN + 2];
#endif
    /* 16-byte aligned: op_d5 reads the block's pixels with 128-bit LDS loads (a member added in front of them in round 6
     * shifted them by four bytes: init_range +10 %) */
    __attribute__((aligned(16))) float pixels[FC_PIXELS];
    float    norms[FC_PIXELS / 32];  /* squared norms of the sub-blocks, heap order (NS <= 127) */
    unsigned long long tk[16];     /* ticks per op (lane 0) */
    struct {
        unsigned long long bytes_mp, bytes_img, bytes_gram, n_mp, n_steps, n_blocks, n_appends,
                           n_fulleval, n_blockevals, t_mpA, t_mpB;
    } cnt;                         /* DevFrame counters of the same names */
#ifdef FC_PM
    unsigned long long pm[8], pm_t;
    int pm_prev;
#endif
#ifdef FC_SERIAL_PROFILE
    unsigned long long tk_ph[8], ph_t0, tk_init[2], tk_apx[4];
    int      ph_prev;
#endif
    /* colour frames (codec/coder.c:775-800): band being coded, its dynamic minimum block
     * level, root states of the finished bands, states that own tables (= end of Y band) */
    int      band, lc_min, tree_band[3], ystates, after_chroma;
    short    dl[64];               /* candidate list of a chroma call: pool + luminance state */
#if !FC_SPEC
    /* chroma bands: the states whose <sub-block, state> entries of the current block anybody reads (chroma_need) */
    short    cl[FC_CLMAX];
    int      cln;
#endif
    unsigned long long red[B / 64];
    /* term lists of the state being appended (uniform for the whole workgroup) */
    int      gs_idx[2][MAXED + 1], gs_n[2], gs_c[2], gs_raw_idx[2][MAXED + 1];
    float    gs_raw_w[2][MAXED + 1];
    float    gs_w[2][MAXED + 1];
    /* parameters the serial lane reads per range, copied from the frame descriptor once (a
     * field of the descriptor is a global-memory round trip in the out-of-line search code) */
    struct {
        int lc_max, width, height, limit_states, PA, P, ML; float price, chroma_decrease;
        /* the same for the matching pursuit: table bases and quantiser parameters */
        float *gram, *diag, *ipis; int16_t *pos; unsigned gram_ls;
        float *gcol;               /* triangular build: DevFrame.gcol */
        float *d5, *d4;            /* big build: the active level-5 / level-4 dot tables */
        const unsigned *l2_keys; const double *l2_vals; unsigned l2_mask;
        int max_elements, rpf_mant, dc_mant, sy, dcs, gl0, images_level, lc_min_opt, trace_on;
        int snap_b1;               /* default build: first "after the linear combination" snapshot slot minus its depth */
        /* automaton arrays for the serial lane: through the frame descriptor (a generic reference in
         * the out-of-line search code) every access is a flat_ instruction behind a descriptor read */
        int16_t *at_tree, *at_into, *at_pool; float *at_weight, *at_final; uint8_t *at_los, *at_dtype, *at_ycol;
        uint16_t *at_x, *at_y; int color;
        float rpf_range, dc_range;
        /* rtob(0.5) in the two RPF formats of the ACTIVE coefficient model: the symbol of the placeholder weight of
         * the stage-1 estimates (codec/approx.c:457) -- a constant of the frame (and of the model set), not of the call */
        int half_nd, half_dc;
    } par;
#if FC_GM
    /* generic models (frame_coder.h FC_GM): kinds of the ACTIVE [0] and the resting [1] model set (pool, coefficients),
     * which of the two current probability-index arrays of DevFrame.gq is the active set's, and -- per call of the
     * matching pursuit -- the price of the empty domain list, of the kept vectors of the running step, log2(1 / n) */
    struct { int pk[2], ck[2], qa; float base, kept; double lg1; int16_t *gq; int P; } gm;
#endif
    int      states;               /* wfa->states */
    int      flim;                 /* Gram tables: states below it have mirrored entries */
    int      failed;
#if FC_SPEC
    /* A verifier sees the states the frame had at the entry of its block, [0, gap_lo), and the
     * states its own search appends, which get ids from gap_hi on (a private index range of every
     * table of the shared slab); the ids in between belong to the chain, which is ahead and still
     * writes them: nothing may look at them.  Chain: gap_lo == gap_hi == 0. */
    int      gap_lo, gap_hi, gap_shift;        /* gap_shift = gap_hi - gap_lo: what the gap adds to a state count */
    unsigned deadmask;             /* scan slots (B candidates each) that lie inside the gap */
    int      cap;                  /* state ids of this workgroup end here (FC_ERR_CAPACITY) */
    int      blk;                  /* chain: blocks of the largest block level entered so far (index into the host's list) */
    int      tab_shared;           /* the block's tables are in a buffer of the frame's ring (sh.par.ipis / d5) */
    int      tab_from;
    struct SpecLocal {
        FcSpecCtl *ctl;
        char     *slots;           /* FC_SPEC_W checkpoints of sizeof(Sh) bytes */
        int       role, on;        /* 0 chain, 1 .. T table workers, then verifiers; on: the frame speculates at all */
        int       mode;            /* the same for the partition search: 0, 1 (chain, on), 2 + floor (verifier) */
        int       T;
        int       chroma_tabs;     /* chain, chroma bands of a colour frame: the other workgroups build the blocks' tables */
        char     *tabs;            /* FC_SPEC_R table buffers */
        unsigned  rb_s[32];        /* chain: state count it returned to at the end of epoch e, [e % 32] */
        unsigned  blkof[FC_SPEC_W];    /* chain: block index of the checkpoint in a slot */
        unsigned  sk[FC_SPEC_W];       /* chain: states at that checkpoint */
        unsigned long long n_tab_used, n_tab_missed, n_adopted;
        int       floor;           /* verifier: stack depth of the block it verifies */
        unsigned  head, commit;    /* chain: checkpoints published / verdicts consumed */
        unsigned  spec_mask;       /* chain: per slot, the block's subtree was left to its verifier */
        int       nospec;          /* chain: the block being entered is searched here (wrong guess before) */
        unsigned  epoch;           /* chain: its count of returns; verifier: the epoch of its task */
        int       verdict, abort, busy;  /* verifier; busy: counted in FcSpecCtl.busy */
        unsigned  ops;
        /* chain: which blocks to guess about.  A wrong guess costs the blocks the chain ran ahead plus
         * the search of the block; searching a block here costs that search alone.  The costs of a
         * block's combination tell the two kinds apart fairly well: blocks whose combination costs more
         * than SPEC_THR x the running mean over the blocks that kept theirs are searched here. */
        float     mlc, lin[FC_SPEC_W];
        unsigned  nlc;
        float     learn;           /* costs of a combination that won in a search of the chain's own, not yet in mlc */
        unsigned long long n_tasks, n_confirmed, n_wrong, n_timeout, n_inline, t_wait;
        /* chain: append helpers (FcSpecCtl.app_*): how many, from which row length, rows published, given up */
        unsigned  app_H, app_min, app_seq, app_off;
        unsigned long long n_app_dealt, t_app_wait;
    } sl;
#endif
};
#if FC_GM
/* generic models: kinds, and the probability-index arrays of the quasi-arithmetic pools in DevFrame.gq */
#define GM_QAC(k)   ((k) == FC_PK_ADAPTIVE || (k) == FC_PK_BASIS)
#define GM_RLE(k)   ((k) == FC_PK_RLE || (k) == FC_PK_RLE_NO_CHROMA)
#define GQ_CUR(sh, set)          ((sh).gm.gq + (size_t) ((sh).gm.qa ^ (set)) * (sh).gm.P)       /* set 0: active, 1: resting */
#define GQ_SNAP(sh, depth, slot) ((sh).gm.gq + (size_t) (2 + (depth) * 5 + (slot)) * (sh).gm.P)
/* snapshot slots of a depth: 0 pool0, 1 pool_lc, 2 dpool0, 3 pool_rec, 4 dpool_rec (SFrame) */
#endif
#if FC_SPEC
#define DEAD(sh, s) ((unsigned) ((int) (s) - (sh).gap_lo) < (unsigned) ((sh).gap_hi - (sh).gap_lo))
#else
#define DEAD(sh, s) false
#endif

/* ------------------------------------------------------------------ small helpers */

__device__ __forceinline__ unsigned width_of_level(int l)  { return 1u << (l >> 1); }
__device__ __forceinline__ unsigned height_of_level(int l) { return 1u << ((l + 1) >> 1); }

/* lib/rpf.c:59-112 (x86 masks variable shift counts to 5 bits; so does this) */
__device__ int rtob_dev(float f, int mant, float range)
{
    f /= range;
    unsigned bits = __float_as_uint(f);
    unsigned m = bits & 0x7fffffu;
    int e = (int) ((bits >> 23) & 0xffu) - 126;
    int sign = (int) (bits >> 31);
    m = (m >> 1) | (1u << 22);
    if (e > 0) m <<= ((unsigned) e & 31u);
    else       m >>= ((unsigned) (-e) & 31u);
    m >>= (23 - mant - 1);
    m += 1;
    m >>= 1;
    if (m == 0) return -1;
    if (m >= (1u << mant)) return sign;
    return (int) (((m & ((1u << mant) - 1)) << 1) | (unsigned) sign);
}

/* lib/misc.c:223-244 */
__device__ __forceinline__ unsigned bits_bin_code(unsigned value, unsigned maxval)
{
    unsigned k = 31u - (unsigned) __clz((int) (maxval + 1));
    unsigned r = (maxval + 1) - (1u << k);
    return value < maxval + 1 - 2 * r ? k : k + 1;
}

/* probability index -> shift n of the quasi-arithmetic model (domain-pool.c:970-999) */
__device__ __forceinline__ int qac_shift(int index)
{
    int n = 1, start = 0;
    while (index >= start + (1 << n)) { start += 1 << n; n++; }
    return n;
}

/* Out-of-line functions get the frame descriptor through a generic reference, so every table
 * pointer they read is per-lane data to the compiler (64-bit address arithmetic in VGPRs for
 * each access).  The pointers ARE uniform: moving them to scalar registers leaves one 32-bit
 * lane offset per access. */
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ GLOBAL_AS T *uniform_ptr(T *p)
{
    unsigned long long v = (unsigned long long) p;
    unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) v);
    unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (v >> 32));
    /* known to be HBM (never LDS/scratch): global_load with a scalar base, not flat_load */
    return (GLOBAL_AS T *) (((unsigned long long) hi << 32) | lo);
}

/* element i of a table behind a scalar base: the byte offset is formed in 32 bits so that
 * the access is `global_load v, v_off, s[base:base+1]` (tables are < 4 GB apart from gram,
 * which is not accessed this way) */
template <typename T>
__device__ __forceinline__ T ldg(GLOBAL_AS const T *base, unsigned i)
{
    return *(GLOBAL_AS const T *) ((GLOBAL_AS const char *) base + i * (unsigned) sizeof(T));
}
template <typename T>
__device__ __forceinline__ void stg(GLOBAL_AS T *base, unsigned i, T v)
{
    *(GLOBAL_AS T *) ((GLOBAL_AS char *) base + i * (unsigned) sizeof(T)) = v;
}

/* ------------------------------------------------------------------ hand-offs between workgroups
 *
 * Per-XCD L2s are not coherent with each other and a CU's vector L1 is never refreshed by another CU's stores
 * (MI355X_MICROARCH.md, "inter-workgroup visibility"): data for another workgroup is PUBLISHED -- every wave drains its
 * stores, the workgroup meets, ONE lane writes the XCD L2's dirty lines back (agent-scope release) and only then stores
 * the flag -- and TAKEN by polling the flag relaxed, ONE agent-scope acquire (drops this CU's L1) and a barrier before
 * the plain loads.  The explicit waits are not decoration: ROCm 7.2 drops the `s_waitcnt vmcnt(0)' behind `buffer_wbl2'
 * whenever its scoreboard says the publishing wave has nothing outstanding, and the flag then overtakes the write-back
 * (round 6: the append helpers read the PREVIOUS row's descriptor until the wait was written out). */
#define WAVE_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
/* lane 0 of a workgroup whose waves have all drained and met (WAVE_DRAIN(); __syncthreads();): after this a relaxed
 * agent-scope store / fetch_add of the flag publishes everything the workgroup has written */
__device__ __forceinline__ void publish_release(void)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
/* the taker's side, one lane, after it has seen the flag (relaxed): nothing stale of the publisher's data in this CU's L1 */
__device__ __forceinline__ void take_acquire(void)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

/* ------------------------------------------------------------------ table access */

/* Gram tables, two layouts (frame_coder.h): full symmetric P x P per level, or -- FC_GRAM_TRI,
 * the build for frames whose full tables HBM cannot hold for every CU (4K) -- the lower triangle
 * with packed rows, <a, b> with a >= b at TRI(a) + b.  The levels are gram_ls floats apart. */
#ifndef FC_GRAM_TRI
#define FC_GRAM_TRI 0
#endif
#define TRI(a)       ((unsigned) (a) * ((unsigned) (a) + 1u) / 2u)
#define GROW(a, P)   (FC_GRAM_TRI ? TRI(a) : (unsigned) (a) * (unsigned) (P))     /* start of row a in a level */
#define GRAM(F, q)   ((F).gram + (size_t) (q) * (F).gram_ls)
/* the same through the LDS copy of the table base (no descriptor read on the hot path) */
#define PGRAM(sh, q)  ((sh).par.gram + (size_t) (q) * (sh).par.gram_ls)
#define TREE(F, s, l)        ((F).tree[(l) * (F).PA + (s)])
#define INTO(F, s, l, e)     ((F).into[((l) * 6 + (e)) * (F).PA + (s)])
#define WEIGHT(F, s, l, e)   ((F).weight[((l) * 6 + (e)) * (F).PA + (s)])

__device__ __forceinline__ float gram_load(const float *G, int P, int a, int b, int flim);
#define NOFLIM 0x7fffffff        /* every entry is stored both ways (the basis states) */

/* one Gram entry at table level q >= 1 from level q-1 (codec/ip.c:213-257) */
__device__ float gram_entry(const DevFrame &F, int q, int s1, int s2)
{
    const float *G = GRAM(F, q - 1);
    const int P = F.P;
    float ip = 0;
    for (int label = 0; label < 2; label++) {
        int d1, d2;
        float sum;
        int t2 = TREE(F, s2, label);
        if ((d1 = TREE(F, s1, label)) != RANGE_) {
            sum = 0;
            if (t2 != RANGE_) sum = gram_load(G, P, d1, t2, NOFLIM);
            for (int e2 = 0; (d2 = INTO(F, s2, label, e2)) != NOEDGE; e2++)
                sum += WEIGHT(F, s2, label, e2) * gram_load(G, P, d1, d2, NOFLIM);
            ip += sum;
        }
        for (int e1 = 0; (d1 = INTO(F, s1, label, e1)) != NOEDGE; e1++) {
            float w1 = WEIGHT(F, s1, label, e1);
            sum = 0;
            if (t2 != RANGE_) sum = gram_load(G, P, d1, t2, NOFLIM);
            for (int e2 = 0; (d2 = INTO(F, s2, label, e2)) != NOEDGE; e2++)
                sum += WEIGHT(F, s2, label, e2) * gram_load(G, P, d1, d2, NOFLIM);
            ip += w1 * sum;
        }
    }
    return ip;
}

/* level-images_level Gram entry: plain sequential dot (codec/ip.c:297-323) */
__device__ float gram_dot(const DevFrame &F, int s1, int s2)
{
    const int n = 1 << F.images_level;
    float ip = 0;
    for (int k = 0; k < n; k++)
        ip += F.imgT[(size_t) k * F.P + s1] * F.imgT[(size_t) k * F.P + s2];
    return ip;
}

#if FC_VARIANT_BIG
/* the same one level lower (block levels down to 4) */
__device__ float gram_dot4(const DevFrame &F, int s1, int s2)
{
    const int n = 1 << (F.images_level - 1);
    float ip = 0;
    for (int k = 0; k < n; k++)
        ip += F.imgT4[(size_t) k * F.P + s1] * F.imgT4[(size_t) k * F.P + s2];
    return ip;
}
#endif

/* s >= t */
__device__ void gram_store(const DevFrame &F, int q, int s, int t, float v)
{
    float *G = GRAM(F, q);
    G[GROW(s, F.P) + (unsigned) t] = v;
#if !FC_GRAM_TRI
    G[(size_t) t * F.P + s] = v;
#else
    if (t < FC_TRI_HOT && t < s) F.gcol[((size_t) q * FC_TRI_HOT + t) * F.P + s] = v;
#endif
    if (s == t) F.diag[(size_t) q * F.P + s] = v;
}

/*
 *  Symmetric Gram tables without scattered writes.  A new state s writes only its ROW
 *  (entries t <= s, contiguous).  The mirrored entries G[t][s] -- one 4-byte store per
 *  128-byte line when written directly, i.e. 32x write amplification in HBM -- are produced
 *  later in blocks of GRAM_FB states by gram_flush(): 128-byte segments, full lines.
 *  Invariant: with flim = sh.flim, G[a][b] is stored if b <= a or max(a, b) < flim; an entry
 *  outside that set is read through its mirror image.
 */
#define GRAM_FB 32

/* position of <a, b> in a level.  Triangle: whichever of the two is larger names the row. */
__device__ __forceinline__ unsigned gram_idx(int P, int a, int b, int flim)
{
#if FC_GRAM_TRI
    return a >= b ? TRI(a) + (unsigned) b : TRI(b) + (unsigned) a;
#else
    const bool mirror = b > a && b >= flim;
    return mirror ? (unsigned) b * (unsigned) P + (unsigned) a : (unsigned) a * (unsigned) P + (unsigned) b;
#endif
}
__device__ __forceinline__ float gram_load(const float *G, int P, int a, int b, int flim)
{
    return G[gram_idx(P, a, b, flim)];
}

#if FC_GRAM_TRI
/*
 *  The triangle.  A new state writes its row (entries t <= s, contiguous) and nothing else; the
 *  sweep of a matching-pursuit step reads the chosen state's row up to the diagonal and, for the
 *  candidates behind it, the chosen state's COLUMN -- one 4-byte gather per candidate, a whole
 *  line of HBM traffic each.  Half the memory per frame: at 4K, where the full tables allow slabs
 *  for only half the CUs, that doubles the frames in flight (16.2 -> 24.6 frames/s); at 1080p,
 *  where every CU has its four frames anyway, the gathers cost 28 % (547 -> 392 frames/s) --
 *  which is why the layout is a property of the kernel build and the launcher picks by memory.
 */
__device__ __forceinline__ void gram_flush(const DevFrame &, Sh &, int) { }
#else

__device__ void gram_flush(const DevFrame &__restrict__ F, Sh &__restrict__ sh, int upto)
{
    const int tid = threadIdx.x, P = __builtin_amdgcn_readfirstlane(F.P);
    int flim = sh.flim;
#if FC_SPEC
    if (sh.sl.role > 0) return;      /* a verifier reads what its own states have in their own rows */
#endif
    if (upto - flim < GRAM_FB) return;                      /* uniform */
    __syncthreads();                                        /* the rows are complete */
    while (upto - flim >= GRAM_FB) {
        for (int q = 0; q < F.NL; q++) {
            /* (a level is P x P floats, < 4 GB: 32-bit element offsets behind a scalar base) */
            GLOBAL_AS float *G = uniform_ptr(GRAM(F, q));
            for (int t = tid; t < flim + GRAM_FB; t += B) {
                if (t < flim) {
                    float v[GRAM_FB];
#pragma unroll
                    for (int j = 0; j < GRAM_FB; j++) v[j] = ldg((GLOBAL_AS const float *) G, (unsigned) ((flim + j) * P + t));
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    GLOBAL_AS f4 *dst = (GLOBAL_AS f4 *) ((GLOBAL_AS char *) G + (unsigned) (t * P + flim) * 4u);
#pragma unroll
                    for (int j = 0; j < GRAM_FB / 4; j++) {
                        const f4 w = { v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3] };
                        dst[j] = w;
                    }
                } else {
                    for (int j = t - flim + 1; j < GRAM_FB; j++)
                        stg(G, (unsigned) (t * P + flim + j), ldg((GLOBAL_AS const float *) G, (unsigned) ((flim + j) * P + t)));
                }
            }
        }
        flim += GRAM_FB;
    }
    __syncthreads();
    if (tid == 0) sh.flim = flim;
}
#endif

/* state image element (codec/control.c:205-258): level l >= 1, position i */
__device__ float image_elem(const DevFrame &F, int s, int l, int i)
{
    int half = 1 << (l - 1);
    int label = i >= half;
    int pos = i - label * half;
    int base = half - 1;                 /* address_of_level(l-1) */
    float v = 0;
    int dom;
    if ((dom = TREE(F, s, label)) != RANGE_) v = F.img[(size_t) dom * F.NI + base + pos];
    for (int e = 0; (dom = INTO(F, s, label, e)) != NOEDGE; e++)
        v += F.img[(size_t) dom * F.NI + base + pos] * WEIGHT(F, s, label, e);
    return v;
}

#if FC_VARIANT_BIG
/* ---- a basis that travels as the memory image of its rows (DevFrame.bx; data/medium.fco, large.fco) ----
 * The edge list of (state, label) starts at entry (2 state + label) * 6 and ends at the first NO_EDGE -- beyond
 * the row's own six entries where the reference's append_edge ran on into the next row (codec/wfalib.c:253-273).
 * Basis states have no tree children.  Their rows in the automaton arrays of the slab stay empty: the table
 * passes below take the basis states' terms from here, in the reference's order of additions. */
struct BxView { int nb; const float *final_d; const int *dtype; const float *w; const int16_t *into; };
__device__ __forceinline__ BxView bx_view(const DevFrame &F)
{
    BxView v;
    const int *b = F.bx;
    v.nb = b[0];
    v.final_d = (const float *) (b + 4); v.dtype = b + 4 + v.nb; v.w = (const float *) (b + 4 + 2 * v.nb);
    v.into = (const int16_t *) (b + 4 + 2 * v.nb + b[1]);
    return v;
}

/* image_elem() of a basis state */
__device__ float image_elem_bx(const DevFrame &F, const BxView &V, int s, int l, int i)
{
    const int half = 1 << (l - 1), label = i >= half, pos = i - label * half, base = half - 1;
    float v = 0;
    int dom;
    for (int e = (s * 2 + label) * 6; (dom = V.into[e]) != NOEDGE; e++)
        v += F.img[(size_t) dom * F.NI + base + pos] * V.w[e];
    return v;
}

/* gram_entry() of two basis states */
__device__ float gram_entry_bx(const DevFrame &F, const BxView &V, int q, int s1, int s2)
{
    const float *G = GRAM(F, q - 1);
    const int P = F.P;
    float ip = 0;
    for (int label = 0; label < 2; label++) {
        int d1, d2;
        for (int e1 = (s1 * 2 + label) * 6; (d1 = V.into[e1]) != NOEDGE; e1++) {
            float sum = 0;
            for (int e2 = (s2 * 2 + label) * 6; (d2 = V.into[e2]) != NOEDGE; e2++)
                sum += V.w[e2] * gram_load(G, P, d1, d2, NOFLIM);
            ip += V.w[e1] * sum;
        }
    }
    return ip;
}
#endif

/* ------------------------------------------------------------------ parallel ops */

/* <sub-block, state> tables in use: the block's, or -- big build, while the residual of a
 * predicted range is searched (codec/prediction.c:302-309,443-450) -- the second set */
#if FC_VARIANT_BIG
#define ACT_IPIS(F, sh) ((sh).par.ipis)
#define ACT_D5(F, sh)   ((sh).par.d5)
#define ACT_D4(F, sh)   ((sh).par.d4)
#elif FC_SPEC               /* the block's tables live in one buffer of the frame's ring, or in the workgroup's own */
#define ACT_IPIS(F, sh) ((sh).par.ipis)
#define ACT_D5(F, sh)   ((sh).par.d5)
#define ACT_D4(F, sh)   ((F).d4)
#else
#define ACT_IPIS(F, sh) ((F).ipis)
#define ACT_D5(F, sh)   ((F).d5)
#define ACT_D4(F, sh)   ((F).d4)
#endif

/* states that can own tables: chroma states are all auxiliary (codec/subdivide.c:433-436) */
__device__ __forceinline__ int table_states(const Sh &sh) { return sh.band ? sh.ystates : sh.states; }

/* the automaton arrays of a frame behind uniform global pointers */
struct AutoTabs {
    GLOBAL_AS const int16_t *tree, *into;
    GLOBAL_AS const float   *weight;
    GLOBAL_AS const uint8_t *domain_type;
    int PA;
};

__device__ __forceinline__ void auto_tabs(const DevFrame &F, AutoTabs &t)
{
    t.tree = uniform_ptr((const int16_t *) F.tree); t.into = uniform_ptr((const int16_t *) F.into);
    t.weight = uniform_ptr((const float *) F.weight);
    t.domain_type = uniform_ptr((const uint8_t *) F.domain_type);
    t.PA = __builtin_amdgcn_readfirstlane(F.PA);
}

/* the automaton rows of one state as they come out of memory: all edge slots are read
 * unconditionally (independent, coalesced loads; what lies behind the terminator is ignored) */
template <int E> struct EdgeRowsT {
    int   tree[2], rd[2][E];
    float rw[2][E];
    int   dt;
};
typedef EdgeRowsT<FC_MAXE> EdgeRows;

template <int E> __device__ __forceinline__ void load_edge_rows(const AutoTabs &T, int s, EdgeRowsT<E> &r)
{
    unsigned us = (unsigned) s;
    /* opaque to loop strength reduction: otherwise every array gets its own 64-bit pointer
     * induction variable in VGPRs (46 registers) instead of scalar base + this one offset */
    asm volatile("" : "+v"(us));
    r.dt = ldg(T.domain_type, us);
#pragma unroll
    for (int l = 0; l < 2; l++) {
        /* one scalar base per array, the row offset goes into the lane offset */
        r.tree[l] = ldg(T.tree, us + (unsigned) (l * T.PA));
#pragma unroll
        for (int e = 0; e < E; e++) {
            r.rd[l][e] = ldg(T.into, us + (unsigned) ((l * 6 + e) * T.PA));
            r.rw[l][e] = ldg(T.weight, us + (unsigned) ((l * 6 + e) * T.PA));
        }
    }
}

/* <sub-block, state> tables for states [from, states) and the heap subtree under `image`
 * (codec/ip.c:72-154).  Per slot the additions run label 0 {child, edges}, label 1 {...}
 * onto zero, which is the reference's accumulation order onto its zeroed slots. */
/* E: edge slots per label read and summed (the build's E; 3 in the big build when neither the options nor
 * the basis allow more: dead slots still cost a gather per slot and state) */
template <int E> __device__ __noinline__ void op_ipis_t(const DevFrame &__restrict__ F, Sh &__restrict__ sh, int image, int address, int level, int from, int lv_first)
{
    const int tid = threadIdx.x, il = F.images_level;
    const int P = __builtin_amdgcn_readfirstlane(F.P), states = __builtin_amdgcn_readfirstlane(table_states(sh));
    image = __builtin_amdgcn_readfirstlane(image); address = __builtin_amdgcn_readfirstlane(address);
    level = __builtin_amdgcn_readfirstlane(level); from = __builtin_amdgcn_readfirstlane(from);
    GLOBAL_AS float *const ipis = uniform_ptr(ACT_IPIS(F, sh));
    GLOBAL_AS const float *const d5 = uniform_ptr((const float *) ACT_D5(F, sh));
    AutoTabs T;
    auto_tabs(F, T);
    lv_first = __builtin_amdgcn_readfirstlane(lv_first);
    for (int lv = lv_first > il + 1 ? lv_first : il + 1; lv <= level; lv++) {
        int delta = level - lv;
        int cnt = 1 << delta;
        int slot0 = ((image + 1) << delta) - 1;
        int adr0 = address << delta;
#if FC_D5T
        const bool first = lv == il + 1;
        const unsigned NAu = (unsigned) __builtin_amdgcn_readfirstlane(F.NA), NAh = NAu >> 1;
        const bool vec4 = first && cnt >= 4 && (NAh & 3u) == 0;          /* adr0 is a multiple of cnt */
        GLOBAL_AS const float *src0 = first ? d5 : ipis + (size_t) (slot0 * 2 + 1) * P;
#else
        GLOBAL_AS const float *src0 = (lv == il + 1) ? d5 + (size_t) (adr0 * 2) * P
                                                     : ipis + (size_t) (slot0 * 2 + 1) * P;
#endif
        int s = from + tid;
#if FC_VARIANT_BIG
        if (F.bx) {                        /* basis states with their terms in DevFrame.bx (see bx_view) */
            const BxView V = bx_view(F);
            for (int i = tid; i < V.nb * cnt; i += B) {
                const int bs = i / cnt, j = i - bs * cnt;
                if (bs < from || !V.dtype[bs]) continue;
                float acc = 0;
                for (int l = 0; l < 2; l++) {
                    int dom;
                    for (int e = (bs * 2 + l) * 6; (dom = V.into[e]) != NOEDGE; e++)
                        acc += V.w[e] * ldg(src0, (unsigned) dom + (unsigned) ((j * 2 + l) * P));
                }
                stg(ipis, (unsigned) bs + (unsigned) ((slot0 + j) * P), acc);
            }
            if (from < V.nb) s = V.nb + tid;
        }
#endif
        /* the rows of the NEXT state of this lane are requested before the gathers of the
         * current one are waited for (one memory round trip per state instead of two) */
        EdgeRowsT<E> nx;
        if (s < states) load_edge_rows(T, s, nx);
        for (; s < states; s += B) {
            const EdgeRowsT<E> cur = nx;
            if (s + B < states) load_edge_rows(T, s + B, nx);
            const bool tabled = cur.dt && !DEAD(sh, s);
            if (!tabled) continue;
            /* term list of the state: per label the tree child (weight 1, added plain) and
             * the edges in stored order.  Fixed-trip, predicated loops so that all gathers
             * of a group of slots are in flight together (the chain is latency bound). */
            int   idx[2][E + 1];
            float wt[2][E + 1];
            unsigned msk[2];
#pragma unroll
            for (int l = 0; l < 2; l++) {
                int k = cur.tree[l];
                msk[l] = k != RANGE_ ? 1u : 0u;
                idx[l][0] = k != RANGE_ ? k : 0;
                wt[l][0] = 1.0f;
                bool live = true;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    live = live && cur.rd[l][e] != NOEDGE;
                    idx[l][e + 1] = live ? cur.rd[l][e] : 0;
                    wt[l][e + 1] = live ? cur.rw[l][e] : 0.0f;
                    msk[l] |= live ? (2u << e) : 0u;
                }
            }
            constexpr int JG = 4;          /* slots per group: 4 x 2 x (E + 1) gathers in flight per lane (8: -5 %) */
            for (int j0 = 0; j0 < cnt; j0 += JG) {
                float v[JG][2][E + 1];
#if FC_D5T
                if (vec4) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int l = 0; l < 2; l++)
#pragma unroll
                        for (int i = 0; i <= E; i++) {
                            const unsigned o = (unsigned) idx[l][i] * NAu + (unsigned) l * NAh + (unsigned) (adr0 + j0);
                            const f4 q = *(GLOBAL_AS const f4 *) (src0 + o);
                            v[0][l][i] = q.x; v[1][l][i] = q.y; v[2][l][i] = q.z; v[3][l][i] = q.w;
                        }
                } else if (first) {
#pragma unroll
                    for (int jj = 0; jj < JG; jj++)
#pragma unroll
                        for (int l = 0; l < 2; l++)
#pragma unroll
                            for (int i = 0; i <= E; i++) {
                                const int jc = j0 + jj < cnt ? j0 + jj : cnt - 1;
                                v[jj][l][i] = ldg(src0, (unsigned) idx[l][i] * NAu + (unsigned) l * NAh + (unsigned) (adr0 + jc));
                            }
                } else
#endif
#pragma unroll
                for (int jj = 0; jj < JG; jj++)
#pragma unroll
                    for (int l = 0; l < 2; l++)
#pragma unroll
                        for (int i = 0; i <= E; i++) {
                            /* UNCONDITIONAL loads (dead terms read element 0 of the row, slots
                             * past the end re-read the last one): a conditional load becomes a
                             * branch with its own s_waitcnt and the gathers would run one
                             * after the other instead of all in flight */
                            const int jc = j0 + jj < cnt ? j0 + jj : cnt - 1;
                            v[jj][l][i] = ldg(src0, (unsigned) idx[l][i] + (unsigned) ((jc * 2 + l) * P));
                        }
#pragma unroll
                for (int jj = 0; jj < JG; jj++) {
                    if (j0 + jj >= cnt) break;
                    float acc = 0;
#pragma unroll
                    for (int l = 0; l < 2; l++) {
                        if (msk[l] & 1u) acc += v[jj][l][0];
#pragma unroll
                        for (int i = 1; i <= E; i++)
                            if ((msk[l] >> i) & 1u) acc += wt[l][i] * v[jj][l][i];
                    }
                    stg(ipis, (unsigned) s + (unsigned) ((slot0 + j0 + jj) * P), acc);
                }
            }
        }
        __syncthreads();
    }
}

/* lv_first: the levels below it are in the tables already (a cooperative build, FcCoop) */
__device__ __forceinline__ void op_ipis(const DevFrame &__restrict__ F, Sh &__restrict__ sh, int image, int address, int level, int from,
                                        int lv_first = 0)
{
#if FC_VARIANT_BIG
    if (F.maxe_live <= 3) { op_ipis_t<3>(F, sh, image, address, level, from, lv_first); return; }
#endif
    op_ipis_t<FC_MAXE>(F, sh, image, address, level, from, lv_first);
}

/* level-images_level dots of the current pixel block with state images (codec/ip.c:268-295) */
/* na / n4: number of level-images_level and level-(images_level - 1) sub-blocks of the block in
 * sh.pixels (NA and 2 NA for a whole block; fewer for the residual of a predicted range) */
/* abase / pxoff: the addresses start at abase (level images_level; 2 abase one level below) and their pixels at
 * sh.pixels + pxoff -- one subtree of a block (cooperative build, FcCoop) */
__device__ void op_d5(const DevFrame &__restrict__ F, Sh &__restrict__ sh, int from, int to, int na, int n4, int abase = 0, int pxoff = 0)
{
    const int tid = threadIdx.x, P = __builtin_amdgcn_readfirstlane(F.P);
    /* tables behind scalar bases: global_load / global_store with a 32-bit lane offset (through the generic
     * frame reference of this out-of-line code they would be flat_ instructions) */
    GLOBAL_AS float *const D5 = uniform_ptr(ACT_D5(F, sh));
    GLOBAL_AS const float *const imgT = uniform_ptr((const float *) F.imgT);
    GLOBAL_AS const uint8_t *const dtype = uniform_ptr((const uint8_t *) F.domain_type);
    for (int s = from + tid; s < to; s += B) {
        if (DEAD(sh, s) || !ldg(dtype, (unsigned) s)) continue;
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = ldg(imgT, (unsigned) (k * P + s));
        /* two addresses per step: packed fp32 multiply and add (v_pk_mul_f32 / v_pk_add_f32,
         * each half rounded like the scalar op; no fused multiply-add), pixels read in pairs */
        typedef float f2 __attribute__((ext_vector_type(2)));
#if FC_D5T
        const unsigned NAu = (unsigned) __builtin_amdgcn_readfirstlane(F.NA), NAh = NAu >> 1;
        int a = 0;
        if ((NAh & 3u) == 0)
            /* eight addresses per step: the even and the odd ones are four consecutive floats each */
            for (; a + 8 <= na; a += 8) {
                f2 ip[4] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    f2 vv = { v[k], v[k] };
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        f2 px = { sh.pixels[(a + 2 * u) * 32 + k], sh.pixels[(a + 2 * u) * 32 + 32 + k] };
                        ip[u] = ip[u] + px * vv;
                    }
                }
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 ev = { ip[0].x, ip[1].x, ip[2].x, ip[3].x }, od = { ip[0].y, ip[1].y, ip[2].y, ip[3].y };
                const unsigned o = (unsigned) s * NAu + ((unsigned) a >> 1);
                *(GLOBAL_AS f4 *) (D5 + o) = ev;
                *(GLOBAL_AS f4 *) (D5 + o + NAh) = od;
            }
        for (; a < na; a += 2) {
            f2 ip = { 0.0f, 0.0f };
#pragma unroll
            for (int k = 0; k < 32; k++) {
                f2 px = { sh.pixels[a * 32 + k], sh.pixels[a * 32 + 32 + k] };
                f2 vv = { v[k], v[k] };
                ip = ip + px * vv;
            }
            stg(D5, D5_AT(P, NAu, a, s), ip.x);
            if (a + 1 < na) stg(D5, D5_AT(P, NAu, a + 1, s), ip.y);
        }
#else
        for (int a = 0; a < na; a += 2) {
            f2 ip = { 0.0f, 0.0f };
#pragma unroll
            for (int k = 0; k < 32; k++) {
                f2 px = { sh.pixels[pxoff + a * 32 + k], sh.pixels[pxoff + a * 32 + 32 + k] };
                f2 vv = { v[k], v[k] };
                ip = ip + px * vv;
            }
            stg(D5, (unsigned) ((abase + a) * P + s), ip.x);
            if (a + 1 < na) stg(D5, (unsigned) ((abase + a + 1) * P + s), ip.y);
        }
#endif
#if FC_VARIANT_BIG
        if (F.gl0 < F.images_level) {
            float *const D4 = ACT_D4(F, sh);
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = F.imgT4[(size_t) k * P + s];
            for (int a = 0; a < n4; a++) {
                float ip = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) ip += sh.pixels[pxoff + a * 16 + k] * v[k];
                D4[(size_t) (2 * abase + a) * P + s] = ip;
            }
        }
#endif
    }
}

#if FC_VARIANT_BIG
/* ---- several workgroups for the table passes of one frame (frame_coder.h, FcCoop) ---- */
__device__ __forceinline__ float *coop_pixels(FcCoop *c) { return (float *) ((char *) c + FC_COOP_HDR); }

/* the subtrees member, member + W, ... of depth D below a block of 2^level pixels in sh.pixels: level-5 dots and
 * the level recursion up to the subtree's own level, for the states [from, table_states) */
__device__ void coop_share(const DevFrame &__restrict__ F, Sh &__restrict__ sh, int level, int from, unsigned member, unsigned W, int D)
{
    const int il = F.images_level, sub = level - D, nasub = 1 << (sub - il);
    for (int j = (int) member; j < (1 << D); j += (int) W)
        op_d5(F, sh, from, table_states(sh), nasub, 2 * nasub, j * nasub, j << sub);
    __syncthreads();
    for (int j = (int) member; j < (1 << D); j += (int) W)
        op_ipis(F, sh, (1 << D) - 1 + j, j, sub, from);
}

/* The frame's workgroup: hand the block in sh.pixels to the helpers.  Returns the depth D (the caller goes on
 * with coop_finish and adds the levels above level - D), or 0: the block is built the ordinary way. */
__device__ int coop_publish(DevFrame &__restrict__ F, Sh &__restrict__ sh, int level, int from)
{
    const int tid = threadIdx.x, il = F.images_level;
    const unsigned W = sh.coopW;
    FcCoop *c = F.coop;
    if (W < 2 || !c) return 0;
    const int D = sh.coopD;
    if (level - il < D + sh.coop_minsub) return 0;          /* small subtrees: not worth a hand-off */
    float *gp = coop_pixels(c);
    for (int i = tid; i < (1 << level); i += B) gp[i] = sh.pixels[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        c->level = level; c->from = from; c->to = table_states(sh);
        c->ipis = sh.par.ipis; c->d5 = sh.par.d5; c->d4 = sh.par.d4;
        /* everything the helpers read: the pixels, the descriptor, the rows of the states appended so far */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh.coop_seq++;
        __hip_atomic_store(&c->seq, sh.coop_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return D;
}

/* ... after the caller's own LDS-only work (the norms of the block): the own share, then the helpers' */
__device__ void coop_finish(DevFrame &__restrict__ F, Sh &__restrict__ sh, int level, int from, int D)
{
    const int tid = threadIdx.x;
    const unsigned W = sh.coopW;
    FcCoop *c = F.coop;
    coop_share(F, sh, level, from, 0, W, D);
    __syncthreads();
    if (tid == 0) {
        const unsigned want = sh.coop_seq * (W - 1);
        const unsigned long long t_give_up = wall_clock64() + sh.coop_ticks;
        while (__hip_atomic_load(&c->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (wall_clock64() > t_give_up) { sh.failed = FC_ERR_COOP; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      /* the helpers' rows, not this CU's stale lines */
    }
    __syncthreads();
}

/* workgroups 1 .. W - 1 of a frame: build what the frame's workgroup hands over until it is finished */
__device__ void coop_helper(DevFrame &__restrict__ F, Sh &__restrict__ sh, unsigned member, unsigned W)
{
    const int tid = threadIdx.x;
    FcCoop *c = F.coop;
    unsigned seen = 0;
    if (!c) return;
    for (;;) {
        if (tid == 0) {
            const unsigned long long t_give_up = wall_clock64() + FC_COOP_WAIT_TICKS;
            int go = 0;
            for (;;) {
                if (__hip_atomic_load(&c->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { go = -1; break; }
                if (__hip_atomic_load(&c->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seen) { go = 1; break; }
                if (wall_clock64() > t_give_up) { go = -1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (go == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                sh.a0 = c->level; sh.a1 = c->from; sh.a2 = (int) c->depth;
                sh.band = 0; sh.states = c->to; sh.ystates = c->to;
                sh.par.ipis = c->ipis; sh.par.d5 = c->d5; sh.par.d4 = c->d4;
            }
            sh.op = go;
        }
        __syncthreads();
        if (sh.op < 0) return;
        const int level = sh.a0, from = sh.a1, D = sh.a2, sub = level - D;
        const float *gp = coop_pixels(c);
        for (int j = (int) member; j < (1 << D); j += (int) W)
            for (int i = tid; i < (1 << sub); i += B) sh.pixels[(j << sub) + i] = gp[(j << sub) + i];
        __syncthreads();
        coop_share(F, sh, level, from, member, W, D);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&c->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        seen++;
        __syncthreads();                /* sh.op is rewritten by lane 0 at the top */
    }
}
#endif

#if !FC_SPEC
/* ------------------------------------------------------------------ chroma bands: tables for the states that matter
 *
 * init_range builds the <sub-block, state> entries of a block for EVERY state with tables (codec/subdivide.c:612-644,
 * codec/ip.c:72-154).  In a chroma band nobody appends a state with tables (chroma states are auxiliary,
 * codec/subdivide.c:433-436) and nothing is predicted: the entries are read by the matching pursuit of the block's
 * ranges alone -- for the <= chroma_max states of the chroma list plus the co-located luminance state of the range
 * (rle_generate, codec/domain-pool.c:707-735) -- and, building those, for the states they refer to one level
 * down, and so on for (lc_max - images_level) levels.  That closure is 60 .. 150 of the 1200 .. 2700 luminance states
 * of a 720p / 1080p frame (measured with the oracle), the same values as the full tables hold for them, and the
 * tables of a chroma block were 60 % of a colour frame.
 *
 * F.hits[s] (free once the chroma list is chosen): low half = levels at which the entries of s are needed for the
 * chroma list's sake (static, op_chroma_pool), high half = the same for the block at hand (+ the luminance states
 * of the block's subtree).  Bit k <-> level images_level + k; bit 0 = the level-images_level dots (d5, d4). */
__device__ __forceinline__ int hits_ld(const DevFrame &F, int s) { return __hip_atomic_load(&F.hits[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* one step of the closure on half `sft` (0 / 16): what needs level k of s needs level k - 1 of the tree children
 * and edge targets of s */
__device__ void chroma_need_step(const DevFrame &__restrict__ F, Sh &__restrict__ sh, int sft)
{
    const int n = sh.ystates;
    for (int s = threadIdx.x; s < n; s += B) {
        const int m = ((hits_ld(F, s) >> sft) & 0xffff) >> 1;
        if (!m) continue;
        for (int l = 0; l < 2; l++) {
            int d = TREE(F, s, l);
            if (d != RANGE_) __hip_atomic_fetch_or(&F.hits[d], m << sft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int e = 0; (d = INTO(F, s, l, e)) != NOEDGE; e++)
                __hip_atomic_fetch_or(&F.hits[d], m << sft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

/* end of op_chroma_pool: the part of the closure that is the same for every block (the chroma list) */
__device__ void chroma_need_static(const DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    const int tid = threadIdx.x, n = sh.ystates, NB = F.lc_max - F.images_level;
    __syncthreads();
    for (int s = tid; s < n; s += B) __hip_atomic_store(&F.hits[s], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int i = tid; i < (int) sh.pool.n; i += B)
        __hip_atomic_store(&F.hits[F.pool_states[i]], (1 << (NB + 1)) - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int it = 0; it < NB; it++) chroma_need_step(F, sh, 0);
}

/* per block: + the luminance states of the block's subtree (the co-located states of its ranges), compacted into sh.cl.
 * false: more states than sh.cl holds -- the caller builds the full tables */
__device__ bool chroma_need_block(const DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    const int tid = threadIdx.x, n = sh.ystates, il = F.images_level, NB = F.lc_max - il;
    const int y = sh.st[sh.sp].y_state;
    for (int s = tid; s < n; s += B) {
        const int v = hits_ld(F, s) & 0xffff;
        __hip_atomic_store(&F.hits[s], v | (v << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) sh.cln = 0;
    __syncthreads();
    if (y != RANGE_) {
        /* nodes of the block's subtree, heap order.  The block is of level F.lc_max (op_init_range builds no other)
         * and its ranges go down to the band's running minimum level, which the ratchet keeps at or below lc_max
         * (band_advance); clamped all the same: a negative shift count would be undefined */
        const int dlv = F.lc_max > sh.lc_min ? F.lc_max - sh.lc_min : 0;
        const int nheap = (2 << dlv) - 1;
        for (int h = tid; h < nheap; h += B) {
            const int depth = 31 - __clz(h + 1);
            int node = y;
            for (int b = depth - 1; b >= 0 && node != RANGE_; b--) node = TREE(F, node, ((h + 1) >> b) & 1);
            if (node == RANGE_) continue;
            const int lv = F.lc_max - depth, k = lv > il ? lv - il : 0;
            __hip_atomic_fetch_or(&F.hits[node], 1 << (16 + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        for (int it = 0; it < NB; it++) chroma_need_step(F, sh, 16);
    }
    for (int s = tid; s < n; s += B)
        if ((hits_ld(F, s) >> 16) && F.domain_type[s]) {
            const int i = atomicAdd(&sh.cln, 1);
            if (i < FC_CLMAX) sh.cl[i] = (short) s;
        }
    __syncthreads();
    const int cap = F.chroma_cl_cap > 0 && F.chroma_cl_cap < FC_CLMAX ? F.chroma_cl_cap : FC_CLMAX;
    return sh.cln <= cap;
}

/* op_d5 for the states of sh.cl that need their level-images_level dots: (state, group of eight addresses) items */
__device__ void op_d5_sparse(const DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    const int tid = threadIdx.x, P = F.P, NA = F.NA, ngrp = (NA + 7) / 8;
    float *const D5 = ACT_D5(F, sh);
    for (int it = tid; it < sh.cln * ngrp; it += B) {
        const int s = sh.cl[it / ngrp], a0 = (it % ngrp) * 8;
        if (!((hits_ld(F, s) >> 16) & 1)) continue;
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = F.imgT[(size_t) k * P + s];
        for (int a = a0; a < a0 + 8 && a < NA; a++) {
            float ip = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) ip += sh.pixels[a * 32 + k] * v[k];      /* codec/ip.c:268-295, sequential */
            D5[D5_AT(P, NA, a, s)] = ip;
        }
    }
#if FC_VARIANT_BIG
    if (F.gl0 < F.images_level) {
        float *const D4 = ACT_D4(F, sh);
        for (int i = tid; i < sh.cln; i += B) {
            const int s = sh.cl[i];
            if (!((hits_ld(F, s) >> 16) & 1)) continue;
            float v4[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v4[k] = F.imgT4[(size_t) k * P + s];
            for (int a = 0; a < 2 * NA; a++) {
                float ip = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) ip += sh.pixels[a * 16 + k] * v4[k];
                D4[(size_t) a * P + s] = ip;
            }
        }
    }
#endif
}

/* op_ipis for the (state, level) pairs of the closure: per level the items (state, slot); the additions of an entry
 * in the reference's order -- label 0 {tree child, edges}, label 1 {...} onto zero (codec/ip.c:104-146) */
__device__ void op_ipis_sparse(const DevFrame &__restrict__ F, Sh &__restrict__ sh, int level)
{
    const int tid = threadIdx.x, il = F.images_level, P = F.P;
    float *const ipis = ACT_IPIS(F, sh);
    const float *const d5 = ACT_D5(F, sh);
    for (int lv = il + 1; lv <= level; lv++) {
        const int delta = level - lv, cnt = 1 << delta, slot0 = cnt - 1, k = lv - il;
        const bool first = lv == il + 1;
        for (int it = tid; it < sh.cln * cnt; it += B) {
            const int s = sh.cl[it >> delta], j = it & (cnt - 1);
            if (!((hits_ld(F, s) >> (16 + k)) & 1)) continue;
            float acc = 0;
            for (int l = 0; l < 2; l++) {
                int d = TREE(F, s, l);
                /* the entry of state d one level down: sub-block 2 j + l of the level below */
#if FC_D5T
#define SRC(d) (first ? d5[D5_AT(P, F.NA, j * 2 + l, (d))] : ipis[(size_t) ((slot0 * 2 + 1) + j * 2 + l) * P + (d)])
#else
#define SRC(d) (first ? d5[(size_t) (j * 2 + l) * P + (d)] : ipis[(size_t) ((slot0 * 2 + 1) + j * 2 + l) * P + (d)])
#endif
                if (d != RANGE_) acc += SRC(d);
                for (int e = 0; (d = INTO(F, s, l, e)) != NOEDGE; e++) acc += WEIGHT(F, s, l, e) * SRC(d);
#undef SRC
            }
            ipis[(size_t) (slot0 + j) * P + s] = acc;
        }
        __syncthreads();
    }
}
#endif

/* codec/subdivide.c:504-541,612-644 */
/* from: the entries of the states below it are in the tables already (FC_SPEC: a table worker has
 * computed them ahead of the chain); otherwise 0 */
__device__ __noinline__ void op_init_range(DevFrame &__restrict__ F, Sh &__restrict__ sh, int x0, int y0, int from)
{
    const int tid = threadIdx.x;
    const int level = F.lc_max, npx = 1 << level;
    const int16_t *plane = F.pix16 + (size_t) sh.band * F.plane;
#if FC_VARIANT_BIG
    if (F.frame_type && sh.band) plane = F.pix_chroma + (size_t) (sh.band - 1) * F.plane;
#endif
    {   /* all pixel loads of the lane in flight: unconditional loads at clamped coordinates,
         * the outside of the image is zeroed afterwards (codec/subdivide.c:504-541) */
        constexpr int NIT = FC_PIXELS / B;
        GLOBAL_AS const int16_t *const gplane = uniform_ptr(plane);      /* a plane is < 2^31 pixels */
        const int width = F.width, height = F.height;
        int raw[NIT];
        bool inside[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + it * B;
            unsigned xo = 0, yo = 0;
#pragma unroll
            for (int b = 0; b < 13; b++) {
                yo |= ((i >> (2 * b)) & 1u) << b;         /* even bits: rows (mask 0x555555)   */
                xo |= ((i >> (2 * b + 1)) & 1u) << b;     /* odd bits: columns (mask 0xaaaaaa) */
            }
            const int x = x0 + (int) xo, y = y0 + (int) yo;
            inside[it] = i < npx && y < height && x < width;
            const int xc = x < width ? x : width - 1, yc = y < height ? y : height - 1;
            raw[it] = ldg(gplane, (unsigned) (yc * width + xc));
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + it * B;
            if (i < npx) sh.pixels[i] = inside[it] ? (float) (raw[it] / 16) : 0.0f;
        }
    }
    __syncthreads();
#if !FC_SPEC
    /* chroma bands: entries for the states somebody reads only (chroma_need_block) */
    const bool sparse = sh.band && !F.bx && F.chroma_sparse && chroma_need_block(F, sh);
#else
    const bool sparse = false;
#endif
#if FC_VARIANT_BIG
    /* the helpers start on the block while this workgroup sums the norms (LDS only) */
    const int coopD = sparse ? 0 : coop_publish(F, sh, level, from);
#endif
    /* squared norms of every sub-block, sequential as codec/approx.c:388-389 */
    for (int slot = tid; slot < F.NS; slot += B) {
        int depth = 31 - __clz(slot + 1);
        int lv = level - depth, size = 1 << lv;
        int adr = slot + 1 - (1 << depth);
        float nrm = 0;
        const float *px = sh.pixels + adr * size;
        for (int k = 0; k < size; k += 8) {        /* size >= 64: 8 LDS reads in flight, */
            float p0 = px[k], p1 = px[k + 1], p2 = px[k + 2], p3 = px[k + 3];   /* adds stay */
            float p4 = px[k + 4], p5 = px[k + 5], p6 = px[k + 6], p7 = px[k + 7]; /* sequential */
            nrm += p0 * p0; nrm += p1 * p1; nrm += p2 * p2; nrm += p3 * p3;
            nrm += p4 * p4; nrm += p5 * p5; nrm += p6 * p6; nrm += p7 * p7;
        }
        sh.norms[slot] = nrm;              /* LDS: read by lane 0 at the start of every search */
    }
#ifdef FC_SERIAL_PROFILE
    unsigned long long tp0 = wall_clock64();
#endif
#if !FC_SPEC
    if (sparse) {
        op_d5_sparse(F, sh);
        __syncthreads();
        op_ipis_sparse(F, sh, level);
        if (tid == 0) {          /* what was built, not what the reference builds: cln states */
            sh.cnt.bytes_img += (unsigned long long) sh.cln * (4ull * 32 + 4ull * F.NS) + 4ull * npx;
            sh.cnt.n_blocks++;
        }
        return;
    }
#endif
#if FC_VARIANT_BIG
    if (coopD) {
        coop_finish(F, sh, level, from, coopD);
        op_ipis(F, sh, 0, 0, level, from, level - coopD + 1);
    } else {
#endif
    op_d5(F, sh, from, table_states(sh), F.NA, 2 * F.NA);
    __syncthreads();
#ifdef FC_SERIAL_PROFILE
    if (tid == 0) { unsigned long long t = wall_clock64(); sh.tk_init[0] += t - tp0; tp0 = t; }
#endif
    op_ipis(F, sh, 0, 0, level, from);
#if FC_VARIANT_BIG
    }
#endif
#ifdef FC_SERIAL_PROFILE
    if (tid == 0) sh.tk_init[1] += wall_clock64() - tp0;
#endif
    if (tid == 0) {
        sh.cnt.bytes_img += (unsigned long long) table_states(sh) * (4ull * 32 + 4ull * F.NS) + 4ull * npx;
        sh.cnt.n_blocks++;
    }
}

#if FC_VARIANT_BIG
__device__ void pred_save_tables(DevFrame &__restrict__ F, Sh &__restrict__ sh, int s);
__device__ void subtract_mc_dev(DevFrame &__restrict__ F, Sh &__restrict__ sh);
#endif

/* Gram row of the new state s at every table level -- the entries t with (t / B) mod parts == part (all of them:
 * part 0 of 1); level q needs level q-1 of states < s.  The term lists of s are in sh.gs_*.  Out of op_append so that
 * the append helpers of a speculating frame (FcSpecCtl.app_*) run the same code on their shares. */
__device__ __forceinline__ void append_row_part(DevFrame &__restrict__ F, Sh &__restrict__ sh, int s, int part, int parts)
{
    const int tid = threadIdx.x, P = F.P;
#if FC_VARIANT_BIG
    const int il = F.images_level;
#endif
    {
        const int flim = __builtin_amdgcn_readfirstlane(sh.flim);
        const int Pu = __builtin_amdgcn_readfirstlane(P);
        s = __builtin_amdgcn_readfirstlane(s);
        const unsigned LS = (unsigned) __builtin_amdgcn_readfirstlane((int) F.gram_ls);   /* floats per table level */
        const unsigned rs = GROW(s, Pu);               /* start of the new state's row in a level */
        GLOBAL_AS float *const gram = uniform_ptr(F.gram);
        GLOBAL_AS float *const diag = uniform_ptr(F.diag);
        AutoTabs T;
        auto_tabs(F, T);
        /* level-images_level image of s (the same for every lane): 32 loads in flight once;
         * per t the other 32.  A `for (k < 1 << images_level)` loop is not unrolled by the
         * compiler and would wait for every single load. */
        GLOBAL_AS const float *imgT = uniform_ptr((const float *) F.imgT);
        float vs[32];
#pragma unroll
        for (int k = 0; k < 32; k++) vs[k] = ldg(imgT, (unsigned) (k * Pu + s));
        for (int t = tid + part * B; t <= s; t += B * parts) {
            EdgeRows rows;
            load_edge_rows(T, t, rows);
            float vt[32];
#pragma unroll
            for (int k = 0; k < 32; k++) vt[k] = ldg(imgT, (unsigned) (k * Pu + t));
            if (!rows.dt || DEAD(sh, t)) continue;
            /* term lists of t in registers (fixed slots: 0 = tree child, 1.. = edges), loaded
             * once and reused by every table level */
            int   i2[2][FC_MAXE + 1];
            float w2[2][FC_MAXE + 1];
            unsigned m2[2];
#pragma unroll
            for (int l = 0; l < 2; l++) {
                int k = rows.tree[l];
                m2[l] = k != RANGE_ ? 1u : 0u;
                i2[l][0] = k != RANGE_ ? k : 0;
                w2[l][0] = 1.0f;
                bool live = true;
#pragma unroll
                for (int e = 0; e < FC_MAXE; e++) {
                    live = live && rows.rd[l][e] != NOEDGE;
                    i2[l][e + 1] = live ? rows.rd[l][e] : 0;
                    w2[l][e + 1] = live ? rows.rw[l][e] : 0.0f;
                    m2[l] |= live ? (2u << e) : 0u;
                }
            }
            /* only the row of s is written here: see gram_flush() */
            int q1 = 1;
#if FC_VARIANT_BIG
            if (F.gl0 < il) {                      /* levels <= images_level: direct dots */
                GLOBAL_AS const float *imgT4 = uniform_ptr((const float *) F.imgT4);
                float a4[16], b4[16], v4 = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) { a4[k] = ldg(imgT4, (unsigned) (k * Pu + s)); b4[k] = ldg(imgT4, (unsigned) (k * Pu + t)); }
#pragma unroll
                for (int k = 0; k < 16; k++) v4 += a4[k] * b4[k];
                stg(gram, rs + (unsigned) t, v4);
                if (s == t) stg(diag, (unsigned) s, v4);
                q1 = 2;
            }
#endif
            {
                float v0 = 0;                         /* codec/ip.c:297-323, sequential */
#pragma unroll
                for (int k = 0; k < 32; k++) v0 += vs[k] * vt[k];
                stg(gram, (unsigned) (q1 - 1) * LS + rs + (unsigned) t, v0);
                if (s == t) stg(diag, (unsigned) ((q1 - 1) * Pu + s), v0);
#if FC_GRAM_TRI
                if (t < FC_TRI_HOT && t < s) F.gcol[((size_t) (q1 - 1) * FC_TRI_HOT + t) * Pu + s] = v0;
#endif
            }
            for (int q = q1; q < F.NL; q++) {
#if FC_VARIANT_BIG
                if (F.bx && t < F.basis_states) {      /* the terms of a basis state: DevFrame.bx */
                    const BxView V = bx_view(F);
                    GLOBAL_AS const float *Gb = gram + (size_t) (q - 1) * LS;
                    float ipb = 0;
                    for (int l = 0; l < 2; l++) {
                        const int na = sh.gs_n[l], ca = sh.gs_c[l];
                        for (int a = 0; a < na; a++) {
                            const int A = sh.gs_idx[l][a];
                            float sum = 0;
                            int d2;
                            for (int e2 = (t * 2 + l) * 6; (d2 = V.into[e2]) != NOEDGE; e2++)
                                sum += V.w[e2] * ldg(Gb, gram_idx(Pu, A, d2, flim));
                            if (a == 0 && ca) ipb += sum;
                            else ipb += sh.gs_w[l][a] * sum;
                        }
                    }
                    stg(gram, (unsigned) q * LS + rs + (unsigned) t, ipb);
                    continue;
                }
#endif
                /* codec/ip.c:213-257: ip = sum_label sum_{a in terms(s)} [w_a *] ( sum_{b in
                 * terms(t)} [w_b *] G_{q-1}[a][b] ); a tree child enters without a multiply.
                 * All gathers of a label (terms(s) x 6 slots of t) are issued before the first
                 * is used; dead term slots of t read a valid dummy entry (no per-lane branch). */
                GLOBAL_AS const float *G = gram + (size_t) (q - 1) * LS;
                float ip = 0;
                float g[2][FC_MAXE + 1][FC_MAXE + 1];
#pragma unroll
                for (int l = 0; l < 2; l++) {                      /* gathers of both labels */
                    const int na = __builtin_amdgcn_readfirstlane(sh.gs_n[l]);
#pragma unroll
                    for (int a = 0; a <= FC_MAXE; a++) {
                        if (a >= na) break;                            /* uniform */
                        const int A = __builtin_amdgcn_readfirstlane(sh.gs_idx[l][a]);
#pragma unroll
                        for (int b = 0; b <= FC_MAXE; b++) {
                            g[l][a][b] = ldg(G, gram_idx(Pu, A, i2[l][b], flim));
                        }
                    }
                }
#pragma unroll
                for (int l = 0; l < 2; l++) {
                    const int na = __builtin_amdgcn_readfirstlane(sh.gs_n[l]);
                    const int ca = __builtin_amdgcn_readfirstlane(sh.gs_c[l]);
#pragma unroll
                    for (int a = 0; a <= FC_MAXE; a++) {
                        if (a >= na) break;
                        float sum = 0;
                        if (m2[l] & 1u) sum = g[l][a][0];
#pragma unroll
                        for (int b = 1; b <= FC_MAXE; b++)
                            if ((m2[l] >> b) & 1u) sum += w2[l][b] * g[l][a][b];
                        if (a == 0 && ca) ip += sum;
                        else ip += sh.gs_w[l][a] * sum;
                    }
                }
                stg(gram, (unsigned) q * LS + rs + (unsigned) t, ip);
                if (s == t) stg(diag, (unsigned) (q * Pu + s), ip);
#if FC_GRAM_TRI
                if (t < FC_TRI_HOT && t < s) F.gcol[((size_t) q * FC_TRI_HOT + t) * Pu + s] = ip;
#endif
            }
        }
    }
}

#if FC_SPEC
/* ... as a call: the shares of a dealt row (chain and append helpers) */
__device__ __noinline__ void append_row_part_ool(DevFrame &__restrict__ F, Sh &__restrict__ sh, int s, int part, int parts)
{
    append_row_part(F, sh, s, part, parts);
}
#endif

/* codec/control.c:48-131 for a non-auxiliary state s whose edges are already stored */
__device__ __noinline__ void op_append(DevFrame &__restrict__ F, Sh &__restrict__ sh, int s)
{
    const int tid = threadIdx.x, il = F.images_level, P = F.P;
#if FC_VARIANT_BIG
    pred_save_tables(F, sh, s);         /* residual search: the id may belong to a displaced state */
#endif
    /* term lists of the new state s (slot 0 = tree child with weight 1 if any, then the
     * edges): twelve lanes read one row slot each (one memory round trip instead of a chain of
     * dependent ones), two lanes compact them into LDS; uniform for the whole workgroup */
#if !FC_VARIANT_BIG
    /* default build: store_new_state() has left the term lists in sh.gs_* */
#else
    if (tid < 12) {
        const int l = tid / 6, e = tid % 6;
        sh.gs_raw_idx[l][e] = e == 0 ? (int) TREE(F, s, l) : (int) INTO(F, s, l, e - 1);
        sh.gs_raw_w[l][e] = e == 0 ? 1.0f : WEIGHT(F, s, l, e - 1);
    }
    __syncthreads();
    if (tid < 2) {
        const int l = tid;
        int m = 0;
        sh.gs_c[l] = sh.gs_raw_idx[l][0] != RANGE_;
        if (sh.gs_c[l]) { sh.gs_idx[l][0] = sh.gs_raw_idx[l][0]; sh.gs_w[l][0] = 1.0f; m = 1; }
        for (int e = 1; e <= MAXED && sh.gs_raw_idx[l][e] != NOEDGE; e++) {
            sh.gs_idx[l][m] = sh.gs_raw_idx[l][e]; sh.gs_w[l][m] = sh.gs_raw_w[l][e]; m++;
        }
        sh.gs_n[l] = m;
        for (; m <= MAXED; m++) { sh.gs_idx[l][m] = 0; sh.gs_w[l][m] = 0.0f; }   /* valid dummies */
    }
    __syncthreads();
#endif
    /* images: level 0 is the final distribution (control.c:97); a level l >= 1 element
     * depends on level l-1 of OTHER states only (codec/control.c:205-258) */
    GLOBAL_AS float *const gimg = uniform_ptr(F.img);
    GLOBAL_AS float *const gimgT = uniform_ptr(F.imgT);
    const int NIu = __builtin_amdgcn_readfirstlane(F.NI);
    if (tid == B - 1) stg(gimg, (unsigned) (s * NIu), F.final_d[s]);
    for (int i = tid; i < NIu - 1; i += B) {
        int l = 31 - __clz(i + 2);                      /* offset 2^l - 1 + pos = i + 1 */
        int pos = i + 1 - ((1 << l) - 1);
        const int half = 1 << (l - 1), label = pos >= half;
        const int off = half - 1 + (pos - label * half);
        const int n = sh.gs_n[label];
        float t[FC_MAXE + 1];
#pragma unroll
        for (int a = 0; a <= FC_MAXE; a++)              /* all term images in flight */
            t[a] = ldg((GLOBAL_AS const float *) gimg, (unsigned) (sh.gs_idx[label][a] * NIu + off));      /* dead slots: state 0 */
        float v = 0;
#pragma unroll
        for (int a = 0; a <= FC_MAXE; a++)
            if (a < n) v = (a == 0 && sh.gs_c[label]) ? t[0] : v + t[a] * sh.gs_w[label][a];
        stg(gimg, (unsigned) (s * NIu + i + 1), v);
        if (l == il) stg(gimgT, (unsigned) (pos * P + s), v);
#if FC_VARIANT_BIG
        if (l == il - 1 && F.gl0 < il) F.imgT4[(size_t) pos * P + s] = v;
#endif
    }
    __syncthreads();
    /* Gram row/column of s at every table level; level q needs level q-1 of states < s */
#if FC_SPEC
    {
        /* a long row of the chain of a frame with append helpers: dealt (FcSpecCtl.app_*) */
        FcSpecCtl *const c = sh.sl.ctl;
        const bool deal = sh.sl.role == 0 && sh.sl.on && c && sh.sl.app_H > 0 && !sh.sl.app_off
                          && (unsigned) (s + 1) >= sh.sl.app_min;                      /* uniform */
        if (!deal) append_row_part(F, sh, s, 0, 1);
        else {
            const unsigned H = sh.sl.app_H;
            WAVE_DRAIN();                       /* images of s, its automaton row: in L2 before the row is published */
            __syncthreads();
            if (tid == 0) {
                c->app_s = s; c->app_flim = sh.flim;
                for (int l = 0; l < 2; l++) {
                    c->app_n[l] = sh.gs_n[l]; c->app_c[l] = sh.gs_c[l];
                    for (int e = 0; e <= MAXED; e++) { c->app_idx[l][e] = sh.gs_idx[l][e]; c->app_w[l][e] = sh.gs_w[l][e]; }
                }
                publish_release();
                __hip_atomic_store(&c->app_seq, ++sh.sl.app_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            append_row_part_ool(F, sh, s, 0, (int) H + 1);
            __syncthreads();
            if (tid == 0) {
                const unsigned want = sh.sl.app_seq * H;
                const unsigned wait_ticks = c->app_wait;
                const unsigned long long t0 = wall_clock64();
                int late = 0;
                while (__hip_atomic_load(&c->app_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
                    if (wall_clock64() - t0 > wait_ticks) { late = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                take_acquire();                 /* the helpers' entries, not this CU's stale lines */
                sh.sl.t_app_wait += wall_clock64() - t0;
                sh.sl.n_app_dealt++;
                if (late) {
                    /* helpers that do not answer (not resident: masked CUs, a busy device).  A helper that turns up
                     * later could write a row the chain has re-made since: the frame is given up -- FC_ERR_COOP, the
                     * host searches it again without helpers (core_hip.cpp complete_wave) -- and the helpers are sent home */
                    sh.sl.app_off = 1;
                    sh.failed = FC_ERR_COOP;
                    __hip_atomic_store(&c->app_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            if (c->app_dbg)                     /* developer: the helpers' shares once more, here */
                for (int p = 1; p <= (int) H; p++) append_row_part_ool(F, sh, s, p, (int) H + 1);
        }
    }
#else
    append_row_part(F, sh, s, 0, 1);
#endif
    {
        GLOBAL_AS float *const gd5 = uniform_ptr(ACT_D5(F, sh));
        for (int a = tid; a < F.NA; a += B) {
            float vs[32], ip = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) vs[k] = ldg((GLOBAL_AS const float *) gimgT, (unsigned) (k * P + s));
#pragma unroll
            for (int k = 0; k < 32; k++) ip += sh.pixels[a * 32 + k] * vs[k];
            stg(gd5, D5_AT(P, F.NA, a, s), ip);
        }
    }
#if FC_VARIANT_BIG
    if (F.gl0 < il)
        for (int a = tid; a < 2 * F.NA; a += B) {
            float v4[16], ip = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) v4[k] = F.imgT4[(size_t) k * P + s];
#pragma unroll
            for (int k = 0; k < 16; k++) ip += sh.pixels[a * 16 + k] * v4[k];
            ACT_D4(F, sh)[(size_t) a * P + s] = ip;
        }
#endif
    if (tid == 0) {
        const int E = sh.gs_n[0] + sh.gs_n[1];       /* tree children + edges of the new state */
        /* B_gram = 5 * 4 * N * (1 + E) read + 5 * 4 * N written -- the five table levels of the
         * reference (6..lc_max), one row per level, no mirrored entries.  (Until round 3 this counted what THIS
         * layout writes, 8 * 6 * (s + 1): the cached level-5 row and the mirror; 6 % more bytes per frame.) */
        sh.cnt.bytes_gram += (unsigned long long) (F.NL - 1) * 4ull * (s + 1) * (1 + E) + 4ull * (s + 1) * (F.NL - 1);
        sh.cnt.n_appends++;
    }
    gram_flush(F, sh, s + 1);
}

/* Start of the chroma bands: rle_chroma (codec/domain-pool.c:854-879) keeps the chroma_max
 * most referenced states as the domain list -- compute_hits (codec/wfalib.c:182-231): state 0
 * first, then by edge-target count descending (ties: lower state, the order glibc's stable
 * qsort leaves), only counts > 0, the kept ones ascending -- and the minimum block level
 * becomes the finest level the luminance band used (codec/coder.c:785-797). */
__device__ __noinline__ void op_chroma_pool(DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int states = sh.states, to = states - 1;
    Pool &m = sh.pool;
    const int maxd = F.chroma_max;
#if FC_VARIANT_BIG
    if (F.frame_type) { subtract_mc_dev(F, sh); __syncthreads(); }     /* codec/coder.c:798-799 */
#endif
    if (tid == 0) { sh.lc_min = F.ML; sh.ystates = states; }
    /* chroma dictionaries of more than 63 states (cfiasco --chroma-dictionary 64 ..; big builds): the list does not
     * fit sh.dl / one wave -- it lives in F.pool_states, the search is mp_steps_list_global */
    const bool longl = FC_GM || (FC_VARIANT_BIG && maxd > 63);
#if FC_GM
    /* default_chroma (codec/domain-pool.c:964-968): the constant, the uniform and the rle-no-chroma pool stay as they are */
    const bool keep_pool = sh.gm.pk[0] == FC_PK_CONSTANT || sh.gm.pk[0] == FC_PK_UNIFORM || sh.gm.pk[0] == FC_PK_RLE_NO_CHROMA;
#else
    const bool keep_pool = false;
#endif
    const int oldn = (int) m.n;
    (void) oldn;
    if (keep_pool) {
    } else
    if (longl && maxd < (int) m.n) {
        uint8_t *const mark = F.used;                     /* [P] scratch of the general scan: free between the bands */
        for (int d = tid; d < to; d += B) { __hip_atomic_store(&F.hits[d], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mark[d] = 0; }
        __syncthreads();
        for (int s = F.basis_states + tid; s <= to; s += B)
            for (int l = 0; l < 2; l++)
                for (int e = 0, d; (d = INTO(F, s, l, e)) != NOEDGE; e++)
                    __hip_atomic_fetch_add(&F.hits[d], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        unsigned long long best = 0;
        for (int d = 1 + tid; d < to; d += B) {
            int k = (short) __hip_atomic_load(&F.hits[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long pk = ((unsigned long long) (unsigned) k << 32) | (0xffffffffu - (unsigned) d);
            if (k > 0 && pk > best) best = pk;
        }
        int n = maxd < to ? maxd : to, npick = 0;
        if (n > 0) { if (tid == 0) mark[0] = 1; npick = 1; }
        unsigned long long *red = sh.red;
        while (npick < n) {                              /* the same rounds as below; a pick is a mark */
            unsigned long long w = best;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                unsigned long long t = __shfl_xor(w, o);
                if (t > w) w = t;
            }
            if (lane == 0) red[wave] = w;
            __syncthreads();
            unsigned long long g = red[0];
#pragma unroll
            for (int i = 1; i < B / 64; i++) if (red[i] > g) g = red[i];
            if (g == 0) break;
            int d = (int) (0xffffffffu - (unsigned) (g & 0xffffffffu));
            npick++;
            if (best == g) {
                mark[d] = 1;
                __hip_atomic_store(&F.hits[d], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                best = 0;
                for (int dd = 1 + tid; dd < to; dd += B) {
                    int k = (short) __hip_atomic_load(&F.hits[dd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long pk = ((unsigned long long) (unsigned) k << 32) | (0xffffffffu - (unsigned) dd);
                    if (k > 0 && pk > best) best = pk;
                }
            }
            __syncthreads();
        }
        __syncthreads();
        /* the kept states in ascending order (wfalib.c:226): every lane compacts its share of the marks */
        int *const scr = (int *) sh.pixels;              /* B counters; the block's pixels are not needed between the bands */
        const int chunk = (to + B - 1) / B, lo = tid * chunk, hi = lo + chunk < to ? lo + chunk : to;
        int cnt = 0;
        for (int d = lo; d < hi; d++) cnt += mark[d];
        scr[tid] = cnt;
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int t = 0; t < B; t++) { const int c = scr[t]; scr[t] = acc; acc += c; }
            m.n = (unsigned short) acc;
            scr[B] = 0x7fffffff;
        }
        __syncthreads();
        int o = scr[tid];
#if FC_GM
        if (GM_QAC(sh.gm.pk[0])) {
            /* qac_chroma (codec/domain-pool.c:466-498): a kept state keeps the probability index it had; through a
             * snapshot slot nobody uses between the bands (the compaction moves entries in place) */
            int16_t *q = GQ_CUR(sh, 0), *tmp = GQ_SNAP(sh, 0, 0);
            /* the reference walks the old and the new list side by side (:480-486): behind the first kept state that
             * the pool did not hold (a full pool) every index stays 0 */
            int miss = 0x7fffffff, oo = o;
            for (int d = lo; d < hi; d++)
                if (mark[d]) { const int pd = F.pos[d]; if ((pd < 0 || pd >= oldn) && oo < miss) miss = oo; oo++; }
            atomicMin(&scr[B], miss);
            __syncthreads();
            const int fm = scr[B];
            for (int d = lo; d < hi; d++) if (mark[d]) { tmp[o] = o < fm ? q[F.pos[d]] : (int16_t) 0; F.pool_states[o++] = (short) d; }
            __syncthreads();
            for (int i = tid; i < (int) m.n; i += B) q[i] = tmp[i];
        } else
#endif
        for (int d = lo; d < hi; d++) if (mark[d]) F.pool_states[o++] = (short) d;
    } else if (longl) {
        /* every pool state stays in the list (F.pool_states as it is) */
    } else
    if (maxd < (int) m.n) {
        /* histogram in HBM with device-scope atomics; read back past the L1 */
        for (int d = tid; d < to; d += B) __hip_atomic_store(&F.hits[d], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (int s = F.basis_states + tid; s <= to; s += B)
            for (int l = 0; l < 2; l++)
                for (int e = 0, d; (d = INTO(F, s, l, e)) != NOEDGE; e++)
                    __hip_atomic_fetch_add(&F.hits[d], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        /* lane-private best (count, lowest state) over the states d = 1 + tid, + B, ...; the
         * reference's counters are int16 (wfalib.c:187): wrap like them */
        unsigned long long best = 0;
        for (int d = 1 + tid; d < to; d += B) {
            int k = (short) __hip_atomic_load(&F.hits[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long pk = ((unsigned long long) (unsigned) k << 32) | (0xffffffffu - (unsigned) d);
            if (k > 0 && pk > best) best = pk;
        }
        int n = maxd < to ? maxd : to, npick = 0;
        if (n > 0) { if (tid == 0) sh.dl[0] = 0; npick = 1; }
        unsigned long long *red = sh.red;
        while (npick < n) {
            unsigned long long w = best;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                unsigned long long t = __shfl_xor(w, o);
                if (t > w) w = t;
            }
            if (lane == 0) red[wave] = w;
            __syncthreads();
            unsigned long long g = red[0];
#pragma unroll
            for (int i = 1; i < B / 64; i++) if (red[i] > g) g = red[i];
            if (g == 0) break;                           /* no state with a count > 0 left */
            int d = (int) (0xffffffffu - (unsigned) (g & 0xffffffffu));
            if (tid == 0) sh.dl[npick] = (short) d;
            npick++;
            if (best == g) {                             /* owner: retire it, rescan its share */
                __hip_atomic_store(&F.hits[d], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                best = 0;
                for (int dd = 1 + tid; dd < to; dd += B) {
                    int k = (short) __hip_atomic_load(&F.hits[dd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long pk = ((unsigned long long) (unsigned) k << 32) | (0xffffffffu - (unsigned) dd);
                    if (k > 0 && pk > best) best = pk;
                }
            }
            __syncthreads();
        }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < npick; i++) {            /* ascending, wfalib.c:226 */
                short v = sh.dl[i];
                int j = i;
                while (j > 0 && sh.dl[j - 1] > v) { sh.dl[j] = sh.dl[j - 1]; j--; }
                sh.dl[j] = v;
            }
            for (int i = 0; i < npick; i++) F.pool_states[i] = sh.dl[i];
            m.n = (unsigned short) npick;
        }
    } else if (tid < (int) m.n) {
        sh.dl[tid] = F.pool_states[tid];                 /* n <= chroma_max <= 63 */
    }
    __syncthreads();
    if (tid == 0 && !keep_pool) { m.y_index = 0; m.max_domains = m.n; }
    for (int s = tid; s < states; s += B) F.pos[s] = -1;
    /* finest level with a linear combination in the luminance band */
    int mn = F.ML;
    for (int s = F.basis_states + tid; s < states; s += B) {
        int lin = (TREE(F, s, 0) == RANGE_) + (TREE(F, s, 1) == RANGE_);
        unsigned lv = (unsigned) ((int) F.level_of_state[s] - 1);
        if (lin && lv < (unsigned) mn) mn = (int) lv;
    }
    atomicMin(&sh.lc_min, mn);
    __syncthreads();
    if (longl) { for (int i = tid; i < (int) m.n; i += B) F.pos[F.pool_states[i]] = (short) i; }
    else if (tid < (int) m.n) F.pos[sh.dl[tid]] = (short) tid;
#if !FC_SPEC
    if (!F.bx && F.chroma_sparse) chroma_need_static(F, sh);
#endif
}

#if FC_VARIANT_BIG
/* ------------------------------------------------------------------ prediction (codec/prediction.c)
 *
 * predict_range (:96-208) tries a third alternative for a range after its linear combination
 * and its subdivision: approximate the range coarsely (its DC part for an intra frame, a motion
 * compensated block of the reference frame otherwise), run the SAME partition search on the
 * residual (`delta' = YES: delta pool, delta coefficient model), keep what is cheapest.  The
 * states the subdivision appended are put aside meanwhile (store_state_data, :502-565) and the
 * residual search re-uses their ids.  Here:
 *   OP_PRED_SETUP   block pixels + norms -> F.pix_save, residual -> sh.pixels, tables of the
 *                   residual block into the SECOND table set (ipis_alt / d5_alt / d4_alt: the
 *                   reference swaps the per-state table pointers, :302-309,443-450), automaton
 *                   rows of the displaced states -> F.sv_auto, delta models become active
 *   op_append       copies the table rows of a displaced id to F.sv_gram / F.sv_img the first
 *                   time the residual search appends a state with that id (copy on write)
 *   OP_PRED_FINISH  everything back; on failure the saved rows return, on success the new
 *                   states get zeroed <sub-block, state> rows (:342-345,481-484)
 */

/* squared norms of the sub-blocks of a block of 2^level pixels in sh.pixels, heap order */
__device__ void block_norms(Sh &sh, int level, int ns)
{
    const int tid = threadIdx.x;
    for (int slot = tid; slot < ns; slot += B) {
        int depth = 31 - __clz(slot + 1);
        int lv = level - depth, size = 1 << lv;
        int adr = slot + 1 - (1 << depth);
        float nrm = 0;
        const float *px = sh.pixels + adr * size;
        for (int k = 0; k < size; k++) nrm += px[k] * px[k];      /* sequential, codec/approx.c:388-389 */
        sh.norms[slot] = nrm;
    }
}

/* exchange the active and the resting model set (all lanes; barriers by the caller) */
__device__ void swap_model_sets(Sh &sh)
{
    const int tid = threadIdx.x;
#if FC_HM           /* models of more 16-byte units than lanes */
    for (int i = tid; i < sh.n16; i += B) {
        uint4 a = ((uint4 *) &sh.cb)[i], b = ((uint4 *) &sh.dcb)[i];
        ((uint4 *) &sh.cb)[i] = b; ((uint4 *) &sh.dcb)[i] = a;
    }
    if (tid == 128) {
#else
    if (tid < sh.n16) {
        uint4 a = ((uint4 *) &sh.cb)[tid], b = ((uint4 *) &sh.dcb)[tid];
        ((uint4 *) &sh.cb)[tid] = b; ((uint4 *) &sh.dcb)[tid] = a;
    } else if (tid == 128) {
#endif
        Pool t = sh.pool; sh.pool = sh.dpool; sh.dpool = t;
#if FC_GM
        { int k = sh.gm.pk[0]; sh.gm.pk[0] = sh.gm.pk[1]; sh.gm.pk[1] = k; k = sh.gm.ck[0]; sh.gm.ck[0] = sh.gm.ck[1]; sh.gm.ck[1] = k; }
        sh.gm.qa ^= 1;
#endif
    } else if (tid == 129) {
        int i; float f;
        i = sh.par.rpf_mant; sh.par.rpf_mant = sh.dq.rpf_mant; sh.dq.rpf_mant = i;
        i = sh.par.dc_mant; sh.par.dc_mant = sh.dq.dc_mant; sh.dq.dc_mant = i;
        i = sh.par.sy; sh.par.sy = sh.dq.sy; sh.dq.sy = i;
        i = sh.par.dcs; sh.par.dcs = sh.dq.dcs; sh.dq.dcs = i;
        f = sh.par.rpf_range; sh.par.rpf_range = sh.dq.rpf_range; sh.dq.rpf_range = f;
        f = sh.par.dc_range; sh.par.dc_range = sh.dq.dc_range; sh.dq.dc_range = f;
        i = sh.par.half_nd; sh.par.half_nd = sh.dq.half_nd; sh.dq.half_nd = i;
        i = sh.par.half_dc; sh.par.half_dc = sh.dq.half_dc; sh.dq.half_dc = i;
    }
}

/* ---- motion compensation (codec/mwfa.c; P frames, full-pixel vectors) ---- */

/* MPEG's vector-component code lengths (mv_code_table[][1], codec/mwfa.c:40-50) */
__device__ __forceinline__ float mv_bits(int v, int sr)
{
    /* lengths 11 11 11 11 11 11 10 10 10 8 8 8 7 5 4 3 | 1 | mirrored: one nibble per code */
    const unsigned long long len = 0xbbbbbbaaa8887543ull;
    const int i = v + sr;
    if (i == 16) return 1.0f;
    return (float) ((len >> (4 * (i < 16 ? 15 - i : i - 17))) & 15u);
}

/* fill_norms_table (codec/mwfa.c:545-602): squared norm of original - displaced reference block
 * for every displacement of the search window, 0 outside the frame.  One displacement per lane
 * and pass; per displacement the pixels are summed in raster order like mcpe_norm (:658-684). */
__device__ void fill_norms(const DevFrame &__restrict__ F, int x0, int y0, int level)
{
    const int tid = threadIdx.x, sr = F.search_range, n = 4 * sr * sr;
    const int bw = (int) width_of_level(level), bh = (int) height_of_level(level), W = F.width, H = F.height;
    float *dst = F.mc_fwd + (size_t) (level - F.p_min) * n;
    float *dstb = F.mc_bwd + (size_t) (level - F.p_min) * n;
    const int16_t *orig = F.pix16 + (size_t) y0 * W + x0;
    const bool bframe = F.frame_type == 2;
    for (int idx = tid; idx < n; idx += B) {
        const int mx = idx % (2 * sr) - sr, my = idx / (2 * sr) - sr;
        float norm = 0.0f, normb = 0.0f;
        if (!(x0 + mx < 0 || x0 + mx + bw > W || y0 + my < 0 || y0 + my + bh > H)) {
            const int16_t *ref = F.past + (size_t) (y0 + my) * W + (x0 + mx);
            for (int y = 0; y < bh; y++)
                for (int x = 0; x < bw; x++) {
                    const int q = (int) (short) (orig[(size_t) y * W + x] - ref[(size_t) y * W + x]) / 16;
                    norm += (float) (q * q);
                }
            if (bframe) {
                const int16_t *reb = F.future + (size_t) (y0 + my) * W + (x0 + mx);
                for (int y = 0; y < bh; y++)
                    for (int x = 0; x < bw; x++) {
                        const int q = (int) (short) (orig[(size_t) y * W + x] - reb[(size_t) y * W + x]) / 16;
                        normb += (float) (q * q);
                    }
            }
            dst[idx] = norm;
            if (bframe) dstb[idx] = normb;
        } else {
            dst[idx] = 0.0f;                 /* both tables, whatever the frame type (:576-577) */
            if (F.mc_bwd) dstb[idx] = 0.0f;
        }
    }
}

/* after a child of a motion compensated node: table of the child's level if the child was not
 * searched (subdivide.c:311-315), then update_norms_table (prediction.c:229-254); `first`
 * stands for the clear_norms_table at the node's entry (0 + x == x) */
__device__ __noinline__ void op_norms(DevFrame &__restrict__ F, Sh &__restrict__ sh, int level, int first,
                                      int fill_level, int xy)
{
    const int tid = threadIdx.x, n = 4 * F.search_range * F.search_range;
    if (fill_level >= 0) {
        fill_norms(F, xy & 0xffff, xy >> 16, fill_level);
        __syncthreads();
    }
    if (level > F.p_min) {
        float *dst = F.mc_fwd + (size_t) (level - F.p_min) * n;
        const float *src = dst - n;
        for (int i = tid; i < n; i += B) dst[i] = first ? 0.0f + src[i] : dst[i] + src[i];
        if (F.frame_type == 2) {
            float *dstb = F.mc_bwd + (size_t) (level - F.p_min) * n;
            const float *srcb = dstb - n;
            for (int i = tid; i < n; i += B) dstb[i] = first ? 0.0f + srcb[i] : dstb[i] + srcb[i];
        } else if (first && F.mc_bwd) {      /* clear_norms_table clears both */
            float *dstb = F.mc_bwd + (size_t) (level - F.p_min) * n;
            for (int i = tid; i < n; i += B) dstb[i] = 0.0f;
        }
    }
}

/* find_best_mv (codec/mwfa.c:686-798): first displacement, in scan order, with the smallest
 * costs norm + (bits_x + bits_y) * price.  All lanes; result on lane 0. */
__device__ void best_mv(const DevFrame &__restrict__ F, Sh &__restrict__ sh, const float *norms, int x0, int y0,
                        int bw, int bh, float price, int &mx_out, int &my_out, float &bits, float &costs_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sr = F.search_range, n = 4 * sr * sr;
    const int W = F.width, H = F.height;
    unsigned long long best = ~0ull;
    for (int idx = tid; idx < n; idx += B) {
        const int mx = idx % (2 * sr) - sr, my = idx / (2 * sr) - sr;
        if (x0 + mx >= 0 && y0 + my >= 0 && x0 + mx + bw <= W && y0 + my + bh <= H) {
            const float costs = norms[idx] + (mv_bits(mx, sr) + mv_bits(my, sr)) * price;
            /* costs >= 0: the float's bit pattern orders like the value */
            const unsigned long long key = ((unsigned long long) __float_as_uint(costs) << 32) | (unsigned) idx;
            if (key < best) best = key;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(best, o);
        if (t < best) best = t;
    }
    __syncthreads();                         /* sh.mcred of an earlier call has been read */
    if (lane == 0) sh.mcred[wave] = best;
    __syncthreads();
    unsigned long long g = sh.mcred[0];
    for (int i = 1; i < B / 64; i++) if (sh.mcred[i] < g) g = sh.mcred[i];
    mx_out = my_out = 0;
    costs_out = MAXCOSTS;
    if (g != ~0ull && __uint_as_float((unsigned) (g >> 32)) < MAXCOSTS) {
        const int idx = (int) (g & 0xffffffffu);
        mx_out = idx % (2 * sr) - sr; my_out = idx / (2 * sr) - sr;
        costs_out = __uint_as_float((unsigned) (g >> 32));
    }
    bits = mv_bits(mx_out, sr) + mv_bits(my_out, sr);
}

/* find_P_frame_mc / find_B_frame_mc (codec/mwfa.c:302-543; cross_B_search is never set: the
 * reference copies it from half_pixel_prediction, codec/coder.c:359, and half-pixel vectors are
 * refused by the host).  Result in sh.mc. */
__device__ __noinline__ void op_mc_search(DevFrame &__restrict__ F, Sh &__restrict__ sh, int level, int xy, int fill)
{
    const int tid = threadIdx.x, sr = F.search_range, n = 4 * sr * sr;
    const int x0 = xy & 0xffff, y0 = xy >> 16;
    const int bw = (int) width_of_level(level), bh = (int) height_of_level(level), W = F.width;
    const float price = sh.st[sh.sp].price;
    if (fill & 1) { fill_norms(F, x0, y0, level); __syncthreads(); }
    if (fill & 2) {
        /* a node above p_min_level whose children were never visited (its level is not above the
         * smallest block level, which a colour stream ratchets upwards, codec/coder.c:785-797):
         * the table is what clear_norms_table left at the entry (prediction.c:210-227) */
        float *t = F.mc_fwd + (size_t) (level - F.p_min) * n;
        for (int i = tid; i < n; i += B) t[i] = 0.0f;
        if (F.mc_bwd) { t = F.mc_bwd + (size_t) (level - F.p_min) * n; for (int i = tid; i < n; i += B) t[i] = 0.0f; }
        __syncthreads();
    }
    int fx, fy, bx = 0, by = 0;
    float fbits, bbits = 0, fcosts, bcosts = 0;
    best_mv(F, sh, F.mc_fwd + (size_t) (level - F.p_min) * n, x0, y0, bw, bh, price, fx, fy, fbits, fcosts);
    if (F.frame_type != 2) {
        if (tid == 0) { sh.mc.type = MV_FORWARD; sh.mc.fx = fx; sh.mc.fy = fy; sh.mc.bx = sh.mc.by = 0;
                        sh.mc.bits = fbits; sh.mc.tree_bits = 1.0f; }
        return;
    }
    best_mv(F, sh, F.mc_bwd + (size_t) (level - F.p_min) * n, x0, y0, bw, bh, price, bx, by, bbits, bcosts);
    /* both vectors together: norm of original - (forward block + backward block) / 2, summed in
     * raster order (mcpe_norm :658-684).  The terms are integers: as long as the total stays
     * below 2^24 every partial sum is exact and the order does not matter -- summed in parallel;
     * otherwise lane 0 repeats the sum in order. */
    const int16_t *orig = F.pix16 + (size_t) y0 * W + x0;
    const int16_t *r1 = F.past + (size_t) (y0 + fy) * W + (x0 + fx);
    const int16_t *r2 = F.future + (size_t) (y0 + by) * W + (x0 + bx);
    unsigned long long part = 0;
    for (int i = tid; i < bw * bh; i += B) {
        const int x = i % bw, y = i / bw;
        const int q = (int) (short) (orig[(size_t) y * W + x] - ((int) r1[(size_t) y * W + x] + (int) r2[(size_t) y * W + x]) / 2) / 16;
        part += (unsigned long long) (q * q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();
    if ((tid & 63) == 0) sh.mcred[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        unsigned long long total = 0;
        for (int i = 0; i < B / 64; i++) total += sh.mcred[i];
        float inorm;
        if (total < (1ull << 24)) inorm = (float) total;
        else {
            inorm = 0.0f;
            for (int y = 0; y < bh; y++)
                for (int x = 0; x < bw; x++) {
                    const int q = (int) (short) (orig[(size_t) y * W + x] - ((int) r1[(size_t) y * W + x] + (int) r2[(size_t) y * W + x]) / 2) / 16;
                    inorm += (float) (q * q);
                }
        }
        const float forward_costs = fcosts + 3 * price, backward_costs = bcosts + 3 * price;
        const float interp_bits = fbits + bbits;
        const float interp_costs = inorm + (interp_bits + 2) * price;
        int type;
        if (forward_costs <= interp_costs) type = forward_costs <= backward_costs ? MV_FORWARD : MV_BACKWARD;
        else type = backward_costs <= interp_costs ? MV_BACKWARD : MV_INTERPOLATED;
        sh.mc.type = type;
        sh.mc.fx = type != MV_BACKWARD ? fx : 0; sh.mc.fy = type != MV_BACKWARD ? fy : 0;
        sh.mc.bx = type != MV_FORWARD ? bx : 0; sh.mc.by = type != MV_FORWARD ? by : 0;
        sh.mc.tree_bits = type == MV_INTERPOLATED ? 2.0f : 3.0f;
        sh.mc.bits = type == MV_FORWARD ? fbits : type == MV_BACKWARD ? bbits : interp_bits;
    }
}

/* subtract_mc (codec/mwfa.c:156-300): private chroma planes = original chroma - luminance motion
 * compensation with the vector components rounded to even; called once, at the first chroma band */
__device__ void subtract_mc_dev(DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    const int tid = threadIdx.x, W = F.width;
    const size_t npix = (size_t) F.plane;
    for (size_t i = tid; i < 2 * npix; i += B) F.pix_chroma[i] = F.pix16[npix + i];
    __syncthreads();
    for (int s = F.basis_states; s < sh.states; s++)                 /* blocks do not overlap */
        for (int l = 0; l < 2; l++) {
            const int type = F.mv[(0 * 2 + l) * F.PA + s];
            if (type == MV_NONE) continue;       /* uniform */
            const int lv = (int) F.level_of_state[s] - 1;
            const int bw = (int) width_of_level(lv), bh = (int) height_of_level(lv);
            const int x0 = F.x[l * F.PA + s], y0 = F.y[l * F.PA + s];
            const int fx = (F.mv[(1 * 2 + l) * F.PA + s] / 2) * 2, fy = (F.mv[(2 * 2 + l) * F.PA + s] / 2) * 2;
            const int bx = (F.mv[(3 * 2 + l) * F.PA + s] / 2) * 2, by = (F.mv[(4 * 2 + l) * F.PA + s] / 2) * 2;
            for (int b = 0; b < 2; b++) {
                int16_t *o = F.pix_chroma + (size_t) b * npix + (size_t) y0 * W + x0;
                const int16_t *r1 = type == MV_BACKWARD ? F.future + (size_t) (b + 1) * npix + (size_t) (y0 + by) * W + (x0 + bx)
                                                        : F.past + (size_t) (b + 1) * npix + (size_t) (y0 + fy) * W + (x0 + fx);
                const int16_t *r2 = type == MV_INTERPOLATED ? F.future + (size_t) (b + 1) * npix + (size_t) (y0 + by) * W + (x0 + bx) : r1;
                for (int i = tid; i < bw * bh; i += B) {
                    const size_t p = (size_t) (i / bw) * W + (i % bw);
                    o[p] = (int16_t) (o[p] - (type == MV_INTERPOLATED ? ((int) r1[p] + (int) r2[p]) / 2 : (int) r1[p]));
                }
            }
        }
}

/* a0 = level of the range, a1 = its address in the block; the frame on top of the stack holds
 * the DC weight (nd_w).  States [fr.states, fr.rec_states) are the ones the subdivision made. */
__device__ __noinline__ void op_pred_setup(DevFrame &__restrict__ F, Sh &__restrict__ sh, int level, int address)
{
    const bool mc = sh.st[sh.sp].try_pred == 2;
    const int tid = threadIdx.x, il = F.images_level;
    SFrame &fr = sh.st[sh.sp];
    const int size = 1 << level, npx = 1 << F.lc_max;
    /* block pixels and norms aside */
    for (int i = tid; i < npx; i += B) F.pix_save[i] = sh.pixels[i];
    for (int i = tid; i < FC_PIXELS / 32; i += B) F.pix_save[FC_PIXELS + i] = sh.norms[i];
    /* automaton rows of the displaced states aside (store_state_data) */
    for (int s = fr.states + tid; s < fr.rec_states; s += B) {
        FcSavedRow &r = F.sv_auto[s - fr.states];
        for (int l = 0; l < 2; l++) {
            r.tree[l] = TREE(F, s, l);
            r.x[l] = F.x[l * F.PA + s]; r.y[l] = F.y[l * F.PA + s];
            r.ycol[l] = F.color ? F.ycol[l * F.PA + s] : 0;
            for (int e = 0; e < 6; e++) { r.into[l][e] = INTO(F, s, l, e); r.weight[l][e] = WEIGHT(F, s, l, e); }
        }
        r.final_d = F.final_d[s]; r.level = F.level_of_state[s]; r.dtype = F.domain_type[s];
        r.pos = F.pos[s]; r.tables = 0;
        if (F.frame_type)
            for (int l = 0; l < 2; l++)
                for (int k = 0; k < 5; k++) r.mv[l][k] = F.mv[(k * 2 + l) * F.PA + s];
    }
    /* residual: range pixels + w, w = - weight * <image of state 0 at level 0> (:417-427) */
    if (mc) {
        /* motion compensated prediction error of the range, bintree order, / 16 truncated
         * (get_mcpe + cut_to_bintree, codec/mwfa.c:610-656, codec/subdivide.c:504-541) */
        const Range &rg = fr.rg;
        const int W = F.width, type = fr.prange.mv[0];
        const int16_t *orig = F.pix16 + (size_t) rg.y * W + rg.x;
        const int16_t *r1 = type == MV_BACKWARD ? F.future + (size_t) (rg.y + fr.prange.mv[4]) * W + (rg.x + fr.prange.mv[3])
                                                : F.past + (size_t) (rg.y + fr.prange.mv[2]) * W + (rg.x + fr.prange.mv[1]);
        const int16_t *r2 = type == MV_INTERPOLATED ? F.future + (size_t) (rg.y + fr.prange.mv[4]) * W + (rg.x + fr.prange.mv[3]) : r1;
        for (int i = tid; i < size; i += B) {
            unsigned xo = 0, yo = 0;
#pragma unroll
            for (int b = 0; b < 13; b++) {
                yo |= ((i >> (2 * b)) & 1u) << b;
                xo |= ((i >> (2 * b + 1)) & 1u) << b;
            }
            const size_t o = (size_t) yo * W + xo;
            const short d = type == MV_INTERPOLATED ? (short) (orig[o] - ((int) r1[o] + (int) r2[o]) / 2)
                                                    : (short) (orig[o] - r1[o]);
            sh.pixels[i] = (float) ((int) d / 16);
        }
    } else {
        const float w = -fr.nd_w * F.img[0];
        float v[FC_PIXELS / B];
#pragma unroll
        for (int it = 0; it < FC_PIXELS / B; it++) {
            const int i = tid + it * B;
            v[it] = i < size ? sh.pixels[address * size + i] + w : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < FC_PIXELS / B; it++) {
            const int i = tid + it * B;
            if (i < size) sh.pixels[i] = v[it];
        }
    }
    if (tid == 0) {
        sh.par.ipis = F.ipis_alt; sh.par.d5 = F.d5_alt; sh.par.d4 = F.d4_alt;
        sh.pred_active = 1; sh.pred_lo = fr.states; sh.pred_rec = fr.rec_states;
        for (int i = 0; i < FC_MAXSAVE / 32; i++) sh.pred_saved[i] = 0;
    }
    swap_model_sets(sh);
    __syncthreads();
    /* tables of the residual block for every state (compute_ip_images_state(0, 0, level, 1, 0)) */
    {
        const int coopD = coop_publish(F, sh, level, 0);
        if (level > il) block_norms(sh, level, (1 << (level - il)) - 1);
        if (coopD) {
            coop_finish(F, sh, level, 0, coopD);
            op_ipis(F, sh, 0, 0, level, 0, level - coopD + 1);
            return;
        }
    }
    op_d5(F, sh, 0, table_states(sh), level >= il ? 1 << (level - il) : 0, level >= il - 1 ? 1 << (level - il + 1) : 0);
    __syncthreads();
    if (level > il) op_ipis(F, sh, 0, 0, level, 0);
}

/* copy-on-write of the table rows of a displaced state id (called by all lanes from op_append) */
__device__ void pred_save_tables(DevFrame &__restrict__ F, Sh &__restrict__ sh, int s)
{
    const int tid = threadIdx.x, P = F.P, idx = s - sh.pred_lo;
    if (!sh.pred_active || s < sh.pred_lo || s >= sh.pred_rec || idx >= F.max_save) return;   /* uniform */
    if ((sh.pred_saved[idx >> 5] >> (idx & 31)) & 1u) return;
    if (!F.sv_auto[idx].dtype) return;                 /* the displaced state had no tables */
    for (int q = 0; q < F.NL; q++) {
        const float *G = GRAM(F, q) + GROW(s, P);
        float *dst = F.sv_gram + ((size_t) idx * F.NL + q) * P;
        for (int t = tid; t <= s; t += B) dst[t] = G[t];
    }
    {
        float *dst = F.sv_img + (size_t) idx * (F.NI + 48 + F.NL);
        for (int i = tid; i < F.NI; i += B) dst[i] = F.img[(size_t) s * F.NI + i];
        if (tid < 32) dst[F.NI + tid] = F.imgT[(size_t) tid * P + s];
        else if (tid < 48 && F.gl0 < F.images_level) dst[F.NI + tid] = F.imgT4[(size_t) (tid - 32) * P + s];
        else if (tid >= 64 && tid < 64 + F.NL) dst[F.NI + 48 + tid - 64] = F.diag[(size_t) (tid - 64) * P + s];
    }
    __syncthreads();
    if (tid == 0) { sh.pred_saved[idx >> 5] |= 1u << (idx & 31); F.sv_auto[idx].tables = 1; }
    __syncthreads();
}

/* a0 = the prediction is kept */
__device__ __noinline__ void op_pred_finish(DevFrame &__restrict__ F, Sh &__restrict__ sh, int keep)
{
    const int tid = threadIdx.x, P = F.P;
    SFrame &fr = sh.st[sh.sp];
    const int npx = 1 << F.lc_max;
    const int new_states = sh.states;           /* states of the residual search */
    swap_model_sets(sh);
    for (int i = tid; i < npx; i += B) sh.pixels[i] = F.pix_save[i];
    for (int i = tid; i < FC_PIXELS / 32; i += B) sh.norms[i] = F.pix_save[FC_PIXELS + i];
    if (tid == 0) {
        sh.par.ipis = F.ipis; sh.par.d5 = F.d5; sh.par.d4 = F.d4;
        sh.pred_active = 0;
    }
    __syncthreads();
    if (keep) {
        /* the delta pool saw every append; the normal pool holds the same list */
#if !FC_GM              /* (generic models: both pools were offered every state, each by its own rule -- gm_offer) */
        if (tid == 0) { sh.pool.n = sh.dpool.n; }
#endif
        /* rows of the new states in the block's tables: zero (:342-345,481-484); their level-5
         * dots with the block are what later table updates start from */
        for (int s = fr.states + tid; s < new_states; s += B)
            if (F.domain_type[s])
                for (int slot = 0; slot < F.NS; slot++) F.ipis[(size_t) slot * P + s] = 0.0f;
        op_d5(F, sh, fr.states, new_states, F.NA, 2 * F.NA);
    } else {
        /* restore_state_data (:567-625) */
        for (int s = fr.states + tid; s < fr.rec_states; s += B) {
            const FcSavedRow &r = F.sv_auto[s - fr.states];
            for (int l = 0; l < 2; l++) {
                TREE(F, s, l) = r.tree[l];
                F.x[l * F.PA + s] = r.x[l]; F.y[l * F.PA + s] = r.y[l];
                if (F.color) F.ycol[l * F.PA + s] = r.ycol[l];
                for (int e = 0; e < 6; e++) { INTO(F, s, l, e) = r.into[l][e]; WEIGHT(F, s, l, e) = r.weight[l][e]; }
            }
            F.final_d[s] = r.final_d; F.level_of_state[s] = r.level; F.domain_type[s] = r.dtype;
            F.pos[s] = r.pos;
            if (r.pos >= 0) F.pool_states[r.pos] = (short) s;
            if (F.frame_type)
                for (int l = 0; l < 2; l++)
                    for (int k = 0; k < 5; k++) F.mv[(k * 2 + l) * F.PA + s] = r.mv[l][k];
        }
        for (int idx = 0; idx < fr.rec_states - fr.states && idx < F.max_save; idx++) {
            if (!((sh.pred_saved[idx >> 5] >> (idx & 31)) & 1u)) continue;      /* uniform */
            const int s = fr.states + idx;
            for (int q = 0; q < F.NL; q++) {
                float *G = GRAM(F, q) + GROW(s, P);
                const float *src = F.sv_gram + ((size_t) idx * F.NL + q) * P;
                for (int t = tid; t <= s; t += B) G[t] = src[t];
            }
            const float *src = F.sv_img + (size_t) idx * (F.NI + 48 + F.NL);
            for (int i = tid; i < F.NI; i += B) F.img[(size_t) s * F.NI + i] = src[i];
            if (tid < 32) F.imgT[(size_t) tid * P + s] = src[F.NI + tid];
            else if (tid < 48 && F.gl0 < F.images_level) F.imgT4[(size_t) (tid - 32) * P + s] = src[F.NI + tid];
            else if (tid >= 64 && tid < 64 + F.NL) F.diag[(size_t) (tid - 64) * P + s] = src[F.NI + 48 + tid - 64];
        }
    }
}
#endif

#if FC_VARIANT_BIG
#define SNAP(sh) ((sh).snap)
#define NSLOT(sh) ((sh).nslot)
#define SNAP_TM(sh) ((sh).snap_tm_p)
#define TM_SLOTS(sh) ((sh).nslot == 5 ? 2 : 1)
#else
#define SNAP(sh) ((sh).snap_pool)
#define NSLOT(sh) 2
#define SNAP_TM(sh) ((uint4 *) (sh).snap_tm)
#define TM_SLOTS(sh) 1
#endif
/* aac snapshot slots of a depth: 0 entry, 1 after the linear combination; with prediction (big
 * build) 2 = resting model at entry, 3 / 4 = active / resting model after the recursion
 * (rec_coeff_model, rec_d_coeff_model of predict_range) */
#if FC_VARIANT_BIG
#define SNAP_AT(sh, depth, which) (SNAP(sh) + ((depth) * NSLOT(sh) + (which)) * (sh).n16)
#else
/* slot 0 of depth d is slot d; slot 1 exists for the block levels with children only and follows
 * the depth slots: snap_b1 + d (the depth of a node is frame level - node level) */
#define SNAP_AT(sh, depth, which) (SNAP(sh) + ((which) ? (sh).par.snap_b1 + (depth) : (depth)) * (sh).n16)
#endif
/* tree-model snapshots: slot 0 entry, slot 1 (prediction) after the recursion */
/* uint4 per tree-model snapshot: both models (4 ML words) in the big build, the first one in the
 * default build */
#if FC_VARIANT_BIG
#define TM_N16(ML) (ML)
#else
#define TM_N16(ML) ((2 * (ML) + 3) / 4)
#endif
#define TM_AT(sh, depth, which, ML) (SNAP_TM(sh) + ((depth) * TM_SLOTS(sh) + (which)) * TM_N16(ML))

#if FC_GM
/* ---- generic models (frame_coder.h FC_GM) ------------------------------------------------------------
 * The two model sets keep ONE list of states (F.pool_states, F.pos) like the two `rle' pools of the other builds:
 * every pool that keeps a list takes the states it is offered in the same order until it is full
 * (codec/subdivide.c:571-581; rle_append codec/domain-pool.c:832-852, qac_append :448-464, default_append
 * :957-962), so the list of a pool is the first Pool.n entries of the common one.  A `uniform' pool has no model:
 * its list is every usable state (:578-590), Pool.n counts them; the `constant' pool is the list {0} (:518-528). */

__device__ __forceinline__ float gm_m0(const Sh &sh, int idx) { return sh.m0tab[qac_shift(idx)]; }   /* matrix_0, domain-pool.c:970-999 */
__device__ __forceinline__ float gm_m1(int idx) { return (float) qac_shift(idx); }                   /* matrix_1 */

/* lane 0: n probability indices, 16 bytes at a time (the arrays are P int16 apart, P a multiple of 64) */
__device__ void gq_copy(int16_t *dst, const int16_t *src, int n)
{
    const int n16 = (n + 7) / 8;
    for (int i = 0; i < n16; i++) ((uint4 *) dst)[i] = ((const uint4 *) src)[i];
}
/* qac_model_duplicate (codec/domain-pool.c:318-331) beside the copy of the Pool struct: lane 0 */
__device__ void gq_save(Sh &sh, int set, int depth, int slot)
{
    if (GM_QAC(sh.gm.pk[set])) gq_copy(GQ_SNAP(sh, depth, slot), GQ_CUR(sh, set), set ? sh.dpool.n : sh.pool.n);
}
__device__ void gq_load(Sh &sh, int set, int depth, int slot)      /* AFTER the Pool struct is back: its n says how many */
{
    if (GM_QAC(sh.gm.pk[set])) gq_copy(GQ_CUR(sh, set), GQ_SNAP(sh, depth, slot), set ? sh.dpool.n : sh.pool.n);
}
/* the same by all lanes of the workgroup */
__device__ __forceinline__ void gq_save_par(Sh &sh, int set, int depth, int slot)
{
    if (!GM_QAC(sh.gm.pk[set])) return;
    const int n = set ? sh.dpool.n : sh.pool.n;
    int16_t *d = GQ_SNAP(sh, depth, slot);
    const int16_t *c = GQ_CUR(sh, set);
    for (int i = threadIdx.x; i < n; i += B) d[i] = c[i];
}

/* would the pool take another state?  (the constant and the uniform pool take everything) */
__device__ __forceinline__ bool gm_accepts(const Pool &m, int kind)
{
    return kind == FC_PK_CONSTANT || kind == FC_PK_UNIFORM || m.n < m.max_domains;
}
/* ->append; true: the pool's list grew */
__device__ bool gm_take(Sh &sh, Pool &m, int kind, int set, int state)
{
    if (kind == FC_PK_CONSTANT) return false;
    if (kind != FC_PK_UNIFORM && m.n >= m.max_domains) return false;
    if (GM_QAC(kind)) { int16_t *q = GQ_CUR(sh, set); q[m.n] = m.n > 0 ? q[m.n - 1] : (int16_t) 0; }
    if (GM_RLE(kind) && state == 0) { m.d0_index = 0; m.d0_n = 1; }
    m.n++;
    return true;
}
/* a non-auxiliary state is offered to both pools (both of normal_domains / delta_domains are on) */
__device__ void gm_offer(DevFrame &F, Sh &sh, int s)
{
    const int L = sh.pool.n > sh.dpool.n ? sh.pool.n : sh.dpool.n;        /* length of the common list */
    bool grow = gm_take(sh, sh.pool, sh.gm.pk[0], 0, s);
    if (F.pred_on) grow = gm_take(sh, sh.dpool, sh.gm.pk[1], 1, s) || grow;
    F.pos[s] = -1;
    if (grow) { F.pos[s] = (short) L; F.pool_states[L] = (short) s; }
}
#endif

/* the same snapshots taken by the whole workgroup around a linear-combination search
 * (codec/subdivide.c:188-237): before it, models -> slot 0 (+ tree model); after it, models ->
 * slot 1 and slot 0 -> models.  One 16-byte element per lane. */
__device__ __forceinline__ void snap_coop_before(Sh &sh, SFrame &fr, int depth, int ML)
{
    const int tid = threadIdx.x;
#if FC_HM
    for (int i = tid; i < sh.n16; i += B) SNAP_AT(sh, depth, 0)[i] = ((const uint4 *) &sh.cb)[i];
    if (tid >= 96 && tid < 96 + TM_N16(ML)) TM_AT(sh, depth, 0, ML)[tid - 96] = ((const uint4 *) sh.tm)[tid - 96];
    if (tid == 128) fr.pool0 = sh.pool;
#if FC_GM
    /* The reference duplicates all four models at every node (codec/subdivide.c:185-192).  Inside a residual search
     * the resting (normal) models are not touched -- except that the normal pool is offered the states the search
     * appends (gm_offer): its length goes back with the active pool's */
    if (tid == 130) fr.rn0 = sh.dpool.n;
    gq_save_par(sh, 0, depth, 0);
#endif
    if (sh.nslot == 5 && !fr.delta) {
        if (tid == 129) fr.dpool0 = sh.dpool;
        for (int i = tid; i < sh.n16; i += B) SNAP_AT(sh, depth, 2)[i] = ((const uint4 *) &sh.dcb)[i];
#if FC_GM
        gq_save_par(sh, 1, depth, 2);
#endif
    }
    return;
#endif
    if (tid < sh.n16) SNAP_AT(sh, depth, 0)[tid] = ((const uint4 *) &sh.cb)[tid];
    else if (tid >= 96 && tid < 96 + TM_N16(ML))   /* n16 <= 82 (FC_MAXCOEFF_BIG), ML <= 26 */
        TM_AT(sh, depth, 0, ML)[tid - 96] = ((const uint4 *) sh.tm)[tid - 96];
    else if (tid == 128) fr.pool0 = sh.pool;
#if FC_VARIANT_BIG
    /* a node outside a residual search also keeps the resting (delta) models: a prediction
     * further down may change them, and this node may have to go back (subdivide.c:189-191) */
    else if (sh.nslot == 5 && !fr.delta) {
        if (tid == 129) fr.dpool0 = sh.dpool;
        else if (tid >= 160 && tid < 160 + sh.n16) SNAP_AT(sh, depth, 2)[tid - 160] = ((const uint4 *) &sh.dcb)[tid - 160];
    }
#endif
}

__device__ __forceinline__ void snap_coop_after(Sh &sh, SFrame &fr, int depth)
{
    const int tid = threadIdx.x;
#if FC_HM
    for (int i = tid; i < sh.n16; i += B) {
        SNAP_AT(sh, depth, 1)[i] = ((const uint4 *) &sh.cb)[i];
        ((uint4 *) &sh.cb)[i] = SNAP_AT(sh, depth, 0)[i];
    }
#if FC_GM
    if (GM_QAC(sh.gm.pk[0])) {           /* pool_lc <- the pool after the combination; the pool <- pool0 (below) */
        const int n1 = sh.pool.n, n0 = fr.pool0.n;
        int16_t *cur = GQ_CUR(sh, 0), *s1 = GQ_SNAP(sh, depth, 1);
        const int16_t *s0 = GQ_SNAP(sh, depth, 0);
        for (int i = tid; i < (n1 > n0 ? n1 : n0); i += B) {
            if (i < n1) s1[i] = cur[i];
            if (i < n0) cur[i] = s0[i];
        }
    }
    __syncthreads();                     /* sh.pool.n was read above: it changes now */
#endif
    if (tid == 128) {
#else
    if (tid < sh.n16) {
        SNAP_AT(sh, depth, 1)[tid] = ((const uint4 *) &sh.cb)[tid];
        ((uint4 *) &sh.cb)[tid] = SNAP_AT(sh, depth, 0)[tid];
    } else if (tid == 128) {
#endif
        fr.pool_lc = sh.pool;
        sh.pool = fr.pool0;
#if FC_GM
        if (fr.delta) sh.dpool.n = (unsigned short) fr.rn0;
#endif
    }
}

__device__ float tree_bits_dev(const Sh &sh, int ML, int child, int level, int which);

#ifdef FC_PM
#define PM0(sh) do { (sh).pm_t = wall_clock64(); } while (0)
#define PM(sh, i, g) do { if (FC_PM == (g)) { unsigned long long t_ = wall_clock64(); (sh).pm[i] += t_ - (sh).pm_t; (sh).pm_t = t_; } } while (0)
#else
#define PM0(sh) do { } while (0)
#define PM(sh, i, g) do { } while (0)
#endif
#include "mp_device.inc"

/* ------------------------------------------------------------------ serial state machine */

__device__ float tree_bits_dev(const Sh &sh, int ML, int child, int level, int which)
{
    const unsigned *counts = sh.tm + which * 2 * ML;
    float prob = counts[level] / (float) counts[ML + level];
    return child ? (float) -log2((double) prob) : (float) -log2((double) (1 - prob));
}

__device__ void tree_update_dev(Sh &sh, int ML, int child, int level, int which)
{
    unsigned *counts = sh.tm + which * 2 * ML;
    if (child) counts[level]++;
    counts[ML + level]++;
}

/* model snapshots: short runs of 128-bit LDS copies (lane 0) */
__device__ __forceinline__ void copy16(uint4 *dst, const uint4 *src, int n)
{
    /* 8 independent 128-bit LDS reads in flight, then 8 writes: a dependent read->write
     * chain per element would cost one LDS latency (~64 cycles) each */
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        /* named temporaries: an indexed local array ends up in scratch memory */
        uint4 t0 = src[i], t1 = src[i + 1], t2 = src[i + 2], t3 = src[i + 3];
        uint4 t4 = src[i + 4], t5 = src[i + 5], t6 = src[i + 6], t7 = src[i + 7];
        dst[i] = t0; dst[i + 1] = t1; dst[i + 2] = t2; dst[i + 3] = t3;
        dst[i + 4] = t4; dst[i + 5] = t5; dst[i + 6] = t6; dst[i + 7] = t7;
    }
    for (; i < n; i++) dst[i] = src[i];
}

__device__ void snap_save(const DevFrame &F, Sh &sh, int depth, int which)
{
    copy16(SNAP_AT(sh, depth, which), (const uint4 *) &sh.cb, sh.n16);
}

__device__ void snap_load(const DevFrame &F, Sh &sh, int depth, int which)
{
    copy16((uint4 *) &sh.cb, SNAP_AT(sh, depth, which), sh.n16);
}

#if FC_VARIANT_BIG
/* the resting coefficient model (sh.dcb) */
__device__ void snap_save_d(Sh &sh, int depth, int which)
{
    copy16(SNAP_AT(sh, depth, which), (const uint4 *) &sh.dcb, sh.n16);
}

__device__ void snap_load_d(Sh &sh, int depth, int which)
{
    copy16((uint4 *) &sh.dcb, SNAP_AT(sh, depth, which), sh.n16);
}
#endif

__device__ __forceinline__ void tm_save(Sh &sh, int depth, int ML, int which = 0)
{
    copy16(TM_AT(sh, depth, which, ML), (const uint4 *) sh.tm, TM_N16(ML));
}

__device__ __forceinline__ void tm_load(Sh &sh, int depth, int ML, int which = 0)
{
    copy16((uint4 *) sh.tm, TM_AT(sh, depth, which, ML), TM_N16(ML));
}

/* wfalib.c:152-180 */
__device__ float final_distribution_dev(const DevFrame &F, int s)
{
    float f = 0;
    int dom;
    for (int l = 0; l < 2; l++) {
        if ((dom = TREE(F, s, l)) != RANGE_) f += F.final_d[dom];
        for (int e = 0; (dom = INTO(F, s, l, e)) != NOEDGE; e++)
            f += WEIGHT(F, s, l, e) * F.final_d[dom];
    }
    return f / 2;
}

/* init_new_state (codec/subdivide.c:549-610): store the new state's rows; edge lists are
 * kept sorted by target like append_edge (codec/wfalib.c:233-275) */
#if !FC_VARIANT_BIG
/* The default build's form (<= 3 edges per label): table bases from LDS as global pointers, the
 * edge lists sorted in registers (a dynamically indexed private array lives in scratch memory),
 * the final distribution from the values at hand -- the terms' entries are read in one batch, not
 * found again one dependent read at a time through the rows just stored. */
__device__ void store_new_state(DevFrame &__restrict__ F, Sh &sh, SFrame &fr, int aux)
{
    const int s = sh.states, PA = sh.par.PA;
    GLOBAL_AS int16_t *const tree = (GLOBAL_AS int16_t *) sh.par.at_tree, *const into = (GLOBAL_AS int16_t *) sh.par.at_into;
    GLOBAL_AS int16_t *const posv = (GLOBAL_AS int16_t *) sh.par.pos, *const pool = (GLOBAL_AS int16_t *) sh.par.at_pool;
    GLOBAL_AS float *const weight = (GLOBAL_AS float *) sh.par.at_weight, *const fin = (GLOBAL_AS float *) sh.par.at_final;
    GLOBAL_AS uint16_t *const xs = (GLOBAL_AS uint16_t *) sh.par.at_x, *const ys = (GLOBAL_AS uint16_t *) sh.par.at_y;
    short p = -1;
#if FC_BLKEST
    if ((s & 63) == 0 && (s >> 6) <= NBLOCKMIN) sh.cum[s >> 6] = sh.pool.n;
#endif
    if (!aux && sh.pool.n < sh.pool.max_domains) {
        p = (short) sh.pool.n;
        pool[sh.pool.n++] = (short) s;
    }
    posv[s] = p;
    fr.rrange.into[0] = NOEDGE;
    fr.rrange.tree = s;
    int   tr[2], i[2][3];
    float w[2][3], fd_t[2], fd_e[2][3];
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const Range &ch = fr.child[l];
        tr[l] = ch.tree;
        bool live = true;
#pragma unroll
        for (int e = 0; e < 3; e++) {                     /* NOEDGE terminated; missing: sorts last */
            live = live && ch.into[e] != NOEDGE;
            i[l][e] = live ? (int) ch.into[e] : 0x7fff;
            w[l][e] = live ? ch.weight[e] : 0.0f;
        }
        /* ascending by target (append_edge, codec/wfalib.c:233-275; targets are distinct) */
#define CSWAP(a, b) if (i[l][a] > i[l][b]) { int ti = i[l][a]; i[l][a] = i[l][b]; i[l][b] = ti; float tw = w[l][a]; w[l][a] = w[l][b]; w[l][b] = tw; }
        CSWAP(0, 1) CSWAP(1, 2) CSWAP(0, 1)
#undef CSWAP
        /* every term's final distribution entry requested at once (missing terms: state 0) */
        fd_t[l] = fin[tr[l] != RANGE_ ? tr[l] : 0];
#pragma unroll
        for (int e = 0; e < 3; e++) fd_e[l][e] = fin[i[l][e] != 0x7fff ? i[l][e] : 0];
    }
    float f = 0;                                          /* wfalib.c:152-180 */
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const Range &ch = fr.child[l];
        tree[l * PA + s] = (short) tr[l];
        xs[l * PA + s] = (uint16_t) ch.x;
        ys[l * PA + s] = (uint16_t) ch.y;
        if (tr[l] != RANGE_) f += fd_t[l];
#pragma unroll
        for (int e = 0; e < 3; e++)
            if (i[l][e] != 0x7fff) {
                into[(l * 6 + e) * PA + s] = (short) i[l][e];
                weight[(l * 6 + e) * PA + s] = w[l][e];
                f += w[l][e] * fd_e[l][e];
            }
        const int ne = (i[l][0] != 0x7fff) + (i[l][1] != 0x7fff) + (i[l][2] != 0x7fff);
        into[(l * 6 + ne) * PA + s] = NOEDGE;
        /* y_column (codec/subdivide.c:560-567), see the general form below */
        if (sh.par.color) {
            int yc = 0;
#pragma unroll
            for (int e = 0; e < 3; e++) if (i[l][e] != 0x7fff && i[l][e] == fr.ny[l]) yc = 1;
            ((GLOBAL_AS uint8_t *) sh.par.at_ycol)[l * PA + s] = (uint8_t) yc;
        }
    }
    fin[s] = f / 2;
    /* the term lists op_append works from (slot 0 = tree child with weight 1 if any, then the edges
     * in stored order; unused slots: valid dummies), so that it need not read the rows back */
#pragma unroll
    for (int l = 0; l < 2; l++) {
        int m = 0;
        const int c = tr[l] != RANGE_;
        sh.gs_c[l] = c;
        int   gi[4]; float gw[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { gi[a] = 0; gw[a] = 0.0f; }
        if (c) { gi[0] = tr[l]; gw[0] = 1.0f; m = 1; }
#pragma unroll
        for (int e = 0; e < 3; e++)
            if (i[l][e] != 0x7fff) {
#pragma unroll
                for (int a = 0; a < 4; a++) if (a == m) { gi[a] = i[l][e]; gw[a] = w[l][e]; }
                m++;
            }
        sh.gs_n[l] = m;
#pragma unroll
        for (int a = 0; a <= MAXED; a++) { sh.gs_idx[l][a] = a < 4 ? gi[a < 4 ? a : 0] : 0; sh.gs_w[l][a] = a < 4 ? gw[a < 4 ? a : 0] : 0.0f; }
    }
    ((GLOBAL_AS uint8_t *) sh.par.at_los)[s] = (uint8_t) fr.rrange.level;
    ((GLOBAL_AS uint8_t *) sh.par.at_dtype)[s] = aux ? 0 : 2;
}
#else
__device__ void store_new_state(DevFrame &__restrict__ F, Sh &sh, SFrame &fr, int aux)
{
    const int s = sh.states;
    F.pos[s] = -1;
#if FC_GM
    if (!aux) gm_offer(F, sh, s);
#else
    if (!aux && sh.pool.n < sh.pool.max_domains) {
        F.pos[s] = (short) sh.pool.n;
        F.pool_states[sh.pool.n++] = (short) s;
#if FC_VARIANT_BIG
        if (sh.nslot == 5) sh.dpool.n = sh.pool.n;      /* one list, two sets of counters */
#endif
    }
#endif
    fr.rrange.into[0] = NOEDGE;
    fr.rrange.tree = s;
    for (int l = 0; l < 2; l++) {
        const Range &ch = fr.child[l];
        TREE(F, s, l) = (short) ch.tree;
        F.x[l * F.PA + s] = (uint16_t) ch.x;
        F.y[l * F.PA + s] = (uint16_t) ch.y;
        short si[MAXED + 1]; float sw[MAXED + 1];
        int ne = 0;
        for (int e = 0; ch.into[e] != NOEDGE; e++) {
            int pos = 0;
            while (pos < ne && si[pos] < ch.into[e]) pos++;
            for (int j = ne; j > pos; j--) { si[j] = si[j - 1]; sw[j] = sw[j - 1]; }
            si[pos] = ch.into[e]; sw[pos] = ch.weight[e];
            ne++;
        }
        for (int e = 0; e < ne; e++) { INTO(F, s, l, e) = si[e]; WEIGHT(F, s, l, e) = sw[e]; }
        INTO(F, s, l, ne) = NOEDGE;
        /* y_column (codec/subdivide.c:560-567).  The flag stays with the state ID when the
         * state is removed again (remove_states, codec/wfalib.c:283-309, does not clear it)
         * and the join states of a colour frame never set theirs: they show what an earlier,
         * removed state of the same ID left behind, and the stream writer reads it. */
        if (F.color) {
            int yc = 0;
            for (int e = 0; ch.into[e] != NOEDGE; e++) if (ch.into[e] == fr.ny[l]) yc = 1;
            F.ycol[l * F.PA + s] = (uint8_t) yc;
        }
#if FC_VARIANT_BIG
        if (F.frame_type)                              /* wfa->mv_tree, codec/subdivide.c:592 */
            for (int k = 0; k < 5; k++) F.mv[(k * 2 + l) * F.PA + s] = ch.mv[k];
#endif
    }
    F.final_d[s] = final_distribution_dev(F, s);
    F.level_of_state[s] = (uint8_t) fr.rrange.level;
    F.domain_type[s] = aux ? 0 : 2;
}
#endif

/* auxiliary state joining two band trees (codec/coder.c:803-833) */
__device__ int append_join_state(DevFrame &__restrict__ F, Sh &__restrict__ sh, int t0, int t1, int level)
{
    const int s = sh.states;
    if (s >= F.PA) { sh.failed = FC_ERR_CAPACITY; return 0; }
    TREE(F, s, 0) = (short) t0; TREE(F, s, 1) = (short) t1;
    for (int l = 0; l < 2; l++) {
        INTO(F, s, l, 0) = NOEDGE;
        F.x[l * F.PA + s] = 0; F.y[l * F.PA + s] = 0;
    }
    F.final_d[s] = final_distribution_dev(F, s);
    F.level_of_state[s] = (uint8_t) level;
    F.domain_type[s] = 0;
    F.pos[s] = -1;
#if FC_VARIANT_BIG
    if (F.frame_type) for (int k = 0; k < 10; k++) F.mv[k * F.PA + s] = 0;
#endif
    sh.states++;
    if (sh.states >= F.limit_states) { sh.failed = FC_ERR_STATES; return 0; }
    return 1;
}

__device__ void push_root(DevFrame &__restrict__ F, Sh &__restrict__ sh, int y_state)
{
    SFrame &r = sh.st[0];
    r.rg.x = r.rg.y = r.rg.image = r.rg.address = 0;
    r.rg.level = F.level; r.rg.tree = RANGE_;
    for (int i = 0; i < RANGE_E; i++) { r.rg.weight[i] = 0; r.rg.into[i] = 0; }
    r.rg.err = r.rg.tree_bits = r.rg.matrix_bits = r.rg.weights_bits = 0;
    r.max_costs = MAXCOSTS;
    r.y_state = y_state;
    r.phase = PH_ENTER;
#if FC_SPINE
    sh.spine_n = 0; sh.spine_use = 0;
    if (sh.band == 0) { for (int k = 0; k < 3; k++) sh.spine_t[k] = 0; for (int k = 0; k < 4; k++) sh.spine_c[k] = 0; }
#endif
#if FC_SPEC
    r.ckpt = 0;
#endif
#if FC_VARIANT_BIG
    r.rg.nd_tree_bits = r.rg.nd_weights_bits = r.rg.mv_tree_bits = r.rg.mv_coord_bits = 0; r.rg.prediction = 0;
    for (int i = 0; i < 5; i++) r.rg.mv[i] = 0;
    r.pred = sh.band == 0 ? F.pred_root : 0;     /* codec/coder.c:743-745,805-806 */
    r.delta = 0;
#endif
    sh.sp = 0;
}

/* a band of the frame is finished (codec/coder.c:738-833): record it, start the next one.
 * Returns 0 when the frame is complete (or has failed). */
__device__ int band_advance(DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    if (!sh.after_chroma) {
        const SFrame &r = sh.st[0];
        const Range &rg = r.rg;
        const int band = sh.band;
        if (band == 0) {
            F.costs = r.ret; F.err = rg.err; F.tree_bits = rg.tree_bits;
            F.matrix_bits = rg.matrix_bits; F.weights_bits = rg.weights_bits;
            F.root_state = rg.tree;
        } else {
            F.c_costs[band - 1] = r.ret; F.c_err[band - 1] = rg.err;
            F.c_tree_bits[band - 1] = rg.tree_bits; F.c_matrix_bits[band - 1] = rg.matrix_bits;
            F.c_weights_bits[band - 1] = rg.weights_bits;
        }
        if (sh.failed) return 0;
        if (rg.tree == RANGE_) { sh.failed = FC_ERR_NOROOT; return 0; }
        if (!F.color) return 0;
        sh.tree_band[band] = rg.tree;
        if (band == 1) {
            if (!append_join_state(F, sh, sh.tree_band[0], sh.tree_band[1], F.level + 1)) return 0;
            sh.tree_band[1] = sh.states - 1;             /* from here on: the Y+Cb state */
        }
        if (band == 2) {
            if (!append_join_state(F, sh, sh.tree_band[2], RANGE_, F.level + 1)) return 0;
            if (!append_join_state(F, sh, sh.tree_band[1], sh.states - 1, F.level + 2)) return 0;
            F.root_state = sh.states - 1;
            return 0;
        }
        sh.band = band + 1;
        if (band == 0) { sh.op = OP_CHROMA; sh.after_chroma = 1; return 1; }
    }
    sh.after_chroma = 0;
    sh.op = OP_NOP;
    push_root(F, sh, sh.tree_band[0]);
    return 1;
}

#if FC_SPEC
/* The chain raises `epoch` and then reads `busy`; a verifier counts itself into `busy` and then reads
 * `epoch`: a store followed by a load of ANOTHER word on each side (Dekker).  Release / acquire alone
 * order neither pair; a sequentially consistent fence between the two accesses does -- at least one of
 * the two sides then sees the other's write, so either the verifier drops the block or the chain waits
 * for it.  (Before round 4 this held only because both words share a cache line of FcSpecCtl.) */
#define SPEC_DEKKER_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent")
#define SPEC_TIMEOUT_TICKS 20000000ull      /* 0.2 s of the 100 MHz wall clock: then the chain does the block itself */
/* Chain, lane 0: consume the verdicts that have arrived, in block order.  `drain`: wait for all of
 * them (end of the frame); otherwise wait only while every checkpoint slot is taken.  Returns 0, or 1 +
 * the checkpoint slot the chain has to go back to.  (Called from the workgroup's operation loop, not from
 * the partition search: a call inside serial_advance() costs every one of its invocations the saving
 * and restoring of registers through scratch memory -- 7 % of a frame, measured.) */
__device__ __noinline__ int spec_poll(Sh &sh, bool drain)
{
    Sh::SpecLocal &sl = sh.sl;
    FcSpecCtl *c = sl.ctl;
    unsigned long long t0 = 0;
    if (sl.learn != 0.0f) { sl.mlc = sl.nlc ? 0.9f * sl.mlc + 0.1f * sl.learn : sl.learn; sl.nlc++; sl.learn = 0.0f; }
    while (sl.commit != sl.head) {
        const unsigned slot = sl.commit % FC_SPEC_W;
        const bool mine = (sl.spec_mask >> slot) & 1u;
        if (!mine) { sl.commit++; t0 = 0; continue; }     /* searched here anyway: whatever the verifier says */
        /* relaxed: the word guards no data (an acquire would drop the chain's L1 at every look; what a
         * return reads is the chain's own checkpoint, behind a fence of its own) */
        const unsigned v = __hip_atomic_load(&c->verdict[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 11) != sl.commit + 1) {                  /* not there yet */
            if (!drain && sl.head - sl.commit < FC_SPEC_W) {
                __hip_atomic_store(&c->committed, sl.commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0;
            }
            const unsigned long long now = wall_clock64();
            if (!t0) t0 = now;
            if (now - t0 < SPEC_TIMEOUT_TICKS) { __builtin_amdgcn_s_sleep(16); continue; }
            sl.t_wait += now - t0;
            sl.n_timeout++;                                /* no verifier in sight: the chain is not held up by it */
        } else {
            if (t0) { sl.t_wait += wall_clock64() - t0; t0 = 0; }
            if ((v & 3u) == 1u) {
                /* colour: every state the block's search appended (and removed again) had its y_column
                 * flags written -- zeros, in the luminance band (codec/subdivide.c:560-567) -- under the id it
                 * had in the chain's numbering; the flags outlive the states (codec/wfalib.c:283-309) and
                 * the stream shows them.  The verifier used ids of its own and wrote nothing: here, for as
                 * many ids as its search ever used. */
                if (sh.par.color) {
                    GLOBAL_AS uint8_t *yc = (GLOBAL_AS uint8_t *) sh.par.at_ycol;
                    const unsigned used = (v >> 2) & 63u, s0 = sl.sk[slot];
                    for (unsigned j = 0; j < used; j++) { yc[s0 + j] = 0; yc[(unsigned) sh.par.PA + s0 + j] = 0; }
                }
                sl.mlc = sl.nlc ? 0.9f * sl.mlc + 0.1f * sl.lin[slot] : sl.lin[slot]; sl.nlc++;
                sl.commit++; sl.n_confirmed++;
                continue;
            }
            sl.n_wrong++;
            /* take over the verifier's state -- unless the states its search appended do not fit below the
             * verifiers' ids any more: then the chain searches the block itself and runs out of ids the
             * ordinary way (FC_ERR_CAPACITY, the host stages the frame again with more) */
            if ((v & 3u) == 3u && sl.sk[slot] + ((v >> 2) & 63u) <= (unsigned) sh.cap)
                return (1 + (int) slot) | 0x100 | (int) (((v >> 2) & 63u) << 16) | (int) (((v >> 8) & 7u) << 24);
        }
        return 1 + (int) slot;
    }
    /* (verifiers that wait with a result the chain did not ask for -- a block it searched itself -- go on) */
    __hip_atomic_store(&c->committed, sl.commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
}
#endif

#if FC_SPEC
/* Chain, lane 0, end of the luminance band of a colour frame: the chroma bands number their states on
 * into the verifiers' id ranges.  No block search may start from here on (epoch), none may still be
 * running (a verifier looks at the epoch every few operations). */
__device__ __noinline__ void spec_luminance_done(Sh &sh)
{
    FcSpecCtl *c = sh.sl.ctl;
    sh.sl.epoch++;
    __hip_atomic_store(&c->epoch, sh.sl.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    SPEC_DEKKER_FENCE();
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&c->busy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (wall_clock64() - t0 > 100000000ull) { sh.failed = FC_ERR_INTERNAL; break; }     /* 1 s */
        __builtin_amdgcn_s_sleep(32);
    }
    sh.sl.on = 0; sh.sl.mode = 0;        /* no more guesses: the chroma bands are searched by the chain ... */
    sh.sl.chroma_tabs = c->n_tabs > c->n_blocks;      /* ... with tables from all the other workgroups (OP_CHROMA) */
}
#endif

/* One transition of the state machine per call; 1 = call again, 0 = sh.op holds the next parallel
 * operation (or OP_DONE).  One transition per call on purpose: as a loop inside one function the
 * compiler hoists every constant and LDS address of every phase into registers for the whole
 * loop, ~120 VGPRs, and the function then saves and restores 48 callee-saved registers through
 * scratch memory on every call (a memory round trip on the serial path, ~100 k times per frame). */
#ifndef FC_SERIAL_LOOP
#define FC_SERIAL_LOOP 0
#endif
/* sp / phase: the stack pointer and the phase of the frame on top of the stack, held by the caller
 * (in registers across the transitions of one serial_advance()); in memory the stack pointer is
 * current between calls of serial_advance(), the phase of a frame while it is not on top */
__device__ __forceinline__ int serial_step(DevFrame &__restrict__ F, Sh &__restrict__ sh, int &sp, int &phase)
{
    const int ML = sh.par.ML;
    {
#if FC_SPEC
        /* what this workgroup is: 0 one workgroup per frame (or a chain that guesses no more), 1 a chain
         * that guesses, 2 + floor a verifier.  Read afresh in every transition: a plain read is hoisted
         * out of the loop of transitions (FC_SERIAL_LOOP) and then lives in a register of its own for
         * the whole of serial_advance() -- one more callee-saved register to save and restore through
         * scratch memory per call, three calls per range */
        const int spec_mode = *(volatile int *) &sh.sl.mode;
#endif
        if (sp < 0) {
            sh.sp = sp;
#if FC_SPEC
            if (spec_mode == 1) {                /* end of the band: every verdict, then no more guesses */
                sh.op = OP_SPEC_CKPT; sh.a0 = 2;
                return 0;
            }
#endif
            const int more = band_advance(F, sh);        /* may push the root of the next band */
            sp = sh.sp; phase = sp >= 0 ? sh.st[sp].phase : 0;
            if (!more) { sh.op = OP_DONE; return 0; }
            if (sh.op == OP_CHROMA) return 0;
            return 1;
        }
        SFrame &fr = sh.st[sp];
#if defined(FC_PM) && FC_PM == 4
        /* developer micro-benchmark of the serial lane under the live load of the CU: every 1024th
         * transition, 64 dependent LDS reads / 64 dependent float adds / 64 independent LDS reads /
         * 64 dependent int ops; ticks (100 MHz) in pm[0..3], samples in pm[7] */
        if ((sh.pm_prev++ & 1023) == 0) {
            volatile int *chain = (volatile int *) sh.pixels;      /* scratch area of the block, unused here */
            int keep[8];
            for (int k = 0; k < 8; k++) keep[k] = chain[k];
            for (int k = 0; k < 8; k++) chain[k] = (k + 1) & 7;
            unsigned long long t0 = wall_clock64();
            int idx = 0;
            for (int k = 0; k < 64; k++) idx = chain[idx];
            unsigned long long t1 = wall_clock64();
            float a = __int_as_float(idx + 0x3f800000);
            for (int k = 0; k < 64; k++) a = a + 1.25f;
            asm volatile("" : "+v"(a));
            unsigned long long t2 = wall_clock64();
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 64; k++) sum += chain[k & 7];
            asm volatile("" : "+v"(sum));
            unsigned long long t3 = wall_clock64();
            int x = sum;
            for (int k = 0; k < 64; k++) x = x * 3 + 1;
            asm volatile("" : "+v"(x));
            unsigned long long t4 = wall_clock64();
            {   /* dependent global loads: 32 lines of the Gram table far apart (cold: HBM or L2),
                 * then the same 32 again (warm: L1/L2) */
                const volatile float *g = F.gram;
                const size_t stride = (size_t) F.P * 8 + 64;
                unsigned long long u0 = wall_clock64();
                size_t o = (size_t) (x & 1);
                for (int k = 0; k < 32; k++) { float v = g[o]; o = (size_t) (k + 1) * stride + (size_t) (__float_as_int(v) & 1); }
                unsigned long long u1 = wall_clock64();
                o = (size_t) (o & 1);
                for (int k = 0; k < 32; k++) { float v = g[o]; o = (size_t) (k + 1) * stride + (size_t) (__float_as_int(v) & 1); }
                unsigned long long u2 = wall_clock64();
                sh.pm[4] += u1 - u0; sh.pm[5] += u2 - u1; sh.pm[6] += o & 1;
            }
            for (int k = 0; k < 8; k++) chain[k] = keep[k];
            sh.pm[0] += t1 - t0; sh.pm[1] += t2 - t1; sh.pm[2] += t3 - t2; sh.pm[3] += t4 - t3;
            sh.pm[7] += 1; sh.pm[6] += (unsigned long long) ((x & 1) + (__float_as_int(a) & 1));
        }
#endif
#if defined(FC_PM) && FC_PM == 3
        { unsigned long long t_ = wall_clock64(); sh.pm[sh.pm_prev & 7] += t_ - sh.pm_t; sh.pm_t = t_; sh.pm_prev = phase; }
#endif
#ifdef FC_SERIAL_PROFILE
        {   /* developer profile: ticks per phase of the state machine (previous phase ends here) */
            unsigned long long t = wall_clock64();
            sh.tk_ph[sh.ph_prev] += t - sh.ph_t0; sh.ph_t0 = t; sh.ph_prev = phase;
        }
#endif
        switch (phase) {
        case PH_ENTER: {
            Range &rg = fr.rg;
            rg.into[0] = NOEDGE;
            rg.tree = RANGE_;
            fr.ret = MAXCOSTS;
            if (sh.failed || rg.level < 3) { fr.ret = MAXCOSTS; goto pop; }
            if (rg.x >= sh.par.width || rg.y >= sh.par.height) { fr.ret = 0; goto pop; }
#if FC_SPEC
            /* entry of a block of the largest block level: verdicts that have arrived, then the
             * checkpoint of this block (OP_SPEC_CKPT; the node is entered again afterwards) */
#endif
            fr.price = sh.par.price;
            if (sh.band) fr.price *= sh.par.chroma_decrease;
#if FC_VARIANT_BIG
            /* try_nd / try_mc (codec/subdivide.c:141-154) */
            fr.try_pred = 0;
            if (fr.pred && rg.level >= F.p_min && rg.level <= F.p_max) {
                if (F.frame_type == 0) fr.try_pred = 1;
                else if (rg.x + (int) width_of_level(rg.level) <= sh.par.width
                         && rg.y + (int) height_of_level(rg.level) <= sh.par.height) fr.try_pred = 2;
            }
            fr.pred_done = 0;
            fr.norm_first = 1; fr.norm_done = 0;     /* clear_norms_table, folded into the first update */
#endif
            phase = PH_AFTER_INIT;
            if (rg.level == sh.par.lc_max) {
                rg.address = rg.image = 0;
                sh.op = OP_INIT_RANGE; sh.a0 = rg.x; sh.a1 = rg.y;
#if FC_SPEC
                sh.a2 = (!sh.band || sh.sl.chroma_tabs) ? sh.blk++ : -1;      /* index of the block in the host's list (+ blocks per band) */
#endif
                return 0;
            }
            break;
        }
        case PH_AFTER_INIT: {
            Range &rg = fr.rg;
#if FC_SPEC
            /* a block of the largest block level, its tables done: verdicts that have arrived, then
             * the checkpoint of this block (OP_SPEC_CKPT; the phase is entered again afterwards) */
            if (spec_mode == 1 && rg.level == sh.par.lc_max && !sh.band && !fr.ckpt && sp > 0) {
                fr.ckpt = 1; sh.op = OP_SPEC_CKPT; sh.a0 = 1;        /* the operation sets ckpt = 2 if it takes a checkpoint */
                return 0;
            }
#endif
            /* A range that cannot be subdivided needs no model snapshots at all: a rejected
             * linear combination leaves every model untouched (codec/approx.c:264-268), an
             * accepted one is exactly the state to continue from, and the tree model is not
             * touched without children -- the reference's duplicate/restore pairs
             * (codec/subdivide.c:188-237,404-468) are no-ops for it. */
            fr.leaf = rg.level <= sh.lc_min && rg.level <= sh.par.lc_max;
#if FC_VARIANT_BIG
            if (fr.try_pred) fr.leaf = 0;       /* predict_range goes back to the entry models */
#endif
            /* the snapshots around a linear-combination search are taken by all lanes inside
             * OP_APPROX (snap_coop_*), not by this one */
            fr.coop = !fr.leaf && rg.level <= sh.par.lc_max;
            if (!fr.leaf && !fr.coop) {
                fr.pool0 = sh.pool;
                snap_save(F, sh, sp, 0);
                tm_save(sh, sp, ML);
#if FC_GM
                fr.rn0 = sh.dpool.n;
                gq_save(sh, 0, sp, 0);
#endif
#if FC_VARIANT_BIG
                if (sh.nslot == 5 && !fr.delta) {
                    fr.dpool0 = sh.dpool; snap_save_d(sh, sp, 2);
#if FC_GM
                    gq_save(sh, 1, sp, 2);
#endif
                }
#endif
            }
            fr.states = sh.states;
            for (int l = 0; l < 2; l++)                 /* codec/subdivide.c:167-173 */
                fr.ny[l] = (sh.band && fr.y_state != RANGE_) ? (int) TREE(F, fr.y_state, l) : RANGE_;
            phase = PH_AFTER_LC;
            if (rg.level <= sh.par.lc_max) {
                fr.lrange = rg;
                fr.lrange.tree = RANGE_;
#if FC_VARIANT_BIG
                fr.lrange.tree_bits = tree_bits_dev(sh, ML, 0, rg.level, 0);
#endif                              /* default build: priced by an idle lane of OP_APPROX (mp_tables, sh.tb) */
                fr.lrange.matrix_bits = 0;
                fr.lrange.weights_bits = 0;
#if FC_VARIANT_BIG
                fr.lrange.nd_tree_bits = 0; fr.lrange.nd_weights_bits = 0; fr.lrange.prediction = 0;
                fr.lrange.mv_tree_bits = fr.try_pred == 2 ? 1.0f : 0.0f;   /* mc allowed but not used */
                fr.lrange.mv_coord_bits = 0;
#endif
#if FC_SPINE
                {   /* Is this range the next one on the spine that was searched when its top node was entered
                     * (mp_wave.inc)?  It is iff the search has descended from the last range picked up into
                     * its FIRST child: models, dictionary and tree model are then what the spine's search saw
                     * (codec/subdivide.c:226-237 restores them before :303-310 recurses).  Anything else
                     * drops what is parked. */
                    int use = 0;
                    if (sh.spine_n) {
                        if (sp == sh.spine_next && sp - sh.spine_top < sh.spine_n && sh.st[sp - 1].label == 0) {
                            use = sp - sh.spine_top; sh.spine_next = sp + 1;
                        } else sh.spine_n = 0;
                    }
                    sh.spine_use = use;
                }
#endif
                sh.op = OP_APPROX;
                return 0;
            }
            fr.lincomb = MAXCOSTS;
            break;
        }
        case PH_AFTER_LC: {
            Range &rg = fr.rg;
            if (fr.leaf) {
                fr.subdiv = MAXCOSTS;
                phase = PH_DECIDE;
                break;
            }
            if (!fr.coop) {
#if FC_VARIANT_BIG
                fr.pool_lc = sh.pool;
                snap_save(F, sh, sp, 1);
#if FC_GM
                gq_save(sh, 0, sp, 1);
#endif
                sh.pool = fr.pool0;
                snap_load(F, sh, sp, 0);
#if FC_GM
                gq_load(sh, 0, sp, 0);
                if (fr.delta) sh.dpool.n = (unsigned short) fr.rn0;
#endif
#else
                /* a node above the largest block level: no linear combination has touched the
                 * models since the snapshot of its entry */
                fr.pool_lc = sh.pool;
#endif
            }
#if FC_SPEC
            if (fr.ckpt == 3) {                  /* the guess (spec_guess, end of OP_APPROX): the combination wins */
                fr.subdiv = MAXCOSTS;
                phase = PH_DECIDE;
                break;
            }
#endif
            if (rg.level > sh.lc_min) {
                Range z;
                z.x = z.y = z.image = z.address = z.level = 0; z.tree = 0;
                for (int i = 0; i < RANGE_E; i++) { z.weight[i] = 0; z.into[i] = 0; }
                z.err = z.tree_bits = z.matrix_bits = z.weights_bits = 0;
#if FC_VARIANT_BIG
                z.nd_tree_bits = z.nd_weights_bits = z.mv_tree_bits = z.mv_coord_bits = 0; z.prediction = 0;
                for (int i = 0; i < 5; i++) z.mv[i] = 0;
#endif
                fr.child[0] = z; fr.child[1] = z;
                fr.rrange = rg;
#if FC_VARIANT_BIG
                fr.rrange.tree_bits = tree_bits_dev(sh, ML, 1, rg.level, 0);
#else
                /* the tree model has not changed since the node's OP_APPROX priced both symbols
                 * (only finished children update it) */
                fr.rrange.tree_bits = rg.level <= sh.par.lc_max ? sh.tb[1] : tree_bits_dev(sh, ML, 1, rg.level, 0);
#endif
                fr.rrange.matrix_bits = 0;
                fr.rrange.weights_bits = 0;
                fr.rrange.err = 0;
#if FC_VARIANT_BIG
                /* codec/subdivide.c:257-271 */
                fr.rrange.mv_tree_bits = fr.try_pred == 2 ? 1.0f : 0.0f;
                fr.rrange.mv_coord_bits = 0;
                fr.rrange.nd_tree_bits = fr.try_pred == 1 ? tree_bits_dev(sh, ML, 1, rg.level, 1) : 0.0f;
                fr.rrange.nd_weights_bits = 0;
                fr.rrange.prediction = 0;
                fr.subdiv = (fr.rrange.tree_bits + fr.rrange.weights_bits + fr.rrange.matrix_bits
                             + fr.rrange.mv_tree_bits + fr.rrange.mv_coord_bits + fr.rrange.nd_tree_bits
                             + fr.rrange.nd_weights_bits) * fr.price;
#else
                fr.subdiv = (fr.rrange.tree_bits + fr.rrange.weights_bits + fr.rrange.matrix_bits) * fr.price;
#endif
                fr.label = 0;
                phase = PH_CHILD;
            } else {
                fr.subdiv = MAXCOSTS;
                phase = PH_DECIDE;
            }
            break;
        }
        case PH_CHILD: {
            const Range &rr = fr.rrange;
            const int label = fr.label;
            Range &ch = fr.child[label];
            ch.image = rr.image * 2 + label + 1;
            ch.address = rr.address * 2 + label;
            ch.level = rr.level - 1;
            ch.x = (rr.level & 1) ? rr.x : rr.x + label * (int) width_of_level(rr.level - 1);
            ch.y = (rr.level & 1) ? rr.y + label * (int) height_of_level(rr.level - 1) : rr.y;
            phase = PH_CHILD2;
            if (label && rr.level <= sh.par.lc_max && sh.states > fr.states && !sh.band) {
                sh.op = OP_IPIS_INCR; sh.a0 = ch.image; sh.a1 = ch.address; sh.a2 = ch.level;
                sh.a3 = fr.states;
                return 0;
            }
            break;
        }
        case PH_CHILD2: {
            float lim = fr.lincomb > fr.max_costs ? fr.max_costs : fr.lincomb;
            float remaining = lim - fr.subdiv;
            phase = PH_CHILD_RET;
            fr.ret = 0;
            if (remaining > 0) {
                if (sp + 1 >= FC_DEPTH) { sh.failed = FC_ERR_INTERNAL; break; }
                SFrame &cf = sh.st[sp + 1];
                cf.rg = fr.child[fr.label];
                cf.y_state = fr.ny[fr.label];
                cf.max_costs = remaining;
                cf.phase = PH_ENTER;
#if FC_SPEC
                cf.ckpt = 0;
#endif
#if FC_VARIANT_BIG
                cf.pred = fr.pred; cf.delta = fr.delta;
#endif
                fr.phase = phase; sp++; phase = PH_ENTER;   /* this frame rests: its phase goes to memory */
                break;                              /* child result arrives in fr.ret */
            }
            fr.ret = -1;                            /* marker: no recursion happened */
            break;
        }
        case PH_CHILD_RET: {
            const int label = fr.label;
            float lim = fr.lincomb > fr.max_costs ? fr.max_costs : fr.lincomb;
#if FC_VARIANT_BIG
            if (fr.try_pred == 2 && !fr.norm_done) {
                /* a child that was not searched gets its displacement table here
                 * (subdivide.c:311-315); then update_norms_table (:317-318) */
                const Range &c0 = fr.child[label];
                sh.op = OP_NORMS; sh.a0 = fr.rg.level; sh.a1 = fr.norm_first;
                sh.a2 = (fr.ret < 0 && c0.level >= F.p_min) ? c0.level : -1;
                sh.a3 = c0.x | (c0.y << 16);
                fr.norm_first = 0; fr.norm_done = 1;
                return 0;
            }
            fr.norm_done = 0;
#endif
            if (fr.ret >= 0) fr.subdiv += fr.ret;
            if (fr.subdiv >= lim) {
                fr.subdiv = MAXCOSTS;
                phase = PH_DECIDE;
                break;
            }
            const Range &ch = fr.child[label];
            fr.rrange.err          += ch.err;
            fr.rrange.tree_bits    += ch.tree_bits;
            fr.rrange.matrix_bits  += ch.matrix_bits;
            fr.rrange.weights_bits += ch.weights_bits;
#if FC_VARIANT_BIG
            fr.rrange.mv_tree_bits    += ch.mv_tree_bits;
            fr.rrange.mv_coord_bits   += ch.mv_coord_bits;
            fr.rrange.nd_weights_bits += ch.nd_weights_bits;
            fr.rrange.nd_tree_bits    += ch.nd_tree_bits;
            tree_update_dev(sh, ML, ch.tree != RANGE_, ch.level, 0);
            tree_update_dev(sh, ML, !ch.prediction, ch.level, 1);     /* subdivide.c:371-372 */
#else
            /* (the second tree model, codec/subdivide.c:371-372, prices nothing without prediction and is
             * not part of the default build's snapshots: not kept) */
            tree_update_dev(sh, ML, ch.tree != RANGE_, ch.level, 0);
#endif
            fr.label = label + 1;
            phase = fr.label < 2 ? PH_CHILD : PH_DECIDE;
            break;
        }
        case PH_DECIDE: {
            Range &rg = fr.rg;
#if FC_SPEC
            if (spec_mode >= 2 && sp == spec_mode - 2) {
                /* the verifier's block: all the chain needs to know is whether the combination wins
                 * (the branch `lincomb < subdiv` below) */
                sh.sl.verdict = (!sh.failed && !fr.leaf && fr.lincomb < MAXCOSTS && fr.lincomb < fr.subdiv) ? 1 : 2;
                /* 3: the subdivision wins (the last branch below).  The chain need not search the block
                 * again: it takes over this workgroup's state as it stands here (OP_SPEC_CKPT) */
                if (!sh.failed && !fr.leaf && fr.subdiv < MAXCOSTS && !(fr.lincomb < fr.subdiv)) sh.sl.verdict = 3;
                sh.op = OP_DONE;
                return 0;
            }
#endif
#if FC_VARIANT_BIG
            if (fr.try_pred && !fr.pred_done && !sh.failed) { phase = PH_PRED_BEGIN; break; }
#endif
            if (fr.leaf) {                       /* models are already what they have to be */
                if (fr.lincomb < MAXCOSTS) { rg = fr.lrange; fr.ret = fr.lincomb; }
                else fr.ret = MAXCOSTS;
                goto pop;
            } else if (fr.lincomb >= MAXCOSTS && fr.subdiv >= MAXCOSTS) {
                sh.pool = fr.pool0;
                snap_load(F, sh, sp, 0);
                tm_load(sh, sp, ML);
#if FC_GM
                gq_load(sh, 0, sp, 0);
                if (fr.delta) sh.dpool.n = (unsigned short) fr.rn0;
#endif
#if FC_VARIANT_BIG
                if (sh.nslot == 5 && !fr.delta) {
                    sh.dpool = fr.dpool0; snap_load_d(sh, sp, 2);
#if FC_GM
                    gq_load(sh, 1, sp, 2);
#endif
                }
#endif
                sh.states = fr.states;
                if (sh.flim > sh.states) sh.flim = sh.states & ~(GRAM_FB - 1);
                fr.ret = MAXCOSTS;
                goto pop;
            } else if (fr.lincomb < fr.subdiv) {
#if FC_SPEC
                if (fr.ckpt == 2 && spec_mode == 1) sh.sl.learn = fr.lincomb;    /* searched here, kept its combination */
#endif
                sh.pool = fr.pool_lc;
                snap_load(F, sh, sp, 1);
                tm_load(sh, sp, ML);
#if FC_GM
                gq_load(sh, 0, sp, 1);
                if (fr.delta) sh.dpool.n = (unsigned short) fr.rn0;
#endif
#if FC_VARIANT_BIG
                /* the linear combination left the resting models as they were at the entry */
                if (sh.nslot == 5 && !fr.delta) {
                    sh.dpool = fr.dpool0; snap_load_d(sh, sp, 2);
#if FC_GM
                    gq_load(sh, 1, sp, 2);
#endif
                }
#endif
                rg = fr.lrange;
                sh.states = fr.states;
                if (sh.flim > sh.states) sh.flim = sh.states & ~(GRAM_FB - 1);
                fr.ret = fr.lincomb;
                goto pop;
            } else {
                int aux = sh.band > 0 || rg.x + (int) width_of_level(rg.level) > sh.par.width
                          || rg.y + (int) height_of_level(rg.level) > sh.par.height;
#if FC_VARIANT_BIG
                /* with a second rle pool as delta pool a state that neither pool takes keeps no
                 * tables (codec/subdivide.c:571-583,607; the constant pool takes every state) */
#if FC_GM
                /* ... in general: a state that neither pool takes (without prediction the delta pool is the
                 * constant pool, which takes everything) */
                if (!(gm_accepts(sh.pool, sh.gm.pk[0]) || !F.pred_on || gm_accepts(sh.dpool, sh.gm.pk[1]))) aux = 1;
#else
                if (F.pred_on && sh.pool.n >= sh.pool.max_domains) aux = 1;
#endif
#endif
#if FC_SPEC
                /* (volatile: see spec_mode) */
                if (sh.states >= (sh.band ? sh.par.PA : *(volatile int *) &sh.cap)) { sh.failed = FC_ERR_CAPACITY; fr.ret = MAXCOSTS; goto pop; }
#else
                if (sh.states >= (sh.band ? sh.par.PA : sh.par.P)) { sh.failed = FC_ERR_CAPACITY; fr.ret = MAXCOSTS; goto pop; }
#endif
                store_new_state(F, sh, fr, aux);
                phase = PH_AFTER_APPEND;
                if (!aux) { sh.op = OP_APPEND; sh.a0 = sh.states; return 0; }
                break;
            }
        }
#if FC_SPEC
        case PH_SPEC_END: {                  /* a verifier's block has left the stack without a verdict (not reached:
                                              * it ends in PH_DECIDE of the block): the chain does the block itself */
            sh.sl.verdict = 2; sh.op = OP_DONE;
            return 0;
        }
#endif
        case PH_AFTER_APPEND: {
            sh.states++;
#if FC_SPEC
            if (sh.states - *(volatile int *) &sh.gap_shift >= sh.par.limit_states) sh.failed = FC_ERR_STATES;
#else
            if (sh.states >= sh.par.limit_states) sh.failed = FC_ERR_STATES;
#endif
            fr.rg = fr.rrange;
            fr.ret = fr.subdiv;
            goto pop;
        }
#if FC_VARIANT_BIG
        case PH_PRED_BEGIN: {                /* predict_range + nd_prediction, prediction.c:96-150,371-404 */
            Range &rg = fr.rg;
            const int il = sh.par.images_level, P = sh.par.P;
            float maxc = fr.lincomb > fr.subdiv ? fr.subdiv : fr.lincomb;
            if (maxc > fr.max_costs) maxc = fr.max_costs;
            fr.pred_done = 1;
            fr.pred_max = maxc;
            fr.rec_states = sh.states;
            /* what the recursion left behind */
            fr.pool_rec = sh.pool; fr.dpool_rec = sh.dpool;
            snap_save(F, sh, sp, 3); snap_save_d(sh, sp, 4); tm_save(sh, sp, ML, 1);
#if FC_GM
            gq_save(sh, 0, sp, 3); gq_save(sh, 1, sp, 4);
#endif
            /* back to the models of the entry */
            sh.pool = fr.pool0; sh.dpool = fr.dpool0;
            snap_load(F, sh, sp, 0); snap_load_d(sh, sp, 2); tm_load(sh, sp, ML, 0);
#if FC_GM
            gq_load(sh, 0, sp, 0); gq_load(sh, 1, sp, 2);
#endif
            sh.states = fr.states;
            if (sh.flim > sh.states) sh.flim = sh.states & ~(GRAM_FB - 1);
            if (fr.try_pred == 2) {          /* mc_prediction, prediction.c:262-289 */
                sh.op = OP_MC_SEARCH; sh.a0 = rg.level; sh.a1 = rg.x | (rg.y << 16);
                sh.a2 = (rg.level == F.p_min ? 1 : 0) | (rg.level > F.p_min && fr.norm_first ? 2 : 0);
                phase = PH_PRED_MC2;
                return 0;
            }
            {   /* the range's DC part in the DC format of the normal model */
                const float x = rg.level > il ? sh.par.ipis[(size_t) rg.image * P]
                                              : (rg.level == il ? sh.par.d5 : sh.par.d4)[(size_t) rg.address * P];
                const float y = sh.par.diag[(size_t) (rg.level - sh.par.gl0) * P];
                const int sym = rtob_dev(x / y, sh.par.dc_mant, sh.par.dc_range);
                const int cnt = sym < 0 ? 0 : (int) sh.cb.cnt[sym];      /* RPF_ZERO: see coeff_bits of the oracle */
                fr.nd_w = btor_fast(sym, sh.par.dc_mant, sh.par.dc_range);
                fr.nd_tbits = tree_bits_dev(sh, ML, 0, rg.level, 1);
                fr.nd_wbits = (float) (0.0 - log2((double) (cnt / (float) sh.cb.tot[0])));
#if FC_GM
                if (sh.gm.ck[0] == FC_CK_UNIFORM) fr.nd_wbits = (float) (sh.par.dc_mant + 1);     /* uniform_bits, codec/coeff.c:155-170 */
#endif
            }
            fr.pred_costs = fr.price * (fr.nd_wbits + fr.nd_tbits);
            phase = PH_PRED_GO;
            break;
        }
        case PH_PRED_MC2: {                  /* find_P_frame_mc done: vector in sh.mc (prediction.c:282-289) */
            fr.prange = fr.rg;
            fr.prange.mv[0] = (short) sh.mc.type; fr.prange.mv[1] = (short) sh.mc.fx; fr.prange.mv[2] = (short) sh.mc.fy;
            fr.prange.mv[3] = (short) sh.mc.bx; fr.prange.mv[4] = (short) sh.mc.by;
            fr.prange.mv_tree_bits = sh.mc.tree_bits; fr.prange.mv_coord_bits = sh.mc.bits;
            fr.nd_tbits = sh.mc.tree_bits; fr.nd_wbits = sh.mc.bits;      /* mvt, mvc kept for PH_PRED_DONE */
            fr.pred_costs = (fr.prange.mv_tree_bits + fr.prange.mv_coord_bits) * fr.price;
            phase = PH_PRED_GO;
            break;
        }
        case PH_PRED_GO: {
            if (fr.pred_costs < fr.pred_max) {
                if (fr.rec_states - fr.states > F.max_save || sp + 1 >= FC_DEPTH) { sh.failed = FC_ERR_INTERNAL; }
                else {
                    sh.op = OP_PRED_SETUP; sh.a0 = fr.rg.level; sh.a1 = fr.rg.address;
                    phase = PH_PRED_RECURSE;
                    return 0;
                }
            }
            /* no residual search: everything back as the recursion left it */
            sh.pool = fr.pool_rec; sh.dpool = fr.dpool_rec;
            snap_load(F, sh, sp, 3); snap_load_d(sh, sp, 4); tm_load(sh, sp, ML, 1);
#if FC_GM
            gq_load(sh, 0, sp, 3); gq_load(sh, 1, sp, 4);
#endif
            sh.states = fr.rec_states;
            fr.rg.prediction = 0;
            phase = PH_DECIDE;
            break;
        }
        case PH_PRED_RECURSE: {              /* subdivide (max_costs - costs, ..., NO, YES), :432-456 */
            SFrame &cf = sh.st[sp + 1];
            cf.rg = fr.rg;
            if (fr.try_pred == 2) for (int i = 0; i < 5; i++) cf.rg.mv[i] = fr.prange.mv[i];
            cf.rg.tree_bits = cf.rg.matrix_bits = cf.rg.weights_bits = 0;
            cf.rg.nd_tree_bits = cf.rg.nd_weights_bits = cf.rg.mv_tree_bits = cf.rg.mv_coord_bits = 0;
            cf.rg.image = 0; cf.rg.address = 0;
            cf.y_state = fr.y_state;
            cf.max_costs = fr.pred_max - fr.pred_costs;
            cf.phase = PH_ENTER;
            cf.pred = 0; cf.delta = 1;
            fr.phase = PH_PRED_RET;
            sp++; phase = PH_ENTER;
            break;
        }
        case PH_PRED_RET: {
            const float costs = fr.pred_costs + fr.ret;
            /* nd: only a subdivided residual counts (:460); mc: any (:329) */
            const int keep = !sh.failed && costs < fr.pred_max && (fr.try_pred == 2 || fr.prange.tree != RANGE_);
            fr.pred_costs = costs;
            sh.op = OP_PRED_FINISH; sh.a0 = keep;
            phase = PH_PRED_DONE;
            fr.label = keep;                 /* remembered for PH_PRED_DONE */
            return 0;
        }
        case PH_PRED_DONE: {
            Range &rg = fr.rg;
            if (fr.label) {                  /* use the prediction, prediction.c:460-485,152-180 */
                const int img = rg.image, adr = rg.address;
                const float mvt = fr.try_pred == 2 ? fr.nd_tbits : 0.0f, mvc = fr.try_pred == 2 ? fr.nd_wbits : 0.0f;
                rg = fr.prange;
                rg.image = img; rg.address = adr;
                if (fr.try_pred == 2) {      /* prediction.c:333-340 */
                    rg.mv_coord_bits = mvc; rg.mv_tree_bits = mvt;
                } else {
                    rg.nd_tree_bits += fr.nd_tbits;
                    rg.nd_weights_bits += fr.nd_wbits;
                    rg.into[0] = 0; rg.weight[0] = fr.nd_w; rg.into[1] = NOEDGE;
                }
                rg.prediction = 1;
                if (sh.flim > sh.states) sh.flim = sh.states & ~(GRAM_FB - 1);
                fr.ret = (rg.tree_bits + rg.matrix_bits + rg.weights_bits + rg.mv_tree_bits + rg.mv_coord_bits
                          + rg.nd_tree_bits + rg.nd_weights_bits) * fr.price + rg.err;
                goto pop;
            }
            sh.pool = fr.pool_rec; sh.dpool = fr.dpool_rec;
            snap_load(F, sh, sp, 3); snap_load_d(sh, sp, 4); tm_load(sh, sp, ML, 1);
#if FC_GM
            gq_load(sh, 0, sp, 3); gq_load(sh, 1, sp, 4);
#endif
            sh.states = fr.rec_states;
            {   /* columns of ids the residual search used are stale in older rows */
                const int lim = fr.states & ~(GRAM_FB - 1);
                if (sh.flim > lim) sh.flim = lim;
            }
            rg.prediction = 0;
            phase = PH_DECIDE;
            break;
        }
#endif
        }
        return 1;
    pop:
        if (sp > 0) {
            SFrame &pf = sh.st[sp - 1];
#if FC_VARIANT_BIG
            if (pf.phase == PH_PRED_RET) pf.prange = fr.rg;
            else
#endif
            pf.child[pf.label] = fr.rg;
            pf.ret = fr.ret;
        }
        sp--;
        if (sp >= 0) phase = sh.st[sp].phase;
    }
    return 1;
}

/* advance the partition search until a data-parallel operation is required */
#if FC_SERIAL_LOOP
__device__ __noinline__ void serial_advance(DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    int sp = sh.sp, phase = sp >= 0 ? sh.st[sp].phase : 0;
    while (serial_step(F, sh, sp, phase))
        ;
    sh.sp = sp;
    if (sp >= 0) sh.st[sp].phase = phase;
}
#else
/* One transition per out-of-line call on purpose (builds with machine LICM): as a loop inside one
 * function the compiler hoists every constant and LDS address of every phase into registers for the
 * whole loop, ~120 VGPRs, and the function then saves and restores 48 callee-saved registers
 * through scratch memory on every call. */
__device__ __noinline__ int serial_step_call(DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    int sp = sh.sp, phase = sp >= 0 ? sh.st[sp].phase : 0;
    const int r = serial_step(F, sh, sp, phase);
    sh.sp = sp;
    if (sp >= 0) sh.st[sp].phase = phase;
    return r;
}
__device__ __forceinline__ void serial_advance(DevFrame &__restrict__ F, Sh &__restrict__ sh)
{
    while (serial_step_call(F, sh))
        ;
}
#endif

#if FC_SPEC
/* Chain, all lanes: whose tables does block `blk` get?  The buffer a table worker has filled for it
 * (sh.tab_from = the states whose entries are good: what the worker saw, less what a return of the
 * chain has replaced since), or -- no worker got there in time -- the chain's own tables, from scratch. */
__device__ void spec_tables(DevFrame &__restrict__ F, Sh &__restrict__ sh, int blk)
{
    if (threadIdx.x == 0) {
        Sh::SpecLocal &sl = sh.sl;
        FcSpecCtl *c = sl.ctl;
        const unsigned b = (unsigned) blk % FC_SPEC_R;
        int from = -1;
        {   /* for the table workers: where the chain is, and which buffers it needs no more (those of the
             * blocks below the oldest one that still waits for its verdict; in the chroma bands none does) */
            unsigned oldest = (unsigned) blk;
            if (!sh.band && sl.commit != sl.head) oldest = sl.blkof[sl.commit % FC_SPEC_W];
            __hip_atomic_store(&c->tab_free, oldest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c->blk_cur, (unsigned) blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((unsigned) blk < c->n_tabs) {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                if (__hip_atomic_load(&c->tab_seq[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned) blk + 1) {   /* (fence below) */
                    unsigned S = __hip_atomic_load(&c->tab_s[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned te = __hip_atomic_load(&c->tab_epoch[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (sl.epoch - te > 32u) S = 0;
                    else for (unsigned e = te; e != sl.epoch; e++) if (sl.rb_s[e % 32u] < S) S = sl.rb_s[e % 32u];
                    from = (int) S < table_states(sh) ? (int) S : table_states(sh);
                    break;
                }
                if (wall_clock64() - t0 > c->tab_wait) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (from >= 0) {
            sh.par.ipis = (float *) (sl.tabs + (size_t) b * c->tab_stride);
            sh.par.d5 = sh.par.ipis + (size_t) F.NS * F.P;
            sh.tab_shared = 1; sl.n_tab_used++;
        } else {
            sh.par.ipis = F.ipis; sh.par.d5 = F.d5;
            sh.tab_shared = 0; sl.n_tab_missed++; from = 0;
        }
        sh.tab_from = from;
        take_acquire();                 /* the worker's entries, not this CU's stale lines (one lane; the barrier follows) */
    }
    __syncthreads();
}
#endif

#if FC_SPEC
/* Table worker (all lanes; returns when the chain is done).  Luminance band: workgroup `role` of the T
 * workers builds the tables of every T-th block of the host's list, ahead of the chain, for the states
 * the chain has published, into the buffer blk % FC_SPEC_R.  Chroma bands of a colour frame (block
 * indices from n_blocks on; dynamic: a verifier that has turned worker): blocks handed out one by one. */
__device__ __noinline__ void spec_worker(DevFrame &__restrict__ F, Sh &__restrict__ sh, unsigned role, unsigned T, bool dynamic)
{
    __shared__ int tw_x, tw_y, tw_go;
    __shared__ unsigned tw_e0;
    const int tid = threadIdx.x;
    FcSpecCtl *const c = F.spec;
    unsigned j = role - 1;                      /* lane 0's */
    const unsigned short *blocks = (const unsigned short *) ((const char *) c + c->off_blocks);
    if (tid == 0) { sh.band = 0; sh.gap_lo = sh.gap_hi = 0; sh.deadmask = 0; sh.sl.role = (int) role; sh.ystates = 0; }
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            int go = 1;
            const unsigned nb = c->n_blocks;
            bool have = false;                  /* dynamic: j is a block taken from tab_next */
            for (;;) {
                if (__hip_atomic_load(&c->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { go = 0; break; }
                if (!dynamic) {
                    const unsigned cur = __hip_atomic_load(&c->blk_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (j + 1 < cur) j += T;           /* the chain is past these (it may still wait for block cur - 1) */
                    if (j >= nb) dynamic = true;
                }
                if (dynamic) {
                    if (!__hip_atomic_load(&c->chroma_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || c->n_tabs <= nb) {
                        __builtin_amdgcn_s_sleep(64);
                        continue;
                    }
                    if (!have) { j = nb + atomicAdd(&c->tab_next, 1u); have = true; }
                    if (j >= c->n_tabs) { __builtin_amdgcn_s_sleep(64); continue; }       /* nothing left: wait for the end */
                }
                if (j < __hip_atomic_load(&c->tab_free, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + FC_SPEC_R
                    && __hip_atomic_load(&c->s_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                __builtin_amdgcn_s_sleep(32);
            }
            tw_go = go;
            if (go) {
                const unsigned b = j % FC_SPEC_R;
                __hip_atomic_store(&c->tab_seq[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                /* the epoch first: a return of the chain lowers s_pub before it raises the epoch */
                tw_e0 = __hip_atomic_load(&c->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned S = __hip_atomic_load(&c->s_pub, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned band = j / nb;
                if (band) S = c->ystates;                 /* the finished luminance dictionary */
                sh.band = (int) band; sh.states = (int) S; sh.ystates = (int) S;
                sh.par.ipis = (float *) ((char *) c + c->off_tabs + (size_t) b * c->tab_stride);
                sh.par.d5 = sh.par.ipis + (size_t) F.NS * F.P;
                tw_x = blocks[2 * (j % nb)]; tw_y = blocks[2 * (j % nb) + 1];
            }
        }
        __syncthreads();
        if (!tw_go) break;              /* (lane 0's acquire loads of epoch / s_pub have dropped this CU's L1) */
        op_init_range(F, sh, tw_x, tw_y, 0);
        WAVE_DRAIN();
        __syncthreads();
        if (tid == 0) {
            const unsigned b = j % FC_SPEC_R;
            publish_release();
            __hip_atomic_store(&c->tab_s[b], (unsigned) sh.states, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c->tab_epoch[b], tw_e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            WAVE_DRAIN();                       /* tab_s / tab_epoch before tab_seq */
            __hip_atomic_store(&c->tab_seq[b], j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dynamic) j = c->n_tabs + 1;     /* take the next one */
            else j += T;
        }
    }
}
#endif

#if FC_SPEC
/* Append helper h of the H helpers of a frame (all lanes; returns when the chain is done): the entries t with
 * (t / B) mod (H + 1) == h + 1 of every Gram row the chain publishes (FcSpecCtl.app_*, frame_coder.h).  F is the CHAIN's
 * descriptor, read only; of sh only what append_row_part looks at is set up. */
__device__ __noinline__ void spec_append_helper(DevFrame &__restrict__ F, Sh &__restrict__ sh, unsigned h, unsigned H)
{
    __shared__ int ah_go, ah_s;
    const int tid = threadIdx.x;
    FcSpecCtl *const c = F.spec;
    unsigned seen = 0;
    if (!c) return;                                  /* the launch speculates without its buffers: nothing to help with */
    if (tid == 0) { sh.gap_lo = sh.gap_hi = 0; sh.gap_shift = 0; sh.deadmask = 0; sh.band = 0; }
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            int go = 0;
            for (;;) {                               /* relaxed polls, ONE acquire once there is something to take */
                if (__hip_atomic_load(&c->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                    || __hip_atomic_load(&c->app_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                const unsigned q = __hip_atomic_load(&c->app_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q != seen) { seen = q; go = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (go) {
                take_acquire();                      /* the chain's rows, images, automaton and the descriptor: nothing stale */
                ah_s = c->app_s; sh.flim = c->app_flim;
                for (int l = 0; l < 2; l++) {
                    sh.gs_n[l] = c->app_n[l]; sh.gs_c[l] = c->app_c[l];
                    for (int e = 0; e <= MAXED; e++) { sh.gs_idx[l][e] = c->app_idx[l][e]; sh.gs_w[l][e] = c->app_w[l][e]; }
                }
            }
            ah_go = go;
        }
        __syncthreads();
        if (!ah_go) break;                           /* uniform */
        if (c->app_dbg != 1) append_row_part_ool(F, sh, ah_s, (int) h + 1, (int) H + 1);
        WAVE_DRAIN();
        __syncthreads();
        if (tid == 0) publish_release();
        if (tid == 0) __hip_atomic_fetch_add(&c->app_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#endif

/* basis states: images, Gram tables (codec/control.c:133-173); lane 0, a few hundred flops */
__device__ void basis_init(DevFrame &F, Sh &sh)
{
    const int nb = F.basis_states, il = F.images_level;
    for (int s = 0; s < nb; s++) {
        F.img[(size_t) s * F.NI] = F.final_d[s];
        if (il == 0) F.imgT[s] = F.final_d[s];
    }
    for (int l = 1; l <= il; l++)
        for (int s = 0; s < nb; s++)
            for (int i = 0; i < (1 << l); i++) {
                float v = image_elem(F, s, l, i);
                F.img[(size_t) s * F.NI + (1 << l) - 1 + i] = v;
                if (l == il) F.imgT[(size_t) i * F.P + s] = v;
#if FC_VARIANT_BIG
                if (l == il - 1 && F.gl0 < il) F.imgT4[(size_t) i * F.P + s] = v;
#endif
            }
    for (int q = 0; q < F.NL; q++)
        for (int s1 = 0; s1 < nb; s1++)
            for (int s2 = 0; s2 <= s1; s2++) {
                if (!F.domain_type[s2]) continue;
#if FC_VARIANT_BIG
                if (F.gl0 < il) {
                    gram_store(F, q, s1, s2, q == 0 ? gram_dot4(F, s1, s2) : q == 1 ? gram_dot(F, s1, s2)
                                                                   : gram_entry(F, q, s1, s2));
                    continue;
                }
#endif
                gram_store(F, q, s1, s2, q == 0 ? gram_dot(F, s1, s2) : gram_entry(F, q, s1, s2));
            }
    sh.states = nb;
}

#if FC_VARIANT_BIG
/* the same for a basis in DevFrame.bx (hundreds of states with edge lists of up to 33 entries): all lanes; a level
 * of the images / of the Gram tables needs the level below complete for ALL basis states (codec/control.c:205-258,
 * codec/ip.c:213-257) */
__device__ void basis_init_bx(DevFrame &F, Sh &sh)
{
    const int tid = threadIdx.x, il = F.images_level;
    const BxView V = bx_view(F);
    const int nb = V.nb;
    for (int s = tid; s < nb; s += B) {
        F.img[(size_t) s * F.NI] = F.final_d[s];
        if (il == 0) F.imgT[s] = F.final_d[s];
    }
    __syncthreads();
    for (int l = 1; l <= il; l++) {
        for (int k = tid; k < (nb << l); k += B) {
            const int s = k >> l, i = k & ((1 << l) - 1);
            const float v = image_elem_bx(F, V, s, l, i);
            F.img[(size_t) s * F.NI + (1 << l) - 1 + i] = v;
            if (l == il) F.imgT[(size_t) i * F.P + s] = v;
            if (l == il - 1 && F.gl0 < il) F.imgT4[(size_t) i * F.P + s] = v;
        }
        __syncthreads();
    }
    for (int q = 0; q < F.NL; q++) {
        for (int k = tid; k < nb * nb; k += B) {
            const int s1 = k / nb, s2 = k - s1 * nb;
            if (s2 > s1 || !F.domain_type[s2]) continue;
            float v;
            if (F.gl0 < il) v = q == 0 ? gram_dot4(F, s1, s2) : q == 1 ? gram_dot(F, s1, s2) : gram_entry_bx(F, V, q, s1, s2);
            else v = q == 0 ? gram_dot(F, s1, s2) : gram_entry_bx(F, V, q, s1, s2);
            gram_store(F, q, s1, s2, v);
        }
        __syncthreads();
    }
}
#endif

/*
 *  One workgroup per frame.  A launch may hold more frames than slabs (more than the chip runs at
 *  once, or than HBM holds): the first `nlend` frames own a slab each, the others borrow one --
 *  the hardware's workgroup dispatcher is the queue.  A workgroup that finishes hands its slab to
 *  a ring of free slabs (ring[ctr[1]++] = base); a borrower takes the next ticket (ctr[0]++) and
 *  waits for that entry.  With as many slabs as resident workgroups the entry is always there
 *  already: the borrower only became resident because another workgroup had left.  The borrower
 *  re-bases every slab pointer of its own descriptor (ptrmask: one bit per 8-byte word) onto the
 *  slab it got.  No tail of idle CUs waiting for the slowest of the first frames, no limit on the
 *  size of a launch from the 200 MB slabs.
 */
#if FC_SPEC
#define SPEC_STRIDE ((unsigned) ((sizeof(Sh) + 255) / 256 * 256))          /* bytes per checkpoint slot */
#define SPEC_SLOTS(ctl) ((char *) (ctl) + (sizeof(FcSpecCtl) + 255) / 256 * 256)
#endif

__global__ void __launch_bounds__(B, FC_WG_PER_CU)
#if FC_SPEC
/* G workgroups per frame: workgroup f * G is the chain of frame f (descriptor frames[f]), the G - 1
 * after it are its verifiers (descriptors vframes[f * (G - 1) ..]: the chain's with private
 * <sub-block, state> tables, scratch and state-id range).  G == 1: no speculation. */
FC_KERNEL(DevFrame *frames, DevFrame *vframes, unsigned G, unsigned n, unsigned H)
#else
FC_KERNEL(DevFrame *frames, unsigned nlend, unsigned long long *ring, unsigned *ctr, const unsigned *ptrmask,
          unsigned long long queue_wait_ticks, unsigned coopW)
#endif
{
    /* the LDS budget of the build: FC_WG_PER_CU workgroups share the 160 KB of a CU.  Beside Sh the kernel keeps a few
     * words (and, speculating builds, one SpecLocal): the big 256-thread build is at 81 736 of its 81 920 bytes */
#if FC_SPEC
    constexpr unsigned FC_LDS_EXTRA = sizeof(Sh::SpecLocal) + 64;
#else
    constexpr unsigned FC_LDS_EXTRA = 64;
#endif
    static_assert(sizeof(Sh) + FC_LDS_EXTRA <= (160u * 1024u) / FC_WG_PER_CU, "Sh outgrows the LDS share of a workgroup of this build");
    __shared__ Sh sh;
#if FC_SPEC
    __shared__ Sh::SpecLocal sl_keep;
    __shared__ unsigned task_seq, spec_slot, spec_used, spec_vid;
    __shared__ int task_go, spec_act, spec_bad;
    if (threadIdx.x == 0) spec_bad = 0;
    if (blockIdx.x >= n * G) {
        /* behind the n * G workgroups of the frames: H append helpers per frame, on the chain's descriptor (read only) */
        const unsigned k = blockIdx.x - n * G;
        spec_append_helper(frames[k / H], sh, k % H, H);
        return;
    }
    const unsigned role = blockIdx.x % G;
    DevFrame &F = role ? vframes[(blockIdx.x / G) * (G - 1) + role - 1] : frames[blockIdx.x / G];
    unsigned long long *const ring = nullptr;
#elif FC_VARIANT_BIG
    /* coopW > 1: that many workgroups per frame (FcCoop; nlend = frames, no queue).  The workgroups of a frame get
     * block ids that differ by multiples of 8: the same XCD where the dispatcher deals blocks round robin -- a
     * matter of speed only, the hand-offs are agent-scope release / acquire */
    const unsigned cW = coopW > 1 ? coopW : 1;
    unsigned fidx = blockIdx.x, member = 0;
    if (cW > 1) {
        fidx = (blockIdx.x / (8 * cW)) * 8 + blockIdx.x % 8;
        member = (blockIdx.x / 8) % cW;
        if (fidx >= nlend) return;
    }
    DevFrame &F = frames[fidx];
    if (member) { coop_helper(F, sh, member, cW); return; }
    if (threadIdx.x == 0) {
        sh.coopW = F.coop ? cW : 1; sh.coop_seq = 0;
        sh.coopD = cW > 1 && F.coop ? (int) F.coop->depth : 0; sh.coop_minsub = cW > 1 && F.coop ? F.coop->minsub : 0;
        sh.coop_ticks = cW > 1 && F.coop ? F.coop->done_ticks : 0;
    }
#else
    DevFrame &F = frames[blockIdx.x];
#endif
    const int tid = threadIdx.x;

#if !FC_SPEC
    if (ring && blockIdx.x >= nlend) {
        /* a frame without a slab: wait for the next free one and move the descriptor onto it */
        __shared__ unsigned long long got;
        if (tid == 0) {
            const unsigned t = atomicAdd(&ctr[0], 1u);
            unsigned long long b;
            /* Forward progress rests on an observation, not a promise of HIP: workgroups become
             * resident in blockIdx order, so every slab owner is resident (or done) before a
             * borrower spins here.  Should that ever not hold the wait is bounded (wall clock,
             * 100 MHz): the frame gives up with FC_ERR_QUEUE and the host encodes it again in a slab
             * of its own (complete_wave) -- a diagnostic and a retry instead of a hung GPU. */
            const unsigned long long t_give_up = wall_clock64() + queue_wait_ticks;
            while ((b = __hip_atomic_load(&ring[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                if (wall_clock64() > t_give_up) break;
                __builtin_amdgcn_s_sleep(32);
            }
            got = b;                                        /* (the acquire has dropped the CU's L1: nothing stale of the slab's earlier users) */
        }
        __syncthreads();
        if (got == 0) {                                     /* uniform: no slab arrived */
            if (tid == 0) { F.status = FC_ERR_QUEUE; F.states = 0; }
            return;
        }
        {
            unsigned long long *d = (unsigned long long *) &F;
            const unsigned long long lo = (unsigned long long) F.slab_base, span = F.slab_bytes;
            const unsigned long long delta = got - lo;
            unsigned long long v[(FC_DESC_WORDS + B - 1) / B];
#pragma unroll
            for (unsigned k = 0; k < (FC_DESC_WORDS + B - 1) / B; k++) {      /* read everything first: slab_base moves too */
                const unsigned w = k * B + tid;
                v[k] = w < FC_DESC_WORDS ? d[w] : 0;
            }
            __syncthreads();
#pragma unroll
            for (unsigned k = 0; k < (FC_DESC_WORDS + B - 1) / B; k++) {
                const unsigned w = k * B + tid;
                if (w < FC_DESC_WORDS && ((ptrmask[w >> 5] >> (w & 31)) & 1u) && v[k] - lo < span) d[w] = v[k] + delta;
            }
        }
        __threadfence();
        __syncthreads();
        __builtin_amdgcn_s_dcache_inv();                    /* the descriptor is read through the scalar cache */
    }
#endif

#if FC_SPEC
    if (tid == 0) {
        Sh::SpecLocal &sl = sh.sl;
        sh.gap_lo = sh.gap_hi = 0; sh.gap_shift = 0; sh.deadmask = 0;
        sh.cap = F.spec ? F.spec_cap : F.P;
        sl.ctl = F.spec; sl.slots = F.spec ? SPEC_SLOTS(F.spec) : nullptr;
        sl.role = (int) role; sl.on = F.spec != nullptr && G > 1;
        sl.mode = role == 0 && sl.on ? 1 : 0;
        sl.T = F.spec_T; sl.tabs = F.spec ? (char *) F.spec + F.spec->off_tabs : nullptr;
        sl.chroma_tabs = 0;
        sl.n_tab_used = sl.n_tab_missed = sl.n_adopted = 0;
        sl.app_H = role == 0 && F.spec && G > 1 ? F.spec->app_H : 0u; sl.app_min = F.spec ? F.spec->app_min : 0u;
        sl.app_seq = 0; sl.app_off = 0; sl.n_app_dealt = sl.t_app_wait = 0;
        for (int k = 0; k < 32; k++) sl.rb_s[k] = 0;
        sh.blk = 0; sh.tab_shared = 0; sh.tab_from = 0;
        sl.floor = 0; sl.head = sl.commit = 0; sl.spec_mask = 0; sl.nospec = 0; sl.epoch = 0;
        sl.verdict = 0; sl.abort = 0; sl.ops = 0; sl.mlc = 0.0f; sl.nlc = 0; sl.learn = 0.0f;
        sl.n_tasks = sl.n_confirmed = sl.n_wrong = sl.n_timeout = sl.n_inline = sl.t_wait = 0;
        sl_keep = sl;
    }
    if (role == 0) {
#endif
    if (tid < 10 && tid >= 1)
        sh.m0tab[tid] = (float) -log2((double) (1 - 1 / (float) (1 << tid)));
    if (tid == 0) {
        const int ML = F.ML;
        static const unsigned c0[22] = {20,17,15,10,5,4,3,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1};
        static const unsigned c1[22] = {1,1,1,1,1,1,1,1,1,2,3,5,10,15,20,25,30,35,60,60,60,60};
        sh.failed = 0;
        /* a staged frame may be encoded several times: start from clean counters */
        /* the roofline / profile counters are accumulated in LDS (a global read-modify-write
         * per call is a memory round trip of the serial lane) and stored once at the end */
        sh.cnt.bytes_mp = sh.cnt.bytes_img = sh.cnt.bytes_gram = 0;
        sh.cnt.n_mp = sh.cnt.n_steps = sh.cnt.n_blocks = sh.cnt.n_appends = sh.cnt.n_fulleval = 0;
        sh.cnt.t_mpA = sh.cnt.t_mpB = sh.cnt.n_blockevals = 0;
        F.trace_n = 0;
        for (int k = 0; k < 8; k++) F.dbg[k] = 0;
#ifdef FC_LATENCY_PROBE
        {   /* developer probe: dependent-load latency on a small global array (L2 resident)
             * and on a freshly stored one */
            volatile float *a = F.est;
            for (int k = 0; k < 64; k++) a[k] = (float) ((k * 7 + 3) & 63);
            __builtin_amdgcn_s_waitcnt(0);
            unsigned long long c0 = __builtin_readcyclecounter();
            int idx = 0;
            for (int k = 0; k < 64; k++) idx = (int) a[idx];
            unsigned long long c1 = __builtin_readcyclecounter();
            F.dbg[5] = (c1 - c0) / 64 + (idx & 0);
            /* store -> load of the same word */
            c0 = __builtin_readcyclecounter();
            for (int k = 0; k < 64; k++) { a[64] = (float) k; idx += (int) a[64]; }
            c1 = __builtin_readcyclecounter();
            F.dbg[6] = (c1 - c0) / 64 + (idx & 0);
        }
#endif
        /* rows of the basis states (input/basis.c:61-114, input/read.c:219-340) */
#if FC_VARIANT_BIG
        if (F.bx) {                          /* their edges stay in DevFrame.bx (bx_view) */
            const BxView V = bx_view(F);
            for (int s = 0; s < V.nb; s++) {
                F.final_d[s] = V.final_d[s];
                F.domain_type[s] = (uint8_t) V.dtype[s];
                F.level_of_state[s] = 0xff;
                for (int l = 0; l < 2; l++) { TREE(F, s, l) = RANGE_; INTO(F, s, l, 0) = NOEDGE; }
            }
        } else
#endif
        for (int s = 0; s < F.basis_states; s++) {
            F.final_d[s] = F.b_final[s];
            F.domain_type[s] = F.b_dtype[s];
            F.level_of_state[s] = 0xff;
            for (int l = 0; l < 2; l++) {
                TREE(F, s, l) = F.b_tree[s][l];
                for (int e = 0; e < 6; e++) {
                    INTO(F, s, l, e) = F.b_into[s][l][e];
                    WEIGHT(F, s, l, e) = F.b_weight[s][l][e];
                    if (F.b_into[s][l][e] == NOEDGE) break;
                }
            }
        }
        for (int w = 0; w < 2; w++)
            for (int l = 0; l < ML; l++) {
                int k = l < 22 ? l : 21;
                sh.tm[w * 2 * ML + l] = c1[k];
                sh.tm[w * 2 * ML + ML + l] = c0[k] + c1[k];
            }
        for (int i = 4 * ML; i < TM_WORDS; i++) sh.tm[i] = 0;
        /* rle pool over the usable basis states (domain-pool.c:632-676) */
        Pool &m = sh.pool;
        m.total = 0;
        for (int i = 0; i <= MAXED; i++) { m.count[i] = 1; m.total++; }
        m.n = 0; m.max_domains = (unsigned short) F.pool_max; m.y_index = 0;
        m.d0_index = 0; m.d0_yindex = 0; m.d0_n = 0;
#if FC_BLKEST
        sh.cum[0] = 0;
#endif
#if FC_GM
        /* alloc_domain_pool and the allocators behind it (codec/domain-pool.c:203-236) for the kinds of both sets */
        sh.gm.pk[0] = F.gm_pool[0]; sh.gm.pk[1] = F.pred_on ? F.gm_pool[1] : FC_PK_CONSTANT;
        sh.gm.ck[0] = F.gm_coeff[0]; sh.gm.ck[1] = F.gm_coeff[1];
        sh.gm.qa = 0; sh.gm.gq = F.gq; sh.gm.P = F.P;
        sh.gm.base = sh.gm.kept = 0.0f; sh.gm.lg1 = 0.0;
        sh.dpool = m;
        for (int set = 0; set < 2; set++) {
            Pool &pm = set ? sh.dpool : sh.pool;
            const int k = sh.gm.pk[set];
            int maxd = F.pool_max ? F.pool_max : 1;                 /* "Using at least DC component.", :221-226 */
            if (k == FC_PK_BASIS) maxd = F.basis_states;
            if (k == FC_PK_UNIFORM || k == FC_PK_CONSTANT) maxd = 0xffff;
            pm.max_domains = (unsigned short) maxd;
            if (!GM_RLE(k)) { pm.total = 0; for (int i = 0; i <= MAXED; i++) pm.count[i] = 0; }
        }
        for (int s = 0; s < F.basis_states; s++) {
            F.pos[s] = -1;
            if (F.domain_type[s] & 2) gm_offer(F, sh, s);
        }
        if ((F.domain_type[0] & 2) && F.pos[0] < 0) { F.pos[0] = 0; F.pool_states[0] = 0; }   /* no pool keeps a list */
        if (sh.gm.pk[0] == FC_PK_CONSTANT) sh.pool.n = 1;          /* the list {0} */
        if (sh.gm.pk[1] == FC_PK_CONSTANT) sh.dpool.n = 0;
#else
        for (int s = 0; s < F.basis_states; s++) {
            F.pos[s] = -1;
            if ((F.domain_type[s] & 2) && m.n < m.max_domains) {
                F.pos[s] = (short) m.n;
                F.pool_states[m.n++] = (short) s;
                if (s == 0) m.d0_n = 1;
            }
        }
#endif
        /* aac model, all-ones (coeff.c:297-310) */
        sh.n16 = (32 + 2 * F.coeff_size + 15) / 16;
#if FC_VARIANT_BIG
        if (F.pred_on && (32 + 2 * F.d_coeff_size + 15) / 16 > sh.n16) sh.n16 = (32 + 2 * F.d_coeff_size + 15) / 16;
        sh.nslot = F.pred_on ? 5 : 2;
        /* snapshots that outgrow LDS (wide level window x many mantissa symbols) live in HBM; with
         * prediction (5 slots per depth, deeper stack) always: aac snapshots in the first part of
         * the area, tree-model snapshots behind them */
        sh.snap = (F.pred_on || (F.level - F.lc_min + 3) * 2 * sh.n16 > SNAP_POOL16) && F.snap_hbm
                  ? (uint4 *) F.snap_hbm : sh.snap_pool;
        sh.snap_tm_p = F.pred_on && F.snap_hbm ? (uint4 *) F.snap_hbm + FC_DEPTH * 5 * FC_N16MAX : (uint4 *) sh.snap_tm;
        sh.pred_active = 0; sh.pred_lo = sh.pred_rec = 0;
#endif
        {
            const int depth_need = F.level - F.lc_min + 2
#if FC_VARIANT_BIG
                                   + (F.pred_on ? F.p_max - F.lc_min + 2 : 0)
#endif
                                   ;
#if FC_VARIANT_BIG
            const bool snap_lds = sh.snap == &sh.snap_pool[0];
            const bool tm_lds = (const void *) sh.snap_tm_p == (const void *) &sh.snap_tm[0];
#else
            const bool snap_lds = true, tm_lds = true;
#endif
#if FC_VARIANT_BIG
            const int snap_need = (F.level - F.lc_min + 3) * 2 * sh.n16;
#else
            const int snap_need = (F.level - F.lc_min + 3 + F.lc_max - F.lc_min) * sh.n16;
            sh.par.snap_b1 = (F.level - F.lc_min + 3) - (F.level - F.lc_max);
#endif
            if ((snap_lds && snap_need > SNAP_POOL16)
                || (tm_lds && (F.level - F.lc_min + 3) * 4 * TM_N16(ML) > SNAP_TM_WORDS) || F.coeff_nt > 16
                || (F.P + 63) / 64 > NBLOCKMIN            /* block minima of the general scan */
                || depth_need > FC_DEPTH
                || F.max_elements > FC_MAXE               /* term slots of the table ops */
                || (!FC_VARIANT_BIG && F.pred_on))        /* prediction needs the big build */
                sh.failed = FC_ERR_INTERNAL;
        }
        for (int i = 0; i < (FC_VARIANT_BIG ? FC_MAXCOEFF_BIG : FC_MAXCOEFF); i++) sh.cb.cnt[i] = 0;
        for (int i = 0; i < 16; i++) sh.cb.tot[i] = 0;
        for (int i = 0; i < F.coeff_size; i++) sh.cb.cnt[i] = 1;
        sh.cb.tot[0] = (short) F.dcs;
        for (int i = 1; i < F.coeff_nt; i++) sh.cb.tot[i] = (short) F.sy;
#if FC_VARIANT_BIG
        /* d_coeff (codec/coder.c:732-736) and the second rle pool over the same basis states */
        for (int i = 0; i < FC_MAXCOEFF_BIG; i++) sh.dcb.cnt[i] = 0;
        for (int i = 0; i < 16; i++) sh.dcb.tot[i] = 0;
        for (int i = 0; i < F.d_coeff_size; i++) sh.dcb.cnt[i] = 1;
        sh.dcb.tot[0] = (short) F.d_dcs;
        for (int i = 1; i < F.coeff_nt; i++) sh.dcb.tot[i] = (short) F.d_sy;
#if !FC_GM
        sh.dpool = sh.pool;
#endif
        sh.dq.rpf_mant = F.d_rpf_mant; sh.dq.dc_mant = F.d_dc_mant; sh.dq.sy = F.d_sy; sh.dq.dcs = F.d_dcs;
        sh.dq.rpf_range = F.d_rpf_range; sh.dq.dc_range = F.d_dc_range;
        sh.dq.half_nd = rtob_dev(0.5f, F.d_rpf_mant, F.d_rpf_range); sh.dq.half_dc = rtob_dev(0.5f, F.d_dc_mant, F.d_dc_range);
        if (F.pred_on && F.d_coeff_size > FC_MAXCOEFF_BIG) sh.failed = FC_ERR_INTERNAL;
#endif
#if FC_VARIANT_BIG
        if (F.bx) sh.states = F.basis_states;        /* tables: basis_init_bx below, all lanes */
        else
#else
        if (F.bx) sh.failed = FC_ERR_INTERNAL;       /* a long basis needs a big build (core_hip.cpp routes) */
#endif
        basis_init(F, sh);
        /* root range (codec/coder.c:738-745) */
        sh.flim = 0;
        sh.par.lc_max = F.lc_max; sh.par.width = F.width; sh.par.height = F.height;
        sh.par.limit_states = F.limit_states; sh.par.PA = F.PA; sh.par.P = F.P; sh.par.ML = F.ML;
        sh.par.price = F.price; sh.par.chroma_decrease = F.chroma_decrease;
        sh.par.gram = F.gram; sh.par.gram_ls = F.gram_ls; sh.par.diag = F.diag; sh.par.ipis = F.ipis; sh.par.pos = F.pos;
        sh.par.gcol = F.gcol;
        sh.par.d5 = F.d5; sh.par.d4 = F.d4;
        sh.par.at_tree = F.tree; sh.par.at_into = F.into; sh.par.at_pool = F.pool_states; sh.par.at_weight = F.weight;
        sh.par.at_final = F.final_d; sh.par.at_los = F.level_of_state; sh.par.at_dtype = F.domain_type;
        sh.par.at_ycol = F.ycol; sh.par.at_x = F.x; sh.par.at_y = F.y; sh.par.color = F.color;
        sh.par.l2_keys = F.l2_keys; sh.par.l2_vals = F.l2_vals; sh.par.l2_mask = F.l2_mask;
        sh.par.max_elements = F.max_elements; sh.par.rpf_mant = F.rpf_mant; sh.par.dc_mant = F.dc_mant;
        sh.par.sy = F.sy; sh.par.dcs = F.dcs; sh.par.gl0 = F.gl0; sh.par.images_level = F.images_level;
        sh.par.lc_min_opt = F.lc_min; sh.par.trace_on = F.trace != nullptr;
        sh.par.rpf_range = F.rpf_range; sh.par.dc_range = F.dc_range;
        sh.par.half_nd = rtob_dev(0.5f, F.rpf_mant, F.rpf_range); sh.par.half_dc = rtob_dev(0.5f, F.dc_mant, F.dc_range);
        sh.band = 0; sh.lc_min = F.lc_min; sh.after_chroma = 0; sh.ystates = 0;
        push_root(F, sh, RANGE_);
        sh.op = OP_NOP;                      /* first pass: no parallel op, just run the search */
    }
    if (F.color)                              /* calloc'ed in the reference (codec/wfa.h) */
        for (int i = tid; i < 2 * F.PA; i += B) F.ycol[i] = F.ycol0 ? F.ycol0[i] : (uint8_t) 0;
#if FC_VARIANT_BIG
    if (F.bx) { __syncthreads(); basis_init_bx(F, sh); }
#endif
#if FC_SPEC
    }
#endif
    /* per-op tick counters live in LDS: a private array indexed by `op` would be scratch */
    unsigned long long *tk = sh.tk;
    if (tid == 0) for (int k = 0; k < 16; k++) tk[k] = 0;
#ifdef FC_SERIAL_PROFILE
    if (tid == 0) { for (int k = 0; k < 8; k++) sh.tk_ph[k] = 0; sh.ph_prev = 0; sh.ph_t0 = 0; sh.tk_init[0] = sh.tk_init[1] = 0; for (int k = 0; k < 4; k++) sh.tk_apx[k] = 0; }
#endif
#ifdef FC_PM
    if (tid == 0) for (int k = 0; k < 8; k++) sh.pm[k] = 0;
#endif
    unsigned long long t_begin = wall_clock64();
    /* everything below is inlined into this one loop (a single call site per op keeps the
     * kernel argument visible to the compiler: DevFrame fields come through scalar loads and
     * table accesses are global_load, not flat_load through a generic reference) */
#if FC_SPEC
    const unsigned T = F.spec ? (unsigned) F.spec_T : 0u;
    if (role == 0 && F.spec && G > 1) {
        /* the rows of the basis states are complete: table workers may start */
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&F.spec->s_pub, (unsigned) sh.states, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (role >= 1 && role <= T) { spec_worker(F, sh, role, T, false); return; }
    for (;;) {          /* chain: once.  Verifier: once per block it verifies, until the chain is done. */
    if (role) {
        FcSpecCtl *const c = F.spec;
        __syncthreads();
        if (tid == 0) {
            unsigned t = atomicAdd(&c->next, 1u);
            int go = 1;
            for (;;) {                  /* relaxed polls (an acquire per look drops the CU's L1 every time), ONE acquire on a find */
                const unsigned q = __hip_atomic_load(&c->slot_seq[t % FC_SPEC_W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q == t + 1) { take_acquire(); break; }        /* nothing stale in this CU's L1: the slot, the rows of the states */
                /* the slot already holds a LATER block: the chain went back behind block t, dropped it and has
                 * come round to the slot again before anybody looked at it.  Waiting for it would be for ever. */
                if (q > t + 1) { t = atomicAdd(&c->next, 1u); continue; }
                if (__hip_atomic_load(&c->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { go = 0; break; }
                /* colour frame, luminance band done: nothing left to verify, the chroma bands' tables to build */
                if (__hip_atomic_load(&c->chroma_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { take_acquire(); go = 2; break; }
                __builtin_amdgcn_s_sleep(32);
            }
            task_seq = t; task_go = go;
        }
        __syncthreads();
        if (task_go != 1) break;                                  /* uniform */
        {   /* the chain's LDS state at the entry of the block */
            const uint4 *src = (const uint4 *) (SPEC_SLOTS(c) + (size_t) (task_seq % FC_SPEC_W) * SPEC_STRIDE);
            for (unsigned i = tid; i < sizeof(Sh) / 16; i += B) ((uint4 *) &sh)[i] = src[i];
        }
        __syncthreads();                /* the copy is complete (its loads were waited for by the LDS stores) before lane 0 looks at the slot again */
        if (tid == 0) {
            const unsigned task_epoch = sh.sl.epoch;              /* the chain's, as of the checkpoint */
            /* the slot was not taken for a later block while it was read, and the chain has not gone
             * back behind this block since */
            bool valid = __hip_atomic_load(&c->slot_seq[task_seq % FC_SPEC_W], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == task_seq + 1
                         && __hip_atomic_load(&c->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == task_epoch;
            sh.sl = sl_keep;
            sh.sl.epoch = task_epoch;
            const int S0 = sh.states, TB = F.spec_tb;
            if (valid && (S0 > TB || sh.sp < 0 || sh.sp >= FC_DEPTH)) valid = false;
            sh.sl.floor = sh.sp; sh.sl.verdict = 2; sh.sl.abort = valid ? 0 : 1; sh.sl.ops = 0; sh.sl.busy = 0;
            sh.sl.mode = 2 + sh.sp;
            if (sh.sp > 0) sh.st[sh.sp - 1].phase = PH_SPEC_END;     /* what the block's parent does should the block ever return */
            sh.gap_lo = S0; sh.gap_hi = TB; sh.gap_shift = TB - S0; sh.states = TB; sh.cap = TB + FC_SPEC_TEMPS;
            unsigned dm = 0;
            for (int k = 0; k < 32; k++) if (k * B >= S0 && (k + 1) * B <= TB) dm |= 1u << k;
            sh.deadmask = dm;
            sh.par.at_pool = F.pool_states;                       /* private pool list */
            sh.par.trace_on = 0;
            sh.par.color = 0;                                     /* no y_column flags from here: spec_poll */
            /* which of this workgroup's ids the search uses is read off their level entries afterwards */
            for (int k = 0; k < FC_SPEC_TEMPS; k++) F.level_of_state[TB + k] = 0;
            sh.op = valid ? OP_NOP : OP_DONE;
            if (valid) {
                atomicAdd(&c->busy, 1u);
                SPEC_DEKKER_FENCE();
                /* (the chain may have raised the epoch between the check above and this count: once more) */
                if (__hip_atomic_load(&c->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != task_epoch) {
                    atomicSub(&c->busy, 1u); valid = false; sh.sl.abort = 1; sh.op = OP_DONE;
                } else sh.sl.busy = 1;
            }
            if (valid && !sh.tab_shared) {
                /* the chain built this block's tables in its own memory: this workgroup builds them
                 * again in its own (otherwise they are in a buffer of the ring, complete for every
                 * state below gap_lo, and the states appended here add their entries behind gap_hi) */
                const Range &rg = sh.st[sh.sp].rg;
                sh.par.ipis = F.ipis; sh.par.d5 = F.d5;
                sh.op = OP_INIT_RANGE; sh.a0 = rg.x; sh.a1 = rg.y; sh.a2 = -1;
            }
        }
    }
#endif
    for (;;) {
        __syncthreads();
        const int op = sh.op;
        if (op == OP_DONE) break;
        unsigned long long t0 = wall_clock64();
#if FC_PRIO_ROTATE
        {
            /* Four frames share a CU, one wave of each per SIMD, and the instruction arbiter takes the OLDEST ready
             * wave: the frame whose workgroup arrived first ran 18 % faster than the one that arrived last (1.52 /
             * 1.61 / 1.70 / 1.79 s by block id / 256), and a launch lasts as long as its slowest frame.  The user
             * priority (s_setprio, above age in the arbitration) of the co-resident frames rotates with the wall
             * clock -- every frame is first, second, third and last a quarter of the time -- so that they finish
             * together.  What a frame computes does not depend on when its instructions issue. */
            const unsigned pr = ((unsigned) (t0 >> FC_PRIO_SHIFT) + (blockIdx.x >> 8)) & 3u;       /* wave uniform */
            if (pr == 0) __builtin_amdgcn_s_setprio(0);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
#endif
        switch (op) {
#if FC_SPEC
        case OP_SPEC_CKPT: {
            /* a0 = 1, a block of the largest block level with its tables done: the verdicts that have
             * arrived, then the checkpoint of this block -- the complete LDS state of the chain: what a
             * verifier starts from, and what the chain returns to if its guess about the block is wrong.
             * a0 = 2, end of the band: every verdict.  Either way a verdict may send the chain back. */
            FcSpecCtl *const c = sh.sl.ctl;
            if (tid == 0) {
                int act = 0;
                const int back = spec_poll(sh, sh.a0 == 2);
                if (back) {
                    act = (back & 0x100) ? 3 : 2; spec_slot = (unsigned) ((back & 0xff) - 1);
                    spec_used = (unsigned) (back >> 16) & 63u; spec_vid = (unsigned) (back >> 24) & 7u;
                    if (act == 3) {
                        /* the state to take over is what this block's verifier left: anything else is a bug */
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        const Sh *im = (const Sh *) (sh.sl.slots + (size_t) (FC_SPEC_W + spec_vid) * SPEC_STRIDE);
                        const int TB = im->gap_hi, S0 = im->gap_lo, m = im->states - TB;
                        int bad = 0;
                        if (S0 != (int) sh.sl.sk[spec_slot]) bad |= 1;
                        if (TB < F.spec_cap || TB + FC_SPEC_TEMPS > F.P || ((TB - F.spec_cap) % FC_SPEC_TEMPS)) bad |= 2;
                        if (m < 0 || m > FC_SPEC_TEMPS || m > (int) spec_used) bad |= 4;
                        if (im->sp < 1 || im->sp >= FC_DEPTH) bad |= 8;
                        if (im->sl.epoch != sh.sl.epoch) bad |= 16;
                        if (S0 + m > F.spec_cap) bad |= 32;
                        if (im->sl.verdict != 3 || im->failed) bad |= 64;
                        if (bad) { spec_bad |= bad; act = 2; }
                    }
                }
                else if (sh.a0 == 1) {
                    SFrame &fr = sh.st[sh.sp];
                    if (!sh.sl.nospec && fr.rg.level > sh.lc_min) { fr.ckpt = 2; act = 1; }
                } else if (sh.par.color && !sh.band && !sh.after_chroma) spec_luminance_done(sh);
                else { sh.sl.on = 0; sh.sl.mode = 0; }       /* a gray frame: it is over */
                spec_act = act;
            }
            __syncthreads();
            if (spec_act == 1) {
                const unsigned seq = sh.sl.head, slot = seq % FC_SPEC_W;
                /* (hand-off recipe at the top of the file: the waves drain, ONE lane releases -- this runs once per block
                 * of the largest level, 8 100 times per 4K frame; until round 6 every lane fenced twice here.)  The slot is
                 * marked "being written" with a write-through store that is complete before any of its new bytes exist */
                if (tid == 0) __hip_atomic_store(&c->slot_seq[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WAVE_DRAIN();
                __syncthreads();
                uint4 *dst = (uint4 *) (sh.sl.slots + (size_t) slot * SPEC_STRIDE);
                for (unsigned i = tid; i < sizeof(Sh) / 16; i += B) dst[i] = ((const uint4 *) &sh)[i];
                WAVE_DRAIN();                    /* the slot + every table row written so far: in L2 */
                __syncthreads();
                if (tid == 0) {
                    /* every table row of the states so far is complete and visible: table workers may use them */
                    publish_release();
                    __hip_atomic_store(&c->s_pub, (unsigned) sh.states, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&c->slot_seq[slot], seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh.sl.head = seq + 1; sh.sl.spec_mask &= ~(1u << slot); sh.sl.n_tasks++;
                    sh.sl.blkof[slot] = (unsigned) (sh.blk - 1);
                    sh.sl.sk[slot] = (unsigned) sh.states;
                }
            } else if (spec_act == 3) {
                /* A wrong guess whose verifier found the subdivision to win: instead of going back to the
                 * checkpoint and searching the block again, the chain takes over the verifier's state at the
                 * decision of the block -- models, stack, the block's children -- and the states its search
                 * has appended: their rows move from the verifier's ids (from TB on) to the chain's (from the
                 * block's state count on), references to them with them.  Same values as a search of the
                 * chain's own: same code on the same inputs. */
                if (tid == 0) sl_keep = sh.sl;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
                const uint4 *src = (const uint4 *) (sl_keep.slots + (size_t) (FC_SPEC_W + spec_vid) * SPEC_STRIDE);
                for (unsigned i = tid; i < sizeof(Sh) / 16; i += B) ((uint4 *) &sh)[i] = src[i];
                __syncthreads();
                const int TB = sh.gap_hi, S0 = sh.gap_lo, m = sh.states - TB, shift = TB - S0;
                __syncthreads();
                if (tid == 0) {
                    sh.sl = sl_keep;
                    sh.sl.nospec = 0;
                    sh.sl.commit = sh.sl.head;
                    sh.sl.rb_s[sh.sl.epoch % 32u] = (unsigned) S0;
                    __hip_atomic_store(&c->s_pub, (unsigned) S0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    /* Rows from S0 on are about to change under every verification in flight (all void: they
                     * started from later checkpoints).  Half-moved rows must not be searched -- a state's
                     * position in the pool and the counters of a later checkpoint need not agree: first the
                     * epoch, then wait until every search has seen it.  The verifier whose rows move waits
                     * for `adopting` to clear before it uses its ids again. */
                    __hip_atomic_store(&c->adopting, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    sh.sl.epoch++;
                    __hip_atomic_store(&c->epoch, sh.sl.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    SPEC_DEKKER_FENCE();
                    while (__hip_atomic_load(&c->busy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(8);
                    sh.sl.n_adopted++;
                    sh.gap_lo = sh.gap_hi = sh.gap_shift = 0; sh.deadmask = 0;
                    sh.cap = F.spec_cap;
                    sh.states = S0 + m;
                    sh.par.at_pool = F.pool_states; sh.par.color = F.color; sh.par.trace_on = 0;
                    if (!sh.tab_shared) { sh.par.ipis = F.ipis; sh.par.d5 = F.d5; }
                    sh.st[sh.sp - 1].phase = PH_CHILD_RET;               /* (the verifier's terminal phase) */
                    SFrame &fr = sh.st[sh.sp];
                    fr.states = S0;
                    for (int l = 0; l < 2; l++) {
                        Range &ch = fr.child[l];
                        if (ch.tree >= TB) ch.tree -= shift;
                        for (int e = 0; e < RANGE_E; e++) if (ch.into[e] >= TB) ch.into[e] = (short) (ch.into[e] - shift);
                    }
                    if (F.color) {                                       /* see spec_poll: flags of the ids the search used */
                        GLOBAL_AS uint8_t *yc = (GLOBAL_AS uint8_t *) F.ycol;
                        for (unsigned j = 0; j < spec_used; j++) { yc[S0 + j] = 0; yc[(unsigned) F.PA + S0 + j] = 0; }
                    }
                }
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                /* the rows */
                for (int j = 0; j < m; j++) {
                    const int sid = TB + j, did = S0 + j;
                    if (tid < 2) {
                        const int l = tid;
                        const int t = TREE(F, sid, l);
                        TREE(F, did, l) = (short) (t >= TB ? t - shift : t);
                        for (int e = 0; e < 6; e++) {
                            const int d = INTO(F, sid, l, e);
                            INTO(F, did, l, e) = (short) (d >= TB ? d - shift : d);
                            WEIGHT(F, did, l, e) = WEIGHT(F, sid, l, e);
                        }
                        F.x[l * F.PA + did] = F.x[l * F.PA + sid]; F.y[l * F.PA + did] = F.y[l * F.PA + sid];
                    } else if (tid == 2) {
                        F.final_d[did] = F.final_d[sid]; F.level_of_state[did] = F.level_of_state[sid];
                        F.domain_type[did] = F.domain_type[sid];
                        const short p = F.pos[sid];
                        F.pos[did] = p;
                        if (p >= 0) F.pool_states[p] = (short) did;      /* the chain's list (the verifier kept its own) */
                    }
                    if (!F.domain_type[sid]) continue;                   /* an auxiliary state: no tables (uniform) */
                    for (int i = tid; i < F.NI; i += B) F.img[(size_t) did * F.NI + i] = F.img[(size_t) sid * F.NI + i];
                    if (tid < 32) F.imgT[(size_t) tid * F.P + did] = F.imgT[(size_t) tid * F.P + sid];
                    for (int q = 0; q < F.NL; q++) {
                        const float *Gs = GRAM(F, q) + GROW(sid, F.P);
                        float *Gd = GRAM(F, q) + GROW(did, F.P);
                        for (int t = tid; t < S0; t += B) Gd[t] = Gs[t];
                        if (tid <= j) Gd[S0 + tid] = Gs[TB + tid];
                        if (tid == 0) F.diag[(size_t) q * F.P + did] = F.diag[(size_t) q * F.P + sid];
                    }
                }
                __threadfence();
                __syncthreads();
                if (tid == 0)            /* the verifier has its ids back */
                    __hip_atomic_store(&c->adopting, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else if (spec_act == 2) {
                const unsigned slot = spec_slot;
                if (tid == 0) sl_keep = sh.sl;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
                const uint4 *src = (const uint4 *) (sl_keep.slots + (size_t) slot * SPEC_STRIDE);
                for (unsigned i = tid; i < sizeof(Sh) / 16; i += B) ((uint4 *) &sh)[i] = src[i];
                __syncthreads();
                if (tid == 0) {
                    sh.sl = sl_keep;
                    sh.sl.nospec = 1;                     /* this block is searched here */
                    sh.sl.commit = sh.sl.head;            /* every verification in flight is void ... */
                    /* the rows of the states from here on will be written again: tables computed from them
                     * in this epoch or before count up to here only (spec_tables), and nothing beyond is
                     * offered to the table workers until the next checkpoint */
                    sh.sl.rb_s[sh.sl.epoch % 32u] = (unsigned) sh.states;
                    __hip_atomic_store(&c->s_pub, (unsigned) sh.states, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    sh.sl.epoch++;                        /* ... and its verifier should drop it */
                    __hip_atomic_store(&c->epoch, sh.sl.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    SPEC_DEKKER_FENCE();
                    /* ... before the search here appends a state: rows from this state count on, read by a
                     * search that has not looked at the epoch yet, would change under it */
                    while (__hip_atomic_load(&c->busy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(8);
                }
                __syncthreads();
                /* the checkpoint was taken with the block's tables done.  In a buffer of the ring they still
                 * are; the chain's own tables have been those of later blocks since: once more */
                if (!sh.tab_shared) op_init_range(F, sh, sh.st[sh.sp].rg.x, sh.st[sh.sp].rg.y, 0);
                /* Verdicts that do not come (every wait for one is bounded, but 0.2 s each): the verifiers are
                 * not resident, or too few for a chain this fast.  Three of them and this frame goes on
                 * without guesses -- one workgroup, as if it had no others. */
                if (tid == 0 && sh.sl.n_timeout >= 3) { sh.sl.on = 0; sh.sl.mode = 0; }
            }
            break;
        }
#endif
#if FC_SPEC
        case OP_INIT_RANGE:
            if (((sh.sl.on && sh.sl.T > 0) || sh.sl.chroma_tabs) && sh.sl.role == 0 && sh.a2 >= 0) spec_tables(F, sh, sh.a2);
            else if (tid == 0) { sh.tab_from = 0; if (sh.sl.role == 0) { sh.par.ipis = F.ipis; sh.par.d5 = F.d5; sh.tab_shared = 0; } }
            __syncthreads();
            op_init_range(F, sh, sh.a0, sh.a1, sh.tab_from);
            break;
#else
        case OP_INIT_RANGE: op_init_range(F, sh, sh.a0, sh.a1, 0); FC_DUP(OP_INIT_RANGE, op_init_range(F, sh, sh.a0, sh.a1, 0)); break;
#endif
        case OP_APPROX:     op_approx(F, sh); break;
        case OP_IPIS_INCR:  op_ipis(F, sh, sh.a0, sh.a1, sh.a2, sh.a3); FC_DUP(OP_IPIS_INCR, op_ipis(F, sh, sh.a0, sh.a1, sh.a2, sh.a3)); break;
        case OP_APPEND:     op_append(F, sh, sh.a0); FC_DUP(OP_APPEND, op_append(F, sh, sh.a0)); break;
        case OP_CHROMA:
            op_chroma_pool(F, sh);
#if FC_SPEC
            if (sh.sl.chroma_tabs) {                 /* uniform: the luminance dictionary is final and visible */
                __threadfence();
                __syncthreads();
                if (tid == 0) {
                    FcSpecCtl *c = sh.sl.ctl;
                    c->ystates = (unsigned) sh.ystates;
                    __hip_atomic_store(&c->chroma_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#endif
            break;
#if FC_VARIANT_BIG
        case OP_PRED_SETUP:  op_pred_setup(F, sh, sh.a0, sh.a1); break;
        case OP_PRED_FINISH: op_pred_finish(F, sh, sh.a0); break;
        case OP_NORMS:       op_norms(F, sh, sh.a0, sh.a1, sh.a2, sh.a3); break;
        case OP_MC_SEARCH:   op_mc_search(F, sh, sh.a0, sh.a1, sh.a2); break;
#endif
        default: break;                      /* OP_NOP */
        }
        __syncthreads();
#if FC_SPEC
        if (tid == 0 && role && (++sh.sl.ops & 1u) == 0
            && __hip_atomic_load(&F.spec->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sh.sl.epoch) {
            sh.sl.abort = 1; sh.op = OP_DONE;         /* the chain has gone back behind this block */
        } else
#endif
        if (tid == 0) {                       /* partition search, lane 0 */
            unsigned long long t1 = wall_clock64();
            tk[op] += t1 - t0;
#ifdef FC_SERIAL_PROFILE
            sh.ph_t0 = t1;
#endif
#if defined(FC_PM) && FC_PM == 3
            sh.pm_t = wall_clock64(); sh.pm_prev = 7;
#endif
            serial_advance(F, sh);
#if defined(FC_PM) && FC_PM == 3
            { unsigned long long t_ = wall_clock64(); sh.pm[sh.pm_prev & 7] += t_ - sh.pm_t; }
#endif
#ifdef FC_SERIAL_PROFILE
            { unsigned long long t = wall_clock64(); sh.tk_ph[sh.ph_prev] += t - sh.ph_t0; }
#endif
            tk[0] += wall_clock64() - t1;
        }
    }
#if FC_SPEC
    if (!role) break;
    __syncthreads();
    /* A search of a block the chain has dropped meanwhile may get here without having looked at the epoch
     * (it does every few operations): its result slot is the slot of a LATER block by now -- it must not
     * write there.  (A return raises the epoch long before the later block's own result can be written:
     * what still slips through between this look and the copy lands first and is overwritten.) */
    if (tid == 0 && !sh.sl.abort
        && __hip_atomic_load(&F.spec->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != sh.sl.epoch) sh.sl.abort = 1;
    __syncthreads();
    if (!sh.sl.abort && sh.sl.verdict == 3) {            /* uniform: this workgroup's state, for the chain to take over */
        /* (a buffer per verifier, behind the checkpoint slots: nobody else ever writes it, and this workgroup
         * not again before the chain has read it -- see the wait below) */
        uint4 *dst = (uint4 *) (SPEC_SLOTS(F.spec) + (size_t) (FC_SPEC_W + (role - T - 1)) * SPEC_STRIDE);
        for (unsigned i = tid; i < sizeof(Sh) / 16; i += B) dst[i] = ((const uint4 *) &sh)[i];
        __threadfence();                                 /* + the rows of the states the search has appended */
        __syncthreads();
    }
    if (tid == 0) {
        if (!sh.sl.abort) {
            /* (seq + 1) << 11 | verifier << 8 | ids the search used << 2 | verdict */
            unsigned used = 0;
            while (used < FC_SPEC_TEMPS && F.level_of_state[sh.gap_hi + (int) used] != 0) used++;
            __hip_atomic_store(&F.spec->verdict[task_seq % FC_SPEC_W],
                               ((task_seq + 1) << 11) | ((role - T - 1) << 8) | (used << 2) | (unsigned) sh.sl.verdict,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (sh.sl.busy) { __threadfence(); atomicSub(&F.spec->busy, 1u); }      /* its rows are written */
        if (!sh.sl.abort && sh.sl.verdict == 3) {
            /* the rows of the states this search appended wait under this workgroup's ids for the chain to
             * move them: no new search (it would write the same ids) before the chain has -- it raises the
             * epoch when it is done with them, as it does when it drops the block for another reason */
            while (__hip_atomic_load(&F.spec->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == sh.sl.epoch
                   && __hip_atomic_load(&F.spec->committed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= task_seq
                   && !__hip_atomic_load(&F.spec->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT))
                __builtin_amdgcn_s_sleep(16);
            while (__hip_atomic_load(&F.spec->adopting, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(16);
        }
    }
    }
    if (role) {
        if (task_go == 2) spec_worker(F, sh, role, T, true);
        return;
    }
    if (tid == 0 && sh.sl.ctl && G > 1) {
        FcSpecCtl *const c = sh.sl.ctl;
        /* verifiers that are still at a block the chain went back behind drop it; the others leave */
        __hip_atomic_store(&c->epoch, sh.sl.epoch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&c->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        c->n_tasks = sh.sl.n_tasks; c->n_confirmed = sh.sl.n_confirmed; c->n_wrong = sh.sl.n_wrong;
        c->n_timeout = sh.sl.n_timeout; c->n_inline = sh.sl.n_inline; c->t_wait = sh.sl.t_wait;
        c->n_tab_used = sh.sl.n_tab_used; c->n_tab_missed = sh.sl.n_tab_missed; c->n_adopted = sh.sl.n_adopted;
        c->n_app_dealt = sh.sl.n_app_dealt; c->t_app_wait = sh.sl.t_app_wait;
    }
#endif
    if (tid == 0) {
        F.t_serial = tk[0]; F.t_init = tk[OP_INIT_RANGE]; F.t_approx = tk[OP_APPROX];
        F.t_ipis = tk[OP_IPIS_INCR]; F.t_append = tk[OP_APPEND];
        F.t_total = wall_clock64() - t_begin;
        F.bytes_mp = sh.cnt.bytes_mp; F.bytes_img = sh.cnt.bytes_img; F.bytes_gram = sh.cnt.bytes_gram;
        F.n_mp = sh.cnt.n_mp; F.n_steps = sh.cnt.n_steps; F.n_blocks = sh.cnt.n_blocks;
        F.n_appends = sh.cnt.n_appends; F.n_fulleval = sh.cnt.n_fulleval;
        F.n_blockevals = sh.cnt.n_blockevals; F.t_mpA = sh.cnt.t_mpA; F.t_mpB = sh.cnt.t_mpB;
#if FC_VARIANT_BIG && !defined(FC_SERIAL_PROFILE) && !defined(FC_PM)
        /* the ops only this build has (ticks): chroma set-up, prediction set-up / finish, norms, motion search */
        F.dbg[2] = tk[OP_CHROMA]; F.dbg[3] = tk[OP_PRED_SETUP]; F.dbg[4] = tk[OP_PRED_FINISH];
        F.dbg[5] = tk[OP_NORMS]; F.dbg[6] = tk[OP_MC_SEARCH];
#endif
#ifdef FC_SERIAL_PROFILE
        for (int k = 0; k < 8; k++) F.dbg[k] = sh.tk_ph[k];
        F.dbg[0] = sh.tk_init[0]; F.dbg[7] = sh.tk_init[1];      /* d5 / ipis of init_range */
        /* OP_APPROX: tables, init, steps, finalize (replace the CHILD* phase slots) */
        F.dbg[3] = sh.tk_apx[0]; F.dbg[4] = sh.tk_apx[1]; F.dbg[5] = sh.tk_apx[2]; F.dbg[2] = sh.tk_apx[3];
#endif
    }
#ifdef FC_PM
    if (tid == 0) for (int k = 0; k < 8; k++) F.dbg[k] = sh.pm[k];
#elif FC_SPINE && !defined(FC_SERIAL_PROFILE) && !defined(FC_BLKEST_CHECK)
    if (tid == 0) {
        for (int k = 0; k < 3; k++) { F.dbg[2 * k] = sh.spine_t[k]; F.dbg[2 * k + 1] = sh.spine_c[k]; }
        F.dbg[6] = sh.spine_c[3];
        /* which SIMD runs wave 0 (the serial lane): histogram over the frames of a launch, 16 bits per SIMD
         * (HW_REG_HW_ID, bits 5:4) */
        F.dbg[7] = 1ull << (16 * ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3));
    }
#endif
#if FC_SPEC && !defined(FC_PM) && !defined(FC_SERIAL_PROFILE)
    if (tid == 0) { F.dbg[0] = tk[OP_SPEC_CKPT]; F.dbg[1] = 0; }   /* ticks of the chain in checkpoints, verdicts and returns */
#endif
    if (tid == 0) {
        /* per-band results and the root state were recorded by band_advance() */
#if FC_VARIANT_BIG && !FC_SPEC
        if (sh.coopW > 1 && F.coop) __hip_atomic_store(&F.coop->quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        F.states = sh.states;
        F.ystates_out = sh.band ? sh.ystates : sh.states;
        F.lc_min_out = sh.lc_min;
        F.status = sh.failed ? sh.failed : FC_OK;
#if FC_SPEC
        if (spec_bad) F.status = 128 + spec_bad;
#endif
    }
    if (F.pack_dst) {                         /* the automaton for the host writer, packed */
        const uint4 *src = (const uint4 *) F.pack_src;
        uint4 *dst = (uint4 *) F.pack_dst;
        const unsigned n16 = F.pack_bytes / 16;
        for (unsigned i = tid; i < n16; i += B) dst[i] = src[i];
    }
#if !FC_SPEC
    if (ring) {                               /* the slab is free for the next frame without one */
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const unsigned i = atomicAdd(&ctr[1], 1u);
            __hip_atomic_store(&ring[i], (unsigned long long) F.slab_base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
}

/* workgroups of this build a CU holds at once: what the build was compiled for (its launch bound
 * caps the registers) unless its LDS allows fewer.  The launcher sizes the frame queue and the
 * number of slabs by it.  (hipOccupancyMaxActiveBlocksPerMultiprocessor answered 1 and 2 for
 * builds that demonstrably run 4 and 5 workgroups per CU: not used.) */
extern "C" int FC_OCCUPANCY(void)
{
    const int by_lds = (int) (163840 / sizeof(Sh));
    return by_lds < FC_WG_PER_CU ? (by_lds < 1 ? 1 : by_lds) : FC_WG_PER_CU;
}

#if FC_SPEC
/* n frames, G workgroups each (all n * G must be resident at once: a chain whose verifiers are not
 * does their blocks itself after a bounded wait, see spec_poll) */
extern "C" void FC_LAUNCH(DevFrame *d_frames, DevFrame *d_vframes, unsigned n, unsigned G, unsigned H, hipStream_t stream)
{
    /* ... and H append helpers per frame behind them (FcSpecCtl.app_H of every frame of the launch; 0: none) */
    hipLaunchKernelGGL(FC_KERNEL, dim3(n * (G + H)), dim3(B), 0, stream, d_frames, d_vframes, G, n, H);
}
extern "C" unsigned FC_SPEC_SLOT_BYTES(void) { return SPEC_STRIDE; }
#if !FC_VARIANT_WIDE
extern "C" unsigned fc_spec_ctl_bytes(void) { return (unsigned) ((sizeof(FcSpecCtl) + 255) / 256 * 256); }
#endif
#else
extern "C" void FC_LAUNCH(DevFrame *d_frames, unsigned n, unsigned nlend, unsigned long long *ring, unsigned *ctr,
                          const unsigned *ptrmask, unsigned long long queue_wait_ticks, unsigned coopW, hipStream_t stream)
{
    /* coopW > 1 (big builds, no queue): coopW workgroups per frame, frames in groups of eight (kernel entry) */
    const unsigned grid = coopW > 1 ? (n + 7) / 8 * 8 * coopW : n;
    hipLaunchKernelGGL(FC_KERNEL, dim3(grid), dim3(B), 0, stream, d_frames, nlend, ring, ctr, ptrmask, queue_wait_ticks, coopW);
}
#endif
