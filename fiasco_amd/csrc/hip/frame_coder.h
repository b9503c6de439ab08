/*
 *  frame_coder.h -- device-side data layout of the MI355X frame coder (shared between the
 *  kernel and its host launcher).  One persistent workgroup encodes one frame; all tables
 *  of that frame live in one HBM slab described by DevFrame.
 *
 *  HBM layout per frame (P = pitch = capacity for states that own tables, PA = capacity of the
 *  automaton arrays; PA > P only for colour frames, whose chroma states are all auxiliary):
 *    gram   [NL][P][P] f32  symmetric <state,state> tables, level images_level..lc_max
 *                           (reference ip_states_state, codec/cwfa.h:86, is lower
 *                           triangular per level; stored full so that both the
 *                           Gram-Schmidt row sweep and the row append read contiguous rows)
 *           or [NL][P(P+1)/2 + P]: the lower triangle with packed rows, for the frames whose
 *                           full tables HBM cannot hold for every CU (4K): half the memory, the
 *                           sweep gathers the part of a row behind the diagonal from the column
 *    diag   [NL][P]    f32  Gram diagonal (matching-pursuit denominators)
 *    ipis   [NS][P]    f32  <range sub-block, state>, heap slot major
 *                           (reference ip_images_state, codec/cwfa.h:89, is state major)
 *    d5     [NA][P]    f32  level-images_level dots of the current block, address major
 *    img    [P][NI]    f32  state images levels 0..images_level (reference layout)
 *    imgT   [2^il][P]  f32  level-images_level slice of img, pixel major (coalesced dots)
 *    d4, imgT4             the same one level lower, only when min block level is 4
 *    num/den/est [P], ipdo [MAXED][P], used [P]   matching-pursuit scratch
 *    tree [2][PA] i16, into [2][6][PA] i16, weight [2][6][PA] f32, ...  automaton, SoA
 *  Nothing in the slab needs host-side initialisation: the kernel writes every cell before
 *  it reads it (the basis automaton travels inside DevFrame).
 */
#ifndef FRAME_CODER_H
#define FRAME_CODER_H
#include <stdint.h>

#define FC_MAXED    5
#define FC_BLOCK    256         /* threads per frame workgroup */
#define FC_MAXDEPTH 22          /* recursion depth bound: level <= 26, lc_min >= 6 */
#define FC_MAXDEPTH_BIG 32      /* big build: + the residual search of a predicted range */
#define FC_MAXDEPTH_NARROW 18   /* 256-thread default build: level <= 22 with block levels from 6 */
/* aac snapshot pool (16-byte units) of the default build, 256- and 512-thread variant */
#define FC_SNAP16_NARROW 480    /* (19 depths + 4 block levels with children) x 20 uint4 = 460 at level 22, CLI models */
#define FC_SNAP16_WIDE   840
/* tree-model snapshots (words) of the default build: depths x 2 x MAXLEVEL, rounded to 16 bytes */
#define FC_SNAPTM_NARROW 988    /* 19 depths x 13 uint4 (MAXLEVEL 26: a frame the stock limits accept, coded under the limits extension -- 1080p colour, BASELINE config 3 -- stays in the 256-thread build) */
#define FC_SNAPTM_WIDE   1092   /* 21 depths x 13 uint4 */
#define FC_MAXSAVE  512         /* states a prediction attempt can displace: 2^(12 - 4 + 1) */
#define FC_MAXCOEFF 224         /* int16 entries of the aac model kept in LDS: default build */
/* FC_HM: the "high mantissa" build of the kernel (RPF mantissas of 6 .. 8 bits, codec/options.c:510-553: up to
 * 512 symbols per context, 9 x 512 + 512 counters) -- the 512-thread big build with the coefficient models
 * and their log2 tables sized for that (one frame per CU has the LDS for it) */
#ifndef FC_HM
#define FC_HM 0
#endif
/* FC_GM: the "generic models" build (the FC_HM build + the other entries of the reference's model registries,
 * codec/domain-pool.c:188-236 and codec/coeff.c:97-131: `adaptive', `basis', `uniform', `rle-no-chroma' and
 * `constant' domain pools, `uniform' coefficients; no setter of fiasco.h reaches them, fiasco_amd_c_options_set_models
 * does).  The candidate scan is the list scan with scratch in HBM; prices and model updates follow the kind of the
 * ACTIVE model set (csrc/hip/mp_device.inc, `#if FC_GM'). */
#ifndef FC_GM
#define FC_GM 0
#endif
/* DevFrame.gm_pool[] / gm_coeff[]: fa_pool_kind / fa_coeff_kind of csrc/host/fa_host.h */
enum { FC_PK_ADAPTIVE = 0, FC_PK_CONSTANT = 1, FC_PK_BASIS = 2, FC_PK_UNIFORM = 3, FC_PK_RLE = 4, FC_PK_RLE_NO_CHROMA = 5 };
enum { FC_CK_ADAPTIVE = 0, FC_CK_UNIFORM = 1 };
#define FC_GQ_SLOTS (2 + FC_MAXDEPTH_BIG * 5)   /* DevFrame.gq: two current arrays + five snapshots per depth */
#define FC_MAXCOEFF_BIG_STD 640     /* big builds: 9 levels x 64 symbols + 64 (mantissas up to 5 bits) */
#define FC_MAXSYM_STD   64          /* symbols per context (mantissa <= 5) */
#define FC_MAXCOEFF_HM  5120        /* FC_HM: 9 levels x 512 symbols + 512 (mantissas up to 8 bits) */
#define FC_MAXSYM_HM    512
#define FC_MAXCOEFF_BIG (FC_HM ? FC_MAXCOEFF_HM : FC_MAXCOEFF_BIG_STD)
#define FC_MAXSYM       (FC_HM ? FC_MAXSYM_HM : FC_MAXSYM_STD)
/* uint4 (16-byte units) of an aac snapshot of the largest model: totals + counters */
#define FC_N16(coeffs)  ((32 + 2 * (coeffs) + 15) / 16)
#define FC_N16MAX       FC_N16(FC_MAXCOEFF_BIG)       /* 82, FC_HM: 643 */
#define FC_MAXBASIS 16          /* states of an initial basis that travels inside DevFrame (else DevFrame.bx) */
#define FC_BX_BYTES (64 * 1024)  /* room for DevFrame.bx in a slab: 818 basis states (12 lists of 6+1 ints each = 80 bytes) */
#ifndef FC_TRI_HOT
#define FC_TRI_HOT  8           /* states whose Gram columns the triangular layout also keeps as rows (DevFrame.gcol) */
#endif

enum { FC_OK = 1, FC_ERR_STATES = 2, FC_ERR_CAPACITY = 3, FC_ERR_NOROOT = 4, FC_ERR_INTERNAL = 5,
       FC_ERR_QUEUE = 6,      /* frame queue: no free slab arrived in time, encode again with a slab of its own */
       FC_ERR_COOP = 7 };     /* the helper workgroups of a frame did not answer in time (FcCoop) */
/* Several workgroups for the TABLE passes of one frame (big build, launches that leave the chip empty; round 4).
 * The <sub-block, state> tables of a block (init_range, codec/ip.c:72-154) and of a residual block
 * (codec/prediction.c:302-309) are a third of a predicted frame's time, and the subtrees below depth D of the block
 * do not depend on each other: workgroup m of the frame builds the level-5 dots and the level recursion of the
 * subtrees m, m + W, ... for ALL states; the frame's own workgroup (m = 0) then adds the top D levels.  One
 * hand-off each way per block: pixels + descriptor -> release -> seq; helpers: acquire, work, release -> done.
 * Same operations on the same operands in the same order per table entry: the values cannot differ. */
struct FcCoop {
    unsigned seq;            /* table builds published by the frame's workgroup */
    unsigned done;           /* arrivals of the helpers: W - 1 per build */
    unsigned quit;           /* the frame is finished */
    unsigned depth;          /* D: 2^D subtrees */
    int      level, from, to;
    int      minsub;         /* blocks whose subtrees have fewer than 2^minsub level-5 addresses are built the ordinary way */
    float   *ipis, *d5, *d4; /* the active table set (prediction swaps it) */
    unsigned long long done_ticks;   /* how long the frame waits for its helpers (100 MHz wall clock; host) */
    /* the block's pixels (floats) follow at byte FC_COOP_HDR */
};
#define FC_COOP_HDR 128
#define FC_COOP_WAIT_TICKS 3000000000ull        /* helpers give up after 30 s without work (100 MHz wall clock) */
#define FC_COOP_DONE_TICKS 500000000ull         /* the frame waits 5 s for its helpers, then fails with FC_ERR_COOP */
#define FC_QUEUE_WAIT_TICKS 60000000000ull      /* default bound of that wait: 10 minutes of the 100 MHz wall clock */

struct FcTrace;
typedef struct DevFrame {
    /* ---- parameters ---- */
    float    price;
    int      lc_min, lc_max, images_level, max_elements;
    int      maxe_live;    /* edges per label any state of this frame can have: max(max_elements, basis) */
    int      gl0;          /* lowest level with a Gram table: min(lc_min, images_level) */
    int      second_domain_block;   /* codec/approx.c:103-118 (cfiasco -z 2) */
    int      check_underflow, check_overflow, full_search;   /* :119-206,420 (cfiasco -z 3) */
    int      level, width, height;
    int      pool_max, limit_states, ML;
    int      rpf_mant, dc_mant;
    float    rpf_range, dc_range;
    int      P;            /* pitch / capacity of the per-state TABLES (states with images) */
    unsigned gram_ls;      /* floats per level of the Gram tables: P x P (full symmetric layout), or
                            * P (P + 1) / 2 + P (lower triangle with packed rows -- the FC_GRAM_TRI build of
                            * the kernel --, + one row of slack) */
    int      PA;           /* pitch / capacity of the automaton arrays (all states, PA >= P) */
    int      color;        /* 3 bands Y, Cb, Cr (codec/coder.c:775-800) */
    int      chroma_sparse; /* chroma bands: <sub-block, state> entries only for the states somebody reads (frame_coder.hip,
                            * chroma_need_block); 0: the full tables (FIASCO_AMD_CHROMA_FULL, tests) */
    int      chroma_cl_cap; /* tests (FIASCO_AMD_CLMAX): a chroma block with more needed states than this takes the full tables, as
                            * one with more than the build's FC_CLMAX (the capacity of Sh::cl) does; 0 = FC_CLMAX */
    int      chroma_max;   /* size of the chroma domain list (rle_chroma, domain-pool.c:854-879) */
    float    chroma_decrease;
    unsigned long long plane;   /* pixels per band plane in pix16 */
    int      NL, NS, NA, NI;
    int      coeff_size, coeff_nt, dcs, sy;
    int      basis_states;
    /* ---- initial basis automaton (codec/wfa.h:112-138 rows of the basis states) ---- */
    int16_t  b_tree[FC_MAXBASIS][2];
    int16_t  b_into[FC_MAXBASIS][2][6];
    float    b_weight[FC_MAXBASIS][2][6];
    float    b_final[FC_MAXBASIS];
    uint8_t  b_dtype[FC_MAXBASIS];
    /* ---- or, for a basis that does not fit the rows above (more than FC_MAXBASIS states, or labels with more than
     * MAXEDGES edges: data/medium.fco, data/large.fco), the reference's own memory image of the basis rows.  The
     * reference never checks MAXEDGES (codec/wfalib.c:253-273): the rows into[state][label][MAXEDGES + 1] lie back
     * to back in one block (codec/wfa.h:131-133, codec/wfalib.c:70-75), a label's sixth and later edges run on into
     * the row of the next label, every reader walks a list up to the first NO_EDGE -- so the edge list of (state,
     * label) is into[(2 state + label) * 6 ..] up to the terminator, up to 33 entries in large.fco.  Big kernel
     * builds only.  Words: [0] basis states nb, [1] row entries n = 12 nb + 12, [2..3] 0; float final[nb];
     * int domain_type[nb]; float weight[n]; int16 into[n].  null: the basis is in b_tree .. b_dtype. ---- */
    const int *bx;
    /* ---- FC_GM build: model kinds of the normal [0] and the delta [1] set; gq: the probability indices of the
     * quasi-arithmetic pools (qac_model_t.index, codec/domain-pool.c:259-274), FC_GQ_SLOTS arrays of P int16 -- [0]
     * and [1] the current ones of the two sets, the rest the snapshots of the partition search (model_duplicate);
     * lginv[n] = log2(1.0 / n) as the HOST's libm gives it (uniform_bits, :592-615), n <= limit_states ---- */
    int      gm_pool[2], gm_coeff[2];
    int16_t *gq;
    const double *lginv;
    /* ---- tables ---- */
    const int16_t *pix16;
    float   *gram, *diag, *ipis, *d5, *img, *imgT, *norms;
    float   *gcol;         /* triangular Gram tables only: [NL][FC_TRI_HOT][P], the COLUMNS of the first FC_TRI_HOT states
                            * as contiguous rows -- <d, h> for d > h at [q][h][d].  The state a matching-pursuit step
                            * chooses first is state 0 in 41 % and one of the first eight in 46 % of the searches
                            * (measured with the oracle); their sweep reads this row instead of one 4-byte gather
                            * (a whole HBM line) per candidate */
    float   *d4, *imgT4;   /* level images_level-1 twins of d5 / imgT (block levels down to 4) */
    float   *num, *den, *est, *ipdo;
    uint8_t *used;
    int16_t *tree, *into;
    float   *weight, *final_d;
    uint8_t *level_of_state, *domain_type;
    uint16_t *x, *y;
    uint8_t *ycol;         /* [2][PA] y_column of wfa_t, kept per state ID across removals */
    const uint8_t *ycol0;  /* what ycol starts from: the flags the previous frame of a colour stream
                            * left behind (fa_job.ycol_carry); null = zeros (fresh wfa_t) */
    void    *snap_hbm;     /* big build: home of the aac model snapshots when they outgrow LDS */
    /* ---- prediction (codec/prediction.c; big build only) ---- */
    int      pred_root;    /* `prediction' argument of the band-0 root subdivide() call */
    int      pred_on;      /* options.prediction || frame is not intra: second rle pool, ND section */
    int      frame_type;   /* 0 I, 1 P, 2 B */
    int      p_min, p_max; /* wi->p_min_level, p_max_level */
    int      d_rpf_mant, d_dc_mant;        /* delta coefficient model (d_rpf / d_dc_rpf) */
    float    d_rpf_range, d_dc_range;
    int      d_coeff_size, d_dcs, d_sy;
    float   *ipis_alt, *d5_alt, *d4_alt;   /* tables of the residual block of a predicted range */
    float   *pix_save;     /* [FC_PIXELS + FC_PIXELS/32] block pixels + norms while a residual is searched */
    /* rows of the states a prediction attempt displaces (store_state_data, prediction.c:502-565) */
    float   *sv_gram;      /* [max_save][NL][P] */
    float   *sv_img;       /* [max_save][NI + 48 + NL]  image row, imgT / imgT4 columns, diagonal */
    struct FcSavedRow *sv_auto;   /* [max_save] automaton rows */
    int      max_save;
    /* ---- motion compensation (codec/mwfa.c, codec/motion.c; P frames) ---- */
    const int16_t *past, *future;     /* reconstructed reference frames, planes like pix16 */
    float   *mc_fwd, *mc_bwd;         /* [p_max - p_min + 1][4 sr^2] displacement cost tables */
    int16_t *mv;                      /* [5][2][PA] type, fx, fy, bx, by of (state, label) */
    int16_t *pix_chroma;              /* [2][plane] chroma planes minus the luminance motion */
    int      search_range;
    int16_t *pool_states;
    int16_t *pos;          /* state -> position in the domain pool list, -1 = not a candidate */
    int     *hits;         /* [P] edge-target histogram for the chroma domain list */
    /* finished automaton, packed: at the end of the frame the workgroup copies its automaton
     * arrays (tree .. ycol, one contiguous part of the slab) to pack_dst, a per-launch buffer
     * that holds the automata of all frames back to back -- ONE device->host copy per launch,
     * and the next launch does not have to wait for it (core_hip.cpp).  pack_dst may be null. */
    const void *pack_src;
    void    *pack_dst;
    unsigned pack_bytes;
    /* ---- log2 of a float probability exactly as the HOST's libm gives it (core_hip.cpp) ----
     * open-addressing table of the arguments for which the device's log2 differs: key = bits of
     * the float (0 = empty slot), value = the host's double; null = no table */
    const unsigned *l2_keys;
    const double   *l2_vals;
    unsigned        l2_mask;
    /* ---- results ---- */
    int      status;
    int      states, root_state;
    FcCoop  *coop;         /* several workgroups per frame for the table passes (nullptr: one) */
    int      ystates_out;  /* states that own tables (gray / Y band) at the end: the host's capacity memory */
    float    costs, err, tree_bits, matrix_bits, weights_bits;      /* band 0 (gray / Y) */
    float    c_costs[2], c_err[2], c_tree_bits[2], c_matrix_bits[2], c_weights_bits[2];  /* Cb, Cr */
    int      lc_min_out;   /* min block level after the frame (codec/coder.c:785-797 ratchet) */
    /* ---- counters for the roofline model ---- */
    unsigned long long bytes_mp, bytes_img, bytes_gram;
    unsigned long long n_mp, n_steps, n_blocks, n_appends, n_fulleval;
    /* ---- time per phase in 100 MHz wall-clock ticks (lane 0) ---- */
    unsigned long long t_init, t_approx, t_ipis, t_append, t_serial, t_total;
    unsigned long long t_mpA, t_mpB, n_blockevals;
    unsigned long long dbg[8];          /* free-form developer counters (shader clock ticks) */
    /* ---- optional per-call trace (FIASCO_AMD_TRACE), compared with the oracle's ---- */
    struct FcTrace *trace;
    int      trace_cap, trace_n;
    /* ---- frame queue (core_hip.cpp): the slab the pointers above were computed for.  A launch
     * with more frames than slabs runs one workgroup per SLAB; each takes frames off a queue and
     * re-bases the slab pointers of the frame's descriptor onto its own slab. ---- */
    char    *slab_base;
    unsigned long long slab_bytes;
    /* ---- block-level speculation (FC_SPEC build of the kernel, see FcSpecCtl below): the frame is
     * served by spec_G workgroups that share its slab -- role 0 runs the partition search and
     * speculates, the others verify.  spec == null: one workgroup, no speculation. ---- */
    struct FcSpecCtl *spec;
    int      spec_role, spec_G;
    int      spec_tb;      /* verifier: first of the FC_SPEC_TEMPS state ids it may use for the states its
                            * subtree search appends (all tables of the slab, private index range) */
    int      spec_cap;     /* chain: state capacity left of the verifiers' index ranges */
    int      spec_T;       /* workgroups 1 .. spec_T of the frame are table workers, the rest verifiers */
} DevFrame;

#define FC_DESC_WORDS ((sizeof(DevFrame) + 7) / 8)      /* 8-byte words of a descriptor */

/*
 *  Block-level speculation.  At the CLI's geometry nine of ten blocks of the largest block level end
 *  as ONE linear combination, and everything the partition search does below such a block (20 of
 *  its 21 matching-pursuit calls on average) leaves no trace: models, tree model and state count go
 *  back to what the combination left (codec/subdivide.c:431-459).  The chain workgroup therefore
 *  runs only the combination of a block, assumes that it wins, and goes on to the next block; the
 *  complete search of the block -- from the SAME entry state, with the same code -- is done by a
 *  verifier workgroup in parallel.  Its verdict is exact: "the combination wins" confirms what the
 *  chain assumed, anything else takes the chain back to the checkpoint of that block, which it
 *  then searches itself.  The bytes of the stream cannot depend on any of this.
 *
 *  Per frame, in HBM: this control block, then FC_SPEC_W checkpoint slots of sizeof(Sh) bytes (the
 *  complete LDS state of the chain at the entry of a block: input of the verifier, and what the
 *  chain returns to after a wrong guess), then one result slot of the same size PER VERIFIER (index
 *  role - T - 1; a verifier that finds the subdivision of its block to win leaves its own LDS state
 *  there: the chain takes it over instead of searching the block again).
 *
 *  Table workers.  A third of what is left to the chain is init_range: the <sub-block, state> tables
 *  of the next block for every state of the dictionary (codec/ip.c:72-154, codec/subdivide.c:612-644).
 *  They are a function of the block's pixels and of the states alone -- not of the models -- and the
 *  entries of a state depend on older states only.  Workgroups 1 .. T of the frame therefore build the
 *  tables of the blocks AHEAD of the chain (the host lists the blocks in the order of the search),
 *  for the states the chain has published so far, into a ring of FC_SPEC_R buffers; the chain adds
 *  the entries of the states that have come since (and of those a return has replaced: the buffer is
 *  tagged with the epoch it was computed in) and hands the buffer on to the block's verifier with
 *  its checkpoint, which then needs no init_range of its own.
 */
#define FC_SPEC_W      16       /* checkpoints / verifications in flight per frame */
#define FC_SPEC_TEMPS  16       /* state ids per verifier: a block has 2 + 4 + 8 inner nodes below its root */
#define FC_SPEC_MAXG   8        /* workgroups per frame: chain + table workers + verifiers */
#define FC_SPEC_R      24       /* table buffers per frame: blocks waiting for their verdict + blocks ahead */
typedef struct FcSpecCtl {
    unsigned next;              /* verifiers: next sequence number to take */
    unsigned epoch;             /* bumped by the chain whenever it goes back: verifications in flight are void */
    unsigned done;              /* the chain has finished the frame */
    unsigned committed;         /* blocks below this sequence number have had their verdict looked at, or were not waited for (chain) */
    unsigned adopting;          /* the chain moves the rows of a verifier's states to its own ids: nobody searches, the verifier keeps off its ids */
    unsigned busy;              /* verifiers inside a block search (the chain waits for 0 before the chroma bands re-use their state ids) */
    unsigned slot_bytes;        /* size of a checkpoint slot */
    unsigned slot_seq[FC_SPEC_W];   /* seq + 1 once the checkpoint of block `seq` is complete, 0 while written */
    unsigned verdict[FC_SPEC_W];    /* (seq + 1) << 11 | verifier (role - T - 1) << 8 | state ids the search used << 2 | code:
                                     * 1 the combination wins, 2 anything else, 3 the subdivision wins and the verifier's
                                     * finished search waits in ITS result slot (frame_coder.hip, end of a verifier's task; spec_poll) */
    /* statistics (chain) */
    unsigned long long n_tasks, n_confirmed, n_wrong, n_timeout, n_inline, t_wait;
    unsigned long long n_tab_used, n_tab_missed;      /* blocks whose tables came from a worker / were not there in time */
    unsigned long long n_adopted;                     /* wrong guesses after which the chain took over the verifier's state */
    /* table workers */
    unsigned s_pub;             /* states whose table rows are complete and visible (chain, at its checkpoints) */
    unsigned tab_free;          /* blocks below this index need their table buffer no more (chain) */
    unsigned blk_cur;           /* the block the chain is at (chain) */
    /* chroma bands of a colour frame: their tables are a function of the pixels and of the finished
     * luminance dictionary alone -- once the chain has set chroma_ready, EVERY other workgroup of the
     * frame builds them (block list once more per band; blocks handed out by tab_next) */
    unsigned chroma_ready, tab_next, ystates;
    unsigned n_tabs;            /* blocks with tables from workers: n_blocks, or 3 n_blocks for a colour frame (host) */
    unsigned n_blocks;          /* entries of the block list (host) */
    unsigned tab_stride;        /* bytes per table buffer: ipis [NS][P], then d5 [NA][P] (host) */
    unsigned tab_wait;          /* ticks (100 MHz) the chain waits for a worker's tables before it builds them itself (host) */
    unsigned long long off_blocks, off_tabs;   /* byte offsets from this struct: block list (x, y as 2 x u16), buffers (host) */
    unsigned tab_seq[FC_SPEC_R];    /* block index + 1 once the buffer holds that block's tables */
    unsigned tab_s[FC_SPEC_R];      /* ... for the states below this */
    unsigned tab_epoch[FC_SPEC_R];  /* ... read in this epoch or later */
    /* Append helpers (round 6).  A third of what is left to the chain of a speculating 4K frame is the Gram row of every
     * state it appends (codec/control.c:48-131, codec/ip.c:184-260): entries t = 0 .. s of the row are independent --
     * each is a function of older rows, of the automaton rows of s and t and of the level-images_level images alone;
     * codec/ip.c:213-257 fixes the order of operations per ENTRY, not the order of entries -- and one CU's gather rate
     * bounds them.  app_H further workgroups of the frame (launched behind the chains / workers / verifiers of the
     * launch, frame_coder.hip spec_append_helper) take the entries t with (t / B) mod (app_H + 1) == h + 1 of every
     * row the chain publishes: term lists of the new state -> release -> app_seq; helpers: acquire, build, release ->
     * app_done.  The chain builds its own share meanwhile and waits for app_seq * app_H arrivals -- bounded (app_wait): a
     * frame whose helpers do not answer fails with FC_ERR_COOP and is searched again without helpers (a helper that
     * turned up later could otherwise write into a row the chain has re-made since).  Same operations per entry, so
     * the same values. */
    unsigned app_H;                 /* helpers of this frame (host; 0: none) */
    unsigned app_min;               /* rows with fewer entries than this are the chain's alone (host) */
    unsigned app_wait;              /* ticks (100 MHz) the chain waits for its helpers before it gives the frame up (host) */
    unsigned app_seq;               /* rows published by the chain */
    unsigned app_done;              /* arrivals of the helpers (cumulative) */
    unsigned app_off;               /* the chain has given up: helpers go home */
    unsigned app_dbg;               /* developer (FIASCO_AMD_SPEC_APPDBG): 1 helpers answer without building, 2 the chain builds every share again after them (host) */
    int      app_s, app_flim;       /* the row: state id; sh.flim of the chain */
    int      app_n[2], app_c[2], app_idx[2][FC_MAXED + 1];      /* Sh::gs_n, gs_c, gs_idx, gs_w of the new state */
    float    app_w[2][FC_MAXED + 1];
    unsigned long long n_app_dealt, t_app_wait;                  /* statistics (chain): rows dealt, ticks waited for helpers */
} FcSpecCtl;

/* automaton row of one state, as store_state_data keeps it */
typedef struct FcSavedRow {
    int16_t  tree[2], into[2][6];
    float    weight[2][6];
    float    final_d;
    uint16_t x[2], y[2];
    int16_t  pos;
    int16_t  mv[2][5];
    uint8_t  level, dtype, ycol[2], tables;      /* tables: table rows were saved too */
} FcSavedRow;

/* one record per approximate_range call; identical layout in oracle/oracle_core.c */
typedef struct FcTrace {
    int   seq, level, image, D, states, nedges;
    float cost, err, mbits, wbits;
    short into[6];
    float w[5];
} FcTrace;

#endif
