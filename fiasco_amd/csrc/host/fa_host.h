/*
 *  fa_host.h -- internal host-side declarations of libfiasco_amd.
 *
 *  Host C keeps what the reference keeps around the hot path: options object, PNM input,
 *  basis automaton, frame driver and the .fco stream writer.  Everything from
 *  subdivide() downwards (partition search, matching pursuit, inner-product tables,
 *  rate models) lives behind fa_core_encode_frames(), which is implemented
 *    - by the HIP device coder (csrc/hip/core_hip.hip) in the product library, and
 *    - by the CPU restatement (oracle/oracle_core.c) in the test-only oracle library.
 */
#ifndef FA_HOST_H
#define FA_HOST_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "libfiasco_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FA_MAXEDGES      5     /* reference codec/wfa.h:20 */
#define FA_MAXLABELS     2     /* reference codec/wfa.h:22 */
#define FA_STOCK_STATES  6000  /* reference codec/wfa.h:21 */
#define FA_STOCK_LEVEL   22    /* reference codec/wfa.h:23 */
#define FA_CAP_STATES    32000 /* word_t state ids */
#define FA_CAP_LEVEL     26
#define FA_NO_EDGE       (-1)
#define FA_RANGE         (-1)
#define FA_MAXCOSTS      1e20f /* reference codec/coder.c:53 */
#define FA_USE_DOMAIN    2     /* reference codec/wfa.h:41 USE_DOMAIN_MASK */
#define FA_AUXILIARY     1

enum { FA_I_FRAME = 0, FA_P_FRAME = 1, FA_B_FRAME = 2 };
enum { FA_GRAY = 0, FA_Y = 0, FA_CB = 1, FA_CR = 2 };

/* level geometry, reference lib/macros.h:48-52 */
#define fa_width_of_level(l)   (1u << ((l) >> 1))
#define fa_height_of_level(l)  (1u << (((l) + 1) >> 1))
#define fa_size_of_level(l)    (1u << (l))
#define fa_address_of_level(l) (fa_size_of_level(l) - 1u)
#define fa_size_of_tree(l)     (fa_address_of_level((l) + 1))

/* ---------------- error / messages (reference lib/error.c) ---------------- */
void fa_set_error(const char *fmt, ...);
/* getenv() for developer / test switches: null unless FIASCO_AMD_DEBUG is set (core_hip.cpp; the oracle's
 * core has its own) */
const char *fa_knob(const char *name);
void fa_warning(const char *fmt, ...);
void fa_message(const char *fmt, ...);
void fa_debug(const char *fmt, ...);
void fa_progress(const char *fmt, ...);

/* ---------------- reduced precision format (reference lib/rpf.h) ---------------- */
typedef struct fa_rpf {
    unsigned mantissa_bits;
    float    range;
    int      range_e;
} fa_rpf;
void fa_rpf_init(fa_rpf *r, unsigned mantissa, int range_e);
int  fa_rtob(float f, const fa_rpf *r);   /* used by the stream writer (weights) */

/* ---------------- options (reference codec/options.h:20-65) ---------------- */
typedef struct fa_options {
    char     id[9];
    char    *basis_name;
    unsigned lc_min_level, lc_max_level;
    unsigned p_min_level, p_max_level;
    unsigned images_level;
    unsigned max_states, chroma_max_states, max_elements;
    unsigned tiling_exponent;
    int      tiling_method;
    char    *id_domain_pool, *id_d_domain_pool, *id_rpf_model, *id_d_rpf_model;
    unsigned rpf_mantissa;      int rpf_range;
    unsigned dc_rpf_mantissa;   int dc_rpf_range;
    unsigned d_rpf_mantissa;    int d_rpf_range;
    unsigned d_dc_rpf_mantissa; int d_dc_rpf_range;
    float    chroma_decrease;
    int      prediction, delta_domains, normal_domains;
    unsigned search_range, fps;
    char    *pattern;
    int      half_pixel_prediction, cross_B_search, B_as_past_ref;
    int      check_for_underflow, check_for_overflow, second_domain_block, full_search;
    int      progress_meter;
    char    *title, *comment;
    unsigned smoothing;
} fa_options;
fa_options *fa_cast_options(const fiasco_c_options_t *o);

/* ---------------- image (reference lib/image.h, 12.4 fixed point int16) ---------- */
typedef struct fa_image {
    unsigned width, height;
    int      color;
    int16_t *pixels[3];
    int      borrowed;       /* planes belong to somebody else (upload staging buffer) */
    void    *dev;            /* decoded frames: the same planes [bands][height][width] on device dev_id, or NULL
                                (fa_core_decode_frames; released by fa_image_free through fa_core_release_dev) */
    int      dev_id;
} fa_image;
/* parse raw P5/P6 from memory; returns NULL + error message on failure */
fa_image *fa_image_from_pnm(const unsigned char *buf, size_t len, const char *name);
fa_image *fa_image_from_pnm_into(const unsigned char *buf, size_t len, const char *name,
                                 unsigned want_w, unsigned want_h, int want_color, int16_t *planes);
int       fa_pnm_header(const unsigned char *buf, size_t len, const char *name,
                        unsigned *w, unsigned *h, int *color, size_t *data_off);
void      fa_image_free(fa_image *im);
fa_image *fa_image_alloc(unsigned width, unsigned height, int color);   /* zeroed planes */

/* ---------------- reconstruction (reference codec/decoder.c, codec/motion.c) ------- */
unsigned char *fa_read_whole_file(const char *name, const char *env_var, size_t *len);

/* ---------------- automaton container handed to the writer ---------------- */
/* motion vector of a (state, label): reference codec/wfa.h:43-63 mv_t */
enum { FA_MV_NONE = 0, FA_MV_FORWARD = 1, FA_MV_BACKWARD = 2, FA_MV_INTERPOLATED = 3 };
typedef struct fa_mv { int16_t type, fx, fy, bx, by; } fa_mv;

typedef struct fa_wfa {
    unsigned cap;                 /* allocated states */
    unsigned states, basis_states, root_state;
    int      frame_type;
    float   *final_distribution;  /* [cap] */
    uint8_t *level_of_state;      /* [cap] */
    uint8_t *domain_type;         /* [cap] */
    uint8_t *delta_state;         /* [cap] */
    int16_t *tree;                /* [cap][2] */
    uint16_t *x, *y;              /* [cap][2] */
    int16_t *into;                /* [cap][2][6] */
    float   *weight;              /* [cap][2][6] */
    int16_t *y_state;             /* [cap][2] */
    uint8_t *y_column;            /* [cap][2] */
    uint8_t *prediction;          /* [cap][2] */
    fa_mv   *mv;                  /* [cap][2] mv_tree */
} fa_wfa;
#define FA_TREE(w, s, l)      ((w)->tree[(s) * 2 + (l)])
#define FA_INTO(w, s, l, e)   ((w)->into[((s) * 2 + (l)) * 6 + (e)])
#define FA_WEIGHT(w, s, l, e) ((w)->weight[((s) * 2 + (l)) * 6 + (e)])
fa_wfa *fa_wfa_alloc(unsigned cap);
void    fa_wfa_free(fa_wfa *w);
void    fa_wfa_remove_states(fa_wfa *w, unsigned from);
int     fa_wfa_append_edge(fa_wfa *w, unsigned from, unsigned into, float weight, unsigned label);
                                  /* 0: the label already has FA_MAXEDGES edges */
int     fa_load_basis(const char *name, fa_wfa *w);   /* 1 ok / 0 error */

/* TEST ORACLE ONLY (oracle/oracle_decoder.c; the product decodes on the device, fa_core_decode_frames below):
 * the frame an automaton describes, 4:4:4, cropped to the coded size (decode_image,
 * codec/decoder.c:411-536); NULL + message on failure */
fa_image *fa_decode_image(unsigned orig_width, unsigned orig_height, const fa_wfa *w, int color);
/* add the motion compensation of a P/B frame (restore_mc, codec/motion.c:37-229) */
int       fa_restore_mc(fa_image *image, const fa_image *past, const fa_image *future,
                        const fa_wfa *w, unsigned p_max_level);
void      fa_extract_mc_block(int16_t *mcblock, unsigned width, unsigned height,
                              const int16_t *reference, unsigned ref_width,
                              unsigned xo, unsigned yo, int mx, int my);
double    fa_plane_mse(const int16_t *a, const int16_t *b, size_t n);

/* the decoder as the core runs it (device library: csrc/hip/frame_decoder.inc; test oracle: the host decoder
 * above): decode_image for every job, restore_mc for the P/B frames among them (codec/coder.c:647-651) */
typedef struct fa_dec_job {
    const fa_wfa   *wfa;                  /* in  */
    unsigned        width, height;        /* in: the coded size (the decoder crops to it) */
    int             color, frame_type;    /* in  */
    const fa_image *past, *future;        /* in: references of a P/B frame */
    unsigned        p_max_level;          /* in  */
    int             skip;                 /* in: leave this job alone (keeps the index -> device dealing) */
    int             keep_dev;             /* in: leave the planes on the device too (out->dev): reference frames */
    unsigned        share_key;            /* in: as fa_job.share_key (0 = dealt by index) */
    fa_image       *out;                  /* out: the frame, or NULL + errmsg */
    char            errmsg[160];
} fa_dec_job;
int  fa_core_decode_frames(unsigned n, fa_dec_job *jobs);    /* number of frames decoded */
void fa_core_release_dev(void *dev, int dev_id);

/* ---------------- stream info (reference codec/wfa.h:65-110 wfa_info_t) ----------- */
typedef struct fa_info {
    char    *basis_name, *title, *comment;
    unsigned max_states, chroma_max_states;
    int      color;
    unsigned width, height, level;
    fa_rpf   rpf, dc_rpf, d_rpf, d_dc_rpf;
    unsigned frames, fps, p_min_level, p_max_level, search_range;
    int      half_pixel, cross_B_search, B_as_past_ref;
    unsigned smoothing;
} fa_info;

/* ---------------- model registries (reference codec/domain-pool.c:188-236, codec/coeff.c:97-131) -----
 * The reference names its domain pools and coefficient models by strings (c_options_t id_domain_pool,
 * id_d_domain_pool, id_rpf_model, id_d_rpf_model; codec/options.c:77-80) and looks them up in two tables;
 * an unknown name gives a warning and the FIRST entry of the table.  Same names, same order, same rule. */
typedef enum fa_pool_kind {
    FA_POOL_ADAPTIVE = 0,       /* "adaptive": quasi-arithmetic model per domain (qac), :259-498 */
    FA_POOL_CONSTANT,           /* "constant": domain list {0}, no model, :504-544 */
    FA_POOL_BASIS,              /* "basis": the adaptive pool over the basis states, :546-560 */
    FA_POOL_UNIFORM,            /* "uniform": every usable state, uniform price, :562-615 */
    FA_POOL_RLE,                /* "rle": the default, :621-879 */
    FA_POOL_RLE_NO_CHROMA       /* "rle-no-chroma": rle whose list is not cut down for the chroma bands, :886-899 */
} fa_pool_kind;
typedef enum fa_coeff_kind { FA_COEFF_ADAPTIVE = 0, FA_COEFF_UNIFORM } fa_coeff_kind;
/* name -> kind; *known = 0 and the table's first entry for a name the table does not hold (the caller warns) */
fa_pool_kind  fa_pool_kind_of(const char *name, int *known);
fa_coeff_kind fa_coeff_kind_of(const char *name, int *known);
const char   *fa_pool_name(fa_pool_kind k);
const char   *fa_coeff_name(fa_coeff_kind k);

/* ---------------- parameters of one frame for the core coder ---------------- */
typedef struct fa_cparams {
    float    price;                       /* 128*64/quality (codec/coder.c:164) */
    unsigned lc_min_level, lc_max_level;  /* after clamping (codec/coder.c:261-281) */
    unsigned images_level, products_level;
    unsigned max_elements;
    unsigned pool_max_states;             /* wi->max_states */
    unsigned chroma_max_states;
    float    chroma_decrease;
    fa_rpf   rpf, dc_rpf, d_rpf, d_dc_rpf;
    int      second_domain_block, check_for_underflow, check_for_overflow, full_search;
    unsigned level;                       /* bintree level of the whole image */
    unsigned limit_states, limit_level;   /* MAXSTATES / MAXLEVEL in force */
    /* prediction (codec/prediction.c) and motion compensation (codec/mwfa.c) */
    int      prediction;                  /* options.prediction: intra (ND) prediction */
    unsigned p_min_level, p_max_level;    /* wi->p_min_level / p_max_level */
    int      delta_domains, normal_domains;
    unsigned search_range;
    int      half_pixel, cross_B_search;
    /* the models (registries above): pool of the normal / of the delta approximation, their coefficient
     * models.  Through fiasco.h only rle / rle / adaptive / adaptive can be had (no setter exists,
     * codec/options.c:77-80); fiasco_amd_c_options_set_models() chooses the others. */
    fa_pool_kind  pool_kind, d_pool_kind;
    fa_coeff_kind coeff_kind, d_coeff_kind;
} fa_cparams;

/* result statistics of one coded band (root range), used for -V 2 style reporting */
typedef struct fa_stats {
    float costs, err, tree_bits, matrix_bits, weights_bits;
} fa_stats;

typedef struct fa_job {
    const fa_image   *image;      /* in  */
    int               frame_type; /* in: FA_I_FRAME / FA_P_FRAME / FA_B_FRAME */
    const fa_image   *past, *future;   /* in: reconstructed reference frames of a P/B frame */
    fa_cparams        cp;         /* in  (lc_min_level may be ratcheted: colour) */
    fa_wfa           *wfa;        /* in: basis states loaded; out: finished automaton */
    fa_stats          stats[3];   /* out */
    int               status;     /* out: 1 ok, 0 failed */
    char              errmsg[160];/* out */
    unsigned          lc_min_level_out; /* out: value of options.lc_min_level after the
                                           frame (colour carry, codec/coder.c:785-797) */
    int               ycol_carry; /* in: wfa->y_column holds what the previous frame of the
                                     stream left there.  The reference keeps ONE wfa_t for a
                                     whole stream and remove_states() (codec/wfalib.c:277-310)
                                     does not clear y_column, so the flags of a colour frame
                                     show through on states of the next frame that are not made
                                     by init_new_state (the three join states).  On return the
                                     array holds this frame's flags for EVERY state id. */
    unsigned          share_key;  /* in: 0 = none, else key + 1: jobs with the same key go to the same device share of
                                     the process whatever their index in the call (fa_share_of) -- the sequence engine
                                     passes the GOP number, so that the frames of a GOP and the decoded reference
                                     frames between them stay on ONE device while other GOPs end */
} fa_job;

/* Which of `shares' device shares (core_hip.cpp: one per device of the process) takes a job: a pure function of the
 * job's key when it has one, of its index in the call otherwise (round robin).  The search
 * (fa_core_stage) and the decoder (fa_core_decode_frames) both deal with it, so a keyed job is decoded where its
 * successor is searched. */
static inline unsigned fa_share_of(unsigned share_key, unsigned index, unsigned shares)
{
    return shares ? (share_key ? share_key - 1u : index) % shares : 0u;
}

/* THE SEAM.  Encode n independent frames.  Returns number of successful jobs.
 * Staged form: stage() makes the inputs resident where the core computes (HBM for the HIP
 * core), run() encodes every staged frame (may be called repeatedly), unstage() releases.
 * jobs[] must stay alive between stage() and unstage(). */
int   fa_core_encode_frames(unsigned n, fa_job *jobs);
void *fa_core_stage(unsigned n, fa_job *jobs);
int   fa_core_run(void *staged);            /* == submit + finish */
int   fa_core_submit(void *staged);         /* start encoding all staged frames, do not wait */
int   fa_core_finish2(void *staged, int resubmit);   /* finish, and start the next pass as
                                             * early as possible when resubmit != 0 */
int   fa_core_finish(void *staged);         /* wait, bring every frame to completion; the jobs'
                                             * automata are then in host memory and the core is
                                             * free for the next submit */
void  fa_core_unstage(void *staged);
/* Replacing the inputs of a staged batch while a pass runs (a stream of batches):
 *   upload_buffer: host staging memory for `bytes` of int16 planes (pinned for the HIP core),
 *                  valid until the next upload_buffer()/unstage(); NULL = not available
 *   upload_commit: jobs[i].image now describe the new frames, their planes lie inside the
 *                  upload buffer; the core copies them to where it computes WITHOUT waiting
 *                  (overlaps the running pass); the next submit uses them.  1 ok / 0 failed */
int16_t *fa_core_upload_buffer(void *staged, size_t bytes);
int   fa_core_upload_commit(void *staged);
const char *fa_core_name(void);

/* ---------------- bit writer (reference lib/bit-io.c, memory backed) -------------- */
typedef struct fa_bitw {
    unsigned char *buf;
    size_t   cap;
    size_t   bytes;      /* index of current byte + 1 (0 before the first bit) */
    unsigned bitpos;     /* reference semantics: 8 initially, counts down      */
    unsigned long long nbits;
} fa_bitw;
void fa_bw_init(fa_bitw *b);
void fa_bw_free(fa_bitw *b);
void fa_bw_append(fa_bitw *b, const fa_bitw *src);   /* b byte-aligned: what src holds, behind it */
void fa_bw_put_bit(fa_bitw *b, unsigned v);
void fa_bw_put_bits(fa_bitw *b, unsigned v, unsigned n);
void fa_bw_align(fa_bitw *b);
size_t fa_bw_finish(fa_bitw *b);   /* number of bytes the reference would write */
void fa_bw_rice(fa_bitw *b, unsigned value, unsigned k);
void fa_bw_bincode(fa_bitw *b, unsigned value, unsigned maxval);

/* ---------------- .fco writer (reference output/ directory) ---------------- */
void fa_write_header(const fa_info *wi, fa_bitw *out);
/* returns 1 ok, 0 on the reference's "total != ..." sanity error */
int  fa_write_frame(const fa_wfa *wfa, const fa_info *wi, int frame_type, unsigned number,
                    int prediction, int normal_domains, int delta_domains, fa_bitw *out);
/* stable (count desc, state asc) top-n hits, reference codec/wfalib.c:182-231 */
int16_t *fa_compute_hits(unsigned from, unsigned to, unsigned n, const fa_wfa *wfa);

/* ---------------- sequences: frames in coding order, GOPs side by side (fa_sequence.c) ----- */
typedef struct fa_seq fa_seq;
fa_seq *fa_seq_open(const fa_options *op, float quality, unsigned nframes, const unsigned char *const *bufs,
                    const size_t *lens, char const *const *names, unsigned rank, unsigned world);
void    fa_seq_free(fa_seq *s);
int     fa_seq_search(fa_seq *s, const unsigned *carry_in, const uint8_t *todo);
int     fa_seq_probe(fa_seq *s, unsigned *level);
int     fa_seq_encode_all(fa_seq *s, fa_bitw *out,
                          void (*report)(const fa_wfa *, const fa_stats *, const fa_info *));
unsigned fa_seq_gops(const fa_seq *s);
unsigned fa_seq_frames(const fa_seq *s);
unsigned fa_seq_ycol_size(const fa_seq *s);
unsigned fa_seq_initial_level(const fa_seq *s);
unsigned fa_seq_gop_of(const fa_seq *s, unsigned k);
int      fa_seq_is_mine(const fa_seq *s, unsigned gop);
void     fa_seq_gop_result(const fa_seq *s, unsigned gop, unsigned *carry_out, int *failed);
const char *fa_seq_gop_error(const fa_seq *s, unsigned gop);
const fa_stats *fa_seq_stats(const fa_seq *s, unsigned k);
const fa_info *fa_seq_info(const fa_seq *s);
const uint8_t *fa_seq_ycol_raw(const fa_seq *s, unsigned k);
void     fa_seq_ycol_resolve(uint8_t *chain, const uint8_t *raw, unsigned n);
int      fa_seq_write(fa_seq *s, unsigned k, const uint8_t *ycol, fa_bitw *out);

/* ---------------- frame driver pieces shared by fiasco_coder and the batch API ----- */
unsigned fa_image_level(unsigned width, unsigned height);      /* codec/coder.c:247-255 */
int fa_setup_params(const fa_options *op, float quality, unsigned width, unsigned height,
                    int color, unsigned frames, fa_info *wi, fa_cparams *cp);
void fa_info_free(fa_info *wi);
void fa_limits(unsigned *max_states, unsigned *max_level);

#ifdef __cplusplus
}
#endif
#endif
