/*
 *  libfiasco_amd.h -- C ABI of the MI355X-native FIASCO encoder library.
 *
 *  This is the drop-in boundary: a C-ABI shared library whose encoder
 *  entry points have the same names, argument meaning, return convention (1 = ok,
 *  0 = failure + fiasco_get_error_message()) and the same struct layouts as the
 *  reference's public header, so that the reference's `cfiasco` objects link against it
 *  unchanged.  Each declaration cites the reference interface it replaces.
 *
 *  Only the ENCODER side is provided (hot path = encode-side matching pursuit).
 *  Decoder / image / renderer entry points of the reference (fiasco.h:223-296) are out of
 *  scope and not exported.
 */
#ifndef LIBFIASCO_AMD_H
#define LIBFIASCO_AMD_H 1

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: values are ABI (reference fiasco.h:56-80) ---- */
typedef enum { FIASCO_NO_VERBOSITY, FIASCO_SOME_VERBOSITY,
               FIASCO_ULTIMATE_VERBOSITY } fiasco_verbosity_e;
typedef enum { FIASCO_TILING_SPIRAL_ASC, FIASCO_TILING_SPIRAL_DSC,
               FIASCO_TILING_VARIANCE_ASC, FIASCO_TILING_VARIANCE_DSC } fiasco_tiling_e;
typedef enum { FIASCO_RPF_RANGE_0_75, FIASCO_RPF_RANGE_1_00,
               FIASCO_RPF_RANGE_1_50, FIASCO_RPF_RANGE_2_00 } fiasco_rpf_range_e;
typedef enum { FIASCO_PROGRESS_NONE, FIASCO_PROGRESS_BAR,
               FIASCO_PROGRESS_PERCENT } fiasco_progress_e;

/* ---- coder options object: vtable order is ABI (reference fiasco.h:132-174) ---- */
typedef struct fiasco_c_options {
    void (*delete_)(struct fiasco_c_options *o);   /* reference member name: delete */
    int (*set_tiling)(struct fiasco_c_options *o, fiasco_tiling_e method, unsigned exponent);
    int (*set_frame_pattern)(struct fiasco_c_options *o, const char *pattern);
    int (*set_basisfile)(struct fiasco_c_options *o, const char *filename);
    int (*set_chroma_quality)(struct fiasco_c_options *o, float quality_factor,
                              unsigned dictionary_size);
    int (*set_optimizations)(struct fiasco_c_options *o, unsigned min_block_level,
                             unsigned max_block_level, unsigned max_elements,
                             unsigned dictionary_size, unsigned optimization_level);
    int (*set_prediction)(struct fiasco_c_options *o, int intra_prediction,
                          unsigned min_block_level, unsigned max_block_level);
    int (*set_video_param)(struct fiasco_c_options *o, unsigned frames_per_second,
                           int half_pixel_prediction, int cross_B_search, int B_as_past_ref);
    int (*set_quantization)(struct fiasco_c_options *o, unsigned mantissa,
                            fiasco_rpf_range_e range, unsigned dc_mantissa,
                            fiasco_rpf_range_e dc_range);
    int (*set_progress_meter)(struct fiasco_c_options *o, fiasco_progress_e type);
    int (*set_smoothing)(struct fiasco_c_options *o, int smoothing);
    int (*set_comment)(struct fiasco_c_options *o, const char *comment);
    int (*set_title)(struct fiasco_c_options *o, const char *title);
    void *private_;                                /* reference member name: private */
} fiasco_c_options_t;

/* ---- misc (reference fiasco.h:210-216, lib/error.c:178-186,300-310) ---- */
const char *fiasco_get_error_message(void);
void fiasco_set_verbosity(fiasco_verbosity_e level);
fiasco_verbosity_e fiasco_get_verbosity(void);

/* ---- the encoder entry point (reference fiasco.h:303-306, codec/coder.c:85-182) ----
 * inputname: NULL-terminated list of raw PGM/PPM names or "prefix[a-b{+,-}s]suffix"
 * templates; outputname NULL or "-" = stdout; quality > 0; options may be NULL.       */
int fiasco_coder(char const *const *inputname, const char *outputname, float quality,
                 const fiasco_c_options_t *options);

/* ---- options constructor / setters (reference fiasco.h:313-398, codec/options.c) ---- */
fiasco_c_options_t *fiasco_c_options_new(void);
void fiasco_c_options_delete(fiasco_c_options_t *options);
int fiasco_c_options_set_smoothing(fiasco_c_options_t *options, int smoothing);
int fiasco_c_options_set_frame_pattern(fiasco_c_options_t *options, const char *pattern);
int fiasco_c_options_set_tiling(fiasco_c_options_t *options, fiasco_tiling_e method,
                                unsigned exponent);
int fiasco_c_options_set_basisfile(fiasco_c_options_t *options, const char *filename);
int fiasco_c_options_set_chroma_quality(fiasco_c_options_t *options, float quality_factor,
                                        unsigned dictionary_size);
int fiasco_c_options_set_optimizations(fiasco_c_options_t *options, unsigned min_block_level,
                                       unsigned max_block_level, unsigned max_elements,
                                       unsigned dictionary_size, unsigned optimization_level);
int fiasco_c_options_set_prediction(fiasco_c_options_t *options, int intra_prediction,
                                    unsigned min_block_level, unsigned max_block_level);
int fiasco_c_options_set_video_param(fiasco_c_options_t *options, unsigned frames_per_second,
                                     int half_pixel_prediction, int cross_B_search,
                                     int B_as_past_ref);
int fiasco_c_options_set_quantization(fiasco_c_options_t *options, unsigned mantissa,
                                      fiasco_rpf_range_e range, unsigned dc_mantissa,
                                      fiasco_rpf_range_e dc_range);
int fiasco_c_options_set_progress_meter(fiasco_c_options_t *options, fiasco_progress_e type);
int fiasco_c_options_set_comment(fiasco_c_options_t *options, const char *comment);
int fiasco_c_options_set_title(fiasco_c_options_t *options, const char *title);

/* ---- two non-public symbols the reference CLI objects import (nm cwfa.o params.o;
 *      reference lib/misc.c:51-71, lib/bit-io.c:48-146) ---- */
void *fiasco_calloc(size_t n, size_t size);
typedef enum { READ_ACCESS, WRITE_ACCESS } openmode_e;
FILE *open_file(const char *filename, const char *env_var, openmode_e mode);

/* =====================================================================================
 *  Extensions of this library (not in the reference).  Plain pointers and sizes only.
 * ===================================================================================== */

/* Limits extension: the reference hard-codes MAXSTATES 6000 /
 * MAXLEVEL 22 (codec/wfa.h:21,23), which rules out 4K frames and 1080p colour at CLI
 * defaults.  Default here = the stock limits (bit parity with the stock reference).
 * max_states in [16, 32000], max_level in [22, 26].  Returns 1 on success.            */
int fiasco_amd_set_limits(unsigned max_states, unsigned max_level);
void fiasco_amd_get_limits(unsigned *max_states, unsigned *max_level);

/* Batch encoder: encode `n` independent still images (raw PNM bytes in host memory) into
 * `n` .fco byte strings, all frames in flight on the GPU at once (one persistent
 * workgroup per frame).  Semantically identical to n separate fiasco_coder() calls.
 * out[i] is malloc()ed by the library (caller frees with fiasco_amd_free), out_len[i] its
 * length.  Returns the number of frames encoded successfully (== n on success).        */
int fiasco_amd_encode_batch(unsigned n, const unsigned char *const *pnm,
                            const size_t *pnm_len, float quality,
                            const fiasco_c_options_t *options,
                            unsigned char **out, size_t *out_len);
void fiasco_amd_free(void *p);

/* Staged form of the batch encoder, for callers that keep frames resident on the GPU:
 *   stage : parse the PNM buffers, upload the pixel planes into HBM        -> handle
 *   encode: run the hot path over every staged frame and write the .fco byte strings
 *           (same out/out_len contract as fiasco_amd_encode_batch); repeatable
 *   free  : release the handle (HBM slabs go back to the library's pool)
 * fiasco_amd_encode_batch(...) == stage + encode + free.                               */
typedef struct fiasco_amd_batch fiasco_amd_batch_t;
fiasco_amd_batch_t *fiasco_amd_batch_stage(unsigned n, const unsigned char *const *pnm,
                                           const size_t *pnm_len, float quality,
                                           const fiasco_c_options_t *options);
int  fiasco_amd_batch_encode(fiasco_amd_batch_t *batch, unsigned char **out, size_t *out_len);
/* Pipelined form of encode for repeated passes (a stream of batches): submit starts the
 * device search and returns; collect waits for it, optionally starts the next pass
 * (resubmit != 0) and then writes the streams of the finished pass on the host, so the
 * host-side .fco writer (output/write.c in the reference) overlaps the next device pass.
 * submit + collect(.., 0) == encode. */
int  fiasco_amd_batch_submit(fiasco_amd_batch_t *batch);
int  fiasco_amd_batch_collect(fiasco_amd_batch_t *batch, unsigned char **out, size_t *out_len,
                              int resubmit);
/* A stream of batches: replace the frame of EVERY slot of a staged batch (same number of
 * frames, same sizes, colour model and options) with new raw PNM buffers in host memory.
 * The buffers are parsed by a few host threads into pinned staging memory and copied to HBM
 * without waiting, so a call between submit and collect overlaps the pass that is running;
 * the next submit (or collect with resubmit) encodes the new frames.  Returns 1 / 0 + message;
 * on failure the batch keeps its previous frames. */
int  fiasco_amd_batch_upload(fiasco_amd_batch_t *batch, const unsigned char *const *pnm,
                             const size_t *pnm_len);
/* Root-range statistics of frame i, band 0..2 (Y, Cb, Cr), of the last finished pass: the
 * figures the reference prints at verbosity 2 (codec/coder.c:918-923): costs, squared error
 * `err` (coder-side PSNR = 10 log10(255^2 / (err / (width*height)))), and width/height.
 * Returns 1, or 0 when i/band are out of range or the frame failed.                     */
int  fiasco_amd_batch_stats(const fiasco_amd_batch_t *batch, unsigned i, unsigned band,
                            float *costs, float *err, unsigned *width, unsigned *height);

/* Decoded PSNR of frame i after a successful pass (reference tools: dfiasco -s 0 followed
 * by pnmpsnr, codec/decoder.c:411-536 and bin/pnmpsnr.c:36-163): the automaton is decoded without
 * smoothing and compared with the input as bytes.  psnr_db / mse: one entry per band (gray: [0] only; may
 * be NULL).  Intra frames only.  1 ok / 0 + error message. */
int  fiasco_amd_batch_decode_psnr(const fiasco_amd_batch_t *batch, unsigned i, double psnr_db[3], double mse[3]);
/* ... of every frame of the batch, decoded by ONE call of the device decoder (csrc/hip/
 * frame_decoder.inc replaces decode_image, codec/decoder.c:411-536): psnr_db / mse are [n][3] (either may be NULL),
 * zeros for frames without a finished intra automaton.  Returns the number of frames decoded. */
int  fiasco_amd_batch_decode_psnr_all(const fiasco_amd_batch_t *batch, double *psnr_db, double *mse);
/* ... and the decoded frame itself: one band as width x height bytes, clip((pixel >> 4) + 128) -- for a gray frame
 * the payload of the PGM `dfiasco -s 0 -o` writes (lib/image.c:449-483). */
int  fiasco_amd_batch_decode_plane(const fiasco_amd_batch_t *batch, unsigned i, unsigned band, unsigned char *out);

/* The model names of the reference's c_options_t (codec/options.h:36-39; registries codec/domain-pool.c:188-236
 * "adaptive", "constant", "basis", "uniform", "rle", "rle-no-chroma" and codec/coeff.c:97-131 "adaptive",
 * "uniform").  The reference has the fields and the registries but no setter; through fiasco.h a coder
 * always runs rle / rle / adaptive / adaptive.  NULL keeps a name; an unknown name is a warning and the
 * first entry of its table, as in the reference.  The HIP device coder runs every entry of both registries (the
 * `FC_GM' build of the kernel, csrc/hip/frame_coder.h; streams pinned against builds of the reference with other
 * registry defaults, tests/golden/MANIFEST.json "model_cases"). */
int  fiasco_amd_c_options_set_models(fiasco_c_options_t *options, const char *domain_pool, const char *d_domain_pool,
                                     const char *rpf_model, const char *d_rpf_model);
void fiasco_amd_batch_free(fiasco_amd_batch_t *batch);

/* ---- sequences across processes (one process per GPU) ---------------------------------------
 * fiasco_coder() codes a video by searching its groups of pictures (an I frame and what follows
 * up to the next I frame; reference codec/coder.c:581-592) side by side on one GPU.  With these
 * entries process `rank` of `world` searches and writes every world-th GOP of the same sequence;
 * what the processes exchange is left to the caller (fiasco_amd/sharding.py: torch.distributed):
 *   - colour streams hand the minimum block level from frame to frame (codec/coder.c:785-797):
 *     search with a speculated level per GOP (carry_in), gather what every GOP left
 *     (gop_result), search again from the first GOP whose start value was wrong;
 *   - the y_column flags that show through from frame to frame (codec/wfalib.c:277-310): gather
 *     the raw flags of every frame (seq_ycol: 2 = "not written by this frame"), resolve them in
 *     coding order, hand the resolved flags of a frame to seq_write;
 *   - the byte strings of the frames in coding order are the stream.
 * Frames are given as raw PNM buffers in DISPLAY order; they and `options` must stay alive until
 * seq_free.  All functions return 1 / non-NULL on success, 0 / NULL + message otherwise. */
typedef struct fiasco_amd_seq fiasco_amd_seq_t;
fiasco_amd_seq_t *fiasco_amd_seq_open(unsigned n, const unsigned char *const *pnm, const size_t *pnm_len,
                                      float quality, const fiasco_c_options_t *options,
                                      unsigned rank, unsigned world);
void     fiasco_amd_seq_free(fiasco_amd_seq_t *seq);
unsigned fiasco_amd_seq_gops(const fiasco_amd_seq_t *seq);
unsigned fiasco_amd_seq_frames(const fiasco_amd_seq_t *seq);          /* in coding order */
unsigned fiasco_amd_seq_gop_of(const fiasco_amd_seq_t *seq, unsigned frame);
unsigned fiasco_amd_seq_ycol_size(const fiasco_amd_seq_t *seq);       /* 0 for gray streams */
unsigned fiasco_amd_seq_initial_level(const fiasco_amd_seq_t *seq);
/* the level worth speculating for every GOP but the first: what the first frame alone leaves */
int      fiasco_amd_seq_probe(fiasco_amd_seq_t *seq, unsigned *level);
int      fiasco_amd_seq_search(fiasco_amd_seq_t *seq, const unsigned *carry_in, const unsigned char *todo);
int      fiasco_amd_seq_gop_result(const fiasco_amd_seq_t *seq, unsigned gop, unsigned *carry_out, int *failed);
const unsigned char *fiasco_amd_seq_ycol(const fiasco_amd_seq_t *seq, unsigned frame);
int      fiasco_amd_seq_write(fiasco_amd_seq_t *seq, unsigned frame, const unsigned char *ycol,
                              unsigned char **out, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif /* LIBFIASCO_AMD_H */
