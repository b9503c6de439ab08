/*
 *  fa_options.c -- the fiasco_c_options_* object (reference codec/options.c:36-708).
 *
 *  Same defaults (options.c:66-107), same validation rules and error texts, same
 *  "COFIASCO" type tag check (options.c:682-708).
 */
#include <stdlib.h>
#include <string.h>
#include "fa_host.h"

static char *dupstr(const char *s) { return strdup(s ? s : ""); }

fa_options *fa_cast_options(const fiasco_c_options_t *o)
{
    fa_options *op;
    if (!o) { fa_set_error("Parameter `%s' not defined (NULL).", "options"); return NULL; }
    op = (fa_options *) o->private_;
    if (!op) { fa_set_error("Parameter `%s' not defined (NULL).", "options"); return NULL; }
    if (strcmp(op->id, "COFIASCO") != 0) {
        fa_set_error("Parameter `options' doesn't match required type.");
        return NULL;
    }
    return op;
}

fiasco_c_options_t *fiasco_c_options_new(void)
{
    fiasco_c_options_t *pub = (fiasco_c_options_t *) calloc(1, sizeof *pub);
    fa_options *op = (fa_options *) calloc(1, sizeof *op);
    if (!pub || !op) { free(pub); free(op); fa_set_error("Out of memory!"); return NULL; }

    pub->private_           = op;
    pub->delete_            = fiasco_c_options_delete;
    pub->set_tiling         = fiasco_c_options_set_tiling;
    pub->set_frame_pattern  = fiasco_c_options_set_frame_pattern;
    pub->set_basisfile      = fiasco_c_options_set_basisfile;
    pub->set_chroma_quality = fiasco_c_options_set_chroma_quality;
    pub->set_optimizations  = fiasco_c_options_set_optimizations;
    pub->set_prediction     = fiasco_c_options_set_prediction;
    pub->set_video_param    = fiasco_c_options_set_video_param;
    pub->set_quantization   = fiasco_c_options_set_quantization;
    pub->set_progress_meter = fiasco_c_options_set_progress_meter;
    pub->set_smoothing      = fiasco_c_options_set_smoothing;
    pub->set_title          = fiasco_c_options_set_title;
    pub->set_comment        = fiasco_c_options_set_comment;

    strcpy(op->id, "COFIASCO");
    /* library defaults: note they differ from the CLI defaults (levels 4..12, 5 edges) */
    op->basis_name        = dupstr("small.fco");
    op->lc_min_level      = 4;
    op->lc_max_level      = 12;
    op->p_min_level       = 8;
    op->p_max_level       = 10;
    op->images_level      = 5;
    op->max_states        = FA_STOCK_STATES;
    op->chroma_max_states = 40;
    op->max_elements      = FA_MAXEDGES;
    op->tiling_exponent   = 4;
    op->tiling_method     = FIASCO_TILING_VARIANCE_DSC;
    op->id_domain_pool    = dupstr("rle");
    op->id_d_domain_pool  = dupstr("rle");
    op->id_rpf_model      = dupstr("adaptive");
    op->id_d_rpf_model    = dupstr("adaptive");
    op->rpf_mantissa      = 3;  op->rpf_range      = FIASCO_RPF_RANGE_1_50;
    op->dc_rpf_mantissa   = 5;  op->dc_rpf_range   = FIASCO_RPF_RANGE_1_00;
    op->d_rpf_mantissa    = 3;  op->d_rpf_range    = FIASCO_RPF_RANGE_1_50;
    op->d_dc_rpf_mantissa = 5;  op->d_dc_rpf_range = FIASCO_RPF_RANGE_1_00;
    op->chroma_decrease   = 2.0f;
    op->prediction        = 0;
    op->delta_domains     = 1;
    op->normal_domains    = 1;
    op->search_range      = 16;
    op->fps               = 25;
    op->pattern           = dupstr("IPPPPPPPPP");
    op->half_pixel_prediction = 0;
    op->cross_B_search    = 1;
    op->B_as_past_ref     = 1;
    op->progress_meter    = FIASCO_PROGRESS_NONE;
    op->smoothing         = 70;
    op->comment           = dupstr("");
    op->title             = dupstr("");
    return pub;
}

void fiasco_c_options_delete(fiasco_c_options_t *options)
{
    fa_options *op;
    if (!options) return;
    op = fa_cast_options(options);
    if (!op) return;
    free(op->basis_name);
    free(op->id_domain_pool);  free(op->id_d_domain_pool);
    free(op->id_rpf_model);    free(op->id_d_rpf_model);
    free(op->pattern); free(op->comment); free(op->title);
    free(op);
    free(options);
}

int fiasco_c_options_set_tiling(fiasco_c_options_t *options, fiasco_tiling_e method,
                                unsigned exponent)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if ((int) method < FIASCO_TILING_SPIRAL_ASC || (int) method > FIASCO_TILING_VARIANCE_DSC) {
        fa_set_error("Invalid tiling method `%d' specified (valid methods are 0, 1, 2, or 3).",
                     (int) method);
        return 0;
    }
    /* Stored and written nowhere: the reference's alloc_tiling() drops both values
     * (codec/tiling.c:68-91), so tiling is inert; reproduced as a no-op. */
    op->tiling_method   = method;
    op->tiling_exponent = exponent;
    return 1;
}

int fiasco_c_options_set_frame_pattern(fiasco_c_options_t *options, const char *pattern)
{
    fa_options *op = fa_cast_options(options);
    const char *s;
    if (!op) return 0;
    if (!pattern) { fa_set_error("Parameter `%s' not defined (NULL).", "pattern"); return 0; }
    if (!*pattern) { fa_set_error("Frame type pattern doesn't contain any character."); return 0; }
    for (s = pattern; *s; s++)
        if (!strchr("iIbBpP", *s)) {
            fa_set_error("Frame type pattern contains invalid character `%c' (choose I, B or P).", *s);
            return 0;
        }
    free(op->pattern);
    op->pattern = dupstr(pattern);
    return 1;
}

int fiasco_c_options_set_basisfile(fiasco_c_options_t *options, const char *filename)
{
    fa_options *op = fa_cast_options(options);
    FILE *f;
    if (!op) return 0;
    if (!filename) { fa_set_error("Parameter `%s' not defined (NULL).", "filename"); return 0; }
    /* the reference insists the file can be opened even for the compiled-in basis */
    f = open_file(filename, "FIASCO_DATA", READ_ACCESS);
    if (!f) {
        fa_set_error("Can't read basis file `%s'.\n%s.", filename, "No such file or directory");
        return 0;
    }
    fclose(f);
    free(op->basis_name);
    op->basis_name = dupstr(filename);
    return 1;
}

int fiasco_c_options_set_chroma_quality(fiasco_c_options_t *options, float quality_factor,
                                        unsigned dictionary_size)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if (!dictionary_size) {
        fa_set_error("Size of chroma compression dictionary has to be a positive number.");
        return 0;
    }
    if (quality_factor <= 0) {
        fa_set_error("Quality of chroma channel compression has to be positive value.");
        return 0;
    }
    op->chroma_decrease   = quality_factor;
    op->chroma_max_states = dictionary_size;
    return 1;
}

int fiasco_c_options_set_optimizations(fiasco_c_options_t *options, unsigned min_block_level,
                                       unsigned max_block_level, unsigned max_elements,
                                       unsigned dictionary_size, unsigned optimization_level)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if (!dictionary_size) { fa_set_error("Size of dictionary has to be a positive number."); return 0; }
    if (!max_elements) {
        fa_set_error("At least one dictionary element has to be used in an approximation.");
        return 0;
    }
    if (max_block_level < 4) { fa_set_error("Maximum image block size has to be at least level 4."); return 0; }
    if (min_block_level < 4) { fa_set_error("Minimum image block size has to be at least level 4."); return 0; }
    if (max_block_level < min_block_level) {
        fa_set_error("Maximum block size has to be larger or equal minimum block size.");
        return 0;
    }
    op->lc_min_level = min_block_level;
    op->lc_max_level = max_block_level;
    op->max_states   = dictionary_size;
    op->max_elements = max_elements;
    op->second_domain_block = optimization_level > 0;
    op->check_for_overflow  = optimization_level > 1;
    op->check_for_underflow = optimization_level > 1;
    op->full_search         = optimization_level > 1;
    return 1;
}

int fiasco_c_options_set_prediction(fiasco_c_options_t *options, int intra_prediction,
                                    unsigned min_block_level, unsigned max_block_level)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if (max_block_level < 6) { fa_set_error("Maximum prediction block size has to be at least level 6"); return 0; }
    if (min_block_level < 6) { fa_set_error("Minimum prediction block size has to be at least level 6"); return 0; }
    if (max_block_level < min_block_level) {
        fa_set_error("Maximum prediction block size has to be larger or equal minimum block size.");
        return 0;
    }
    op->p_min_level = min_block_level;
    op->p_max_level = max_block_level;
    op->prediction  = intra_prediction;
    return 1;
}

int fiasco_c_options_set_video_param(fiasco_c_options_t *options, unsigned frames_per_second,
                                     int half_pixel_prediction, int cross_B_search,
                                     int B_as_past_ref)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    op->fps = frames_per_second;
    op->half_pixel_prediction = half_pixel_prediction;
    op->cross_B_search = cross_B_search;
    op->B_as_past_ref = B_as_past_ref;
    return 1;
}

static int valid_range(int r) { return r >= FIASCO_RPF_RANGE_0_75 && r <= FIASCO_RPF_RANGE_2_00; }

int fiasco_c_options_set_quantization(fiasco_c_options_t *options, unsigned mantissa,
                                      fiasco_rpf_range_e range, unsigned dc_mantissa,
                                      fiasco_rpf_range_e dc_range)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if (mantissa < 2 || mantissa > 8 || dc_mantissa < 2 || dc_mantissa > 8) {
        fa_set_error("Number of RPF mantissa bits `%d', `%d' have to be in the interval [2,8].",
                     (int) mantissa, (int) dc_mantissa);
        return 0;
    }
    if (!valid_range((int) range) || !valid_range((int) dc_range)) {
        fa_set_error("Invalid RPF ranges `%d', `%d' specified.", (int) range, (int) dc_range);
        return 0;
    }
    op->rpf_range = range;       op->dc_rpf_range = dc_range;
    op->rpf_mantissa = mantissa; op->dc_rpf_mantissa = dc_mantissa;
    return 1;
}

int fiasco_c_options_set_progress_meter(fiasco_c_options_t *options, fiasco_progress_e type)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if ((int) type < FIASCO_PROGRESS_NONE || (int) type > FIASCO_PROGRESS_PERCENT) {
        fa_set_error("Invalid progress meter `%d' specified (valid values are 0, 1, or 2).", (int) type);
        return 0;
    }
    op->progress_meter = type;
    return 1;
}

int fiasco_c_options_set_smoothing(fiasco_c_options_t *options, int smoothing)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if (smoothing < -1 || smoothing > 100) {
        fa_set_error("Smoothing percentage must be in the range [-1, 100].");
        return 0;
    }
    op->smoothing = (unsigned) smoothing;
    return 1;
}

int fiasco_c_options_set_comment(fiasco_c_options_t *options, const char *comment)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if (!comment) { fa_set_error("Parameter `%s' not defined (NULL).", "title"); return 0; }
    free(op->comment);
    op->comment = dupstr(comment);
    return 1;
}

int fiasco_c_options_set_title(fiasco_c_options_t *options, const char *title)
{
    fa_options *op = fa_cast_options(options);
    if (!op) return 0;
    if (!title) { fa_set_error("Parameter `%s' not defined (NULL).", "title"); return 0; }
    free(op->title);
    op->title = dupstr(title);
    return 1;
}


/* ---------------- model registries (codec/domain-pool.c:188-236, codec/coeff.c:97-131) ---------------- */

static const char *const pool_names[] = { "adaptive", "constant", "basis", "uniform", "rle", "rle-no-chroma", NULL };
static const char *const coeff_names[] = { "adaptive", "uniform", NULL };

fa_pool_kind fa_pool_kind_of(const char *name, int *known)
{
    unsigned n;
    if (known) *known = 1;
    for (n = 0; pool_names[n]; n++)
        if (name && strcasecmp(pool_names[n], name) == 0) return (fa_pool_kind) n;
    if (known) *known = 0;
    return FA_POOL_ADAPTIVE;            /* "Using default value": the first entry */
}

fa_coeff_kind fa_coeff_kind_of(const char *name, int *known)
{
    unsigned n;
    if (known) *known = 1;
    for (n = 0; coeff_names[n]; n++)
        if (name && strcasecmp(coeff_names[n], name) == 0) return (fa_coeff_kind) n;
    if (known) *known = 0;
    return FA_COEFF_ADAPTIVE;
}

const char *fa_pool_name(fa_pool_kind k) { return pool_names[k]; }
const char *fa_coeff_name(fa_coeff_kind k) { return coeff_names[k]; }

/* The four model names of c_options_t (codec/options.h:36-39): the reference has the fields and the
 * registries but no fiasco_c_options_set_*() for them.  NULL keeps a name. */
int fiasco_amd_c_options_set_models(fiasco_c_options_t *options, const char *domain_pool, const char *d_domain_pool,
                                    const char *rpf_model, const char *d_rpf_model)
{
    fa_options *op = (fa_options *) fa_cast_options(options);
    if (!op) return 0;
    if (domain_pool)   { free(op->id_domain_pool);   op->id_domain_pool = dupstr(domain_pool); }
    if (d_domain_pool) { free(op->id_d_domain_pool); op->id_d_domain_pool = dupstr(d_domain_pool); }
    if (rpf_model)     { free(op->id_rpf_model);     op->id_rpf_model = dupstr(rpf_model); }
    if (d_rpf_model)   { free(op->id_d_rpf_model);   op->id_d_rpf_model = dupstr(d_rpf_model); }
    return 1;
}
