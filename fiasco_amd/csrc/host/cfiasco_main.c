/*
 *  cfiasco_main.c -- thin command line front end over the C API (own implementation of
 *  the subset of reference bin/cwfa.c options that matter for the encode path).
 *  CLI defaults follow bin/cwfa.c:36-98 and the option mapping of bin/cwfa.c:252-393
 *  (e.g. --optimize 0 => block levels [6,10], 3 elements; >=1 => [4,12], 5 elements).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <getopt.h>
#include "libfiasco_amd.h"

static fiasco_rpf_range_e to_range(float r)
{
    if (r < 1) return FIASCO_RPF_RANGE_0_75;
    if (r < 1.5) return FIASCO_RPF_RANGE_1_00;
    if (r < 2.0) return FIASCO_RPF_RANGE_1_50;
    return FIASCO_RPF_RANGE_2_00;
}

int main(int argc, char **argv)
{
    const char *outname = "-", *pattern = "ippppppppp", *basis = NULL;
    const char *title = "", *comment = "";
    float quality = 20.0f, chroma_q = 2, rpf_r = 1.5f, dc_r = 1.0f;
    int optimize = 0, dict = 10000, cdict = 40, pmin = 6, pmax = 10, pred = 0, verbose = 1;
    int smooth = 70, progress = 2, rpf_m = 3, dc_m = 5, tile_e = 4;
    int lim_states = 0, lim_level = 0;
    const char *m_pool = NULL, *m_dpool = NULL, *m_rpf = NULL, *m_drpf = NULL;
    static struct option lo[] = {
        {"output-name", 1, 0, 'o'}, {"quality", 1, 0, 'q'}, {"optimize", 1, 0, 'z'},
        {"verbose", 1, 0, 'V'}, {"title", 1, 0, 't'}, {"comment", 1, 0, 'c'},
        {"pattern", 1, 0, 1}, {"dictionary-size", 1, 0, 2}, {"chroma-dictionary", 1, 0, 3},
        {"chroma-qfactor", 1, 0, 4}, {"progress-meter", 1, 0, 5}, {"smooth", 1, 0, 6},
        {"rpf-range", 1, 0, 7}, {"rpf-mantissa", 1, 0, 8}, {"dc-rpf-range", 1, 0, 9},
        {"dc-rpf-mantissa", 1, 0, 10}, {"min-level", 1, 0, 11}, {"max-level", 1, 0, 12},
        {"prediction", 0, 0, 13}, {"basis-name", 1, 0, 14}, {"tiling-exponent", 1, 0, 15},
        {"limit-states", 1, 0, 16}, {"limit-level", 1, 0, 17}, {"tiling-method", 1, 0, 18},
        /* parsed and dropped like bin/cwfa.c does (it never calls set_video_param) */
        {"half-pixel", 0, 0, 18}, {"cross-B-search", 0, 0, 18}, {"B-as-past-ref", 0, 0, 18},
        {"fps", 1, 0, 18},
        /* the model names of c_options_t the reference CLI cannot set (fiasco_amd_c_options_set_models) */
        {"domain-pool", 1, 0, 19}, {"d-domain-pool", 1, 0, 20}, {"rpf-model", 1, 0, 21}, {"d-rpf-model", 1, 0, 22},
        {0, 0, 0, 0}
    };
    int ch;
    fiasco_c_options_t *o;
    const char **inputs;
    int i, n;

    while ((ch = getopt_long(argc, argv, "o:q:z:V:t:c:", lo, NULL)) != -1)
        switch (ch) {
        case 'o': outname = optarg; break;
        case 'q': quality = (float) atof(optarg); break;
        case 'z': optimize = atoi(optarg); break;
        case 'V': verbose = atoi(optarg); break;
        case 't': title = optarg; break;
        case 'c': comment = optarg; break;
        case 1: pattern = optarg; break;
        case 2: dict = atoi(optarg); break;
        case 3: cdict = atoi(optarg); break;
        case 4: chroma_q = (float) atof(optarg); break;
        case 5: progress = atoi(optarg); break;
        case 6: smooth = atoi(optarg); break;
        case 7: rpf_r = (float) atof(optarg); break;
        case 8: rpf_m = atoi(optarg); break;
        case 9: dc_r = (float) atof(optarg); break;
        case 10: dc_m = atoi(optarg); break;
        case 11: pmin = atoi(optarg); break;
        case 12: pmax = atoi(optarg); break;
        case 13: pred = 1; break;
        case 14: basis = optarg; break;
        case 15: tile_e = atoi(optarg); break;
        case 16: lim_states = atoi(optarg); break;
        case 17: lim_level = atoi(optarg); break;
        case 18: break;
        case 19: m_pool = optarg; break;
        case 20: m_dpool = optarg; break;
        case 21: m_rpf = optarg; break;
        case 22: m_drpf = optarg; break;
        default:
            fprintf(stderr, "usage: %s [-o out.fco] [-q quality] [-z level] [-V n] [--pattern p] "
                            "image.pgm ...\n", argv[0]);
            return 2;
        }
    fiasco_set_verbosity((fiasco_verbosity_e) verbose);
    if (lim_states || lim_level) {
        unsigned s, l;
        fiasco_amd_get_limits(&s, &l);
        if (!fiasco_amd_set_limits(lim_states ? (unsigned) lim_states : s,
                                   lim_level ? (unsigned) lim_level : l)) {
            fprintf(stderr, "%s\n", fiasco_get_error_message());
            return 1;
        }
    }
    o = fiasco_c_options_new();
#define CK(x) do { if (!(x)) { fprintf(stderr, "%s\n", fiasco_get_error_message()); return 1; } } while (0)
    CK(fiasco_c_options_set_frame_pattern(o, pattern));
    if (basis) CK(fiasco_c_options_set_basisfile(o, basis));   /* default basis is compiled in */
    CK(fiasco_c_options_set_chroma_quality(o, chroma_q, cdict > 0 ? (unsigned) cdict : 0));
    CK(fiasco_c_options_set_smoothing(o, smooth > 0 ? smooth : 0));
    CK(fiasco_c_options_set_progress_meter(o, (fiasco_progress_e) (progress > 0 ? progress : 0)));
    if (*title) CK(fiasco_c_options_set_title(o, title));
    if (*comment) CK(fiasco_c_options_set_comment(o, comment));
    CK(fiasco_c_options_set_tiling(o, FIASCO_TILING_VARIANCE_DSC, tile_e > 0 ? (unsigned) tile_e : 0));
    if (optimize <= 0)
        CK(fiasco_c_options_set_optimizations(o, 6, 10, 3, dict > 0 ? (unsigned) dict : 0, 0));
    else
        CK(fiasco_c_options_set_optimizations(o, 4, 12, 5, dict > 0 ? (unsigned) dict : 0,
                                              (unsigned) (optimize - 1)));
    CK(fiasco_c_options_set_prediction(o, pred, pmin > 0 ? (unsigned) pmin : 0,
                                       pmax > 0 ? (unsigned) pmax : 0));
    CK(fiasco_c_options_set_quantization(o, rpf_m > 0 ? (unsigned) rpf_m : 0, to_range(rpf_r),
                                         dc_m > 0 ? (unsigned) dc_m : 0, to_range(dc_r)));
    if (m_pool || m_dpool || m_rpf || m_drpf) CK(fiasco_amd_c_options_set_models(o, m_pool, m_dpool, m_rpf, m_drpf));
    n = argc - optind;
    inputs = (const char **) calloc((size_t) n + 1, sizeof *inputs);
    for (i = 0; i < n; i++) inputs[i] = argv[optind + i];
    inputs[n] = NULL;
    if (!fiasco_coder(n ? inputs : NULL, outname, quality, o)) {
        fprintf(stderr, "%s\n", fiasco_get_error_message());
        return 1;
    }
    fiasco_c_options_delete(o);
    free(inputs);
    return 0;
}
