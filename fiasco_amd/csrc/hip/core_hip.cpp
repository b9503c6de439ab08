/*
 *  core_hip.cpp -- host launcher of the device frame coder: implements the seam
 *  fa_core_encode_frames() of fa_host.h on top of the HIP runtime.
 *
 *  For every job it carves one HBM slab (layout: frame_coder.h), uploads the int16 pixel
 *  plane and the basis automaton, launches ONE persistent kernel with one workgroup per
 *  frame (all frames of the call in flight at once), and copies the finished automaton
 *  back for the host stream writer.  There is no CPU fallback: without a usable GPU every
 *  job fails with an error message.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fa_host.h"
#include "frame_coder.h"
#include "libfiasco_amd_hip.h"

extern "C" void fc_launch(DevFrame *d_frames, unsigned n, hipStream_t stream);

static fiasco_amd_stats g_stats;

extern "C" void fiasco_amd_get_stats(fiasco_amd_stats *out) { *out = g_stats; }
extern "C" void fiasco_amd_reset_stats(void) { memset(&g_stats, 0, sizeof g_stats); }
extern "C" const char *fa_core_name(void) { return "hip-gfx950"; }

/* one process per GPU: bind this process's coder to a device of the node */
extern "C" int fiasco_amd_set_device(int device)
{
    if (hipSetDevice(device) != hipSuccess) {
        fa_set_error("libfiasco_amd: cannot select HIP device %d", device);
        return 0;
    }
    return 1;
}

#define HIPCK(call)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(errbuf, sizeof errbuf, "HIP error %s at %s:%d", hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                  \
            goto hip_fail;                                                                 \
        }                                                                                  \
    } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Slab {
    char  *base = nullptr;      /* device */
    size_t bytes = 0;
    int    P = 0;
};

struct Layout {
    size_t gram, diag, ipis, d5, img, imgT, norms, num, den, est, ipdo, used, tree, into, weight,
           final_d, level_of_state, domain_type, x, y, pool_states, pix16, total;
};

static Layout make_layout(int P, int NL, int NS, int NA, int NI, int il, size_t npix)
{
    Layout L;
    size_t o = 0;
#define CARVE(field, bytes) do { L.field = o; o = align_up(o + (bytes), 256); } while (0)
    CARVE(gram, (size_t) NL * P * P * 4);
    CARVE(diag, (size_t) NL * P * 4);
    CARVE(ipis, (size_t) NS * P * 4);
    CARVE(d5, (size_t) NA * P * 4);
    CARVE(img, (size_t) P * NI * 4);
    CARVE(imgT, ((size_t) 1 << il) * P * 4);
    CARVE(norms, (size_t) NS * 4);
    CARVE(num, (size_t) P * 4);
    CARVE(den, (size_t) P * 4);
    CARVE(est, (size_t) P * 4);
    CARVE(ipdo, (size_t) FC_MAXED * P * 4);
    CARVE(used, (size_t) P);
    CARVE(tree, (size_t) 2 * P * 2);
    CARVE(into, (size_t) 12 * P * 2);
    CARVE(weight, (size_t) 12 * P * 4);
    CARVE(final_d, (size_t) P * 4);
    CARVE(level_of_state, (size_t) P);
    CARVE(domain_type, (size_t) P);
    CARVE(x, (size_t) 2 * P * 2);
    CARVE(y, (size_t) 2 * P * 2);
    CARVE(pool_states, (size_t) (P + 8) * 2);
    CARVE(pix16, npix * 2);
#undef CARVE
    L.total = o;
    return L;
}

static int device_supported(const fa_job *job, char *why, size_t n)
{
    const fa_cparams *cp = &job->cp;
    if (job->image->color) { snprintf(why, n, "colour frames are not supported by the device coder yet"); return 0; }
    if (cp->images_level != 5 || cp->lc_min_level <= cp->images_level) {
        snprintf(why, n, "device coder needs images_level 5 and min block level > 5 (CLI -z 0 levels)");
        return 0;
    }
    if (cp->second_domain_block || cp->check_for_underflow || cp->check_for_overflow || cp->full_search) {
        snprintf(why, n, "optimisation levels > 0 are not supported by the device coder yet");
        return 0;
    }
    if (cp->lc_max_level > 10) { snprintf(why, n, "max block level > 10 is not supported by the device coder yet"); return 0; }
    if (cp->rpf.mantissa_bits > 5 || cp->dc_rpf.mantissa_bits > 5) {
        snprintf(why, n, "RPF mantissa > 5 bits is not supported by the device coder yet");
        return 0;
    }
    return 1;
}

extern "C" int fa_core_encode_frames(unsigned n, fa_job *jobs)
{
    char errbuf[200] = "";
    int ndev = 0, good = 0;
    std::vector<Slab> slabs(n);
    std::vector<Layout> lay(n);
    std::vector<DevFrame> hf(n);
    std::vector<int> todo;
    DevFrame *d_frames = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t free_b = 0, total_b = 0;

    for (unsigned i = 0; i < n; i++) { jobs[i].status = 0; jobs[i].errmsg[0] = 0; }
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        for (unsigned i = 0; i < n; i++)
            snprintf(jobs[i].errmsg, sizeof jobs[i].errmsg,
                     "libfiasco_amd: no HIP device available (the hot path has no CPU fallback)");
        return 0;
    }
    HIPCK(hipStreamCreate(&stream));
    HIPCK(hipEventCreate(&ev0));
    HIPCK(hipEventCreate(&ev1));
    HIPCK(hipMalloc((void **) &d_frames, sizeof(DevFrame) * (n ? n : 1)));

    for (unsigned i = 0; i < n; i++) {
        if (!device_supported(&jobs[i], jobs[i].errmsg, sizeof jobs[i].errmsg)) continue;
        todo.push_back((int) i);
        /* first guess of the state capacity: the partition needs one state per bintree
         * node above the largest block level plus the splits inside the blocks */
        const fa_cparams *cp = &jobs[i].cp;
        unsigned bw = fa_width_of_level(cp->lc_max_level), bh = fa_height_of_level(cp->lc_max_level);
        size_t blocks = (size_t) ((jobs[i].image->width + bw - 1) / bw) * ((jobs[i].image->height + bh - 1) / bh);
        size_t guess = blocks + blocks / 2 + 512;
        if (guess > cp->limit_states) guess = cp->limit_states;
        slabs[i].P = (int) align_up(guess, 64);
    }

    while (!todo.empty()) {
        /* ---- admit as many frames as fit in free HBM (keep 2 GiB headroom) ---- */
        std::vector<int> batch;
        HIPCK(hipMemGetInfo(&free_b, &total_b));
        size_t budget = free_b > ((size_t) 2 << 30) ? free_b - ((size_t) 2 << 30) : 0;
        for (size_t k = 0; k < todo.size(); k++) {
            int i = todo[k];
            const fa_cparams *cp = &jobs[i].cp;
            int il = (int) cp->images_level, P = slabs[i].P;
            int NL = (int) (cp->lc_max_level - cp->images_level + 1);
            int NS = (int) fa_size_of_tree(cp->products_level);
            int NA = 1 << (cp->lc_max_level - cp->images_level);
            int NI = (int) fa_size_of_tree(cp->images_level);
            lay[i] = make_layout(P, NL, NS, NA, NI, il, (size_t) jobs[i].image->width * jobs[i].image->height);
            if (lay[i].total > budget) {
                if (batch.empty()) {
                    snprintf(jobs[i].errmsg, sizeof jobs[i].errmsg,
                             "frame needs %.1f GiB of HBM, only %.1f GiB free",
                             lay[i].total / 1073741824.0, free_b / 1073741824.0);
                    todo.erase(todo.begin() + (long) k);
                    k--;
                }
                continue;
            }
            budget -= lay[i].total;
            batch.push_back(i);
        }
        if (batch.empty()) break;

        /* ---- build slabs ---- */
        for (size_t b = 0; b < batch.size(); b++) {
            int i = batch[b];
            fa_job *job = &jobs[i];
            const fa_cparams *cp = &job->cp;
            const fa_wfa *w = job->wfa;
            const Layout &L = lay[i];
            const int P = slabs[i].P;
            DevFrame &F = hf[b];
            memset(&F, 0, sizeof F);
            HIPCK(hipMalloc((void **) &slabs[i].base, L.total));
            slabs[i].bytes = L.total;
            char *base = slabs[i].base;
            F.price = cp->price;
            F.lc_min = (int) cp->lc_min_level; F.lc_max = (int) cp->lc_max_level;
            F.images_level = (int) cp->images_level; F.max_elements = (int) cp->max_elements;
            F.level = (int) cp->level; F.width = (int) job->image->width; F.height = (int) job->image->height;
            F.pool_max = (int) cp->pool_max_states; F.limit_states = (int) cp->limit_states;
            F.ML = (int) cp->limit_level;
            F.rpf_mant = (int) cp->rpf.mantissa_bits; F.dc_mant = (int) cp->dc_rpf.mantissa_bits;
            F.rpf_range = cp->rpf.range; F.dc_range = cp->dc_rpf.range;
            F.P = P;
            F.NL = (int) (cp->lc_max_level - cp->images_level + 1);
            F.NS = (int) fa_size_of_tree(cp->products_level);
            F.NA = 1 << (cp->lc_max_level - cp->images_level);
            F.NI = (int) fa_size_of_tree(cp->images_level);
            F.dcs = 1 << (1 + F.dc_mant); F.sy = 1 << (1 + F.rpf_mant);
            F.coeff_size = (F.lc_max - F.lc_min + 1) * F.sy + F.dcs;
            F.coeff_nt = F.lc_max - F.lc_min + 2;
            F.basis_states = (int) w->basis_states;
            F.pix16 = (const int16_t *) (base + L.pix16);
            F.gram = (float *) (base + L.gram); F.diag = (float *) (base + L.diag);
            F.ipis = (float *) (base + L.ipis); F.d5 = (float *) (base + L.d5);
            F.img = (float *) (base + L.img); F.imgT = (float *) (base + L.imgT);
            F.norms = (float *) (base + L.norms);
            F.num = (float *) (base + L.num); F.den = (float *) (base + L.den);
            F.est = (float *) (base + L.est); F.ipdo = (float *) (base + L.ipdo);
            F.used = (uint8_t *) (base + L.used);
            F.tree = (int16_t *) (base + L.tree); F.into = (int16_t *) (base + L.into);
            F.weight = (float *) (base + L.weight); F.final_d = (float *) (base + L.final_d);
            F.level_of_state = (uint8_t *) (base + L.level_of_state);
            F.domain_type = (uint8_t *) (base + L.domain_type);
            F.x = (uint16_t *) (base + L.x); F.y = (uint16_t *) (base + L.y);
            F.pool_states = (int16_t *) (base + L.pool_states);
            if (F.coeff_size > FC_MAXCOEFF || F.dcs > FC_MAXSYM || F.sy > FC_MAXSYM || F.ML > 26) {
                snprintf(job->errmsg, sizeof job->errmsg, "coefficient model too large for the device coder");
                continue;
            }
            /* everything but the Gram tables starts zeroed (the reference callocs) */
            HIPCK(hipMemsetAsync(base + L.diag, 0, L.total - L.diag, stream));
            HIPCK(hipMemcpyAsync(base + L.pix16, job->image->pixels[0],
                                 (size_t) F.width * F.height * 2, hipMemcpyHostToDevice, stream));
            {   /* basis automaton, SoA */
                std::vector<int16_t> tree((size_t) 2 * P, (int16_t) FA_RANGE), into((size_t) 12 * P, (int16_t) FA_NO_EDGE);
                std::vector<float> weight((size_t) 12 * P, 0.0f), fin((size_t) P, 0.0f);
                std::vector<uint8_t> dt((size_t) P, 0), los((size_t) P, 0);
                for (unsigned s = 0; s < w->basis_states; s++) {
                    fin[s] = w->final_distribution[s];
                    dt[s] = w->domain_type[s];
                    los[s] = 0xff;
                    for (int l = 0; l < 2; l++) {
                        tree[(size_t) l * P + s] = FA_TREE(w, s, l);
                        for (int e = 0; e < 6; e++) {
                            into[(size_t) (l * 6 + e) * P + s] = FA_INTO(w, s, l, e);
                            weight[(size_t) (l * 6 + e) * P + s] = FA_WEIGHT(w, s, l, e);
                            if (FA_INTO(w, s, l, e) == FA_NO_EDGE) break;
                        }
                    }
                }
                HIPCK(hipMemcpy(base + L.tree, tree.data(), tree.size() * 2, hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(base + L.into, into.data(), into.size() * 2, hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(base + L.weight, weight.data(), weight.size() * 4, hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(base + L.final_d, fin.data(), fin.size() * 4, hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(base + L.domain_type, dt.data(), dt.size(), hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(base + L.level_of_state, los.data(), los.size(), hipMemcpyHostToDevice));
            }
        }
        FcTrace *d_trace = nullptr;
        const char *trace_path = getenv("FIASCO_AMD_TRACE");
        const int trace_cap = 400000;
        if (trace_path && !batch.empty()) {
            HIPCK(hipMalloc((void **) &d_trace, sizeof(FcTrace) * trace_cap));
            hf[0].trace = d_trace; hf[0].trace_cap = trace_cap; hf[0].trace_n = 0;
        }
        HIPCK(hipMemcpyAsync(d_frames, hf.data(), sizeof(DevFrame) * batch.size(), hipMemcpyHostToDevice, stream));

        /* ---- one persistent launch: one workgroup per frame ---- */
        HIPCK(hipEventRecord(ev0, stream));
        fc_launch(d_frames, (unsigned) batch.size(), stream);
        HIPCK(hipGetLastError());
        HIPCK(hipEventRecord(ev1, stream));
        HIPCK(hipStreamSynchronize(stream));
        {
            float ms = 0;
            HIPCK(hipEventElapsedTime(&ms, ev0, ev1));
            g_stats.kernel_ms += ms;
            g_stats.launches += 1;
        }
        HIPCK(hipMemcpy(hf.data(), d_frames, sizeof(DevFrame) * batch.size(), hipMemcpyDeviceToHost));
        if (d_trace) {
            std::vector<FcTrace> tr((size_t) hf[0].trace_n);
            HIPCK(hipMemcpy(tr.data(), d_trace, sizeof(FcTrace) * tr.size(), hipMemcpyDeviceToHost));
            FILE *tf = fopen(trace_path, "wb");
            if (tf) { fwrite(tr.data(), sizeof(FcTrace), tr.size(), tf); fclose(tf); }
            (void) hipFree(d_trace);
        }

        /* ---- collect ---- */
        std::vector<int> retry;
        for (size_t b = 0; b < batch.size(); b++) {
            int i = batch[b];
            fa_job *job = &jobs[i];
            DevFrame &F = hf[b];
            const Layout &L = lay[i];
            const int P = slabs[i].P;
            char *base = slabs[i].base;
            if (F.status == FC_ERR_CAPACITY && (unsigned) P < align_up(job->cp.limit_states, 64)) {
                size_t np = align_up((size_t) P * 2, 64), cap = align_up(job->cp.limit_states, 64);
                slabs[i].P = (int) (np > cap ? cap : np);
                retry.push_back(i);
            } else if (F.status == FC_OK) {
                fa_wfa *w = job->wfa;
                unsigned ns = (unsigned) F.states;
                std::vector<int16_t> tree((size_t) 2 * P), into((size_t) 12 * P);
                std::vector<float> weight((size_t) 12 * P), fin((size_t) P);
                std::vector<uint8_t> dt((size_t) P), los((size_t) P);
                std::vector<uint16_t> xs((size_t) 2 * P), ys((size_t) 2 * P);
                HIPCK(hipMemcpy(tree.data(), base + L.tree, tree.size() * 2, hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(into.data(), base + L.into, into.size() * 2, hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(weight.data(), base + L.weight, weight.size() * 4, hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(fin.data(), base + L.final_d, fin.size() * 4, hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(dt.data(), base + L.domain_type, dt.size(), hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(los.data(), base + L.level_of_state, los.size(), hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(xs.data(), base + L.x, xs.size() * 2, hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(ys.data(), base + L.y, ys.size() * 2, hipMemcpyDeviceToHost));
                for (unsigned s = w->basis_states; s < ns; s++) {
                    w->final_distribution[s] = fin[s];
                    w->domain_type[s] = dt[s];
                    w->level_of_state[s] = los[s];
                    w->delta_state[s] = 0;
                    for (int l = 0; l < 2; l++) {
                        FA_TREE(w, s, l) = tree[(size_t) l * P + s];
                        w->x[s * 2 + l] = xs[(size_t) l * P + s];
                        w->y[s * 2 + l] = ys[(size_t) l * P + s];
                        w->y_state[s * 2 + l] = FA_RANGE;
                        w->y_column[s * 2 + l] = 0;
                        w->prediction[s * 2 + l] = 0;
                        for (int e = 0; e < 6; e++) {
                            FA_INTO(w, s, l, e) = into[(size_t) (l * 6 + e) * P + s];
                            FA_WEIGHT(w, s, l, e) = weight[(size_t) (l * 6 + e) * P + s];
                            if (FA_INTO(w, s, l, e) == FA_NO_EDGE) break;
                        }
                    }
                }
                w->states = ns;
                w->root_state = (unsigned) F.root_state;
                job->stats[0].costs = F.costs; job->stats[0].err = F.err;
                job->stats[0].tree_bits = F.tree_bits; job->stats[0].matrix_bits = F.matrix_bits;
                job->stats[0].weights_bits = F.weights_bits;
                job->lc_min_level_out = job->cp.lc_min_level;
                job->status = 1;
                good++;
                g_stats.frames += 1;
                g_stats.bytes_mp += F.bytes_mp; g_stats.bytes_img += F.bytes_img;
                g_stats.bytes_gram += F.bytes_gram;
                g_stats.n_mp += F.n_mp; g_stats.n_steps += F.n_steps; g_stats.n_blocks += F.n_blocks;
                g_stats.n_appends += F.n_appends; g_stats.n_fulleval += F.n_fulleval;
                g_stats.t_init += F.t_init; g_stats.t_approx += F.t_approx; g_stats.t_ipis += F.t_ipis;
                g_stats.t_append += F.t_append; g_stats.t_serial += F.t_serial; g_stats.t_total += F.t_total;
            } else {
                const char *msg = "device coder failed";
                if (F.status == FC_ERR_STATES) msg = "Maximum number of states reached!";
                else if (F.status == FC_ERR_NOROOT) msg = "No root state generated!";
                else if (F.status == FC_ERR_CAPACITY) msg = "Maximum number of states reached!";
                else if (F.status == FC_ERR_INTERNAL) msg = "device coder: recursion depth exceeded";
                snprintf(job->errmsg, sizeof job->errmsg, "%s", msg);
            }
            HIPCK(hipFree(slabs[i].base));
            slabs[i].base = nullptr;
        }
        /* drop finished frames from the work list, keep the ones that need more room */
        std::vector<int> next;
        for (size_t k = 0; k < todo.size(); k++) {
            int i = todo[k];
            bool in_batch = false;
            for (size_t b = 0; b < batch.size(); b++) if (batch[b] == i) in_batch = true;
            if (!in_batch) next.push_back(i);
        }
        for (size_t k = 0; k < retry.size(); k++) next.push_back(retry[k]);
        todo.swap(next);
    }
    goto cleanup;

hip_fail:
    for (unsigned i = 0; i < n; i++)
        if (!jobs[i].status && !jobs[i].errmsg[0]) snprintf(jobs[i].errmsg, sizeof jobs[i].errmsg, "%s", errbuf);
cleanup:
    for (unsigned i = 0; i < n; i++) if (slabs[i].base) (void) hipFree(slabs[i].base);
    if (d_frames) (void) hipFree(d_frames);
    if (ev0) (void) hipEventDestroy(ev0);
    if (ev1) (void) hipEventDestroy(ev1);
    if (stream) (void) hipStreamDestroy(stream);
    return good;
}
