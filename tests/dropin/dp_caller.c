/*
 * tests/dropin/dp_caller.c -- a plain-C caller of the data-parallel path: one vit_engine per listed device, device-resident
 * forwards through vit_engine_forward_device(), the per-image top-1 records gathered with vit_dp_gather_top1() (RCCL
 * all-gather on device memory, include/vit_dp.h), checked on the host against every engine's own records.
 *
 *     dp_caller <images per device> <device> [<device> ...]        (a reduced model with seeded synthetic weights and images)
 *
 * Shard axis: the reference's image loop (ViT_opencl.c:802, ViT_seq.c:354) -- image i of the batch goes to device i / per.
 * Prints "dp_caller ok devices=N images=M" and returns 0, or a message naming what differed and a non-zero code.
 * Test infrastructure (tests/test_gpu_dropin.py compiles and runs it); links -lvit_mi355x -lvit_mi355x_dp.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vit_dp.h"
#include "vit_engine.h"
#include "vit_hip_kernels.h"
#include "vit_io.h"

#define MAXDEV 16
#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != 0) {                                                              \
            fprintf(stderr, "dp_caller: %s failed with %d (line %d)\n", #call, rc_, __LINE__); \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 3 || argc - 2 > MAXDEV) {
        fprintf(stderr, "usage: dp_caller <images per device> <device> [<device> ...]\n");
        return 64;
    }
    const int per = atoi(argv[1]), nd = argc - 2;
    int devices[MAXDEV];
    for (int d = 0; d < nd; ++d) devices[d] = atoi(argv[2 + d]);

    vit_config cfg = vit_config_b16(); /* reduced model: 64x64 images, 17 tokens, D = 192, 3 layers, 100 classes */
    cfg.img_size = 64; cfg.num_classes = 100; cfg.embed_dim = 192; cfg.depth = 3; cfg.num_heads = 3; cfg.hidden_dim = 768;
    const int count = VIT_WEIGHT_COUNT(cfg.depth);
    Network *net = (Network *)calloc((size_t)count, sizeof(Network));
    if (!net || vit_synth_weights(&cfg, 7, net, count) != 0) return 3;
    ImageData *images = vit_synth_images(&cfg, per * nd, 8);
    if (!images) return 3;
    const size_t img_elems = (size_t)cfg.in_chans * cfg.img_size * cfg.img_size;

    vit_engine *eng[MAXDEV];
    vithip_stream_t stream[MAXDEV];
    float *d_images[MAXDEV], *d_probs[MAXDEV];
    int *d_top1[MAXDEV], *d_all[MAXDEV];
    for (int d = 0; d < nd; ++d) {
        vit_engine_options opt;
        vit_engine_default_options(&opt);
        opt.device = devices[d];
        opt.max_batch = per;
        CHECK(vit_engine_create(&eng[d], &cfg, &opt));
        if (d == 0) CHECK(vit_engine_load_weights(eng[0], net, count));
        else CHECK(vit_engine_copy_weights(eng[d], eng[0]));           /* device to device: one upload for the node */
        CHECK(vithip_set_device(devices[d]));
        CHECK(vithip_stream_create(&stream[d]));
        CHECK(vithip_malloc((void **)&d_images[d], (size_t)per * img_elems * sizeof(float)));
        CHECK(vithip_malloc((void **)&d_probs[d], (size_t)per * cfg.num_classes * sizeof(float)));
        CHECK(vithip_malloc((void **)&d_top1[d], (size_t)2 * per * sizeof(int)));              /* [labels | probability bits] */
        CHECK(vithip_malloc((void **)&d_all[d], (size_t)nd * 2 * per * sizeof(int)));
        for (int i = 0; i < per; ++i)
            CHECK(vithip_memcpy_h2d(d_images[d] + (size_t)i * img_elems, images[d * per + i].data, img_elems * sizeof(float), stream[d]));
    }
    vit_dp *dp = NULL;
    CHECK(vit_dp_create(&dp, devices, nd));
    if (vit_dp_size(dp) != nd) return 4;

    /* forwards on every device's own stream, then ONE grouped all-gather enqueued behind them: no host synchronisation between */
    for (int d = 0; d < nd; ++d)
        CHECK(vit_engine_forward_device(eng[d], d_images[d], per, d_probs[d], d_top1[d], (float *)(d_top1[d] + per), stream[d]));
    if (vit_dp_gather_top1(dp, (const void *const *)d_top1, (void *const *)d_all, (size_t)per, (void *const *)stream) != 0) {
        fprintf(stderr, "dp_caller: %s\n", vit_dp_last_error(dp));
        return 5;
    }

    int *own = (int *)malloc((size_t)nd * 2 * per * sizeof(int)), *got = (int *)malloc((size_t)nd * 2 * per * sizeof(int));
    float *probs = (float *)malloc((size_t)per * cfg.num_classes * sizeof(float));
    if (!own || !got || !probs) return 3;
    for (int d = 0; d < nd; ++d) {   /* every engine's own records, and what its probabilities say they should be */
        CHECK(vithip_set_device(devices[d]));
        CHECK(vithip_stream_sync(stream[d]));
        CHECK(vithip_memcpy_d2h(own + (size_t)d * 2 * per, d_top1[d], (size_t)2 * per * sizeof(int), stream[d]));
        CHECK(vithip_memcpy_d2h(probs, d_probs[d], (size_t)per * cfg.num_classes * sizeof(float), stream[d]));
        CHECK(vithip_stream_sync(stream[d]));
        for (int i = 0; i < per; ++i) {
            const float *p = probs + (size_t)i * cfg.num_classes;
            const int label = vit_argmax(p, cfg.num_classes);
            float pbits;
            memcpy(&pbits, &own[(size_t)d * 2 * per + per + i], sizeof(float));
            if (own[(size_t)d * 2 * per + i] != label || pbits != p[label]) {
                fprintf(stderr, "dp_caller: device %d image %d: record (%d, %g) but probabilities say (%d, %g)\n", devices[d], i,
                        own[(size_t)d * 2 * per + i], pbits, label, p[label]);
                return 6;
            }
        }
    }
    for (int d = 0; d < nd; ++d) {   /* every device's gathered buffer holds every device's records, in rank order */
        CHECK(vithip_set_device(devices[d]));
        CHECK(vithip_memcpy_d2h(got, d_all[d], (size_t)nd * 2 * per * sizeof(int), stream[d]));
        CHECK(vithip_stream_sync(stream[d]));
        if (memcmp(got, own, (size_t)nd * 2 * per * sizeof(int)) != 0) {
            fprintf(stderr, "dp_caller: the gathered records on device %d differ from the devices' own\n", devices[d]);
            return 7;
        }
    }
    printf("dp_caller ok devices=%d images=%d\n", nd, per * nd);
    vit_dp_destroy(dp);
    for (int d = 0; d < nd; ++d) {
        CHECK(vithip_set_device(devices[d]));
        vithip_free(d_images[d]); vithip_free(d_probs[d]); vithip_free(d_top1[d]); vithip_free(d_all[d]);
        vithip_stream_destroy(stream[d]);
        vit_engine_destroy(eng[d]);
    }
    free_image_data(images);
    free_weights(net, count);
    free(net); free(own); free(got); free(probs);
    return 0;
}
