/*
 * host/vit_facade.c -- the reference-shaped global-singleton API over vit_engine.
 *
 * initialize_opencl / ViT_opencl / Release_opencl of the reference (ViT_opencl.c:74-113,785-883)
 * become initialize_hip / ViT_hip / Release_hip; the reference's names are exported as aliases.
 * Error convention of CHECK_ERROR (ViT_opencl.h:7-11): print and exit(EXIT_FAILURE).
 */
#include "ViT_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vit_engine.h"

static struct {
    vit_engine *engine;
    const Network *cached_weights; /* weights already resident on the device */
} g_vit;

#define DIE_ON(rc, what)                                                                   \
    do {                                                                                   \
        if ((rc) != VIT_OK) {                                                              \
            printf("[%s:%d] %s failed: %s\n", __FILE__, __LINE__, (what),                  \
                   g_vit.engine ? vit_engine_last_error(g_vit.engine) : "no engine");      \
            exit(EXIT_FAILURE);                                                            \
        }                                                                                  \
    } while (0)

static int env_int(const char *name, int fallback) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

void initialize_hip(void) {
    if (g_vit.engine) return;
    vit_engine_options opt;
    vit_engine_default_options(&opt);
    opt.device = env_int("VIT_HIP_DEVICE", opt.device);
    opt.max_batch = env_int("VIT_HIP_MAX_BATCH", opt.max_batch);
    opt.lanes = env_int("VIT_HIP_LANES", 2); /* two concurrent sub-batches: bit-identical, +0.3-0.5 % img/s */
    opt.prune_last_layer = env_int("VIT_HIP_PRUNE_LAST_LAYER", 0); /* bit-identical output, 7 % less arithmetic */
    const char *dt = getenv("VIT_HIP_DTYPE"); /* "bf16" = bf16 matrix pipe (own tolerance); default: the reference's fp32 */
    if (dt && (!strcmp(dt, "bf16") || !strcmp(dt, "BF16"))) opt.dtype = VIT_DTYPE_BF16;
    vit_config cfg = vit_config_b16(); /* the reference's compile-time model (ViT_opencl.c:12-23) */
    int rc = vit_engine_create(&g_vit.engine, &cfg, &opt);
    DIE_ON(rc, "initialize_hip");
    g_vit.cached_weights = NULL;
}

void ViT_hip(ImageData *image, Network *networks, float **prb) {
    if (!g_vit.engine) initialize_hip(); /* the reference requires the explicit call; be lenient */
    if (!image || !networks || !prb) {
        printf("[%s:%d] ViT_hip: NULL argument\n", __FILE__, __LINE__);
        exit(EXIT_FAILURE);
    }
    const vit_config *cfg = vit_engine_config(g_vit.engine);
    const int n = image[0].n; /* ViT_opencl.c:802 loops i < image->n */
    if (n <= 0) return;
    for (int i = 0; i < n; ++i) {
        if (image[i].c != cfg->in_chans || image[i].h != cfg->img_size || image[i].w != cfg->img_size) {
            printf("[%s:%d] ViT_hip: image %d is %dx%dx%d, the model needs %dx%dx%d\n", __FILE__, __LINE__, i,
                   image[i].c, image[i].h, image[i].w, cfg->in_chans, cfg->img_size, cfg->img_size);
            exit(EXIT_FAILURE);
        }
    }
    if (g_vit.cached_weights != networks) {
        int rc = vit_engine_load_weights(g_vit.engine, networks, VIT_WEIGHT_COUNT(cfg->depth));
        DIE_ON(rc, "ViT_hip (weight upload)");
        g_vit.cached_weights = networks;
    }
    const float **imgs = (const float **)malloc(sizeof(float *) * (size_t)n);
    if (!imgs) {
        printf("[%s:%d] ViT_hip: out of memory\n", __FILE__, __LINE__);
        exit(EXIT_FAILURE);
    }
    for (int i = 0; i < n; ++i) imgs[i] = image[i].data;
    int rc = vit_engine_forward_host(g_vit.engine, imgs, n, prb);
    free(imgs);
    DIE_ON(rc, "ViT_hip (forward)");
}

void Release_hip(void) {
    if (!g_vit.engine) return;
    vit_engine_destroy(g_vit.engine);
    g_vit.engine = NULL;
    g_vit.cached_weights = NULL;
}

void ViT_hip_invalidate_weights(void) { g_vit.cached_weights = NULL; }

/* The reference's own symbol names, so its Main.c links unchanged. */
void initialize_opencl(void) { initialize_hip(); }
void ViT_opencl(ImageData *image, Network *networks, float **prb) { ViT_hip(image, networks, prb); }
void Release_opencl(void) { Release_hip(); }
