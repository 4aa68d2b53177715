/*
 * host/vit_facade.c -- the reference-shaped global-singleton API over vit_engine.
 *
 * initialize_opencl / ViT_opencl / Release_opencl of the reference (ViT_opencl.c:74-113,785-883)
 * become initialize_hip / ViT_hip / Release_hip; the reference's names are exported as aliases.
 * Error convention of CHECK_ERROR (ViT_opencl.h:7-11): print and exit(EXIT_FAILURE).
 *
 * Devices.  The reference initialises exactly one device (ViT_opencl.c:74-101) and loops over its images one by
 * one (ViT_opencl.c:802).  Here VIT_HIP_DEVICES ("all", or a comma list of ordinals such as "0,1,2,3") selects N
 * devices of the node: the facade keeps one vit_engine per device, uploads the weights to the first and replicates
 * them device-to-device (xGMI) to the others, and a forward splits image[0..n) into N contiguous slices -- the
 * reference's image loop, cut across devices -- each driven by its own host thread through the pinned, double-buffered
 * host path of its engine.  Images are independent, so no device ever waits for another.  Without VIT_HIP_DEVICES the
 * facade is the reference's single device (VIT_HIP_DEVICE, default 0).
 */
#include "ViT_hip.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vit_engine.h"
#include "vit_hip_kernels.h"

#define VIT_MAX_DEVICES 16

static struct {
    int n_dev;
    vit_engine *engine[VIT_MAX_DEVICES];
    int device[VIT_MAX_DEVICES];
    const Network *cached_weights; /* weights already resident on the device(s) */
    int image_loaded;              /* weights came from a cache file (ViT_hip_load_weight_cache): any `networks` is accepted */
} g_vit;

#define DIE(...)                                       \
    do {                                               \
        printf("[%s:%d] ", __FILE__, __LINE__);        \
        printf(__VA_ARGS__);                           \
        printf("\n");                                  \
        exit(EXIT_FAILURE);                            \
    } while (0)

static int env_int(const char *name, int fallback) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

/* VIT_HIP_DEVICES -> ordinals; returns the count (0 = variable unset or empty) */
static int parse_devices(int *out) {
    const char *v = getenv("VIT_HIP_DEVICES");
    if (!v || !*v) return 0;
    int n = 0;
    if (!strcmp(v, "all") || !strcmp(v, "ALL")) {
        int ndev = 0;
        if (vithip_device_count(&ndev) != 0 || ndev < 1) DIE("initialize_hip: no HIP device");
        for (int d = 0; d < ndev && n < VIT_MAX_DEVICES; ++d) out[n++] = d;
        return n;
    }
    const char *p = v;
    while (*p) {
        char *end = NULL;
        const long d = strtol(p, &end, 10);
        if (end == p || d < 0) DIE("initialize_hip: cannot parse VIT_HIP_DEVICES=\"%s\" (\"all\" or e.g. \"0,1,2,3\")", v);
        if (n == VIT_MAX_DEVICES) DIE("initialize_hip: more than %d devices in VIT_HIP_DEVICES", VIT_MAX_DEVICES);
        out[n++] = (int)d;
        p = end;
        while (*p == ',' || *p == ' ') ++p;
    }
    return n;
}

void initialize_hip(void) {
    if (g_vit.n_dev) return;
    vit_engine_options opt;
    vit_engine_default_options(&opt);
    opt.max_batch = env_int("VIT_HIP_MAX_BATCH", opt.max_batch);
    opt.prune_last_layer = env_int("VIT_HIP_PRUNE_LAST_LAYER", 0); /* bit-identical output, 7 % less arithmetic */
    const char *dt = getenv("VIT_HIP_DTYPE"); /* "bf16" = bf16 matrix pipe (own tolerance); default: the reference's fp32 */
    if (dt && (!strcmp(dt, "bf16") || !strcmp(dt, "BF16"))) opt.dtype = VIT_DTYPE_BF16;
    /* concurrent sub-batches (bit-identical): fp32 1 -- its GEMMs balance their own tails and assume their workgroups resident;
     * bf16 2 -- the HBM-bound LayerNorm / residual kernels of one lane overlap the other's GEMMs (+6 %) */
    opt.lanes = env_int("VIT_HIP_LANES", opt.dtype == VIT_DTYPE_BF16 ? 2 : 1);
    opt.ln_fold = env_int("VIT_HIP_LN_FOLD", 0); /* 0 auto (encoder LayerNorms folded into the GEMMs either side), -1 a LayerNorm kernel each */
    vit_config cfg = vit_config_b16(); /* the reference's compile-time model (ViT_opencl.c:12-23) */
    int n = parse_devices(g_vit.device);
    if (n == 0) {
        g_vit.device[0] = env_int("VIT_HIP_DEVICE", opt.device);
        n = 1;
    }
    for (int d = 0; d < n; ++d) {
        opt.device = g_vit.device[d];
        const int rc = vit_engine_create(&g_vit.engine[d], &cfg, &opt);
        if (rc != VIT_OK)
            DIE("initialize_hip failed on device %d: %s", opt.device,
                g_vit.engine[d] ? vit_engine_last_error(g_vit.engine[d]) : "no engine");
        g_vit.n_dev = d + 1;
    }
    g_vit.cached_weights = NULL;
    g_vit.image_loaded = 0;
}

/* ---- one host thread per device ------------------------------------------------------------ */

typedef struct {
    int d;                 /* slot in g_vit */
    int mode;              /* 0 = replicate weights from engine 0, 1 = forward a slice */
    const float *const *images;
    float *const *probs;
    int n;
    int rc;
} dev_job;

static void *dev_thread(void *arg) {
    dev_job *j = (dev_job *)arg;
    vit_engine *e = g_vit.engine[j->d];
    if (j->mode == 0) j->rc = vit_engine_copy_weights(e, g_vit.engine[0]);
    else j->rc = vit_engine_forward_host(e, j->images, j->n, j->probs);
    return NULL;
}

/* run jobs[0..n): jobs[0] on the calling thread, every other job on a thread of its own (one host thread per device);
 * exits through DIE(what ...) when a job failed */
static void run_jobs(dev_job *jobs, int n, const char *what) {
    pthread_t th[VIT_MAX_DEVICES];
    int started[VIT_MAX_DEVICES] = {0};
    for (int i = 1; i < n; ++i)
        started[i] = pthread_create(&th[i], NULL, dev_thread, &jobs[i]) == 0;
    if (n > 0) dev_thread(&jobs[0]);
    for (int i = 1; i < n; ++i) {
        if (started[i]) pthread_join(th[i], NULL);
        else dev_thread(&jobs[i]); /* no thread to be had: do it here, later but correct */
    }
    for (int i = 0; i < n; ++i)
        if (jobs[i].rc != VIT_OK)
            DIE("%s failed on device %d: %s", what, g_vit.device[jobs[i].d], vit_engine_last_error(g_vit.engine[jobs[i].d]));
}

static void replicate_weights(void) {
    if (g_vit.n_dev < 2) return;
    dev_job jobs[VIT_MAX_DEVICES];
    memset(jobs, 0, sizeof(jobs));
    for (int d = 1; d < g_vit.n_dev; ++d) { jobs[d - 1].d = d; jobs[d - 1].mode = 0; }
    run_jobs(jobs, g_vit.n_dev - 1, "ViT_hip (weight replication)");
}

void ViT_hip(ImageData *image, Network *networks, float **prb) {
    if (!g_vit.n_dev) initialize_hip(); /* the reference requires the explicit call; be lenient */
    if (!image || !prb || (!networks && !g_vit.image_loaded)) DIE("ViT_hip: NULL argument");
    const vit_config *cfg = vit_engine_config(g_vit.engine[0]);
    const int n = image[0].n; /* ViT_opencl.c:802 loops i < image->n */
    if (n <= 0) return;
    for (int i = 0; i < n; ++i)
        if (image[i].c != cfg->in_chans || image[i].h != cfg->img_size || image[i].w != cfg->img_size)
            DIE("ViT_hip: image %d is %dx%dx%d, the model needs %dx%dx%d", i, image[i].c, image[i].h, image[i].w,
                cfg->in_chans, cfg->img_size, cfg->img_size);
    if (!g_vit.image_loaded && g_vit.cached_weights != networks) {
        const int rc = vit_engine_load_weights(g_vit.engine[0], networks, VIT_WEIGHT_COUNT(cfg->depth));
        if (rc != VIT_OK) DIE("ViT_hip (weight upload) failed: %s", vit_engine_last_error(g_vit.engine[0]));
        replicate_weights();
        g_vit.cached_weights = networks;
    }
    const float **imgs = (const float **)malloc(sizeof(float *) * (size_t)n);
    if (!imgs) DIE("ViT_hip: out of memory");
    for (int i = 0; i < n; ++i) imgs[i] = image[i].data;
    /* contiguous split of the reference's image loop: device slot d forwards images [n*d/N, n*(d+1)/N) */
    dev_job jobs[VIT_MAX_DEVICES];
    memset(jobs, 0, sizeof(jobs));
    int nj = 0;
    for (int d = 0; d < g_vit.n_dev; ++d) {
        const int lo = (int)((long)n * d / g_vit.n_dev), hi = (int)((long)n * (d + 1) / g_vit.n_dev);
        if (hi <= lo) continue;
        jobs[nj].d = d; jobs[nj].mode = 1; jobs[nj].images = imgs + lo; jobs[nj].probs = prb + lo; jobs[nj].n = hi - lo;
        nj++;
    }
    run_jobs(jobs, nj, "ViT_hip (forward)");
    free(imgs);
}

void Release_hip(void) {
    for (int d = 0; d < g_vit.n_dev; ++d) {
        vit_engine_destroy(g_vit.engine[d]);
        g_vit.engine[d] = NULL;
    }
    g_vit.n_dev = 0;
    g_vit.cached_weights = NULL;
    g_vit.image_loaded = 0;
}

void ViT_hip_invalidate_weights(void) {
    g_vit.cached_weights = NULL;
    g_vit.image_loaded = 0;
}

int ViT_hip_device_count(void) { return g_vit.n_dev; }

int ViT_hip_load_weight_cache(const char *path, const char *source_dir) {
    if (!g_vit.n_dev) initialize_hip();
    vit_weight_image img;
    if (vit_weight_image_load(&img, path, vit_engine_config(g_vit.engine[0]), source_dir) != 0) return -1;
    const int rc = vit_engine_load_weight_image(g_vit.engine[0], &img);
    vit_weight_image_free(&img);
    if (rc != VIT_OK) DIE("ViT_hip_load_weight_cache failed: %s", vit_engine_last_error(g_vit.engine[0]));
    replicate_weights();
    g_vit.image_loaded = 1;
    g_vit.cached_weights = NULL;
    return 0;
}

int ViT_hip_save_weight_cache(const char *path, const char *source_dir) {
    if (!g_vit.n_dev || (!g_vit.cached_weights && !g_vit.image_loaded)) return -1;
    vit_weight_image img;
    if (vit_engine_read_weight_image(g_vit.engine[0], &img) != VIT_OK) return -1;
    const int rc = vit_weight_image_save(&img, path, source_dir);
    vit_weight_image_free(&img);
    return rc;
}

/* The reference's own symbol names, so its Main.c links unchanged. */
void initialize_opencl(void) { initialize_hip(); }
void ViT_opencl(ImageData *image, Network *networks, float **prb) { ViT_hip(image, networks, prb); }
void Release_opencl(void) { Release_hip(); }
