/*
 * host/vit_engine.c -- batched ViT forward, host orchestration in C over the HIP C-ABI.
 *
 * Follows the stage order of the reference's forward (ViT_seq.c:337-439 / ViT_opencl.c:785-883)
 * with the whole chunk of images as the GEMM M dimension:
 *
 *   patch_embed (conv_proj + flatten_transpose + class_token + pos_emb, one implicit GEMM)
 *   depth x { LN1 -> QKV GEMM -> fused attention -> out_proj GEMM (+bias +residual, in place)
 *             LN2 -> fc1 GEMM (+bias +GELU) -> fc2 GEMM (+bias +residual, in place) }
 *   LN on the class-token rows only -> head GEMM -> softmax + top-1
 *
 * Weights are validated and uploaded once (the reference re-uploads them per op per image,
 * e.g. ViT_opencl.c:136,630-631); activations never leave HBM between stages (the reference
 * reads every stage back to the host).
 */
#include "vit_engine.h"

#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <omp.h>
#include <string.h>

#include "vit_hip_kernels.h"
#include "vit_io.h"

#define VIT_MAX_LANES 4
#define MAX_EVENTS 4096   /* stage brackets kept in flight before they are read back */

struct vit_engine {
    vit_config cfg;
    vit_engine_options opt;
    int tokens;
    int n_weights;
    char err[512];

    vithip_stream_t stream;      /* engine-owned in-order stream */
    vithip_stream_t aux_stream[VIT_MAX_LANES - 1]; /* extra lanes of a chunk (forward_chunk) */
    vithip_event_t ev_fork, ev_join[VIT_MAX_LANES - 1];
    float *wblob;                /* all weights, ONE allocation: [fp32 section | bf16 GEMM operands] (vit_weight_image) */
    size_t wblob_bytes;          /* bytes of it that carry data (the allocation has a read-only tail pad behind) */
    float **w;                   /* device pointer per weight index */
    unsigned short *wblob16;     /* the bf16 section inside wblob (dtype bf16 only) */
    int fold;                    /* LayerNorm fold active (vit_engine_options.ln_fold) */
    unsigned short *wfold16;     /* per layer [gamma1-folded in_proj 3D x D | gamma2-folded fc1 H x D] (bf16 engines) */
    float *wfold32;              /* the same two operands as fp32 products gamma * W (fp32 engines) */
    float *wfoldf;               /* per layer [colsum qkv 3D | bias qkv 3D | colsum fc1 H | bias fc1 H] */
    float *ln_rows32;            /* fp32 engines: (rstd, mean) per token row [max_batch * tokens][2], then per class row [max_batch][2] */
    float *ln_part32;            /* ... and the residual GEMMs' scratch for them: [embed_dim / 64][rows][2] per lane (vithip_gemm_args.stats_partials) */
    int lane_cap;                /* most images one lane may hold (32-bit buffer offsets of the fp32 kernels) */
    void *gemm_ws[VIT_MAX_LANES]; /* per lane in use (= per stream): vithip_gemm_args.workspace handles */
    long handover_taken, handover_recomputed;
    int n_cus;                   /* compute units of the device */
    /* use_graph: the captured forward and what it was captured for */
    vithip_graph_t graph;
    const float *g_images; float *g_probs; int *g_label; float *g_prob; int g_n;
    unsigned short **w16;        /* per weight index; NULL for tensors that stay fp32 */
    int weights_loaded;

    /* workspace for max_batch images */
    float *x, *y, *qkv, *hbuf, *z, *logits;
    /* host-pointer path: double-buffered staging so that gather + H2D of piece i+1 overlap compute of piece i */
    float *in_stage[2], *out_stage[2];   /* device */
    float *pin_in[2], *pin_out[2];       /* pinned host */
    vithip_stream_t copy_stream;
    vithip_event_t ev_h2d[2], ev_done[2];
    int last_rows;

    /* stage profiling */
    vithip_event_t ev[2 * MAX_EVENTS];
    int ev_stage[MAX_EVENTS];
    int ev_used;
    int ev_ready;
    long pending_images;         /* images whose brackets are still in the pool */
    vit_stage_times times;
};

/* ------------------------------------------------------------------------------------------ */

vit_config vit_config_b16(void) {
    vit_config c = {224, 16, 3, 1000, 768, 12, 12, 3072};
    return c;
}

int vit_config_tokens(const vit_config *cfg) {
    int g = cfg->img_size / cfg->patch_size;
    return g * g + 1;
}

size_t vit_config_weight_size(const vit_config *cfg, int index) {
    const size_t D = (size_t)cfg->embed_dim, H = (size_t)cfg->hidden_dim;
    const size_t T = (size_t)vit_config_tokens(cfg);
    const size_t PK = (size_t)cfg->in_chans * cfg->patch_size * cfg->patch_size;
    const int base = 4 + VIT_WEIGHTS_PER_LAYER * cfg->depth;
    if (index < 0 || index >= base + 4) return 0;
    if (index < 4) {
        const size_t s[4] = {D, D * PK, D, T * D};
        return s[index];
    }
    if (index >= base) {
        const size_t s[4] = {D, D, (size_t)cfg->num_classes * D, (size_t)cfg->num_classes};
        return s[index - base];
    }
    {
        /* ln1 w,b | in_proj w,b | out_proj w,b | ln2 w,b | fc1 w,b | fc2 w,b  (ViT_seq.c:366-426) */
        const size_t s[12] = {D, D, 3 * D * D, 3 * D, D * D, D, D, D, H * D, H, D * H, D};
        return s[(index - 4) % VIT_WEIGHTS_PER_LAYER];
    }
}

unsigned long long vit_config_macs_per_image(const vit_config *cfg) {
    const unsigned long long T = (unsigned long long)vit_config_tokens(cfg), D = cfg->embed_dim,
                             H = cfg->hidden_dim, hd = D / cfg->num_heads,
                             PK = (unsigned long long)cfg->in_chans * cfg->patch_size * cfg->patch_size;
    const unsigned long long layer = T * D * 3 * D + 2 * cfg->num_heads * T * T * hd + T * D * D + 2 * T * D * H;
    return (T - 1) * PK * D + cfg->depth * layer + D * cfg->num_classes;
}

unsigned long long vit_config_macs_per_image_pruned(const vit_config *cfg) {
    const unsigned long long T = (unsigned long long)vit_config_tokens(cfg), D = cfg->embed_dim, H = cfg->hidden_dim,
                             hd = D / cfg->num_heads;
    /* last layer: Q projection, both attention products, out_proj, fc1, fc2 for one row instead of T */
    const unsigned long long saved = (T - 1) * (D * D + 2 * cfg->num_heads * T * hd + D * D + 2 * D * H);
    return vit_config_macs_per_image(cfg) - saved;
}

void vit_engine_default_options(vit_engine_options *opt) {
    opt->device = 0;
    opt->max_batch = 256;
    opt->profile = 0;
    opt->lanes = 1;
    opt->dtype = VIT_DTYPE_F32;
    opt->prune_last_layer = 0;
    opt->use_graph = 0;
    opt->gemm_tile = 0;
    opt->ln_fold = 0;
    opt->gemm_handover_test = 0;
    opt->host_first_piece = 0;
}

static int fail(vit_engine *e, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e->err, sizeof(e->err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(e, call)                                                                     \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != 0)                                                                        \
            return fail((e), VIT_ERR_HIP, "[%s:%d] HIP error %d (%s) in %s", __FILE__, __LINE__, \
                        rc_, vithip_error_string(rc_), #call);                               \
    } while (0)

const char *vit_engine_last_error(const vit_engine *e) { return e ? e->err : "null engine"; }
const vit_config *vit_engine_config(const vit_engine *e) { return &e->cfg; }

#define WEIGHT_TAIL_PAD (1u << 20) /* GEMM loaders read (never use) up to a tile of rows past the last tensor */

static int check_config(vit_engine *e) {
    const vit_config *c = &e->cfg;
    if (c->img_size <= 0 || c->patch_size <= 0 || c->in_chans <= 0 || c->num_classes <= 0 ||
        c->embed_dim <= 0 || c->depth <= 0 || c->num_heads <= 0 || c->hidden_dim <= 0)
        return fail(e, VIT_ERR_ARG, "vit_config: all dimensions must be positive");
    if (c->img_size % c->patch_size) return fail(e, VIT_ERR_ARG, "img_size must be a multiple of patch_size");
    if (c->embed_dim % c->num_heads || c->embed_dim / c->num_heads != 64)
        return fail(e, VIT_ERR_ARG, "HIP attention kernel needs head_dim == 64 (got %d/%d)", c->embed_dim, c->num_heads);
    if (c->embed_dim % 32 || c->hidden_dim % 32)
        return fail(e, VIT_ERR_ARG, "embed_dim and hidden_dim must be multiples of 32");
    if (c->patch_size % 4 || c->img_size % 4 || (c->in_chans * c->patch_size * c->patch_size) % 32)
        return fail(e, VIT_ERR_ARG, "patch geometry unsupported (patch%%4, img%%4, C*P*P%%32 must be 0)");
    if (c->embed_dim > 2048) return fail(e, VIT_ERR_ARG, "embed_dim > 2048 unsupported by the LayerNorm kernel");
    return VIT_OK;
}

/* One hand-over workspace per lane in use (fp32 engines; 32 MB of uncached device memory each on a 256-CU device). */
static int ensure_gemm_workspaces(vit_engine *e) {
    if (e->opt.dtype != VIT_DTYPE_F32) return VIT_OK;
    for (int j = 0; j < e->opt.lanes && j < VIT_MAX_LANES; ++j)
        if (!e->gemm_ws[j]) HIP_TRY(e, vithip_gemm_f32_workspace_create(&e->gemm_ws[j]));
    return VIT_OK;
}

int vit_engine_create(vit_engine **out, const vit_config *cfg, const vit_engine_options *opt) {
    static vit_engine scratch;  /* error text holder when allocation itself fails */
    if (!out) return VIT_ERR_ARG;
    *out = NULL;
    vit_engine *e = (vit_engine *)calloc(1, sizeof(*e));
    if (!e) return fail(&scratch, VIT_ERR_NOMEM, "out of host memory");
    *out = e; /* returned even on failure so the caller can read the message, then destroy */
    e->cfg = cfg ? *cfg : vit_config_b16();
    if (opt) e->opt = *opt; else vit_engine_default_options(&e->opt);
    if (e->opt.max_batch <= 0) e->opt.max_batch = 256;
    if (e->opt.lanes < 1) e->opt.lanes = 1;
    int rc = check_config(e);
    if (rc) return rc;
    e->tokens = vit_config_tokens(&e->cfg);
    e->n_weights = VIT_WEIGHT_COUNT(e->cfg.depth);
    {
        /* The fp32 GEMMs (and the patch gather) address an A operand through a buffer descriptor with 32-bit byte
         * offsets (csrc/vit_gemm.hip): one launch may span < 2 GiB of images, LN output, attention output or MLP hidden
         * rows.  That bounds the images of one LANE; larger chunks are cut down in forward_device/forward_host. */
        const size_t T_ = (size_t)e->tokens, per[3] = {T_ * (size_t)e->cfg.hidden_dim, T_ * (size_t)e->cfg.embed_dim,
                                                         (size_t)e->cfg.in_chans * e->cfg.img_size * e->cfg.img_size};
        size_t worst = per[0] > per[1] ? per[0] : per[1];
        if (per[2] > worst) worst = per[2];
        const size_t cap = ((size_t)0x7fffffff - 4096) / (worst * sizeof(float));
        if (cap < 1) return fail(e, VIT_ERR_ARG, "model too large: one image needs %zu bytes of fp32 rows, the fp32 kernels "
                                                 "address < 2 GiB per launch", worst * sizeof(float));
        e->lane_cap = cap > (size_t)INT_MAX ? INT_MAX : (int)cap;
    }

    int ndev = 0;
    HIP_TRY(e, vithip_device_count(&ndev));
    if (e->opt.device < 0 || e->opt.device >= ndev)
        return fail(e, VIT_ERR_ARG, "device %d not available (%d HIP devices)", e->opt.device, ndev);
    HIP_TRY(e, vithip_set_device(e->opt.device));
    {
        vithip_device_info di;
        HIP_TRY(e, vithip_get_device_info(e->opt.device, &di));
        e->n_cus = di.compute_units;
    }
    HIP_TRY(e, vithip_stream_create(&e->stream));
    for (int j = 0; j < VIT_MAX_LANES - 1; ++j) {
        HIP_TRY(e, vithip_stream_create(&e->aux_stream[j]));
        HIP_TRY(e, vithip_event_create(&e->ev_join[j]));
    }
    HIP_TRY(e, vithip_event_create(&e->ev_fork));

    const size_t B = (size_t)e->opt.max_batch, T = (size_t)e->tokens, D = (size_t)e->cfg.embed_dim,
                 H = (size_t)e->cfg.hidden_dim, NC = (size_t)e->cfg.num_classes;
    const size_t img = (size_t)e->cfg.in_chans * e->cfg.img_size * e->cfg.img_size;
    HIP_TRY(e, vithip_malloc((void **)&e->x, B * T * D * sizeof(float)));
    HIP_TRY(e, vithip_malloc((void **)&e->y, B * T * D * sizeof(float)));
    HIP_TRY(e, vithip_malloc((void **)&e->qkv, B * T * 3 * D * sizeof(float)));
    HIP_TRY(e, vithip_malloc((void **)&e->hbuf, B * T * H * sizeof(float)));
    HIP_TRY(e, vithip_malloc((void **)&e->z, B * D * sizeof(float)));
    HIP_TRY(e, vithip_malloc((void **)&e->logits, B * NC * sizeof(float)));
    if (e->opt.lanes > VIT_MAX_LANES) e->opt.lanes = VIT_MAX_LANES;
    {
        int rc_ws = ensure_gemm_workspaces(e);
        if (rc_ws) return rc_ws;
    }
    if (e->opt.dtype == VIT_DTYPE_BF16 && e->opt.ln_fold >= 0) {
        /* the fold lives in the ping-pong GEMM (two K steps at least); its scratch (bf16 copy of x, row sums) uses the
         * idle halves of the y and qkv allocations, which bf16 activations only half fill */
        const int ok = e->cfg.embed_dim >= 128 && e->cfg.hidden_dim >= 128 && e->cfg.embed_dim % 64 == 0 && e->cfg.hidden_dim % 64 == 0;
        if (!ok && e->opt.ln_fold > 0) return fail(e, VIT_ERR_ARG, "ln_fold needs embed_dim and hidden_dim >= 128 and multiples of 64");
        e->fold = ok;
        if (e->fold) {
            const size_t L = (size_t)e->cfg.depth;
            HIP_TRY(e, vithip_malloc((void **)&e->wfold16, L * (3 * D * D + H * D) * sizeof(unsigned short) + WEIGHT_TAIL_PAD));
            HIP_TRY(e, vithip_memset((char *)e->wfold16 + L * (3 * D * D + H * D) * sizeof(unsigned short), 0, WEIGHT_TAIL_PAD, e->stream));
            HIP_TRY(e, vithip_malloc((void **)&e->wfoldf, L * (6 * D + 2 * H) * sizeof(float)));
        }
    }
    if (e->opt.dtype == VIT_DTYPE_F32 && e->opt.ln_fold >= 0) {
        /* fp32 fold: the consumer epilogue exists in every fp32 GEMM kernel; the row statistics kernel wants whole 64-column strips */
        const int ok = e->cfg.embed_dim % 64 == 0 && e->cfg.embed_dim <= 2048;
        if (!ok && e->opt.ln_fold > 0) return fail(e, VIT_ERR_ARG, "ln_fold (fp32) needs embed_dim to be a multiple of 64, at most 2048");
        e->fold = ok;
        if (e->fold) {
            const size_t L = (size_t)e->cfg.depth;
            HIP_TRY(e, vithip_malloc((void **)&e->wfold32, L * (3 * D * D + H * D) * sizeof(float) + WEIGHT_TAIL_PAD));
            HIP_TRY(e, vithip_memset((char *)e->wfold32 + L * (3 * D * D + H * D) * sizeof(float), 0, WEIGHT_TAIL_PAD, e->stream));
            HIP_TRY(e, vithip_malloc((void **)&e->wfoldf, L * (6 * D + 2 * H) * sizeof(float)));
            HIP_TRY(e, vithip_malloc((void **)&e->ln_rows32, (B * T + B) * 2 * sizeof(float)));
            HIP_TRY(e, vithip_malloc((void **)&e->ln_part32, (D / 64) * B * T * 2 * sizeof(float)));
        }
    }
    HIP_TRY(e, vithip_stream_create(&e->copy_stream));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(e, vithip_malloc((void **)&e->in_stage[b], B * img * sizeof(float)));
        HIP_TRY(e, vithip_malloc((void **)&e->out_stage[b], B * NC * sizeof(float)));
        HIP_TRY(e, vithip_host_alloc((void **)&e->pin_in[b], B * img * sizeof(float)));
        HIP_TRY(e, vithip_host_alloc((void **)&e->pin_out[b], B * NC * sizeof(float)));
        HIP_TRY(e, vithip_event_create(&e->ev_h2d[b]));
        HIP_TRY(e, vithip_event_create(&e->ev_done[b]));
    }

    e->w = (float **)calloc((size_t)e->n_weights, sizeof(float *));
    e->w16 = (unsigned short **)calloc((size_t)e->n_weights, sizeof(unsigned short *));
    if (!e->w || !e->w16) return fail(e, VIT_ERR_NOMEM, "out of host memory");
    if (e->opt.dtype != VIT_DTYPE_F32 && e->opt.dtype != VIT_DTYPE_BF16)
        return fail(e, VIT_ERR_ARG, "dtype must be VIT_DTYPE_F32 or VIT_DTYPE_BF16");
    if (e->opt.dtype == VIT_DTYPE_BF16 && (e->cfg.embed_dim % 64 || e->cfg.hidden_dim % 64))
        return fail(e, VIT_ERR_ARG, "bf16 path needs embed_dim and hidden_dim to be multiples of 64");
    if (e->opt.profile) return vit_engine_set_profile(e, 1);
    return VIT_OK;
}

void vit_engine_destroy(vit_engine *e) {
    if (!e) return;
    vithip_set_device(e->opt.device);
    if (e->stream) vithip_stream_sync(e->stream);
    if (e->ev_ready)
        for (int i = 0; i < 2 * MAX_EVENTS; ++i) vithip_event_destroy(e->ev[i]);
    if (e->graph) vithip_graph_destroy(e->graph);
    vithip_free(e->x); vithip_free(e->y); vithip_free(e->qkv); vithip_free(e->hbuf);
    vithip_free(e->z); vithip_free(e->logits);
    for (int j = 0; j < VIT_MAX_LANES; ++j) vithip_gemm_f32_workspace_destroy(e->gemm_ws[j]);
    if (e->copy_stream) { vithip_stream_sync(e->copy_stream); vithip_stream_destroy(e->copy_stream); }
    for (int b = 0; b < 2; ++b) {
        vithip_free(e->in_stage[b]); vithip_free(e->out_stage[b]);
        if (e->pin_in[b]) vithip_host_free(e->pin_in[b]);
        if (e->pin_out[b]) vithip_host_free(e->pin_out[b]);
        if (e->ev_h2d[b]) vithip_event_destroy(e->ev_h2d[b]);
        if (e->ev_done[b]) vithip_event_destroy(e->ev_done[b]);
    }
    vithip_free(e->wblob);
    vithip_free(e->wfold16);
    vithip_free(e->wfold32);
    vithip_free(e->ln_rows32);
    vithip_free(e->ln_part32);
    vithip_free(e->wfoldf);
    free(e->w16);
    for (int j = 0; j < VIT_MAX_LANES - 1; ++j) {
        if (e->aux_stream[j]) { vithip_stream_sync(e->aux_stream[j]); vithip_stream_destroy(e->aux_stream[j]); }
        if (e->ev_join[j]) vithip_event_destroy(e->ev_join[j]);
    }
    if (e->ev_fork) vithip_event_destroy(e->ev_fork);
    if (e->stream) vithip_stream_destroy(e->stream);
    free(e->w);
    free(e);
}

int vit_engine_set_lanes(vit_engine *e, int lanes) {
    if (!e) return VIT_ERR_ARG;
    if (lanes < 1 || lanes > VIT_MAX_LANES) return fail(e, VIT_ERR_ARG, "lanes must be 1..%d", VIT_MAX_LANES);
    e->opt.lanes = lanes;
    HIP_TRY(e, vithip_set_device(e->opt.device));
    return ensure_gemm_workspaces(e);
}

int vit_engine_set_profile(vit_engine *e, int on) {
    if (on && !e->ev_ready) {
        for (int i = 0; i < 2 * MAX_EVENTS; ++i) HIP_TRY(e, vithip_event_create(&e->ev[i]));
        e->ev_ready = 1;
    }
    e->opt.profile = on ? 1 : 0;
    return VIT_OK;
}

/* ------------------------------------------------------------------------------------------ */

/* A captured forward holds the old weight addresses in its kernel arguments: drop it with them. */
static void drop_graph(vit_engine *e) {
    if (e->graph) { vithip_graph_destroy(e->graph); e->graph = NULL; }
    e->g_n = 0; e->g_images = NULL; e->g_probs = NULL; e->g_label = NULL; e->g_prob = NULL;
}

/* (Re)allocate the device blob for this model and point w[] / w16[] into it (no data yet). */
static int alloc_weight_blob(vit_engine *e, const size_t *off, size_t f32_floats, size_t gemm_floats) {
    const int bf16 = e->opt.dtype == VIT_DTYPE_BF16;
    const size_t bytes = f32_floats * sizeof(float) + (bf16 ? gemm_floats * sizeof(unsigned short) : 0);
    HIP_TRY(e, vithip_set_device(e->opt.device));
    if (e->stream) HIP_TRY(e, vithip_stream_sync(e->stream));
    drop_graph(e);
    e->weights_loaded = 0;
    if (e->wblob && e->wblob_bytes != bytes) { vithip_free(e->wblob); e->wblob = NULL; }
    if (!e->wblob) {
        HIP_TRY(e, vithip_malloc((void **)&e->wblob, bytes + WEIGHT_TAIL_PAD));
        HIP_TRY(e, vithip_memset((char *)e->wblob + bytes, 0, WEIGHT_TAIL_PAD, e->stream));
    }
    e->wblob_bytes = bytes;
    e->wblob16 = bf16 ? (unsigned short *)(e->wblob + f32_floats) : NULL;
    for (int i = 0; i < e->n_weights; ++i) {
        e->w[i] = e->wblob + off[i];
        e->w16[i] = (bf16 && off[i] < gemm_floats) ? e->wblob16 + off[i] : NULL;
    }
    return VIT_OK;
}

/* LayerNorm fold: Wf = bf16(gamma * W), column sums and beta-folded biases of every layer's in_proj (LN1) and fc1 (LN2), from
 * the resident fp32 tensors.  Runs after every upload / replication; 2 launches per layer. */
static int fold_ln_weights(vit_engine *e) {
    if (!e->fold) return VIT_OK;
    const size_t D = (size_t)e->cfg.embed_dim, H = (size_t)e->cfg.hidden_dim;
    for (int l = 0; l < e->cfg.depth; ++l) {
        float **lw = e->w + 4 + VIT_WEIGHTS_PER_LAYER * l;
        float *ff = e->wfoldf + (size_t)l * (6 * D + 2 * H);
        if (e->opt.dtype == VIT_DTYPE_F32) { /* fp32 products gamma * W; sums in double, rounded once */
            float *f32 = e->wfold32 + (size_t)l * (3 * D * D + H * D);
            /* centred weights (vit_hip_kernels.h): the GEMMs deliver x . (gamma W)^T - mean * colsum themselves, their epilogues only scale;
             * the colsum slots of wfoldf receive what the weights' rounding left of the column sums and are not read again */
            HIP_TRY(e, vithip_ln_fold_weights_f32_centered(e->stream, lw[2], lw[3], lw[0], lw[1], f32, ff, ff + 3 * D, (int)(3 * D), (int)D));
            HIP_TRY(e, vithip_ln_fold_weights_f32_centered(e->stream, lw[8], lw[9], lw[6], lw[7], f32 + 3 * D * D, ff + 6 * D, ff + 6 * D + H, (int)H, (int)D));
            continue;
        }
        unsigned short *f16 = e->wfold16 + (size_t)l * (3 * D * D + H * D);
        /* in_proj: the Q rows also carry the factor of the scores' exponent (the attention kernels are told: _qscaled) */
        HIP_TRY(e, vithip_ln_fold_weights_scaled(e->stream, lw[2], lw[3], lw[0], lw[1], f16, ff, ff + 3 * D, (int)(3 * D), (int)D, (int)D, VITHIP_QSCALE));
        HIP_TRY(e, vithip_ln_fold_weights(e->stream, lw[8], lw[9], lw[6], lw[7], f16 + 3 * D * D, ff + 6 * D, ff + 6 * D + H, (int)H, (int)D));
    }
    return VIT_OK;
}

int vit_engine_load_weight_image(vit_engine *e, const vit_weight_image *img) {
    if (!e || !img || !img->f32) return e ? fail(e, VIT_ERR_ARG, "null weight image") : VIT_ERR_ARG;
    if (memcmp(&img->cfg, &e->cfg, sizeof(vit_config)) != 0 || img->count != e->n_weights)
        return fail(e, VIT_ERR_WEIGHTS, "weight image was built for another model configuration");
    int rc = alloc_weight_blob(e, img->off, img->f32_floats, img->gemm_floats);
    if (rc) return rc;
    const int bf16 = e->opt.dtype == VIT_DTYPE_BF16;
    /* ONE host-to-device copy: the image is the device layout ([fp32 | bf16] adjacent on both sides) */
    const size_t up = img->f32_floats * sizeof(float) + ((bf16 && img->bf16_elems) ? img->bf16_elems * sizeof(unsigned short) : 0);
    HIP_TRY(e, vithip_memcpy_h2d(e->wblob, img->f32, up, e->stream));
    if (bf16 && !img->bf16_elems) /* image without a bf16 section: convert the GEMM-operand region on the device, one launch */
        HIP_TRY(e, vithip_f32_to_bf16(e->stream, e->wblob, e->wblob16, img->gemm_floats));
    if ((rc = fold_ln_weights(e))) return rc;
    HIP_TRY(e, vithip_stream_sync(e->stream));
    e->weights_loaded = 1;
    return VIT_OK;
}

int vit_engine_load_weights(vit_engine *e, const Network *weights, int count) {
    if (!e || !weights) return e ? fail(e, VIT_ERR_ARG, "null weights") : VIT_ERR_ARG;
    if (count != e->n_weights)
        return fail(e, VIT_ERR_WEIGHTS, "expected %d weight tensors for depth %d, got %d", e->n_weights, e->cfg.depth, count);
    for (int i = 0; i < count; ++i) {
        const size_t want = vit_config_weight_size(&e->cfg, i);
        if (!weights[i].data)
            return fail(e, VIT_ERR_WEIGHTS, "weight %d is missing (Network[%d].data == NULL; expected %zu floats)", i, i, want);
        if (weights[i].size != want)
            return fail(e, VIT_ERR_WEIGHTS, "weight %d has %zu floats, expected %zu", i, weights[i].size, want);
    }
    /* the separately malloc'd tensors of the reference's Network[] (Network.c:147-191) are packed into the device
     * layout on the host (8 threads), then uploaded with one copy; bf16 operands are converted on the device */
    vit_weight_image img;
    if (vit_weight_image_build(&img, &e->cfg, weights, count, 0) != 0) return fail(e, VIT_ERR_NOMEM, "out of host memory packing the weights");
    const int rc = vit_engine_load_weight_image(e, &img);
    vit_weight_image_free(&img);
    return rc;
}

int vit_engine_copy_weights(vit_engine *dst, vit_engine *src) {
    if (!dst || !src) return VIT_ERR_ARG;
    if (!src->weights_loaded) return fail(dst, VIT_ERR_STATE, "copy_weights: the source engine has no weights");
    if (memcmp(&dst->cfg, &src->cfg, sizeof(vit_config)) != 0 || dst->opt.dtype != src->opt.dtype)
        return fail(dst, VIT_ERR_ARG, "copy_weights: engines differ in model configuration or dtype");
    size_t *off = (size_t *)malloc(sizeof(size_t) * (size_t)dst->n_weights);
    if (!off) return fail(dst, VIT_ERR_NOMEM, "out of host memory");
    size_t gemm_floats = 0;
    const size_t f32_floats = vit_weight_layout(&dst->cfg, off, NULL, &gemm_floats);
    int rc = alloc_weight_blob(dst, off, f32_floats, gemm_floats);
    free(off);
    if (rc) return rc;
    /* device-to-device: on a multi-GPU node this crosses xGMI once per replica instead of PCIe (the in-process form of
     * "upload to GPU 0 + broadcast") */
    HIP_TRY(dst, vithip_memcpy_peer(dst->wblob, dst->opt.device, src->wblob, src->opt.device, dst->wblob_bytes, dst->stream));
    if ((rc = fold_ln_weights(dst))) return rc; /* recomputed from the replica's own fp32 tensors: 2 launches per layer */
    HIP_TRY(dst, vithip_stream_sync(dst->stream));
    dst->weights_loaded = 1;
    return VIT_OK;
}

int vit_engine_read_weight_image(vit_engine *e, vit_weight_image *img) {
    if (!e || !img) return VIT_ERR_ARG;
    if (!e->weights_loaded) return fail(e, VIT_ERR_STATE, "read_weight_image: no weights loaded");
    /* an empty image of the right shape (built from a zero-filled model would be wasteful: allocate through the loader) */
    Network *tmp = (Network *)calloc((size_t)e->n_weights, sizeof(Network));
    if (!tmp) return fail(e, VIT_ERR_NOMEM, "out of host memory");
    size_t *off = (size_t *)malloc(sizeof(size_t) * (size_t)e->n_weights), *size = (size_t *)malloc(sizeof(size_t) * (size_t)e->n_weights);
    size_t gemm_floats = 0;
    const size_t f32_floats = (off && size) ? vit_weight_layout(&e->cfg, off, size, &gemm_floats) : 0;
    float *host = f32_floats ? (float *)malloc(e->wblob_bytes) : NULL;
    int rc = host ? VIT_OK : fail(e, VIT_ERR_NOMEM, "out of host memory");
    if (!rc) {
        rc = vithip_set_device(e->opt.device) || vithip_memcpy_d2h(host, e->wblob, e->wblob_bytes, e->stream) ||
             vithip_stream_sync(e->stream);
        if (rc) rc = fail(e, VIT_ERR_HIP, "read_weight_image: device-to-host copy failed");
    }
    if (!rc) {
        for (int i = 0; i < e->n_weights; ++i) { tmp[i].data = host + off[i]; tmp[i].size = size[i]; }
        const int bf16 = e->opt.dtype == VIT_DTYPE_BF16;
        if (vit_weight_image_build(img, &e->cfg, tmp, e->n_weights, bf16) != 0) rc = fail(e, VIT_ERR_NOMEM, "out of host memory");
        /* the bf16 section is taken from the DEVICE (its own conversion), not re-derived on the host */
        if (!rc && bf16) memcpy(img->bf16, host + f32_floats, gemm_floats * sizeof(unsigned short));
    }
    free(host); free(off); free(size); free(tmp);
    return rc;
}

/* ---- stage launch helpers -------------------------------------------------------------------- */

static int stage_begin(vit_engine *e, vithip_stream_t s, int stage) {
    if (!e->opt.profile || e->ev_used >= MAX_EVENTS) return 0;
    e->ev_stage[e->ev_used] = stage;
    return vithip_event_record(e->ev[2 * e->ev_used], s);
}
static int stage_end(vit_engine *e, vithip_stream_t s) {
    if (!e->opt.profile || e->ev_used >= MAX_EVENTS) return 0;
    int rc = vithip_event_record(e->ev[2 * e->ev_used + 1], s);
    e->ev_used++;
    return rc;
}

static int gemm(vit_engine *e, vithip_stream_t s, int stage, const float *A, int lda, const float *W,
                const float *bias, const float *res, float *C, int ldc, int M, int N, int K, int epi) {
    vithip_gemm_args a;
    memset(&a, 0, sizeof(a));
    /* the lane's workspace: launches on one stream are ordered, which is what sharing it needs.  Lane 0 runs on the caller's
     * stream (whatever it is), lane j on aux_stream[j - 1]. */
    a.workspace = e->gemm_ws[0];
    for (int j = 0; j < VIT_MAX_LANES - 1; ++j)
        if (s == e->aux_stream[j]) a.workspace = e->gemm_ws[j + 1];
    a.handover_test = e->opt.gemm_handover_test;
    a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = bias; a.residual = res; a.ldr = ldc;
    a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.epilogue = epi;
    a.tile = e->opt.gemm_tile; a.group_m = 0;
    HIP_TRY(e, stage_begin(e, s, stage));
    HIP_TRY(e, vithip_gemm_f32(s, &a));
    HIP_TRY(e, stage_end(e, s));
    return VIT_OK;
}

/* fp32 consumer of the LayerNorm fold: A = the un-normalised rows x, Wf / bias_f / colsum the folded operands, rows = (rstd,
 * mean) per row of A (vithip_gemm_args.ln_rows) */
static int gemm_fold(vit_engine *e, vithip_stream_t s, int stage, const float *A, int lda, const float *Wf, const float *bias_f,
                     const float *colsum, const float *rows, float *C, int ldc, int M, int N, int K, int epi) {
    vithip_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.workspace = e->gemm_ws[0];
    for (int j = 0; j < VIT_MAX_LANES - 1; ++j)
        if (s == e->aux_stream[j]) a.workspace = e->gemm_ws[j + 1];
    a.handover_test = e->opt.gemm_handover_test;
    a.A = A; a.lda = lda; a.W = Wf; a.ldw = K; a.bias = bias_f; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.epilogue = epi;
    a.tile = e->opt.gemm_tile; a.group_m = 0;
    (void)colsum; /* the engine's fp32 weights are the CENTRED ones (vit_engine_load_weights): nothing to subtract in the epilogue */
    a.ln_rows = rows; a.ln_colsum = NULL;
    HIP_TRY(e, stage_begin(e, s, stage));
    HIP_TRY(e, vithip_gemm_f32(s, &a));
    HIP_TRY(e, stage_end(e, s));
    return VIT_OK;
}
/* fp32 residual GEMM in place (x += A . W^T + b) that also leaves the (rstd, mean) of the rows it stored in `rows` WHEN its
 * kernel can take the sums in its epilogue (the persistent walk: large batches); *took says whether it did -- otherwise the
 * caller runs the statistics pass (rowstats32), so that small batches keep their launch list and the stage profile its meaning */
static int gemm_res_stats(vit_engine *e, vithip_stream_t s, int stage, const float *A, int lda, const float *W, const float *bias,
                          float *x, int ldx, int M, int N, int K, float *rows, float *partials, int *took) {
    vithip_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.workspace = e->gemm_ws[0];
    for (int j = 0; j < VIT_MAX_LANES - 1; ++j)
        if (s == e->aux_stream[j]) a.workspace = e->gemm_ws[j + 1];
    a.handover_test = e->opt.gemm_handover_test;
    a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = bias; a.residual = x; a.ldr = ldx; a.C = x; a.ldc = ldx;
    a.M = M; a.N = N; a.K = K; a.epilogue = VITHIP_EPI_BIAS_RESIDUAL; a.tile = e->opt.gemm_tile;
    a.stats_out = rows; a.stats_partials = partials;
    *took = rows != NULL && vithip_gemm_f32_stats_in_epilogue(&a);
    if (!*took) a.stats_out = a.stats_partials = NULL;
    HIP_TRY(e, stage_begin(e, s, stage));
    HIP_TRY(e, vithip_gemm_f32(s, &a));
    HIP_TRY(e, stage_end(e, s));
    return VIT_OK;
}
static int rowstats32(vit_engine *e, vithip_stream_t s, const float *x, size_t ldx, float *rows, int n_rows, int D) {
    HIP_TRY(e, stage_begin(e, s, VIT_STAGE_LN));
    HIP_TRY(e, vithip_rowstats_f32(s, x, ldx, rows, n_rows, D));
    HIP_TRY(e, stage_end(e, s));
    return VIT_OK;
}

static int gemm16(vit_engine *e, vithip_stream_t s, int stage, const unsigned short *A, int lda, const unsigned short *W,
                  const float *bias, const float *res, void *C, int ldc, int M, int N, int K, int epi) {
    vithip_gemm_bf16_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = bias; a.residual = res; a.ldr = ldc;
    a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.epilogue = epi;
    HIP_TRY(e, stage_begin(e, s, stage));
    HIP_TRY(e, vithip_gemm_bf16(s, &a));
    HIP_TRY(e, stage_end(e, s));
    return VIT_OK;
}

/* The two GEMM roles of the LayerNorm fold (contiguous rows only).  Consumer: A = un-normalised bf16 rows, W / bias / colsum = the
 * folded operands, rows = (rstd, mean*rstd) per row.  Producer: the residual GEMM also stores bf16(x) and the row sums, which
 * one small launch turns into `rows` for the consumer behind it (accounted to the LayerNorm stage). */
static int gemm16_ln(vit_engine *e, vithip_stream_t s, int stage, const unsigned short *A, int lda, const unsigned short *Wf,
                     const float *bias_f, const float *colsum, const float *rows, unsigned short *C, int ldc, int M, int N, int K, int epi) {
    vithip_gemm_bf16_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = Wf; a.ldw = K; a.bias = bias_f; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.epilogue = epi;
    a.ln_rows = rows; a.ln_colsum = colsum;
    HIP_TRY(e, stage_begin(e, s, stage));
    HIP_TRY(e, vithip_gemm_bf16(s, &a));
    HIP_TRY(e, stage_end(e, s));
    return VIT_OK;
}
static int gemm16_res_stats(vit_engine *e, vithip_stream_t s, int stage, const unsigned short *A, int lda, const unsigned short *W,
                            const float *bias, float *x, unsigned short *x16, int ldx, float *partials, float *rows, int M, int N, int K) {
    vithip_gemm_bf16_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = bias; a.residual = x; a.ldr = ldx; a.C = x; a.ldc = ldx;
    a.M = M; a.N = N; a.K = K; a.epilogue = VITHIP_BF16_EPI_F32_RESIDUAL;
    a.x16 = x16; a.ldx16 = ldx; a.row_partials = partials;
    HIP_TRY(e, stage_begin(e, s, stage));
    HIP_TRY(e, vithip_gemm_bf16(s, &a));
    HIP_TRY(e, stage_end(e, s));
    HIP_TRY(e, stage_begin(e, s, VIT_STAGE_LN));
    HIP_TRY(e, vithip_rowstats_finalize(s, partials, vithip_ln_strips(N), M, N, rows));
    HIP_TRY(e, stage_end(e, s));
    return VIT_OK;
}

static int collect_profile(vit_engine *e) {
    if (e->ev_used == 0) return VIT_OK;
    HIP_TRY(e, vithip_device_sync()); /* brackets may sit on several lane streams */
    for (int i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(e, vithip_event_elapsed_ms(&ms, e->ev[2 * i], e->ev[2 * i + 1]));
        e->times.ms[e->ev_stage[i]] += ms;
        e->times.launches[e->ev_stage[i]]++;
    }
    e->times.images += e->pending_images;
    e->pending_images = 0;
    e->ev_used = 0;
    return VIT_OK;
}

/*
 * One chunk of nb <= max_batch images, everything device resident.
 *
 * With opt.lanes > 1 the chunk is cut into that many sub-batches that run the same stage sequence
 * on their own HIP streams (lane 0 on the caller's stream, which forks and joins the others with
 * events).  The sub-batches are independent -- images never interact -- so this adds no
 * synchronisation to the data path; what it buys is that the low-occupancy and HBM-bound kernels of
 * one lane run beside the other lane's GEMMs.  Launches are issued stage by stage across the lanes so
 * that the queues advance together.
 *
 * The layer loop picks ONE of six layer bodies (fp32 / bf16 with LayerNorm kernels / bf16 with the
 * LayerNorm fold, each in its full and its class-rows-only "pruned last layer" form); every body is a
 * sequence of stages, each stage issued for every lane.
 */
typedef struct {
    vithip_stream_t s;
    int off, n; /* first image of the lane inside the chunk, image count */
    int stats_ready; /* fp32 fold: the residual GEMM in front has left the (rstd, mean) of the lane's token rows (its epilogue took the sums) */
} vit_lane;

typedef struct {
    vit_engine *e;
    vit_lane lane[VIT_MAX_LANES];
    int L;                       /* lanes in use for this chunk */
    int T, D, H, NC;
    /* bf16 views of the activation buffers (bf16 activations half fill the fp32-sized allocations) */
    unsigned short *y16, *qkv16, *h16;
    /* LayerNorm fold scratch, in the idle halves: bf16 copy of x (y allocation); row sums, (rstd, mean*rstd) pairs per token
     * and -- pruned last layer -- per class row (qkv allocation).  A lane uses its own rows of each. */
    unsigned short *x16;
    float *ln_part, *ln_rows, *cls_rows;
    int strips;
} chunk_ctx;

#define LANES for (int j = 0; j < c->L; ++j)
#define LN_ (c->lane[j])
#define ROWS(j) ((size_t)c->lane[j].off * c->T)
#define PART(j) (c->ln_part + ROWS(j) * (size_t)c->strips * 2)
#define RUN(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

/* conv_proj + flatten_transpose + class_token + pos_emb (ViT_seq.c:25-101) */
static int stage_embed(chunk_ctx *c, const float *d_images) {
    vit_engine *e = c->e;
    const vit_config *cfg = &e->cfg;
    float **w = e->w;
    const size_t img = (size_t)cfg->in_chans * cfg->img_size * cfg->img_size;
    /* bf16 patch embedding needs K = chans*patch^2 to be a multiple of 64 (two K steps at least) and patch % 8 == 0;
     * its bf16 patch rows live in the (still unused) hidden-layer buffer */
    const int pk = cfg->in_chans * cfg->patch_size * cfg->patch_size;
    const int embed16 = e->opt.dtype == VIT_DTYPE_BF16 && pk % 64 == 0 && pk >= 128 && cfg->patch_size % 8 == 0 &&
                        (size_t)pk <= 2 * (size_t)c->H;
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_EMBED));
        if (embed16)
            HIP_TRY(e, vithip_patch_embed_bf16(LN_.s, d_images + LN_.off * img, e->w16[1], w[2], w[0], w[3],
                                               e->x + ROWS(j) * c->D, (unsigned short *)e->hbuf + (size_t)LN_.off * (c->T - 1) * pk,
                                               LN_.n, cfg->img_size, cfg->patch_size, cfg->in_chans, c->D));
        else
            HIP_TRY(e, vithip_patch_embed_f32(LN_.s, d_images + LN_.off * img, w[1], w[2], w[0], w[3],
                                              e->x + ROWS(j) * c->D, LN_.n, cfg->img_size, cfg->patch_size, cfg->in_chans, c->D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    return VIT_OK;
}

/* ---- fp32 layer (the reference's arithmetic, ViT_seq.c:276-300) ---- */
static int layer_f32(chunk_ctx *c, float **lw) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    LANES { /* LN1 (ViT_seq.c:281) */
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32(LN_.s, e->x + ROWS(j) * D, (size_t)D, e->y + ROWS(j) * D, (size_t)D, lw[0], lw[1], LN_.n * T, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES /* QKV in_proj (ViT_seq.c:134-147) */
        RUN(gemm(e, LN_.s, VIT_STAGE_QKV, e->y + ROWS(j) * D, D, lw[2], lw[3], NULL, e->qkv + ROWS(j) * 3 * D, 3 * D, LN_.n * T, 3 * D, D, VITHIP_EPI_BIAS));
    LANES { /* scores, softmax, P.V (ViT_seq.c:156-215) -> y */
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_f32(LN_.s, e->qkv + ROWS(j) * 3 * D, e->y + ROWS(j) * D, LN_.n, T, heads));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES /* out_proj + residual (ViT_seq.c:219-227,286-288): x = x + (y.Wo^T + bo) */
        RUN(gemm(e, LN_.s, VIT_STAGE_OUTPROJ, e->y + ROWS(j) * D, D, lw[4], lw[5], e->x + ROWS(j) * D, e->x + ROWS(j) * D, D, LN_.n * T, D, D, VITHIP_EPI_BIAS_RESIDUAL));
    LANES { /* LN2 (ViT_seq.c:291) */
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32(LN_.s, e->x + ROWS(j) * D, (size_t)D, e->y + ROWS(j) * D, (size_t)D, lw[6], lw[7], LN_.n * T, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES /* fc1 + GELU (ViT_seq.c:258-264) */
        RUN(gemm(e, LN_.s, VIT_STAGE_FC1, e->y + ROWS(j) * D, D, lw[8], lw[9], NULL, e->hbuf + ROWS(j) * H, H, LN_.n * T, H, D, VITHIP_EPI_BIAS_GELU));
    LANES /* fc2 + residual (ViT_seq.c:266,297-299): x = x + (h.W2^T + b2) */
        RUN(gemm(e, LN_.s, VIT_STAGE_FC2, e->hbuf + ROWS(j) * H, H, lw[10], lw[11], e->x + ROWS(j) * D, e->x + ROWS(j) * D, D, LN_.n * T, D, H, VITHIP_EPI_BIAS_RESIDUAL));
    return VIT_OK;
}

/* prune_last_layer (vit_engine_options): K and V of every token, everything else for the class rows only.  The class rows of a
 * [n*T][D] buffer are rows 0, T, 2T, ... = a matrix with leading dimension T*D, which every operator takes as it is. */
static int layer_f32_pruned(chunk_ctx *c, float **lw) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32(LN_.s, e->x + ROWS(j) * D, (size_t)D, e->y + ROWS(j) * D, (size_t)D, lw[0], lw[1], LN_.n * T, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES { /* K and V of every token (in_proj rows D..3D), Q of the class rows only */
        RUN(gemm(e, LN_.s, VIT_STAGE_QKV, e->y + ROWS(j) * D, D, lw[2] + (size_t)D * D, lw[3] + D, NULL, e->qkv + ROWS(j) * 3 * D + D, 3 * D, LN_.n * T, 2 * D, D, VITHIP_EPI_BIAS));
        RUN(gemm(e, LN_.s, VIT_STAGE_QKV, e->y + ROWS(j) * D, T * D, lw[2], lw[3], NULL, e->qkv + ROWS(j) * 3 * D, T * 3 * D, LN_.n, D, D, VITHIP_EPI_BIAS));
    }
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_f32_rows(LN_.s, e->qkv + ROWS(j) * 3 * D, e->y + ROWS(j) * D, LN_.n, T, heads, 1));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm(e, LN_.s, VIT_STAGE_OUTPROJ, e->y + ROWS(j) * D, T * D, lw[4], lw[5], e->x + ROWS(j) * D, e->x + ROWS(j) * D, T * D, LN_.n, D, D, VITHIP_EPI_BIAS_RESIDUAL));
    LANES { /* LN2 of the class rows -> compact [n][D] at the head of the lane's y region */
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32(LN_.s, e->x + ROWS(j) * D, (size_t)T * D, e->y + ROWS(j) * D, (size_t)D, lw[6], lw[7], LN_.n, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm(e, LN_.s, VIT_STAGE_FC1, e->y + ROWS(j) * D, D, lw[8], lw[9], NULL, e->hbuf + ROWS(j) * H, H, LN_.n, H, D, VITHIP_EPI_BIAS_GELU));
    LANES
        RUN(gemm(e, LN_.s, VIT_STAGE_FC2, e->hbuf + ROWS(j) * H, H, lw[10], lw[11], e->x + ROWS(j) * D, e->x + ROWS(j) * D, T * D, LN_.n, D, H, VITHIP_EPI_BIAS_RESIDUAL));
    return VIT_OK;
}

/* ---- fp32 layer with the LayerNorm fold (vit_hip_kernels.h, vithip_gemm_args.ln_rows): in_proj and fc1 read the raw rows x with
 * the gamma/beta-folded operands (f32 / ff) and the rows' (rstd, mean); each LayerNorm (ViT_seq.c:281, 291) is one pass
 * that reads x and writes 8 bytes per row.  R32(j) = the pairs of lane j's token rows, C32(j) = of its class rows. ---- */
#define R32(j) (e->ln_rows32 + ROWS(j) * 2)
#define C32(j) (e->ln_rows32 + ((size_t)e->opt.max_batch * c->T + (size_t)c->lane[j].off) * 2)
#define P32(j) (e->ln_part32 + ROWS(j) * (size_t)(c->D / 64) * 2)
static int layer_f32_folded(chunk_ctx *c, float **lw, const float *f32, const float *ff, int feeds_next) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    int ln2_ready[VIT_MAX_LANES];
    LANES /* LN1 (ViT_seq.c:281): its statistics, unless the fc2 in front has left them */
        if (!LN_.stats_ready) RUN(rowstats32(e, LN_.s, e->x + ROWS(j) * D, (size_t)D, R32(j), LN_.n * T, D));
    LANES /* QKV in_proj (ViT_seq.c:134-147) on LN1(x) */
        RUN(gemm_fold(e, LN_.s, VIT_STAGE_QKV, e->x + ROWS(j) * D, D, f32, ff + 3 * D, ff, R32(j), e->qkv + ROWS(j) * 3 * D, 3 * D, LN_.n * T, 3 * D, D, VITHIP_EPI_BIAS));
    LANES { /* scores, softmax, P.V (ViT_seq.c:156-215) -> y */
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_f32(LN_.s, e->qkv + ROWS(j) * 3 * D, e->y + ROWS(j) * D, LN_.n, T, heads));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES /* out_proj + residual (ViT_seq.c:219-227,286-288); the statistics of LN2 (ViT_seq.c:291) on the way when the kernel can */
        RUN(gemm_res_stats(e, LN_.s, VIT_STAGE_OUTPROJ, e->y + ROWS(j) * D, D, lw[4], lw[5], e->x + ROWS(j) * D, D, LN_.n * T, D, D, R32(j), P32(j), &ln2_ready[j]));
    LANES
        if (!ln2_ready[j]) RUN(rowstats32(e, LN_.s, e->x + ROWS(j) * D, (size_t)D, R32(j), LN_.n * T, D));
    LANES /* fc1 + GELU (ViT_seq.c:258-264) on LN2(x) */
        RUN(gemm_fold(e, LN_.s, VIT_STAGE_FC1, e->x + ROWS(j) * D, D, f32 + (size_t)3 * D * D, ff + 6 * D + H, ff + 6 * D, R32(j), e->hbuf + ROWS(j) * H, H, LN_.n * T, H, D, VITHIP_EPI_BIAS_GELU));
    LANES /* fc2 + residual (ViT_seq.c:266,297-299); the statistics of the next layer's LN1 on the way when there is one */
        RUN(gemm_res_stats(e, LN_.s, VIT_STAGE_FC2, e->hbuf + ROWS(j) * H, H, lw[10], lw[11], e->x + ROWS(j) * D, D, LN_.n * T, D, H,
                           feeds_next ? R32(j) : NULL, P32(j), &LN_.stats_ready));
    return VIT_OK;
}

/* the sequence of layer_f32_pruned in folded form: class rows = rows 0, T, 2T, ... */
static int layer_f32_folded_pruned(chunk_ctx *c, float **lw, const float *f32, const float *ff) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    LANES
        if (!LN_.stats_ready) RUN(rowstats32(e, LN_.s, e->x + ROWS(j) * D, (size_t)D, R32(j), LN_.n * T, D));
    LANES { /* K and V of every token (folded in_proj rows D..3D); Q of the class rows, whose pairs are made compact first */
        RUN(gemm_fold(e, LN_.s, VIT_STAGE_QKV, e->x + ROWS(j) * D, D, f32 + (size_t)D * D, ff + 3 * D + D, ff + D, R32(j), e->qkv + ROWS(j) * 3 * D + D, 3 * D, LN_.n * T, 2 * D, D, VITHIP_EPI_BIAS));
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_gather_rows_f32(LN_.s, R32(j), (size_t)2 * T, C32(j), 2, LN_.n, 2));
        HIP_TRY(e, stage_end(e, LN_.s));
        RUN(gemm_fold(e, LN_.s, VIT_STAGE_QKV, e->x + ROWS(j) * D, T * D, f32, ff + 3 * D, ff, C32(j), e->qkv + ROWS(j) * 3 * D, T * 3 * D, LN_.n, D, D, VITHIP_EPI_BIAS));
    }
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_f32_rows(LN_.s, e->qkv + ROWS(j) * 3 * D, e->y + ROWS(j) * D, LN_.n, T, heads, 1));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm(e, LN_.s, VIT_STAGE_OUTPROJ, e->y + ROWS(j) * D, T * D, lw[4], lw[5], e->x + ROWS(j) * D, e->x + ROWS(j) * D, T * D, LN_.n, D, D, VITHIP_EPI_BIAS_RESIDUAL));
    LANES RUN(rowstats32(e, LN_.s, e->x + ROWS(j) * D, (size_t)T * D, C32(j), LN_.n, D));   /* LN2 of the class rows */
    LANES
        RUN(gemm_fold(e, LN_.s, VIT_STAGE_FC1, e->x + ROWS(j) * D, T * D, f32 + (size_t)3 * D * D, ff + 6 * D + H, ff + 6 * D, C32(j), e->hbuf + ROWS(j) * H, H, LN_.n, H, D, VITHIP_EPI_BIAS_GELU));
    LANES
        RUN(gemm(e, LN_.s, VIT_STAGE_FC2, e->hbuf + ROWS(j) * H, H, lw[10], lw[11], e->x + ROWS(j) * D, e->x + ROWS(j) * D, T * D, LN_.n, D, H, VITHIP_EPI_BIAS_RESIDUAL));
    return VIT_OK;
}

/* ---- bf16 layer with LayerNorm kernels: LN output, qkv, attention output and the MLP hidden layer are bf16; the residual
 * stream x, LayerNorm statistics, softmax and every accumulation stay fp32 ---- */
static int layer_bf16(chunk_ctx *c, float **lw, unsigned short **lw16) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32_bf16out(LN_.s, e->x + ROWS(j) * D, (size_t)D, c->y16 + ROWS(j) * D, (size_t)D, lw[0], lw[1], LN_.n * T, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_QKV, c->y16 + ROWS(j) * D, D, lw16[2], lw[3], NULL, c->qkv16 + ROWS(j) * 3 * D, 3 * D, LN_.n * T, 3 * D, D, VITHIP_BF16_EPI_BF16));
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_bf16io(LN_.s, c->qkv16 + ROWS(j) * 3 * D, c->y16 + ROWS(j) * D, LN_.n, T, heads));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_OUTPROJ, c->y16 + ROWS(j) * D, D, lw16[4], lw[5], e->x + ROWS(j) * D, e->x + ROWS(j) * D, D, LN_.n * T, D, D, VITHIP_BF16_EPI_F32_RESIDUAL));
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32_bf16out(LN_.s, e->x + ROWS(j) * D, (size_t)D, c->y16 + ROWS(j) * D, (size_t)D, lw[6], lw[7], LN_.n * T, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_FC1, c->y16 + ROWS(j) * D, D, lw16[8], lw[9], NULL, c->h16 + ROWS(j) * H, H, LN_.n * T, H, D, VITHIP_BF16_EPI_BF16_GELU));
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_FC2, c->h16 + ROWS(j) * H, H, lw16[10], lw[11], e->x + ROWS(j) * D, e->x + ROWS(j) * D, D, LN_.n * T, D, H, VITHIP_BF16_EPI_F32_RESIDUAL));
    return VIT_OK;
}

static int layer_bf16_pruned(chunk_ctx *c, float **lw, unsigned short **lw16) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32_bf16out(LN_.s, e->x + ROWS(j) * D, (size_t)D, c->y16 + ROWS(j) * D, (size_t)D, lw[0], lw[1], LN_.n * T, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES { /* K and V of every token (in_proj rows D..3D), Q of the class rows only */
        RUN(gemm16(e, LN_.s, VIT_STAGE_QKV, c->y16 + ROWS(j) * D, D, lw16[2] + (size_t)D * D, lw[3] + D, NULL, c->qkv16 + ROWS(j) * 3 * D + D, 3 * D, LN_.n * T, 2 * D, D, VITHIP_BF16_EPI_BF16));
        RUN(gemm16(e, LN_.s, VIT_STAGE_QKV, c->y16 + ROWS(j) * D, T * D, lw16[2], lw[3], NULL, c->qkv16 + ROWS(j) * 3 * D, T * 3 * D, LN_.n, D, D, VITHIP_BF16_EPI_BF16));
    }
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_bf16io_rows(LN_.s, c->qkv16 + ROWS(j) * 3 * D, c->y16 + ROWS(j) * D, LN_.n, T, heads, 1));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_OUTPROJ, c->y16 + ROWS(j) * D, T * D, lw16[4], lw[5], e->x + ROWS(j) * D, e->x + ROWS(j) * D, T * D, LN_.n, D, D, VITHIP_BF16_EPI_F32_RESIDUAL));
    LANES { /* LN2 of the class rows -> compact [n][D] at the head of the lane's y region */
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32_bf16out(LN_.s, e->x + ROWS(j) * D, (size_t)T * D, c->y16 + ROWS(j) * D, (size_t)D, lw[6], lw[7], LN_.n, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_FC1, c->y16 + ROWS(j) * D, D, lw16[8], lw[9], NULL, c->h16 + ROWS(j) * H, H, LN_.n, H, D, VITHIP_BF16_EPI_BF16_GELU));
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_FC2, c->h16 + ROWS(j) * H, H, lw16[10], lw[11], e->x + ROWS(j) * D, e->x + ROWS(j) * D, T * D, LN_.n, D, H, VITHIP_BF16_EPI_F32_RESIDUAL));
    return VIT_OK;
}

/* ---- bf16 layer with the LayerNorm fold (vit_hip_kernels.h, "LayerNorm folding"): in_proj and fc1 read the raw bf16 rows x16
 * with the gamma/beta-folded operands (f16 / ff) and the per-row (rstd, mean*rstd) pairs; out_proj and fc2 store bf16(x) and
 * the row sums for the LayerNorm behind them.  `first`: layer 0, whose LN1 has no residual GEMM in front; `feeds_next`: fc2's
 * output is read by another folded layer. ---- */
static int layer_bf16_folded(chunk_ctx *c, float **lw, unsigned short **lw16, const unsigned short *f16, const float *ff, int first, int feeds_next) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    if (first)
        LANES { /* one pass for bf16(x) and the row pairs */
            HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
            HIP_TRY(e, vithip_rowstats_bf16(LN_.s, e->x + ROWS(j) * D, (size_t)D, c->x16 + ROWS(j) * D, (size_t)D, c->ln_rows + ROWS(j) * 2, LN_.n * T, D));
            HIP_TRY(e, stage_end(e, LN_.s));
        }
    LANES /* LN1 + in_proj */
        RUN(gemm16_ln(e, LN_.s, VIT_STAGE_QKV, c->x16 + ROWS(j) * D, D, f16, ff + 3 * D, ff, c->ln_rows + ROWS(j) * 2, c->qkv16 + ROWS(j) * 3 * D, 3 * D, LN_.n * T, 3 * D, D, VITHIP_BF16_EPI_BF16));
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_bf16io_qscaled(LN_.s, c->qkv16 + ROWS(j) * 3 * D, c->y16 + ROWS(j) * D, LN_.n, T, heads, T));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES /* out_proj + residual; bf16(x) and row sums for LN2 */
        RUN(gemm16_res_stats(e, LN_.s, VIT_STAGE_OUTPROJ, c->y16 + ROWS(j) * D, D, lw16[4], lw[5], e->x + ROWS(j) * D, c->x16 + ROWS(j) * D, D, PART(j), c->ln_rows + ROWS(j) * 2, LN_.n * T, D, D));
    LANES /* LN2 + fc1 + GELU */
        RUN(gemm16_ln(e, LN_.s, VIT_STAGE_FC1, c->x16 + ROWS(j) * D, D, f16 + 3 * (size_t)D * D, ff + 6 * D + H, ff + 6 * D, c->ln_rows + ROWS(j) * 2, c->h16 + ROWS(j) * H, H, LN_.n * T, H, D, VITHIP_BF16_EPI_BF16_GELU));
    LANES { /* fc2 + residual; bf16(x) and row sums for the next layer's LN1 */
        if (feeds_next)
            RUN(gemm16_res_stats(e, LN_.s, VIT_STAGE_FC2, c->h16 + ROWS(j) * H, H, lw16[10], lw[11], e->x + ROWS(j) * D, c->x16 + ROWS(j) * D, D, PART(j), c->ln_rows + ROWS(j) * 2, LN_.n * T, D, H));
        else
            RUN(gemm16(e, LN_.s, VIT_STAGE_FC2, c->h16 + ROWS(j) * H, H, lw16[10], lw[11], e->x + ROWS(j) * D, e->x + ROWS(j) * D, D, LN_.n * T, D, H, VITHIP_BF16_EPI_F32_RESIDUAL));
    }
    return VIT_OK;
}

/* the sequence of layer_bf16_pruned in folded form: class rows = rows 0, T, 2T, ... */
static int layer_bf16_folded_pruned(chunk_ctx *c, float **lw, unsigned short **lw16, const unsigned short *f16, const float *ff, int first) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, H = c->H, heads = e->cfg.num_heads;
    if (first)
        LANES {
            HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
            HIP_TRY(e, vithip_rowstats_bf16(LN_.s, e->x + ROWS(j) * D, (size_t)D, c->x16 + ROWS(j) * D, (size_t)D, c->ln_rows + ROWS(j) * 2, LN_.n * T, D));
            HIP_TRY(e, stage_end(e, LN_.s));
        }
    LANES { /* K and V of every token (folded in_proj rows D..3D); Q of the class rows, whose (rstd, mean*rstd) pairs are copied
             * out of the per-token array first */
        float *cls = c->cls_rows + (size_t)LN_.off * 2;
        RUN(gemm16_ln(e, LN_.s, VIT_STAGE_QKV, c->x16 + ROWS(j) * D, D, f16 + (size_t)D * D, ff + 3 * D + D, ff + D, c->ln_rows + ROWS(j) * 2, c->qkv16 + ROWS(j) * 3 * D + D, 3 * D, LN_.n * T, 2 * D, D, VITHIP_BF16_EPI_BF16));
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_gather_rows_f32(LN_.s, c->ln_rows + ROWS(j) * 2, (size_t)T * 2, cls, 2, LN_.n, 2));
        HIP_TRY(e, stage_end(e, LN_.s));
        RUN(gemm16_ln(e, LN_.s, VIT_STAGE_QKV, c->x16 + ROWS(j) * D, T * D, f16, ff + 3 * D, ff, cls, c->qkv16 + ROWS(j) * 3 * D, T * 3 * D, LN_.n, D, D, VITHIP_BF16_EPI_BF16));
    }
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_ATTN));
        HIP_TRY(e, vithip_attention_bf16io_qscaled(LN_.s, c->qkv16 + ROWS(j) * 3 * D, c->y16 + ROWS(j) * D, LN_.n, T, heads, 1));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm16_res_stats(e, LN_.s, VIT_STAGE_OUTPROJ, c->y16 + ROWS(j) * D, T * D, lw16[4], lw[5], e->x + ROWS(j) * D, c->x16 + ROWS(j) * D, T * D, PART(j), c->cls_rows + (size_t)LN_.off * 2, LN_.n, D, D));
    LANES
        RUN(gemm16_ln(e, LN_.s, VIT_STAGE_FC1, c->x16 + ROWS(j) * D, T * D, f16 + 3 * (size_t)D * D, ff + 6 * D + H, ff + 6 * D, c->cls_rows + (size_t)LN_.off * 2, c->h16 + ROWS(j) * H, H, LN_.n, H, D, VITHIP_BF16_EPI_BF16_GELU));
    LANES
        RUN(gemm16(e, LN_.s, VIT_STAGE_FC2, c->h16 + ROWS(j) * H, H, lw16[10], lw[11], e->x + ROWS(j) * D, e->x + ROWS(j) * D, T * D, LN_.n, D, H, VITHIP_BF16_EPI_F32_RESIDUAL));
    return VIT_OK;
}

/* final LayerNorm on the class-token rows only (ViT_seq.c:429-433 normalises all rows, uses row 0), classifier head
 * (ViT_seq.c:435), Softmax (ViT_seq.c:437) + top-1 (Main.c:62-70) */
static int stage_head(chunk_ctx *c, float *d_probs, int *d_label, float *d_prob) {
    vit_engine *e = c->e;
    const int T = c->T, D = c->D, NC = c->NC;
    float **fw = e->w + 4 + VIT_WEIGHTS_PER_LAYER * e->cfg.depth;
    LANES {
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_LN));
        HIP_TRY(e, vithip_layernorm_f32(LN_.s, e->x + ROWS(j) * D, (size_t)T * D, e->z + (size_t)LN_.off * D, (size_t)D, fw[0], fw[1], LN_.n, D));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    LANES
        RUN(gemm(e, LN_.s, VIT_STAGE_HEAD, e->z + (size_t)LN_.off * D, D, fw[2], fw[3], NULL, e->logits + (size_t)LN_.off * NC, NC, LN_.n, NC, D, VITHIP_EPI_BIAS));
    LANES {
        const size_t o = (size_t)LN_.off;
        HIP_TRY(e, stage_begin(e, LN_.s, VIT_STAGE_SOFTMAX));
        HIP_TRY(e, vithip_softmax_top1_f32(LN_.s, e->logits + o * NC, NC, d_probs + o * NC, NC, d_label ? d_label + o : NULL, d_prob ? d_prob + o : NULL, LN_.n, NC));
        HIP_TRY(e, stage_end(e, LN_.s));
    }
    return VIT_OK;
}

static int forward_chunk(vit_engine *e, vithip_stream_t s, const float *d_images, int nb, float *d_probs,
                         int *d_label, float *d_prob) {
    const vit_config *cfg = &e->cfg;
    chunk_ctx ctx, *c = &ctx;
    c->e = e;
    c->T = e->tokens; c->D = cfg->embed_dim; c->H = cfg->hidden_dim; c->NC = cfg->num_classes;
    c->L = e->opt.lanes > VIT_MAX_LANES ? VIT_MAX_LANES : e->opt.lanes;
    if (c->L < 1 || nb < 2 * c->L) c->L = 1;
    for (int j = 0; j < c->L; ++j) {
        c->lane[j].off = (int)((long)nb * j / c->L);
        c->lane[j].n = (int)((long)nb * (j + 1) / c->L) - c->lane[j].off;
        c->lane[j].s = j == 0 ? s : e->aux_stream[j - 1];
        c->lane[j].stats_ready = 0;
    }
    const size_t B = (size_t)e->opt.max_batch, T = (size_t)c->T, D = (size_t)c->D, H = (size_t)c->H;
    c->y16 = (unsigned short *)e->y; c->qkv16 = (unsigned short *)e->qkv; c->h16 = (unsigned short *)e->hbuf;
    c->strips = vithip_ln_strips((int)D);
    c->x16 = c->y16 + B * T * D;
    c->ln_part = (float *)(c->qkv16 + B * T * 3 * D);
    c->ln_rows = c->ln_part + (size_t)c->strips * B * T * 2;
    c->cls_rows = c->ln_rows + B * T * 2;

    if (c->L > 1) { /* fork: the other lanes start after everything already queued on s */
        HIP_TRY(e, vithip_event_record(e->ev_fork, s));
        for (int j = 1; j < c->L; ++j) HIP_TRY(e, vithip_stream_wait_event(c->lane[j].s, e->ev_fork));
    }
    RUN(stage_embed(c, d_images));
    const int bf16 = e->opt.dtype == VIT_DTYPE_BF16;
    const int prune = e->opt.prune_last_layer && c->T <= 224;
    for (int l = 0; l < cfg->depth; ++l) {
        float **lw = e->w + 4 + VIT_WEIGHTS_PER_LAYER * l;
        unsigned short **lw16 = e->w16 + 4 + VIT_WEIGHTS_PER_LAYER * l;
        const int last_pruned = prune && l == cfg->depth - 1;
        if (!bf16 && e->fold) {
            const float *f32 = e->wfold32 + (size_t)l * (3 * D * D + H * D);
            const float *ff = e->wfoldf + (size_t)l * (6 * D + 2 * H);
            RUN(last_pruned ? layer_f32_folded_pruned(c, lw, f32, ff) : layer_f32_folded(c, lw, f32, ff, l + 1 < cfg->depth));
        } else if (!bf16) {
            RUN(last_pruned ? layer_f32_pruned(c, lw) : layer_f32(c, lw));
        } else if (e->fold) {
            const unsigned short *f16 = e->wfold16 + (size_t)l * (3 * D * D + H * D);
            const float *ff = e->wfoldf + (size_t)l * (6 * D + 2 * H);
            RUN(last_pruned ? layer_bf16_folded_pruned(c, lw, lw16, f16, ff, l == 0)
                            : layer_bf16_folded(c, lw, lw16, f16, ff, l == 0, l + 1 < cfg->depth));
        } else {
            RUN(last_pruned ? layer_bf16_pruned(c, lw, lw16) : layer_bf16(c, lw, lw16));
        }
    }
    RUN(stage_head(c, d_probs, d_label, d_prob));
    for (int j = 1; j < c->L; ++j) { /* join */
        HIP_TRY(e, vithip_event_record(e->ev_join[j - 1], c->lane[j].s));
        HIP_TRY(e, vithip_stream_wait_event(s, e->ev_join[j - 1]));
    }
    e->last_rows = nb;
    return VIT_OK;
}
#undef LANES
#undef LN_
#undef ROWS
#undef PART
#undef RUN

/* Images per forward_chunk call: the workspace holds max_batch, and no lane may exceed lane_cap (see vit_engine_create). */
static int chunk_limit(const vit_engine *e) {
    int lanes = e->opt.lanes < 1 ? 1 : (e->opt.lanes > VIT_MAX_LANES ? VIT_MAX_LANES : e->opt.lanes);
    if (e->opt.dtype == VIT_DTYPE_BF16) return e->opt.max_batch; /* bf16 kernels rebase their descriptors per tile */
    const long cap = (long)e->lane_cap * lanes;
    return cap < e->opt.max_batch ? (int)cap : e->opt.max_batch;
}

int vit_engine_forward_device(vit_engine *e, const float *d_images, int n, float *d_probs,
                              int *d_top1_label, float *d_top1_prob, void *stream) {
    if (!e) return VIT_ERR_ARG;
    if (!d_images || !d_probs || n <= 0) return fail(e, VIT_ERR_ARG, "forward_device: bad arguments (n=%d)", n);
    if (!e->weights_loaded) return fail(e, VIT_ERR_STATE, "forward before vit_engine_load_weights()");
    vithip_stream_t s = stream ? (vithip_stream_t)stream : e->stream;
    const size_t img = (size_t)e->cfg.in_chans * e->cfg.img_size * e->cfg.img_size;
    const size_t NC = (size_t)e->cfg.num_classes;
    HIP_TRY(e, vithip_set_device(e->opt.device)); /* the current device is per host thread: several engines may share a process */
    const int chunk = chunk_limit(e);
    const int graphable = e->opt.use_graph && !e->opt.profile && e->opt.lanes == 1 && s != NULL;
    if (graphable && e->graph && e->g_n == n && e->g_images == d_images && e->g_probs == d_probs &&
        e->g_label == d_top1_label && e->g_prob == d_top1_prob) {
        HIP_TRY(e, vithip_graph_launch(e->graph, s));
        return VIT_OK;
    }
    if (graphable) {
        if (e->graph) { vithip_graph_destroy(e->graph); e->graph = NULL; }
        HIP_TRY(e, vithip_graph_begin(s));
    }
    for (int done = 0; done < n; done += chunk) {
        const int nb = n - done < chunk ? n - done : chunk;
        int rc = forward_chunk(e, s, d_images + (size_t)done * img, nb, d_probs + (size_t)done * NC,
                               d_top1_label ? d_top1_label + done : NULL,
                               d_top1_prob ? d_top1_prob + done : NULL);
        if (rc) {
            if (graphable) { /* never leave the caller's stream in capture mode: end the capture, discard what it recorded */
                vithip_graph_t g = NULL;
                if (vithip_graph_end(s, &g) == 0 && g) vithip_graph_destroy(g);
            }
            return rc;
        }
        if (e->opt.profile) e->pending_images += nb;
        /* read the brackets back lazily (it needs an event sync): only when the pool runs low */
        if (e->opt.profile && e->ev_used > MAX_EVENTS - 256 && (rc = collect_profile(e))) return rc;
    }
    if (graphable) { /* nothing ran yet: the launches above were recorded; instantiate and run them */
        HIP_TRY(e, vithip_graph_end(s, &e->graph));
        e->g_n = n; e->g_images = d_images; e->g_probs = d_probs; e->g_label = d_top1_label; e->g_prob = d_top1_prob;
        HIP_TRY(e, vithip_graph_launch(e->graph, s));
    }
    return VIT_OK;
}

int vit_engine_sync(vit_engine *e) {
    if (!e) return VIT_ERR_ARG;
    HIP_TRY(e, vithip_stream_sync(e->stream));
    return VIT_OK;
}

/* Host-pointer surface helpers.  The reference's ImageData keeps every image in its own malloc (Network.c:66-93), so a
 * piece has to be gathered into pinned memory before it can be uploaded: one memcpy thread manages ~10 GB/s, which made
 * the gather of the FIRST piece (nothing to overlap it with) 15 ms of a 512-image call.  The gather runs on a few OpenMP
 * threads and a piece goes up in sub-pieces of 64 images, so the H2D copy of one sub-piece overlaps the gather of the
 * next. */
#define GATHER_THREADS_MAX 16
#define SUB_PIECE 64
#define SUB_PIECE_FIRST 16 /* the call's first piece goes up in sub-pieces of 16: its upload starts after 10 MB of gathering */
static int gather_threads(void) {
    int n = omp_get_num_procs();
    return n < 1 ? 1 : (n > GATHER_THREADS_MAX ? GATHER_THREADS_MAX : n);
}
static void gather_images(float *dst, const float *const *images, int first, int count, size_t img) {
    const int nt = gather_threads();
#pragma omp parallel for num_threads(nt) schedule(static) if (count >= 4)
    for (int i = 0; i < count; ++i) memcpy(dst + (size_t)i * img, images[first + i], img * sizeof(float));
}
static int stage_piece(vit_engine *e, int slot, const float *const *images, int first, int count, size_t img, int sub) {
    for (int s0 = 0; s0 < count; s0 += sub) {
        const int c = count - s0 < sub ? count - s0 : sub;
        gather_images(e->pin_in[slot] + (size_t)s0 * img, images, first + s0, c, img);
        HIP_TRY(e, vithip_memcpy_h2d(e->in_stage[slot] + (size_t)s0 * img, e->pin_in[slot] + (size_t)s0 * img,
                                     (size_t)c * img * sizeof(float), e->copy_stream));
    }
    HIP_TRY(e, vithip_event_record(e->ev_h2d[slot], e->copy_stream));
    return VIT_OK;
}

int vit_engine_forward_host(vit_engine *e, const float *const *images, int n, float *const *probs) {
    if (!e) return VIT_ERR_ARG;
    if (!images || !probs || n <= 0) return fail(e, VIT_ERR_ARG, "forward_host: bad arguments (n=%d)", n);
    if (!e->weights_loaded) return fail(e, VIT_ERR_STATE, "forward before vit_engine_load_weights()");
    const size_t img = (size_t)e->cfg.in_chans * e->cfg.img_size * e->cfg.img_size;
    const size_t NC = (size_t)e->cfg.num_classes;
    for (int i = 0; i < n; ++i)
        if (!images[i] || !probs[i]) return fail(e, VIT_ERR_ARG, "forward_host: image or output row %d is NULL", i);
    HIP_TRY(e, vithip_set_device(e->opt.device));
    /*
     * Pieces of up to max_batch images flow through two staging slots: while the GPU computes piece i,
     * the host gathers the separately allocated images of piece i+1 into pinned memory and the copy
     * stream uploads them; the results of piece i-1 are scattered to the caller's rows meanwhile.
     * Nothing overlaps the gather + upload of the FIRST piece, so it is a small one, uploaded in sub-pieces of 16 images (the copy
     * of one overlaps the gather of the next), and the rest arrives behind its compute in pieces as large as the workspace allows
     * (large pieces keep the GEMMs' tile walks long).  Rows are bit-identical whatever the cut.
     */
    const int chunk = chunk_limit(e);
    /* Round 5: what the first piece has to do is cover, with its compute, the gather + upload of the piece behind it -- and no
     * more than that, because a small piece computes badly (ViT-B/16 fp32, device-resident: 8 images run at 56 % of the 256-image
     * rate per image, 40 at 86 %, 64 at 90 %, 192 at 99.5 %: tools/batch_time_sweep.py).  Measured at 256 images
     * (tools/host_path_sweep.py, ms per call, device-resident 64.9): first piece 8: 69.3 (the GPU waits for the second piece),
     * 12: 67.7, 16: 68.7, 40: 68.0, 64: 68.8 (the round-4 choice), 128: 71.2 -- on a box whose host gathers at ~40 GB/s.  On
     * one that gathers at ~16 GB/s the 12-image piece left the GPU idle for 4 ms (71.2 ms per call): the choice is 5 n / 32
     * (40 of 256, at most 64), whose compute covers the next piece's staging down to ~11 GB/s.  The bf16 engines compute an
     * image in a tenth of the time: they keep 64. */
    int first_n = n;
    if (e->opt.host_first_piece > 0) first_n = e->opt.host_first_piece;
    else if (e->opt.dtype == VIT_DTYPE_F32 && n >= 128) {
        first_n = (5 * n / 32 + 2) & ~3;
        if (first_n > 64) first_n = 64;
    } else if (n >= 128) first_n = 64;
    else if (n >= 64) first_n = (n + 1) / 2;
    if (first_n > chunk) first_n = chunk;
    if (first_n > n) first_n = n;
    const int np = 1 + (n - first_n + chunk - 1) / chunk;
#define PIECE_LO(i) ((i) == 0 ? 0 : ((i) >= np ? n : first_n + ((i) - 1) * chunk))
#define PIECE_N(i) ((PIECE_LO((i) + 1) < n ? PIECE_LO((i) + 1) : n) - PIECE_LO(i))
    /* stage piece 0 */
    {
        int rc0 = stage_piece(e, 0, images, 0, PIECE_N(0), img, np > 1 ? SUB_PIECE_FIRST : SUB_PIECE);
        if (rc0) return rc0;
    }
    for (int k = 0; k < np; ++k) {
        const int b = k & 1, nb = PIECE_N(k);
        HIP_TRY(e, vithip_stream_wait_event(e->stream, e->ev_h2d[b]));
        int rc = forward_chunk(e, e->stream, e->in_stage[b], nb, e->out_stage[b], NULL, NULL);
        if (rc) return rc;
        HIP_TRY(e, vithip_memcpy_d2h(e->pin_out[b], e->out_stage[b], (size_t)nb * NC * sizeof(float), e->stream));
        HIP_TRY(e, vithip_event_record(e->ev_done[b], e->stream));
        if (e->opt.profile) e->pending_images += nb;
        if (k >= 1) { /* piece k-1 (slot b^1) is finished by now or soon: hand its rows back */
            HIP_TRY(e, vithip_event_sync(e->ev_done[b ^ 1]));
            const int first = PIECE_LO(k - 1);
            for (int i = 0; i < PIECE_N(k - 1); ++i)
                memcpy(probs[first + i], e->pin_out[b ^ 1] + (size_t)i * NC, NC * sizeof(float));
        }
        if (k + 1 < np) { /* slot b^1 is free again (its H2D, compute and D2H are complete): refill it */
            rc = stage_piece(e, b ^ 1, images, PIECE_LO(k + 1), PIECE_N(k + 1), img, SUB_PIECE);
            if (rc) return rc;
        }
    }
    {
        const int b = (np - 1) & 1, first = PIECE_LO(np - 1);
        HIP_TRY(e, vithip_event_sync(e->ev_done[b]));
        for (int i = 0; i < PIECE_N(np - 1); ++i)
            memcpy(probs[first + i], e->pin_out[b] + (size_t)i * NC, NC * sizeof(float));
    }
#undef PIECE_N
#undef PIECE_LO
    if (e->opt.profile) {
        int rc = collect_profile(e);
        if (rc) return rc;
    }
    return VIT_OK;
}

int vit_engine_handover_stats(vit_engine *e, long *taken, long *recomputed) {
    if (!e) return VIT_ERR_ARG;
    HIP_TRY(e, vithip_set_device(e->opt.device));
    HIP_TRY(e, vithip_device_sync()); /* lanes and caller-provided streams */
    for (int j = 0; j < VIT_MAX_LANES; ++j) {
        int t = 0, r = 0;
        if (!e->gemm_ws[j]) continue;
        HIP_TRY(e, vithip_gemm_f32_workspace_stats(e->gemm_ws[j], &t, &r));
        e->handover_taken += t;
        e->handover_recomputed += r;
    }
    if (taken) *taken = e->handover_taken;
    if (recomputed) *recomputed = e->handover_recomputed;
    e->handover_taken = e->handover_recomputed = 0;
    return VIT_OK;
}

int vit_engine_read_logits(vit_engine *e, float *dst, int rows) {
    if (!e || !dst) return VIT_ERR_ARG;
    if (rows <= 0 || rows > e->last_rows) return fail(e, VIT_ERR_ARG, "read_logits: %d rows requested, last chunk had %d", rows, e->last_rows);
    HIP_TRY(e, vithip_device_sync()); /* the chunk may have run on a caller-provided stream */
    HIP_TRY(e, vithip_memcpy_d2h(dst, e->logits, (size_t)rows * e->cfg.num_classes * sizeof(float), e->stream));
    HIP_TRY(e, vithip_stream_sync(e->stream));
    return VIT_OK;
}

int vit_engine_get_stage_times(vit_engine *e, vit_stage_times *out) {
    if (!e || !out) return VIT_ERR_ARG;
    int rc = collect_profile(e);
    if (rc) return rc;
    *out = e->times;
    return VIT_OK;
}

void vit_engine_reset_stage_times(vit_engine *e) {
    if (!e) return;
    collect_profile(e); /* drain the pool so earlier brackets do not leak into the next window */
    memset(&e->times, 0, sizeof(e->times));
}
