/*
 * include/vit_engine.h -- the batched ViT forward engine (host side, plain C).
 *
 * This is the re-entrant object underneath the reference-shaped facade of ViT_hip.h.  It owns
 * what the reference's OpenCL path re-creates on every call (ViT_opencl.c:126-883: buffers,
 * weight uploads, per-op blocking reads): device-resident weights uploaded ONCE, a workspace
 * sized for a whole batch, one in-order HIP stream, and the layer loop that enqueues the
 * kernels of vit_hip_kernels.h.  The forward is ViT_seq.c:337-439 for a batch of images
 * (the reference loops `for i < image->n`, ViT_seq.c:354; here the batch is the GEMM M dimension).
 *
 * All functions return VIT_OK or an error code; vit_engine_last_error() gives the message.
 * Nothing here prints or exits -- that convention belongs to the facade (ViT_hip.h).
 */
#ifndef VIT_ENGINE_H
#define VIT_ENGINE_H

#include "vit_io.h"
#include "vit_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    VIT_OK = 0,
    VIT_ERR_ARG = 1,      /* bad argument / unsupported configuration */
    VIT_ERR_WEIGHTS = 2,  /* missing or mis-sized weight tensor */
    VIT_ERR_HIP = 3,      /* a HIP call failed (message carries the hipError string) */
    VIT_ERR_NOMEM = 4,
    VIT_ERR_STATE = 5     /* e.g. forward before weights were loaded */
};

typedef struct vit_engine vit_engine;

typedef struct {
    int device;      /* HIP device ordinal */
    int max_batch;   /* images per forward chunk (workspace is sized for it); default 256 */
    int profile;     /* 1: bracket every stage with events and accumulate vit_stage_times */
    int lanes;       /* sub-batches of a chunk run concurrently on separate streams (1..4); default 1 */
    int dtype;       /* VIT_DTYPE_F32 (default: the reference's arithmetic) or VIT_DTYPE_BF16 (bf16 MFMA GEMMs) */
    int prune_last_layer; /* 1: in the LAST encoder layer compute only what the class token needs (default 0).  Only row
                           * 0 of the encoder output feeds the classifier (ViT_seq.c:429-435), so after the last layer's
                           * K/V projections every token-wise operation runs on the n class rows instead of n*tokens:
                           * the probabilities are bit-identical, 7 % of the reference's arithmetic is not executed
                           * (vit_config_macs_per_image_pruned).  Off by default: bench.py's metric counts the
                           * reference's full work.  tokens <= 224. */
    int use_graph;   /* 1: vit_engine_forward_device() captures its launch sequence into a hipGraph the first time it sees
                      * an (n, pointers, stream-kind) combination and replays the graph afterwards (default 0).  For
                      * small batches the ~150 launches of a forward are a visible share of the latency.  Ignored
                      * while profiling, with lanes > 1 and on the NULL stream (not capturable). */
    int gemm_tile;   /* tuning: vithip_gemm_args.tile for every fp32 GEMM of this engine (0 = auto, the default) */
    int ln_fold;     /* fold the encoder LayerNorms into the GEMMs behind them (vit_hip_kernels.h, "LayerNorm folding"): in_proj and
                      * fc1 multiply the un-normalised rows with gamma-folded weights and rescale in their epilogues, a LayerNorm
                      * is then one pass that reads x and writes 8 bytes per row (fp32) or nothing at all (bf16: the residual
                      * GEMM in front produces the row sums).  0 = auto (fp32: on when embed_dim % 64 == 0 and <= 2048; bf16: on
                      * when embed_dim and hidden_dim >= 128), 1 = on (error when the shapes do not allow it), -1 = off: a
                      * LayerNorm kernel per LayerNorm, i.e. the reference's operation order (ViT_seq.c:103-147) -- both orders
                      * meet the 1e-4 bar against ViT_seq.c, the folded one is not the same bits as the unfolded one. */
    int gemm_handover_test; /* testing: vithip_gemm_args.handover_test for every fp32 GEMM (1 = helper pieces arrive too late and
                             * every owner computes its whole tile; results must not change) */
    int host_first_piece;   /* vit_engine_forward_host(): images in the FIRST piece of a call (nothing overlaps its gather and upload,
                             * so it is a small one; the rest follows behind its compute in pieces of max_batch).  0 = auto (the
                             * measured choice, host/vit_engine.c), otherwise clamped to [1, max_batch].  Rows are bit-identical
                             * whatever the cut. */
} vit_engine_options;

enum { VIT_DTYPE_F32 = 0, VIT_DTYPE_BF16 = 1 };

/* Per-stage device time of the profiled forwards (ms, summed) and launch counts. */
enum {
    VIT_STAGE_EMBED = 0, VIT_STAGE_LN, VIT_STAGE_QKV, VIT_STAGE_ATTN, VIT_STAGE_OUTPROJ,
    VIT_STAGE_FC1, VIT_STAGE_FC2, VIT_STAGE_HEAD, VIT_STAGE_SOFTMAX, VIT_STAGE_COUNT
};
typedef struct {
    double ms[VIT_STAGE_COUNT];
    long launches[VIT_STAGE_COUNT];
    long images;  /* images covered by the profiled forwards */
} vit_stage_times;

/* ViT-B/16-224: the reference's macros (ViT_seq.c:10-21). */
vit_config vit_config_b16(void);
/* tokens = (img/patch)^2 + 1 */
int vit_config_tokens(const vit_config *cfg);
/* Expected element count of weight tensor `index` (reference index map); 0 if out of range. */
size_t vit_config_weight_size(const vit_config *cfg, int index);
/* Algorithmic MACs of one image (17,563,828,224 for ViT-B/16-224). */
unsigned long long vit_config_macs_per_image(const vit_config *cfg);
/* MACs actually executed per image with vit_engine_options.prune_last_layer = 1. */
unsigned long long vit_config_macs_per_image_pruned(const vit_config *cfg);

void vit_engine_default_options(vit_engine_options *opt);
int vit_engine_create(vit_engine **out, const vit_config *cfg, const vit_engine_options *opt);
void vit_engine_destroy(vit_engine *e);
const char *vit_engine_last_error(const vit_engine *e);
const vit_config *vit_engine_config(const vit_engine *e);

/*
 * Validate (non-NULL, exact element count -- the reference's loader validates nothing,
 * Network.c:147) and upload all `count` = VIT_WEIGHT_COUNT(depth) tensors once.
 * The host arrays are only borrowed during the call.
 */
int vit_engine_load_weights(vit_engine *e, const Network *weights, int count);
/*
 * The same from a device-layout weight image (vit_io.h: built once, or read from the cache file): ONE host-to-device
 * copy of [fp32 tensors | bf16 GEMM operands]; an image without a bf16 section is converted on the device (one launch).
 */
int vit_engine_load_weight_image(vit_engine *e, const vit_weight_image *img);
/* Replicate the resident weights of `src` into `dst` (same model and dtype, any two devices of the process) with one
 * device-to-device copy -- over xGMI between GPUs, instead of another upload from the host. */
int vit_engine_copy_weights(vit_engine *dst, vit_engine *src);
/* Read the resident weights back as an image (bf16 section = the device's own conversion), e.g. to write the cache. */
int vit_engine_read_weight_image(vit_engine *e, vit_weight_image *img);

/*
 * Device-resident forward: d_images [n][C][S][S] fp32 -> d_probs [n][classes] fp32, both in
 * HBM, n arbitrary (processed in chunks of max_batch).  Asynchronous on `stream`
 * (a hipStream_t; NULL = the engine's own stream, then call vit_engine_sync()).
 * d_top1_label / d_top1_prob (device, n entries each) may be NULL.
 */
int vit_engine_forward_device(vit_engine *e, const float *d_images, int n, float *d_probs,
                              int *d_top1_label, float *d_top1_prob, void *stream);
int vit_engine_sync(vit_engine *e);

/*
 * Host-pointer forward with the reference's ownership rules (ViT_opencl.h:18): images[i] are
 * separately allocated CHW buffers, probs[i] caller-allocated [classes] rows.  Stages through
 * pinned buffers; blocking.
 */
int vit_engine_forward_host(vit_engine *e, const float *const *images, int n, float *const *probs);

/*
 * The fp32 GEMMs' helper-piece hand-over (csrc/vit_gemm_persistent.hip) since the last call, summed over the lanes: tiles whose
 * first K-steps came from a helper workgroup / tiles whose owner found no piece when it looked and computed all of it.  The
 * second number is lost time, never a wrong result (nothing in the hand-over waits or gives up).  Synchronises the device.
 */
int vit_engine_handover_stats(vit_engine *e, long *taken, long *recomputed);

/* Debug/test taps: copy the logits of the most recent chunk (rows = images of that chunk). */
int vit_engine_read_logits(vit_engine *e, float *dst, int rows);

int vit_engine_get_stage_times(vit_engine *e, vit_stage_times *out);  /* syncs, then reports */
void vit_engine_reset_stage_times(vit_engine *e);
int vit_engine_set_profile(vit_engine *e, int on);
int vit_engine_set_lanes(vit_engine *e, int lanes);

#ifdef __cplusplus
}
#endif
#endif /* VIT_ENGINE_H */
