/*
 * include/ViT_hip.h -- the drop-in forward surface (C-ABI shared library libvit_mi355x.so).
 *
 * Replaces the reference's OpenCL entry points one for one (ViT_opencl.h:18,21,22):
 *
 *   reference (ViT_opencl.h)                         this library
 *   void initialize_opencl();                 :21    void initialize_hip(void);
 *   void ViT_opencl(ImageData*, Network*, float**); :18    void ViT_hip(ImageData*, Network*, float**);
 *   void Release_opencl();                    :22    void Release_hip(void);
 *
 * and ALSO exports the reference's exact names (initialize_opencl / ViT_opencl / Release_opencl)
 * as aliases, so a Main.c-shaped caller (Main.c:19,57,86) links against this library unchanged.
 * get_source_code() and build_error() (ViT_opencl.h:19-20) are OpenCL build plumbing that
 * Main.c never calls; they are not part of the replacement surface.
 *
 * Contract kept from the reference (ViT_opencl.c:785-883, ViT_seq.h:20):
 *   - reads image[0].n images image[i].data (CHW fp32, model geometry), reads networks[0..151]
 *     (torchvision vit_b_16 order, already rounded by load_weights), writes 1000 softmax
 *     probabilities into the caller-allocated prb[i]; borrows every pointer; returns nothing;
 *   - global singleton state, not re-entrant, not thread-safe, one device, one in-order stream;
 *   - errors print "[file:line] HIP error <code> ..." and exit(EXIT_FAILURE) (CHECK_ERROR,
 *     ViT_opencl.h:7-11).
 * Differences, all additive: numerics follow ViT_seq.c (eps = 1e-6 LayerNorm, exact-erf GELU),
 * not kernel.cl; weights are uploaded once and cached per `networks` pointer;
 * a NULL / mis-sized weight is reported by index instead of crashing; all image[0].n images
 * are processed as one batch.
 *
 * Environment: VIT_HIP_DEVICE (device ordinal, default 0); VIT_HIP_DEVICES ("all" or a comma list of ordinals, e.g.
 * "0,1,2,3,4,5,6,7": one engine and one host thread per device, the weights uploaded once and replicated device to
 * device, image[0..n) split into contiguous slices -- the reference's image loop, ViT_opencl.c:802, cut across the
 * GPUs of the node; results are bit-identical to the single-device run); VIT_HIP_MAX_BATCH (chunk size per device,
 * default 256), VIT_HIP_LANES (concurrent sub-batches per chunk, default 1 for fp32, 2 for bf16), VIT_HIP_DTYPE ("bf16" selects the bf16
 * matrix-pipe variant: same top-1, |dprob| <= 2e-2 against the fp32 reference instead of 1e-4; default fp32),
 * VIT_HIP_PRUNE_LAST_LAYER (1: vit_engine_options.prune_last_layer, bit-identical probabilities; default 0).
 */
#ifndef VIT_HIP_H
#define VIT_HIP_H

#include "vit_types.h"

#ifdef __cplusplus
extern "C" {
#endif

void initialize_hip(void);
void ViT_hip(ImageData *image, Network *networks, float **prb);
void Release_hip(void);

/* Reference-named aliases (ViT_opencl.h:18,21,22). */
void initialize_opencl(void);
void ViT_opencl(ImageData *image, Network *networks, float **prb);
void Release_opencl(void);

/* Drop the cached device copy of the weights (call after modifying `networks` in place). */
void ViT_hip_invalidate_weights(void);
/* Number of devices the facade drives (0 before initialize_hip). */
int ViT_hip_device_count(void);
/*
 * Packed weight cache (vit_io.h, vit_weight_image): make the weights resident straight from a cache file -- one read,
 * one host-to-device copy, one device-to-device copy per further device -- without load_weights() at all; afterwards
 * ViT_hip() accepts any `networks` (NULL included).  source_dir (may be NULL) = the Network/ directory the file must
 * still match.  Returns 0, or -1 when the file is absent, stale or for another model (then load_weights() as usual).
 * ViT_hip_save_weight_cache writes the currently resident weights (0 / -1).
 */
int ViT_hip_load_weight_cache(const char *path, const char *source_dir);
int ViT_hip_save_weight_cache(const char *path, const char *source_dir);

#ifdef __cplusplus
}
#endif
#endif /* VIT_HIP_H */
