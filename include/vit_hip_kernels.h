/*
 * include/vit_hip_kernels.h -- the thin C-ABI over HIP.
 *
 * The host side of this project is C (as the reference's is); everything that needs the HIP
 * runtime or a gfx950 kernel sits behind these extern "C" entry points, compiled by hipcc
 * (csrc/ *.hip).  Only plain pointers, sizes and ints cross the boundary.  Each launcher is
 * asynchronous on the given stream (NULL = the default stream) and returns 0 on success or
 * a hipError_t value; vithip_error_string() renders it.
 *
 * What each kernel launcher replaces in the reference's OpenCL path:
 *   vithip_patch_embed_f32   Conv2d_opencl + CPU flatten_transpose/class_token/pos_emb
 *                            (ViT_opencl.c:126-180,806-810; kernel.cl:120-175)
 *   vithip_layernorm_f32     layer_norm_opencl (ViT_opencl.c:233-291; kernel.cl:6-80) -- but with
 *                            ViT_seq.c:103-121 numerics (eps = 1e-6, added in double)
 *   vithip_gemm_f32          enqueue_gemm_stage/add_bias_helper, fc1/gelu/fc2 kernels,
 *                            linear_layer_opencl and the CPU residual adds
 *                            (ViT_opencl.c:294-335,369-380,449-497,607-729,758-777;
 *                             kernel.cl:208-284,374-533) with exact-erf GELU (ViT_seq.c:231-233)
 *   vithip_attention_f32     the per-head scores GEMM / softmax / P.V chain
 *                            (ViT_opencl.c:499-602; kernel.cl:289-365), ViT_seq.c:156-215 numerics
 *   vithip_softmax_top1_f32  CPU Softmax (ViT_opencl.c:881 -> ViT_seq.c:304-324) and the
 *                            argmax loop of Main.c:62-72
 */
#ifndef VIT_HIP_KERNELS_H
#define VIT_HIP_KERNELS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *vithip_stream_t; /* hipStream_t     */
typedef void *vithip_event_t;  /* hipEvent_t      */
typedef void *vithip_graph_t;  /* hipGraphExec_t  */

typedef struct {
    char name[256];
    char arch[64];            /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    int compute_units;
    int clock_mhz;            /* max engine clock */
    int wavefront;
    int lds_per_block;        /* bytes */
    unsigned long long hbm_bytes;
} vithip_device_info;

/* ---- runtime plumbing --------------------------------------------------------------- */
const char *vithip_error_string(int code);
int vithip_device_count(int *count);
int vithip_set_device(int device);
int vithip_get_device_info(int device, vithip_device_info *info);
int vithip_malloc(void **ptr, size_t bytes);
int vithip_free(void *ptr);
int vithip_host_alloc(void **ptr, size_t bytes);   /* pinned */
int vithip_host_free(void *ptr);
int vithip_memcpy_h2d(void *dst, const void *src, size_t bytes, vithip_stream_t stream);
int vithip_memcpy_d2h(void *dst, const void *src, size_t bytes, vithip_stream_t stream);
int vithip_memcpy_d2d(void *dst, const void *src, size_t bytes, vithip_stream_t stream);
int vithip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, vithip_stream_t stream);
int vithip_memset(void *dst, int value, size_t bytes, vithip_stream_t stream);
int vithip_stream_create(vithip_stream_t *stream);
int vithip_stream_destroy(vithip_stream_t stream);
int vithip_stream_sync(vithip_stream_t stream);
int vithip_device_sync(void);
int vithip_event_create(vithip_event_t *event);
int vithip_event_destroy(vithip_event_t event);
int vithip_event_record(vithip_event_t event, vithip_stream_t stream);
int vithip_event_sync(vithip_event_t event);
int vithip_event_elapsed_ms(float *ms, vithip_event_t start, vithip_event_t stop);
int vithip_stream_wait_event(vithip_stream_t stream, vithip_event_t event);
int vithip_graph_begin(vithip_stream_t stream);
int vithip_graph_end(vithip_stream_t stream, vithip_graph_t *graph);
int vithip_graph_launch(vithip_graph_t graph, vithip_stream_t stream);
int vithip_graph_destroy(vithip_graph_t graph);

/* ---- kernels ------------------------------------------------------------------------- */

enum { VITHIP_EPI_BIAS = 0, VITHIP_EPI_BIAS_GELU = 1, VITHIP_EPI_BIAS_RESIDUAL = 2 };

/*
 * C[M][ldc] (first N columns) = epilogue(A[M][K] . W[N][K]^T + bias[N]):
 * the shape of every linear layer of the model (weights are [out][in], K contiguous in both
 * operands).  fp32 in, fp32 MFMA accumulate (v_mfma_f32_32x32x2_f32), fp32 out.
 *   EPI_BIAS_GELU:     y = 0.5*y*(1+erff(y/sqrtf(2)))                 (ViT_seq.c:231-233)
 *   EPI_BIAS_RESIDUAL: y += residual[m][n]; residual may alias C      (ViT_seq.c:286-288,297-299)
 * Requirements: K % 32 == 0; lda, ldw, ldc, ldr % 4 == 0; W must provide ceil(N/32)*32 rows
 * (pad rows are read, never used); all pointers 16-byte aligned.
 */
typedef struct {
    const float *A; int lda;
    const float *W; int ldw;
    const float *bias;
    const float *residual; int ldr;
    float *C; int ldc;
    int M, N, K;
    int epilogue;
    /* tuning, per call (the library keeps no process-wide tuning state); 0 = auto:
     * tile: 0 = auto (persistent walk of 128x128 tiles from ~1,280 tiles on, 64x64 / 32x32 tiles as the problem
     *   shrinks); one software-pipelined tile per workgroup: 10 = 128x128, 6 = 128x128 with K step 16, 7 = 256x128, 8 = 128x64,
     *   11 = 64x64; 12 = 32x32 on 16x16x4 MFMA (K % 128 == 0); 9 = persistent 128x128.  All of them give the same bits.
     * group_m: tile rows per L2 group of the XCD-aware tile walk (0 = default: plain N-fastest order when N is at most 8 tiles of
     *   128 wide, groups of 4 tile rows beyond that; 1 = plain N-fastest order). */
    int tile, group_m;
    /* optional scratch: the HANDLE made by vithip_gemm_f32_workspace_create() on the device the launch runs on (one per
     * stream: launches that share it must be ordered).  With it, large problems whose tile count is not a multiple of the
     * workgroup count hand the first K-steps of the last round's tiles to the workgroups that would idle (bit-identical
     * results: the accumulation chain moves between workgroups, it is not split; csrc/vit_gemm_persistent.hip).  Nobody
     * waits in that hand-over: an owner that does not find its piece computes the whole tile.  NULL = off. */
    void *workspace;
    /* testing: 1 = the helper workgroups run their pieces LAST, so that owners look for them too early and take the
     * compute-it-yourself path (results must not change); 0 = normal. */
    int handover_test;
    /* LayerNorm folded into the GEMM behind it (both NULL = off; EPI_BIAS and EPI_BIAS_GELU only): A holds the UN-normalised
     * rows x, W and bias are the gamma- and beta-folded operands of vithip_ln_fold_weights_f32(), ln_rows [M][2] = (rstd, mean)
     * per row of A (vithip_rowstats_f32), ln_colsum [N] -- or NULL when W is the CENTRED weight of
     * vithip_ln_fold_weights_f32_centered(), whose product needs no centring (the term below is then skipped); the epilogue computes
     *     fmaf(rstd, fmaf(-mean, colsum, acc), bias) = LayerNorm(x) . W^T + b                 (ViT_seq.c:103-121 is the LayerNorm)
     * with the same two roundings in every kernel (the 32x32 kernels take the inner one as a rank-1 matrix instruction, four per
     * wave and tile: csrc/vit_gemm_common.hpp), so that the tile shapes stay bit-identical to each other.
     * (Rows of near-zero variance: the cancellation error of the inner term is scaled by rstd, up to 1e3 there, where
     * LayerNorm-then-GEMM gives exact zeros; real residual rows have rstd of order 1.  Measured, not asserted away:
     * tests/test_gpu_lnfold.py::test_fp32_fold_on_near_constant_rows_stays_inside_its_stated_bound.)  The normalised activations never exist in memory: the pass that wrote them (read x, write y: 310 MB at batch 256, 24 times per
     * ViT-B/16 forward) becomes a pass that reads x and writes 8 bytes per row. */
    const float *ln_rows, *ln_colsum;
    /* ... and the producer side (NULL = off; EPI_BIAS_RESIDUAL only, N % 64 == 0, N <= 2048): stats_out [M][2] receives (rstd,
     * mean) of the rows of C as stored -- what vithip_rowstats_f32(C) would write, bit for bit.  With stats_partials
     * ([N / 64][M][2] floats of scratch) the persistent walk takes the sums in its epilogue, where the row is in the accumulators
     * (one small launch then finalises them: vithip_gemm_f32_stats_in_epilogue() says whether a call would); any other kernel,
     * or no scratch, and the call runs vithip_rowstats_f32 behind the GEMM. */
    float *stats_out, *stats_partials;
} vithip_gemm_args;
int vithip_gemm_f32(vithip_stream_t stream, const vithip_gemm_args *args);
int vithip_gemm_f32_stats_in_epilogue(const vithip_gemm_args *args);  /* 1 / 0 (0 also for arguments vithip_gemm_f32 would refuse) */
/* ---- LayerNorm folding, fp32: LN(x) . W^T + b = rstd * (x . (gamma*W)^T) - rstd * mean * colsum(gamma*W) + (b + W . beta).
 * Wf[n][k] = gamma[k] * W[n][k] (fp32 product); colsum[n] = sum_k Wf[n][k] and bias_f[n] = bias[n] + sum_k beta[k] * W[n][k], both
 * accumulated in double and rounded once.  W fp32 [N][K], K % 4 == 0, 16-byte aligned. */
int vithip_ln_fold_weights_f32(vithip_stream_t stream, const float *W, const float *bias, const float *gamma, const float *beta,
                               float *Wf, float *colsum, float *bias_f, int N, int K);
/* The CENTRED form of the same fold (round 5; what vit_engine uses): Wc[n][k] = gamma[k] * W[n][k] - cbar[n] with
 * cbar[n] = sum_k gamma[k] * W[n][k] / K (sum and difference in double, one rounding per weight).  Since mean(x) = sum_k x[k] / K,
 *     x . Wc^T = x . (gamma*W)^T - mean * colsum(gamma*W):
 * the GEMM delivers the centred product itself and its epilogue only scales: pass Wc and bias_f with ln_rows set and
 * ln_colsum = NULL.  residual_colsum[n] = sum_k Wc[n][k] (what the rounding of the weights left of the column sum, for the
 * record: ~1e-7 of |W|).  Same alignment rules. */
int vithip_ln_fold_weights_f32_centered(vithip_stream_t stream, const float *W, const float *bias, const float *gamma,
                                        const float *beta, float *Wc, float *residual_colsum, float *bias_f, int N, int K);
/* rows_out[m] = (rstd, mean) of x [rows][ldx], dim % 64 == 0, dim <= 2048: mean and E[x^2] - mean^2 as ViT_seq.c:103-121
 * takes them, 1 / sqrtf((double)var + 1e-6).  The sums run in ONE documented order (per 64-column strip: columns c and c + 32
 * added first, then a 32-lane butterfly 16, 8, 4, 2, 1; strips in ascending order), the order a GEMM epilogue that holds the
 * row in its accumulators can reproduce -- so that whoever produces the statistics produces the same bits. */
int vithip_rowstats_f32(vithip_stream_t stream, const float *x, size_t ldx, float *rows_out, int rows, int dim);
/* partials [dim / 64][rows][2] (sum, sum of squares per 64-column strip, in the order above) -> rows_out [rows][2] */
int vithip_rowstats_finalize_f32(vithip_stream_t stream, const float *partials, int rows, int dim, float *rows_out);
size_t vithip_gemm_f32_workspace_bytes(void);             /* device bytes a workspace takes on the current device */
int vithip_gemm_f32_workspace_create(void **workspace);   /* on the current device */
int vithip_gemm_f32_workspace_destroy(void *workspace);
/* Hand-over counters since the last call (blocking; the stream's launches should be complete): tiles finished from a parked
 * piece / tiles an owner computed whole because the piece was not there yet.  Either pointer may be NULL.  Clears them. */
int vithip_gemm_f32_workspace_stats(void *workspace, int *taken, int *recomputed);
void *vithip_gemm_f32_workspace_device_ptr(void *workspace);  /* [flag per owner ... ][64 KB slots]; tests */
/* ---- bf16 variant (BASELINE.json configs[2]) ---------------------------------
 * bf16 values are raw uint16 (upper half of the fp32 bit pattern, round-to-nearest-even). */
enum { VITHIP_BF16_EPI_BF16 = 0, VITHIP_BF16_EPI_BF16_GELU = 1, VITHIP_BF16_EPI_F32_RESIDUAL = 2,
       VITHIP_BF16_EPI_F32_EMBED = 3 /* internal to vithip_patch_embed_bf16 */ };
typedef struct {
    const unsigned short *A; int lda;   /* bf16 [M][lda], K contiguous */
    const unsigned short *W; int ldw;   /* bf16 [N][ldw] */
    const float *bias;                  /* fp32 [N] */
    const float *residual; int ldr;     /* fp32, EPI_F32_RESIDUAL only; may alias C */
    void *C; int ldc;                   /* bf16 (EPI_BF16, EPI_BF16_GELU) or fp32 (EPI_F32_RESIDUAL) */
    int M, N, K;                        /* K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0 */
    int epilogue;
    /* testing, per call: 0 auto (ping-pong kernel whenever K >= 128), 1 two-stage kernel (vit_gemm_bf16.hip: the fallback for
     * K < 128), 2 ping-pong kernel (vit_gemm_bf16_pp.hip; invalid-value error when K < 128) */
    int variant;
    /* LayerNorm folded into the GEMMs either side of it (all NULL = off; ping-pong kernel only, i.e. K >= 128):
     * producer, EPI_F32_RESIDUAL with x16 and row_partials set: besides C the epilogue stores bf16(C) to x16 [M][ldx16]
     *   (ldx16 % 8 == 0, 16-byte aligned) and the partial (sum, sum of squares) of every row over every 64-column strip to
     *   row_partials [vithip_ln_strips(N)][M][2]; vithip_rowstats_finalize() turns them into ln_rows;
     * consumer, EPI_BF16 / EPI_BF16_GELU with ln_rows and ln_colsum set: A is the UN-normalised bf16 row, W and bias are the
     *   gamma- and beta-folded operands of vithip_ln_fold_weights(), ln_rows [M][2] = (rstd, mean * rstd) per row of A and
     *   ln_colsum [N]; the epilogue computes rstd * acc - (mean * rstd) * colsum + bias = LayerNorm(x) . W^T + b. */
    const float *ln_rows, *ln_colsum;
    unsigned short *x16; int ldx16;
    float *row_partials;
} vithip_gemm_bf16_args;
/* C = epilogue(A . W^T + bias) on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16 in the ping-pong kernel,
 * v_mfma_f32_32x32x16_bf16 in the two-stage one), fp32 accumulate.  BF16_GELU rounds gelu(acc + bias) to bf16 (a
 * polynomial erfc whose error stays below 5 % of half a bf16 ulp); F32_RESIDUAL adds an fp32 residual in fp32. */
int vithip_gemm_bf16(vithip_stream_t stream, const vithip_gemm_bf16_args *args);
/* ---- LayerNorm folding (bf16 forward): LN(x) . W^T + b = rstd * (x . (gamma*W)^T) - rstd * mean * colsum(gamma*W) + (b + W . beta),
 * so the normalised activations never exist in memory: the residual GEMM in front stores bf16(x) and row sums, the GEMM
 * behind multiplies the raw bf16 rows with the folded weight and rescales in its epilogue (ViT_seq.c:103-121 is the
 * LayerNorm being folded: mean, var = E[x^2] - mean^2, 1/sqrt(var + 1e-6)). */
/* Wf[n][k] = bf16(gamma[k] * W[n][k]); colsum[n] = sum_k float(Wf[n][k]) (the ROUNDED values: the mean term then cancels
 * exactly what the matrix pipe accumulates); bias_f[n] = bias[n] + sum_k beta[k] * W[n][k].  W fp32 [N][K], K % 4 == 0. */
int vithip_ln_fold_weights(vithip_stream_t stream, const float *W, const float *bias, const float *gamma, const float *beta,
                           unsigned short *Wf, float *colsum, float *bias_f, int N, int K);
/* The same with output features [0, scale_rows) made `scale` times as large (rows of Wf and their bias_f; colsum is taken of the
 * scaled, rounded rows): the engine folds the factor of the attention scores' exponent, VITHIP_QSCALE, into the Q rows of in_proj,
 * so that q is still rounded to bf16 once and the attention kernels need no scaling pass (vithip_attention_bf16io_qscaled). */
#define VITHIP_QSCALE 0.18033688011112042f /* (1/sqrtf(64)) * log2(e): softmax(q.k / 8) = 2^(QSCALE q.k - max) / sum */
int vithip_ln_fold_weights_scaled(vithip_stream_t stream, const float *W, const float *bias, const float *gamma, const float *beta,
                                  unsigned short *Wf, float *colsum, float *bias_f, int N, int K, int scale_rows, float scale);
/* x16 = bf16(x) and rows[m] = (rstd, mean * rstd) of fp32 rows x [rows][ldx] (the first LayerNorm of the stack, which has no
 * residual GEMM in front of it). */
int vithip_rowstats_bf16(vithip_stream_t stream, const float *x, size_t ldx, unsigned short *x16, size_t ldx16,
                         float *rows_out, int rows, int dim);
/* strips = vithip_ln_strips(N) = 4 * ceil(N / 256): partials [strips][rows][2] (sum, sum of squares) -> rows_out [rows][2]. */
int vithip_ln_strips(int N);
int vithip_rowstats_finalize(vithip_stream_t stream, const float *partials, int strips, int rows, int dim, float *rows_out);
/* dst[r][0..width) = src[r * src_stride .. + width): a strided row subset made compact (e.g. the pairs of the class rows). */
int vithip_gather_rows_f32(vithip_stream_t stream, const float *src, size_t src_stride, float *dst, size_t dst_stride, int rows,
                           int width);
/* LayerNorm with fp32 statistics and a bf16 store; attention reading bf16 Q/K/V [n*tokens][3*heads*64] and
 * writing bf16 [n*tokens][heads*64]: both products on bf16 MFMA with fp32 softmax (P rounded to bf16 once);
 * vithip_attention_bf16io_f32math: same I/O, K/V widened to fp32 in LDS and the fp32 kernel's arithmetic (cross-check). */
int vithip_layernorm_f32_bf16out(vithip_stream_t stream, const float *x, size_t ldx, unsigned short *y, size_t ldy,
                                 const float *gamma, const float *beta, int rows, int dim);
int vithip_attention_bf16io(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out,
                            int n_images, int tokens, int heads);
int vithip_attention_bf16io_f32math(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out,
                                    int n_images, int tokens, int heads);
/* As vithip_attention_f32 / vithip_attention_bf16io, but only the first q_rows query rows of every image are computed
 * and stored (rows q_rows.. of `out` are left untouched); tokens <= 224.  q_rows = 1 is the class token. */
int vithip_attention_f32_rows(vithip_stream_t stream, const float *qkv, float *out, int n_images, int tokens, int heads,
                              int q_rows);
int vithip_attention_bf16io_rows(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out, int n_images,
                                 int tokens, int heads, int q_rows);
/* As vithip_attention_bf16io (q_rows = tokens) / _rows, for Q columns that already hold VITHIP_QSCALE * q.  For the streamed
 * kernel (225..704 tokens) this removes the scale-and-subtract of every score: the score accumulators start at -max; the
 * resident kernel (up to 224) and the chunked one (beyond 704) take the factor 1 in place of VITHIP_QSCALE. */
int vithip_attention_bf16io_qscaled(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out, int n_images,
                                    int tokens, int heads, int q_rows);
/* Patch embedding on the bf16 matrix pipe (same result layout as vithip_patch_embed_f32: x[n][tokens][D] fp32 with
 * class token and pos_emb applied; ViT_seq.c:25-101): the images are cut into bf16 patch rows
 * (patches16: workspace of n * (img/patch)^2 * chans*patch^2 bf16), multiplied with the bf16 conv weight
 * conv_w16 [D][chans*patch^2] with fp32 accumulation, bias / pos_emb / class token added in fp32.
 * Needs chans*patch^2 % 64 == 0, chans*patch^2 >= 128, D % 4 == 0, patch % 8 == 0. */
int vithip_patch_embed_bf16(vithip_stream_t stream, const float *images, const unsigned short *conv_w16,
                            const float *conv_b, const float *cls, const float *pos, float *x,
                            unsigned short *patches16, int n_images, int img_size, int patch_size, int in_chans,
                            int embed_dim);
/* The same result as ONE implicit GEMM that reads the NCHW fp32 images directly (pixels rounded to bf16 in the A-tile loader; no
 * staging pass, no workspace).  A tested alternative, not the engine's default: 2.4 ms against 1.0 ms at batch 2048, bound by
 * the fp32 pixel bytes every 128-wide N tile re-reads (csrc/vit_patch_embed_bf16.hip).
 * Needs patch % 4 == 0, chans*patch^2 % 32 == 0, img % 4 == 0, n * patches <= 2^24. */
int vithip_patch_embed_bf16_implicit(vithip_stream_t stream, const float *images, const unsigned short *conv_w16,
                                     const float *conv_b, const float *cls, const float *pos, float *x,
                                     int n_images, int img_size, int patch_size, int in_chans, int embed_dim);
/* dst[i] = bf16(src[i]), round to nearest even; count % 4 == 0. */
int vithip_f32_to_bf16(vithip_stream_t stream, const float *src, unsigned short *dst, size_t count);

/*
 * Patch embedding straight from NCHW images (implicit GEMM over the 16x16 patches), with the
 * embedding tail fused into the store:
 *   x[img][0][:]     = cls[:] + pos[0][:]
 *   x[img][1+p][:]   = conv_bias[:] + sum_{ic,kh,kw} image * conv_w + pos[1+p][:]
 * images: [n][C][S][S]; conv_w: [D][C*P*P]; x: [n][T][D], T = (S/P)^2 + 1.
 */
int vithip_patch_embed_f32(vithip_stream_t stream, const float *images, const float *conv_w,
                           const float *conv_b, const float *cls, const float *pos, float *x,
                           int n_images, int img_size, int patch_size, int in_chans, int embed_dim);

/*
 * y[r][0..dim) = (x[r] - mean) * inv_std * gamma + beta for rows r = 0..rows-1 where row r
 * starts at x + r*ldx (ldx lets the final LayerNorm touch only the class-token rows).
 * mean/var as ViT_seq.c:103-121: var = E[x^2] - mean^2, inv_std = 1/sqrtf((double)var + 1e-6).
 * dim % 4 == 0, dim <= 2048.
 */
int vithip_layernorm_f32(vithip_stream_t stream, const float *x, size_t ldx, float *y, size_t ldy,
                         const float *gamma, const float *beta, int rows, int dim);

/*
 * Fused scaled-dot-product attention, one workgroup per (image, head) (x blocks of 256 queries when chunked).
 * qkv: [n*T][3*D] rows = tokens, columns [Q | K | V], head h = columns 64h..64h+63 of each.
 * out: [n*T][D].  scores = q.k / sqrtf(64); row softmax with max subtraction; out = P.V.
 * head_dim must be 64.  Up to 224 tokens K and V of one head stay resident in LDS; longer sequences
 * (ViT-L/16-384: 577) stream K/V through LDS in 224-key chunks with an online softmax.
 */
int vithip_attention_f32(vithip_stream_t stream, const float *qkv, float *out,
                         int n_images, int tokens, int heads);

/*
 * probs[r][0..classes) = softmax(logits[r]) (ViT_seq.c:304-324) and the top-1 record
 * (first index of the maximum probability, Main.c:62-70).  top1_label / top1_prob may be NULL.
 */
int vithip_softmax_top1_f32(vithip_stream_t stream, const float *logits, int ld_logits,
                            float *probs, int ld_probs, int *top1_label, float *top1_prob,
                            int rows, int classes);

#ifdef __cplusplus
}
#endif
#endif /* VIT_HIP_KERNELS_H */
