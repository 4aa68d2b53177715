/*
 * oracle/vit_cpu_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's scalar forward pass (ViT_seq.c) used ONLY as
 * the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  The shipped library (vision-transformer-opencl_amd/) never links, loads or
 * calls anything declared here.
 *
 * Parity status: PINNED.  In the build container the restatement is checked
 * bit-for-bit against the reference's own ViT_seq.c compiled from
 * /root/reference (oracle/_ref/libvitseq_ref.so, recipe in oracle/Makefile) by
 * tests/test_oracle_vs_reference.py, and against the golden vectors that the
 * compiled reference emitted (tests/golden/, generator oracle/gen_golden.py).
 * The reference's own end-to-end fixture (Data/answer_result.txt) needs
 * Data/input-100.bin and 36 weight blobs that are absent from the mount, so that
 * one fixture cannot be exercised here (tests auto-skip).
 *
 * Every function cites the ViT_seq.c lines it follows.  Arithmetic order is the
 * reference's: sequential fp32 accumulation in index order, linear layers start
 * their accumulator at the bias, LayerNorm adds a *double* eps.  Compile with
 * -ffp-contract=off (see oracle/Makefile).  The only generalisation is that the
 * compile-time macros of ViT_seq.c:10-21 become a runtime config so that the
 * same code also serves reduced-size test models and ViT-L/16-384.
 */
#ifndef VIT_CPU_REF_H
#define VIT_CPU_REF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ViT_seq.c:10-21 as a struct.  hidden_dim = (int)(embed_dim * mlp_ratio). */
typedef struct {
    int img_size;
    int patch_size;
    int in_chans;
    int num_classes;
    int embed_dim;
    int depth;
    int num_heads;
    int hidden_dim;
} vitref_config;

/* Same layout as the reference's Network (Network.h:18-21). */
typedef struct {
    float *data;
    size_t size;
} vitref_tensor;

/* Threads used by the outer (token / output-channel) loops.  Each output element's
 * accumulation chain stays sequential, so results are bit-identical for any count. */
void vitref_set_threads(int n);
int vitref_tokens(const vitref_config *cfg);

void vitref_round_weights(float *w, size_t n);                       /* Network.c:184-187 */
void vitref_conv2d(const vitref_config *cfg, const float *input, float *output,
                   const float *weight, const float *bias);          /* ViT_seq.c:25-50   */
void vitref_flatten_transpose(const vitref_config *cfg, const float *input, float *output); /* :52-70 */
void vitref_class_token(const vitref_config *cfg, const float *patch_tokens, float *final_tokens,
                        const float *cls);                           /* ViT_seq.c:72-90   */
void vitref_pos_emb(const vitref_config *cfg, const float *input, float *output,
                    const float *pos);                               /* ViT_seq.c:92-101  */
void vitref_layer_norm(const float *input, float *output, int tokens, int dim,
                       const float *weight, const float *bias);      /* ViT_seq.c:103-121 */
void vitref_multihead_attn(const float *input, float *output, int tokens, int dim, int heads,
                           const float *in_weight, const float *in_bias,
                           const float *out_weight, const float *out_bias); /* :123-229 */
/* Attention core only (scores, softmax, P.V) on given Q,K,V [tokens][dim]: ViT_seq.c:156-215 */
void vitref_attention_core(const float *Q, const float *K, const float *V, float *attn_output,
                           int tokens, int dim, int heads);
float vitref_gelu(float x);                                          /* ViT_seq.c:231-233 */
void vitref_linear(const float *input, float *output, int tokens, int in_features,
                   int out_features, const float *weight, const float *bias); /* :240-250 */
void vitref_mlp_block(const float *input, float *output, int tokens, int dim, int hidden,
                      const float *fc1_w, const float *fc1_b,
                      const float *fc2_w, const float *fc2_b);       /* ViT_seq.c:251-268 */
/* w[0..11] = ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b */
void vitref_encoder(const float *input, float *output, int tokens, int dim, int heads, int hidden,
                    const float *const w[12]);                       /* ViT_seq.c:271-302 */
void vitref_softmax(const float *logits, float *probabilities, int length); /* :304-324 */

/*
 * ViT_seq.c:337-439 for one image.  weights[] uses the reference's index map
 * (0 cls, 1/2 conv, 3 pos, 4+12l+k encoder l, 4+12*depth ln_w, +1 ln_b, +2 head_w, +3 head_b).
 * Optional taps (may be NULL): stages = (depth+1) x tokens x dim floats: the embedding
 * followed by every encoder output; logits = num_classes floats.
 */
void vitref_forward_image(const vitref_config *cfg, const float *image,
                          const vitref_tensor *weights, float *probabilities,
                          float *logits, float *stages);
/* Batch loop of ViT_seq.c:354: images[i] are separate CHW buffers, prob[i] caller-allocated. */
void vitref_forward(const vitref_config *cfg, const float *const *images, int n,
                    const vitref_tensor *weights, float *const *prob);

#ifdef __cplusplus
}
#endif
#endif
