/*
 * oracle/vit_cpu_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see vit_cpu_ref.h).
 *
 * Scalar fp32 ViT forward with the arithmetic order of the reference's ViT_seq.c.
 * Parity: PINNED against the compiled reference (bit-identical, tests/test_oracle_vs_reference.py)
 * and against tests/golden/ vectors emitted by the compiled reference.
 *
 * Rules kept from the reference:
 *   - every dot product is a left-to-right fp32 chain in index order;
 *   - conv / linear chains start at the bias (ViT_seq.c:31,137,221,243), attention
 *     chains start at 0.0f (ViT_seq.c:167,198);
 *   - LayerNorm: fp32 sum and sum of squares, var = sumsq/dim - mean*mean, the eps
 *     1e-6 is a double literal so the add happens in double before sqrtf
 *     (ViT_seq.c:21,115);
 *   - scores are divided by sqrtf(head_dim); softmax divides by the sum;
 *   - GELU = 0.5f*x*(1.0f+erff(x/sqrtf(2.0f))).
 * OpenMP only splits loops over independent output elements.
 */
#include "vit_cpu_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void vitref_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

int vitref_tokens(const vitref_config *cfg) {
    int g = cfg->img_size / cfg->patch_size;
    return g * g + 1;
}

static float *falloc(size_t n) {
    float *p = (float *)malloc(sizeof(float) * (n ? n : 1));
    if (!p) abort();
    return p;
}

/* Network.c:184-187: buffer[i] = roundf(buffer[i] * 1000000.0f) / 1000000.0f */
void vitref_round_weights(float *w, size_t n) {
    for (size_t i = 0; i < n; ++i) w[i] = roundf(w[i] * 1000000.0f) / 1000000.0f;
}

/* ViT_seq.c:25-50.  out[oc][oh][ow] = bias[oc] + sum over (ic, kh, kw) in that order. */
void vitref_conv2d(const vitref_config *cfg, const float *input, float *output,
                   const float *weight, const float *bias) {
    const int P = cfg->patch_size, S = cfg->img_size, C = cfg->in_chans, D = cfg->embed_dim;
    const int G = S / P;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int oc = 0; oc < D; ++oc) {
        for (int oh = 0; oh < G; ++oh) {
            for (int ow = 0; ow < G; ++ow) {
                float acc = bias[oc];
                for (int ic = 0; ic < C; ++ic)
                    for (int kh = 0; kh < P; ++kh)
                        for (int kw = 0; kw < P; ++kw) {
                            int in_idx = (ic * S + (oh * P + kh)) * S + (ow * P + kw);
                            int w_idx = ((oc * C + ic) * P + kh) * P + kw;
                            acc += input[in_idx] * weight[w_idx];
                        }
                output[(oc * G + oh) * G + ow] = acc;
            }
        }
    }
}

/* ViT_seq.c:52-70: [D][G*G] -> [G*G][D] */
void vitref_flatten_transpose(const vitref_config *cfg, const float *input, float *output) {
    const int D = cfg->embed_dim, G = cfg->img_size / cfg->patch_size, NP = G * G;
    for (int p = 0; p < NP; ++p)
        for (int oc = 0; oc < D; ++oc) output[p * D + oc] = input[oc * NP + p];
}

/* ViT_seq.c:72-90: row 0 = class token, rows 1.. = patch tokens */
void vitref_class_token(const vitref_config *cfg, const float *patch_tokens, float *final_tokens,
                        const float *cls) {
    const int D = cfg->embed_dim, G = cfg->img_size / cfg->patch_size, NP = G * G;
    memcpy(final_tokens, cls, sizeof(float) * D);
    memcpy(final_tokens + D, patch_tokens, sizeof(float) * (size_t)D * NP);
}

/* ViT_seq.c:92-101 */
void vitref_pos_emb(const vitref_config *cfg, const float *input, float *output, const float *pos) {
    const int total = vitref_tokens(cfg) * cfg->embed_dim;
    for (int i = 0; i < total; ++i) output[i] = input[i] + pos[i];
}

/* ViT_seq.c:103-121 */
void vitref_layer_norm(const float *input, float *output, int tokens, int dim,
                       const float *weight, const float *bias) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int t = 0; t < tokens; ++t) {
        const float *x = input + (size_t)t * dim;
        float *y = output + (size_t)t * dim;
        float sum = 0.0f, sum_sq = 0.0f;
        for (int i = 0; i < dim; ++i) {
            float v = x[i];
            sum += v;
            sum_sq += v * v;
        }
        float mean = sum / dim;                    /* int -> float, fp32 divide (:113) */
        float var = sum_sq / dim - mean * mean;    /* :114 */
        float inv_std = 1.0f / sqrtf(var + 1e-6);  /* double add, then float sqrt (:115) */
        for (int i = 0; i < dim; ++i) y[i] = (x[i] - mean) * inv_std * weight[i] + bias[i];
    }
}

/* ViT_seq.c:156-215 for every head.  Q,K,V,attn_output are [tokens][dim]. */
void vitref_attention_core(const float *Q, const float *K, const float *V, float *attn_output,
                           int tokens, int dim, int heads) {
    const int hd = dim / heads;
    const float scale_div = sqrtf((float)hd);      /* :174 */
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1) collapse(2)
    for (int h = 0; h < heads; ++h) {
        for (int i = 0; i < tokens; ++i) {
            const int off = h * hd;
            float *row = falloc((size_t)tokens);
            for (int j = 0; j < tokens; ++j) {
                float s = 0.0f;
                for (int d = 0; d < hd; ++d) s += Q[(size_t)i * dim + off + d] * K[(size_t)j * dim + off + d];
                row[j] = s / scale_div;
            }
            float mx = row[0];                     /* :178-182 */
            for (int j = 1; j < tokens; ++j)
                if (row[j] > mx) mx = row[j];
            float denom = 0.0f;                    /* :183-187 */
            for (int j = 0; j < tokens; ++j) {
                row[j] = expf(row[j] - mx);
                denom += row[j];
            }
            for (int j = 0; j < tokens; ++j) row[j] /= denom; /* :188-190 */
            for (int d = 0; d < hd; ++d) {         /* :196-204 */
                float acc = 0.0f;
                for (int j = 0; j < tokens; ++j) acc += row[j] * V[(size_t)j * dim + off + d];
                attn_output[(size_t)i * dim + off + d] = acc;
            }
            free(row);
        }
    }
}

/* ViT_seq.c:123-229 */
void vitref_multihead_attn(const float *input, float *output, int tokens, int dim, int heads,
                           const float *in_weight, const float *in_bias,
                           const float *out_weight, const float *out_bias) {
    const size_t n = (size_t)tokens * dim;
    float *Q = falloc(n), *K = falloc(n), *V = falloc(n), *A = falloc(n);
    /* :134-147 -- rows 0..dim-1 of in_weight are Q, dim..2dim-1 K, 2dim..3dim-1 V; bias first */
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int t = 0; t < tokens; ++t) {
        const float *x = input + (size_t)t * dim;
        for (int i = 0; i < dim; ++i) {
            float q = in_bias[i], k = in_bias[dim + i], v = in_bias[2 * dim + i];
            const float *wq = in_weight + (size_t)i * dim;
            const float *wk = in_weight + (size_t)(dim + i) * dim;
            const float *wv = in_weight + (size_t)(2 * dim + i) * dim;
            for (int j = 0; j < dim; ++j) {
                q += x[j] * wq[j];
                k += x[j] * wk[j];
                v += x[j] * wv[j];
            }
            Q[(size_t)t * dim + i] = q;
            K[(size_t)t * dim + i] = k;
            V[(size_t)t * dim + i] = v;
        }
    }
    vitref_attention_core(Q, K, V, A, tokens, dim, heads);
    vitref_linear(A, output, tokens, dim, dim, out_weight, out_bias); /* :219-227, same chain */
    free(Q); free(K); free(V); free(A);
}

/* ViT_seq.c:231-233 */
float vitref_gelu(float x) { return 0.5f * x * (1.0f + erff(x / sqrtf(2.0f))); }

/* ViT_seq.c:240-250 (and :219-227): y[t][o] = b[o] + sum_i x[t][i]*W[o][i], bias first */
void vitref_linear(const float *input, float *output, int tokens, int in_features,
                   int out_features, const float *weight, const float *bias) {
#pragma omp parallel for num_threads(g_threads) schedule(static) collapse(2)
    for (int t = 0; t < tokens; ++t) {
        for (int o = 0; o < out_features; ++o) {
            const float *x = input + (size_t)t * in_features;
            const float *w = weight + (size_t)o * in_features;
            float acc = bias[o];
            for (int i = 0; i < in_features; ++i) acc += x[i] * w[i];
            output[(size_t)t * out_features + o] = acc;
        }
    }
}

/* ViT_seq.c:251-268 */
void vitref_mlp_block(const float *input, float *output, int tokens, int dim, int hidden,
                      const float *fc1_w, const float *fc1_b, const float *fc2_w, const float *fc2_b) {
    const size_t n = (size_t)tokens * hidden;
    float *h = falloc(n);
    vitref_linear(input, h, tokens, dim, hidden, fc1_w, fc1_b);
    for (size_t i = 0; i < n; ++i) h[i] = vitref_gelu(h[i]);
    vitref_linear(h, output, tokens, hidden, dim, fc2_w, fc2_b);
    free(h);
}

/* ViT_seq.c:271-302 */
void vitref_encoder(const float *input, float *output, int tokens, int dim, int heads, int hidden,
                    const float *const w[12]) {
    const size_t n = (size_t)tokens * dim;
    float *ln1 = falloc(n), *attn = falloc(n), *res = falloc(n), *ln2 = falloc(n), *mlp = falloc(n);
    vitref_layer_norm(input, ln1, tokens, dim, w[0], w[1]);
    vitref_multihead_attn(ln1, attn, tokens, dim, heads, w[2], w[3], w[4], w[5]);
    for (size_t i = 0; i < n; ++i) res[i] = input[i] + attn[i];        /* :286-288 */
    vitref_layer_norm(res, ln2, tokens, dim, w[6], w[7]);
    vitref_mlp_block(ln2, mlp, tokens, dim, hidden, w[8], w[9], w[10], w[11]);
    for (size_t i = 0; i < n; ++i) output[i] = res[i] + mlp[i];        /* :297-299 */
    free(ln1); free(attn); free(res); free(ln2); free(mlp);
}

/* ViT_seq.c:304-324 */
void vitref_softmax(const float *logits, float *probabilities, int length) {
    float mx = logits[0];
    for (int i = 1; i < length; ++i)
        if (logits[i] > mx) mx = logits[i];
    float denom = 0.0f;
    for (int i = 0; i < length; ++i) {
        probabilities[i] = expf(logits[i] - mx);
        denom += probabilities[i];
    }
    for (int i = 0; i < length; ++i) probabilities[i] /= denom;
}

/* ViT_seq.c:354-438 for one image (nothing leaks, no per-layer printf). */
void vitref_forward_image(const vitref_config *cfg, const float *image,
                          const vitref_tensor *weights, float *probabilities,
                          float *logits, float *stages) {
    const int D = cfg->embed_dim, T = vitref_tokens(cfg), NP = T - 1;
    const size_t n = (size_t)T * D;
    float *conv = falloc((size_t)D * NP), *flat = falloc((size_t)D * NP);
    float *tok = falloc(n), *x = falloc(n), *y = falloc(n);

    vitref_conv2d(cfg, image, conv, weights[1].data, weights[2].data);   /* :356 */
    vitref_flatten_transpose(cfg, conv, flat);                           /* :358 */
    vitref_class_token(cfg, flat, tok, weights[0].data);                 /* :360 */
    vitref_pos_emb(cfg, tok, x, weights[3].data);                        /* :362 */
    if (stages) memcpy(stages, x, sizeof(float) * n);

    for (int l = 0; l < cfg->depth; ++l) {                               /* :366-426 */
        const float *w[12];
        for (int k = 0; k < 12; ++k) w[k] = weights[4 + 12 * l + k].data;
        vitref_encoder(x, y, T, D, cfg->num_heads, cfg->hidden_dim, w);
        if (stages) memcpy(stages + (size_t)(l + 1) * n, y, sizeof(float) * n);
        float *tmp = x; x = y; y = tmp;
    }

    const int base = 4 + 12 * cfg->depth;
    vitref_layer_norm(x, y, T, D, weights[base].data, weights[base + 1].data); /* :429 */
    float *lg = falloc((size_t)cfg->num_classes);
    vitref_linear(y, lg, 1, D, cfg->num_classes, weights[base + 2].data, weights[base + 3].data); /* :435 */
    vitref_softmax(lg, probabilities, cfg->num_classes);                 /* :437 */
    if (logits) memcpy(logits, lg, sizeof(float) * cfg->num_classes);

    free(lg); free(conv); free(flat); free(tok); free(x); free(y);
}

void vitref_forward(const vitref_config *cfg, const float *const *images, int n,
                    const vitref_tensor *weights, float *const *prob) {
    for (int i = 0; i < n; ++i) vitref_forward_image(cfg, images[i], weights, prob[i], NULL, NULL);
}
