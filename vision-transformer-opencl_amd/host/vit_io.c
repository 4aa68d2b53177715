/*
 * host/vit_io.c -- image / weight loaders, result writer, comparator, synthetic tensors.
 *
 * Portable-C restatement of the reference's boundary code (formats in SURVEY.md Appendix A):
 *   load_image_data  <- Network.c:24-97      load_weights <- Network.c:99-194
 *   comparator       <- comparator.c:11-80   result lines <- Main.c:62-72
 * Behavioural differences, all deliberate hardening: an unopenable weight file
 * is skipped (the reference dereferences a NULL FILE*, Network.c:152-158); the image header is
 * range-checked; everything that is malloc'd can be freed.
 */
#include "vit_io.h"

#include <dirent.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- images -------------------------------------------------------------------------------- */

struct vit_image_reader {
    FILE *f;
    int n, c, h, w;
    int next; /* index of the first image not handed out yet */
};

vit_image_reader *vit_image_reader_open(const char *filename, int *n, int *c, int *h, int *w) {
    FILE *f = fopen(filename, "rb");
    if (!f) {
        perror("load_image_data: cannot open image file");
        return NULL;
    }
    int32_t hdr[4];
    if (fread(hdr, sizeof(int32_t), 4, f) != 4) {
        perror("load_image_data: cannot read the n,c,h,w header");
        fclose(f);
        return NULL;
    }
    if (hdr[0] <= 0 || hdr[1] <= 0 || hdr[2] <= 0 || hdr[3] <= 0 || (double)hdr[1] * hdr[2] * hdr[3] > 1e9) {
        fprintf(stderr, "load_image_data: implausible header n=%d c=%d h=%d w=%d\n", hdr[0], hdr[1], hdr[2], hdr[3]);
        fclose(f);
        return NULL;
    }
    vit_image_reader *r = (vit_image_reader *)calloc(1, sizeof(*r));
    if (!r) {
        perror("load_image_data: out of memory");
        fclose(f);
        return NULL;
    }
    r->f = f; r->n = hdr[0]; r->c = hdr[1]; r->h = hdr[2]; r->w = hdr[3];
    if (n) *n = r->n;
    if (c) *c = r->c;
    if (h) *h = r->h;
    if (w) *w = r->w;
    return r;
}

ImageData *vit_image_reader_next(vit_image_reader *r, int max_images, int *first) {
    if (!r || max_images <= 0 || r->next >= r->n) return NULL;
    const int k = r->n - r->next < max_images ? r->n - r->next : max_images;
    const size_t per = (size_t)r->c * r->h * r->w;
    ImageData *images = (ImageData *)calloc((size_t)k, sizeof(ImageData));
    if (!images) {
        perror("load_image_data: out of memory");
        return NULL;
    }
    for (int i = 0; i < k; ++i) {
        images[i].n = k; /* every element carries the count of its array (Network.c:77) */
        images[i].c = r->c;
        images[i].h = r->h;
        images[i].w = r->w;
        images[i].data = (float *)malloc(per * sizeof(float));
        if (!images[i].data || fread(images[i].data, sizeof(float), per, r->f) != per) {
            perror("load_image_data: short read or out of memory");
            free_image_data(images);
            return NULL;
        }
    }
    if (first) *first = r->next;
    r->next += k;
    return images;
}

void vit_image_reader_close(vit_image_reader *r) {
    if (!r) return;
    fclose(r->f);
    free(r);
}

ImageData *load_image_data(const char *filename) { /* the whole file as one array (Network.c:24-97) */
    int n = 0;
    vit_image_reader *r = vit_image_reader_open(filename, &n, NULL, NULL, NULL);
    if (!r) return NULL;
    ImageData *images = vit_image_reader_next(r, n, NULL);
    vit_image_reader_close(r);
    return images;
}

void free_image_data(ImageData *images) {
    if (!images) return;
    const int n = images[0].n;
    for (int i = 0; i < n; ++i) free(images[i].data);
    free(images);
}

/* ---- weights ------------------------------------------------------------------------------- */

void vit_round_weights(float *data, size_t count) {
    /* element-wise: the 86.6 M (ViT-B) / 304.7 M (ViT-L) roundf calls of a load are spread over the host cores */
#pragma omp parallel for schedule(static) if (count > (1u << 16))
    for (size_t i = 0; i < count; ++i) data[i] = roundf(data[i] * 1000000.0f) / 1000000.0f;
}

/* "Weight_<idx>_..." -> idx, or -1 (Network.c:99-117) */
static int weight_index(const char *name) {
    if (strncmp(name, "Weight_", 7) != 0) return -1;
    const char *p = name + 7;
    const char *us = strchr(p, '_');
    if (!us) return -1;
    char digits[16] = {0};
    size_t len = (size_t)(us - p);
    if (len >= sizeof(digits)) len = sizeof(digits) - 1;
    memcpy(digits, p, len);
    return atoi(digits);
}

void load_weights(const char *directory, Network network[], int count) {
    DIR *dir = opendir(directory);
    if (!dir) {
        perror("load_weights: cannot open the weight directory");
        exit(EXIT_FAILURE);
    }
    for (int i = 0; i < count; ++i) {
        network[i].data = NULL;
        network[i].size = 0;
    }
    struct dirent *ent;
    while ((ent = readdir(dir)) != NULL) {
        const char *ext = strrchr(ent->d_name, '.');
        if (strncmp(ent->d_name, "Weight_", 7) != 0 || !ext || strcmp(ext, ".bin") != 0) continue;
        const int idx = weight_index(ent->d_name);
        if (idx < 0 || idx >= count) continue;

        char path[1024];
        snprintf(path, sizeof(path), "%s/%s", directory, ent->d_name);
        FILE *fp = fopen(path, "rb");
        if (!fp) continue;
        if (fseek(fp, 0, SEEK_END) != 0) { fclose(fp); continue; }
        const long bytes = ftell(fp);
        rewind(fp);
        if (bytes < 0) { fclose(fp); continue; }
        const size_t nfloat = (size_t)bytes / sizeof(float);
        float *buf = (float *)malloc(nfloat ? nfloat * sizeof(float) : sizeof(float));
        if (!buf) {
            perror("load_weights: out of memory");
            fclose(fp);
            exit(EXIT_FAILURE);
        }
        if (fread(buf, sizeof(float), nfloat, fp) != nfloat) {
            perror("load_weights: short read");
            free(buf);
            fclose(fp);
            continue;
        }
        fclose(fp);
        vit_round_weights(buf, nfloat); /* Network.c:184-187 */
        free(network[idx].data);        /* two files with one index: last one wins, no leak */
        network[idx].data = buf;
        network[idx].size = nfloat;
    }
    closedir(dir);
}

void free_weights(Network network[], int count) {
    for (int i = 0; i < count; ++i) {
        free(network[i].data);
        network[i].data = NULL;
        network[i].size = 0;
    }
}

/* ---- results ------------------------------------------------------------------------------- */

int vit_argmax(const float *probs, int classes) {
    int best = 0;
    for (int j = 1; j < classes; ++j)
        if (probs[j] > probs[best]) best = j;
    return best;
}

int vit_write_results_from(FILE *fp, float *const *probs, int n, int classes, int fix_argmax, int first_index, int *pred_io) {
    int pred = pred_io ? *pred_io : 0; /* Main.c:62 declares it once, outside the image loop */
    for (int i = 0; i < n; ++i) {
        if (fix_argmax) pred = 0;
        for (int j = 1; j < classes; ++j)
            if (probs[i][j] > probs[i][pred]) pred = j;
        if (fprintf(fp, "[%d] label: %d / prob: %.6f\n", first_index + i, pred, probs[i][pred]) < 0) return -1;
    }
    if (pred_io) *pred_io = pred;
    return 0;
}

int vit_write_results(FILE *fp, float *const *probs, int n, int classes, int fix_argmax) {
    return vit_write_results_from(fp, probs, n, classes, fix_argmax, 0, NULL);
}

int vit_write_results_file(const char *path, float *const *probs, int n, int classes, int fix_argmax) {
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    const int rc = vit_write_results(fp, probs, n, classes, fix_argmax);
    return fclose(fp) == 0 ? rc : -1;
}

/* ---- comparator ---------------------------------------------------------------------------- */

static int parse_result_line(const char *line, int *label, float *prob) {
    return sscanf(line, "[%*d] label: %d / prob: %f)", label, prob); /* comparator.c:11-14 */
}

int vit_compare_results(const char *result_path, const char *answer_path, int lines, float tol) {
    FILE *fr = fopen(result_path, "r");
    if (!fr) {
        fprintf(stderr, "Error: Cannot open %s\n", result_path);
        return 1;
    }
    FILE *fa = fopen(answer_path, "r");
    if (!fa) {
        fprintf(stderr, "Error: Cannot open %s\n", answer_path);
        fclose(fr);
        return 1;
    }
    char lr[1024], la[1024];
    int errors = 0;
    for (int ln = 0; ln < lines; ++ln) {
        if (!fgets(lr, sizeof(lr), fr) || !fgets(la, sizeof(la), fa)) {
            fprintf(stderr, "Line %d: the two files do not have the same number of lines.\n", ln);
            errors++;
            break;
        }
        int label_r, label_a;
        float prob_r, prob_a;
        if (parse_result_line(lr, &label_r, &prob_r) != 2 || parse_result_line(la, &label_a, &prob_a) != 2) {
            fprintf(stderr, "Line %d: parse error\n", ln);
            errors++;
            continue;
        }
        if (label_r != label_a) {
            fprintf(stderr, "Line %d: Label mismatch (Result: %d, Answer: %d)\n", ln, label_r, label_a);
            errors++;
        }
        if (fabs(prob_r - prob_a) > tol) {
            fprintf(stderr, "Line %d: Probability mismatch (Result: %.6f, Answer: %.6f)\n", ln, prob_r, prob_a);
            errors++;
        }
    }
    fclose(fr);
    fclose(fa);
    return errors;
}

int comparator(void) {
    return vit_compare_results("./Data/opencl_result.txt", "./Data/answer_result.txt", 1, 0.01f);
}

/* ---- synthetic tensors --------------------------------------------------------------------- */

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void vit_synth_uniform(unsigned long long seed, int index, size_t n, float lo, float hi, float *out) {
    const uint64_t gamma = 0x9E3779B97F4A7C15ULL;
    const uint64_t base = mix64(((uint64_t)seed * 0x100000001B3ULL + (uint64_t)index + 1) * gamma);
    const float span = hi - lo;
    /* counter-based: element i depends on (stream, i) alone, so the cores split the range */
#pragma omp parallel for schedule(static) if (n > (1u << 16))
    for (size_t i = 0; i < n; ++i) {
        const uint64_t z = mix64(base + (uint64_t)(i + 1) * gamma);
        const float u = (float)(z >> 40) * 0x1p-24f;
        out[i] = lo + span * u;
    }
}

/* Tensor kinds and ranges: keep in step with _kind/_RANGE of synth.py. */
static void synth_range(const vit_config *cfg, int idx, float *lo, float *hi) {
    static const float lin = 0.035f, qkv = 0.07f, bias = 0.035f, lnb = 0.05f;
    const int base = 4 + VIT_WEIGHTS_PER_LAYER * cfg->depth;
    float a;
    int ln_w = 0;
    if (idx < 4) {
        const float first[4] = {0.05f, 0.035f, bias, 0.087f}; /* cls, conv_w, conv_b, pos */
        a = first[idx];
    } else if (idx >= base) {
        const float last[4] = {0.f, lnb, 0.28f, bias};         /* ln_w, ln_b, head_w, head_b */
        a = last[idx - base];
        ln_w = (idx == base);
    } else {
        const float layer[12] = {0.f, lnb, qkv, bias, lin, bias, 0.f, lnb, lin, bias, lin, bias};
        const int k = (idx - 4) % VIT_WEIGHTS_PER_LAYER;
        a = layer[k];
        ln_w = (k == 0 || k == 6);
    }
    if (ln_w) { *lo = 0.5f; *hi = 1.0f; } else { *lo = -a; *hi = a; }
}

static size_t synth_weight_size(const vit_config *cfg, int idx) {
    const size_t D = (size_t)cfg->embed_dim, H = (size_t)cfg->hidden_dim;
    const size_t G = (size_t)(cfg->img_size / cfg->patch_size), T = G * G + 1;
    const size_t PK = (size_t)cfg->in_chans * cfg->patch_size * cfg->patch_size;
    const int base = 4 + VIT_WEIGHTS_PER_LAYER * cfg->depth;
    if (idx < 4) { const size_t s[4] = {D, D * PK, D, T * D}; return s[idx]; }
    if (idx >= base) { const size_t s[4] = {D, D, (size_t)cfg->num_classes * D, (size_t)cfg->num_classes}; return s[idx - base]; }
    { const size_t s[12] = {D, D, 3 * D * D, 3 * D, D * D, D, D, D, H * D, H, D * H, D}; return s[(idx - 4) % VIT_WEIGHTS_PER_LAYER]; }
}

int vit_synth_weights(const vit_config *cfg, unsigned long long seed, Network network[], int count) {
    for (int i = 0; i < count; ++i) { network[i].data = NULL; network[i].size = 0; }
    for (int i = 0; i < count && i < VIT_WEIGHT_COUNT(cfg->depth); ++i) {
        const size_t n = synth_weight_size(cfg, i);
        float lo, hi;
        synth_range(cfg, i, &lo, &hi);
        float *buf = (float *)malloc(n * sizeof(float));
        if (!buf) { free_weights(network, count); return -1; }
        vit_synth_uniform(seed, i, n, lo, hi, buf);
        vit_round_weights(buf, n);
        network[i].data = buf;
        network[i].size = n;
    }
    return 0;
}

ImageData *vit_synth_images(const vit_config *cfg, int n, unsigned long long seed) {
    if (n <= 0) return NULL;
    const size_t per = (size_t)cfg->in_chans * cfg->img_size * cfg->img_size;
    ImageData *images = (ImageData *)calloc((size_t)n, sizeof(ImageData));
    if (!images) return NULL;
    for (int i = 0; i < n; ++i) {
        images[i].n = n; images[i].c = cfg->in_chans; images[i].h = cfg->img_size; images[i].w = cfg->img_size;
        images[i].data = (float *)malloc(per * sizeof(float));
        if (!images[i].data) { free_image_data(images); return NULL; }
        vit_synth_uniform(seed, 1000000 + i, per, -2.1f, 2.6f, images[i].data);
    }
    return images;
}
