/*
 * host/vit_wimage.c -- the device-layout weight image and its cache file (include/vit_io.h).
 *
 * What it replaces: the reference reads 152 Weight_*.bin files and rounds 86.6 M values on every start
 * (Network.c:119-194), then re-uploads tensors op by op (ViT_opencl.c:136,630-631).  Here the tensors are packed once
 * into the exact bytes the engine keeps in HBM -- one fp32 section with every tensor at a 512-byte aligned offset,
 * GEMM operands first, and their bf16 copies right behind -- so a start is one read and one host-to-device copy
 * (and, with several devices, one peer copy per further device: vit_engine_copy_weights).
 */
#include "vit_io.h"

#include <dirent.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#define VITW_MAGIC 0x57544956u /* "VITW" little endian */
#define VITW_VERSION 2u
#define SLOT 128u              /* floats: 512-B fp32 slots = 256-B bf16 slots */
#define PAYLOAD_ALIGN 4096u

/* ---- layout -------------------------------------------------------------------------------- */

static size_t expect_size(const vit_config *cfg, int idx) {
    const size_t D = (size_t)cfg->embed_dim, H = (size_t)cfg->hidden_dim;
    const size_t G = (size_t)(cfg->img_size / cfg->patch_size), T = G * G + 1;
    const size_t PK = (size_t)cfg->in_chans * cfg->patch_size * cfg->patch_size;
    const int base = 4 + VIT_WEIGHTS_PER_LAYER * cfg->depth;
    if (idx < 4) { const size_t s[4] = {D, D * PK, D, T * D}; return s[idx]; }
    if (idx >= base) { const size_t s[4] = {D, D, (size_t)cfg->num_classes * D, (size_t)cfg->num_classes}; return s[idx - base]; }
    { const size_t s[12] = {D, D, 3 * D * D, 3 * D, D * D, D, D, D, H * D, H, D * H, D}; return s[(idx - 4) % VIT_WEIGHTS_PER_LAYER]; }
}

static int is_gemm_operand(const vit_config *cfg, int idx) {
    const int base = 4 + VIT_WEIGHTS_PER_LAYER * cfg->depth;
    if (idx == 1) return 1; /* conv_proj weight: patch embedding as a GEMM */
    if (idx < 4 || idx >= base) return 0;
    const int k = (idx - 4) % VIT_WEIGHTS_PER_LAYER;
    return k == 2 || k == 4 || k == 8 || k == 10; /* in_proj, out_proj, fc1, fc2 weights */
}

size_t vit_weight_layout(const vit_config *cfg, size_t *off, size_t *size, size_t *gemm_floats) {
    const int count = VIT_WEIGHT_COUNT(cfg->depth);
    size_t at = 0;
    for (int pass = 0; pass < 2; ++pass) { /* GEMM operands first, then everything that stays fp32 only */
        for (int i = 0; i < count; ++i) {
            if (is_gemm_operand(cfg, i) != (pass == 0)) continue;
            const size_t n = expect_size(cfg, i);
            if (off) off[i] = at;
            if (size) size[i] = n;
            at += (n + SLOT - 1) / SLOT * SLOT;
        }
        if (pass == 0 && gemm_floats) *gemm_floats = at;
    }
    return at;
}

/* ---- image --------------------------------------------------------------------------------- */

static uint16_t bf16_rne(float f) { /* what v_cvt_pk_bf16_f32 does for finite values: round to nearest, ties to even */
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN stays NaN (quiet) */
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static int image_alloc(vit_weight_image *img, const vit_config *cfg, int with_bf16) {
    memset(img, 0, sizeof(*img));
    img->cfg = *cfg;
    img->count = VIT_WEIGHT_COUNT(cfg->depth);
    img->off = (size_t *)calloc((size_t)img->count, sizeof(size_t));
    img->size = (size_t *)calloc((size_t)img->count, sizeof(size_t));
    if (!img->off || !img->size) { vit_weight_image_free(img); return -1; }
    img->f32_floats = vit_weight_layout(cfg, img->off, img->size, &img->gemm_floats);
    img->bf16_elems = with_bf16 ? img->gemm_floats : 0;
    const size_t bytes = img->f32_floats * sizeof(float) + img->bf16_elems * sizeof(uint16_t);
    void *p = NULL;
    if (posix_memalign(&p, PAYLOAD_ALIGN, (bytes + PAYLOAD_ALIGN - 1) / PAYLOAD_ALIGN * PAYLOAD_ALIGN)) {
        vit_weight_image_free(img);
        return -1;
    }
    img->f32 = (float *)p;
    img->bf16 = with_bf16 ? (unsigned short *)(img->f32 + img->f32_floats) : NULL;
    return 0;
}

void vit_weight_image_free(vit_weight_image *img) {
    if (!img) return;
    free(img->off);
    free(img->size);
    free(img->f32);
    memset(img, 0, sizeof(*img));
}

int vit_weight_image_build(vit_weight_image *img, const vit_config *cfg, const Network network[], int count, int with_bf16) {
    if (!img || !cfg || !network || count != VIT_WEIGHT_COUNT(cfg->depth)) return -1;
    for (int i = 0; i < count; ++i)
        if (!network[i].data || network[i].size != expect_size(cfg, i)) return -1;
    if (image_alloc(img, cfg, with_bf16)) return -1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(8)
    for (int i = 0; i < count; ++i) {
        float *dst = img->f32 + img->off[i];
        const size_t n = img->size[i], slot = (n + SLOT - 1) / SLOT * SLOT;
        memcpy(dst, network[i].data, n * sizeof(float));
        memset(dst + n, 0, (slot - n) * sizeof(float)); /* slot padding is read by the GEMM's row padding: keep it zero */
    }
    if (with_bf16) {
        const size_t g = img->gemm_floats;
#pragma omp parallel for schedule(static) num_threads(8)
        for (size_t i = 0; i < g; ++i) img->bf16[i] = bf16_rne(img->f32[i]);
    }
    return 0;
}

void vit_weight_image_view(const vit_weight_image *img, Network network[], int count) {
    for (int i = 0; i < count; ++i) {
        network[i].data = i < img->count ? img->f32 + img->off[i] : NULL;
        network[i].size = i < img->count ? img->size[i] : 0;
    }
}

/* ---- source fingerprints ------------------------------------------------------------------- */

typedef struct {
    char name[104];
    uint64_t bytes;
    int64_t mtime_ns;
    uint32_t index, pad;
} vitw_source;

static int source_cmp(const void *a, const void *b) { return strcmp(((const vitw_source *)a)->name, ((const vitw_source *)b)->name); }

static int weight_file_index(const char *name) {
    if (strncmp(name, "Weight_", 7) != 0) return -1;
    const char *us = strchr(name + 7, '_');
    const char *ext = strrchr(name, '.');
    if (!us || !ext || strcmp(ext, ".bin") != 0 || us == name + 7) return -1;
    return atoi(name + 7);
}

/* every Weight_<idx>_*.bin of `dir` with idx < count, sorted by name; returns the number found or -1 */
static int scan_sources(const char *dir, int count, vitw_source **out) {
    *out = NULL;
    DIR *d = opendir(dir);
    if (!d) return -1;
    int n = 0, cap = 0;
    vitw_source *v = NULL;
    struct dirent *ent;
    while ((ent = readdir(d)) != NULL) {
        const int idx = weight_file_index(ent->d_name);
        if (idx < 0 || idx >= count || strlen(ent->d_name) >= sizeof(v->name)) continue;
        char path[1200];
        struct stat st;
        snprintf(path, sizeof(path), "%s/%s", dir, ent->d_name);
        if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) continue;
        if (n == cap) {
            cap = cap ? 2 * cap : 256;
            vitw_source *nv = (vitw_source *)realloc(v, (size_t)cap * sizeof(*v));
            if (!nv) { free(v); closedir(d); return -1; }
            v = nv;
        }
        memset(&v[n], 0, sizeof(v[n]));
        strcpy(v[n].name, ent->d_name);
        v[n].bytes = (uint64_t)st.st_size;
        v[n].mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
        v[n].index = (uint32_t)idx;
        n++;
    }
    closedir(d);
    if (n) qsort(v, (size_t)n, sizeof(*v), source_cmp);
    *out = v;
    return n;
}

/* ---- file ---------------------------------------------------------------------------------- */

typedef struct {
    uint32_t magic, version, count, n_sources;
    int32_t cfg[8];
    uint64_t f32_floats, gemm_floats, bf16_elems, payload_offset;
} vitw_header;

int vit_weight_image_save(const vit_weight_image *img, const char *path, const char *source_dir) {
    if (!img || !img->f32 || !path) return -1;
    vitw_source *src = NULL;
    int ns = 0;
    if (source_dir && (ns = scan_sources(source_dir, img->count, &src)) < 0) return -1;
    vitw_header h;
    memset(&h, 0, sizeof(h));
    h.magic = VITW_MAGIC; h.version = VITW_VERSION; h.count = (uint32_t)img->count; h.n_sources = (uint32_t)ns;
    memcpy(h.cfg, &img->cfg, sizeof(h.cfg));
    h.f32_floats = img->f32_floats; h.gemm_floats = img->gemm_floats; h.bf16_elems = img->bf16_elems;
    const size_t meta = sizeof(h) + (size_t)ns * sizeof(vitw_source);
    h.payload_offset = (meta + PAYLOAD_ALIGN - 1) / PAYLOAD_ALIGN * PAYLOAD_ALIGN;
    char tmp[1100];
    snprintf(tmp, sizeof(tmp), "%s.tmp", path); /* written beside, renamed when complete: readers never see a torn file */
    FILE *fp = fopen(tmp, "wb");
    if (!fp) { free(src); return -1; }
    int ok = fwrite(&h, sizeof(h), 1, fp) == 1;
    if (ok && ns) ok = fwrite(src, sizeof(vitw_source), (size_t)ns, fp) == (size_t)ns;
    free(src);
    for (size_t i = meta; ok && i < h.payload_offset; ++i) ok = fputc(0, fp) != EOF;
    const size_t bytes = img->f32_floats * sizeof(float) + img->bf16_elems * sizeof(uint16_t);
    if (ok) ok = fwrite(img->f32, 1, bytes, fp) == bytes;
    if (fclose(fp) != 0) ok = 0;
    if (ok && rename(tmp, path) != 0) ok = 0;
    if (!ok) remove(tmp);
    return ok ? 0 : -1;
}

int vit_weight_image_load(vit_weight_image *img, const char *path, const vit_config *cfg, const char *source_dir) {
    if (!img || !path || !cfg) return -1;
    memset(img, 0, sizeof(*img));
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    vitw_header h;
    vitw_source *stored = NULL, *now = NULL;
    int ok = fread(&h, sizeof(h), 1, fp) == 1 && h.magic == VITW_MAGIC && h.version == VITW_VERSION &&
             h.count == (uint32_t)VIT_WEIGHT_COUNT(cfg->depth) && memcmp(h.cfg, cfg, sizeof(h.cfg)) == 0 &&
             h.n_sources <= 65536;
    if (ok && h.n_sources) {
        stored = (vitw_source *)malloc((size_t)h.n_sources * sizeof(vitw_source));
        ok = stored && fread(stored, sizeof(vitw_source), h.n_sources, fp) == h.n_sources;
    }
    if (ok && source_dir) { /* the directory must still hold exactly the files the image was built from */
        const int n = scan_sources(source_dir, (int)h.count, &now);
        ok = n >= 0 && (uint32_t)n == h.n_sources && (n == 0 || memcmp(now, stored, (size_t)n * sizeof(vitw_source)) == 0);
    }
    free(stored);
    free(now);
    if (ok) ok = image_alloc(img, cfg, h.bf16_elems != 0) == 0;
    if (ok) ok = img->f32_floats == h.f32_floats && img->gemm_floats == h.gemm_floats && img->bf16_elems == h.bf16_elems;
    if (ok) {
        const size_t bytes = img->f32_floats * sizeof(float) + img->bf16_elems * sizeof(uint16_t);
        ok = fseek(fp, (long)h.payload_offset, SEEK_SET) == 0 && fread(img->f32, 1, bytes, fp) == bytes;
    }
    fclose(fp);
    if (!ok) vit_weight_image_free(img);
    return ok ? 0 : -1;
}

/* ---- load_weights() through the cache ------------------------------------------------------- */

void load_weights_cached(const vit_config *cfg, const char *directory, Network network[], int count, const char *cache_path) {
    vit_config b16 = {224, 16, 3, 1000, 768, 12, 12, 3072};
    if (!cfg) cfg = &b16;
    char def[1100];
    if (!cache_path) {
        snprintf(def, sizeof(def), "%s/vit_weights.cache", directory);
        cache_path = def;
    }
    vit_weight_image img;
    if (count == VIT_WEIGHT_COUNT(cfg->depth) && vit_weight_image_load(&img, cache_path, cfg, directory) == 0) {
        int ok = 1;
        for (int i = 0; i < count; ++i) { network[i].data = NULL; network[i].size = 0; }
        for (int i = 0; ok && i < count; ++i) {
            network[i].data = (float *)malloc(img.size[i] * sizeof(float));
            if (!network[i].data) { ok = 0; break; }
            memcpy(network[i].data, img.f32 + img.off[i], img.size[i] * sizeof(float));
            network[i].size = img.size[i];
        }
        vit_weight_image_free(&img);
        if (ok) return;
        free_weights(network, count);
    }
    load_weights(directory, network, count);
    /* only a COMPLETE set is cached (the reference tree ships 116 of its 152 blobs: caching that would pin the holes) */
    if (count == VIT_WEIGHT_COUNT(cfg->depth) && vit_weight_image_build(&img, cfg, network, count, 1) == 0) {
        (void)vit_weight_image_save(&img, cache_path, directory); /* best effort: a read-only location just stays uncached */
        vit_weight_image_free(&img);
    }
}
