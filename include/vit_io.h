/*
 * include/vit_io.h -- the data formats either side of the forward pass.
 *
 * Same names, signatures, formats and ownership as the reference's loaders and comparator so a
 * Main.c-shaped caller links unchanged; portable C instead of the MSVC CRT / Win32 dirent.
 *
 *   load_image_data   Network.h:15, Network.c:24-97
 *   load_weights      Network.h:34, Network.c:119-194   (includes the 1e-6 rounding, :184-187)
 *   comparator        comparator.h:5, comparator.c:23-80
 *
 * Additions (not in the reference): free helpers, a comparator with explicit paths / line count /
 * tolerance (the reference hard-codes ./Data/... and IMAGE_COUNT 1, comparator.c:8,26-27), the
 * result-file writer of Main.c:62-72, and the seeded synthetic-tensor generator used by the
 * tests and benchmarks.
 */
#ifndef VIT_IO_H
#define VIT_IO_H

#include <stdio.h>

#include "vit_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* File = int32 n,c,h,w + n*c*h*w little-endian fp32, NCHW.  Returns an array of n ImageData,
 * each owning its own CHW copy, or NULL after perror() on any failure. */
ImageData *load_image_data(const char *filename);
void free_image_data(ImageData *images);
/*
 * Chunked reading of the same file: load_image_data() materialises all n images (the
 * reference's input-100.bin is 60 MB, a production file need not fit in memory).  A reader keeps the file open and hands
 * out consecutive ImageData arrays of at most max_images images; each array is shaped exactly like a
 * load_image_data() result (every element's .n = the length of THAT array), so it can be passed to ViT_opencl() and
 * freed with free_image_data() as it is.  vit_image_reader_next() returns NULL at the end of the file or (after
 * perror()) on a short read; *first receives the file-wide index of the array's first image.
 */
typedef struct vit_image_reader vit_image_reader;
vit_image_reader *vit_image_reader_open(const char *filename, int *n, int *c, int *h, int *w);
ImageData *vit_image_reader_next(vit_image_reader *reader, int max_images, int *first);
void vit_image_reader_close(vit_image_reader *reader);

/* Scans `directory` for Weight_<idx>_*.bin (raw LE fp32), rounds every value to 1e-6 with
 * roundf(), stores network[idx] = {data,size}; entries without a file stay {NULL,0}.
 * exit(EXIT_FAILURE) after perror() when the directory cannot be opened or memory runs out. */
void load_weights(const char *directory, Network network[], int count);
void free_weights(Network network[], int count);
/*
 * Device-layout weight image = the packed weight cache.
 *
 * load_weights() opens 152 files and rounds 86.6 M values on every start, and the engine then needs the tensors at
 * 256-byte aligned offsets of ONE device allocation, the GEMM operands also as bf16.  A vit_weight_image is exactly
 * those device bytes on the host, in one allocation: an fp32 section holding every tensor (already rounded as
 * Network.c:184-187 does) and, directly behind it, a bf16 section with the GEMM operands (conv_proj, in_proj,
 * out_proj, fc1, fc2 weights: the leading `gemm_floats` floats of the fp32 section, converted round-to-nearest-even,
 * same relative offsets).  Uploading it is one host-to-device copy (vit_engine_load_weight_image); written to a
 * file ("VITW" version 2) it is read back with one read.
 *
 * The file also records the Weight_*.bin files it was built from (name, byte size, mtime); vit_weight_image_load()
 * given a source directory rejects the file when that directory no longer matches (a blob added, replaced or
 * removed), so a stale cache is never used.  A cache is only written from a complete set of tensors.
 */
typedef struct {
    vit_config cfg;
    int count;               /* VIT_WEIGHT_COUNT(cfg.depth) */
    size_t f32_floats;       /* floats in the fp32 section */
    size_t gemm_floats;      /* leading floats of the fp32 section that are bf16 GEMM operands */
    size_t bf16_elems;       /* 0 (section absent) or gemm_floats */
    size_t *off;             /* [count] float offset of tensor i inside the fp32 section (multiple of 128) */
    size_t *size;            /* [count] element count of tensor i */
    float *f32;              /* the payload: f32_floats floats ... */
    unsigned short *bf16;    /* ... followed by bf16_elems bf16 values (same allocation), or NULL */
} vit_weight_image;

/* Offsets of the device layout for `cfg` (off[count], sizes[count] caller-provided); returns the fp32 section size in
 * floats and stores the size of the leading GEMM-operand region in *gemm_floats. */
size_t vit_weight_layout(const vit_config *cfg, size_t *off, size_t *size, size_t *gemm_floats);
/* Build from loaded tensors (all `count` present and of the expected size, else -1 with a message on stderr when
 * `verbose`).  with_bf16: also fill the bf16 section. */
int vit_weight_image_build(vit_weight_image *img, const vit_config *cfg, const Network network[], int count, int with_bf16);
void vit_weight_image_free(vit_weight_image *img);
/* network[i] = {pointer INTO the image, size}: a borrowed view, never pass it to free_weights(). */
void vit_weight_image_view(const vit_weight_image *img, Network network[], int count);
/* Write / read the cache file.  source_dir may be NULL (no fingerprints written / no staleness check).
 * Both return 0, or -1 on any I/O, format, configuration or staleness mismatch (nothing is left allocated). */
int vit_weight_image_save(const vit_weight_image *img, const char *path, const char *source_dir);
int vit_weight_image_load(vit_weight_image *img, const char *path, const vit_config *cfg, const char *source_dir);
/* load_weights() through a cache file: when `cache_path` holds an image for `cfg` (NULL = ViT-B/16, the reference's
 * model) that matches the current contents of `directory`, every
 * network[i] is filled from it (separately malloc'd, so free_weights() applies); otherwise the directory is scanned as
 * load_weights() does and -- only when every tensor was found -- the cache is (re)written, best effort.
 * cache_path == NULL means <directory>/vit_weights.cache. */
void load_weights_cached(const vit_config *cfg, const char *directory, Network network[], int count, const char *cache_path);
/* The loader's rounding on its own (Network.c:184-187). */
void vit_round_weights(float *data, size_t count);

/* "[%d] label: %d / prob: %.6f\n" (Main.c:71).  With fix_argmax == 0 the running maximum is NOT
 * reset between images, exactly as Main.c:62-70 behaves; fix_argmax == 1 resets it per image. */
int vit_argmax(const float *probs, int classes);
int vit_write_results(FILE *fp, float *const *probs, int n, int classes, int fix_argmax);
/* The same for a slice of a longer run: lines are numbered from first_index, and *pred_io carries the running
 * arg-max index of the fix_argmax == 0 behaviour from one slice to the next (start it at 0). */
int vit_write_results_from(FILE *fp, float *const *probs, int n, int classes, int fix_argmax, int first_index, int *pred_io);
/* Same, creating/truncating `path`; returns 0, or -1 if the file cannot be written. */
int vit_write_results_file(const char *path, float *const *probs, int n, int classes, int fix_argmax);

/* Compare the first `lines` lines of two result files: label must match and |dprob| <= tol.
 * Returns the number of differences (comparator.c semantics: unreadable file -> 1). */
int vit_compare_results(const char *result_path, const char *answer_path, int lines, float tol);
/* The reference's entry: ./Data/opencl_result.txt vs ./Data/answer_result.txt, 1 line, tol 0.01. */
int comparator(void);

/* Counter-based splitmix64 uniform fill: out[i] = lo + (hi-lo) * u24(seed, index, i). */
void vit_synth_uniform(unsigned long long seed, int index, size_t n, float lo, float hi, float *out);
/* A whole synthetic model / image set with the scales of SURVEY.md 8d (same bytes as
 * synth.make_weights / synth.make_images in the Python package), rounded like load_weights().
 * Returns 0, or -1 when out of memory.  Free with free_weights() / free_image_data(). */
int vit_synth_weights(const vit_config *cfg, unsigned long long seed, Network network[], int count);
ImageData *vit_synth_images(const vit_config *cfg, int n, unsigned long long seed);

#ifdef __cplusplus
}
#endif
#endif /* VIT_IO_H */
