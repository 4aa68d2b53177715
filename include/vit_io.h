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

/* Scans `directory` for Weight_<idx>_*.bin (raw LE fp32), rounds every value to 1e-6 with
 * roundf(), stores network[idx] = {data,size}; entries without a file stay {NULL,0}.
 * exit(EXIT_FAILURE) after perror() when the directory cannot be opened or memory runs out. */
void load_weights(const char *directory, Network network[], int count);
void free_weights(Network network[], int count);
/*
 * Packed weight cache: load_weights() opens 152 files and rounds 86.6 M values on
 * every start.  vit_save_weight_cache() writes the already-rounded tensors of `network` into ONE file
 * (header: magic "VITW", version, count, per-tensor element counts; then the raw fp32 data);
 * vit_load_weight_cache() restores {data,size} for every entry from it with a single read.
 * Both return 0 on success, -1 on any I/O or format error (the caller then falls back to load_weights()).
 */
int vit_save_weight_cache(const char *path, const Network network[], int count);
int vit_load_weight_cache(const char *path, Network network[], int count);
/* load_weights() through the cache: uses <directory>/vit_weights.cache when present and consistent with
 * `count`, otherwise scans the directory as load_weights() does and (best effort) writes the cache. */
void load_weights_cached(const char *directory, Network network[], int count);
/* The loader's rounding on its own (Network.c:184-187). */
void vit_round_weights(float *data, size_t count);

/* "[%d] label: %d / prob: %.6f\n" (Main.c:71).  With fix_argmax == 0 the running maximum is NOT
 * reset between images, exactly as Main.c:62-70 behaves; fix_argmax == 1 resets it per image. */
int vit_argmax(const float *probs, int classes);
int vit_write_results(FILE *fp, float *const *probs, int n, int classes, int fix_argmax);
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
