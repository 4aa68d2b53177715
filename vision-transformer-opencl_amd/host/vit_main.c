/*
 * host/vit_main.c -- command-line driver with the flow of the reference's Main.c:18-88, written against the
 * reference's OWN symbol names (initialize_opencl, load_image_data, load_weights, ViT_opencl, comparator-style
 * check, Release_opencl) as exported by libvit_mi355x.so:
 *   initialize -> load images -> load weights -> forward (timed) -> argmax -> result file -> comparator -> release
 * Differences: paths and counts come from argv (the reference hard-codes them, Main.c:22,30,40), all images of
 * the file are processed unless --count is given (the reference forces n = 1, Main.c:45-46), every written line is
 * compared (the reference checks one, comparator.c:8), a seeded synthetic model can stand in for the weight blobs the
 * reference's repository lacks, and the input file can be streamed in chunks (--chunk) with the next chunk read from
 * disk while the GPUs work on the current one.
 *
 *   vit_main [--images FILE] [--weights DIR] [--out FILE] [--answer FILE] [--count N] [--chunk N]
 *            [--synthetic SEED] [--repeat R] [--reference-argmax] [--cache FILE] [--devices LIST]
 *
 *   --cache FILE    opt-in packed weight cache (vit_io.h): used when it matches the weight directory, otherwise the
 *                   directory is loaded as usual and the cache (re)written from the complete set
 *   --devices LIST  "all" or "0,1,..." = VIT_HIP_DEVICES (one engine per GPU, images split across them)
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ViT_hip.h"
#include "vit_engine.h"
#include "vit_io.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    vit_image_reader *reader;
    int max_images, first;
    ImageData *out;
} prefetch_job;

static void *prefetch_thread(void *arg) {
    prefetch_job *j = (prefetch_job *)arg;
    j->out = vit_image_reader_next(j->reader, j->max_images, &j->first);
    return NULL;
}

int main(int argc, char **argv) {
    const char *img_path = "./Data/input-100.bin", *weight_dir = "./Network";
    const char *out_path = "./Data/opencl_result.txt", *answer_path = "./Data/answer_result.txt";
    const char *cache_path = NULL;
    int count = 0, repeat = 1, synthetic = 0, fix_argmax = 1, chunk = 0;
    unsigned long long seed = 1234;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--images") && i + 1 < argc) img_path = argv[++i];
        else if (!strcmp(argv[i], "--weights") && i + 1 < argc) weight_dir = argv[++i];
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
        else if (!strcmp(argv[i], "--answer") && i + 1 < argc) answer_path = argv[++i];
        else if (!strcmp(argv[i], "--count") && i + 1 < argc) count = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--chunk") && i + 1 < argc) chunk = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--synthetic") && i + 1 < argc) { synthetic = 1; seed = strtoull(argv[++i], NULL, 10); }
        else if (!strcmp(argv[i], "--reference-argmax")) fix_argmax = 0;
        else if (!strcmp(argv[i], "--cache") && i + 1 < argc) cache_path = argv[++i];
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) setenv("VIT_HIP_DEVICES", argv[++i], 1);
        else {
            fprintf(stderr, "usage: %s [--images F] [--weights D] [--out F] [--answer F] [--count N] [--chunk N] "
                            "[--synthetic SEED] [--repeat R] [--reference-argmax] [--cache F] [--devices LIST]\n", argv[0]);
            return 2;
        }
    }

    initialize_opencl(); /* Main.c:19 */
    const vit_config cfg = vit_config_b16();
    const int nw = VIT_WEIGHT_COUNT(cfg.depth);

    /* ---- weights (Main.c:29-30) ---- */
    Network *network = (Network *)calloc((size_t)nw, sizeof(Network));
    if (!network) return 1;
    int from_cache = 0;
    const double tw0 = now_s();
    if (synthetic) {
        if (vit_synth_weights(&cfg, seed, network, nw)) return 1;
    } else if (cache_path && ViT_hip_load_weight_cache(cache_path, weight_dir) == 0) {
        from_cache = 1; /* resident already: one read + one upload, no load_weights() */
    } else {
        load_weights(weight_dir, network, nw);
        if (cache_path) { /* (re)write the cache, from a complete set only */
            vit_weight_image img;
            if (vit_weight_image_build(&img, &cfg, network, nw, 1) == 0) {
                if (vit_weight_image_save(&img, cache_path, weight_dir) != 0) fprintf(stderr, "note: cannot write %s\n", cache_path);
                vit_weight_image_free(&img);
            } else {
                fprintf(stderr, "note: weight set incomplete, %s not written\n", cache_path);
            }
        }
    }
    printf("weights: %s, %.3f s\n", synthetic ? "synthetic" : (from_cache ? "packed cache" : "Weight_*.bin files"), now_s() - tw0);

    FILE *fp = fopen(out_path, "w");
    if (!fp) {
        printf("Error: cannot open %s for writing\n", out_path);
        return 1;
    }

    /* ---- images (Main.c:23) + forward (Main.c:57) + result lines (Main.c:62-72) ---- */
    int written = 0, pred_state = 0;
    printf("=====================Start========================\n");
    if (synthetic || chunk <= 0) {
        ImageData *images = synthetic ? vit_synth_images(&cfg, count > 0 ? count : 8, seed + 1) : load_image_data(img_path);
        if (!images) return 1;
        int n = images[0].n;
        if (count > 0 && count < n) n = count;
        float **prob = (float **)malloc(sizeof(float *) * (size_t)n);
        for (int i = 0; i < n; ++i) prob[i] = (float *)malloc(sizeof(float) * (size_t)cfg.num_classes);
        const int total = images[0].n;
        images[0].n = n; /* the forward reads the count from the first element (Main.c:46) */
        for (int r = 0; r < repeat; ++r) {
            const double t0 = now_s();
            ViT_opencl(images, from_cache ? NULL : network, prob);
            const double dt = now_s() - t0;
            printf("HIP time: %f sec (%d images, %.1f img/s on %d device%s%s)\n", dt, n, n / dt, ViT_hip_device_count(),
                   ViT_hip_device_count() == 1 ? "" : "s", (r == 0 && !from_cache) ? ", includes weight upload" : "");
        }
        vit_write_results_from(fp, prob, n, cfg.num_classes, fix_argmax, 0, &pred_state);
        written = n;
        images[0].n = total;
        for (int i = 0; i < n; ++i) free(prob[i]);
        free(prob);
        free_image_data(images);
    } else {
        /* streamed: chunk k+1 is read from disk by a helper thread while chunk k is on the GPUs */
        int total = 0;
        vit_image_reader *reader = vit_image_reader_open(img_path, &total, NULL, NULL, NULL);
        if (!reader) return 1;
        if (count > 0 && count < total) total = count;
        float **prob = (float **)malloc(sizeof(float *) * (size_t)chunk);
        for (int i = 0; i < chunk; ++i) prob[i] = (float *)malloc(sizeof(float) * (size_t)cfg.num_classes);
        int first = 0;
        ImageData *cur = vit_image_reader_next(reader, chunk < total ? chunk : total, &first);
        const double t0 = now_s();
        while (cur) {
            const int n = cur[0].n;
            prefetch_job job = {reader, 0, 0, NULL};
            pthread_t th;
            int have_thread = 0;
            const int left = total - (first + n);
            if (left > 0) {
                job.max_images = left < chunk ? left : chunk;
                have_thread = pthread_create(&th, NULL, prefetch_thread, &job) == 0;
            }
            ViT_opencl(cur, from_cache ? NULL : network, prob);
            vit_write_results_from(fp, prob, n, cfg.num_classes, fix_argmax, first, &pred_state);
            written += n;
            free_image_data(cur);
            cur = NULL;
            if (left > 0) {
                if (have_thread) pthread_join(th, NULL);
                else prefetch_thread(&job);
                cur = job.out;
                first = job.first;
            }
        }
        const double dt = now_s() - t0;
        printf("HIP time: %f sec (%d images in chunks of %d, %.1f img/s on %d device%s, file read overlapped)\n", dt, written,
               chunk, written / dt, ViT_hip_device_count(), ViT_hip_device_count() == 1 ? "" : "s");
        for (int i = 0; i < chunk; ++i) free(prob[i]);
        free(prob);
        vit_image_reader_close(reader);
    }
    fclose(fp);

    if (!synthetic) {
        const int cmp = vit_compare_results(out_path, answer_path, written, 0.01f); /* Main.c:75, every line */
        if (cmp == 0) printf("Comparator: the two files agree on all %d lines.\n", written);
        else printf("Comparator: %d differences between the two files.\n", cmp);
    }

    free_weights(network, nw);
    free(network);
    Release_opencl(); /* Main.c:86 */
    return 0;
}
