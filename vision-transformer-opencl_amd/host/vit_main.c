/*
 * host/vit_main.c -- command-line driver with the flow of the reference's Main.c:18-88:
 *   initialize -> load images -> load weights -> forward (timed) -> argmax -> result file -> comparator -> release
 * Differences: paths and counts come from argv (the reference hard-codes them, Main.c:22,30,40),
 * all images of the file are processed unless --count is given (the reference forces n = 1,
 * Main.c:45-46), every written line is compared (the reference checks one, comparator.c:8), and a
 * seeded synthetic model can stand in for the weight blobs the reference's repository lacks.
 *
 *   vit_main [--images FILE] [--weights DIR] [--out FILE] [--answer FILE] [--count N]
 *            [--synthetic SEED] [--repeat R] [--reference-argmax]
 */
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

int main(int argc, char **argv) {
    const char *img_path = "./Data/input-100.bin", *weight_dir = "./Network";
    const char *out_path = "./Data/opencl_result.txt", *answer_path = "./Data/answer_result.txt";
    int count = 0, repeat = 1, synthetic = 0, fix_argmax = 1;
    unsigned long long seed = 1234;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--images") && i + 1 < argc) img_path = argv[++i];
        else if (!strcmp(argv[i], "--weights") && i + 1 < argc) weight_dir = argv[++i];
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
        else if (!strcmp(argv[i], "--answer") && i + 1 < argc) answer_path = argv[++i];
        else if (!strcmp(argv[i], "--count") && i + 1 < argc) count = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--synthetic") && i + 1 < argc) { synthetic = 1; seed = strtoull(argv[++i], NULL, 10); }
        else if (!strcmp(argv[i], "--reference-argmax")) fix_argmax = 0;
        else {
            fprintf(stderr, "usage: %s [--images F] [--weights D] [--out F] [--answer F] [--count N] "
                            "[--synthetic SEED] [--repeat R] [--reference-argmax]\n", argv[0]);
            return 2;
        }
    }

    initialize_hip(); /* Main.c:19 */
    const vit_config cfg = vit_config_b16();
    const int nw = VIT_WEIGHT_COUNT(cfg.depth);

    ImageData *images = synthetic ? vit_synth_images(&cfg, count > 0 ? count : 8, seed + 1)
                                  : load_image_data(img_path); /* Main.c:23 */
    if (!images) return 1;
    Network *network = (Network *)calloc((size_t)nw, sizeof(Network));
    if (!network) return 1;
    if (synthetic) {
        if (vit_synth_weights(&cfg, seed, network, nw)) return 1;
    } else {
        load_weights_cached(weight_dir, network, nw); /* Main.c:30, through the packed cache */
    }

    int n = images[0].n;
    if (count > 0 && count < n) n = count;
    float **prob = (float **)malloc(sizeof(float *) * (size_t)n);
    for (int i = 0; i < n; ++i) prob[i] = (float *)malloc(sizeof(float) * (size_t)cfg.num_classes);
    FILE *fp = fopen(out_path, "w");
    if (!fp) {
        printf("Error: cannot open %s for writing\n", out_path);
        return 1;
    }
    const int total = images[0].n;
    images[0].n = n; /* the forward reads the count from the first element (Main.c:46) */

    printf("=====================Start========================\n");
    for (int r = 0; r < repeat; ++r) {
        const double t0 = now_s();
        ViT_hip(images, network, prob); /* Main.c:57 */
        const double dt = now_s() - t0;
        printf("HIP time: %f sec (%d images, %.1f img/s%s)\n", dt, n, n / dt, r == 0 ? ", includes weight upload" : "");
    }
    vit_write_results(fp, prob, n, cfg.num_classes, fix_argmax); /* Main.c:62-72 */
    fclose(fp);

    if (!synthetic) {
        const int cmp = vit_compare_results(out_path, answer_path, n, 0.01f); /* Main.c:75 */
        if (cmp == 0) printf("Comparator: the two files agree on all %d lines.\n", n);
        else printf("Comparator: %d differences between the two files.\n", cmp);
    }

    images[0].n = total;
    for (int i = 0; i < n; ++i) free(prob[i]);
    free(prob);
    free_weights(network, nw);
    free(network);
    free_image_data(images);
    Release_hip(); /* Main.c:86 */
    return 0;
}
