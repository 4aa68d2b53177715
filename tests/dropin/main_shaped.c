/*
 * tests/dropin/main_shaped.c -- a caller shaped like the reference's Main.c:18-88, compiled by
 * tests/test_gpu_dropin.py against THE REFERENCE'S HEADER NAMES ("Network.h", "ViT_opencl.h", "comparator.h": the
 * one-line forwarding headers of INTEGRATION.md section 2, generated next to this file by the test) and linked with
 * -lvit_mi355x.  It calls only what Main.c calls, with Main.c's relative paths and result-line format:
 *
 *   initialize_opencl -> load_image_data("./Data/input-<N>.bin") -> load_weights("./Network", network, 152)
 *   -> ViT_opencl(images, network, probabilities) -> "./Data/opencl_result.txt" -> comparator() -> Release_opencl
 *
 * Two deliberate differences from Main.c, both switchable: the input file name comes from argv[1] (Main.c:22 hard-codes
 * input-100.bin, which is absent upstream), and with -DALL_IMAGES every image is forwarded (Main.c:45-46 forces n = 1).
 */
#include <stdio.h>
#include <stdlib.h>

#include "Network.h"
#include "ViT_opencl.h"
#include "comparator.h"

int main(int argc, char **argv) {
    initialize_opencl();
    const char *img_filename = argc > 1 ? argv[1] : "./Data/input-100.bin";
    ImageData *images = load_image_data(img_filename);
    if (images == NULL) return 1;

    Network network[152];
    load_weights("./Network", network, 152);

    int n = images->n;
    float **probabilities = (float **)malloc(sizeof(float *) * n);
    for (int i = 0; i < n; i++) probabilities[i] = (float *)malloc(sizeof(float) * 1000);

    FILE *fp_output = fopen("./Data/opencl_result.txt", "w");
    if (fp_output == NULL) {
        printf("Error: cannot open ./Data/opencl_result.txt for writing\n");
        return 1;
    }
#ifndef ALL_IMAGES
    n = 1;
    images->n = n;
#endif
    ViT_opencl(images, network, probabilities);

    for (int i = 0; i < n; i++) {
        int pred_idx = 0;
        for (int j = 1; j < 1000; j++)
            if (probabilities[i][j] > probabilities[i][pred_idx]) pred_idx = j;
        fprintf(fp_output, "[%d] label: %d / prob: %.6f\n", i, pred_idx, probabilities[i][pred_idx]);
    }
    fclose(fp_output);

    int cmp = comparator();
    printf("comparator=%d images=%d\n", cmp, n);
    Release_opencl();
    return cmp == 0 ? 0 : 3;
}
