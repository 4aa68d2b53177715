/*
 * include/vit_types.h -- data model shared by the loaders and the forward entry points.
 *
 * These two structs ARE the reference's (Network.h:7-13 and Network.h:18-21, duplicated at
 * Network.c:9-20): a caller written against the reference passes them unchanged.
 *
 *   ImageData: load_image_data() returns an ARRAY of n structs; every element carries the
 *              total count in .n and owns a separately malloc'd CHW fp32 image in .data
 *              (Network.c:66-93).  The forward entry reads image[0].n images.
 *   Network:   one weight tensor; .size is the ELEMENT count; {NULL,0} when the file was
 *              absent (Network.c:127-130,190-191).
 */
#ifndef VIT_TYPES_H
#define VIT_TYPES_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int n;        /* total number of images in the array this element belongs to */
    int c;        /* channels */
    int h;        /* height   */
    int w;        /* width    */
    float *data;  /* this image, CHW fp32, separately allocated */
} ImageData;

typedef struct {
    float *data;  /* raw fp32 tensor in PyTorch layout (Linear.weight = [out][in]) */
    size_t size;  /* number of floats */
} Network;

/*
 * Model dimensions.  The reference fixes them with macros (ViT_seq.c:10-21,
 * ViT_opencl.c:12-23); here they are a runtime struct so that the same engine serves
 * ViT-B/16-224 (the default), reduced test models and ViT-L/16-384.
 * Constraints of the HIP path: embed_dim / num_heads == 64, embed_dim % 32 == 0, embed_dim <= 2048,
 * hidden_dim % 32 == 0, patch_size % 4 == 0, in_chans*patch_size^2 % 32 == 0.
 */
typedef struct {
    int img_size;     /* 224 */
    int patch_size;   /* 16  */
    int in_chans;     /* 3   */
    int num_classes;  /* 1000 */
    int embed_dim;    /* 768 */
    int depth;        /* 12  */
    int num_heads;    /* 12  */
    int hidden_dim;   /* 3072 = (int)(embed_dim * mlp_ratio), ViT_seq.c:254 */
} vit_config;

#define VIT_WEIGHTS_PER_LAYER 12
/* number of Network entries: cls, conv w/b, pos, 12 per layer, ln w/b, head w/b (Main.c:29-30 => 152) */
#define VIT_WEIGHT_COUNT(depth) (4 + VIT_WEIGHTS_PER_LAYER * (depth) + 4)

#ifdef __cplusplus
}
#endif
#endif /* VIT_TYPES_H */
