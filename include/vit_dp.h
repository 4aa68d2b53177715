/*
 * include/vit_dp.h -- the ONE exchange of the data-parallel path, behind the C-ABI: an RCCL all-gather of the per-image
 * top-1 records over the GPUs of a node (libvit_mi355x_dp.so = host/vit_dp.c + librccl; the forward library itself,
 * libvit_mi355x.so, links no collective library).
 *
 * What it is for.  The reference has no multi-device code: its batch is a serial loop over independent images
 * (ViT_seq.c:354; the OpenCL path's is ViT_opencl.c:802).  Cut across N GPUs that loop needs no data-path collective -- every
 * device holds a full weight replica and forwards its own contiguous slice -- and the only thing the devices exchange is what
 * BASELINE.json's north star names: "RCCL over xGMI only to gather top-1".  A C caller that drives N engines through
 * vit_engine_forward_device() (one stream per device, results left in HBM) gathers them with this: one grouped
 * ncclAllGather of 8 bytes per image, enqueued on the engines' own streams, so it is ordered behind the forwards without
 * a host synchronisation.  (bench.py's one-process-per-GPU form does the same exchange through torch.distributed: dp.py.)
 *
 *     int devices[8] = {0, 1, 2, 3, 4, 5, 6, 7};
 *     vit_dp *dp;  vit_dp_create(&dp, devices, 8);                      // ncclCommInitAll
 *     for (d = 0; d < 8; ++d)                                            // each on its own device and stream
 *         vit_engine_forward_device(eng[d], images[d], n, probs[d], top1[d], (float *)(top1[d] + n), stream[d]);
 *     vit_dp_gather_top1(dp, (const void *const *)top1, (void *const *)all, n, stream);   // all[d]: [8][2][n] int32 on device d
 *
 * Record layout: per device a packed int32 [2][n] block -- row 0 the labels, row 1 the probabilities' bit patterns -- i.e. the
 * d_top1_label / d_top1_prob outputs of vit_engine_forward_device when they are handed the two halves of ONE allocation
 * (Main.c:62-70 computes exactly these two numbers per image on the host).
 *
 * All functions return 0 or a negative vit_dp error / positive ncclResult_t; vit_dp_last_error() renders it.
 */
#ifndef VIT_DP_H
#define VIT_DP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vit_dp vit_dp;

enum { VIT_DP_OK = 0, VIT_DP_ERR_ARG = -1, VIT_DP_ERR_NOMEM = -2, VIT_DP_ERR_HIP = -3 };

/* One communicator per listed device, in one process (ncclCommInitAll).  `devices` = n distinct HIP ordinals; rank r of the
 * group is devices[r].  n = 1 is allowed (the gather is then a copy on that device). */
int vit_dp_create(vit_dp **out, const int *devices, int n);
void vit_dp_destroy(vit_dp *dp);
int vit_dp_size(const vit_dp *dp);
int vit_dp_device(const vit_dp *dp, int rank);
const char *vit_dp_last_error(const vit_dp *dp);

/*
 * For every rank r: all-gather the `records` packed 8-byte top-1 records at send[r] (device memory of devices[r], int32
 * [2][records]) into recv[r] (device memory of devices[r], int32 [n][2][records], slot q = what rank q sent), enqueued on
 * streams[r] (a hipStream_t of devices[r]; NULL entries / a NULL array = that device's default stream).  One
 * ncclGroupStart ... ncclGroupEnd around the n calls; asynchronous like the forward it follows.  Every rank sends the same
 * count (ragged splits: pad to the largest shard and trim after, as dp.py does).
 */
int vit_dp_gather_top1(vit_dp *dp, const void *const *send, void *const *recv, size_t records, void *const *streams);

#ifdef __cplusplus
}
#endif
#endif /* VIT_DP_H */
