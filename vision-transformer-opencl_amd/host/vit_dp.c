/*
 * host/vit_dp.c -- RCCL all-gather of the per-image top-1 records across the GPUs of one node (include/vit_dp.h).
 *
 * Plain C over rccl.h; built into its own small library (libvit_mi355x_dp.so) so that the forward library carries no
 * collective-library dependency.  One process drives every device: ncclCommInitAll makes one communicator per device, and a
 * gather is one ncclAllGather per communicator inside ncclGroupStart / ncclGroupEnd -- the single-process form of the exchange
 * SURVEY.md 8(e) describes (16 KB per rank at 2,048 images per GPU: latency-bound on the xGMI mesh, so no bucketing, no ring
 * tuning, one call per batch on the stream of the forward that produced the records).
 * Shard axis: the reference's image loop, ViT_opencl.c:802 / ViT_seq.c:354.
 */
#define __HIP_PLATFORM_AMD__ 1
#include "vit_dp.h"

#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct vit_dp {
    int n;
    int *devices;
    ncclComm_t *comms;
    char err[256];
};

static int fail(vit_dp *dp, int code, const char *what, const char *detail) {
    if (dp) snprintf(dp->err, sizeof(dp->err), "%s: %s", what, detail ? detail : "");
    return code;
}

int vit_dp_create(vit_dp **out, const int *devices, int n) {
    if (!out) return VIT_DP_ERR_ARG;
    *out = NULL;
    if (!devices || n <= 0) return VIT_DP_ERR_ARG;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return VIT_DP_ERR_HIP;
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= have) return VIT_DP_ERR_ARG;
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) return VIT_DP_ERR_ARG; /* a communicator group holds a device once */
    }
    vit_dp *dp = (vit_dp *)calloc(1, sizeof(*dp));
    if (!dp) return VIT_DP_ERR_NOMEM;
    dp->n = n;
    dp->devices = (int *)malloc((size_t)n * sizeof(int));
    dp->comms = (ncclComm_t *)calloc((size_t)n, sizeof(ncclComm_t));
    if (!dp->devices || !dp->comms) {
        vit_dp_destroy(dp);
        return VIT_DP_ERR_NOMEM;
    }
    memcpy(dp->devices, devices, (size_t)n * sizeof(int));
    int before = 0;
    (void)hipGetDevice(&before);
    const ncclResult_t r = ncclCommInitAll(dp->comms, n, dp->devices);
    (void)hipSetDevice(before); /* the caller's current device is not ours to change */
    if (r != ncclSuccess) {
        fprintf(stderr, "vit_dp_create: ncclCommInitAll over %d device(s): %s\n", n, ncclGetErrorString(r));
        memset(dp->comms, 0, (size_t)n * sizeof(ncclComm_t));
        vit_dp_destroy(dp);
        return (int)r;
    }
    *out = dp;
    return VIT_DP_OK;
}

void vit_dp_destroy(vit_dp *dp) {
    if (!dp) return;
    if (dp->comms)
        for (int i = 0; i < dp->n; ++i)
            if (dp->comms[i]) (void)ncclCommDestroy(dp->comms[i]);
    free(dp->comms);
    free(dp->devices);
    free(dp);
}

int vit_dp_size(const vit_dp *dp) { return dp ? dp->n : 0; }
int vit_dp_device(const vit_dp *dp, int rank) { return (dp && rank >= 0 && rank < dp->n) ? dp->devices[rank] : -1; }
const char *vit_dp_last_error(const vit_dp *dp) { return dp ? dp->err : "null vit_dp"; }

int vit_dp_gather_top1(vit_dp *dp, const void *const *send, void *const *recv, size_t records, void *const *streams) {
    if (!dp) return VIT_DP_ERR_ARG;
    if (!send || !recv || records == 0) return fail(dp, VIT_DP_ERR_ARG, "vit_dp_gather_top1", "null buffers or no records");
    for (int r = 0; r < dp->n; ++r)
        if (!send[r] || !recv[r]) return fail(dp, VIT_DP_ERR_ARG, "vit_dp_gather_top1", "a rank's buffer is NULL");
    int before = 0;
    (void)hipGetDevice(&before);
    ncclResult_t res = ncclGroupStart();
    if (res != ncclSuccess) return fail(dp, (int)res, "ncclGroupStart", ncclGetErrorString(res));
    ncclResult_t first_bad = ncclSuccess;
    for (int r = 0; r < dp->n; ++r) {
        /* 2 x records int32 per rank: [labels | probability bits] */
        (void)hipSetDevice(dp->devices[r]);
        res = ncclAllGather(send[r], recv[r], 2 * records, ncclInt32, dp->comms[r], streams ? (hipStream_t)streams[r] : (hipStream_t)0);
        if (res != ncclSuccess && first_bad == ncclSuccess) first_bad = res;
    }
    res = ncclGroupEnd();
    (void)hipSetDevice(before);
    if (first_bad != ncclSuccess) return fail(dp, (int)first_bad, "ncclAllGather", ncclGetErrorString(first_bad));
    if (res != ncclSuccess) return fail(dp, (int)res, "ncclGroupEnd", ncclGetErrorString(res));
    return VIT_DP_OK;
}
