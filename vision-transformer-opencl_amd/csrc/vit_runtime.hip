// csrc/vit_runtime.hip -- HIP runtime plumbing behind the C-ABI of include/vit_hip_kernels.h.
//
// Replaces the OpenCL context/queue/buffer management of the reference
// (ViT_opencl.c:25-124: global g_opencl, clCreateBuffer(COPY_HOST_PTR) per call, blocking
// reads) with explicit device memory, streams, events and graphs that the C engine drives.
#include <hip/hip_runtime.h>

#include <cstring>

#include "vit_hip_kernels.h"

#define RET(expr)                              \
    do {                                       \
        hipError_t e_ = (expr);                \
        return static_cast<int>(e_);           \
    } while (0)

extern "C" {

const char *vithip_error_string(int code) { return hipGetErrorString(static_cast<hipError_t>(code)); }

int vithip_device_count(int *count) { RET(hipGetDeviceCount(count)); }

int vithip_set_device(int device) { RET(hipSetDevice(device)); }

int vithip_get_device_info(int device, vithip_device_info *info) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return static_cast<int>(e);
    std::memset(info, 0, sizeof(*info));
    std::strncpy(info->name, p.name, sizeof(info->name) - 1);
    std::strncpy(info->arch, p.gcnArchName, sizeof(info->arch) - 1);
    info->compute_units = p.multiProcessorCount;
    info->clock_mhz = p.clockRate / 1000;
    info->wavefront = p.warpSize;
    info->lds_per_block = static_cast<int>(p.sharedMemPerBlock);
    info->hbm_bytes = static_cast<unsigned long long>(p.totalGlobalMem);
    return 0;
}

int vithip_malloc(void **ptr, size_t bytes) { RET(hipMalloc(ptr, bytes)); }
int vithip_free(void *ptr) { RET(hipFree(ptr)); }
int vithip_host_alloc(void **ptr, size_t bytes) { RET(hipHostMalloc(ptr, bytes, hipHostMallocDefault)); }
int vithip_host_free(void *ptr) { RET(hipHostFree(ptr)); }

int vithip_memcpy_h2d(void *dst, const void *src, size_t bytes, vithip_stream_t stream) {
    RET(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
}
int vithip_memcpy_d2h(void *dst, const void *src, size_t bytes, vithip_stream_t stream) {
    RET(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
}
int vithip_memcpy_d2d(void *dst, const void *src, size_t bytes, vithip_stream_t stream) {
    RET(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
}
int vithip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, vithip_stream_t stream) {
    if (dst_device == src_device)
        RET(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    RET(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, static_cast<hipStream_t>(stream)));
}
int vithip_memset(void *dst, int value, size_t bytes, vithip_stream_t stream) {
    RET(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
}

int vithip_stream_create(vithip_stream_t *stream) {
    hipStream_t s;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *stream = s;
    return static_cast<int>(e);
}
int vithip_stream_destroy(vithip_stream_t stream) { RET(hipStreamDestroy(static_cast<hipStream_t>(stream))); }
int vithip_stream_sync(vithip_stream_t stream) { RET(hipStreamSynchronize(static_cast<hipStream_t>(stream))); }
int vithip_device_sync(void) { RET(hipDeviceSynchronize()); }

int vithip_event_create(vithip_event_t *event) {
    hipEvent_t ev;
    hipError_t e = hipEventCreate(&ev);
    *event = ev;
    return static_cast<int>(e);
}
int vithip_event_destroy(vithip_event_t event) { RET(hipEventDestroy(static_cast<hipEvent_t>(event))); }
int vithip_event_record(vithip_event_t event, vithip_stream_t stream) {
    RET(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
}
int vithip_event_sync(vithip_event_t event) { RET(hipEventSynchronize(static_cast<hipEvent_t>(event))); }
int vithip_event_elapsed_ms(float *ms, vithip_event_t start, vithip_event_t stop) {
    RET(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
}
int vithip_stream_wait_event(vithip_stream_t stream, vithip_event_t event) {
    RET(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
}

int vithip_graph_begin(vithip_stream_t stream) {
    RET(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
}
int vithip_graph_end(vithip_stream_t stream, vithip_graph_t *graph) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(static_cast<hipStream_t>(stream), &g);
    if (e != hipSuccess) return static_cast<int>(e);
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    *graph = exec;
    return static_cast<int>(e);
}
int vithip_graph_launch(vithip_graph_t graph, vithip_stream_t stream) {
    RET(hipGraphLaunch(static_cast<hipGraphExec_t>(graph), static_cast<hipStream_t>(stream)));
}
int vithip_graph_destroy(vithip_graph_t graph) { RET(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph))); }

}  // extern "C"
