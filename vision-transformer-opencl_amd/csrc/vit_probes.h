/*
 * csrc/vit_probes.h -- entry points that exist ONLY in the probe build (make -C vision-transformer-opencl_amd probes:
 * -DVIT_PROBES, output libvit_mi355x_probe.so).  They are measurement instruments for the tools/ scripts: process-wide
 * overrides of the per-call tuning fields, instrumented kernel builds (cycle stamps, event logs, kernels with a stage
 * removed -- wrong results by construction) and register/store micro-benchmarks.  None of them is compiled into
 * libvit_mi355x.so, which carries no mutable process-wide state.
 *
 *   VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_probe.so python tools/gemm_probe.py ...
 */
#ifndef VIT_PROBES_H
#define VIT_PROBES_H

#include <stddef.h>

#include "vit_hip_kernels.h"

#ifdef __cplusplus
extern "C" {
#endif

/* process-wide override of vithip_gemm_args.tile (0 = none); 101-105, 125, 126, 129 = timing-only / stamped builds,
 * 131-136 = the persistent walk with one part switched off (tools/gemm_f32_switchoff.py) */
int vithip_gemm_set_tile(int tile);
int vithip_gemm_set_group(int group_m);           /* override of vithip_gemm_args.group_m (0 = none) */
int vithip_gemm_set_debug_buffer(void *buf);      /* 8 x u64 stamps per workgroup for the stamped builds */

/* override of vithip_gemm_bf16_args.variant (0 none; 3, 4 = stamped / event-log builds) */
int vithip_gemm_bf16_set_variant(int variant);
int vithip_gemm_bf16_set_group(int group_m);      /* tile rows per L2 group of the bf16 tile walk (0 = the launcher's choice) */
int vithip_gemm_bf16_set_max_workgroups(int n);   /* cap on persistent workgroups of the event-log build */
int vithip_gemm_bf16_set_debug_buffer(void *buf);

int vithip_attention_set_debug_buffer(void *buf);
/* timing experiments of the resident fp32 attention kernel (results wrong by construction): bit 0 no softmax arithmetic,
 * bit 1 no LDS fragment reads, bit 2 waves 4-7 idle, bit 3 no LDS-DMA after the first item */
int vithip_attention_set_probe_mode(int mode); /* 8 x u64 cycle stamps per (image, head) workgroup; NULL disables */

/* register-only fp32 MFMA loop; each wave issues iters*32 v_mfma_f32_32x32x2_f32 (4096 flop each) */
int vithip_probe_mfma_f32(vithip_stream_t stream, float *out, int blocks, int threads, int iters);
/* per wave `iters` 16-B-per-lane stores; mode 0 = 1 KB contiguous, 1 = 16 rows x 64 B, 2 = 8 rows x 128 B at row stride
 * `stride` bytes; cycles[2*wave] = issue span, [2*wave+1] = until complete */
int vithip_probe_store(vithip_stream_t stream, void *out, int blocks, int threads, int iters, int mode, size_t stride, void *cycles);
/* waves 0-3 of every 512-thread block issue iters*32 MFMAs, waves 4-7 valu_iters*64 independent v_fma_f32 */
int vithip_probe_mfma_vs_valu(vithip_stream_t stream, float *out, int blocks, int iters, int valu_iters);

#ifdef __cplusplus
}
#endif
#endif /* VIT_PROBES_H */
