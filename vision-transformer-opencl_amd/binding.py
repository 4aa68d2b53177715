"""ctypes binding of libvit_mi355x.so -- the C-ABI library is the product, this is plumbing.

Everything numeric goes through the shared library (HIP kernels on gfx950).  There is no Python
or CPU fallback: if the library has not been built, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from .synth import ModelConfig

HERE = os.path.dirname(os.path.abspath(__file__))
# VIT_HIP_LIBRARY: the probe build (make probes -> libvit_mi355x_probe.so) for the tools/ scripts; default = the product
LIB_PATH = os.environ.get("VIT_HIP_LIBRARY") or os.path.join(HERE, "libvit_mi355x.so")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)

STAGES = ("embed", "ln", "qkv", "attn", "outproj", "fc1", "fc2", "head", "softmax")
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL = 0, 1, 2


class VitError(RuntimeError):
    pass


class CConfig(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("img_size", "patch_size", "in_chans", "num_classes",
                                       "embed_dim", "depth", "num_heads", "hidden_dim")]

    @classmethod
    def of(cls, cfg: ModelConfig) -> "CConfig":
        return cls(cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.num_classes, cfg.embed_dim,
                   cfg.depth, cfg.num_heads, cfg.hidden_dim)


class CNetwork(C.Structure):  # Network.h:18-21
    _fields_ = [("data", f32p), ("size", C.c_size_t)]


class CImageData(C.Structure):  # Network.h:7-13
    _fields_ = [("n", C.c_int), ("c", C.c_int), ("h", C.c_int), ("w", C.c_int), ("data", f32p)]


class COptions(C.Structure):
    _fields_ = [("device", C.c_int), ("max_batch", C.c_int), ("profile", C.c_int), ("lanes", C.c_int),
                ("dtype", C.c_int), ("prune_last_layer", C.c_int), ("use_graph", C.c_int), ("gemm_tile", C.c_int), ("ln_fold", C.c_int),
                ("gemm_handover_test", C.c_int), ("host_first_piece", C.c_int)]


class CStageTimes(C.Structure):
    _fields_ = [("ms", C.c_double * len(STAGES)), ("launches", C.c_long * len(STAGES)), ("images", C.c_long)]


class CDeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("arch", C.c_char * 64), ("compute_units", C.c_int),
                ("clock_mhz", C.c_int), ("wavefront", C.c_int), ("lds_per_block", C.c_int),
                ("hbm_bytes", C.c_ulonglong)]


class CGemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("W", C.c_void_p), ("ldw", C.c_int),
                ("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("epilogue", C.c_int), ("tile", C.c_int), ("group_m", C.c_int), ("workspace", C.c_void_p),
                ("handover_test", C.c_int), ("ln_rows", C.c_void_p), ("ln_colsum", C.c_void_p),
                ("stats_out", C.c_void_p), ("stats_partials", C.c_void_p)]


_lib: Optional[C.CDLL] = None


DP_LIB_PATH = os.path.join(HERE, "libvit_mi355x_dp.so")
_dp_lib = None


def dp_lib() -> C.CDLL:
    """libvit_mi355x_dp.so (include/vit_dp.h): the RCCL all-gather of the top-1 records behind the C-ABI."""
    global _dp_lib
    if _dp_lib is None:
        if not os.path.exists(DP_LIB_PATH):
            raise VitError(f"{DP_LIB_PATH} is missing: run `make -C {HERE}`")
        L = C.CDLL(DP_LIB_PATH)
        L.vit_dp_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
        L.vit_dp_destroy.argtypes = [C.c_void_p]
        L.vit_dp_destroy.restype = None
        L.vit_dp_size.argtypes = [C.c_void_p]
        L.vit_dp_device.argtypes = [C.c_void_p, C.c_int]
        L.vit_dp_last_error.argtypes = [C.c_void_p]
        L.vit_dp_last_error.restype = C.c_char_p
        L.vit_dp_gather_top1.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p)]
        _dp_lib = L
    return _dp_lib


class DpGroup:
    """vit_dp (include/vit_dp.h): one RCCL communicator per listed device of this process."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        self._h = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        rc = dp_lib().vit_dp_create(C.byref(self._h), arr, len(self.devices))
        if rc != 0:
            raise VitError(f"vit_dp_create({self.devices}) failed: {rc}")

    def gather_top1(self, send_ptrs, recv_ptrs, records: int, streams=None) -> None:
        n = len(self.devices)
        send = (C.c_void_p * n)(*send_ptrs)
        recv = (C.c_void_p * n)(*recv_ptrs)
        st = (C.c_void_p * n)(*streams) if streams is not None else None
        rc = dp_lib().vit_dp_gather_top1(self._h, send, recv, records, st)
        if rc != 0:
            raise VitError(f"vit_dp_gather_top1: {dp_lib().vit_dp_last_error(self._h).decode()} ({rc})")

    def close(self) -> None:
        if self._h:
            dp_lib().vit_dp_destroy(self._h)
            self._h = C.c_void_p()


def lib() -> C.CDLL:
    """Load the native library; never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VitError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                           "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C "
                           f"{HERE}`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.vithip_error_string.restype = C.c_char_p
        L.vit_engine_last_error.restype = C.c_char_p
        L.vit_engine_last_error.argtypes = [C.c_void_p]
        L.vit_engine_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(CConfig), C.POINTER(COptions)]
        L.vit_engine_destroy.argtypes = [C.c_void_p]
        L.vit_engine_load_weights.argtypes = [C.c_void_p, C.POINTER(CNetwork), C.c_int]
        L.vit_engine_forward_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p]
        L.vit_engine_forward_host.argtypes = [C.c_void_p, C.POINTER(f32p), C.c_int, C.POINTER(f32p)]
        L.vit_engine_read_logits.argtypes = [C.c_void_p, f32p, C.c_int]
        L.vit_engine_sync.argtypes = [C.c_void_p]
        L.vit_engine_get_stage_times.argtypes = [C.c_void_p, C.POINTER(CStageTimes)]
        L.vit_engine_reset_stage_times.argtypes = [C.c_void_p]
        L.vit_engine_set_profile.argtypes = [C.c_void_p, C.c_int]
        L.vit_config_weight_size.restype = C.c_size_t
        L.vit_config_weight_size.argtypes = [C.POINTER(CConfig), C.c_int]
        L.vit_config_macs_per_image.restype = C.c_ulonglong
        L.vit_config_macs_per_image.argtypes = [C.POINTER(CConfig)]
        L.vit_config_macs_per_image_pruned.restype = C.c_ulonglong
        L.vit_config_macs_per_image_pruned.argtypes = [C.POINTER(CConfig)]
        L.vit_config_b16.restype = CConfig
        L.load_image_data.restype = C.POINTER(CImageData)
        L.load_image_data.argtypes = [C.c_char_p]
        L.free_image_data.argtypes = [C.POINTER(CImageData)]
        L.load_weights.argtypes = [C.c_char_p, C.POINTER(CNetwork), C.c_int]
        L.free_weights.argtypes = [C.POINTER(CNetwork), C.c_int]
        L.vit_round_weights.argtypes = [f32p, C.c_size_t]
        L.vit_compare_results.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_float]
        L.vit_synth_uniform.argtypes = [C.c_ulonglong, C.c_int, C.c_size_t, C.c_float, C.c_float, f32p]
        L.vit_synth_weights.argtypes = [C.POINTER(CConfig), C.c_ulonglong, C.POINTER(CNetwork), C.c_int]
        L.vit_synth_images.restype = C.POINTER(CImageData)
        L.vit_synth_images.argtypes = [C.POINTER(CConfig), C.c_int, C.c_ulonglong]
        L.vit_argmax.argtypes = [f32p, C.c_int]
        L.ViT_hip.argtypes = [C.POINTER(CImageData), C.POINTER(CNetwork), C.POINTER(f32p)]
        L.ViT_opencl.argtypes = [C.POINTER(CImageData), C.POINTER(CNetwork), C.POINTER(f32p)]
        for fn in ("vithip_malloc", "vithip_host_alloc"):
            getattr(L, fn).argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.vithip_free.argtypes = [C.c_void_p]
        for fn in ("vithip_memcpy_h2d", "vithip_memcpy_d2h", "vithip_memcpy_d2d"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.vithip_gemm_f32.argtypes = [C.c_void_p, C.POINTER(CGemmArgs)]
        L.vithip_patch_embed_f32.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int] * 5
        L.vithip_layernorm_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.vithip_attention_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.vithip_softmax_top1_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_int]
        L.vithip_event_create.argtypes = [C.POINTER(C.c_void_p)]
        L.vithip_event_record.argtypes = [C.c_void_p, C.c_void_p]
        L.vithip_event_sync.argtypes = [C.c_void_p]
        L.vithip_event_destroy.argtypes = [C.c_void_p]
        L.vithip_event_elapsed_ms.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        L.vithip_stream_sync.argtypes = [C.c_void_p]
        L.vithip_get_device_info.argtypes = [C.c_int, C.POINTER(CDeviceInfo)]
        _lib = L
    return _lib


def hip_check(rc: int, what: str = "HIP call") -> None:
    if rc != 0:
        raise VitError(f"{what}: HIP error {rc} ({lib().vithip_error_string(rc).decode()})")


def _as_f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def networks_from(weights: Sequence[np.ndarray]):
    """(ctypes Network[count], keep-alive list).  `None` entries become {NULL, 0}."""
    arr = (CNetwork * len(weights))()
    keep = []
    for i, w in enumerate(weights):
        if w is None:
            arr[i].data = None
            arr[i].size = 0
        else:
            w = _as_f32(w)
            keep.append(w)
            arr[i].data = w.ctypes.data_as(f32p)
            arr[i].size = w.size
    return arr, keep


# --------------------------------------------------------------------------------------------------
# Device buffers for the op-level entry points
# --------------------------------------------------------------------------------------------------
class DeviceArray:
    """A float32/int32 array in HBM owned through the C-ABI (vithip_malloc / vithip_free)."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        hip_check(lib().vithip_malloc(C.byref(p), max(self.nbytes, 16)), "vithip_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        hip_check(lib().vithip_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes, None), "h2d")
        hip_check(lib().vithip_device_sync(), "sync")
        return d

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        hip_check(lib().vithip_device_sync(), "sync")
        hip_check(lib().vithip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, None), "d2h")
        hip_check(lib().vithip_device_sync(), "sync")
        return out

    def free(self) -> None:
        if self.ptr:
            lib().vithip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def gemm(A, W, bias, residual=None, epilogue=EPI_BIAS, tile: int = 0, group_m: int = 0, workspace: bool = False,
         handover_test: int = 0, stats: Optional[dict] = None, ln=None, row_stats: Optional[dict] = None) -> np.ndarray:
    """C = epilogue(A . W^T + bias) through vithip_gemm_f32 (tile / group_m: per-call tuning fields, 0 = auto;
    workspace: lend the scratch that enables the helper pieces of the persistent walk; handover_test: see
    vithip_gemm_args; stats: receives the hand-over counters {"taken", "recomputed"} of the launch;
    ln = (rows [M][2], colsum [N] or None): the consumer side of the LayerNorm fold, W / bias being the folded operands (None: the
    CENTRED weight of ln_fold_weights_f32_centered, nothing to subtract);
    row_stats: a dict that receives "rows" = vithip_gemm_args.stats_out [M][2] and "in_epilogue" = what
    vithip_gemm_f32_stats_in_epilogue said; its key "scratch" (default True) lends stats_partials)."""
    A, W, bias = _as_f32(A), _as_f32(W), _as_f32(bias)
    M, K = A.shape
    N = W.shape[0]
    dA, dW, db = DeviceArray.from_numpy(A), DeviceArray.from_numpy(W), DeviceArray.from_numpy(bias)
    dC = DeviceArray((M, N))
    dR = DeviceArray.from_numpy(_as_f32(residual)) if residual is not None else None
    ws = gemm_workspace() if workspace else None
    dRows = DeviceArray.from_numpy(_as_f32(ln[0])) if ln is not None else None
    dCs = DeviceArray.from_numpy(_as_f32(ln[1])) if ln is not None and ln[1] is not None else None   # None: centred weights
    dSt = DeviceArray((M, 2)) if row_stats is not None else None
    dPart = DeviceArray((max(N // 64, 1), M, 2)) if row_stats is not None and row_stats.get("scratch", True) else None
    args = CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, dR.ptr if dR else None, N, dC.ptr, N, M, N, K, epilogue, tile, group_m, ws,
                     handover_test, dRows.ptr if dRows else None, dCs.ptr if dCs else None, dSt.ptr if dSt else None,
                     dPart.ptr if dPart else None)
    if row_stats is not None:
        row_stats["in_epilogue"] = int(lib().vithip_gemm_f32_stats_in_epilogue(C.byref(args)))
    hip_check(lib().vithip_gemm_f32(None, C.byref(args)), "vithip_gemm_f32")
    out = dC.numpy()
    if row_stats is not None:
        row_stats["rows"] = dSt.numpy()
    if ws:
        if stats is not None:
            stats.update(gemm_workspace_stats(ws))
        lib().vithip_gemm_f32_workspace_destroy(C.c_void_p(ws))
    return out


def gemm_workspace() -> int:
    """vithip_gemm_f32_workspace_create: the handle for vithip_gemm_args.workspace (free with vithip_gemm_f32_workspace_destroy)."""
    p = C.c_void_p()
    hip_check(lib().vithip_gemm_f32_workspace_create(C.byref(p)), "vithip_gemm_f32_workspace_create")
    return p.value


def gemm_workspace_stats(ws: int) -> dict:
    """vithip_gemm_f32_workspace_stats: hand-overs since the last call (syncs the device first)."""
    L = lib()
    hip_check(L.vithip_device_sync(), "vithip_device_sync")
    t, r = C.c_int(), C.c_int()
    L.vithip_gemm_f32_workspace_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    hip_check(L.vithip_gemm_f32_workspace_stats(ws, C.byref(t), C.byref(r)), "vithip_gemm_f32_workspace_stats")
    return {"taken": t.value, "recomputed": r.value}


def gemm_workspace_flags(ws: int, count: int = 1016) -> np.ndarray:
    """The per-owner flags of a workspace (0 = empty after every complete launch)."""
    L = lib()
    L.vithip_gemm_f32_workspace_device_ptr.restype = C.c_void_p
    L.vithip_gemm_f32_workspace_device_ptr.argtypes = [C.c_void_p]
    flags = np.empty(count, np.int32)
    hip_check(L.vithip_device_sync(), "sync")
    hip_check(L.vithip_memcpy_d2h(flags.ctypes.data, L.vithip_gemm_f32_workspace_device_ptr(ws), flags.nbytes, None), "d2h")
    hip_check(L.vithip_device_sync(), "sync")
    return flags


def gemm_workspace_set_flags(ws: int, values) -> None:
    """Overwrite the first len(values) per-owner flags (tests: what an aborted launch may have left behind)."""
    L = lib()
    L.vithip_gemm_f32_workspace_device_ptr.restype = C.c_void_p
    L.vithip_gemm_f32_workspace_device_ptr.argtypes = [C.c_void_p]
    v = np.ascontiguousarray(values, np.int32)
    hip_check(L.vithip_device_sync(), "sync")
    hip_check(L.vithip_memcpy_h2d(L.vithip_gemm_f32_workspace_device_ptr(ws), v.ctypes.data, v.nbytes, None), "h2d")
    hip_check(L.vithip_device_sync(), "sync")


class CGemmBf16Args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("W", C.c_void_p), ("ldw", C.c_int), ("bias", C.c_void_p),
                ("residual", C.c_void_p), ("ldr", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("epilogue", C.c_int),
                ("variant", C.c_int),
                ("ln_rows", C.c_void_p), ("ln_colsum", C.c_void_p), ("x16", C.c_void_p), ("ldx16", C.c_int),
                ("row_partials", C.c_void_p)]


BF16_EPI_BF16, BF16_EPI_BF16_GELU, BF16_EPI_F32_RESIDUAL = 0, 1, 2


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round to nearest even (host model of v_cvt_pk_bf16_f32)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_device(x: np.ndarray) -> np.ndarray:
    """The device conversion kernel (vithip_f32_to_bf16)."""
    x = _as_f32(x)
    L = lib()
    L.vithip_f32_to_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    dx = DeviceArray.from_numpy(x)
    dy = DeviceArray(x.shape, np.uint16)
    hip_check(L.vithip_f32_to_bf16(None, dx.ptr, dy.ptr, x.size), "vithip_f32_to_bf16")
    return dy.numpy()


def gemm_bf16(A_bits, W_bits, bias, residual=None, epilogue=BF16_EPI_BF16, variant: int = 0, ln_rows=None, ln_colsum=None,
              ln_producer: bool = False):
    """vithip_gemm_bf16 on bf16 bit patterns; returns bf16 bits (uint16) or fp32 for the residual epilogue.
    variant: 0 auto, 1 two-stage kernel, 2 ping-pong kernel (fails for K < 128).
    LayerNorm fold: ln_rows [M][2] + ln_colsum [N] make this the consumer; ln_producer (residual epilogue) also returns
    (C, bf16(C) bits, row partials [strips][M][2])."""
    L = lib()
    L.vithip_gemm_bf16.argtypes = [C.c_void_p, C.POINTER(CGemmBf16Args)]
    M, K = A_bits.shape
    N = W_bits.shape[0]
    dA, dW = DeviceArray.from_numpy(np.ascontiguousarray(A_bits, np.uint16)), DeviceArray.from_numpy(np.ascontiguousarray(W_bits, np.uint16))
    db = DeviceArray.from_numpy(_as_f32(bias))
    out_f32 = epilogue == BF16_EPI_F32_RESIDUAL
    dC = DeviceArray((M, N), np.float32 if out_f32 else np.uint16)
    dR = DeviceArray.from_numpy(_as_f32(residual)) if residual is not None else None
    dRows = DeviceArray.from_numpy(_as_f32(ln_rows)) if ln_rows is not None else None
    dCs = DeviceArray.from_numpy(_as_f32(ln_colsum)) if ln_colsum is not None else None
    strips = ln_strips(N)
    dX16 = DeviceArray((M, N), np.uint16) if ln_producer else None
    dPart = DeviceArray((strips, M, 2), np.float32) if ln_producer else None
    args = CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, dR.ptr if dR else None, N, dC.ptr, N, M, N, K, epilogue,
                         variant, dRows.ptr if dRows else None, dCs.ptr if dCs else None,
                         dX16.ptr if dX16 else None, N, dPart.ptr if dPart else None)
    hip_check(L.vithip_gemm_bf16(None, C.byref(args)), "vithip_gemm_bf16")
    if ln_producer:
        return dC.numpy(), dX16.numpy(), dPart.numpy()
    return dC.numpy()


def ln_strips(N: int) -> int:
    return int(lib().vithip_ln_strips(int(N)))


def ln_fold_weights(W, bias, gamma, beta):
    """vithip_ln_fold_weights -> (Wf bf16 bits [N][K], colsum [N], bias_f [N])."""
    W = _as_f32(W)
    N, K = W.shape
    L = lib()
    L.vithip_ln_fold_weights.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int]
    dW, db, dg, dbe = (DeviceArray.from_numpy(_as_f32(a)) for a in (W, bias, gamma, beta))
    dWf, dcs, dbf = DeviceArray((N, K), np.uint16), DeviceArray((N,), np.float32), DeviceArray((N,), np.float32)
    hip_check(L.vithip_ln_fold_weights(None, dW.ptr, db.ptr, dg.ptr, dbe.ptr, dWf.ptr, dcs.ptr, dbf.ptr, N, K), "vithip_ln_fold_weights")
    return dWf.numpy(), dcs.numpy(), dbf.numpy()


def rowstats_bf16(x):
    """vithip_rowstats_bf16 -> (bf16(x) bits, rows [M][2] = (rstd, mean * rstd))."""
    x = _as_f32(x)
    rows, dim = x.shape
    L = lib()
    L.vithip_rowstats_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    dx, d16, dr = DeviceArray.from_numpy(x), DeviceArray((rows, dim), np.uint16), DeviceArray((rows, 2), np.float32)
    hip_check(L.vithip_rowstats_bf16(None, dx.ptr, dim, d16.ptr, dim, dr.ptr, rows, dim), "vithip_rowstats_bf16")
    return d16.numpy(), dr.numpy()


def rowstats_finalize(partials, dim: int):
    """vithip_rowstats_finalize: partials [strips][M][2] -> rows [M][2]."""
    partials = _as_f32(partials)
    strips, rows, _ = partials.shape
    L = lib()
    L.vithip_rowstats_finalize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dp, dr = DeviceArray.from_numpy(partials), DeviceArray((rows, 2), np.float32)
    hip_check(L.vithip_rowstats_finalize(None, dp.ptr, strips, rows, dim, dr.ptr), "vithip_rowstats_finalize")
    return dr.numpy()


def layernorm_bf16out(x, gamma, beta) -> np.ndarray:
    """vithip_layernorm_f32_bf16out -> bf16 bits."""
    x = _as_f32(x)
    rows, dim = x.shape
    L = lib()
    L.vithip_layernorm_f32_bf16out.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    dx, dg, db = DeviceArray.from_numpy(x), DeviceArray.from_numpy(_as_f32(gamma)), DeviceArray.from_numpy(_as_f32(beta))
    dy = DeviceArray((rows, dim), np.uint16)
    hip_check(L.vithip_layernorm_f32_bf16out(None, dx.ptr, dim, dy.ptr, dim, dg.ptr, db.ptr, rows, dim),
              "vithip_layernorm_f32_bf16out")
    return dy.numpy()


QSCALE = 0.18033688011112042  # VITHIP_QSCALE: (1/sqrt(64)) * log2(e)


def attention_bf16io(qkv_bits, n_images: int, tokens: int, heads: int, f32math: bool = False, q_scaled: bool = False,
                     q_rows: Optional[int] = None) -> np.ndarray:
    """vithip_attention_bf16io (bf16 MFMA), vithip_attention_bf16io_f32math (fp32 arithmetic) or, q_scaled,
    vithip_attention_bf16io_qscaled (the Q columns hold QSCALE * q) -> bf16 bits."""
    D = heads * 64
    dq = DeviceArray.from_numpy(np.ascontiguousarray(qkv_bits, np.uint16))
    do = DeviceArray((n_images * tokens, D), np.uint16)
    if q_scaled:
        fn = lib().vithip_attention_bf16io_qscaled
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        hip_check(fn(None, dq.ptr, do.ptr, n_images, tokens, heads, q_rows or tokens), "vithip_attention_bf16io_qscaled")
        return do.numpy()
    fn = getattr(lib(), "vithip_attention_bf16io_f32math" if f32math else "vithip_attention_bf16io")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    hip_check(fn(None, dq.ptr, do.ptr, n_images, tokens, heads), "vithip_attention_bf16io")
    return do.numpy()


def layernorm(x, gamma, beta) -> np.ndarray:
    x = _as_f32(x)
    rows, dim = x.shape
    dx, dg, db = DeviceArray.from_numpy(x), DeviceArray.from_numpy(_as_f32(gamma)), DeviceArray.from_numpy(_as_f32(beta))
    dy = DeviceArray((rows, dim))
    hip_check(lib().vithip_layernorm_f32(None, dx.ptr, dim, dy.ptr, dim, dg.ptr, db.ptr, rows, dim), "vithip_layernorm_f32")
    return dy.numpy()


def ln_fold_weights_f32(W, bias, gamma, beta):
    """vithip_ln_fold_weights_f32 -> (Wf [N][K], colsum [N], bias_f [N])."""
    W, bias, gamma, beta = _as_f32(W), _as_f32(bias), _as_f32(gamma), _as_f32(beta)
    N, K = W.shape
    dW, db, dg, dbe = (DeviceArray.from_numpy(a) for a in (W, bias, gamma, beta))
    dWf, dcs, dbf = DeviceArray((N, K)), DeviceArray((N,)), DeviceArray((N,))
    lib().vithip_ln_fold_weights_f32.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int]
    hip_check(lib().vithip_ln_fold_weights_f32(None, dW.ptr, db.ptr, dg.ptr, dbe.ptr, dWf.ptr, dcs.ptr, dbf.ptr, N, K),
              "vithip_ln_fold_weights_f32")
    return dWf.numpy(), dcs.numpy(), dbf.numpy()


def ln_fold_weights_f32_centered(W, bias, gamma, beta):
    """vithip_ln_fold_weights_f32_centered -> (Wc [N][K] = gamma * W - column mean, residual colsum [N], bias_f [N]): pass Wc and bias_f
    to gemm(..., ln=(rows, None))."""
    W, bias, gamma, beta = _as_f32(W), _as_f32(bias), _as_f32(gamma), _as_f32(beta)
    N, K = W.shape
    dW, db, dg, dbe = (DeviceArray.from_numpy(a) for a in (W, bias, gamma, beta))
    dWf, dcs, dbf = DeviceArray((N, K)), DeviceArray((N,)), DeviceArray((N,))
    lib().vithip_ln_fold_weights_f32_centered.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int]
    hip_check(lib().vithip_ln_fold_weights_f32_centered(None, dW.ptr, db.ptr, dg.ptr, dbe.ptr, dWf.ptr, dcs.ptr, dbf.ptr, N, K),
              "vithip_ln_fold_weights_f32_centered")
    return dWf.numpy(), dcs.numpy(), dbf.numpy()


def rowstats_f32(x) -> np.ndarray:
    """vithip_rowstats_f32 -> [rows][2] = (rstd, mean)."""
    x = _as_f32(x)
    rows, dim = x.shape
    L = lib()
    L.vithip_rowstats_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    dx, dr = DeviceArray.from_numpy(x), DeviceArray((rows, 2))
    hip_check(L.vithip_rowstats_f32(None, dx.ptr, dim, dr.ptr, rows, dim), "vithip_rowstats_f32")
    return dr.numpy()


def attention(qkv, n_images: int, tokens: int, heads: int) -> np.ndarray:
    qkv = _as_f32(qkv)
    D = heads * 64
    assert qkv.shape == (n_images * tokens, 3 * D)
    dq = DeviceArray.from_numpy(qkv)
    do = DeviceArray((n_images * tokens, D))
    hip_check(lib().vithip_attention_f32(None, dq.ptr, do.ptr, n_images, tokens, heads), "vithip_attention_f32")
    return do.numpy()


def attention_rows(qkv, n_images: int, tokens: int, heads: int, q_rows: int, fill: float = 0.0) -> np.ndarray:
    """vithip_attention_f32_rows: the output buffer is pre-filled with `fill` to show which rows are written."""
    qkv = _as_f32(qkv)
    D = heads * 64
    L = lib()
    L.vithip_attention_f32_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    dq = DeviceArray.from_numpy(qkv)
    do = DeviceArray.from_numpy(np.full((n_images * tokens, D), fill, np.float32))
    hip_check(L.vithip_attention_f32_rows(None, dq.ptr, do.ptr, n_images, tokens, heads, q_rows), "vithip_attention_f32_rows")
    return do.numpy()


def patch_embed(cfg: ModelConfig, images, conv_w, conv_b, cls, pos) -> np.ndarray:
    images = _as_f32(images)
    n = images.shape[0]
    d = [DeviceArray.from_numpy(_as_f32(a)) for a in (images, conv_w, conv_b, cls, pos)]
    dx = DeviceArray((n * cfg.tokens, cfg.embed_dim))
    hip_check(lib().vithip_patch_embed_f32(None, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, dx.ptr, n,
                                           cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.embed_dim),
              "vithip_patch_embed_f32")
    return dx.numpy().reshape(n, cfg.tokens, cfg.embed_dim)


def patch_embed_bf16(cfg: ModelConfig, images, conv_w, conv_b, cls, pos, implicit: bool = False) -> np.ndarray:
    """vithip_patch_embed_bf16 (two passes) or vithip_patch_embed_bf16_implicit (one implicit GEMM over the NCHW pixels):
    conv weight given in fp32 and rounded to bf16 here (as the engine does on upload)."""
    images = _as_f32(images)
    n = images.shape[0]
    L = lib()
    if implicit:
        L.vithip_patch_embed_bf16_implicit.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5
        d = [DeviceArray.from_numpy(_as_f32(a)) for a in (images, conv_b, cls, pos)]
        dw = DeviceArray.from_numpy(to_bf16_bits(_as_f32(conv_w).reshape(cfg.embed_dim, -1)))
        dx = DeviceArray((n * cfg.tokens, cfg.embed_dim))
        hip_check(L.vithip_patch_embed_bf16_implicit(None, d[0].ptr, dw.ptr, d[1].ptr, d[2].ptr, d[3].ptr, dx.ptr, n, cfg.img_size,
                                                     cfg.patch_size, cfg.in_chans, cfg.embed_dim), "vithip_patch_embed_bf16_implicit")
        return dx.numpy().reshape(n, cfg.tokens, cfg.embed_dim)
    L.vithip_patch_embed_bf16.argtypes = [C.c_void_p] * 8 + [C.c_int] * 5
    d = [DeviceArray.from_numpy(_as_f32(a)) for a in (images, conv_b, cls, pos)]
    dw = DeviceArray.from_numpy(to_bf16_bits(_as_f32(conv_w).reshape(cfg.embed_dim, -1)))
    dx = DeviceArray((n * cfg.tokens, cfg.embed_dim))
    dp = DeviceArray((n * cfg.patches, cfg.patch_dim), np.uint16)
    hip_check(L.vithip_patch_embed_bf16(None, d[0].ptr, dw.ptr, d[1].ptr, d[2].ptr, d[3].ptr, dx.ptr, dp.ptr, n,
                                        cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.embed_dim), "vithip_patch_embed_bf16")
    return dx.numpy().reshape(n, cfg.tokens, cfg.embed_dim)


def softmax_top1(logits):
    logits = _as_f32(logits)
    rows, classes = logits.shape
    dl = DeviceArray.from_numpy(logits)
    dp = DeviceArray((rows, classes))
    dlab = DeviceArray((rows,), np.int32)
    dpr = DeviceArray((rows,))
    hip_check(lib().vithip_softmax_top1_f32(None, dl.ptr, classes, dp.ptr, classes, dlab.ptr, dpr.ptr, rows, classes),
              "vithip_softmax_top1_f32")
    return dp.numpy(), dlab.numpy(), dpr.numpy()


def device_info(device: int = 0) -> dict:
    info = CDeviceInfo()
    hip_check(lib().vithip_get_device_info(device, C.byref(info)), "vithip_get_device_info")
    return {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
            "clock_mhz": info.clock_mhz, "wavefront": info.wavefront, "lds_per_block": info.lds_per_block,
            "hbm_bytes": int(info.hbm_bytes)}


# --------------------------------------------------------------------------------------------------
# The engine
# --------------------------------------------------------------------------------------------------
class Engine:
    """vit_engine (include/vit_engine.h): weights resident in HBM, batched forward."""

    def __init__(self, cfg: ModelConfig, max_batch: int = 256, device: int = 0, profile: bool = False,
                 lanes: int = 1, dtype: str = "f32", prune_last_layer: bool = False, use_graph: bool = False,
                 gemm_tile: int = 0, ln_fold: int = 0, gemm_handover_test: int = 0, host_first_piece: int = 0):
        self.cfg = cfg
        self._h = C.c_void_p()
        cc = CConfig.of(cfg)
        opt = COptions(device, max_batch, 1 if profile else 0, lanes, {"f32": 0, "bf16": 1}[dtype],
                       1 if prune_last_layer else 0, 1 if use_graph else 0, gemm_tile, ln_fold, gemm_handover_test, host_first_piece)
        rc = lib().vit_engine_create(C.byref(self._h), C.byref(cc), C.byref(opt))
        if rc != 0:
            msg = lib().vit_engine_last_error(self._h).decode() if self._h else "allocation failed"
            if self._h:
                lib().vit_engine_destroy(self._h)
                self._h = C.c_void_p()
            raise VitError(f"vit_engine_create failed ({rc}): {msg}")

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise VitError(f"{what} failed ({rc}): {lib().vit_engine_last_error(self._h).decode()}")

    def load_weights(self, weights: Sequence[np.ndarray]) -> None:
        arr, keep = networks_from(weights)
        self._check(lib().vit_engine_load_weights(self._h, arr, len(weights)), "vit_engine_load_weights")

    def load_weight_image(self, img: "WeightImage") -> None:
        self._check(lib().vit_engine_load_weight_image(self._h, C.byref(img.c)), "vit_engine_load_weight_image")

    def read_weight_image(self) -> "WeightImage":
        img = WeightImage()
        self._check(lib().vit_engine_read_weight_image(self._h, C.byref(img.c)), "vit_engine_read_weight_image")
        return img

    def copy_weights_from(self, other: "Engine") -> None:
        L = lib()
        L.vit_engine_copy_weights.argtypes = [C.c_void_p, C.c_void_p]
        self._check(L.vit_engine_copy_weights(self._h, other._h), "vit_engine_copy_weights")

    def forward(self, images: np.ndarray) -> np.ndarray:
        """Host path (the ViT_opencl-shaped one): per-image pointers in, per-image rows out."""
        images = _as_f32(images)
        n = images.shape[0]
        rows = [images[i] for i in range(n)]
        probs = np.empty((n, self.cfg.num_classes), np.float32)
        in_ptrs = (f32p * n)(*[r.ctypes.data_as(f32p) for r in rows])
        out_ptrs = (f32p * n)(*[probs[i].ctypes.data_as(f32p) for i in range(n)])
        self._check(lib().vit_engine_forward_host(self._h, in_ptrs, n, out_ptrs), "vit_engine_forward_host")
        return probs

    def forward_device(self, d_images: int, n: int, d_probs: int, d_label: int = 0, d_prob: int = 0,
                       stream: int = 0) -> None:
        """Device-resident path: raw HBM addresses (e.g. torch data_ptr()), async on `stream`."""
        self._check(lib().vit_engine_forward_device(self._h, d_images, n, d_probs, d_label or None,
                                                    d_prob or None, stream or None),
                    "vit_engine_forward_device")

    def sync(self) -> None:
        self._check(lib().vit_engine_sync(self._h), "vit_engine_sync")

    def logits(self, rows: int) -> np.ndarray:
        out = np.empty((rows, self.cfg.num_classes), np.float32)
        self._check(lib().vit_engine_read_logits(self._h, out.ctypes.data_as(f32p), rows), "vit_engine_read_logits")
        return out

    def handover_stats(self) -> dict:
        """vit_engine_handover_stats: fp32 GEMM helper pieces taken / recomputed since the last call (syncs the device)."""
        L = lib()
        t, r = C.c_long(), C.c_long()
        L.vit_engine_handover_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        self._check(L.vit_engine_handover_stats(self._h, C.byref(t), C.byref(r)), "vit_engine_handover_stats")
        return {"taken": int(t.value), "recomputed": int(r.value)}

    def set_lanes(self, lanes: int) -> None:
        L = lib()
        L.vit_engine_set_lanes.argtypes = [C.c_void_p, C.c_int]
        self._check(L.vit_engine_set_lanes(self._h, lanes), "vit_engine_set_lanes")

    def set_profile(self, on: bool) -> None:
        self._check(lib().vit_engine_set_profile(self._h, 1 if on else 0), "vit_engine_set_profile")

    def reset_stage_times(self) -> None:
        lib().vit_engine_reset_stage_times(self._h)

    def stage_times(self) -> dict:
        t = CStageTimes()
        self._check(lib().vit_engine_get_stage_times(self._h, C.byref(t)), "vit_engine_get_stage_times")
        return {"images": int(t.images),
                "stages": {s: {"ms": float(t.ms[i]), "launches": int(t.launches[i])} for i, s in enumerate(STAGES)}}

    def close(self) -> None:
        if self._h:
            lib().vit_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------------------------------
# Facade + io
# --------------------------------------------------------------------------------------------------
def facade_forward(images: np.ndarray, weights: Sequence[np.ndarray], use_reference_names: bool = True) -> np.ndarray:
    """initialize_* -> ViT_*(ImageData*, Network*, float**) -> Release_* exactly as Main.c drives it."""
    L = lib()
    images = _as_f32(images)
    n, c, h, w = images.shape
    imgs = (CImageData * n)()
    rows = [images[i] for i in range(n)]
    for i in range(n):
        imgs[i] = CImageData(n, c, h, w, rows[i].ctypes.data_as(f32p))
    nets, keep = networks_from(weights)
    probs = np.empty((n, 1000), np.float32)
    out_ptrs = (f32p * n)(*[probs[i].ctypes.data_as(f32p) for i in range(n)])
    if use_reference_names:
        L.initialize_opencl()
        L.ViT_opencl(imgs, nets, out_ptrs)
        L.Release_opencl()
    else:
        L.initialize_hip()
        L.ViT_hip(imgs, nets, out_ptrs)
        L.Release_hip()
    return probs


def load_image_file(path: str) -> Optional[np.ndarray]:
    L = lib()
    p = L.load_image_data(path.encode())
    if not p:
        return None
    n, c, h, w = p[0].n, p[0].c, p[0].h, p[0].w
    out = np.stack([np.ctypeslib.as_array(p[i].data, shape=(c, h, w)).copy() for i in range(n)])
    L.free_image_data(p)
    return out


def load_weight_dir(directory: str, count: int):
    """load_weights() -> list of arrays (None where the file is absent)."""
    L = lib()
    nets = (CNetwork * count)()
    L.load_weights(directory.encode(), nets, count)
    out = [np.ctypeslib.as_array(nets[i].data, shape=(nets[i].size,)).copy() if nets[i].data else None
           for i in range(count)]
    L.free_weights(nets, count)
    return out


def load_weight_dir_cached(cfg: ModelConfig, directory: str, cache_path: Optional[str] = None):
    """load_weights_cached(): Network[] through the packed cache file (validated against the directory's files)."""
    L = lib()
    L.load_weights_cached.argtypes = [C.POINTER(CConfig), C.c_char_p, C.POINTER(CNetwork), C.c_int, C.c_char_p]
    count = cfg.n_weights
    nets = (CNetwork * count)()
    cc = CConfig.of(cfg)
    L.load_weights_cached(C.byref(cc), directory.encode(), nets, count, cache_path.encode() if cache_path else None)
    out = [np.ctypeslib.as_array(nets[i].data, shape=(nets[i].size,)).copy() if nets[i].data else None
           for i in range(count)]
    L.free_weights(nets, count)
    return out


class CWeightImage(C.Structure):  # include/vit_io.h: vit_weight_image
    _fields_ = [("cfg", CConfig), ("count", C.c_int), ("f32_floats", C.c_size_t), ("gemm_floats", C.c_size_t),
                ("bf16_elems", C.c_size_t), ("off", C.POINTER(C.c_size_t)), ("size", C.POINTER(C.c_size_t)),
                ("f32", f32p), ("bf16", C.POINTER(C.c_ushort))]


class WeightImage:
    """vit_weight_image: the device layout of a model's weights on the host / in the cache file."""

    def __init__(self):
        self.c = CWeightImage()
        L = lib()
        L.vit_weight_image_build.argtypes = [C.POINTER(CWeightImage), C.POINTER(CConfig), C.POINTER(CNetwork), C.c_int, C.c_int]
        L.vit_weight_image_free.argtypes = [C.POINTER(CWeightImage)]
        L.vit_weight_image_save.argtypes = [C.POINTER(CWeightImage), C.c_char_p, C.c_char_p]
        L.vit_weight_image_load.argtypes = [C.POINTER(CWeightImage), C.c_char_p, C.POINTER(CConfig), C.c_char_p]
        L.vit_engine_load_weight_image.argtypes = [C.c_void_p, C.POINTER(CWeightImage)]
        L.vit_engine_read_weight_image.argtypes = [C.c_void_p, C.POINTER(CWeightImage)]

    @classmethod
    def build(cls, cfg: ModelConfig, weights: Sequence[np.ndarray], with_bf16: bool = True) -> "WeightImage":
        img = cls()
        arr, keep = networks_from(weights)
        cc = CConfig.of(cfg)
        if lib().vit_weight_image_build(C.byref(img.c), C.byref(cc), arr, len(weights), 1 if with_bf16 else 0) != 0:
            raise VitError("vit_weight_image_build: incomplete or mis-sized weight set")
        return img

    @classmethod
    def load(cls, cfg: ModelConfig, path: str, source_dir: Optional[str] = None) -> Optional["WeightImage"]:
        img = cls()
        cc = CConfig.of(cfg)
        rc = lib().vit_weight_image_load(C.byref(img.c), path.encode(), C.byref(cc), source_dir.encode() if source_dir else None)
        return img if rc == 0 else None

    def save(self, path: str, source_dir: Optional[str] = None) -> int:
        return lib().vit_weight_image_save(C.byref(self.c), path.encode(), source_dir.encode() if source_dir else None)

    def tensors(self):
        return [np.ctypeslib.as_array(self.c.f32, shape=(self.c.f32_floats,))[self.c.off[i]:self.c.off[i] + self.c.size[i]].copy()
                for i in range(self.c.count)]

    def f32_section(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.c.f32, shape=(self.c.f32_floats,)).copy()

    def bf16_section(self) -> Optional[np.ndarray]:
        if not self.c.bf16_elems:
            return None
        return np.ctypeslib.as_array(self.c.bf16, shape=(self.c.bf16_elems,)).copy()

    def free(self) -> None:
        lib().vit_weight_image_free(C.byref(self.c))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def read_image_file_chunked(path: str, chunk: int):
    """vit_image_reader_*: list of (first_index, array) chunks, or None when the file cannot be opened."""
    L = lib()
    L.vit_image_reader_open.restype = C.c_void_p
    L.vit_image_reader_open.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4
    L.vit_image_reader_next.restype = C.POINTER(CImageData)
    L.vit_image_reader_next.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.vit_image_reader_close.argtypes = [C.c_void_p]
    dims = [C.c_int() for _ in range(4)]
    r = L.vit_image_reader_open(path.encode(), *[C.byref(d) for d in dims])
    if not r:
        return None
    out = []
    while True:
        first = C.c_int()
        p = L.vit_image_reader_next(r, chunk, C.byref(first))
        if not p:
            break
        k, c, h, w = p[0].n, p[0].c, p[0].h, p[0].w
        assert all(p[i].n == k for i in range(k))
        out.append((first.value, np.stack([np.ctypeslib.as_array(p[i].data, shape=(c, h, w)).copy() for i in range(k)])))
        L.free_image_data(p)
    L.vit_image_reader_close(r)
    return dims[0].value, out


def round_weights(w: np.ndarray) -> np.ndarray:
    out = _as_f32(w).copy()
    lib().vit_round_weights(out.ctypes.data_as(f32p), out.size)
    return out


def synth_uniform(seed: int, index: int, n: int, lo: float, hi: float) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().vit_synth_uniform(seed, index, n, lo, hi, out.ctypes.data_as(f32p))
    return out


def synth_weights_c(cfg: ModelConfig, seed: int):
    """The C generator (vit_synth_uniform + vit_round_weights of the host library, the pair vit_synth_weights() runs per tensor),
    filling numpy arrays in place; must equal synth.make_weights(native=False) bit for bit (tests/test_host_io.py)."""
    from .synth import _RANGE, _kind
    L = lib()
    L.vit_synth_uniform.argtypes = [C.c_ulonglong, C.c_int, C.c_size_t, C.c_float, C.c_float, C.c_void_p]
    L.vit_synth_uniform.restype = None
    L.vit_round_weights.argtypes = [C.c_void_p, C.c_size_t]
    L.vit_round_weights.restype = None
    out = []
    for i, shape in enumerate(cfg.weight_shapes()):
        kind = _kind(cfg, i)
        lo, hi = (0.5, 1.0) if kind == "ln_w" else (-_RANGE[kind], _RANGE[kind])
        w = np.empty(shape, np.float32)
        L.vit_synth_uniform(seed, i, w.size, lo, hi, w.ctypes.data)
        L.vit_round_weights(w.ctypes.data, w.size)
        out.append(w)
    return out


def synth_images_c(cfg: ModelConfig, n: int, seed: int) -> np.ndarray:
    L = lib()
    cc = CConfig.of(cfg)
    p = L.vit_synth_images(C.byref(cc), n, seed)
    out = np.stack([np.ctypeslib.as_array(p[i].data, shape=(cfg.in_chans, cfg.img_size, cfg.img_size)).copy()
                    for i in range(n)])
    L.free_image_data(p)
    return out


def write_results(path: str, probs: np.ndarray, fix_argmax: bool = True) -> int:
    probs = _as_f32(probs)
    n, classes = probs.shape
    ptrs = (f32p * n)(*[probs[i].ctypes.data_as(f32p) for i in range(n)])
    L = lib()
    L.vit_write_results_file.argtypes = [C.c_char_p, C.POINTER(f32p), C.c_int, C.c_int, C.c_int]
    return L.vit_write_results_file(path.encode(), ptrs, n, classes, 1 if fix_argmax else 0)


def compare_results(result_path: str, answer_path: str, lines: int, tol: float = 0.01) -> int:
    return lib().vit_compare_results(result_path.encode(), answer_path.encode(), lines, tol)
