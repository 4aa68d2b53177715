"""Seeded synthetic weights / images in the reference's on-disk layout.

The reference ships only part of its weights and the GPU box has none, so tests
and bench.py run on synthetic tensors that (a) follow the reference's 152-entry index map
(wiring ViT_seq.c:356-435), (b) have the measured scale of the real
weights and (c) come from a counter-based splitmix64 stream that the C host
library reproduces bit-for-bit (host/vit_synth.c), so a C driver and the Python tests see the
same bytes.  All weights go through the loader's 1e-6 rounding (Network.c:184-187) afterwards.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def stream_seed(seed: int, index: int) -> int:
    """Per-tensor stream id: one splitmix64 step of (seed, index)."""
    with np.errstate(over="ignore"):
        s = np.array([(seed * 0x100000001B3 + index + 1) & 0xFFFFFFFFFFFFFFFF], np.uint64)
        return int(_mix(s * _GAMMA)[0])


def uniform(seed: int, index: int, n: int, lo: float, hi: float) -> np.ndarray:
    """n floats U[lo,hi): element i = lo + (hi-lo) * (mix(stream + (i+1)*gamma) >> 40) * 2^-24."""
    with np.errstate(over="ignore"):
        base = np.uint64(stream_seed(seed, index))
        out = np.empty(n, np.float32)
        step = 1 << 22
        for s in range(0, n, step):
            i = np.arange(s + 1, min(n, s + step) + 1, dtype=np.uint64)
            z = _mix(base + i * _GAMMA)
            u = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
            out[s:s + i.size] = np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u
    return out


def round6(w: np.ndarray) -> np.ndarray:
    """Network.c:184-187: roundf(w * 1e6f) / 1e6f with C roundf (half away from zero)."""
    w = np.asarray(w, np.float32)
    s = (w * np.float32(1000000.0)).astype(np.float64)  # fp32 product, then exact in double
    r = np.copysign(np.floor(np.abs(s) + 0.5), s).astype(np.float32)
    return (r / np.float32(1000000.0)).astype(np.float32)


@dataclass(frozen=True)
class ModelConfig:
    """Runtime form of ViT_seq.c:10-21."""
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_classes: int = 1000
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    hidden_dim: int = 3072

    @property
    def tokens(self) -> int:
        g = self.img_size // self.patch_size
        return g * g + 1

    @property
    def patches(self) -> int:
        return self.tokens - 1

    @property
    def n_weights(self) -> int:
        return 4 + 12 * self.depth + 4

    @property
    def patch_dim(self) -> int:
        return self.in_chans * self.patch_size * self.patch_size

    @property
    def macs_per_image(self) -> int:
        """Algorithmic MACs: conv + depth*(QKV + QK^T + PV + out + fc1 + fc2) + head."""
        T, D, H = self.tokens, self.embed_dim, self.hidden_dim
        hd = D // self.num_heads
        per_layer = T * D * 3 * D + 2 * self.num_heads * T * T * hd + T * D * D + 2 * T * D * H
        return self.patches * self.patch_dim * D + self.depth * per_layer + D * self.num_classes

    def weight_shapes(self):
        D, H, T = self.embed_dim, self.hidden_dim, self.tokens
        shapes = [(D,), (D, self.patch_dim), (D,), (T, D)]
        for _ in range(self.depth):
            shapes += [(D,), (D,), (3 * D, D), (3 * D,), (D, D), (D,), (D,), (D,),
                       (H, D), (H,), (D, H), (D,)]
        shapes += [(D,), (D,), (self.num_classes, D), (self.num_classes,)]
        return shapes


VIT_B16 = ModelConfig()
VIT_L16_384 = ModelConfig(img_size=384, embed_dim=1024, depth=24, num_heads=16, hidden_dim=4096)
# reduced models for fast live-oracle parity (head_dim stays 64 as in ViT-B/L)
VIT_TINY = ModelConfig(img_size=32, num_classes=10, embed_dim=128, depth=2, num_heads=2, hidden_dim=256)
VIT_SMALL = ModelConfig(img_size=64, num_classes=100, embed_dim=192, depth=3, num_heads=3, hidden_dim=768)


def _kind(cfg: ModelConfig, idx: int) -> str:
    if idx < 4:
        return ("cls", "conv_w", "bias", "pos")[idx]
    base = 4 + 12 * cfg.depth
    if idx >= base:
        return ("ln_w", "ln_b", "head_w", "bias")[idx - base]
    return ("ln_w", "ln_b", "qkv_w", "bias", "lin_w", "bias", "ln_w", "ln_b",
            "lin_w", "bias", "lin_w", "bias")[(idx - 4) % 12]


# half-widths a of U(-a, a) (std = a/sqrt(3)); ln_w is U(0.5, 1.0)
_RANGE = {"cls": 0.05, "conv_w": 0.035, "bias": 0.035, "pos": 0.087, "ln_b": 0.05,
          "qkv_w": 0.07, "lin_w": 0.035, "head_w": 0.28}


def make_weight(cfg: ModelConfig, idx: int, seed: int) -> np.ndarray:
    shape = cfg.weight_shapes()[idx]
    n = int(np.prod(shape))
    kind = _kind(cfg, idx)
    if kind == "ln_w":
        w = uniform(seed, idx, n, 0.5, 1.0)
    else:
        a = _RANGE[kind]
        w = uniform(seed, idx, n, -a, a)
    return round6(w).reshape(shape)


def make_weights(cfg: ModelConfig, seed: int = 1234, native: bool = True):
    """List of cfg.n_weights float32 arrays in the reference's index order, already rounded.

    With native=True the C generator of the host library (vit_synth_weights, same bytes --
    tests/test_host_io.py checks it) is used when the library is built: it is ~4x faster for
    the 86.6 M parameters of ViT-B/16.  This only produces INPUT data; no model arithmetic.
    """
    if native:
        try:
            from . import binding
            if os.path.exists(binding.LIB_PATH):
                return binding.synth_weights_c(cfg, seed)
        except OSError:
            pass
    return [make_weight(cfg, i, seed) for i in range(cfg.n_weights)]


def make_images(cfg: ModelConfig, n: int, seed: int = 99) -> np.ndarray:
    """[n][C][H][W] fp32, U(-2.1, 2.6): the range of ImageNet-normalised pixels."""
    per = cfg.in_chans * cfg.img_size * cfg.img_size
    out = np.empty((n, per), np.float32)
    for i in range(n):
        out[i] = uniform(seed, 1_000_000 + i, per, -2.1, 2.6)
    return out.reshape(n, cfg.in_chans, cfg.img_size, cfg.img_size)


def write_weight_files(directory: str, weights, names=None) -> None:
    """Write Weight_<idx>_<name>.bin files (raw LE fp32), the layout load_weights() scans."""
    os.makedirs(directory, exist_ok=True)
    for i, w in enumerate(weights):
        name = names[i] if names else f"t{i}"
        np.asarray(w, "<f4").tofile(os.path.join(directory, f"Weight_{i}_{name}.bin"))


def write_image_file(path: str, images: np.ndarray) -> None:
    """int32 n,c,h,w header + NCHW fp32 (Network.c:36-58)."""
    n, c, h, w = images.shape
    with open(path, "wb") as f:
        np.array([n, c, h, w], "<i4").tofile(f)
        np.asarray(images, "<f4").tofile(f)
