"""ctypes front-end of the parity CHECKER -- test infrastructure, never product code.

Two libraries are wrapped:

* ``oracle/libvit_cpu_ref.so`` -- our CPU restatement (vit_cpu_ref.c), parametrised by a
  config, travels to the GPU box.  Parity PINNED: bit-identical to the compiled reference
  (tests/test_oracle_vs_reference.py) and to tests/golden/ (emitted by the compiled reference).
* ``oracle/_ref/libvitseq_ref.so`` -- the reference's own ViT_seq.c compiled in the build
  container (``make -C oracle ref``); only present where /root/reference was available
  at build time (plus wherever the prebuilt file was shipped).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvit_cpu_ref.so")
REF_PATH = os.path.join(HERE, "_ref", "libvitseq_ref.so")

_f32p = C.POINTER(C.c_float)


def _p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


class _Cfg(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("img_size", "patch_size", "in_chans", "num_classes",
                                       "embed_dim", "depth", "num_heads", "hidden_dim")]


class _Tensor(C.Structure):  # == reference Network (Network.h:18-21)
    _fields_ = [("data", _f32p), ("size", C.c_size_t)]


@dataclass(frozen=True)
class Config:
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
    def n_weights(self) -> int:
        return 4 + 12 * self.depth + 4

    def c(self) -> _Cfg:
        return _Cfg(self.img_size, self.patch_size, self.in_chans, self.num_classes,
                    self.embed_dim, self.depth, self.num_heads, self.hidden_dim)


def build(ref: bool = True) -> None:
    """Compile the checker (and the reference build when /root/reference is mounted)."""
    subprocess.run(["make", "-C", HERE, "-s"], check=True)
    if ref and os.path.exists("/root/reference/ViT_seq.c"):
        subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        _lib = C.CDLL(LIB_PATH)
        _lib.vitref_gelu.restype = C.c_float
        _lib.vitref_gelu.argtypes = [C.c_float]
        _lib.vitref_tokens.restype = C.c_int
    return _lib


def set_threads(n: int) -> None:
    lib().vitref_set_threads(C.c_int(n))


def _tensors(weights):
    arr = (_Tensor * len(weights))()
    for i, w in enumerate(weights):
        arr[i].data = _p(w)
        arr[i].size = w.size
    return arr


def round_weights(w: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(w, dtype=np.float32).copy()
    lib().vitref_round_weights(_p(out), C.c_size_t(out.size))
    return out


def layer_norm(x, w, b):
    t, d = x.shape
    y = np.empty_like(x)
    lib().vitref_layer_norm(_p(x), _p(y), t, d, _p(w), _p(b))
    return y


def linear(x, w, b):
    t, k = x.shape
    n = w.shape[0]
    y = np.empty((t, n), np.float32)
    lib().vitref_linear(_p(x), _p(y), t, k, n, _p(w), _p(b))
    return y


def gelu(x):
    f = lib().vitref_gelu
    return np.array([f(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def attention_core(q, k, v, heads):
    t, d = q.shape
    o = np.empty_like(q)
    lib().vitref_attention_core(_p(q), _p(k), _p(v), _p(o), t, d, heads)
    return o


def multihead_attn(x, in_w, in_b, out_w, out_b, heads):
    t, d = x.shape
    y = np.empty_like(x)
    lib().vitref_multihead_attn(_p(x), _p(y), t, d, heads, _p(in_w), _p(in_b), _p(out_w), _p(out_b))
    return y


def mlp_block(x, w1, b1, w2, b2):
    t, d = x.shape
    y = np.empty_like(x)
    lib().vitref_mlp_block(_p(x), _p(y), t, d, w1.shape[0], _p(w1), _p(b1), _p(w2), _p(b2))
    return y


def encoder(x, w12, heads):
    t, d = x.shape
    y = np.empty_like(x)
    ptrs = (_f32p * 12)(*[_p(w) for w in w12])
    lib().vitref_encoder(_p(x), _p(y), t, d, heads, w12[8].shape[0], ptrs)
    return y


def softmax(logits):
    p = np.empty_like(logits)
    lib().vitref_softmax(_p(logits), _p(p), logits.size)
    return p


def conv2d(cfg: Config, image, w, b):
    g = cfg.img_size // cfg.patch_size
    out = np.empty((cfg.embed_dim, g, g), np.float32)
    c = cfg.c()
    lib().vitref_conv2d(C.byref(c), _p(image), _p(out), _p(w), _p(b))
    return out


def embed(cfg: Config, image, weights):
    """conv_proj -> flatten_transpose -> class_token -> pos_emb (ViT_seq.c:356-362)."""
    c = cfg.c()
    L = lib()
    g = cfg.img_size // cfg.patch_size
    conv = conv2d(cfg, image, weights[1], weights[2])
    flat = np.empty((g * g, cfg.embed_dim), np.float32)
    L.vitref_flatten_transpose(C.byref(c), _p(conv), _p(flat))
    tok = np.empty((cfg.tokens, cfg.embed_dim), np.float32)
    L.vitref_class_token(C.byref(c), _p(flat), _p(tok), _p(weights[0]))
    out = np.empty_like(tok)
    L.vitref_pos_emb(C.byref(c), _p(tok), _p(out), _p(weights[3]))
    return out


def forward_image(cfg: Config, image, weights, want_stages=False):
    """Returns (probs, logits, stages or None); stages = [depth+1][tokens][dim]."""
    assert len(weights) == cfg.n_weights
    probs = np.empty(cfg.num_classes, np.float32)
    logits = np.empty(cfg.num_classes, np.float32)
    stages = np.empty((cfg.depth + 1, cfg.tokens, cfg.embed_dim), np.float32) if want_stages else None
    c = cfg.c()
    image = np.ascontiguousarray(image, np.float32)
    lib().vitref_forward_image(C.byref(c), _p(image), _tensors(weights), _p(probs), _p(logits),
                               _p(stages) if want_stages else None)
    return probs, logits, stages


def forward(cfg: Config, images, weights):
    """Batch of CHW images -> probs [n][classes]."""
    return np.stack([forward_image(cfg, im, weights)[0] for im in images])


# ----------------------------------------------------------------------------------------------
# The compiled reference itself (ViT-B/16 only: its dimensions are macros, ViT_seq.c:10-21).
# ----------------------------------------------------------------------------------------------
class _RefImage(C.Structure):  # Network.h:7-13
    _fields_ = [("n", C.c_int), ("c", C.c_int), ("h", C.c_int), ("w", C.c_int), ("data", _f32p)]


_ref = None


def have_reference() -> bool:
    return os.path.exists(REF_PATH)


def ref_lib() -> C.CDLL:
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_PATH)
        _ref.gelu.restype = C.c_float
        _ref.gelu.argtypes = [C.c_float]
    return _ref


def _net(a: np.ndarray) -> _Tensor:  # passed BY VALUE to the reference's functions
    return _Tensor(_p(a), a.size)


class Reference:
    """Thin access to the exported functions of the reference's ViT_seq.c (ViT_seq.h:6-20)."""

    cfg = Config()

    def layer_norm(self, x, w, b):
        y = np.empty_like(x)
        ref_lib().layer_norm(_p(x), _p(y), _net(w), _net(b))
        return y

    def linear(self, x, w, b):
        t, k = x.shape
        n = w.shape[0]
        y = np.empty((t, n), np.float32)
        ref_lib().linear_layer(_p(x), _p(y), t, k, n, _net(w), _net(b))
        return y

    def gelu(self, x):
        f = ref_lib().gelu
        return np.array([f(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)

    def multihead_attn(self, x, in_w, in_b, out_w, out_b):
        y = np.empty_like(x)
        ref_lib().multihead_attn(_p(x), _p(y), _net(in_w), _net(in_b), _net(out_w), _net(out_b))
        return y

    def mlp_block(self, x, w1, b1, w2, b2):
        y = np.empty_like(x)
        ref_lib().mlp_block(_p(x), _p(y), _net(w1), _net(b1), _net(w2), _net(b2))
        return y

    def encoder(self, x, w12):
        y = np.empty_like(x)
        ref_lib().Encoder(_p(x), _p(y), *[_net(w) for w in w12])
        return y

    def softmax(self, logits):
        p = np.empty_like(logits)
        ref_lib().Softmax(_p(logits), _p(p), C.c_int(logits.size))
        return p

    def conv2d(self, image, w, b):
        out = np.empty((768, 14, 14), np.float32)
        ref_lib().Conv2d(_p(image), _p(out), _net(w), _net(b))
        return out

    def embed(self, image, weights):
        L = ref_lib()
        conv = self.conv2d(image, weights[1], weights[2])
        flat = np.empty((196, 768), np.float32)
        L.flatten_transpose(_p(conv), _p(flat))
        tok = np.empty((197, 768), np.float32)
        L.class_token(_p(flat), _p(tok), _net(weights[0]))
        out = np.empty_like(tok)
        L.pos_emb(_p(tok), _p(out), _net(weights[3]))
        return out

    def vit_seq(self, images, weights):
        """The reference's ViT_seq(ImageData*, Network*, float**) entry (ViT_seq.c:337)."""
        n = len(images)
        imgs = (_RefImage * n)()
        keep = []
        for i, im in enumerate(images):
            im = np.ascontiguousarray(im, np.float32)
            keep.append(im)
            imgs[i] = _RefImage(n, 3, 224, 224, _p(im))
        nets = _tensors(weights)
        probs = np.empty((n, 1000), np.float32)
        rows = (_f32p * n)(*[_p(probs[i]) for i in range(n)])
        ref_lib().ViT_seq(imgs, nets, rows)
        return probs
