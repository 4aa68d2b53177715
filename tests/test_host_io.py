"""Host side of the boundary (no GPU): loaders, 1e-6 rounding, result writer, comparator, the
synthetic-tensor generator, and that the C-ABI library exports every symbol include/*.h declares.

Reference behaviour being mirrored: Network.c:24-97,119-194 (formats + rounding),
Main.c:62-72 (result lines), comparator.c:23-80 (label equal and |dprob| <= 0.01).
"""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from vit_amd import binding as B
from vit_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    L = B.lib()
    declared, declared_dp = set(), set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            (declared_dp if os.path.basename(h) == "vit_dp.h" else declared).add(m.group(1))
    assert {"ViT_hip", "ViT_opencl", "initialize_opencl", "Release_opencl", "load_weights", "load_image_data",
            "comparator", "vithip_gemm_f32", "vit_engine_forward_device"} <= declared
    missing = [name for name in sorted(declared) if not hasattr(L, name)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    import shutil
    import subprocess
    if shutil.which("readelf"):
        needed = subprocess.run(["readelf", "-d", B.LIB_PATH], capture_output=True, text=True, check=True).stdout
        assert "rccl" not in needed and "nccl" not in needed      # the forward library itself has no collective dependency
    # include/vit_dp.h is the surface of libvit_mi355x_dp.so (the RCCL gather: the only library that links a collective library).
    # That library is an optional target of the Makefile (built where $(ROCM)/include/rccl/rccl.h exists): absent, its checks skip.
    assert {"vit_dp_create", "vit_dp_gather_top1", "vit_dp_destroy"} <= declared_dp
    if not os.path.exists(B.DP_LIB_PATH):
        pytest.skip("libvit_mi355x_dp.so not built on this box (no RCCL): the forward library's exports were checked")
    Ldp = B.dp_lib()
    assert not [name for name in sorted(declared_dp) if not hasattr(Ldp, name)]
    if shutil.which("readelf"):
        assert "librccl" in subprocess.run(["readelf", "-d", B.DP_LIB_PATH], capture_output=True, text=True, check=True).stdout


def test_product_library_has_no_probe_entry_points_or_tuning_setters():
    """Instrumented kernels and process-wide tuning overrides live in the probe build only (csrc/vit_probes.h,
    `make probes` -> libvit_mi355x_probe.so); the product library keeps no mutable process-wide state."""
    import subprocess
    product = os.path.join(os.path.dirname(B.LIB_PATH), "libvit_mi355x.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", product], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1] for ln in syms.splitlines() if ln.strip()}
    banned = [n for n in names if n.startswith("vithip_probe_") or n.endswith("_set_debug_buffer") or
              re.fullmatch(r"vithip_\w+_set_(tile|group|variant|sync|stagger|max_workgroups|mfma)", n)]
    assert "vithip_gemm_f32" in names and "vithip_gemm_bf16" in names
    assert not banned, banned
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "vision-transformer-opencl_amd", "csrc", "vit_probes.h")).read(), flags=re.S)
    probe_decl = set(re.findall(r"^int (\w+)\(", text, flags=re.M))
    assert probe_decl and not (probe_decl & names)


def test_rounding_is_c_roundf_half_away_from_zero():
    # ties at +-x.5e-6 must move away from zero (Network.c:185 uses roundf, not rint)
    x = np.array([0.5e-6, -0.5e-6, 1.5e-6, 2.5e-6, -2.5e-6, 0.1234564, 0.1234565, 0.1234566, 3.0, -7.25], np.float32)
    got = B.round_weights(x)
    s = (x * np.float32(1e6)).astype(np.float64)
    want = (np.copysign(np.floor(np.abs(s) + 0.5), s).astype(np.float32) / np.float32(1e6)).astype(np.float32)
    assert np.array_equal(got, want)
    big = synth.uniform(3, 1, 200000, -0.6, 0.6)
    assert np.array_equal(B.round_weights(big), synth.round6(big))


def test_weight_loader_layout_rounding_and_gaps(tmp_path):
    cfg = synth.VIT_TINY
    raw = [synth.uniform(9, i, int(np.prod(s)), -0.3, 0.3).reshape(s) for i, s in enumerate(cfg.weight_shapes())]
    names = [f"layer_{i}_weight" for i in range(cfg.n_weights)]
    synth.write_weight_files(str(tmp_path), raw, names)
    os.remove(tmp_path / "Weight_6_layer_6_weight.bin")            # a missing blob stays {NULL, 0}
    (tmp_path / "Weight_notanumber.bin").write_bytes(b"xxxx")       # no second '_': ignored
    (tmp_path / "Weight_999_too_big.bin").write_bytes(b"\0" * 8)    # index >= count: ignored
    (tmp_path / "Other_3_file.bin").write_bytes(b"\0" * 8)          # wrong prefix: ignored
    (tmp_path / "Weight_2_wrong.ext").write_bytes(b"\0" * 8)        # wrong extension: ignored
    got = B.load_weight_dir(str(tmp_path), cfg.n_weights)
    assert got[6] is None
    for i, w in enumerate(raw):
        if i == 6:
            continue
        assert got[i].size == w.size
        assert np.array_equal(got[i], synth.round6(w).ravel()), f"tensor {i}"


def _tiny_blobs(tmp_path, seed=9):
    cfg = synth.VIT_TINY
    raw = [synth.uniform(seed, i, int(np.prod(s)), -0.3, 0.3).reshape(s) for i, s in enumerate(cfg.weight_shapes())]
    synth.write_weight_files(str(tmp_path), raw)
    return cfg, raw


def test_packed_weight_cache_roundtrip_and_staleness(tmp_path):
    cfg, raw = _tiny_blobs(tmp_path)
    cache = tmp_path / "vit_weights.cache"
    first = B.load_weight_dir_cached(cfg, str(tmp_path))              # scans the files, writes the cache
    assert cache.exists()
    for i, a in enumerate(first):
        assert np.array_equal(a, synth.round6(raw[i]).ravel())
    # served from the cache: make the blobs unreadable-as-weights without touching name/size/mtime
    keep = {}
    for f in glob.glob(str(tmp_path / "Weight_*.bin")):
        st = os.stat(f)
        keep[f] = (open(f, "rb").read(), st)
        open(f, "r+b").write(b"\xff" * st.st_size)
        os.utime(f, ns=(st.st_atime_ns, st.st_mtime_ns))
    second = B.load_weight_dir_cached(cfg, str(tmp_path))
    for a, b in zip(first, second):
        assert np.array_equal(a, b)
    for f, (data, st) in keep.items():
        open(f, "r+b").write(data)
        os.utime(f, ns=(st.st_atime_ns, st.st_mtime_ns))
    # a replaced blob (new mtime) makes the cache stale: the directory wins and the cache is rewritten
    new7 = synth.uniform(77, 7, raw[7].size, -0.3, 0.3).astype(np.float32)
    name7 = glob.glob(str(tmp_path / "Weight_7_*.bin"))[0]
    new7.tofile(name7)
    os.utime(name7, ns=(1_700_000_000_000_000_000, 1_700_000_000_000_000_000))
    third = B.load_weight_dir_cached(cfg, str(tmp_path))
    assert np.array_equal(third[7], synth.round6(new7)) and not np.array_equal(third[7], first[7])
    fourth = B.load_weight_dir_cached(cfg, str(tmp_path))             # ... and the rewritten cache is valid again
    assert np.array_equal(fourth[7], third[7])
    cache.write_bytes(b"garbage")                                     # a corrupt cache is ignored, not trusted
    fifth = B.load_weight_dir_cached(cfg, str(tmp_path))
    assert np.array_equal(fifth[7], third[7])


def test_packed_weight_cache_is_not_written_from_an_incomplete_set(tmp_path):
    cfg, raw = _tiny_blobs(tmp_path)
    os.remove(glob.glob(str(tmp_path / "Weight_9_*.bin"))[0])         # the reference tree itself ships 116 of 152 blobs
    got = B.load_weight_dir_cached(cfg, str(tmp_path))
    assert got[9] is None and got[8] is not None
    assert not (tmp_path / "vit_weights.cache").exists()              # holes are never pinned into a cache
    synth.write_weight_files(str(tmp_path), raw)                      # the blob arrives later: picked up at once
    got = B.load_weight_dir_cached(cfg, str(tmp_path))
    assert got[9] is not None and (tmp_path / "vit_weights.cache").exists()
    # a cache that predates an added blob of an index it does not know is stale as well
    (tmp_path / "Weight_3_extra_copy.bin").write_bytes(raw[3].astype(np.float32).tobytes())
    assert B.WeightImage.load(cfg, str(tmp_path / "vit_weights.cache"), str(tmp_path)) is None
    assert B.WeightImage.load(cfg, str(tmp_path / "vit_weights.cache"), None) is not None   # no directory: format check only


def test_weight_image_is_the_device_layout(tmp_path):
    cfg = synth.VIT_SMALL
    W = synth.make_weights(cfg, 3)
    img = B.WeightImage.build(cfg, W, with_bf16=True)
    c = img.c
    gemm = [i for i in range(cfg.n_weights) if i == 1 or (4 <= i < 4 + 12 * cfg.depth and (i - 4) % 12 in (2, 4, 8, 10))]
    offs = [c.off[i] for i in range(cfg.n_weights)]
    assert all(o % 128 == 0 for o in offs)                            # 512-B fp32 slots = 256-B bf16 slots
    assert sorted(offs[i] for i in gemm) == sorted(offs)[:len(gemm)]  # GEMM operands lead the blob ...
    assert c.gemm_floats == max(offs[i] + -(-W[i].size // 128) * 128 for i in gemm) == c.bf16_elems
    for i, t in enumerate(img.tensors()):
        assert np.array_equal(t, W[i].ravel())
    f32, b16 = img.f32_section(), img.bf16_section()
    assert np.array_equal(b16, B.to_bf16_bits(f32[:c.gemm_floats]))   # ... and the bf16 section mirrors them, RNE
    path = str(tmp_path / "w.cache")
    assert img.save(path) == 0
    back = B.WeightImage.load(cfg, path)
    assert np.array_equal(back.f32_section(), f32) and np.array_equal(back.bf16_section(), b16)
    assert B.WeightImage.load(synth.VIT_TINY, path) is None           # another model's cache is refused
    with pytest.raises(B.VitError):
        B.WeightImage.build(cfg, W[:-1] + [None])


def test_chunked_image_reader_matches_whole_file_loader(tmp_path):
    cfg = synth.VIT_TINY
    imgs = synth.make_images(cfg, 7, 5)
    path = str(tmp_path / "input-7.bin")
    synth.write_image_file(path, imgs)
    n, chunks = B.read_image_file_chunked(path, 3)
    assert n == 7 and [f for f, _ in chunks] == [0, 3, 6] and [c.shape[0] for _, c in chunks] == [3, 3, 1]
    assert np.array_equal(np.concatenate([c for _, c in chunks]), imgs)
    assert B.read_image_file_chunked(str(tmp_path / "missing.bin"), 3) is None
    (tmp_path / "short.bin").write_bytes(open(path, "rb").read()[:-8])
    n, chunks = B.read_image_file_chunked(str(tmp_path / "short.bin"), 4)
    assert [c.shape[0] for _, c in chunks] == [4]                     # the torn last chunk is refused, not returned short


def test_image_loader_roundtrip_and_failures(tmp_path):
    cfg = synth.VIT_TINY
    imgs = synth.make_images(cfg, 3, 5)
    path = str(tmp_path / "input-3.bin")
    synth.write_image_file(path, imgs)
    back = B.load_image_file(path)
    assert back.shape == imgs.shape and np.array_equal(back, imgs)
    assert B.load_image_file(str(tmp_path / "missing.bin")) is None           # perror + NULL
    (tmp_path / "short.bin").write_bytes(open(path, "rb").read()[:-8])
    assert B.load_image_file(str(tmp_path / "short.bin")) is None              # short read -> NULL
    (tmp_path / "hdr.bin").write_bytes(b"\x01\x00")
    assert B.load_image_file(str(tmp_path / "hdr.bin")) is None


def test_result_lines_and_reference_argmax_quirk(tmp_path):
    probs = np.full((3, 1000), 1e-4, np.float32)
    probs[0, 65] = 0.919345
    probs[1, 0] = 0.5          # class 0 wins image 1 ...
    probs[2, 230] = 0.685105
    fixed, quirk = str(tmp_path / "fixed.txt"), str(tmp_path / "quirk.txt")
    assert B.write_results(fixed, probs, fix_argmax=True) == 0
    assert open(fixed).read().splitlines() == ["[0] label: 65 / prob: 0.919345", "[1] label: 0 / prob: 0.500000",
                                               "[2] label: 230 / prob: 0.685105"]
    # ... but Main.c:62 never resets pred_idx and the scan starts at j = 1: the reference's writer enters
    # image 1 still pointing at class 65 and never looks at class 0 (latent bug for n > 1)
    assert B.write_results(quirk, probs, fix_argmax=False) == 0
    assert open(quirk).read().splitlines()[1] == "[1] label: 65 / prob: 0.000100"
    probs[1, 0] = 1e-4
    probs[1, 3] = 0.3
    B.write_results(quirk, probs, fix_argmax=False)
    assert open(quirk).read().splitlines()[1] == "[1] label: 3 / prob: 0.300000"


def test_comparator_semantics(tmp_path):
    ans = os.path.join(GOLD, "answer_result.txt")
    assert B.compare_results(ans, ans, 100) == 0
    # the reference's own committed OpenCL output differs by 0.00133 on line 0: inside the 0.01 tolerance
    assert B.compare_results(os.path.join(GOLD, "reference_opencl_result.txt"), ans, 1) == 0
    lines = open(ans).read().splitlines()
    bad = list(lines)
    bad[1] = "[1] label: 7 / prob: 0.824735"       # label mismatch          -> +1
    bad[2] = "[2] label: 230 / prob: 0.700000"     # |dprob| = 0.0149 > 0.01 -> +1
    bad[3] = "garbage"                              # parse error             -> +1
    p = tmp_path / "bad.txt"
    p.write_text("\n".join(bad) + "\n")
    assert B.compare_results(str(p), ans, 100) == 3
    assert B.compare_results(str(p), ans, 1) == 0                       # IMAGE_COUNT 1 (comparator.c:8)
    assert B.compare_results(str(tmp_path / "nope.txt"), ans, 1) == 1   # unreadable file counts as one difference
    short = tmp_path / "short.txt"
    short.write_text("\n".join(lines[:5]) + "\n")
    assert B.compare_results(str(short), ans, 10) == 1                  # ran out of lines


def test_synthetic_generator_c_equals_numpy():
    for lo, hi in ((-0.035, 0.035), (0.5, 1.0), (-2.1, 2.6)):
        assert np.array_equal(B.synth_uniform(11, 4, 10007, lo, hi), synth.uniform(11, 4, 10007, lo, hi))
    cfg = synth.VIT_SMALL
    want = synth.make_weights(cfg, 21, native=False)
    for a, b in zip(B.synth_weights_c(cfg, 21), want):
        assert np.array_equal(a, b)
    # vit_synth_weights() itself (what vit_main and a C caller use): the same tensors in a Network[] it allocates
    L = B.lib()
    nets = (B.CNetwork * cfg.n_weights)()
    assert L.vit_synth_weights(C.byref(B.CConfig.of(cfg)), 21, nets, cfg.n_weights) == 0
    for i, b in enumerate(want):
        assert nets[i].size == b.size and np.array_equal(np.ctypeslib.as_array(nets[i].data, shape=(b.size,)), b.ravel())
    L.free_weights(nets, cfg.n_weights)
    assert np.array_equal(B.synth_images_c(cfg, 2, 9), synth.make_images(cfg, 2, 9))


def test_model_constants():
    L = B.lib()
    cc = B.CConfig.of(synth.VIT_B16)
    assert L.vit_config_macs_per_image(C.byref(cc)) == 17_563_828_224 == synth.VIT_B16.macs_per_image
    assert synth.VIT_L16_384.macs_per_image == 191_066_300_416
    # prune_last_layer: Q projection, out_proj, fc1, fc2 and both attention products of 196 of the 197 rows of ONE layer
    T, D, H, heads, hd = 197, 768, 3072, 12, 64
    saved = (T - 1) * (D * D + 2 * heads * T * hd + D * D + 2 * D * H)
    assert L.vit_config_macs_per_image_pruned(C.byref(cc)) == 17_563_828_224 - saved
    sizes = [L.vit_config_weight_size(C.byref(cc), i) for i in range(152)]
    assert sum(sizes) == 86_567_656                       # torchvision vit_b_16 parameter count
    assert sizes == [int(np.prod(s)) for s in synth.VIT_B16.weight_shapes()]
    assert L.vit_config_weight_size(C.byref(cc), 152) == 0
    b16 = L.vit_config_b16()
    assert (b16.img_size, b16.embed_dim, b16.depth, b16.hidden_dim) == (224, 768, 12, 3072)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(B, "_lib", None)
    monkeypatch.setattr(B, "LIB_PATH", "/nonexistent/libvit_mi355x.so")
    with pytest.raises(B.VitError, match="no CPU fallback"):
        B.lib()
