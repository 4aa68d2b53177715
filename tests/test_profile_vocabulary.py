"""bench.py replays counter figures from profiles/<round>/ for its dominant kernel: the kernel names it derives from the engine's stage
brackets (STAGE_KERNEL / STAGE_KERNEL_FOLD) must be the names tools/pmc_summary.py gives the profiler's demangled instantiations, and
the committed profile set must hold an entry for every GEMM kernel of the headline configuration.  No GPU."""
import csv
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_bench_kernel_names_match_the_profile_summaries():
    pmc = _load("pmc_summary_under_test", os.path.join(ROOT, "tools", "pmc_summary.py"))
    bench = _load("bench_under_test", os.path.join(ROOT, "bench.py"))
    # demangled instantiation -> bench vocabulary (template argument 4 of the persistent walk is its epilogue code)
    for code, name in ((0, "EPI_BIAS"), (1, "EPI_BIAS_GELU"), (2, "EPI_BIAS_RESIDUAL"), (3, "EPI_BIAS_LN"), (4, "EPI_BIAS_GELU_LN"), (5, "EPI_RESIDUAL_STATS")):
        short = pmc.short_name(f"void vitgemm::gemm_f32_nt_persistent_kernel<128, 128, 64, 64, {code}, false, true, 0>(vitgemm::GemmParams)")
        assert pmc.bench_key(short) == f"gemm_f32_nt_persistent_kernel<{name}>"
    assert pmc.bench_key(pmc.short_name("void (anonymous namespace)::gemm_f32_nt_kernel<128, 64, 64, 32, 0, 1, 0, 32, true>(vitgemm::GemmParams)")) == "gemm_f32_nt_kernel<A_PATCHES>"
    assert pmc.bench_key("gemm_bf16_pp_kernel<2, 0, 0, true>") == "gemm_bf16_pp_kernel<F32_RESIDUAL>"
    # every GEMM stage of the headline configuration (fold on: the default) has its figures in the committed set
    rel = os.path.join(ROOT, "profiles", bench.PROFILE_ROUND)
    traffic = json.load(open(os.path.join(rel, "hbm_traffic_pmc_f32.json")))
    assert traffic["batch"] == bench.PROFILED_BATCH["f32"]
    with open(os.path.join(rel, "mfma_util_f32.csv"), newline="") as f:
        busy = {pmc.bench_key(r["kernel"]) for r in csv.DictReader(f)}
    for stage in ("qkv", "fc1", "outproj", "fc2", "attn", "embed"):
        k = bench.STAGE_KERNEL_FOLD[stage]
        assert k in traffic["kernels"], (stage, k)
        assert k in busy, (stage, k)
    # ... and the replay helpers find them
    dom = bench.STAGE_KERNEL_FOLD["fc2"]
    t, src = bench.pmc_traffic(dom, "f32", 256)
    assert t and t > 0 and src["path"].endswith("hbm_traffic_pmc_f32.json")
    b, clk = bench.pmc_mfma(dom, "f32", 256)
    assert 50.0 < b <= 100.0 and 1.0 < clk < 3.0


def test_other_configs_replay_their_counters_too():
    """BASELINE.json configs[2] and [4] in `other_configs`: the dominant bf16 kernel's HBM bytes, MFMA-busy share and clock come from
    the committed bf16 / ViT-L passes exactly as the headline's do, marked as replayed."""
    bench = _load("bench_under_test2", os.path.join(ROOT, "bench.py"))
    rel = os.path.join(ROOT, "profiles", bench.PROFILE_ROUND)
    for tag, batch in (("bf16", 2048), ("bf16_l16_384", 1024)):
        assert bench.PROFILED_BATCH[tag] == batch
        traffic = json.load(open(os.path.join(rel, f"hbm_traffic_pmc_{tag}.json")))
        assert traffic["batch"] == batch
        for k in ("gemm_bf16_pp_kernel<BF16>", "gemm_bf16_pp_kernel<BF16_GELU>", "gemm_bf16_pp_kernel<F32_RESIDUAL>"):
            assert k in traffic["kernels"], (tag, k)
        c = bench.replayed_counters("gemm_bf16_pp_kernel<F32_RESIDUAL>", tag, batch)
        assert c["traffic"] and c["traffic"] > 1e9
        assert 20.0 < c["mfma_busy_percent_rocprof"] <= 100.0 and 1.0 < c["clock_ghz_rocprof"] < 3.0
        assert c["replayed_from"]["live"] is False and set(c["replayed_from"]["fields"]) == {"traffic", "mfma_busy_percent_rocprof", "clock_ghz_rocprof"}
        assert c["replayed_from"]["path"].endswith(f"hbm_traffic_pmc_{tag}.json")
    # a batch the passes were not collected at replays nothing
    assert bench.replayed_counters("gemm_bf16_pp_kernel<F32_RESIDUAL>", "bf16", 512) == {"traffic": None, "mfma_busy_percent_rocprof": None, "clock_ghz_rocprof": None, "replayed_from": None}
