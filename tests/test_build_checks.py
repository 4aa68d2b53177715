"""Build-time checks that need the compiler but no GPU."""
import importlib.util
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_inline_asm.py")
pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


def _lint():
    spec = importlib.util.spec_from_file_location("check_inline_asm", TOOL)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    """Every csrc/*.hip with `asm volatile`, compiled to gfx950 assembly once (4 at a time) and linted as it stands."""
    d = str(tmp_path_factory.mktemp("asm"))
    r = subprocess.run([sys.executable, TOOL, "--keep", d], capture_output=True, text=True, timeout=900)
    return d, r


def test_every_inline_asm_site_keeps_its_hazard_and_wait_invariants(listings):
    """tools/check_inline_asm.py over every kernel file with inline assembly: (i) no write of a wide asm store's data registers
    within two wait states, (ii) no use of an asm load's destination before the counted wait that retires it (rounds 1-4 checked this for one file and one instruction only), (iii) every asm LDS-DMA sets M0 in its own block, (iv) no compiler-made 16-byte buffer store with a register in soffset in front of a write of its data."""
    d, r = listings
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    for f in ("vit_gemm_bf16_pp.s", "vit_gemm_persistent.s", "vit_attention_stream.s"):
        assert f in out, out
    m = re.search(r"vit_gemm_bf16_pp\.s: (\d+) kernels, asm sites: (\d+) wide stores, (\d+) register loads", out)
    assert m and int(m.group(2)) > 100 and int(m.group(3)) > 500, out          # the sites are really seen
    assert re.search(r"vit_gemm_persistent\.s: .* (\d+) LDS-DMA", out).group(1) != "0", out
    assert re.search(r"vit_attention_stream\.s: .* (\d+) LDS-DMA", out).group(1) != "0", out


def _asm_blocks(lines):
    """(start, end) line indices of ;;#ASMSTART ... ;;#ASMEND blocks."""
    out, st = [], None
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            st = i
        elif "#ASMEND" in l and st is not None:
            out.append((st, i))
            st = None
    return out


def test_the_lint_fails_on_deliberately_broken_sites(listings):
    """Each rule once, on a mutated copy of the generated assembly: a wide store whose `s_nop 1` is gone and whose data register
    is written in the next slot; an asynchronous load whose destination is read before the wait; an LDS-DMA whose `s_nop 0`
    behind the M0 write is gone.  HEAD passes (test above), every mutant must be reported."""
    d, r = listings
    assert r.returncode == 0, r.stdout + r.stderr
    lint = _lint()
    lines = open(os.path.join(d, "vit_gemm_bf16_pp.s")).read().split("\n")
    assert lint.check_text("\n".join(lines))[0] == []

    # (i) store: drop the pad, write the first data register right behind the block
    blk = next((s, e) for s, e in _asm_blocks(lines) if any("global_store_dwordx4" in l for l in lines[s:e]) and any("s_nop 1" in l for l in lines[s:e]))
    store = next(l for l in lines[blk[0]:blk[1]] if "global_store_dwordx4" in l)
    reg = re.search(r"global_store_dwordx4 v\d+, v\[(\d+):\d+\]", store).group(1)
    mut = lines[:blk[0]] + [l for l in lines[blk[0]:blk[1] + 1] if "s_nop 1" not in l] + [f"\tv_mov_b32_e32 v{reg}, 0"] + lines[blk[1] + 1:]
    found = lint.check_text("\n".join(mut))[0]
    assert len(found) == 1 and "writes data registers of the asm store" in found[0], found
    # ... and the pad alone is enough: with the s_nop kept the same write is legal
    ok = lines[:blk[1] + 1] + [f"\tv_mov_b32_e32 v{reg}, 0"] + lines[blk[1] + 1:]
    assert lint.check_text("\n".join(ok))[0] == []

    # (ii) load: read the destination in the slot behind an asynchronous residual load
    blk = next((s, e) for s, e in _asm_blocks(lines) if any("global_load_dwordx4 v[" in l for l in lines[s:e]) and not any("s_waitcnt" in l for l in lines[s:e]))
    reg = re.search(r"global_load_dwordx4 v\[(\d+):", next(l for l in lines[blk[0]:blk[1]] if "global_load_dwordx4" in l)).group(1)
    mut = lines[:blk[1] + 1] + [f"\tv_add_f32_e32 v0, v{reg}, v0"] + lines[blk[1] + 1:]
    found = lint.check_text("\n".join(mut))[0]
    assert len(found) == 1 and "still in flight" in found[0], found

    # (iv) a compiler-made 16-byte buffer store with a REGISTER in soffset (the form hipcc does not pad: it cost the streamed attention
    # its rows 12-15 / 28-31 in round 5), data register written in the next slot
    alines = open(os.path.join(d, "vit_attention_stream.s")).read().split("\n")
    k = next(i for i, l in enumerate(alines) if re.search(r"buffer_store_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\], 0 offen", l))
    reg = re.search(r"buffer_store_dwordx4 v\[(\d+):", alines[k]).group(1)
    mut = alines[:k] + [re.sub(r", 0 offen", ", s44 offen", alines[k], count=1), f"\tv_mul_f32_e32 v{reg}, v1, v1"] + alines[k + 1:]
    found = lint.check_text("\n".join(mut))[0]
    assert len(found) == 1 and "compiler-made store" in found[0], found
    same_with_immediate = alines[:k + 1] + [f"\tv_mul_f32_e32 v{reg}, v1, v1"] + alines[k + 1:]   # soffset 0: hipcc's own padding applies
    assert lint.check_text("\n".join(same_with_immediate))[0] == []

    # (iii) LDS-DMA of the fp32 persistent GEMM (the rows' pairs of the LayerNorm fold): drop the s_nop behind the M0 write
    lines = open(os.path.join(d, "vit_gemm_persistent.s")).read().split("\n")
    blk = next((s, e) for s, e in _asm_blocks(lines) if any(" lds" in l and "buffer_load" in l for l in lines[s:e]))
    mut = lines[:blk[0]] + [l for l in lines[blk[0]:blk[1] + 1] if "s_nop" not in l] + lines[blk[1] + 1:]
    found = lint.check_text("\n".join(mut))[0]
    assert len(found) == 1 and "LDS-DMA" in found[0], found


def test_streamed_attention_vector_memory_operations_are_the_ones_its_counted_waits_assume(listings):
    """csrc/vit_attention_stream.hip proves that a head's Q refills have landed with COUNTED `s_waitcnt vmcnt(N)`: N is computed from
    ST_QDMA = 4 LDS-DMA pieces and ST_STORES = 4 output stores per retired query block, in a fixed order per wave.  That arithmetic
    holds only while (a) every output store is one `buffer_store_dwordx4` -- 4 per block, 3 blocks, 2 copies of the retire code (chunks
    with one / two sub-chunks) = 24 per kernel, and no other store instruction at all; (b) every load is an inline-asm LDS-DMA (no
    register load whose wait hipcc would place itself); (c) hipcc itself waits for nothing: no `vmcnt` outside the asm blocks, no
    scratch (a spill reload is a vector load behind a `vmcnt(0)`).  Checked on the generated assembly of both instantiations."""
    d, r = listings
    assert r.returncode == 0, r.stdout + r.stderr
    lint = _lint()
    text = open(os.path.join(d, "vit_attention_stream.s")).read()
    kernels = [(n, b) for n, b in lint.kernels(text) if "attention_bf16_stream_kernel" in n]
    assert len(kernels) == 2
    src = open(os.path.join(ROOT, "vision-transformer-opencl_amd", "csrc", "vit_attention_stream.hip")).read()
    assert "constexpr int ST_QDMA = 4, ST_STORES = 4;" in src and "constexpr int MAXB = 3;" in src
    for name, body in kernels:
        stores = [i for i in body if lint.VM_STORE.match(i.op)]
        assert len(stores) == 24 and all(i.op == "buffer_store_dwordx4" and not i.in_asm for i in stores), (name, len(stores))
        loads = [i for i in body if lint.VM_LOAD.match(i.op)]
        assert loads and all(i.in_asm and " lds" in i.text for i in loads), name            # LDS-DMA only, all inline asm
        assert not [i for i in body if i.op.startswith("scratch_")], name
        assert not [i for i in body if i.op == "s_waitcnt" and "vmcnt" in i.text and not i.in_asm], name
