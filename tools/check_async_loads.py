#!/usr/bin/env python3
"""Static check of the ping-pong bf16 GEMM's inline-asm residual loads (csrc/vit_gemm_bf16_pp.hip).

The fp32-output epilogues issue `global_load_dwordx4` from inline asm and wait for it later with a counted
`s_waitcnt vmcnt(N)` (hipcc would otherwise drain the LDS-DMA queue).  The compiler does not know the destination
registers are in flight, so any instruction it places between the load and the next wait that touches them would
read stale data (this happened once on predicated edge tiles, now loaded synchronously).  This script disassembles
the kernels and reports such uses.    python tools/check_async_loads.py      (needs hipcc; no GPU)
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "vision-transformer-opencl_amd", "csrc", "vit_gemm_bf16_pp.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "pp.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-Wno-unused-value", "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
bad = 0
for st in [i for i, l in enumerate(lines) if re.match(r"^_ZN7vitgemm\S*gemm_bf16_pp_kernel\S*:", l)]:
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body, pending, i, nloads, issues = lines[st:end], set(), 0, 0, 0
    while i < len(body):
        if "ASMSTART" in body[i]:
            j = i + 1
            while "ASMEND" not in body[j]:
                j += 1
            t = " ".join(body[i + 1:j])
            if "global_load_dwordx4" in t and "s_waitcnt" not in t:
                r = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", t)
                pending |= set(range(int(r.group(1)), int(r.group(2)) + 1))
                nloads += 1
            elif "s_waitcnt vmcnt" in t:
                pending = set()
            i = j + 1
            continue
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", body[i]):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", body[i]))
        if regs & pending and not body[i].strip().startswith(";"):
            issues += 1
            print("  in-flight register touched:", body[i].strip())
        i += 1
    m = re.search(r"kernelILi(\d+)ELi(\d+)ELi(\d+)E", lines[st])
    print(f"EPI {m.group(1)} STAMP {m.group(2)} DBG {m.group(3)}: {nloads} asynchronous loads, {issues} hazards")
    bad += issues
sys.exit(1 if bad else 0)
