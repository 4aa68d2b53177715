#!/usr/bin/env python3
"""ISA-level lint of csrc/*.hip: every `asm volatile` site, and one compiler-made hazard (CPU only: hipcc -S for gfx950, no GPU).

hipcc treats an asm statement as one opaque instruction: it neither counts the memory operations inside nor pads
their hazards (CDNA4 guide 5.7).  The kernels here rely on three hand-kept invariants -- and on one thing hipcc itself gets wrong, (iv) --; this script
re-derives each from the generated assembly, so the next edit (or the next register allocation) cannot silently break them:

  (i)   STORE DATA HAZARD.  A vector-memory store of more than 8 bytes keeps reading its data registers for two wait
        states after issue (gfx940+: LLVM's hazard recognizer pads 2 for its own stores, nothing behind `;;#ASMEND`).
        For every such store inside an asm block, no instruction within the next two wait states may write a register
        of its data operand (`s_nop N` = N + 1 states; the shipped stores end in `s_nop 1`).
  (ii)  IN-FLIGHT LOAD DESTINATIONS.  The destination VGPRs of a load issued inside an asm block (global_/buffer_ loads
        without `lds`, ds_read) are undefined until the wait that retires it.  vmcnt and lgkmcnt are modelled as
        in-order queues over ALL instructions of the kernel (asm or compiler-made; stores and LDS-DMA count on vmcnt):
        `s_waitcnt vmcnt(N)` retires everything but the N youngest.  Any instruction that names a pending register
        before that is reported.  (Straight-line model: the text order of the unrolled epilogues these loads live in.)
  (iii) LDS-DMA THROUGH M0.  An asm `buffer_load ... lds` / `global_load_lds_*` takes its LDS base from M0, which the
        compiler does not preserve: the same asm block must write M0 first, with at least one wait state (`s_nop 0`)
        between the SALU write and the load.

  (iv)  WIDE BUFFER STORE WITH A SCALAR OFFSET (compiler-made code: found in round 5).  LLVM's hazard recognizer pads the two
        wait states of (i) for its own stores EXCEPT for buffer stores whose soffset is a register; on gfx950 that form has the
        hazard too (a v_mul in the next slot overwrote the data of the lanes read last).  Reported like (i), for every
        `buffer_store_dwordx3/x4 ..., sN offen` outside asm.

    python tools/check_inline_asm.py            # every csrc/*.hip; exit 1 on any finding
    python tools/check_inline_asm.py --asm f.s  # lint an assembly file as it is (what the self-test mutates)
"""
from __future__ import annotations

import argparse, concurrent.futures, glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vision-transformer-opencl_amd", "csrc")
PER_FILE_FLAGS = {"vit_attention_stream": ["-fno-slp-vectorize"]}  # = FLAGS_* of the Makefile

VM_STORE = re.compile(r"^(global|buffer|flat|scratch)_store_")
VM_LOAD = re.compile(r"^(global|buffer|flat|scratch)_load_")
VM_ATOMIC = re.compile(r"^(global|buffer|flat)_atomic_")
DS_OP = re.compile(r"^ds_")
SMEM = re.compile(r"^(s_load_|s_buffer_load_|s_memtime|s_memrealtime|s_store_|s_dcache)")
WIDE = re.compile(r"dwordx[34]\b|_b96\b|_b128\b")


def regs_of(operand: str, bank: str = "v") -> set[int]:
    """Registers of bank `bank` ('v' or 'a') named by one operand: v7, v[4:7], a[0:3]."""
    out: set[int] = set()
    for a, b in re.findall(r"\b%s\[(\d+):(\d+)\]" % bank, operand):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\b%s(\d+)\b" % bank, operand))
    return out


class Inst:
    __slots__ = ("line", "text", "op", "operands", "in_asm", "block")

    def __init__(self, line, text, in_asm, block):
        self.line, self.text, self.in_asm, self.block = line, text, in_asm, block
        parts = text.split(None, 1)
        self.op = parts[0]
        self.operands = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []

    def vregs(self) -> set[int]:
        return regs_of(" ".join(self.operands))

    def written_vregs(self) -> set[int]:
        """VGPRs this instruction writes (destination = first operand; swaps write both; stores / waits / scalar ops none)."""
        op = self.op
        if VM_STORE.match(op) or op.startswith(("ds_write", "s_", "v_cmp", "v_readlane", "v_readfirstlane", "buffer_wbl2", "buffer_inv")):
            return set()
        if " lds" in self.text and VM_LOAD.match(op):
            return set()
        if not self.operands:
            return set()
        if op.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")):
            return regs_of(self.operands[0]) | regs_of(self.operands[1])
        return regs_of(self.operands[0])

    def wait_states(self) -> int:
        if self.op == "s_nop":
            return int(self.operands[0], 0) + 1
        return 1


def kernels(text: str):
    """[(name, [Inst])] for every kernel (function ending in s_endpgm) of an AMDGPU assembly listing."""
    out, name, body, in_asm, block = [], None, [], False, 0
    for n, raw in enumerate(text.split("\n"), 1):
        s = raw.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", raw)
        if m and not m.group(1).startswith(".L"):
            name, body, in_asm = m.group(1), [], False
            continue
        if name is None or not s:
            continue
        if "#ASMSTART" in s:
            in_asm, block = True, block + 1
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if s.startswith((";", ".", "//")) or re.match(r"^[.\w$]+:", s):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        body.append(Inst(n, s, in_asm, block))
        if s.startswith("s_endpgm"):
            out.append((name, body))
            name, body = None, []
    return out


def waitcnt_fields(inst: Inst):
    """(vmcnt, lgkmcnt) of an s_waitcnt, None where the field is absent (= not waited for)."""
    t = inst.text
    vm = re.search(r"vmcnt\((\d+)\)", t)
    lg = re.search(r"lgkmcnt\((\d+)\)", t)
    if not vm and not lg and re.match(r"^s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", t):  # raw immediate: gfx9 layout
        imm = int(t.split()[1], 0)
        return (imm & 0xF) | ((imm >> 14) & 0x3) << 4, (imm >> 8) & 0xF
    return (int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None)


def check_kernel(name: str, body: list[Inst]) -> list[str]:
    found: list[str] = []
    short = name if len(name) < 90 else name[:87] + "..."

    # ---- (i) wide asm stores: two wait states before a write of the data registers
    for i, ins in enumerate(body):
        if not (VM_STORE.match(ins.op) and WIDE.search(ins.op)):
            continue
        if not ins.in_asm:  # (iv): the compiler pads its own wide stores, except buffer stores with a register in soffset
            if not (ins.op.startswith("buffer_") and len(ins.operands) >= 4 and re.match(r"^s\d+\b", ins.operands[3])):
                continue
        # data operand: global_store vaddr, vdata, saddr|off ; buffer_store vdata, vaddr, srsrc, ...
        data = ins.operands[0] if ins.op.startswith("buffer_") else ins.operands[1]
        dregs, states, j = regs_of(data), 0, i + 1
        while states < 2 and j < len(body):
            nxt = body[j]
            if nxt.op != "s_nop" and (nxt.written_vregs() & dregs):
                how = "end the asm string in `s_nop 1`" if ins.in_asm else "keep the scalar offset out of soffset: hipcc pads the other forms itself"
                found.append(f"{short}: line {nxt.line}: `{nxt.text}` writes data registers of the {'asm' if ins.in_asm else 'compiler-made'} store at "
                             f"line {ins.line} (`{ins.text}`) after {states} wait state(s); a store of more than 8 bytes needs 2 ({how})")
                break
            states += nxt.wait_states()
            j += 1

    # ---- (ii) asm loads: destination untouched until the counted wait that retires it
    vmq: list[tuple[Inst, set[int]]] = []   # outstanding vmcnt operations, oldest first: (instruction, pending dest regs or empty)
    lgq: list[tuple[Inst, set[int]]] = []
    cur_block_loads: dict[int, set[int]] = {}
    for ins in body:
        op = ins.op
        if op == "s_waitcnt":
            vm, lg = waitcnt_fields(ins)
            if vm is not None:
                vmq = vmq[len(vmq) - vm:] if vm < len(vmq) else vmq
                if vm == 0:
                    vmq = []
            if lg is not None:
                lgq = lgq[len(lgq) - lg:] if lg < len(lgq) else lgq
                if lg == 0:
                    lgq = []
            continue
        pending = set().union(*(r for _, r in vmq), *(r for _, r in lgq)) if (vmq or lgq) else set()
        if pending:
            touched = ins.vregs() & pending
            # the issuing asm block may name its own destinations again (a second load of a pair, its own wait)
            own = cur_block_loads.get(ins.block, set()) if ins.in_asm else set()
            if touched - own:
                src = next(i0 for i0, r in (vmq + lgq) if r & touched)
                found.append(f"{short}: line {ins.line}: `{ins.text}` touches v{sorted(touched - own)} while the asm load at line {src.line} "
                             f"(`{src.text}`) is still in flight (no counted wait has retired it)")
                for q in (vmq, lgq):  # report a register once
                    for k, (i0, r) in enumerate(q):
                        q[k] = (i0, r - touched)
        is_dma = VM_LOAD.match(op) and (" lds" in ins.text or "_lds_" in op)
        if VM_LOAD.match(op) or VM_STORE.match(op) or VM_ATOMIC.match(op):
            dest = set()
            if ins.in_asm and VM_LOAD.match(op) and not is_dma:
                dest = regs_of(ins.operands[0])
                cur_block_loads.setdefault(ins.block, set()).update(dest)
            vmq.append((ins, dest))
            if op.startswith("flat_"):
                lgq.append((ins, set()))
        elif DS_OP.match(op) or SMEM.match(op):
            dest = set()
            if ins.in_asm and op.startswith("ds_read"):
                dest = regs_of(ins.operands[0])
                cur_block_loads.setdefault(ins.block, set()).update(dest)
            lgq.append((ins, dest))

    # ---- (iii) asm LDS-DMA: M0 written in the same block, one wait state before the load
    for i, ins in enumerate(body):
        if not (ins.in_asm and VM_LOAD.match(ins.op) and (" lds" in ins.text or "_lds_" in ins.op)):
            continue
        j, states, ok = i - 1, 0, False
        while j >= 0 and body[j].in_asm and body[j].block == ins.block:
            if re.match(r"^s_(mov|add|or|and|lshl\w*)_\w+\s+m0\b", body[j].text):
                ok = states >= 1
                break
            states += body[j].wait_states()
            j -= 1
        if not ok:
            found.append(f"{short}: line {ins.line}: asm LDS-DMA `{ins.text}` without `s_mov_b32 m0, ...` + one wait state (`s_nop 0`) "
                         f"in front of it inside the same asm block")
    return found


def check_text(text: str) -> tuple[list[str], dict]:
    found, stats = [], {"kernels": 0, "asm_wide_stores": 0, "asm_loads": 0, "asm_lds_dma": 0}
    for name, body in kernels(text):
        stats["kernels"] += 1
        for ins in body:
            if not ins.in_asm:
                continue
            if VM_STORE.match(ins.op) and WIDE.search(ins.op):
                stats["asm_wide_stores"] += 1
            elif VM_LOAD.match(ins.op) and (" lds" in ins.text or "_lds_" in ins.op):
                stats["asm_lds_dma"] += 1
            elif VM_LOAD.match(ins.op) or ins.op.startswith("ds_read"):
                stats["asm_loads"] += 1
        found += check_kernel(name, body)
    return found, stats


def compile_to_asm(src: str, outdir: str, extra: list[str] | None = None) -> str:
    base = os.path.splitext(os.path.basename(src))[0]
    out = os.path.join(outdir, base + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-Wno-unused-value", "-Wno-inline-asm", *PER_FILE_FLAGS.get(base, []), *(extra or []), "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def sources() -> list[str]:
    # every kernel file: (i)-(iii) concern inline asm, (iv) compiler-made stores anywhere
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*", help="HIP sources (default: every csrc/*.hip)")
    ap.add_argument("--asm", action="append", default=[], help="lint this assembly listing instead of compiling")
    ap.add_argument("--keep", help="directory to keep the generated .s files in")
    a = ap.parse_args()
    bad = 0
    listings = list(a.asm)
    with tempfile.TemporaryDirectory() as td:
        outdir = a.keep or td
        os.makedirs(outdir, exist_ok=True)
        if not listings:
            srcs = a.files or sources()
            with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
                listings = list(ex.map(lambda s: compile_to_asm(s, outdir), srcs))
        for path in listings:
            found, stats = check_text(open(path).read())
            print(f"{os.path.basename(path)}: {stats['kernels']} kernels, asm sites: {stats['asm_wide_stores']} wide stores, "
                  f"{stats['asm_loads']} register loads, {stats['asm_lds_dma']} LDS-DMA -> {len(found)} finding(s)")
            for f in found:
                print("   ", f)
            bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
