#!/usr/bin/env python3
"""Static audit of the compiled fused bilinear kernel (csrc/mi_bilinear_flash.h).

The kernel issues its MFMAs from inline asm, so hipcc's hazard recognizer does not see them: nothing stops the compiler
from placing one of ITS instructions (a register copy, a spill, a scheduled VALU op) on the destination registers of a
matrix instruction that is still in the matrix pipe -- which reads stale data or corrupts the accumulator without any
fault.  (It happened once: see the note at fl_mfma_s.)  This script compiles mi_bilinear.hip to assembly and checks, for
every bilinear_flash_kernel instantiation:

  * no instruction other than an MFMA accumulating into the same registers touches the destination of an MFMA issued
    fewer than WAIT_STATES wait states earlier (one per instruction, N + 1 per `s_nop N`);
  * no scratch (spill) instruction inside a loop;
  * no `s_waitcnt vmcnt(0)` inside a loop (it would drain the LDS-DMA prefetch);
  * no compiler-generated instruction uses M0 (the loop's LDS-DMA pieces set M0 from asm without saving it);
  * every score chain is ONE accumulator: the MFMA that starts a chain (C operand 0) and the D / 16 - 1 MFMAs behind it
    write and accumulate the same registers.  (The accumulating statements name the accumulator as an INPUT only -- as
    an output, hipcc pads a wait state between any two consecutive asm statements of a chain -- so nothing but this
    check would notice a register copy slipped in between two of them.)

  * the softmax micro-ops that overwrite a register they declare as an INPUT only (fl_v_exp, fl_v_add, fl_dpp_max: declared
    that way so that hipcc pads no wait state between two of them) form intact chains:
      - `v_exp_f32 vN, vN` exponentiates a register that the last asm `v_fmamk_f32` (the prescale) wrote, the `v_add_f32`
        and `v_cvt_pk_bf16_f32` behind it read exactly that register, and NO compiler-generated instruction (a copy, a
        spill, an accumulator move) reads or writes it between the prescale and its last asm reader;
      - all running-sum additions between two `; fl_opaque` markers (fl_v_opaque: where hipcc is told the sum changed) use
        one and the same accumulator register, untouched by compiler-generated instructions in between;
      - the six `v_max_f32_dpp` steps of a wave maximum and the `v_readlane_b32` behind them name one register, untouched by
        compiler-generated instructions from the first step to the read.

Exit status 0 = clean.  Used by tests/test_flash_isa_audit.py (CPU suite) and by hand after kernel edits.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CSRC = os.path.join(ROOT, "mutual-information-multimodal_amd", "csrc")
WAIT_STATES = 18  # 16-pass rule; v_mfma_f32_32x32x16_bf16 is an 8-pass instruction (11 needed)

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def compile_asm():
    if os.environ.get("MI_AUDIT_ASM"):  # reuse an assembly file (development)
        return open(os.environ["MI_AUDIT_ASM"]).read()
    out = os.path.join(tempfile.mkdtemp(prefix="mi_audit_"), "mi_bilinear.s")
    # extra compiler flags select a build variant: `audit_flash_isa.py -DMI_FL_WAIT_GROUP=4`, `-DMI_STAMPS -DMI_FL_DIAG=0`
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--offload-arch=gfx950", "-S",
           "--cuda-device-only", *[a for a in sys.argv[1:] if a.startswith("-D")], "-o", out,
           os.path.join(CSRC, "mi_bilinear.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    with open(out) as f:
        return f.read()


def kernels(asm):
    for m in re.finditer(r"^(_ZN2mi21bilinear_flash_kernel\w+):[^\n]*\n", asm, re.M):
        end = asm.index(".Lfunc_end", m.end())
        yield m.group(1), asm[m.end():end]


def audit(name, body):
    problems = []
    inflight = []  # [set of regs, wait states since issue]
    in_loop = False
    seen_mfma = False  # the prologue's id-copy loop legitimately waits for its plain loads
    lines = body.split("\n")
    in_asm = False
    m = re.search(r"ILi(\d+)E", name)
    chain_len = int(m.group(1)) // 16 if m else 0
    grad = "ELb1E" in name  # bilinear_flash_kernel<D, GRAD>
    chain = None  # [destination registers, MFMAs seen]
    hidden = {}       # VGPR -> "prescaled" | "exp": values asm statements rewrite behind hipcc's back
    asm_block = []    # the instructions of the CURRENT asm statement
    lsum_touch = None  # a compiler-generated instruction touched the running sum: fine only if an fl_opaque follows
    vreg = lambda t: ("v", int(t[1:])) if re.fullmatch(r"v\d+", t.split()[0] if t else "") else None  # noqa: E731
    lsum = None       # accumulator register of the running-sum additions since the last `; fl_opaque`
    dpp = None        # [register, steps seen] of an open wave-maximum chain
    for ln, raw in enumerate(lines, 1):
        if "#ASMSTART" in raw:
            in_asm = True
            asm_block = []
        elif "#ASMEND" in raw:
            in_asm = False
            if len(asm_block) == 1:  # the micro-ops are one-instruction statements (an `s_nop` in front does not count)
                bln, op, ops, operands, line = asm_block[0]
                d = vreg(ops[0]) if ops else None
                if op == "v_fmamk_f32" and d:
                    hidden[d] = "prescaled"
                elif op == "v_exp_f32" and d and len(ops) == 2 and vreg(ops[1]) == d:
                    if hidden.get(d) == "prescaled":
                        hidden[d] = "exp"
                    else:
                        problems.append(f"{name}:{bln}: `{line}` exponentiates a register no asm prescale wrote")
                elif op == "v_add_f32" and d and len(ops) == 3 and vreg(ops[1]) == d:
                    p = vreg(ops[2])
                    if p is None or hidden.get(p) != "exp":
                        problems.append(f"{name}:{bln}: `{line}` adds a register that is not a fresh asm exponential")
                    if not grad:  # forward-only kernel: no output product, the addition is the exponential's last reader
                        hidden.pop(p, None)
                    if lsum is None:
                        lsum = d
                    elif d != lsum:
                        problems.append(f"{name}:{bln}: `{line}`: the running sum moved from v{lsum[1]} to v{d[1]} between two fl_opaque points")
                    if lsum_touch is not None:
                        problems.append(f"{name}:{lsum_touch[0]}: compiler-generated `{lsum_touch[1]}` touches the running-sum "
                                        f"register between two fl_opaque points")
                        lsum_touch = None
                elif op == "v_cvt_pk_bf16_f32" and len(ops) == 3:
                    for src in (vreg(ops[1]), vreg(ops[2])):
                        if src is None or hidden.get(src) != "exp":
                            problems.append(f"{name}:{bln}: `{line}` packs a register that is not a fresh asm exponential")
                        hidden.pop(src, None)
                elif op == "v_max_f32_dpp" and d:
                    if "quad_perm:[1,0,3,2]" in operands:
                        if dpp is not None:
                            problems.append(f"{name}:{bln}: a wave-maximum chain starts inside another")
                        dpp = [d, 1]
                    elif dpp is None or dpp[0] != d:
                        problems.append(f"{name}:{bln}: `{line}` is not on the register of the open wave-maximum chain")
                    else:
                        dpp[1] += 1
        elif in_asm and "fl_opaque" in raw:
            lsum = None
            lsum_touch = None
        elif not in_asm and re.search(r"\bm0\b", raw.split(";")[0]):
            problems.append(f"{name}:{ln}: compiler-generated use of M0: {raw.strip()}")
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";") else ""
        if "Loop Header" in raw or "in Loop:" in raw:
            in_loop = True
        elif raw.startswith(".LBB") and "Loop" not in raw:
            in_loop = False
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op = line.split()[0]
        operands = line[len(op):]
        states = 1
        if op == "s_nop":
            states = int(operands.strip()) + 1
        if op.startswith("v_mfma"):
            parts = [p.strip() for p in operands.split(",")]
            dst, srcc = regs_of(parts[0]), regs_of(parts[3]) if len(parts) > 3 else set()
            others = regs_of(parts[1]) | regs_of(parts[2])
            for regs, age in inflight:
                if age < WAIT_STATES and (regs & others):
                    problems.append(f"{name}:{ln}: MFMA reads an in-flight MFMA result as A/B: {line}")
                if age < WAIT_STATES and (regs & (dst | srcc)) and regs != dst:
                    problems.append(f"{name}:{ln}: MFMA overlaps an in-flight MFMA destination partially: {line}")
            inflight.append([dst, 0])
            seen_mfma = True
            if len(parts) > 3 and parts[3] == "0":
                if chain is not None:
                    problems.append(f"{name}:{ln}: a score chain starts {chain[1]} MFMAs into the previous one")
                chain = [dst, 1]
            elif chain is not None:
                if dst != chain[0] or srcc != chain[0]:
                    problems.append(f"{name}:{ln}: MFMA {chain[1] + 1} of a score chain does not accumulate the chain's registers: {line}")
                chain[1] += 1
            if chain is not None and chain[1] == chain_len:
                chain = None
        else:
            touched = regs_of(operands)
            ops = [o.strip() for o in operands.split(",")]
            if not in_asm:
                for r in touched:
                    if hidden.get(r) == "exp":  # (a prescaled value is still what hipcc believes it is)
                        problems.append(f"{name}:{ln}: compiler-generated `{line}` touches {r[0]}{r[1]}, which asm micro-ops "
                                        f"rewrite behind hipcc's back (state: {hidden[r]})")
                    if lsum is not None and r == lsum:
                        lsum_touch = (ln, line)  # fine if an fl_opaque marker follows before the next hidden addition
                    if dpp is not None and r == dpp[0]:
                        problems.append(f"{name}:{ln}: compiler-generated `{line}` touches v{dpp[0][1]} inside a wave-maximum chain")
            else:
                if op != "s_nop":
                    asm_block.append((ln, op, ops, operands, line))
                d = vreg(ops[0]) if ops else None
                if op == "v_readlane_b32" and dpp is not None:
                    src = vreg(ops[1]) if len(ops) > 1 else None
                    if src != dpp[0] or dpp[1] != 6:
                        problems.append(f"{name}:{ln}: `{line}` reads v{src[1] if src else '?'} after {dpp[1]} steps on v{dpp[0][1]}")
                    dpp = None
            for regs, age in inflight:
                if age < WAIT_STATES and (regs & touched):
                    problems.append(f"{name}:{ln}: `{line}` touches the destination of an MFMA issued {age} wait states earlier")
            if in_loop and op.startswith("scratch_"):
                problems.append(f"{name}:{ln}: scratch access inside a loop: {line}")
            if in_loop and seen_mfma and op == "s_waitcnt" and "vmcnt(0)" in operands:
                nxt = next((x.strip() for x in lines[ln:] if x.strip() and not x.strip().startswith(";")), "")
                if not nxt.startswith("s_barrier"):  # the hand-placed "last tile" wait is followed by its barrier
                    problems.append(f"{name}:{ln}: s_waitcnt vmcnt(0) inside a loop (drains the LDS-DMA prefetch)")
        for it in inflight:
            it[1] += states
        inflight = [it for it in inflight if it[1] < WAIT_STATES]
    return problems


def main():
    asm = compile_asm()
    found = False
    bad = []
    for name, body in kernels(asm):
        found = True
        p = audit(name, body)
        n_mfma = len(re.findall(r"^\s*v_mfma", body, re.M))
        print(f"{name}: {n_mfma} MFMAs, {len(p)} problem(s)")
        bad += p
    if not found:
        print("no bilinear_flash_kernel instantiation found in the assembly")
        return 2
    for b in bad[:40]:
        print("  " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
