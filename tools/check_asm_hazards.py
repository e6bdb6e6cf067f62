#!/usr/bin/env python3
"""Scan the gfx950 ISA of the hand-scheduled kernels for uses of in-flight ds_read destinations.

The MFMA kernels issue their LDS fragment reads from inline asm and wait for them with counted
`s_waitcnt lgkmcnt(N)`.  hipcc does not know those reads are asynchronous: wherever it decides to move a fragment
register (loop back-edges and exits, operand set-up of a tied asm operand behind a branch) it emits a plain
`v_mov` of a register whose read may not have landed yet -- stale data, timing dependent, found on MI355X as
rare wrong 16-row blocks.  The kernels are therefore written so that every loop boundary and branch follows an
`s_waitcnt lgkmcnt(0)`; this script proves it on the compiled code: it compiles each source to assembly, replays
the LDS read queue (reads return in order; `lgkmcnt(N)` retires all but the N newest) and the queue of global loads into
registers (`vmcnt`), and reports every instruction that touches a register with a pending read.

usage: python tools/check_asm_hazards.py [--hipcc PATH] [--quiet] [file.hip ...]      exit code 1 if any hazard is found
(vltk_amd/csrc/Makefile runs it after the link and fails the build on a hit; validated with hipcc 7.2.26015 / clang 22.0.0git roc-7.2.0)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "vltk_amd", "csrc")
DEFAULT = ["conv_mfma256.hip", "conv_mfma_duo.hip", "conv3x3_panel.hip", "conv_gemm4.hip", "bneck_fused.hip"]


def _regs(tok):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", tok):
        if m.group(1):
            out.append(int(m.group(1)))
        else:
            out += list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan_function(lines):
    """lines: the instructions of one kernel.  Returns [(line_no, text, pending_read_line)].

    A small forward dataflow over the text: the queues of pending reads are carried along every branch -- the state at an
    `s_branch` / `s_cbranch_*` is saved for its target label and merged in when the label is reached; after an unconditional
    branch the fall-through text is unreachable and starts from what its predecessors saved (hipcc lays out a loop's exit block
    before the loop body, and a later phase of a kernel before an earlier one, so a purely linear scan would be wrong both
    ways).  Branches to labels ABOVE (loop back-edges, out-of-order block placement) are handled by repeating the scan with the
    states they saved until nothing changes."""
    def merge(a, b):
        return sorted(set((i, frozenset(r)) for i, r in a) | set((i, frozenset(r)) for i, r in b))

    back = {}                                # label -> (pend, vpend) carried by branches from below (previous passes)
    found_all = {}
    for _pass in range(8):
        pend, vpend, found = [], [], []      # in-flight LDS reads / the in-order queue of vector-memory instructions
        saved = {}                           # label -> (pend, vpend) carried by forward branches
        new_back = {}
        passed = set()
        reachable = True
        in_asm = False                       # only HAND-ISSUED loads are watched: hipcc waits correctly for its own
        for i, raw in enumerate(lines):
            if "#ASMSTART" in raw:
                in_asm = True
            elif "#ASMEND" in raw:
                in_asm = False
            t = raw.split(";")[0].strip()
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                lab = m.group(1)
                sp, sv = saved.pop(lab, ([], []))
                bp, bv = back.get(lab, ([], []))
                sp, sv = merge(sp, bp), merge(sv, bv)
                if reachable:
                    pend, vpend = merge(pend, sp), merge(vpend, sv)
                else:
                    pend, vpend = merge([], sp), merge([], sv)
                reachable = True
                passed.add(lab)
                continue
            if not t or t.endswith(":") or t.startswith("."):
                continue
            parts = t.split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            ops = [a.strip() for a in args.split(",")]
            if op == "s_branch" or op.startswith("s_cbranch"):
                lab = ops[0]
                if lab in passed:             # a label above: its state is merged in on the next pass
                    sp, sv = new_back.get(lab, ([], []))
                    new_back[lab] = (merge(sp, pend), merge(sv, vpend))
                else:
                    sp, sv = saved.get(lab, ([], []))
                    saved[lab] = (merge(sp, pend), merge(sv, vpend))
                if op == "s_branch":
                    reachable = False
                    pend, vpend = [], []
                continue
            if op.startswith("ds_read"):
                if in_asm:
                    pend.append((i, set(_regs(ops[0]))))
                continue
            if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", op):
                # every vector-memory instruction takes a place in the in-order vmcnt queue (LDS-DMA pieces, stores and the
                # compiler's own loads too: they carry no registers to watch, but `vmcnt(N)` counts them); only HAND-ISSUED
                # loads into registers are watched
                is_load_to_regs = "_load" in op and "_lds_" not in op and " lds" not in t
                vpend.append((i, set(_regs(ops[0])) if (in_asm and is_load_to_regs) else set()))
                if in_asm and "_store_dwordx" in op and re.search(r"dwordx[34]", op):
                    # a 96- / 128-bit asm store reads its data registers over the states after issue and hipcc pads nothing
                    # inside or behind an asm statement: the string must end with `s_nop 1` (bneck_fused.hip: a real bug)
                    nxt = next((ln.split(";")[0].strip() for ln in lines[i + 1:i + 4] if ln.split(";")[0].strip() and "#ASM" not in ln), "")
                    mm = re.match(r"s_nop\s+(\d+)", nxt)
                    if not (mm and int(mm.group(1)) >= 1):
                        found.append((i, t + "   [asm store without s_nop 1 behind it]", i))
                continue
            if op.startswith("s_waitcnt"):
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    if n == 0:
                        pend = []
                    elif n < len(pend):
                        pend = pend[len(pend) - n:]
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:          # the queue holds every vector-memory instruction in issue order, so the newest n stay pending
                    n = int(m.group(1))     # (merged paths: the union of their queues, which can only over-state what is pending)
                    if n == 0:
                        vpend = []
                    elif n < len(vpend):
                        vpend = vpend[len(vpend) - n:]
                continue
            if op.startswith("s_"):
                continue
            used = set()
            for a in ops:
                used |= set(_regs(a))
            for li, rs in list(pend) + list(vpend):
                if used & set(rs):
                    found.append((i, t, li))
        for f in found:
            found_all[(f[0], f[2], f[1])] = f
        if new_back == back:
            break
        back = new_back
    return [found_all[k] for k in sorted(found_all)]


def scan_asm_mfma_region(lines, min_states=20):
    """Kernels whose MFMAs are written as asm (conv_gemm4.hip): hipcc does not know those statements are matrix instructions, so
    it inserts none of the wait states an MFMA needs around it and treats an operand register as free the instruction after.
    Flagged: every compiler-generated VALU / accumulator instruction (anything `v_*` outside an asm block, plain v_mov_b32
    between long-lived registers excepted) issued fewer than `min_states` wait states after an asm MFMA (an instruction counts
    1, `s_nop k` counts k + 1).  Both forms were real bugs on MI355X: a v_accvgpr_read one instruction after the MFMA that
    produces the value returned stale results, and an address temporary allocated in a just-consumed operand register
    corrupted that operand.  Returns [(line index, text)]; [] if the function has no asm MFMA."""
    bad = []
    in_asm = False
    since = None                      # wait states since the last asm MFMA
    for i, ln in enumerate(lines):
        t = ln.strip()
        if not t or t.startswith(";") and not t.startswith(";;#ASM"):
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.endswith(":") or t.startswith("."):
            continue
        if in_asm and t.startswith("v_mfma"):
            since = 0
            continue
        if since is None:
            continue
        # (v_readlane / v_writelane: hipcc's SGPR spill traffic through a register it reserves for that)
        if not in_asm and t.startswith("v_") and not t.startswith(("v_mov_b32", "v_readlane_b32", "v_writelane_b32")) and since < min_states:
            bad.append((i, t))
        m = re.match(r"s_nop\s+(\d+)", t)
        since += int(m.group(1)) + 1 if m else 1
    return bad


def scan_file(path, hipcc="hipcc", scan=None):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                        "-S", "--cuda-device-only", path, "-o", out], check=True, stderr=subprocess.DEVNULL)
        src = open(out).read().split("\n")
    res = {}
    starts = [i for i, ln in enumerate(src) if re.match(r"^_Z\w+:", ln)]
    for st in starts:
        name = src[st].split(":")[0]
        end = next(i for i in range(st, len(src)) if "s_endpgm" in src[i])
        res[name] = (scan or scan_function)(src[st:end])
    return res


def main():
    argv = sys.argv[1:]
    hipcc, quiet = "hipcc", False
    if "--hipcc" in argv:
        i = argv.index("--hipcc")
        hipcc = argv[i + 1]
        del argv[i:i + 2]
    if "--quiet" in argv:
        quiet = True
        argv.remove("--quiet")
    files = argv or [os.path.join(CSRC, f) for f in DEFAULT]
    bad = 0
    for f in files:
        for name, found in scan_file(f, hipcc=hipcc).items():
            if found or not quiet:
                print(f"{os.path.basename(f)}  {name}: {len(found)} hazard(s)")
            ablation = "conv_gemm4_kernel" in name and "ILb0ELi0ELi" not in name   # stamp / timing-only DBG builds (dummy reads): not product code
            for i, t, li in found[:0 if ablation else 6]:
                print(f"    line +{i}: {t}    <- ds_read at +{li} still pending")
            bad += 0 if ablation else len(found)
        for name, found in scan_file(f, hipcc=hipcc, scan=scan_asm_mfma_region).items():
            if found:
                print(f"{os.path.basename(f)}  {name}: {len(found)} compiler VALU / accumulator instruction(s) between asm MFMAs")
                for i, t in found[:6]:
                    print(f"    +{i}: {t}")
                if "ILb0ELi0ELi" in name:                   # stamp and ablation builds are not product code
                    bad += len(found)
    if quiet:
        print(f"check_asm_hazards: {len(files)} file(s), {bad} hazard(s) in product builds")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
