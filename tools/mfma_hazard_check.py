#!/usr/bin/env python3
"""CPU-only ISA check for K1m (csrc/match_mfma.hip): the top-2 folds read the matrix-core accumulators through INLINE ASM (v_min3_f32 /
v_med3_f32 / v_min_f32), which the compiler's hazard recogniser does not see — nothing inserts the wait states an XDL write needs before
a vector instruction may read its destination (gfx950, 8-pass v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands: 12; the reverse
kernel once returned a wrong top-2 for lack of them, round 5).  This script compiles the file to gfx950 assembly with the Makefile's
flags and, for every v_mfma, walks every path of the control-flow graph forward from it, counting wait states CONSERVATIVELY (every
instruction 1 — also an independent v_mfma, which really holds the issue port for 4; s_nop N = N + 1) until a non-matrix instruction
reads one of its destination registers.  It prints the smallest distance per kernel and fails below REQUIRED.
    python tools/mfma_hazard_check.py            (tests/test_isa_mfma_hazard.py runs the same on every CPU test run)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stvo-pl_amd", "csrc")
REQUIRED = 12   # wait states between an 8-pass XDL write and a VALU read of its vDst (advisor, round 5; the reverse kernel's fence gives 19)
HORIZON = 24    # paths are followed this far
STORES = ("global_store", "ds_write", "ds_store", "scratch_store", "buffer_store", "flat_store", "global_atomic", "ds_add", "ds_min", "ds_max")


def assemble(src, extra=("-mllvm", "-amdgpu-mfma-vgpr-form")):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", *extra, "-I", os.path.join(ROOT, "include"),
               "-I", CSRC, src, "-o", tmp.name]
        subprocess.run(cmd, check=True, capture_output=True)
        return open(tmp.name, errors="ignore").read().split("\n")


def vregs(operand):
    """Set of VGPR numbers named by one operand (v12, v[32:47]); empty for anything else."""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(lines):
    """{kernel: [(mnemonic, [operands], label or None)]} with labels resolved to instruction indices."""
    kernels, cur, labels, pending = {}, None, None, []
    for line in lines:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, labels = [], {}
            kernels[m.group(1)] = (cur, labels)
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = len(cur)
            continue
        if re.match(r"^\s*\.(end_amdhsa_kernel|section|size)", line) or line.startswith(".Lfunc_end"):
            if line.startswith(".Lfunc_end"):
                cur = None
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(;.*)?$", line)
        if not m or m.group(1).startswith("."):
            continue
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", m.group(2)) if o.strip()]
        ops = [o.split()[0] for o in ops]   # (modifiers such as op_sel_hi:[..] ride on the last operand)
        cur.append((m.group(1), ops))
    return kernels


ASM_READERS = ("v_min3_f32", "v_med3_f32", "v_min_f32", "v_max_f32")   # what the folds are written in (reads the compiler cannot see)


def min_distance(instrs, labels, only=None):
    """Smallest conservative wait-state distance from a v_mfma to a non-matrix read of its destination, over all paths; None if none
    within HORIZON.  Returns (distance, index of the v_mfma, index of the reader)."""
    worst = None
    n = len(instrs)

    def succ(j):
        mn, ops = instrs[j]
        if mn == "s_endpgm":
            return []
        if mn == "s_branch":
            return [labels[ops[0]]] if ops and ops[0] in labels else []
        out = [j + 1] if j + 1 < n else []
        if mn.startswith("s_cbranch") and ops and ops[-1] in labels:
            out.append(labels[ops[-1]])
        return out

    for i, (mn, ops) in enumerate(instrs):
        if not mn.startswith("v_mfma") or not ops:
            continue
        dst = vregs(ops[0])
        if not dst:
            continue
        best = {}
        stack = [(s, 0) for s in succ(i)]
        while stack:
            j, ws = stack.pop()
            if ws >= HORIZON or best.get(j, HORIZON + 1) <= ws:
                continue
            best[j] = ws
            m2, o2 = instrs[j]
            if m2.startswith("v_mfma"):
                if o2 and vregs(o2[0]) & dst:
                    continue   # the destination is rewritten (accumulation into itself is the matrix pipe's own dependency)
            else:
                srcs = o2 if m2.startswith(STORES) else o2[1:]
                if any(vregs(o) & dst for o in srcs):
                    if only and not m2.startswith(only):
                        continue   # a read the compiler made itself: it counted the wait states
                    if worst is None or ws < worst[0]:
                        worst = (ws, i, j)
                    continue
                if o2 and not m2.startswith(STORES) and vregs(o2[0]) >= dst:
                    continue   # overwritten by something else
            step = 1
            if m2 == "s_nop" and o2:
                step = int(o2[0], 0) + 1
            for s in succ(j):
                stack.append((s, ws + step))
    return worst


def check(src=os.path.join(CSRC, "match_mfma.hip")):
    report = {}
    for name, (instrs, labels) in parse(assemble(src)).items():
        if not any(mn.startswith("v_mfma") for mn, _ in instrs):
            continue
        w = min_distance(instrs, labels, ASM_READERS)
        report[name] = None if w is None else {"wait_states": w[0], "mfma": " ".join([instrs[w[1]][0]] + instrs[w[1]][1][:1]),
                                               "reader": " ".join([instrs[w[2]][0]] + instrs[w[2]][1])}
    return report


if __name__ == "__main__":
    rep = check(*sys.argv[1:2])
    bad = 0
    for k, v in rep.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90]
        if v is None:
            print(f"{name}: no inline-asm read of a matrix destination within {HORIZON} wait states")
        else:
            flag = "" if v["wait_states"] >= REQUIRED else f"   <-- BELOW {REQUIRED}"
            bad += bool(flag)
            print(f"{name}: {v['wait_states']} wait states  [{v['mfma']}] -> [{v['reader']}]{flag}")
    sys.exit(1 if bad else 0)
