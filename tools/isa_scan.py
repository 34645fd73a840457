#!/usr/bin/env python3
"""Static hint list: innermost loops of the gfx950 ISA whose body issues vector-memory loads and then waits for all of them
(s_waitcnt vmcnt(0)) — if such a loop runs more than one trip per thread, its memory round trips are taken one after the other
(round 4: the tile staging of orb_fast_nms_kernel and orb_describe_kernel, the neighbour loads of lsd_grow_kernel's two
sub-groups — each was a `for (i = tid; i < n; i += T) lds[i] = global[i]` style loop the compiler did not pipeline, or a
rarely taken branch whose join needed vmcnt(0)).  Whether a hit matters needs the trip count and a profile; double-buffered
loops (a wait for the PREVIOUS trip's loads) are listed too.  `--waits`: per kernel, how many loads are waited for within 8 instructions
of their issue (straight-line code that the loop view cannot see: orb_blur_kernel's unrolled rows).
    python tools/isa_scan.py [--waits] [--max-instr 150] [file.hip ...]   (CPU only: hipcc -S)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stvo-pl_amd", "csrc")
MAX_INSTR = 150  # longer loop bodies are whole phases of a kernel, not copy / staging loops
argv = sys.argv[1:]
WAITS = argv[:1] == ["--waits"]  # second view: per kernel, the loads that are waited for (vmcnt(0)) within 8 instructions of their issue
if WAITS:
    argv = argv[1:]
if argv[:1] == ["--max-instr"]:
    MAX_INSTR, argv = int(argv[1]), argv[2:]
files = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


for src in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
               "-I", CSRC, src, "-o", tmp.name]
        if subprocess.run(cmd, capture_output=True).returncode != 0:
            print(f"{os.path.basename(src)}: does not compile with -S", file=sys.stderr)
            continue
        lines = open(tmp.name, errors="ignore").read().split("\n")
    if WAITS:
        kernel, stats = None, {}
        for i, line in enumerate(lines):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kernel = m.group(1)
                stats[kernel] = [0, 0, 0]  # loads, vmcnt(0) waits, loads waited for at once
            if kernel is None:
                continue
            if re.search(r"\b(global_load|buffer_load|flat_load)", line):
                stats[kernel][0] += 1
                window = [x for x in lines[i + 1:i + 40] if re.match(r"\s+[a-z]", x)][:8]
                if any("vmcnt(0)" in x for x in window):
                    stats[kernel][2] += 1
            if "vmcnt(0)" in line:
                stats[kernel][1] += 1
        for k, (nl, nw, ni) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
            if ni and "rocprim" not in k:
                print(f"{os.path.basename(src):20s} {demangle(k)[:96]:96s} {nl:4d} loads, {nw:4d} x vmcnt(0), {ni:4d} loads waited for within 8 instructions")
        continue
    kernel, labels, seen = None, {}, set()
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, labels = m.group(1), {}
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = i
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
        if not (m and m.group(1) in labels and labels[m.group(1)] < i):
            continue
        a = labels[m.group(1)]
        body = lines[a:i]
        if any(re.match(r"^\.LBB", x) and "Loop Header" in x for x in body[1:]):
            continue  # not an innermost loop
        loads = [k for k, x in enumerate(body) if re.search(r"\b(global_load|buffer_load|flat_load|scratch_load)", x)]
        waits = [k for k, x in enumerate(body) if "vmcnt(0)" in x]
        n_instr = sum(1 for x in body if re.match(r"\s+[a-z]", x))
        if loads and any(w > loads[0] for w in waits) and n_instr <= MAX_INSTR and (kernel, a) not in seen and "rocprim" not in kernel:
            seen.add((kernel, a))
            print(f"{os.path.basename(src):20s} {demangle(kernel)[:100]:100s} loop at asm line {a + 1:6d}: {n_instr:4d} instructions, {len(loads):2d} loads, then vmcnt(0)")
