#!/usr/bin/env python3
"""Instructions per phase of the encoder's front half: compiles tools/count_encode_phases.hip for gfx950 (no GPU needed),
disassembles it and counts the instructions of every probe kernel, by unit.  Loop bodies of the probes' scaffolding are
counted once (they are not part of any phase); the phases themselves are straight-line code.

    python tools/count_encode_phases.py > profiles/r06_encode_phase_instructions.txt
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = "/tmp/count_encode_phases.o"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-fno-strict-aliasing", "-Wno-unused-function",
                       "-c", os.path.join(ROOT, "tools", "count_encode_phases.hip"), "-o", obj])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", obj], stdout=subprocess.DEVNULL, cwd="/tmp")
dev = [f for f in os.listdir("/tmp") if f.startswith("count_encode_phases.o.0.hipv4")][0]
asm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", os.path.join("/tmp", dev)], capture_output=True, text=True, check=True).stdout
for f in os.listdir("/tmp"):
    if f.startswith("count_encode_phases.o.0."):
        os.remove(os.path.join("/tmp", f))
kern = collections.OrderedDict()
cur = None
for line in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
    if m:
        cur = m.group(1)
        kern[cur] = collections.Counter()
        continue
    m = re.match(r"^\s+(\w+)", line)
    if cur and m and not m.group(1).startswith("s_endpgm") and not m.group(1).startswith("s_code_end") and not m.group(1).startswith("s_nop"):
        op = m.group(1)
        unit = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_"))
                else "wait" if op.startswith("s_waitcnt") else "salu")
        kern[cur][unit] += 1
        kern[cur]["all"] += 1
        kern[cur]["op:" + op] += 1


def row(name, k, base=None):
    c = kern[k]
    b = kern[base] if base else collections.Counter()
    return "%-34s %6d %6d %5d %5d   %s" % (name, c["valu"] - b["valu"], c["all"] - b["all"], c["lds"] - b["lds"], c["vmem"] - b["vmem"],
                                          ", ".join("%s %d" % (o[3:], n - b[o]) for o, n in sorted(c.items(), key=lambda t: -t[1]) if o.startswith("op:v_") and n - b[o] >= 8)[:150])


print("Instructions per 8x8 block (one lane's block: the wave executes them once for its 60 blocks), gfx950, hipcc -O3;")
print("each phase compiled as a kernel of its own, its load / store scaffolding (the probe named *_base / probe_copy64) subtracted.\n")
print("%-34s %6s %6s %5s %5s   %s" % ("phase", "VALU", "all", "LDS", "VMEM", "most frequent vector instructions"))
print(row("row pass x 8 (off the packed samples)", "probe_rows", "probe_rows_base"))
print(row("column pass x 8", "probe_cols", "probe_copy64"))
print(row("quantiser x 63 + DC, pairs packed", "probe_quant", "probe_quant_base"))
print(row("non-zero mask off the pairs", "probe_mask", "probe_mask_base"))
print(row("stage 2 as the kernels run it", "probe_transform_block", "probe_unpack_base"))
print()
print("per ten-MCU segment (five trips of 4x2-pixel patches per lane):")
print(row("colour conversion (rgb24 -> planes)", "probe_colour", "probe_colour_base"))
print("  (a STATIC count: it includes the byte-by-byte path of a picture's right edge, which a width that is a multiple of 16 never")
print("   runs, once per trip; the path every patch of the bench stream takes is ~135 instructions per trip -- 8 luma samples of 5,")
print("   2 x 2 chroma samples of 2 x 2-pixel sums, unpacking by SDWA operands, 4 LDS writes -- plus ~40 of placement and the two")
print("   12-byte loads: ~875 per segment, as the round-5 review's 360 + 250 counted them per round of one wave)")
