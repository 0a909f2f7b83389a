#!/usr/bin/env python3
"""Static instruction mix PER PHASE of a rollout kernel: compiles a copy of the kernel header in which
the region stamps of the trace build (MPPI_STAMP / MPPI_PK_STAMP) are assembler comments, and counts
the instructions between them in the listing.  This is how the `if (has_cg)` of pass 2 was found
(an addition and a select per normal that the source does not show).  No GPU needed.

    tools/phase_mix.py packed 3          # k_rollout_packed<3, 4, true>
    tools/phase_mix.py fused 2 4         # k_rollout_ride<2, 7, true, true>, 2^4 lanes per trajectory
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mppi_gpu_amd", "csrc")
kind, A = sys.argv[1], int(sys.argv[2])
logc = sys.argv[3] if len(sys.argv) > 3 else "4"
tmp = tempfile.mkdtemp()
hdr = f"rollout_{kind}_impl.hpp"
src = open(os.path.join(ROOT, hdr)).read()
if kind == "packed":
    src = src.replace('#define MPPI_PK_STAMP(i) do { } while (0)', '#define MPPI_PK_STAMP(i) asm volatile("; MARK " #i)')
    key = f"_ZN4mppi16k_rollout_packedILi{A}E"
else:
    src = src.replace("if (first) MPPI_STAMP(", "MPPI_MARKX(").replace(
        "#pragma once", '#pragma once\n#define MPPI_MARKX(i) asm volatile("; MARK " #i)', 1)
    key = f"_ZN4mppi14k_rollout_rideILi{A}E"
open(os.path.join(tmp, hdr), "w").write(src)
unit = open(os.path.join(ROOT, f"rollout_{kind}_a{A}.hip")).read().replace(f'"{hdr}"', f'"{os.path.join(tmp, hdr)}"')
open(os.path.join(tmp, "unit.hip"), "w").write(unit)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "-std=c++17", "--offload-arch=gfx950",
       "-Wno-unused-function", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-I", ROOT,
       "-o", os.path.join(tmp, "unit.s"), os.path.join(tmp, "unit.hip")]
if kind == "fused":
    cmd.insert(1, f"-DMPPI_ONLY_LOGC={logc}")
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
L = open(os.path.join(tmp, "unit.s")).read().splitlines()
starts = [i for i, l in enumerate(L) if l.startswith(key) and l.rstrip().endswith(":") is False and ":" in l]
for st in starts:
    name = L[st].split(":")[0]
    if kind == "packed":
        if f"ILi{A}ELi4ELb1ELb0E" not in name and f"ILi{A}ELi" not in name:
            continue
        if not re.search(r"ELb1ELb[01]EE", name):   # SAMPLE = true; RAGGED printed with the name
            continue
    elif "Lb1E" not in name:        # the sampling instantiations
        continue
    en = next(i for i in range(st, len(L)) if L[i].startswith(".Lfunc_end"))
    body = L[st:en]
    marks = [(i, l.strip()) for i, l in enumerate(body) if "; MARK" in l]
    print(name, "lines", len(body))
    by_op = collections.Counter()
    for (a, la), (b, lb) in zip(marks, marks[1:] + [(len(body), "END")]):
        c = collections.Counter()
        for l in body[a:b]:
            m = re.match(r"\s+([a-z_0-9]+)", l)
            if not m:
                continue
            k = m.group(1)
            if k.startswith("v_cndmask"): c["cndmask"] += 1
            if k.startswith("v_mov"): c["mov"] += 1
            if k.startswith("v_"): c["valu"] += 1
            elif k.startswith("ds_"): c["lds"] += 1
            elif k.startswith("s_"): c["salu"] += 1
            elif k.startswith(("buffer_", "global_")): c["vmem"] += 1
            # a VALU instruction that READS an SGPR issues at half rate on gfx950 (4.2 against 2.4
            # SIMD cycles, tools/ubench_int.hip; inline constants and literals do not): count them
            if k.startswith("v_") and not k.startswith(("v_readlane", "v_readfirstlane", "v_mad_u64",
                                                         "v_cmp", "v_writelane")):
                ops = l.split(";")[0].strip().split(None, 1)
                srcs = ops[1].split(",")[1:] if len(ops) > 1 else []
                if any(re.match(r"\s*-?\|?s(\d+|\[)", o) for o in srcs):
                    c["valu_sgpr_src"] += 1
                    by_op[k] += 1
        print(f"  {la:12s} -> {lb:12s} {dict(c)}")
    print("  VALU with an SGPR source, by opcode:", dict(by_op.most_common(12)))
