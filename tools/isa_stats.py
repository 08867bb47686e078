#!/usr/bin/env python3
"""What hipcc made of a kernel, in numbers: the device assembly of one translation unit of csrc/ (hipcc -S --cuda-device-only with the
unit's build flags), and per kernel the instruction count, matrix instructions, branches (waterfall loops = s_cbranch_execnz), EXEC-masked
blocks (s_and_saveexec), v_readfirstlane, scratch accesses, s_waitcnt by kind, and the wait states (s_nop cycles) by the instruction
before / after them.  Round 5 found its four largest kernel gains this way: run-time storage forms compiled to a
branch per stored register quad, scalar offsets hipcc could not prove uniform compiled to a waterfall loop per store, compare + select
masks through SGPR pairs, divergent blocks in the middle of a pipelined K loop.

    python tools/isa_stats.py vfn_bwd16 [kernel-name substring] [--dump]        (needs hipcc; no GPU)"""
import os
import re
import subprocess
import sys
from collections import Counter

unit = sys.argv[1] if len(sys.argv) > 1 else "vfn_mlp16"
needle = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
dump = "--dump" in sys.argv
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vf_nerf_amd", "csrc")
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"]
if unit in ("vfn_mlp16", "vfn_bwd16"):
    flags += ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-pragma-unroll-threshold=10000000"]
if unit in ("vfn_rays", "vfn_grid"):
    flags += ["-ffp-contract=off"]
flags += os.environ.get("VFN_%s_EXTRA" % unit[4:].upper(), "").split()
asm = f"/tmp/{unit}.isa.s"
subprocess.run(["hipcc", *flags, "--cuda-device-only", "-S", os.path.join(root, unit + ".hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and "kernel" in l]
for k, (i, l) in enumerate(starts):
    if needle not in l:
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    body = [x.strip() for x in lines[i:end] if x.startswith("\t") and not x.strip().startswith((";", "."))]
    c = Counter(x.split()[0] for x in body)
    name = re.sub(r"^_Z(N12_GLOBAL__N_1)?\d+", "", l.split(":")[0])[:70]
    mfma = sum(v for kk, v in c.items() if kk.startswith("v_mfma"))
    branches = sum(v for kk, v in c.items() if kk.startswith("s_cbranch"))
    scratch = sum(v for kk, v in c.items() if kk.startswith("scratch_"))
    print(f"{name}\n    {len(body)} instructions, {mfma} matrix, {branches} branches ({c['s_cbranch_execnz']} execnz = waterfall / divergent loops), "
          f"{c['s_and_saveexec_b64']} saveexec, {c['v_readfirstlane_b32']} readfirstlane, {scratch} scratch accesses")
    waits = Counter(x for x in body if x.startswith("s_waitcnt"))
    print("    s_waitcnt:", ", ".join(f"{w.split(None, 1)[1]} x {n}" for w, n in waits.most_common(6)))
    prev, nxt, total = Counter(), Counter(), 0
    for j, x in enumerate(body):
        if x.startswith("s_nop"):
            n = int(x.split()[1]) + 1
            total += n
            prev[body[j - 1].split()[0]] += n
            if j + 1 < len(body):
                nxt[body[j + 1].split()[0]] += n
    print(f"    wait states: {total} cycles of s_nop; after {prev.most_common(4)}; before {nxt.most_common(4)}")
    print("    most frequent:", ", ".join(f"{op} x {n}" for op, n in c.most_common(14)))
    if dump:
        out = f"/tmp/{unit}.{k}.s"
        open(out, "w").write("\n".join(lines[i:end]))
        print("    assembly:", out)
