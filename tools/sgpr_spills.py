"""Scalar-register spill report for the step kernel: compiles csrc/smj_kernels.hip to gfx950 assembly (no GPU needed) and
lists, per definition, how often a spilled SGPR is reloaded from its VGPR lane (v_readlane) and in which stage (stages =
intervals between the s_memtime reads of the per-stage cycle counters).  Usage: python tools/sgpr_spills.py [top_n]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stretch_mujoco_amd", "csrc")


def main(top=25):
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(CSRC, "smj_kernels.hip"),
                               "-I", CSRC, "-I", os.path.join(ROOT, "include"), "--save-temps", "-o", "k.o"], cwd=d, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(d, "smj_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "agpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
        m = re.search(r"\.%s:\s+(\d+)" % key, asm)
        print(f"{key}: {m.group(1) if m else '?'}")
    lines = [l.strip() for l in asm.split("\n")]
    wr = re.compile(r"v_writelane_b32 (v\d+), (s\d+), (\d+)")
    rd = re.compile(r"v_readlane_b32 (s\d+), (v\d+), (\d+)")
    spillregs = {r for r, n in collections.Counter(m.group(1) for m in map(wr.match, lines) if m).items() if n >= 16}
    ticks = [i for i, l in enumerate(lines) if "s_memtime" in l]
    stage = lambda i: sum(1 for a in ticks if i >= a)

    def defs(l):
        m = re.match(r"(\S+)\s+(s\[(\d+):(\d+)\]|s(\d+))", l)
        if not m or m.group(1).startswith(("s_cbranch", "s_waitcnt", "s_nop", "s_cmp", "s_bitcmp", "s_setpc", "s_branch")):
            return []
        return ["s%d" % i for i in range(int(m.group(3)), int(m.group(4)) + 1)] if m.group(3) else ["s" + m.group(5)]

    lastdef, slotdef = {}, {}
    agg, where = collections.Counter(), collections.defaultdict(collections.Counter)
    for i, l in enumerate(lines):
        m = wr.match(l)
        if m and m.group(1) in spillregs:
            slotdef[(m.group(1), m.group(3))] = lastdef.get(m.group(2), "?")
            continue
        m = rd.match(l)
        if m and m.group(2) in spillregs:
            d = slotdef.get((m.group(2), m.group(3)), "?")
            agg[d] += 1
            where[d][stage(i)] += 1
            lastdef[m.group(1)] = "reload"
            continue
        for r in defs(l):
            lastdef[r] = l[:60]
    n_instr = sum(1 for l in lines if l and not l.startswith((";", ".")) and not l.endswith(":"))
    print(f"spill VGPRs: {sorted(spillregs)}; static reloads {sum(agg.values())} of {n_instr} instructions")
    for d, v in agg.most_common(top):
        print(f"{v:5d}  {d:62s} stages {dict(where[d].most_common(5))}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 25)
