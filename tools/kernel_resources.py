"""Registers / scratch of every step-kernel variant in the built library: carves the gfx950 code objects out of libsmj.so's
fat binary (clang offload bundle: ELF images after the bundle header) and reads their metadata notes.  No GPU needed.
Usage: python tools/kernel_resources.py [path/to/libsmj.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(so):
    blob = open(so, "rb").read()
    rows = []
    with tempfile.TemporaryDirectory() as d:
        pos = 0
        n = 0
        while True:
            pos = blob.find(b"\x7fELF\x02\x01\x01\x40", pos)   # ELF64, little endian, OS ABI 64 = AMDGPU HSA
            if pos < 0:
                break
            # section header table offset + count * size bounds the image
            shoff = int.from_bytes(blob[pos + 0x28:pos + 0x30], "little")
            shentsize = int.from_bytes(blob[pos + 0x3A:pos + 0x3C], "little")
            shnum = int.from_bytes(blob[pos + 0x3C:pos + 0x3E], "little")
            end = pos + shoff + shentsize * shnum
            path = os.path.join(d, f"co{n}.o")
            open(path, "wb").write(blob[pos:end])
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", out)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name or "step_kernel" not in name.group(1):
                    continue
                g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, "?"])[1]
                rows.append((name.group(1), blk.split()[0], g("vgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
            pos = end
            n += 1
    for r in sorted(set(rows)):
        print(f"{r[0][:48]:48s} agpr {r[1]:>4s} vgpr {r[2]:>4s} sgpr_spill {r[3]:>4s} vgpr_spill {r[4]:>4s} scratch_bytes_per_lane {r[5]:>5s}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "stretch_mujoco_amd", "libsmj.so"))
