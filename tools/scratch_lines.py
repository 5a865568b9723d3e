"""Where a kernel's scratch (spill) accesses come from: maps the scratch_load / scratch_store instructions of an ISA listing
compiled with -gline-tables-only (-S --cuda-device-only) to source lines.  Usage: python tools/scratch_lines.py file.s [top]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
files, loc, hits = {}, None, {}
for ln in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2))
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        loc = (files.get(m.group(1), m.group(1)).split("/")[-1], int(m.group(2)))
    if re.search(r"\bscratch_(load|store)", ln):
        k = (loc, "st" if "scratch_store" in ln else "ld")
        hits[k] = hits.get(k, 0) + 1
tot = sum(hits.values())
print("scratch instructions:", tot)
for (l, kind), v in sorted(hits.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{v:5d} {kind} {l}")
