#!/usr/bin/env python3
"""Where does a kernel spill?  Reads the ISA listing hipcc leaves under -save-temps -gline-tables-only and attributes every scratch load / store
(VGPR spills) and every SGPR-spill v_writelane / v_readlane to the source line its .loc names, summed per file and per bucket of lines.
usage: spill_map.py listing.s [kernel-substring] [--bucket N]"""
import re, sys, collections
def main():
    path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    bucket = int(sys.argv[sys.argv.index("--bucket") + 1]) if "--bucket" in sys.argv else 25
    files = {}; cur = (0, 0); infn = None
    ld = collections.Counter(); st = collections.Counter(); sg = collections.Counter(); ins = collections.Counter()
    for line in open(path, errors="replace"):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
        if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"): infn = m.group(1)
        if want and (infn is None or want not in infn): continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m: cur = (int(m.group(1)), int(m.group(2))); continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        key = (files.get(cur[0], "?"), cur[1] // bucket * bucket)
        ins[key] += 1
        if t.startswith("scratch_load"): ld[key] += 1
        elif t.startswith("scratch_store"): st[key] += 1
        elif ("v_writelane" in t or "v_readlane" in t) and "Spill" in t or "Reload" in t and "v_readlane" in t: sg[key] += 1
    print(f"{'file':24s} {'line':>6s} {'insts':>7s} {'ld':>5s} {'st':>5s} {'sgpr':>6s}")
    for key in sorted(ins, key=lambda k: (k[0], k[1])):
        if ld[key] + st[key] + sg[key] == 0 and "--all" not in sys.argv: continue
        print(f"{key[0]:24s} {key[1]:6d} {ins[key]:7d} {ld[key]:5d} {st[key]:5d} {sg[key]:6d}")
    print("total", sum(ins.values()), sum(ld.values()), sum(st.values()), sum(sg.values()))
main()
