#!/usr/bin/env python3
"""Per scratch slot of a kernel's ISA listing (-save-temps -gline-tables-only): where it is stored and where it is reloaded (file:line lists).
usage: spill_slots.py listing.s kernel-substring"""
import re, sys, collections
path, want = sys.argv[1], sys.argv[2]
files = {}; cur = ("?", 0); infn = None
st = collections.defaultdict(list); ld = collections.defaultdict(list)
for line in open(path, errors="replace"):
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"^(\S+):\s*(;.*)?$", line)
    if m and not m.group(1).startswith(".L"): infn = m.group(1)
    if infn is None or want not in infn: continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m: cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
    t = line.strip()
    m = re.match(r"scratch_(load|store)_dword(x\d)?\s+(.*)", t)
    if not m: continue
    off = re.search(r"offset:(\d+)", t); off = int(off.group(1)) if off else 0
    n = int(m.group(2)[1]) if m.group(2) else 1
    (st if m.group(1) == "store" else ld)[(off, n)].append("%s:%d" % (cur[0].replace(".hip.h", "").replace(".inc", ""), cur[1]))
for k in sorted(set(st) | set(ld)):
    print("%5d x%d  st[%d]: %s\n          ld[%d]: %s" % (k[0], k[1], len(st[k]), " ".join(st[k][:12]), len(ld[k]), " ".join(ld[k][:24])))
