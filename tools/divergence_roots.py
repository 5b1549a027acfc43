#!/usr/bin/env python3
"""Why does the compiler treat a wave-uniform state machine as divergent?  Takes the kernel's LLVM IR with value names
(hipcc --cuda-device-only -fno-discard-value-names -emit-llvm -c x.hip -o x.bc; llvm-dis x.bc) and opt's uniformity report
(opt -mcpu=gfx950 -passes='print<uniformity>' -disable-output x.bc 2> uni.txt), and traces every divergent branch / switch condition of the kernel back
to the values where the divergence starts (a lane id, a load the analysis cannot prove uniform, a phi that joins a divergent branch).
usage: divergence_roots.py x.ll uni.txt kernel-substring [value-to-trace ...]"""
import re, sys, collections
ll, uni, want = sys.argv[1:4]
trace_vals = sys.argv[4:]
div, brs, on = set(), [], False
for l in open(uni):
    if l.startswith("UniformityInfo for function"): on = want in l
    if on and "DIVERGENT:" in l:
        t = l.split("DIVERGENT:", 1)[1].strip()
        m = re.match(r"(%[\w.]+) = ", t)
        if m: div.add(m.group(1))
        if t.startswith("br i1") or t.startswith("switch"): brs.append(t)
defs, on = {}, False
for l in open(ll):
    if l.startswith("define"): on = want in l
    elif l.startswith("}"): on = False
    if on:
        m = re.match(r"\s+(%[\w.]+) = (.*)", l)
        if m: defs[m.group(1)] = m.group(2)
print(len(div), "divergent values,", len(brs), "divergent terminators in", want)
def ops(d): return re.findall(r"%[\w.]+", d)
def trace(v, depth, seen, path):
    if v in seen or depth > 60: return []
    seen.add(v)
    d = defs.get(v)
    if d is None: return [(v, "<argument>")]
    dops = [o for o in ops(d) if o in div and o != v]
    if not dops: return [(v, ("PHI joins a divergent branch: " if d.startswith("phi") else "") + d[:150])]
    out = []
    for o in dops[:4]: out += trace(o, depth + 1, seen, path + [v])
    return out
cnt = collections.Counter()
for t in brs:
    c = re.match(r"br i1 (%[\w.]+)", t) or re.match(r"switch i32 (%[\w.]+)", t)
    if c:
        for r in trace(c.group(1), 0, set(), []): cnt[r] += 1
for (v, d), n in cnt.most_common(60): print("%5d  %s = %s" % (n, v, d))
for v in trace_vals:
    print("---- trace", v)
    seen = set(); cur = [v]
    for depth in range(12):
        nxt = []
        for x in cur:
            d = defs.get(x, "<argument>")
            print("  " * depth + x, "=", d[:200], "[DIVERGENT]" if x in div else "")
            nxt += [o for o in ops(d) if o in div and o != x and o not in seen]
            seen.update(nxt)
        cur = nxt[:6]
        if not cur: break
