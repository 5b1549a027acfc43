"""Scan gfx950 device code for the hazard found in round 5 (HISTORY.md §9): a VALU instruction WRITES an SGPR / VCC (v_cmp*, v_readlane, v_readfirstlane,
carry-outs ...) and a VALU instruction within the next two wait states READS that register — MI355X then sees the OLD value (measured with
v_writelane reading a ballot).  LLVM's hazard recogniser covers the pairs it knows for its own code; this tool lists every such pair in a code object
so that suspicious ones (inline assembly, SGPR spill code) can be looked at.

    python tools/hazard_scan.py file.o [more.o ...]        (device-only ELF objects: hipcc --cuda-device-only -c)
    python tools/hazard_scan.py --bundle x264vfw_amd/csrc/mb_slice_b_umh.o   (a host object with an embedded offload bundle: unbundled first)
"""
import re
import subprocess
import sys
import tempfile
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
SREG = re.compile(r"(?<![a-z0-9_])(s\[\d+:\d+\]|s\d+|vcc_lo|vcc_hi|vcc)(?![a-z0-9_\[])")


def regs_of(tok):
    """the set of 32-bit scalar registers an operand token names"""
    out = set()
    m = re.fullmatch(r"s(\d+)", tok)
    if m: out.add(int(m.group(1)))
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    if tok == "vcc": out.update({"vcc_lo", "vcc_hi"})
    if tok in ("vcc_lo", "vcc_hi"): out.add(tok)
    return out


def scalar_writes(op, ops):
    """scalar registers a VALU instruction writes"""
    w = set()
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        if op.endswith("_e32") or not ops or not SREG.fullmatch(ops[0]): w |= {"vcc_lo", "vcc_hi"}
        else: w |= regs_of(ops[0])
    elif op.startswith(("v_readlane", "v_readfirstlane")):
        w |= regs_of(ops[0])
    elif re.match(r"v_(add|sub|subrev)_co_|v_addc_co|v_subb_co|v_subbrev_co|v_div_scale|v_mad_u64_u32|v_mad_i64_i32", op):
        if len(ops) > 1 and SREG.fullmatch(ops[1]): w |= regs_of(ops[1])
        elif op.endswith("_e32"): w |= {"vcc_lo", "vcc_hi"}
    return w


CARRY_OUT = re.compile(r"v_(add|sub|subrev)_co_|v_addc_co|v_subb_co|v_subbrev_co|v_div_scale|v_mad_u64_u32|v_mad_i64_i32")


def scalar_reads(op, ops):
    r = set()
    start = 1
    if CARRY_OUT.match(op): start = 2          # (operand 1 is the carry-out: written, not read)
    for t in ops[start:]:
        for m in SREG.finditer(t): r |= regs_of(m.group(1))
    if op.endswith("_e32") and re.match(r"v_cndmask", op): r |= {"vcc_lo", "vcc_hi"}
    return r


def wait_states(op, ops):
    if op == "s_nop": return int(ops[0], 0) + 1 if ops else 1
    return 1


def scan(path):
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
    func = "?"
    recent = []          # (wait states since, written regs, text)
    hits = []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m: func = m.group(1); recent = []; continue
        m = re.match(r"^\s+([sv]_\w+|ds_\w+|global_\w+|scratch_\w+|buffer_\w+|flat_\w+)\s*(.*?)\s*(//.*)?$", line)
        if not m: continue
        op, rest = m.group(1), m.group(2)
        ops = [t.strip() for t in rest.split(",")] if rest else []
        if op.startswith("v_"):
            rd = scalar_reads(op, ops)
            for ws, wr, text in recent:
                if ws < 2 and (rd & wr):
                    hits.append((func, text, line.strip(), ws))
        ws_now = wait_states(op, ops)
        recent = [(ws + ws_now, wr, text) for ws, wr, text in recent if ws + ws_now < 3]
        if op.startswith("v_"):
            wr = scalar_writes(op, ops)
            if wr: recent.append((0, wr, line.strip()))
    return hits


def main():
    args = sys.argv[1:]
    bundle = False
    if args and args[0] == "--bundle": bundle = True; args = args[1:]
    total = 0
    for p in args:
        q = p
        if bundle:
            fd, q = tempfile.mkstemp(suffix=".co")
            os.close(fd)
            subprocess.run([BUNDLER, "--unbundle", "--type=o", "--input=" + p, "--output=" + q, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
        hits = scan(q)
        if bundle: os.unlink(q)
        kinds = {}
        for f, a, b, ws in hits:
            key = (a.split()[0], b.split()[0])
            kinds[key] = kinds.get(key, 0) + 1
        print("%s: %d pairs with fewer than two wait states between a VALU scalar write and a VALU read of it" % (p, len(hits)))
        for k, c in sorted(kinds.items(), key=lambda x: -x[1])[:20]: print("   %5d  %s -> %s" % (c, k[0], k[1]))
        for f, a, b, ws in hits[:int(os.environ.get("HAZ_SHOW", "6"))]: print("      [%s] %s  ==>  %s   (%d wait states between)" % (f[:50], a, b, ws))
        total += len(hits)
    return 0


if __name__ == "__main__":
    sys.exit(main())
