"""Instruction counts between `; MARK name` lines of a hipcc -S listing (developer tool; build with -DMIDAS_ISA_MARKS).

usage: python tools/isa/segments.py listing.s kernel-name-substring [first-mark]
Walks the listing in layout order from the first occurrence of `first-mark` (default: settle) to the next one of the same
name -- one body of the unrolled loop -- and prints, per segment, VALU by class (plain / slow = perm, sdwa, sad, dpp /
cmp -> sgpr), SALU, LDS, VMEM and a wall-time estimate from tools/probes/isa_rate.hip's figures.  Straight-line layout: code
in side branches (slow path, partial lanes) is counted where it lies, so read the segments with the source next to them.
"""
import re
import sys

NS = {"plain": 1.17, "slow": 1.88, "cmps": 2.5}


def klass(op, line):
    if not op.startswith("v_"):
        return None
    if "sdwa" in op or "sdwa" in line or "dpp" in line or op.startswith("v_perm") or op.startswith("v_sad") or op.startswith("v_mul_lo") or op.startswith("v_mad_i64") or op.startswith("v_mad_u64"):
        if op.startswith("v_cmp"):
            return "cmps"
        return "slow"
    if op.startswith("v_cmp") and "_e64" in op and re.search(r"\ss\[", line):
        return "cmps"
    return "plain"


def main():
    path, want = sys.argv[1], sys.argv[2]
    first = sys.argv[3] if len(sys.argv) > 3 else "settle"
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if want in ln and re.match(r"^\S+:", ln) and not ln.startswith(".L"):
            start = i
            break
    segs, cur, seen = [], None, 0
    for ln in lines[start:]:
        s = ln.strip()
        if s.startswith("; MARK"):
            name = s[7:].strip()
            if name == first:
                seen += 1
                if seen == 2:
                    break
            if seen:
                cur = {"name": name, "plain": 0, "slow": 0, "cmps": 0, "s": 0, "ds": 0, "vm": 0, "wait": 0}
                segs.append(cur)
            continue
        if cur is None or not s or s.startswith(";") or s.startswith(".") or re.match(r"^\S+:$", s):
            continue
        op = s.split()[0]
        k = klass(op, s)
        if k:
            cur[k] += 1
        elif op.startswith("s_waitcnt"):
            cur["wait"] += 1
        elif op.startswith("s_"):
            cur["s"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vm"] += 1
        if "s_endpgm" in s:
            break
    tot = 0.0
    for g in segs:
        ns = sum(g[k] * NS[k] for k in NS)
        tot += ns
        print("%-10s plain %4d  slow %4d  cmp->s %3d | SALU %4d  LDS %3d  VMEM %2d  waits %2d | ~%6.1f ns" % (
            g["name"], g["plain"], g["slow"], g["cmps"], g["s"], g["ds"], g["vm"], g["wait"], ns))
    print("sum of VALU time ~%.1f ns (static, straight-line)" % tot)


main()
