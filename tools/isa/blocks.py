"""Static instruction counts per basic block of one kernel of a hipcc -S listing (developer tool).

usage: python tools/isa/blocks.py listing.s kernel-name-substring [min_instructions]
Prints every basic block in layout order: label, VALU / SALU / LDS / VMEM / scratch counts, the branch targets that leave it
and the `; MARK name` comments found inside (asm volatile("; MARK x") in the source), so that the hot loop can be read off.
"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lines = open(path).read().split("\n")
    start = end = None
    for i, ln in enumerate(lines):
        if start is None and want in ln and not ln.startswith("\t") and not ln.startswith(".L") and re.match(r"^\S+:", ln):
            start = i
        if start is not None and "s_endpgm" in ln and i > start:
            end = i
            break
    blocks, cur = [], None
    for ln in lines[start:end + 1]:
        s = ln.strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m or cur is None:
            cur = {"label": m.group(1) if m else "entry", "v": 0, "s": 0, "ds": 0, "vm": 0, "scr": 0, "br": [], "marks": [], "wait": 0}
            blocks.append(cur)
            if m:
                continue
        if s.startswith("; MARK"):
            cur["marks"].append(s[7:])
            continue
        if s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            cur["v"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur["br"].append(op.replace("s_cbranch_", "") + "->" + s.split()[-1])
            cur["s"] += 1
        elif op.startswith("s_waitcnt"):
            cur["wait"] += 1
        elif op.startswith("s_"):
            cur["s"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith("scratch_"):
            cur["scr"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
            cur["vm"] += 1
    tot = {"v": 0, "s": 0, "ds": 0, "vm": 0, "scr": 0}
    for b in blocks:
        for k in tot:
            tot[k] += b[k]
        if b["v"] + b["s"] + b["ds"] + b["vm"] < floor and not b["marks"]:
            continue
        print("%-12s V %4d  S %4d  DS %3d  VM %3d  SCR %2d  W %2d  %s  %s" % (b["label"], b["v"], b["s"], b["ds"], b["vm"], b["scr"], b["wait"],
              " ".join(b["br"]), ("MARKS: " + ", ".join(b["marks"])) if b["marks"] else ""))
    print("total", tot, "blocks", len(blocks))


main()
