"""Closes the one pin this repository cannot close in its build image: the hand-derived known-answer cases
(tests/golden/kat_cases.json) and a seeded random CIGAR grammar, checked against REAL pysam.

The arithmetic of the pileup lives in pysam (`AlignmentFile.count_coverage`, `AlignedSegment.query_alignment_sequence`,
`get_aligned_pairs`; the reference names `pysam >= 0.8.1`, /root/reference/setup.py:15, call site
/root/reference/midas/run/snps.py:194-199).  pysam is neither in the reference tree nor in the build image, so the cases
were derived by hand and the oracle (oracle/pileup_oracle.{py,c}) follows pysam's published algorithm.  Wherever
`import pysam` works, this script

  1. writes every case as a coordinate-sorted BAM -- by its own few lines of struct + zlib (SAMv1 sections 4.1 / 4.2), so that
     it needs neither this repository's library nor a GPU --, indexes it with pysam,
  2. calls `AlignmentFile.count_coverage(contig, start=0, end=length, quality_threshold=baseq, read_callback=keep_read)`
     exactly as midas/run/snps.py:194-199 does, with `keep_read` = the reference's own function text when a MIDAS checkout
     is given (--reference DIR: the text of midas/run/snps.py:141-162 is read and executed at run time, nothing of it is
     stored here), else this repository's restatement of it (oracle/pileup_oracle.py keep_read) over the real
     AlignedSegment's attributes,
  3. compares the four count arrays, the aligned / mapped counters and the exception a case expects with kat_cases.json,
     and -- for the random grammar -- with the Python oracle under both pad rules,
  4. prints one line per case and exits 1 on any difference that is not a documented one.

Documented difference: the CIGAR op P.  Cases marked `"pad_rule": "pysam"` state what pysam releases whose
get_aligned_pairs treats BAM_CPAD like an insertion give; their twins without the mark state the SAM specification's rule.
The script reports which of the two the installed pysam follows (run_midas.py snps runs the pysam rule by default).

  python tests/golden/check_kats_against_pysam.py [--reference /path/to/MIDAS] [--random 300] [--keep DIR]

Without pysam: prints SKIPPED and exits 0 (tests/test_kat_pysam_script.py runs it that way, and its flow against a stand-in).
"""
import argparse
import json
import os
import random
import struct
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
NT16 = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=XB"
ERR_NAMES = {1: "TypeError", 2: "KeyError", 3: "ZeroDivisionError", 4: "TypeError", 5: "IndexError"}    # MIDAS_SNPS_ERR_READ_*


# ---- a BAM writer of its own (SAMv1 4.1 BGZF, 4.2 BAM) --------------------------------------------------------------
def reg2bin(beg, end):             # SAMv1 5.3
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def parse_cigar(c):
    if isinstance(c, str):
        import re
        return [(CIGAR_OPS.index(op), int(n)) for n, op in re.findall(r"(\d+)([MIDNSHP=XB])", c)]
    return [(int(op), int(n)) for op, n in c]


def bam_record(refid, read):
    cigar = parse_cigar(read["cigar"])
    seq = read.get("seq") or ""
    qual = read.get("qual", None)
    name = b"r\0"
    l = len(seq)
    span = sum(n for op, n in cigar if op in (0, 2, 3, 7, 8)) or 1
    codes = [NT16.index(ch) if ch in NT16 else 15 for ch in seq.upper()]
    if l & 1:
        codes.append(0)
    seq4 = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    if qual == "absent" or qual is None:
        q = b"\xff" * l
    else:
        q = bytes(int(x) for x in qual)
    aux = b""
    nm = read.get("nm", 0)
    if nm is not None and nm >= 0:
        aux = b"NMI" + struct.pack("<I", nm) if nm > 65535 else (b"NMS" + struct.pack("<H", nm) if nm > 255 else b"NMC" + struct.pack("<B", nm))
    pos = int(read["pos"])
    body = struct.pack("<iiBBHHHIiii", refid, pos, len(name), int(read.get("mapq", 42)), reg2bin(max(pos, 0), max(pos, 0) + span),
                       len(cigar), int(read.get("flag", 0)), l, -1, -1, 0)
    body += name + b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigar) + seq4 + q + aux
    return struct.pack("<I", len(body)) + body


def bgzf(data):
    out = b""
    for k in range(0, max(len(data), 1), 0xff00):
        chunk = data[k:k + 0xff00]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return out + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path, contig, length, reads):
    text = ("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (contig, length)).encode()
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1)
    head += struct.pack("<i", len(contig) + 1) + contig.encode() + b"\0" + struct.pack("<i", length)
    with open(path, "wb") as f:
        f.write(bgzf(head + b"".join(bam_record(0, r) for r in reads)))


# ---- the filter -------------------------------------------------------------------------------------------------------
def load_keep_read(reference):
    """-> (callable(aln), state dict with 'args' and 'stats' setters)."""
    import numpy as np
    if reference:
        src = os.path.join(reference, "midas", "run", "snps.py")
        lines = open(src).read().split("\n")
        a = next(i for i, l in enumerate(lines) if l.startswith("def keep_read("))
        b = next(i for i in range(a + 1, len(lines)) if lines[i].startswith("def "))
        ns = {"np": np, "aln_stats": None, "global_args": None}
        exec("\n".join(lines[a:b]) + "\n", ns)

        def setup(args, stats):
            ns["global_args"], ns["aln_stats"] = args, stats
        return ns["keep_read"], setup, "the reference's own keep_read (%s)" % src
    state = {}

    def keep(aln):
        # the restatement's own steps over pysam's attributes (oracle/pileup_oracle.py keep_read, following snps.py:141-162)
        args, st = state["args"], state["stats"]
        st["aligned_reads"] += 1
        align_len = len(aln.query_alignment_sequence)
        query_len = aln.query_length
        if 100 * (align_len - dict(aln.tags)["NM"]) / float(align_len) < args["mapid"]:
            return False
        if np.mean(aln.query_qualities) < args["readq"]:
            return False
        if aln.mapping_quality < args["mapq"]:
            return False
        if align_len / float(query_len) < args["aln_cov"]:
            return False
        st["mapped_reads"] += 1
        return True

    def setup(args, stats):
        state["args"], state["stats"] = args, stats
    return keep, setup, "this repository's restatement of keep_read over pysam's attributes (pass --reference DIR for the reference's own text)"


def run_pysam(pysam, path, contig, length, args, keep, setup):
    """-> ((A, C, G, T) lists, stats) or the exception's type name."""
    stats = {"aligned_reads": 0, "mapped_reads": 0}
    setup(args, stats)
    if not os.path.exists(path + ".bai"):
        pysam.index(path)
    with pysam.AlignmentFile(path, "rb") as bam:
        try:
            counts = bam.count_coverage(contig, start=0, end=length, quality_threshold=args["baseq"], read_callback=keep)
        except Exception as e:      # noqa: BLE001 -- the exception type IS the observable
            return type(e).__name__, stats
    return [list(c) for c in counts], stats


DEFAULT_ARGS = dict(baseq=30, mapq=20, readq=20, mapid=94.0, aln_cov=0.75)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", help="a snayfach/MIDAS checkout: its keep_read is read and executed at run time")
    ap.add_argument("--random", type=int, default=300, help="reads of the seeded random CIGAR grammar (0: skip)")
    ap.add_argument("--keep", help="write the BAMs here instead of a temporary directory")
    opt = ap.parse_args()
    try:
        import pysam
    except ImportError:
        print("SKIPPED: pysam is not importable here (the build image has none); run this where `import pysam` works")
        return 0
    keep, setup, which = load_keep_read(opt.reference)
    print("pysam %s; filter: %s" % (getattr(pysam, "__version__", "?"), which))
    work = opt.keep or tempfile.mkdtemp(prefix="kat_pysam_")
    os.makedirs(work, exist_ok=True)
    cases = json.load(open(os.path.join(HERE, "kat_cases.json")))["cases"]
    bad, pad_other = 0, {"pysam": 0, "spec": 0}
    for case in cases:
        args = dict(DEFAULT_ARGS)
        args.update(case.get("args", {}))
        reads = sorted(case["reads"], key=lambda r: r["pos"])
        path = os.path.join(work, case["name"] + ".bam")
        write_bam(path, "contig_1", case["contig_len"], reads)
        got, stats = run_pysam(pysam, path, "contig_1", case["contig_len"], args, keep, setup)
        has_pad = any("P" in r["cigar"] if isinstance(r["cigar"], str) else any(op == 6 for op, _ in r["cigar"]) for r in case["reads"])
        rule = case.get("pad_rule", "spec")
        if "error" in case:
            want = ERR_NAMES[case["error"]]
            ok = got == want
            detail = "raises %s (expected %s)" % (got if isinstance(got, str) else "nothing", want)
        else:
            exp = [[0] * case["contig_len"] for _ in range(4)]
            for k, v in case.get("counts", {}).items():
                for j in range(4):
                    exp[j][int(k)] = v[j]
            ok = (not isinstance(got, str)) and got == exp and stats["aligned_reads"] == case["aligned_reads"] and stats["mapped_reads"] == case["mapped_reads"]
            if isinstance(got, str):
                detail = "raises %s" % got
            else:
                diff = [i for i in range(case["contig_len"]) if any(got[j][i] != exp[j][i] for j in range(4))]
                detail = "counts equal" if not diff else "counts differ at sites %s" % diff[:8]
                detail += "; aligned/mapped %d/%d (expected %d/%d)" % (stats["aligned_reads"], stats["mapped_reads"], case["aligned_reads"], case["mapped_reads"])
        if has_pad:
            if not ok:
                pad_other[rule] += 1
            print("%-44s %-9s [op P, states the %s rule] %s" % (case["name"], "same" if ok else "other", rule, detail))
            continue                # (of a pair that tells the rules apart one is expected to differ: judged below)
        print("%-44s %-9s %s" % (case["name"], "ok" if ok else "MISMATCH", detail))
        bad += 0 if ok else 1
    if pad_other["pysam"] and pad_other["spec"]:
        print("op P: this pysam differs from cases of BOTH rules -- inspect the lines above")
        bad += 1
    follows_pysam_rule = pad_other["pysam"] == 0
    print("op P: this pysam follows the %s rule (run_midas.py snps --pad_rule %s)" % (
        ("pysam", "pysam") if follows_pysam_rule else ("specification's", "spec")))
    # ---- the random grammar against the Python oracle under the rule this pysam follows -------------------------------
    if opt.random > 0:
        sys.path.insert(0, ROOT)
        from oracle import pileup_oracle as po
        from tests.test_gpu_parity import _random_cigar
        rng = random.Random(20260927)
        L = 6000
        reads = []
        while len(reads) < opt.random:
            l = rng.choice([150, 100, 60, 33, rng.randint(20, 300)])
            cigar, qlen = _random_cigar(rng, l)
            if qlen > l or not any(op in (0, 7, 8) and n > 0 for op, n in cigar) or any(n == 0 for _, n in cigar):
                continue            # (zero-length ops and reads pysam would refuse to index are left to the unit tests)
            reads.append(dict(pos=rng.randint(0, L - 1), cigar=cigar, seq="".join(rng.choice("ACGTACGTN") for _ in range(l)),
                              qual=[rng.choice([40, 35, 31, 30, 29, 12, 2]) for _ in range(l)], nm=rng.choice([0, 1, 2, 5]),
                              mapq=rng.choice([42, 30, 20, 19])))
        reads.sort(key=lambda r: r["pos"])
        path = os.path.join(work, "random_grammar.bam")
        write_bam(path, "contig_1", L, reads)
        for args in (dict(DEFAULT_ARGS), dict(DEFAULT_ARGS, baseq=0, mapid=50.0, aln_cov=0.2, readq=0, mapq=0)):
            got, stats = run_pysam(pysam, path, "contig_1", L, args, keep, setup)
            before = po.PAD_ADVANCES_QUERY
            po.set_pad_rule(follows_pysam_rule)
            try:
                alns = [po.Aln(pos=r["pos"], cigar=parse_cigar(r["cigar"]), seq=r["seq"], qual=r["qual"], nm=r["nm"], mapq=r["mapq"], flag=0)
                        for r in reads]
                ost = {"aligned_reads": 0, "mapped_reads": 0}
                try:
                    exp = po.count_coverage(alns, L, args["baseq"], lambda a: po.keep_read(a, args, ost))
                    exp = [list(map(int, c)) for c in exp]
                except po.PileupError as e:
                    exp = e.kind
            finally:
                po.set_pad_rule(before)
            same = got == exp
            print("%-44s %-9s %d reads, thresholds %s" % ("random CIGAR grammar vs oracle", "ok" if same else "MISMATCH", len(reads), args))
            bad += 0 if same else 1
    print("%d difference(s)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
