"""Shared test helpers: build C-ABI inputs from readable case descriptions."""
import json
import os
import re

import numpy as np

from midas_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CIGAR_CHARS = "MIDNSHP=XB"
NT16 = "=ACMGRSVTWYHKDBN"


def parse_cigar(s):
    return [(CIGAR_CHARS.index(op), int(n)) for n, op in re.findall(r"(\d+)([MIDNSHP=XB])", s)]


def encode_seq4(seq):
    codes = [NT16.index(c) if c in NT16 else 15 for c in seq.upper()]
    if len(codes) & 1:
        codes.append(0)
    return [(codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2)]


def reads_from_dicts(reads):
    """[{pos,cigar(str or [(op,len)]),seq,qual(list|'absent'),nm(int|None),mapq,flag}] -> ReadsSoA"""
    pos, mapq, flag, nm, l_seq = [], [], [], [], []
    seq4, qual, cigar = [], [], []
    seq_off, qual_off, cigar_off = [0], [0], [0]
    for r in reads:
        cg = parse_cigar(r["cigar"]) if isinstance(r["cigar"], str) else r["cigar"]
        s = r["seq"]
        q = r.get("qual")
        if q is None:
            q = [40] * len(s)
        elif isinstance(q, str) and q == "absent":
            q = [0xFF] * len(s)
        assert len(q) == len(s)
        pos.append(r["pos"]); mapq.append(r.get("mapq", 42)); flag.append(r.get("flag", 0))
        nm.append(-1 if r.get("nm", 0) is None else r.get("nm", 0)); l_seq.append(len(s))
        seq4 += encode_seq4(s); qual += list(q); cigar += [(ln << 4) | op for op, ln in cg]
        seq_off.append(len(seq4)); qual_off.append(len(qual)); cigar_off.append(len(cigar))
    return abi.ReadsSoA(pos=np.array(pos, dtype=np.int32), mapq=np.array(mapq, dtype=np.uint8),
                        flag=np.array(flag, dtype=np.uint16), nm=np.array(nm, dtype=np.int32),
                        l_seq=np.array(l_seq, dtype=np.int32), seq_off=np.array(seq_off, dtype=np.int64),
                        qual_off=np.array(qual_off, dtype=np.int64), cigar_off=np.array(cigar_off, dtype=np.int64),
                        seq4=np.array(seq4, dtype=np.uint8), qual=np.array(qual, dtype=np.uint8),
                        cigar=np.array(cigar, dtype=np.uint32))


def single_contig(length, n_reads, ref=None):
    ref = np.frombuffer((ref or "A" * length).encode(), dtype=np.uint8)
    return abi.ContigTable(length=[length], species=[0], read_begin=[0, n_reads], ref=ref, n_species=1,
                           ids=["contig_1"], species_ids=["sp"])


def load_kat_cases():
    with open(os.path.join(GOLDEN, "kat_cases.json")) as f:
        return json.load(f)["cases"]


def kat_inputs(case):
    reads = reads_from_dicts(case["reads"])
    contigs = single_contig(case["contig_len"], reads.n_reads, case.get("ref"))
    args = dict(abi.DEFAULT_ARGS)
    args.update(case.get("args", {}))
    return contigs, reads, abi.Thresholds.from_args(args), args


def kat_pysam_pad_rule(case):
    """True when the case states what the path does under MIDAS_SNPS_PAD_PYSAM (the CIGAR op P advances the query)."""
    return case.get("pad_rule", "spec") == "pysam"


def kat_expected_counts(case):
    exp = np.zeros((case["contig_len"], 4), dtype=np.uint32)
    for k, v in case.get("counts", {}).items():
        exp[int(k)] = v
    return exp


def kat_expected_stats(case):
    return np.array([[case["aligned_reads"], case["mapped_reads"], case["covered_bases"], case["total_depth"]]],
                    dtype=np.int64)
