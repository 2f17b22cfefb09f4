"""Writes tests/golden/kat_cases.json: HAND-DERIVED known-answer cases for the pileup path.

The expectations below are literals worked out by hand from the reference semantics
(SURVEY.md 8c list; /root/reference/midas/run/snps.py:141-162, 194-213 and the [EXT] pysam
rules) -- this script never calls the oracle or the HIP library, so the file it writes pins
both of them.  Run:  python tests/golden/make_kat_cases.py

Case format: one contig of `contig_len` sites (reference letters `ref`, default all 'A'),
thresholds = CLI defaults overridden by `args`, reads in input order.  `counts` is sparse:
{site: [A, C, G, T]}, every other site is all-zero.  `error` is the MIDAS_SNPS_ERR_READ_*
kind the reference's exception maps to (then counts/stats are not compared).
"""
import json
import os

Q40 = 40


def read(pos, cigar, seq, qual=None, nm=0, mapq=42, flag=0):
    if qual is None:
        qual = [Q40] * len(seq)
    return {"pos": pos, "cigar": cigar, "seq": seq, "qual": qual, "nm": nm, "mapq": mapq, "flag": flag}


def run(seq, start):
    """{site: one-hot} for the bases of `seq` laid down from `start` ('-' = skip a site, lowercase/N = no count)."""
    out = {}
    for i, b in enumerate(seq):
        if b in "ACGT":
            v = [0, 0, 0, 0]
            v["ACGT".index(b)] = 1
            out[str(start + i)] = v
    return out


def add(*maps):
    out = {}
    for m in maps:
        for k, v in m.items():
            cur = out.setdefault(k, [0, 0, 0, 0])
            out[k] = [a + b for a, b in zip(cur, v)]
    return out


cases = []

# (1) one 10M read, all Q40: +1 on its base at 10 sites
cases.append(dict(name="k01_single_10M", contig_len=30, args={},
                  reads=[read(5, "10M", "ACGTACGTAC")],
                  counts=run("ACGTACGTAC", 5), aligned_reads=1, mapped_reads=1, covered_bases=10, total_depth=10))

# (2) 3S7M: align_len 7, query_len 10 -> 0.7 < 0.75 rejects at defaults ...
cases.append(dict(name="k02a_softclip_rejected_by_aln_cov", contig_len=30, args={},
                  reads=[read(10, "3S7M", "TTTACGTACG")],
                  counts={}, aligned_reads=1, mapped_reads=0, covered_bases=0, total_depth=0))
# ... and with aln_cov 0.7 it is kept; the counts use seq[3..9] (qpos starts at the clip length)
cases.append(dict(name="k02b_softclip_kept_qpos_offset", contig_len=30, args={"aln_cov": 0.7},
                  reads=[read(10, "3S7M", "TTTACGTACG")],
                  counts=run("ACGTACG", 10), aligned_reads=1, mapped_reads=1, covered_bases=7, total_depth=7))

# (3) insertion: inserted bases count nowhere.  NM=2, align_len=10 -> pid 80.0; mapid 80 keeps (strict <)
cases.append(dict(name="k03a_insertion", contig_len=20, args={"mapid": 80.0},
                  reads=[read(0, "4M2I4M", "ACGTTTACGT", nm=2)],
                  counts=run("ACGTACGT", 0), aligned_reads=1, mapped_reads=1, covered_bases=8, total_depth=8))
# deletion / ref-skip: the deleted sites get nothing (depth gap).  align_len 8, NM 2 -> pid 75.0
cases.append(dict(name="k03b_deletion", contig_len=20, args={"mapid": 75.0},
                  reads=[read(0, "4M2D4M", "ACGTACGT", nm=2)],
                  counts=add(run("ACGT", 0), run("ACGT", 6)), aligned_reads=1, mapped_reads=1,
                  covered_bases=8, total_depth=8))
cases.append(dict(name="k03c_refskip", contig_len=20, args={"mapid": 75.0},
                  reads=[read(0, "4M2N4M", "ACGTACGT", nm=2)],
                  counts=add(run("ACGT", 0), run("ACGT", 6)), aligned_reads=1, mapped_reads=1,
                  covered_bases=8, total_depth=8))

# (4) H is a no-op; so is P under the SAM specification (the default pad rule, MIDAS_SNPS_PAD_SPEC) ...
cases.append(dict(name="k04_hardclip_pad_noops", contig_len=20, args={},
                  reads=[read(3, "2H4M1P4M2H", "ACGTACGT")],
                  counts=run("ACGTACGT", 3), aligned_reads=1, mapped_reads=1, covered_bases=8, total_depth=8))
# ... while under MIDAS_SNPS_PAD_PYSAM ("pad_rule": "pysam": get_aligned_pairs of the pysam releases of MIDAS's time advances
# the QUERY on BAM_CPAD, in the branch of BAM_CINS / BAM_CSOFT_CLIP) the four bases behind the pad are read one position late:
# sites 7..10 get seq[5], seq[6], seq[7] and -- seq[8] does not exist: the last pair indexes past SEQ inside the contig, which
# is the IndexError of count_coverage for a kept read (status 5)
cases.append(dict(name="k04b_pad_advances_query_pysam_rule_overruns", contig_len=20, args={}, pad_rule="pysam",
                  reads=[read(3, "2H4M1P4M2H", "ACGTACGT")], error=5, error_read=0))
# with one spare base stored behind the aligned ones (9 bases, 4M1P4M covers 8 under the specification) the pysam rule has
# a base for every pair: sites 3..6 = seq[0..3] = ACGT, the pad skips seq[4] = 'T', sites 7..10 = seq[5..8] = CGTA.  The
# filter sees align_len 9 (no clips), NM 0.  Under the specification's rule the same record gives sites 7..10 = seq[4..7] = TCGT
cases.append(dict(name="k04c_pad_pysam_rule_reads_one_late", contig_len=20, args={}, pad_rule="pysam",
                  reads=[read(3, "4M1P4M", "ACGTTCGTA")],
                  counts=add(run("ACGT", 3), run("CGTA", 7)), aligned_reads=1, mapped_reads=1, covered_bases=8, total_depth=8))
cases.append(dict(name="k04d_same_record_spec_rule", contig_len=20, args={},
                  reads=[read(3, "4M1P4M", "ACGTTCGTA")],
                  counts=add(run("ACGT", 3), run("TCGT", 7)), aligned_reads=1, mapped_reads=1, covered_bases=8, total_depth=8))

# (5) N and IUPAC codes count nowhere (depth unchanged at those sites).  NM 2 of 6 -> pid 66.7; mapid 50
cases.append(dict(name="k05_N_and_iupac_not_counted", contig_len=10, args={"mapid": 50.0},
                  reads=[read(0, "6M", "ANCRGT", nm=2)],
                  counts=add(run("A", 0), run("C", 2), run("G", 4), run("T", 5)), aligned_reads=1, mapped_reads=1,
                  covered_bases=4, total_depth=4))

# (6) qual == baseq counts, baseq-1 does not (mean qual 22.5 >= readq 20)
cases.append(dict(name="k06a_baseq_inclusive", contig_len=10, args={},
                  reads=[read(0, "4M", "ACGT", qual=[30, 29, 31, 0])],
                  counts=add(run("A", 0), run("G", 2)), aligned_reads=1, mapped_reads=1, covered_bases=2, total_depth=2))
cases.append(dict(name="k06b_baseq_zero_counts_everything", contig_len=10, args={"baseq": 0},
                  reads=[read(0, "4M", "ACGT", qual=[30, 29, 31, 0])],
                  counts=run("ACGT", 0), aligned_reads=1, mapped_reads=1, covered_bases=4, total_depth=4))

# (7) filter boundaries.  L=150: NM 9 -> 100*141/150.0 == 94.0 is KEPT (strict <); NM 10 rejected
A150 = "A" * 150
cases.append(dict(name="k07a_mapid_boundary", contig_len=200, args={},
                  reads=[read(0, "150M", A150, nm=9), read(0, "150M", A150, nm=10)],
                  counts={str(i): [1, 0, 0, 0] for i in range(150)}, aligned_reads=2, mapped_reads=1,
                  covered_bases=150, total_depth=150))
# 113/150 = 0.7533 kept, 112/150 = 0.7467 rejected at aln_cov 0.75
cases.append(dict(name="k07b_aln_cov_boundary", contig_len=200, args={},
                  reads=[read(0, "37S113M", A150), read(0, "38S112M", A150)],
                  counts={str(i): [1, 0, 0, 0] for i in range(113)}, aligned_reads=2, mapped_reads=1,
                  covered_bases=113, total_depth=113))
# mean qual exactly readq (20) kept -- but every base is below baseq 30, so nothing is counted;
# one base at 19 drops the mean below 20: rejected
cases.append(dict(name="k07c_readq_boundary", contig_len=20, args={},
                  reads=[read(0, "10M", "ACGTACGTAC", qual=[20] * 10),
                         read(0, "10M", "ACGTACGTAC", qual=[20] * 9 + [19])],
                  counts={}, aligned_reads=2, mapped_reads=1, covered_bases=0, total_depth=0))
# mapq == 20 kept, 19 rejected
cases.append(dict(name="k07d_mapq_boundary", contig_len=20, args={},
                  reads=[read(0, "4M", "ACGT", mapq=20), read(0, "4M", "ACGT", mapq=19)],
                  counts=run("ACGT", 0), aligned_reads=2, mapped_reads=1, covered_bases=4, total_depth=4))

# (8) readq averages soft-clipped quals too: aligned bases Q30, clipped bases Q2 -> mean 16 < 20 rejected
cases.append(dict(name="k08_readq_includes_softclip", contig_len=20, args={"aln_cov": 0.5},
                  reads=[read(0, "5S5M", "TTTTTACGTA", qual=[2] * 5 + [30] * 5)],
                  counts={}, aligned_reads=1, mapped_reads=0, covered_bases=0, total_depth=0))

# (9) secondary / duplicate / QC-fail / supplementary reads ARE counted when they pass keep_read
cases.append(dict(name="k09_flags_ignored", contig_len=10, args={},
                  reads=[read(0, "4M", "ACGT", flag=f) for f in (0x100, 0x400, 0x200, 0x800)],
                  counts={"0": [4, 0, 0, 0], "1": [0, 4, 0, 0], "2": [0, 0, 4, 0], "3": [0, 0, 0, 4]},
                  aligned_reads=4, mapped_reads=4, covered_bases=4, total_depth=16))

# (10) aligned_reads counts rejected reads as well (three different rejections + one keep)
cases.append(dict(name="k10_aligned_counts_rejected", contig_len=10, args={},
                  reads=[read(0, "4M", "ACGT", nm=1),            # pid 75 < 94
                         read(0, "4M", "ACGT", qual=[10] * 4),   # mean 10 < 20
                         read(0, "4M", "ACGT", mapq=0),
                         read(0, "4M", "ACGT")],
                  counts=run("ACGT", 0), aligned_reads=4, mapped_reads=1, covered_bases=4, total_depth=4))

# reads hanging over the contig ends are clipped to [0, length)
cases.append(dict(name="k14_clipped_to_contig", contig_len=6, args={},
                  reads=[read(3, "6M", "ACGTAC")],
                  counts=run("ACG", 3), aligned_reads=1, mapped_reads=1, covered_bases=3, total_depth=3))

# two reads overlapping on one site, different bases
cases.append(dict(name="k15_overlap", contig_len=10, args={},
                  reads=[read(0, "4M", "ACGT"), read(2, "4M", "TTAA")],
                  counts={"0": [1, 0, 0, 0], "1": [0, 1, 0, 0], "2": [0, 0, 1, 1], "3": [0, 0, 0, 2],
                          "4": [1, 0, 0, 0], "5": [1, 0, 0, 0]},
                  aligned_reads=2, mapped_reads=2, covered_bases=6, total_depth=8))

# pysam's backward clip walk never looks at cigar[0]: a lone 10S has start 10, end 10 -> align_len 0
cases.append(dict(name="e01_zero_align_len", contig_len=20, args={}, reads=[read(0, "10S", "ACGTACGTAC")],
                  error=3))
cases.append(dict(name="e02_no_nm_tag", contig_len=20, args={}, reads=[read(0, "4M", "ACGT", nm=None)], error=2))
cases.append(dict(name="e03_no_seq", contig_len=20, args={}, reads=[read(0, "4M", "", qual=[])], error=1))
# QUAL absent is only reached when the mapid test passed ...
cases.append(dict(name="e04a_no_qual", contig_len=20, args={}, reads=[read(0, "4M", "ACGT", qual="absent")],
                  error=4))
# ... a read that fails mapid first is just rejected
cases.append(dict(name="e04b_no_qual_but_rejected_first", contig_len=20, args={},
                  reads=[read(0, "4M", "ACGT", qual="absent", nm=4)],
                  counts={}, aligned_reads=1, mapped_reads=0, covered_bases=0, total_depth=0))
# CIGAR claims 12 aligned bases, SEQ holds 10: IndexError on a kept read
cases.append(dict(name="e05_cigar_overrun", contig_len=20, args={}, reads=[read(0, "12M", "ACGTACGTAC")], error=5))
# the error is raised at the FIRST offending read; earlier reads do not matter
cases.append(dict(name="e06_error_read_index", contig_len=20, args={},
                  reads=[read(0, "4M", "ACGT"), read(1, "4M", "ACGT", nm=None), read(2, "4M", "ACGT", nm=None)],
                  error=2, error_read=1))

# ---- the worked example of the SAM v1 specification, section 1.1 (a PUBLISHED vector for the [EXT] semantics) -------------
# The specification prints the alignment column by column:
#
#   Coor     12345678901234  5678901234567890123456789012345
#   ref      AGCATGTTAGATAA**GATAGCTGTGCTAGTAGGCAGTCAGCGCCAT
#   +r001/1        TTAGATAAAGGATA*CTG
#   +r002         aaaAGATAA*GGATA
#   +r003       gcctaAGCTAA
#   +r004                     ATAGCT..............TCAGC
#   -r003                            ttagctTAGGC
#   -r001/2                                        CAGCGGCAT
#
# and the records  r001 pos 7 8M2I4M1D3M TTAGATAAAGGATACTG | r002 pos 9 3S6M1P1I4M AAAAGATAAGGATA | r003 pos 9 5S6M
# GCCTAAGCTAA | r004 pos 16 6M14N5M ATAGCTTCAGC | r003 (supplementary, flag 2064) pos 29 6H5M TAGGC | r001 pos 37 9M
# CAGCGGCAT.  The expected counts below are read off the PRINTED columns (1-based column c = site c - 1), not derived from
# any CIGAR walk of ours: inserted and soft-clipped (lower-case) bases stand over no reference column, the deleted column
# 19 of r001 and the skipped columns 22-35 of r004 carry no base, the pad of r002 is silent.  The example has QUAL '*' and
# NM only on the last record; the path needs both (keep_read), so every read gets Q40 and the NM its printed alignment
# implies (inserted + deleted + mismatched bases: 3, 1, 1, 0, 0, 1).
SAM_REF = "AGCATGTTAGATAAGATAGCTGTGCTAGTAGGCAGTCAGCGCCAT"
SAM_READS = [read(6, "8M2I4M1D3M", "TTAGATAAAGGATACTG", nm=3, mapq=30, flag=99),
             read(8, "3S6M1P1I4M", "AAAAGATAAGGATA", nm=1, mapq=30, flag=0),
             read(8, "5S6M", "GCCTAAGCTAA", nm=1, mapq=30, flag=0),
             read(15, "6M14N5M", "ATAGCTTCAGC", nm=0, mapq=30, flag=0),
             read(28, "6H5M", "TAGGC", nm=0, mapq=17, flag=2064),
             read(36, "9M", "CAGCGGCAT", nm=1, mapq=30, flag=147)]
SAM_COLUMNS = {   # read -> (first 1-based column, printed upper-case bases; '*' / '.' = no base over that column)
    "r001/1": (7, "TTAGATAA" + "GATA*CTG"),      # the two inserted bases AG stand over the ** of the padded reference
    "r002": (9, "AGATAA" + "GATA"),              # aaa clipped; the inserted G stands over the padded reference
    "r003": (9, "AGCTAA"),
    "r004": (16, "ATAGCT" + "." * 14 + "TCAGC"),
    "r003s": (29, "TAGGC"),
    "r001/2": (37, "CAGCGGCAT"),
}
sam_all = add(*[run(bases.replace("*", "-").replace(".", "-"), col - 1) for col, bases in SAM_COLUMNS.values()])
cases.append(dict(name="s01_sam_spec_example_all_reads_kept", contig_len=45, ref=SAM_REF,
                  args={"mapid": 0.0, "aln_cov": 0.0, "mapq": 0, "readq": 0},
                  reads=SAM_READS, counts=sam_all, aligned_reads=6, mapped_reads=6,
                  covered_bases=len(sam_all), total_depth=sum(sum(v) for v in sam_all.values())))
# at the CLI defaults only r004 survives keep_read: identities 14/17 = 82.4 % (r001/1), 10/11 = 90.9 % (r002: aligned
# length 14 - 3 clipped), 5/6 = 83.3 % (r003), 8/9 = 88.9 % (r001/2) are below mapid 94; the supplementary r003
# (5/5 = 100 %) has MAPQ 17 < 20; r004 is 11/11 = 100 %, MAPQ 30, aligned fraction 11/11
sam_r004 = run(SAM_COLUMNS["r004"][1].replace(".", "-"), SAM_COLUMNS["r004"][0] - 1)
# the same example under MIDAS_SNPS_PAD_PYSAM: r002 (3S6M1P1I4M, 14 stored bases) walks 3 + 6 + 1 (pad) + 1 = 11 query
# positions before its last 4M, whose fourth pair would index seq[14]: IndexError if the read is kept -- it is, when
# every read is kept (read index 1) --, and nothing happens at the CLI defaults, where keep_read drops r002 first
cases.append(dict(name="s01b_sam_spec_example_all_reads_kept_pysam_pad_rule", contig_len=45, ref=SAM_REF, pad_rule="pysam",
                  args={"mapid": 0.0, "aln_cov": 0.0, "mapq": 0, "readq": 0}, reads=SAM_READS, error=5, error_read=1))
cases.append(dict(name="s02b_sam_spec_example_default_thresholds_pysam_pad_rule", contig_len=45, ref=SAM_REF, args={},
                  pad_rule="pysam", reads=SAM_READS, counts=sam_r004, aligned_reads=6, mapped_reads=1, covered_bases=11,
                  total_depth=11))
cases.append(dict(name="s02_sam_spec_example_default_thresholds", contig_len=45, ref=SAM_REF, args={},
                  reads=SAM_READS, counts=sam_r004, aligned_reads=6, mapped_reads=1, covered_bases=11, total_depth=11))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_cases.json")
with open(out, "w") as f:
    json.dump({"comment": "hand-derived; written by make_kat_cases.py; do not regenerate from the oracle",
               "cases": cases}, f, indent=1)
print("wrote", out, len(cases), "cases")
