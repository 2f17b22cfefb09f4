"""CPU tests of the host side of the path: BAM decode, row formatter + gzip writer, summary text, contig
ordering, CLI argument surface.  None of them needs a GPU; none of the product code under test touches oracle/."""
import gzip
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from midas_amd import abi, bam, fasta, synth
from midas_amd.run import snps as msnps
from oracle import pileup_oracle as po
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bam_round_trip_through_native_decoder(tmp_path):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=5000, n_reads=3000,
                                        seed=3, var_len=True)
    reads.nm[5] = -1          # a record without NM
    reads.nm[6] = 300         # NM that needs the 'i' aux type
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "t.bam")
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
    names, lens, rid, got = abi.read_bam(path)
    assert names == contigs.ids and lens == [int(x) for x in contigs.length]
    np.testing.assert_array_equal(rid, refid)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(got, k), getattr(reads, k), err_msg=k)


def test_bam_decoder_rejects_garbage(tmp_path):
    p = tmp_path / "bad.bam"
    p.write_bytes(b"this is not a bam file at all")
    with pytest.raises(abi.MidasSnpsError):
        abi.read_bam(str(p))
    with pytest.raises(abi.MidasSnpsError):
        abi.read_bam(str(tmp_path / "missing.bam"))


def test_group_by_contig_regroups_and_drops_foreign_contigs():
    reads = H.reads_from_dicts([dict(pos=5, cigar="4M", seq="ACGT"), dict(pos=1, cigar="4M", seq="CCCC"),
                                dict(pos=2, cigar="4M", seq="GGGG"), dict(pos=9, cigar="4M", seq="TTTT")])
    refid = np.array([2, 0, 1, 2], dtype=np.int32)
    sub, rb = bam.group_by_contig(["a", "b", "c"], refid, reads, ["c", "a"])   # table order c, a; b is foreign
    assert rb.tolist() == [0, 2, 3]
    assert sub.pos.tolist() == [5, 9, 1]


def test_fasta_parser_matches_biopython_conventions():
    text = ">c1 description here\nacgt\nNNAC\n\n>c2\nGG\n>c3\tx\n"
    recs = list(fasta.parse(io.StringIO(text)))
    assert recs == [("c1", "acgtNNAC"), ("c2", "GG"), ("c3", "")]


def _oracle_text(contigs, reads, args):
    """<species>.snps text per species, from the pysam-shaped oracle."""
    alns = po.alns_from_soa(reads.as_dict())
    off = contigs.site_offsets()
    oc, by = {}, {}
    for k, cid in enumerate(contigs.ids):
        seq = bytes(contigs.ref[off[k]:off[k + 1]]).decode().upper()
        oc[cid] = po.OContig(id=cid, seq=seq, species_id=contigs.species_ids[contigs.species[k]])
        by[cid] = alns[int(contigs.read_begin[k]):int(contigs.read_begin[k + 1])]
    return {sp: po.species_pileup(args, sp, oc, by) for sp in contigs.species_ids}


def test_row_writer_reproduces_reference_text(tmp_path):
    """Counts from the C oracle -> native formatter -> must equal, byte for byte after gunzip, the text the
    pysam-shaped oracle emits with the reference's own loop (midas/run/snps.py:201-210)."""
    from oracle import c_oracle
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=3, contig_len=3000, n_reads=900, seed=2,
                                        lowercase_frac=0.2)
    contigs.ids = ["c_10", "C_1", "a|b"]                    # sorted() order != table order
    args = dict(abi.DEFAULT_ARGS)
    thr = abi.Thresholds.from_args(args)
    st, _, counts, allele, stats = c_oracle.pileup(thr, contigs, reads)
    assert st == 0
    out = str(tmp_path / "sp.snps.gz")
    off = contigs.site_offsets()
    first = True
    for cid in sorted(contigs.ids):
        k = contigs.ids.index(cid)
        abi.write_rows(out, not first, cid, allele[off[k]:off[k + 1]], counts[off[k]:off[k + 1]], threads=3)
        first = False
    got = gzip.open(out, "rt").read()
    exp, exp_stats = _oracle_text(contigs, reads, args)[contigs.species_ids[0]]
    assert got == exp
    # the one-call table writer (what the host uses) produces the same text
    out2 = str(tmp_path / "sp2.snps.gz")
    ks = [contigs.ids.index(cid) for cid in sorted(contigs.ids)]
    abi.write_table(out2, [contigs.ids[k] for k in ks], [allele[off[k]:off[k + 1]] for k in ks],
                    [counts[off[k]:off[k + 1]] for k in ks], threads=5)
    assert gzip.open(out2, "rt").read() == exp
    abi.write_table(out2, [], [], [])            # a species without contigs: header only
    assert gzip.open(out2, "rt").read() == exp.splitlines(keepends=True)[0]
    assert [l.split("\t")[0] for l in got.splitlines()[1::3000]] == ["C_1", "a|b", "c_10"]
    assert exp_stats['total_depth'] == int(stats[0, abi.STAT_TOTAL_DEPTH])


def test_row_writer_large_counts_and_many_members(tmp_path):
    n = 200000   # 13 gzip members of 16 384 rows
    counts = np.zeros((n, 4), dtype=np.uint32)
    counts[:, 0] = np.arange(n)
    counts[7] = [4000000000, 4000000000, 4000000000, 4000000000]   # depth needs 64-bit
    allele = np.frombuffer((b"ACGTN" * (n // 5 + 1))[:n], dtype=np.uint8)
    out = str(tmp_path / "big.snps.gz")
    abi.write_rows(out, False, "contig|1", allele, counts, gz_level=1)
    lines = gzip.open(out, "rt").read().splitlines()
    assert len(lines) == n + 1 and lines[0].startswith("ref_id\tref_pos")
    assert lines[8] == "contig|1\t8\tG\t16000000000\t4000000000\t4000000000\t4000000000\t4000000000"
    assert lines[n] == "contig|1\t%d\t%s\t%d\t%d\t0\t0\t0" % (n, "ACGTN"[(n - 1) % 5], n - 1, n - 1)


def test_summary_text_and_derived_floats(tmp_path):
    """snps_summary + the derived floats of pysam_pileup (midas/run/snps.py:231-241, 247-262): repr() floats,
    untouched int 0 for an uncovered species."""
    a, b = msnps.Species("sp_a"), msnps.Species("sp_b")
    a.genome_length, a.covered_bases, a.total_depth, a.aligned_reads, a.mapped_reads = 15, 14, 33, 9, 7
    a.fraction_covered = a.covered_bases / float(a.genome_length)
    a.mean_coverage = a.total_depth / float(a.covered_bases)
    b.genome_length = 20
    b.fraction_covered = b.covered_bases / float(b.genome_length)
    args = {'outdir': str(tmp_path)}
    os.makedirs(tmp_path / "snps")
    msnps.snps_summary(args, {"sp_a": a, "sp_b": b})
    got = open(tmp_path / "snps" / "summary.txt").read()
    exp = po.snps_summary_text({
        "sp_a": dict(genome_length=15, covered_bases=14, total_depth=33, aligned_reads=9, mapped_reads=7),
        "sp_b": dict(genome_length=20, covered_bases=0, total_depth=0, aligned_reads=0, mapped_reads=0)})
    assert got == exp
    assert "0.9333333333333333" in got and got.splitlines()[2] == "sp_b\t20\t0\t0.0\t0\t0\t0"


def test_initialize_contigs_uppercases_and_keys_by_contig_id(tmp_path):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=400, n_reads=10, seed=1,
                                        lowercase_frac=0.5)
    synth.write_sample(str(tmp_path / "out"), str(tmp_path / "db"), contigs, reads, gz_fasta=True)
    args = {'outdir': str(tmp_path / "out"), 'db': str(tmp_path / "db"), 'build_db': False}
    species = msnps.initialize_species(args)
    assert sorted(species) == sorted(contigs.species_ids)
    cs = msnps.initialize_contigs(species)
    assert sorted(cs) == sorted(contigs.ids)
    off = contigs.site_offsets()
    c0 = cs[contigs.ids[0]]
    assert c0.seq == bytes(contigs.ref[off[0]:off[1]]).decode().upper() and c0.length == 400
    assert c0.species_id == contigs.species_ids[0]


def _cli(*argv, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py")] + list(argv),
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def test_cli_help_and_unknown_commands():
    assert _cli("-h").returncode == 0
    assert _cli("snps", "-h").returncode == 0
    r = _cli("bogus")
    assert r.returncode == 1 and "Unrecognized command" in r.stderr
    r = _cli("species", "x")
    assert r.returncode == 1 and "not part of this build" in r.stderr


def test_cli_argument_checks_follow_the_reference(tmp_path):
    r = _cli("snps", str(tmp_path / "o"), "--pileup", env={"MIDAS_DB": ""})
    assert r.returncode == 1 and "reference database" in r.stderr.lower()
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=400, n_reads=5, seed=1)
    synth.write_sample(str(tmp_path / "out"), str(tmp_path / "db"), contigs, reads)
    db = str(tmp_path / "db")
    r = _cli("snps", str(tmp_path / "empty"), "--pileup", "-d", db)
    assert r.returncode == 1 and "no alignments were found" in r.stderr
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--mapid", "0")
    assert r.returncode == 1 and "between 1 and 100" in r.stderr
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--aln_cov", "1.5")
    assert r.returncode == 1 and "ALN_COV" in r.stderr
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--species_id", "Nope_1")
    assert r.returncode == 1 and "not found" in r.stderr
    # dead flags of the reference are accepted
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--discard", "--baq", "--adjust_mq", "--baseq", "101")
    assert r.returncode == 1 and "BASEQ" in r.stderr
