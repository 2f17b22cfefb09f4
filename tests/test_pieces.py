"""Pieces of long contigs (midas_amd/pieces.py, midas_snps_contigs.origin): the same tallies and counters as the whole contig.
CPU part: the oracle's semantics of a piece and the writers' row numbering; the GPU part is tests/test_gpu_pieces.py."""
import gzip

import numpy as np
import pytest

from midas_amd import abi, pieces, synth
from oracle import c_oracle


def _dataset(seed=5, contig_len=400000, n_reads=30000, **kw):
    return synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=contig_len, n_reads=n_reads, seed=synth.BASE_SEED + seed, **kw)


def test_cut_covers_the_contig_on_the_member_grid():
    assert pieces.cut(100, 0) == [(0, 100)]
    assert pieces.cut(65536, 65536) == [(0, 65536)]
    c = pieces.cut(3 * 65536 + 5, 65536)
    assert c == [(0, 65536), (65536, 131072), (131072, 196608), (196608, 196613)]
    assert pieces.piece_length(1) == 65536 and pieces.piece_length(65537) == 131072 and pieces.piece_length(0) == 0
    assert all(lo % abi.ROWS_PER_MEMBER == 0 for lo, _ in c)


@pytest.mark.parametrize("piece_len", [65536, 131072])
def test_oracle_pieces_equal_the_whole(piece_len):
    table, reads = _dataset()
    thr = abi.Thresholds(mapid=94.0, mapq=20, baseq=30, readq=20, aln_cov=0.75)
    st, _, counts, allele, stats = c_oracle.pileup(thr, table, reads)
    assert st == 0
    pt, pr, entries = pieces.split_table(table, reads, piece_len)
    assert pt.n_contigs > table.n_contigs and pr.n_reads > reads.n_reads      # (halo reads are in two pieces)
    st2, _, c2, a2, s2 = c_oracle.pileup(thr, pt, pr)
    assert st2 == 0
    assert np.array_equal(counts, c2) and np.array_equal(allele, a2) and np.array_equal(stats, s2)


def test_reference_span_counts_reference_consuming_ops():
    from tests import helpers
    rd = helpers.reads_from_dicts([dict(pos=3, cigar="5S10M2D3I7M4N2=1X", seq="A" * 28, qual=[30] * 28, nm=0, mapq=40),
                                   dict(pos=9, cigar="12M", seq="A" * 12, qual=[30] * 12, nm=0, mapq=40)])
    assert pieces.reference_span(rd) == 10 + 2 + 7 + 4 + 2 + 1


def test_an_overrun_in_the_next_piece_is_still_reported():
    """A CIGAR longer than SEQ whose first uncovered site lies behind the piece's end: the piece that owns the read cannot
    see it, the next piece (where the read is halo) reports it -- as the whole contig would."""
    from tests import helpers
    L = 2 * 65536
    ref = np.frombuffer(b"ACGT" * (L // 4), np.uint8)
    pos = 65536 - 20
    rd = helpers.reads_from_dicts([dict(pos=pos, cigar="60M", seq="ACGTA" * 6, qual=[40] * 30, nm=0, mapq=40)])
    table = abi.ContigTable(length=[L], species=[0], read_begin=[0, 1], ref=ref, n_species=1, ids=["c"], species_ids=["s"])
    thr = abi.Thresholds(mapid=0.0, mapq=0, baseq=0, readq=0, aln_cov=0.0)
    st, er, *_ = c_oracle.pileup(thr, table, rd)
    assert st == abi.ERR_READ_CIGAR_OVERRUN and er == 0
    pt, pr, _ = pieces.split_table(table, rd, 65536)
    assert pr.n_reads == 2
    st2, er2, *_ = c_oracle.pileup(thr, pt, pr)
    assert st2 == abi.ERR_READ_CIGAR_OVERRUN and er2 == 1      # (the halo copy in the second piece)


def test_writer_rows_of_pieces_concatenate_to_the_whole(tmp_path):
    rng = np.random.default_rng(3)
    n = 3 * 16384 + 77
    allele = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n)
    counts = rng.integers(0, 50, (n, 4)).astype(np.uint32)
    whole = str(tmp_path / "whole.gz")
    abi.write_table(whole, ["contig_x"], [allele], [counts], gz_level=4, threads=2)
    cuts = [(0, 16384), (16384, 3 * 16384), (3 * 16384, n)]
    parts = []
    for k, (lo, hi) in enumerate(cuts):
        p = str(tmp_path / ("p%d.gz" % k))
        abi.write_table(p, ["contig_x"], [allele[lo:hi]], [counts[lo:hi]], gz_level=4, threads=2, header=(k == 0), first_pos=[lo])
        parts.append(open(p, "rb").read())
    assert b"".join(parts) == open(whole, "rb").read()
    rows = gzip.open(whole, "rt").read().splitlines()
    assert rows[16385].split("\t")[1] == "16385"
