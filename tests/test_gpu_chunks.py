"""The direct kernel's chunks (pileup_direct.hip "Work items"): a workgroup takes four consecutive tiles and carries what a
tile's reads add behind its last site (at most 160 sites) over to the next tile, so that a read is visited once.  The cases
here keep every read's reference span within the overhang -- the condition under which the host asks for chunks -- and put the
reads where the hand-over can go wrong: on and around every tile border, at contig ends, in contigs whose lengths are exact
multiples of the tile and of the chunk, in many short contigs (chunks that span contigs), with deletions that stretch a read to
exactly the overhang, with soft clips and insertions, and as pieces with halo reads.  Held to the C oracle, bit for bit."""
import random

import numpy as np
import pytest

from midas_amd import abi, pieces
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
TILE = 2048


def _reads_for(rng, length, n, dense_borders=True):
    out = []
    borders = [b for b in range(TILE, length, TILE)]
    for _ in range(n):
        l = rng.choice([150, 150, 100, 36, 151])
        kind = rng.random()
        if kind < 0.55:
            cigar, span = [(0, l)], l
        elif kind < 0.70:
            a = rng.randint(1, l - 2)
            d = rng.randint(1, 160 - l) if l < 159 else 1
            cigar, span = [(0, a), (2, d), (0, l - a)], l + d            # a deletion: up to exactly 160 sites of span
        elif kind < 0.80:
            a, i = rng.randint(1, l - 10), rng.randint(1, 5)
            cigar, span = [(0, a), (1, i), (0, l - a - i)], l - i
        elif kind < 0.92:
            s, t = rng.randint(0, min(20, (l - 6) // 2)), rng.randint(0, min(20, (l - 6) // 2))
            m = l - s - t
            cigar = ([(4, s)] if s else []) + [(0, m)] + ([(4, t)] if t else [])
            span = m
        else:
            a = rng.randint(5, l - 30)
            cigar, span = [(0, a), (3, 7), (7, 10), (8, 2), (0, l - a - 12)], l + 7      # five ops: walked op by op
        if dense_borders and borders and rng.random() < 0.5:
            b = rng.choice(borders)
            pos = b - rng.randint(0, span + 3) + rng.choice([0, 0, 1, -1, 2])
        else:
            pos = rng.randint(-3, length - 1)
        pos = max(-2, min(length - 1, pos))
        out.append(dict(pos=pos, cigar=cigar, seq="".join(rng.choice("ACGTACGTACGTN") for _ in range(l)),
                        qual=[rng.choice([40, 38, 31, 30, 29, 12]) for _ in range(l)], nm=rng.choice([0, 1, 2, 3]),
                        mapq=rng.choice([42, 42, 30, 19])))
    out.sort(key=lambda r: r["pos"])
    return out


def _table(rng, lengths):
    reads, begin, ref = [], [0], []
    for n in lengths:
        rs = _reads_for(rng, n, max(40, n // 12))
        reads += rs
        begin.append(len(reads))
        ref.append("".join(rng.choice("ACGTacgtN") for _ in range(n)))
    soa = H.reads_from_dicts(reads)
    table = abi.ContigTable(length=lengths, species=[k % 3 for k in range(len(lengths))], read_begin=begin,
                            ref=np.frombuffer("".join(ref).encode(), np.uint8), n_species=3,
                            ids=["c%d" % k for k in range(len(lengths))], species_ids=["s0", "s1", "s2"])
    return table, soa


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_chunked_tiles_hand_their_overhang_over_exactly(hip_ctx, seed):
    rng = random.Random(seed)
    lengths = [TILE * 4, TILE * 8 + 1, TILE * 3 - 1, TILE, 5, TILE + 160, TILE * 2, 700, TILE * 12, TILE * 5 + 161, 1, TILE * 4 + 2047]
    rng.shuffle(lengths)
    table, soa = _table(rng, lengths + [TILE * 40])
    for args in (dict(abi.DEFAULT_ARGS), dict(abi.DEFAULT_ARGS, baseq=0, mapid=50.0, aln_cov=0.2, readq=0, mapq=0)):
        thr = abi.Thresholds.from_args(args)
        st, er, oc, oa, os_ = c_oracle.pileup(thr, table, soa)
        assert st == 0, (st, er)
        b = hip_ctx.batch(table, soa)
        info = b.info()
        assert info.path == abi.PATH_DIRECT and info.direct_reach <= 160 and info.n_tiles >= 80
        assert info.direct_chunk_tiles == 4 and info.direct_overhang == 160                           # (the chunks are on)
        for _ in range(2):
            b.run(thr)
            counts, allele, stats = b.fetch()
            bad = np.nonzero((counts != oc).any(axis=1))[0]
            assert bad.size == 0, "counts differ at %d sites, first %s" % (bad.size, bad[:8])
            assert np.array_equal(allele, oa) and np.array_equal(stats, os_)
        b.select_path(abi.PATH_PACKED)
        b.run(thr)
        c2, _, s2 = b.fetch()
        assert np.array_equal(c2, oc) and np.array_equal(s2, os_)
        b.close()


def test_chunks_and_pieces_with_halo_reads(hip_ctx, thr_default):
    rng = random.Random(9)
    table, soa = _table(rng, [TILE * 70, TILE * 9 + 5])
    st, _, oc, oa, os_ = c_oracle.pileup(thr_default, table, soa)
    assert st == 0
    pt, pr, _ = pieces.split_table(table, soa, 65536)
    assert pt.n_contigs > table.n_contigs
    counts, allele, stats = hip_ctx.pileup(thr_default, pt, pr)
    assert np.array_equal(counts, oc) and np.array_equal(allele, oa) and np.array_equal(stats, os_)


def test_a_few_long_spans_do_not_switch_the_chunks_off(hip_ctx):
    """Real alignments hold the odd read with a 20 bp deletion or an N skip: its span exceeds the overhang.  Such outliers are
    listed once per batch -- the ranges pass keeps to the common span (the tiles' streams stay as short as without them), the
    tiles they reach find them through the list, and only the chunks they touch are piled up tile by tile.  Outliers on tile
    and chunk borders, reaching one tile or many, first and last in their contig; held to the C oracle on both paths."""
    rng = random.Random(21)
    lengths = [TILE * 16, TILE * 4 + 5, TILE * 40, 900, TILE * 9]
    reads, begin, ref = [], [0], []
    n_out = 0
    for n in lengths:
        rs = _reads_for(rng, n, max(40, n // 12))
        borders = [b for b in range(TILE, n, TILE)]
        for k in range(max(2, n // 6000)):
            l = 150
            a = rng.randint(10, l - 10)
            gap = rng.choice([11, 20, 200, 500, 1900, TILE, 3 * TILE + 7])                      # D or N: spans of 161 ... 6 300 sites
            op = rng.choice([2, 3])
            pos = (rng.choice(borders) - rng.choice([0, 1, a, a + gap // 2, l + gap - 1, l + gap, 161])) if borders and rng.random() < 0.7 \
                else rng.randint(0, max(0, n - 1))
            pos = max(0, min(n - 1, pos))
            rs.append(dict(pos=pos, cigar=[(0, a), (op, gap), (0, l - a)], seq="".join(rng.choice("ACGT") for _ in range(l)),
                           qual=[40] * l, nm=gap if op == 2 else 0, mapq=42))
            n_out += 1
        rs.sort(key=lambda r: r["pos"])
        reads += rs
        begin.append(len(reads))
        ref.append("".join(rng.choice("ACGTacgtN") for _ in range(n)))
    soa = H.reads_from_dicts(reads)
    table = abi.ContigTable(length=lengths, species=[k % 2 for k in range(len(lengths))], read_begin=begin,
                            ref=np.frombuffer("".join(ref).encode(), np.uint8), n_species=2,
                            ids=["c%d" % k for k in range(len(lengths))], species_ids=["s0", "s1"])
    args = dict(abi.DEFAULT_ARGS, mapid=0.0)          # (a 500 bp deletion is NM 500: keep those reads)
    thr = abi.Thresholds.from_args(args)
    st, er, oc, oa, os_ = c_oracle.pileup(thr, table, soa)
    assert st == 0, (st, er)
    b = hip_ctx.batch(table, soa)
    info = b.info()
    assert info.path == abi.PATH_DIRECT and info.direct_reach > 3 * TILE
    assert info.direct_chunk_tiles == 4 and info.direct_overhang == 160          # (chunks on, the outliers listed)
    # the streams are those of the common span -- the reads plus the straddlers (half of these reads sit on tile borders) -- not
    # every read within 6 000 sites of a tile, which would be four tiles' worth per tile
    assert info.direct_stream_reads < 2.5 * info.n_reads, (info.direct_stream_reads, info.n_reads)
    for _ in range(2):
        b.run(thr)
        counts, allele, stats = b.fetch()
        bad = np.nonzero((counts != oc).any(axis=1))[0]
        assert bad.size == 0, "counts differ at %d sites, first %s" % (bad.size, bad[:8])
        assert np.array_equal(allele, oa) and np.array_equal(stats, os_)
    b.select_path(abi.PATH_PACKED)
    b.run(thr)
    c2, _, s2 = b.fetch()
    assert np.array_equal(c2, oc) and np.array_equal(s2, os_)
    b.close()


def _long_reads_for(rng, length, n, lens, max_span):
    """as _reads_for, for reads of `lens` bases whose reference spans stay within max_span: dense on tile borders, deletions that
    stretch a read to exactly max_span, insertions, clips, general CIGARs"""
    out = []
    borders = [b for b in range(TILE, length, TILE)]
    for _ in range(n):
        l = rng.choice(lens)
        kind = rng.random()
        if kind < 0.55:
            cigar, span = [(0, l)], l
        elif kind < 0.70:
            a = rng.randint(1, l - 2)
            d = rng.randint(1, max_span - l) if l < max_span - 1 else 1
            cigar, span = [(0, a), (2, d), (0, l - a)], l + d
        elif kind < 0.80:
            a, i = rng.randint(1, l - 10), rng.randint(1, 5)
            cigar, span = [(0, a), (1, i), (0, l - a - i)], l - i
        elif kind < 0.92:
            s_, t = rng.randint(0, min(20, (l - 6) // 2)), rng.randint(0, min(20, (l - 6) // 2))
            m = l - s_ - t
            cigar = ([(4, s_)] if s_ else []) + [(0, m)] + ([(4, t)] if t else [])
            span = m
        else:
            a = rng.randint(5, l - 30)
            cigar, span = [(0, a), (3, 7), (7, 10), (8, 2), (0, l - a - 12)], l + 7
        if borders and rng.random() < 0.5:
            b = rng.choice(borders)
            pos = b - rng.randint(0, span + 3) + rng.choice([0, 0, 1, -1, 2])
        else:
            pos = rng.randint(-3, length - 1)
        pos = max(-2, min(length - 1, pos))
        out.append(dict(pos=pos, cigar=cigar, seq="".join(rng.choice("ACGTACGTACGTN") for _ in range(l)),
                        qual=[rng.choice([40, 38, 31, 30, 29, 12]) for _ in range(l)], nm=rng.choice([0, 1, 2, 3]),
                        mapq=rng.choice([42, 42, 30, 19])))
    out.sort(key=lambda r: r["pos"])
    return out


@pytest.mark.parametrize("seed", [4, 5])
def test_250_bp_reads_are_chunked_with_the_long_overhang(hip_ctx, seed):
    """Reads longer than the common overhang of 160 sites (2 x 250 MiSeq): the batch takes the kernel's instantiation whose tallies
    hold 288 sites behind a tile (three workgroups a CU), chunks stay on -- batch_get_info says so -- and the table is the
    oracle's: reads on every tile and chunk border, deletions that stretch a read to exactly 288 sites, one read with a 500 bp
    deletion among them (an outlier: only the chunks it touches fall back), and 150 bp batches beside it keep the common overhang."""
    rng = random.Random(seed)
    lengths = [TILE * 4, TILE * 8 + 1, TILE * 3 - 1, TILE, 5, TILE + 288, TILE * 2, 700, TILE * 12, TILE * 5 + 289, TILE * 40]
    rng.shuffle(lengths)
    reads, begin, ref = [], [0], []
    for n in lengths:
        rs = _long_reads_for(rng, n, max(40, n // 20), [250, 250, 251, 200, 150, 36], 288) if n > 300 else []
        if n >= TILE * 12:        # outliers: a 500 bp deletion, an N skip over two tiles
            for gap, op in ((500, 2), (2 * TILE + 9, 3)):
                a = rng.randint(10, 240)
                rs.append(dict(pos=rng.choice([TILE * 3 - 100, TILE * 7 + 5]), cigar=[(0, a), (op, gap), (0, 250 - a)],
                               seq="".join(rng.choice("ACGT") for _ in range(250)), qual=[40] * 250, nm=gap if op == 2 else 0, mapq=42))
            rs.sort(key=lambda r: r["pos"])
        reads += rs
        begin.append(len(reads))
        ref.append("".join(rng.choice("ACGTacgtN") for _ in range(n)))
    soa = H.reads_from_dicts(reads)
    table = abi.ContigTable(length=lengths, species=[k % 3 for k in range(len(lengths))], read_begin=begin,
                            ref=np.frombuffer("".join(ref).encode(), np.uint8), n_species=3,
                            ids=["c%d" % k for k in range(len(lengths))], species_ids=["s0", "s1", "s2"])
    # (the third set: an identity threshold so negative that the 16-bit tables of the long-overhang instantiation cannot hold it --
    # that run goes tile by tile through the common instantiation)
    for args in (dict(abi.DEFAULT_ARGS, mapid=0.0), dict(abi.DEFAULT_ARGS, baseq=0, mapid=0.0, aln_cov=0.2, readq=0, mapq=0),
                 dict(abi.DEFAULT_ARGS, mapid=-1.0e6)):
        thr = abi.Thresholds.from_args(args)
        st, er, oc, oa, os_ = c_oracle.pileup(thr, table, soa)
        assert st == 0, (st, er)
        b = hip_ctx.batch(table, soa)
        info = b.info()
        assert info.path == abi.PATH_DIRECT and info.direct_chunk_tiles == 4 and info.direct_overhang == 288, (info.direct_chunk_tiles, info.direct_overhang)
        assert info.direct_reach > 2 * TILE           # (the N skip: listed, not reached over by every tile)
        for _ in range(2):
            b.run(thr)
            counts, allele, stats = b.fetch()
            bad = np.nonzero((counts != oc).any(axis=1))[0]
            assert bad.size == 0, "counts differ at %d sites, first %s" % (bad.size, bad[:8])
            assert np.array_equal(allele, oa) and np.array_equal(stats, os_)
        b.select_path(abi.PATH_PACKED)
        b.run(thr)
        c2, _, s2 = b.fetch()
        assert np.array_equal(c2, oc) and np.array_equal(s2, os_)
        b.close()
    # reads of 300 bases: beyond the long overhang too -- no chunks, still the oracle's table
    rng = random.Random(seed + 100)
    n = TILE * 20
    rs = _long_reads_for(rng, n, 2500, [300, 290, 150], 330)
    soa = H.reads_from_dicts(rs)
    table = abi.ContigTable(length=[n], species=[0], read_begin=[0, len(rs)], ref=np.frombuffer("".join(rng.choice("ACGT") for _ in range(n)).encode(), np.uint8),
                            n_species=1, ids=["c0"], species_ids=["s0"])
    thr = abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, mapid=0.0))
    st, er, oc, oa, os_ = c_oracle.pileup(thr, table, soa)
    assert st == 0
    b = hip_ctx.batch(table, soa)
    info = b.info()
    assert info.direct_chunk_tiles == 1 and info.direct_overhang == 160
    b.run(thr)
    counts, allele, stats = b.fetch()
    assert np.array_equal(counts, oc) and np.array_equal(allele, oa) and np.array_equal(stats, os_)
    b.close()
