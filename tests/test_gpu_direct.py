"""The device paths of a batch, held to the oracle and to each other.

DIRECT: the pileup kernel reads the BAM's own bytes (index_direct.hip + pileup_direct.hip).
PACKED: tile-ordered records + one byte per base (pack_reads.hip + index_reads.hip + pileup_tiles.hip).
LONG:   one thread per read over the caller's arrays (pileup_long.hip) -- the path of batches that hold a read beyond the fast
        paths' limits, and a structurally different third implementation of the same rules for every other batch.
All must give bit-identical counts, alleles, counters and statuses on every input, in any read order; `auto` picks DIRECT
for position-sorted input without coverage hot spots.
"""
import random

import numpy as np
import pytest

from midas_amd import abi, synth
from oracle import c_oracle
from tests import helpers as H
from tests.test_gpu_parity import _random_cigar

pytestmark = pytest.mark.gpu

PATHS = [abi.PATH_DIRECT, abi.PATH_PACKED, abi.PATH_LONG]
CASES = H.load_kat_cases()


@pytest.fixture(scope="module", params=PATHS, ids=[abi.PATH_NAMES[p] for p in PATHS])
def path_ctx(request):
    ctx = abi.Context(0)
    ctx.set_default_path(request.param)
    ctx.forced_path = request.param
    yield ctx
    ctx.close()


def _same(ctx, thr, contigs, reads):
    st, er, oc, oa, os_ = c_oracle.pileup(thr, contigs, reads)
    assert st == 0, "oracle refused the input (%d at read %d)" % (st, er)
    b = ctx.batch(contigs, reads)
    assert b.info().path == ctx.forced_path
    b.run(thr)
    counts, allele, stats = b.fetch()
    b.close()
    bad = np.nonzero((counts != oc).any(axis=1))[0]
    assert bad.size == 0, "counts differ at %d sites, first %s: hip %s oracle %s" % (
        bad.size, bad[:5], counts[bad[:5]].tolist(), oc[bad[:5]].tolist())
    np.testing.assert_array_equal(allele, oa)
    np.testing.assert_array_equal(stats, os_)
    return counts, stats


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hand_derived_cases_on_both_paths(path_ctx, case):
    contigs, reads, thr, _ = H.kat_inputs(case)
    path_ctx.set_pad_rule(abi.PAD_PYSAM if H.kat_pysam_pad_rule(case) else abi.PAD_SPEC)
    try:
        if "error" in case:
            with pytest.raises(abi.MidasSnpsError) as ei:
                path_ctx.pileup(thr, contigs, reads)
            assert ei.value.status == case["error"]
            assert ei.value.read_index == case.get("error_read", 0)
            return
        counts, allele, stats = path_ctx.pileup(thr, contigs, reads)
    finally:
        path_ctx.set_pad_rule(abi.PAD_SPEC)
    np.testing.assert_array_equal(counts, H.kat_expected_counts(case))
    np.testing.assert_array_equal(stats, H.kat_expected_stats(case))


@pytest.mark.parametrize("read_len,var_len", [(150, False), (150, True), (151, False), (100, False), (30, False), (31, True),
                                               (61, False), (250, False), (1000, False), (16, False)])
def test_read_lengths_and_lane_sizes(path_ctx, thr_default, read_len, var_len):
    """30 and 32 bases per lane (direct), ragged tails, reads straddling tiles, indels and clips included."""
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=9000, n_reads=4000 if read_len < 500 else 600,
                                        read_len=read_len, seed=1000 + read_len, var_len=var_len, lowercase_frac=0.2)
    _same(path_ctx, thr_default, contigs, reads)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cigar_grammar(path_ctx, seed):
    rng = random.Random(seed)
    L = 30000
    reads = []
    for _ in range(5000):
        l = rng.choice([150, 150, 150, 100, 60, 33, rng.randint(20, 400)])
        while True:
            cigar, qlen = _random_cigar(rng, l)
            if qlen <= l and any(op in (0, 7, 8) and n > 0 for op, n in cigar):
                break
        pos = rng.choice([rng.randint(0, L - 1), rng.randint(4000, 4200), rng.randint(L - 300, L - 1), rng.randint(8100, 8250)])
        seq = "".join(rng.choice("ACGTACGTACGTNRYKM=") for _ in range(l))
        qual = [rng.choice([40, 38, 35, 31, 30, 29, 12, 2, 0, 93, 200, 254]) for _ in range(l)]
        reads.append(dict(pos=pos, cigar=cigar, seq=seq, qual=qual, nm=rng.choice([0, 1, 2, 5, 9, 30, 1023, 1024, 3000]),
                          mapq=rng.choice([42, 42, 30, 20, 19, 3])))
    reads.sort(key=lambda r: r['pos'])
    soa = H.reads_from_dicts(reads)
    ref = "".join(rng.choice("ACGTacgtN") for _ in range(L))
    contig = H.single_contig(L, len(reads), ref)
    sets = [dict(abi.DEFAULT_ARGS), dict(abi.DEFAULT_ARGS, baseq=0, mapid=80.0, aln_cov=0.3, readq=10, mapq=0)]
    if True:      # (a baseq above 62 on qualities above 62: the packed path hands such a run to the direct kernel)
        sets += [dict(abi.DEFAULT_ARGS, baseq=93, mapid=1.0, readq=0), dict(abi.DEFAULT_ARGS, baseq=201, mapid=1.0, readq=0),
                 dict(abi.DEFAULT_ARGS, baseq=-3, readq=-1), dict(abi.DEFAULT_ARGS, baseq=255, readq=0, mapid=0.0),
                 dict(abi.DEFAULT_ARGS, baseq=256, readq=0, mapid=0.0), dict(abi.DEFAULT_ARGS, baseq=1, readq=300)]
    for args in sets:
        _same(path_ctx, abi.Thresholds.from_args(args), contig, soa)


def test_long_skips_pads_hard_clips_and_positions_off_the_contig(path_ctx):
    rng = np.random.default_rng(3)
    L = 20000

    def rs(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    reads = [dict(pos=-1, cigar="20M", seq=rs(20)),                                   # starts before the contig
             dict(pos=4000, cigar="50M9000N50M", seq=rs(100), nm=0),
             dict(pos=4090, cigar="10M2D10M", seq=rs(20), nm=2),
             dict(pos=4095, cigar="1M", seq="G"), dict(pos=4096, cigar="1M", seq="T"),
             dict(pos=4090, cigar="3S10M2S", seq=rs(15), nm=0),                       # soft clips across the 4096 border
             dict(pos=8100, cigar="5H20S100M3I27M10S2H", seq=rs(160), nm=3),
             dict(pos=8190, cigar="4M1P4M", seq=rs(8)),
             dict(pos=8191, cigar="2H3S4=2X1S1H", seq=rs(10), nm=2),                  # class 0 with hard clips, = and X
             dict(pos=12000, cigar="30=5X30=", seq=rs(65), nm=5),
             dict(pos=19990, cigar="30M", seq=rs(30)),                                # hangs over the contig end
             dict(pos=19999, cigar="5S1M", seq=rs(6)),
             dict(pos=25000, cigar="10M", seq=rs(10))]                                # starts behind the contig
    for _ in range(300):
        l = int(rng.integers(30, 200))
        a = int(rng.integers(5, l - 10))
        d = int(rng.integers(1, 3000))
        reads.append(dict(pos=int(rng.integers(0, L - 200)), cigar="%dM%dD%dM" % (a, d, l - a), seq=rs(l), nm=0))
    reads.sort(key=lambda r: r["pos"])
    soa = H.reads_from_dicts(reads)
    contigs = H.single_contig(L, soa.n_reads, ref=rs(L))
    _same(path_ctx, abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, mapid=1.0, aln_cov=0.0)), contigs, soa)


def test_any_read_order_is_exact(path_ctx, thr_default):
    from oracle import pileup_oracle as po
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=30000, n_reads=5000, seed=21)
    perm = np.random.default_rng(0).permutation(reads.n_reads)
    objs = po.alns_from_soa(reads.as_dict())
    shuffled = [dict(pos=o.pos, cigar=o.cigar, seq=o.seq, qual=list(o.qual), nm=o.nm, mapq=o.mapq, flag=o.flag)
                for o in (objs[j] for j in perm)]
    c_sorted, _ = _same(path_ctx, thr_default, contigs, reads)
    c_shuf, _ = _same(path_ctx, thr_default, contigs, H.reads_from_dicts(shuffled))
    np.testing.assert_array_equal(c_sorted, c_shuf)


def test_deep_coverage_and_reruns(path_ctx, thr_default):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=120000, seed=9)
    st, _, oc, oa, os_ = c_oracle.pileup(thr_default, contigs, reads)
    loose = abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, baseq=0, readq=0, mapq=0, mapid=0.0, aln_cov=0.0))
    st2, _, oc2, _, os2 = c_oracle.pileup(loose, contigs, reads)
    assert st == 0 and st2 == 0
    b = path_ctx.batch(contigs, reads)
    for k in range(4):      # the index pass, the general lists and the tile bounds reset themselves run after run
        t, c, s = (thr_default, oc, os_) if k % 2 == 0 else (loose, oc2, os2)
        b.run(t)
        counts, allele, stats = b.fetch()
        assert np.array_equal(counts, c) and np.array_equal(stats, s) and np.array_equal(allele, oa)
    assert counts.sum(axis=1).max() > 255
    b.close()


def test_errors_lowest_read_wins_on_both_paths(path_ctx, thr_default):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=12000, n_reads=3000, seed=77)
    d = reads.as_dict()
    for idx, what in ((2100, 'nm'), (900, 'qual'), (1500, 'nm')):
        e = {k: v.copy() for k, v in d.items()}
        if what == 'nm':
            e['nm'][idx] = -1
            e['nm'][min(idx + 40, len(e['nm']) - 1)] = -1
            want = abi.ERR_READ_NO_NM
        else:
            e['qual'][e['qual_off'][idx]:e['qual_off'][idx + 1]] = 0xFF
            e['nm'][idx] = 0
            e['mapq'][idx] = 42
            want = abi.ERR_READ_NO_QUAL
        bad = abi.ReadsSoA(**e)
        st, er, *_ = c_oracle.pileup(thr_default, contigs, bad)
        with pytest.raises(abi.MidasSnpsError) as ei:
            path_ctx.pileup(thr_default, contigs, bad)
        assert (ei.value.status, ei.value.read_index) == (st, er) and st == want


def test_empty_batches(path_ctx, thr_default):
    contigs, _ = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=5000, n_reads=10, seed=3)
    contigs.read_begin = np.zeros(5, dtype=np.int64)
    counts, allele, stats = path_ctx.pileup(thr_default, contigs, abi.ReadsSoA.empty())
    assert counts.sum() == 0 and stats.sum() == 0 and allele.size == contigs.n_sites


def test_auto_picks_direct_for_sorted_input_and_packed_otherwise(hip_ctx, thr_default):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=20000, seed=5)
    b = hip_ctx.batch(contigs, reads)
    info = b.info()
    assert info.path == abi.PATH_DIRECT and info.path_auto == abi.PATH_DIRECT
    assert info.packed_bytes == 0                                         # nothing was packed
    assert reads.n_reads <= info.direct_stream_reads <= reads.n_reads * 1.1   # every read once, straddlers twice
    assert info.direct_general_reads == 0          # clips and single indels are settled in registers: nothing is walked op by op
    assert 150 <= info.direct_reach <= 153         # the longest reference span: 150 bp + a deletion of up to three
    b.close()
    # the same reads, contig by contig in random order: the streams no longer add up -> packed
    d = reads.as_dict()
    from oracle import pileup_oracle as po
    objs = po.alns_from_soa(d)
    rng = np.random.default_rng(1)
    out = []
    for c in range(contigs.n_contigs):
        lo, hi = int(contigs.read_begin[c]), int(contigs.read_begin[c + 1])
        for j in rng.permutation(np.arange(lo, hi)):
            o = objs[j]
            out.append(dict(pos=o.pos, cigar=o.cigar, seq=o.seq, qual=list(o.qual), nm=o.nm, mapq=o.mapq, flag=o.flag))
    b = hip_ctx.batch(contigs, H.reads_from_dicts(out))
    assert b.info().path == abi.PATH_PACKED
    b.close()
    # a coverage hot spot: one tile holds far more than a workgroup's share -> packed (it splits the tile)
    hc, hr = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=3000, n_reads=60000, seed=24)
    b = hip_ctx.batch(hc, hr)
    assert b.info().path == abi.PATH_PACKED
    b.close()


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_pad_rule_pysam_on_random_cigars(path_ctx, seed):
    """MIDAS_SNPS_PAD_PYSAM: the CIGAR op P advances the query position (get_aligned_pairs of the pysam releases of MIDAS's
    time).  Reads with pads then take their bases one position late -- or run past SEQ, which is the IndexError status with
    the lowest such read -- exactly as the oracle does under the same rule, on both device paths."""
    rng = random.Random(seed)
    L = 12000
    reads = []
    for _ in range(1500):
        l = rng.choice([150, 100, 60, rng.randint(20, 300)])
        n_pad = rng.choice([0, 0, 1, 1, 2])
        spare = rng.choice([0, 0, 1, 2, 3]) if seed != 5 else 3          # seed 5: every read has room behind its pads
        body = l - spare
        if body < 8:
            continue
        cuts = sorted(rng.sample(range(2, body - 1), min(n_pad, body - 4)))
        ops, prev = [], 0
        for c in cuts + [body]:
            ops.append((0, c - prev))
            prev = c
            if c != body:
                ops.append((6, 1))                                          # 1P between two match runs
        seq = "".join(rng.choice("ACGT") for _ in range(l))
        reads.append(dict(pos=rng.randint(0, L - 320), cigar=ops, seq=seq, qual=[rng.choice([40, 35, 31, 29, 12]) for _ in range(l)],
                          nm=rng.choice([0, 1, 2]), mapq=42))
    reads.sort(key=lambda r: r['pos'])
    soa = H.reads_from_dicts(reads)
    contig = H.single_contig(L, len(reads), "".join(rng.choice("ACGT") for _ in range(L)))
    thr = abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, aln_cov=0.0, mapid=50.0))
    st0, _, oc0, _, _ = c_oracle.pileup(thr, contig, soa)                  # the specification's rule: never an overrun here
    assert st0 == 0
    c_oracle.set_pad_rule(True)
    path_ctx.set_pad_rule(abi.PAD_PYSAM)
    try:
        st, er, oc, oa, os_ = c_oracle.pileup(thr, contig, soa)
        if st != 0:
            assert st == abi.ERR_READ_CIGAR_OVERRUN
            with pytest.raises(abi.MidasSnpsError) as ei:
                path_ctx.pileup(thr, contig, soa)
            assert (ei.value.status, ei.value.read_index) == (st, er)
        else:
            counts, allele, stats = path_ctx.pileup(thr, contig, soa)
            assert np.array_equal(counts, oc) and np.array_equal(stats, os_)
            assert not np.array_equal(oc, oc0)                               # the rule does change the table
    finally:
        c_oracle.set_pad_rule(False)
        path_ctx.set_pad_rule(abi.PAD_SPEC)
    counts, _, _ = path_ctx.pileup(thr, contig, soa)                        # and back: the specification's table again
    assert np.array_equal(counts, oc0)
