"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle.

Bit-exact bar (integer / byte work): per-site A,C,G,T, ref allele, and the four per-species
counters must be IDENTICAL to the oracle's on the same seeded inputs.  The two floating-point
expressions of the path (keep_read's mapid / aln_cov ratios) are IEEE fp64 divisions on both
sides and only feed a comparison, so they are covered by the same bit-exact bar (tolerance 0).
"""
import numpy as np
import pytest

from midas_amd import abi, synth
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = H.load_kat_cases()


def _assert_same(ctx, thr, contigs, reads):
    st, er, oc, oa, os_ = c_oracle.pileup(thr, contigs, reads)
    assert st == 0, "oracle refused the input (%d at read %d)" % (st, er)
    counts, allele, stats = ctx.pileup(thr, contigs, reads)
    bad = np.nonzero((counts != oc).any(axis=1))[0]
    assert bad.size == 0, "counts differ at %d sites, first %s: hip %s oracle %s" % (
        bad.size, bad[:5], counts[bad[:5]].tolist(), oc[bad[:5]].tolist())
    np.testing.assert_array_equal(allele, oa)
    np.testing.assert_array_equal(stats, os_)
    return counts, stats


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hand_derived_cases_through_c_abi(hip_ctx, case):
    contigs, reads, thr, _ = H.kat_inputs(case)
    hip_ctx.set_pad_rule(abi.PAD_PYSAM if H.kat_pysam_pad_rule(case) else abi.PAD_SPEC)
    try:
        if "error" in case:
            with pytest.raises(abi.MidasSnpsError) as ei:
                hip_ctx.pileup(thr, contigs, reads)
            assert ei.value.status == case["error"]
            assert ei.value.read_index == case.get("error_read", 0)
            return
        counts, allele, stats = hip_ctx.pileup(thr, contigs, reads)
    finally:
        hip_ctx.set_pad_rule(abi.PAD_SPEC)
    np.testing.assert_array_equal(counts, H.kat_expected_counts(case))
    np.testing.assert_array_equal(stats, H.kat_expected_stats(case))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_seeded_uniform_reads(hip_ctx, thr_default, seed):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=30000, seed=seed)
    _assert_same(hip_ctx, thr_default, contigs, reads)


def test_ragged_lengths_lowercase_ref_tile_edges(hip_ctx, thr_default):
    # contig length not a multiple of the 4096-site tile; trimmed reads; lower-case reference letters
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=5, contig_len=30011, n_reads=40000,
                                        seed=7, var_len=True, lowercase_frac=0.1)
    _assert_same(hip_ctx, thr_default, contigs, reads)


def test_deep_hotspot_exceeds_u16(hip_ctx, thr_default):
    # ~1000x on a small contig: several sites above 65 535? no -- but far above 255; u32 planes must hold
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=120000, seed=9)
    counts, _ = _assert_same(hip_ctx, thr_default, contigs, reads)
    assert counts.sum(axis=1).max() > 255


def test_hot_spot_among_ordinary_tiles_is_split_and_stays_exact(hip_ctx, thr_default):
    # one 3 kb window at ~2500x inside an otherwise 8x genome: its tiles hold far more than a workgroup's fair share
    # of the reads and are processed as parts merged with atomics -- counts, alleles, covered bases and depth must not
    # notice, run after run (the part counters and the zeroed outputs reset themselves)
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=13000, seed=23,
                                        var_len=True)
    hot_c, hot_r = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=3000, n_reads=50000, seed=24)
    # graft the hot reads onto contig 1 at offset 17000 (BAM order: by contig, then position)
    shift = 17000
    d, h = reads.as_dict(), hot_r.as_dict()
    lo, hi = int(contigs.read_begin[1]), int(contigs.read_begin[2])
    pos = np.concatenate([d['pos'][lo:hi], h['pos'] + shift])
    order = np.argsort(pos, kind='stable')
    def splice(key, off_key=None):
        if off_key is None:
            mid = np.concatenate([d[key][lo:hi], h[key] + (shift if key == 'pos' else 0)])[order]
            return np.concatenate([d[key][:lo], mid, d[key][hi:]])
    def ragged(key, off_key):
        a = [d[key][d[off_key][i]:d[off_key][i + 1]] for i in range(len(d['pos']))]
        b = [h[key][h[off_key][i]:h[off_key][i + 1]] for i in range(len(h['pos']))]
        mid = [(a[lo:hi] + b)[j] for j in order]
        seqs = a[:lo] + mid + a[hi:]
        offs = np.zeros(len(seqs) + 1, np.int64)
        offs[1:] = np.cumsum([len(x) for x in seqs])
        return np.concatenate(seqs), offs
    new = {k: splice(k) for k in ('pos', 'mapq', 'flag', 'nm', 'l_seq')}
    new['seq4'], new['seq_off'] = ragged('seq4', 'seq_off')
    new['qual'], new['qual_off'] = ragged('qual', 'qual_off')
    new['cigar'], new['cigar_off'] = ragged('cigar', 'cigar_off')
    merged = abi.ReadsSoA(**new)
    rb = contigs.read_begin.copy()
    rb[2:] += len(h['pos'])
    contigs.read_begin = rb
    b = hip_ctx.batch(contigs, merged)
    info = b.info()
    assert info.n_work_items > info.n_tiles          # the hot tiles were split
    from oracle import c_oracle
    st, _, oc, oa, os_ = c_oracle.pileup(thr_default, contigs, merged)
    assert st == 0
    for _ in range(3):
        b.run(thr_default)
        counts, allele, stats = b.fetch()
        assert np.array_equal(counts, oc) and np.array_equal(allele, oa) and np.array_equal(stats, os_)
    assert counts.sum(axis=1).max() > 1500
    b.close()


@pytest.mark.parametrize("args", [
    dict(baseq=0), dict(baseq=41), dict(mapq=0, readq=0, mapid=1.0, aln_cov=0.0),
    dict(mapid=99.0), dict(aln_cov=1.0), dict(readq=38), dict(mapq=42),
])
def test_threshold_sweep(hip_ctx, args):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=3, contig_len=20000, n_reads=15000, seed=11)
    a = dict(abi.DEFAULT_ARGS)
    a.update(args)
    _assert_same(hip_ctx, abi.Thresholds.from_args(a), contigs, reads)


def test_empty_inputs(hip_ctx, thr_default):
    # a species with no reads at all: all-zero rows, counters 0
    contigs = abi.ContigTable(length=[5000, 100], species=[0, 1], read_begin=[0, 0, 0],
                              ref=np.frombuffer(b"acgtn" * 1020, dtype=np.uint8), n_species=2)
    counts, allele, stats = hip_ctx.pileup(thr_default, contigs, abi.ReadsSoA.empty())
    assert counts.shape == (5100, 4) and not counts.any()
    assert bytes(allele[:10]) == b"ACGTNACGTN"
    assert not stats.any()


def test_one_site_contigs_and_many_small_contigs(hip_ctx, thr_default):
    rng = np.random.default_rng(5)
    reads = []
    lengths = [1, 2, 3, 150, 151, 4095, 4096, 4097, 1]
    read_begin = [0]
    for ln in lengths:
        n = 0
        for _ in range(6):
            l = int(rng.integers(1, 160))
            pos = int(rng.integers(0, ln))
            seq = "".join("ACGT"[i] for i in rng.integers(0, 4, size=l))
            reads.append(dict(pos=pos, cigar="%dM" % l, seq=seq))
            n += 1
        read_begin.append(read_begin[-1] + n)
    soa = H.reads_from_dicts(reads)
    g = sum(lengths)
    contigs = abi.ContigTable(length=lengths, species=[0] * len(lengths), read_begin=read_begin,
                              ref=np.frombuffer(b"A" * g, dtype=np.uint8), n_species=1)
    _assert_same(hip_ctx, thr_default, contigs, soa)


def test_long_deletions_refskips_and_clips_across_tiles(hip_ctx):
    """Reads whose reference span crosses one or many tile borders; hard clips, pads, N-skips."""
    rng = np.random.default_rng(3)
    L = 20000
    reads = []

    def rs(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))

    reads.append(dict(pos=4000, cigar="50M9000N50M", seq=rs(100), nm=0))        # spans 3 tiles, lands in #0 and #3
    reads.append(dict(pos=4090, cigar="10M2D10M", seq=rs(20), nm=2))             # straddles the 4096 border
    reads.append(dict(pos=4095, cigar="1M", seq="G"))
    reads.append(dict(pos=4096, cigar="1M", seq="T"))
    reads.append(dict(pos=8100, cigar="5H20S100M3I27M10S2H", seq=rs(160), nm=3))  # clips both ends, 160 bases
    reads.append(dict(pos=8190, cigar="4M1P4M", seq=rs(8)))
    reads.append(dict(pos=12000, cigar="30=5X30=", seq=rs(65), nm=5))
    reads.append(dict(pos=19990, cigar="30M", seq=rs(30)))                        # hangs over the contig end
    for _ in range(300):
        l = int(rng.integers(30, 200))
        a = int(rng.integers(5, l - 10))
        d = int(rng.integers(1, 3000))
        reads.append(dict(pos=int(rng.integers(0, L - 200)), cigar="%dM%dD%dM" % (a, d, l - a), seq=rs(l), nm=0))
    reads.sort(key=lambda r: r["pos"])
    soa = H.reads_from_dicts(reads)
    contigs = H.single_contig(L, soa.n_reads, ref=rs(L))
    a = dict(abi.DEFAULT_ARGS)
    a.update(mapid=1.0)
    _assert_same(hip_ctx, abi.Thresholds.from_args(a), contigs, soa)


def test_unsorted_input_is_still_exact(hip_ctx, thr_default):
    """Sortedness only tightens the per-tile read ranges; a shuffled contig must give the same table."""
    from oracle import pileup_oracle as po
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=30000, n_reads=5000, seed=21)
    perm = np.random.default_rng(0).permutation(reads.n_reads)
    objs = po.alns_from_soa(reads.as_dict())
    shuffled = [dict(pos=o.pos, cigar=o.cigar, seq=o.seq, qual=list(o.qual), nm=o.nm, mapq=o.mapq, flag=o.flag)
                for o in (objs[j] for j in perm)]
    soa = H.reads_from_dicts(shuffled)
    c_sorted, _ = _assert_same(hip_ctx, thr_default, contigs, reads)
    c_shuf, _ = _assert_same(hip_ctx, thr_default, contigs, soa)
    np.testing.assert_array_equal(c_sorted, c_shuf)


def test_batch_rerun_is_idempotent_and_thresholds_switch(hip_ctx):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=25000, n_reads=20000, seed=31)
    b = hip_ctx.batch(contigs, reads)
    t1 = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    a2 = dict(abi.DEFAULT_ARGS)
    a2.update(baseq=10, mapq=0)
    t2 = abi.Thresholds.from_args(a2)
    b.run(t1)
    c1, _, s1 = b.fetch()
    b.run(t2)
    c2, _, s2 = b.fetch()
    b.run(t1)
    c3, _, s3 = b.fetch()
    np.testing.assert_array_equal(c1, c3)
    np.testing.assert_array_equal(s1, s3)
    assert c2.sum() > c1.sum()
    for thr, c, s in ((t1, c1, s1), (t2, c2, s2)):
        st, _, oc, _, os_ = c_oracle.pileup(thr, contigs, reads)
        assert st == 0
        np.testing.assert_array_equal(c, oc)
        np.testing.assert_array_equal(s, os_)
    b.close()


def test_long_reads_up_to_1024(hip_ctx, thr_default):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=50000, n_reads=3000,
                                        read_len=1000, seed=41, var_len=True)
    _assert_same(hip_ctx, thr_default, contigs, reads)


@pytest.mark.parametrize("read_len,var_len", [(40, True), (16, False), (17, False), (33, False)])
def test_short_reads(hip_ctx, thr_default, read_len, var_len):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=5000, n_reads=4000,
                                        read_len=read_len, seed=43, var_len=var_len)
    _assert_same(hip_ctx, thr_default, contigs, reads)


@pytest.mark.parametrize("read_len", [32, 64, 96, 125, 128, 151, 160, 250])
def test_read_lengths_on_both_sides_of_the_lane_size(hip_ctx, thr_default, read_len):
    """A batch packs 31 bases per lane, or 32 where that saves a lane per read (32, 64, 96, 125, 128, 160 here): both
    layouts, reads straddling tiles, indels and clips included."""
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=9000, n_reads=5000,
                                        read_len=read_len, seed=47 + read_len, var_len=False)
    _assert_same(hip_ctx, thr_default, contigs, reads)
    b = hip_ctx.batch(contigs, reads)
    # packed path: 31 bases per lane, or 32 where that saves a lane; direct path: 30, or 32 where that saves a lane
    b.select_path(abi.PATH_PACKED)
    assert b.info().lanes_per_read == {32: 1, 64: 2, 96: 3, 125: 4, 128: 4, 151: 5, 160: 5, 250: 9}[read_len]
    b.select_path(abi.PATH_DIRECT)
    assert b.info().lanes_per_read == {32: 1, 64: 2, 96: 3, 125: 4, 128: 4, 151: 5, 160: 5, 250: 8}[read_len]
    assert b.info().lane_bases == 32
    b.close()


def test_unsupported_and_malformed_inputs_are_statuses_not_crashes(hip_ctx, thr_default):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=5000, n_reads=100, seed=1)
    bad = abi.ContigTable(length=contigs.length, species=contigs.species, read_begin=[0, 99], ref=contigs.ref, n_species=1)
    with pytest.raises(abi.MidasSnpsError) as ei:
        hip_ctx.pileup(thr_default, bad, reads)
    assert ei.value.status == abi.ERR_BAD_LAYOUT
    zero = abi.ContigTable(length=[0], species=[0], read_begin=[0, 0], ref=np.zeros(0, np.uint8), n_species=1)
    with pytest.raises(abi.MidasSnpsError) as ei:
        hip_ctx.pileup(thr_default, zero, abi.ReadsSoA.empty())
    assert ei.value.status == abi.ERR_UNSUPPORTED
    # (a 2 000-base read is beyond the fast paths, not beyond the library: the batch runs on the long path, tests/test_gpu_long.py)
    long_read = H.reads_from_dicts([dict(pos=0, cigar="2000M", seq="A" * 2000)])
    _assert_same(hip_ctx, thr_default, H.single_contig(5000, 1), long_read)


def test_qualities_above_62_are_exact_at_every_baseq_on_the_packed_path_too(hip_ctx):
    """The PACKED path keeps a base's quality in six bits (layout.h): Phred 63..93 are stored as 62, which changes no comparison
    with a baseq <= 62; a baseq above 62 on a batch that holds such qualities is run by the direct kernel, which compares the
    quality bytes themselves (round 5; until then it was a status).  Batches without such qualities take any baseq on the
    packed kernel (nothing passes above their maximum, as in the reference)."""
    hip_ctx.set_default_path(abi.PATH_PACKED)
    try:
        _packed_quality_limits(hip_ctx)
    finally:
        hip_ctx.set_default_path(abi.PATH_AUTO)


def _packed_quality_limits(hip_ctx):
    rng = np.random.default_rng(5)
    L = 6000
    reads = []
    for k in range(800):
        l = int(rng.integers(30, 151))
        reads.append(dict(pos=int(rng.integers(0, L - 160)), cigar="%dM" % l, seq="".join("ACGT"[i] for i in rng.integers(0, 4, l)),
                          qual=[int(x) for x in rng.choice([0, 1, 30, 40, 61, 62, 63, 64, 70, 93], size=l)], nm=0))
    reads.sort(key=lambda r: r['pos'])
    soa = H.reads_from_dicts(reads)
    contig = H.single_contig(L, len(reads), "".join("ACGT"[i] for i in rng.integers(0, 4, L)))
    for bq in (0, 1, 41, 61, 62):
        _assert_same(hip_ctx, abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, baseq=bq, readq=0)), contig, soa)
    for bq in (63, 64, 70, 93, 94):
        _assert_same(hip_ctx, abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, baseq=bq, readq=0)), contig, soa)
    low = H.reads_from_dicts([dict(r, qual=[min(q, 62) for q in r['qual']]) for r in reads])
    for bq in (62, 63, 200):
        counts, _ = _assert_same(hip_ctx, abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, baseq=bq, readq=0)), contig, low)
        assert (counts.sum() > 0) == (bq == 62)


# ---- full-size properties (BASELINE configs[1]); the oracle also finishes it in ~1 s, so compare too -----

@pytest.fixture(scope="module")
def c2():
    return synth.make_dataset(**synth.CONFIGS['c2'])


def test_c2_full_size_bit_exact_and_conserved(hip_ctx, thr_default, c2):
    contigs, reads = c2
    counts, stats = _assert_same(hip_ctx, thr_default, contigs, reads)
    depth = counts.sum(axis=1, dtype=np.int64)
    # checksum of checksums: the per-species counters are the reduction of the per-site table
    assert int(depth.sum()) == int(stats[0, abi.STAT_TOTAL_DEPTH])
    assert int((depth > 0).sum()) == int(stats[0, abi.STAT_COVERED_BASES])
    assert int(stats[0, abi.STAT_ALIGNED_READS]) == reads.n_reads
    assert 0 < int(stats[0, abi.STAT_MAPPED_READS]) < reads.n_reads


def test_c2_linearity_in_the_read_set(hip_ctx, thr_default, c2):
    """counts(A u B) == counts(A) + counts(B): split the reads of every contig into even / odd halves."""
    contigs, reads = c2
    full, _, sfull = hip_ctx.pileup(thr_default, contigs, reads, want_allele=False)
    acc = np.zeros_like(full)
    sacc = np.zeros_like(sfull)
    idx = np.arange(reads.n_reads)
    contig_of = np.searchsorted(contigs.read_begin, idx, side='right') - 1
    for par in (0, 1):
        sel = idx[(idx & 1) == par]
        sub = _subset(reads, sel)
        rb = np.zeros(contigs.n_contigs + 1, dtype=np.int64)
        np.cumsum(np.bincount(contig_of[sel], minlength=contigs.n_contigs), out=rb[1:])
        sc = abi.ContigTable(length=contigs.length, species=contigs.species, read_begin=rb, ref=contigs.ref, n_species=1)
        c, _, s = hip_ctx.pileup(thr_default, sc, sub, want_allele=False)
        acc += c
        sacc[:, :2] += s[:, :2]
    np.testing.assert_array_equal(acc, full)
    np.testing.assert_array_equal(sacc[:, :2], sfull[:, :2])


# ---- the two larger single-GPU workloads: BASELINE configs[2] and one rank's share of configs[3] -------------------
# (20 species / 80 Mb / 10.67 M reads at 20x, and 13 species / 52 Mb / 10.4 M reads at 30x; the C oracle needs ~8 s)

@pytest.mark.parametrize("name", ["c3", "c4_rank"])
def test_full_size_configs_bit_exact_and_conserved(hip_ctx, thr_default, name):
    contigs, reads = synth.make_dataset(**synth.CONFIGS[name])
    counts, stats = _assert_same(hip_ctx, thr_default, contigs, reads)
    # checksum of checksums per species: the counters are the reduction of that species' rows of the per-site table
    off = contigs.site_offsets()
    depth = counts.sum(axis=1, dtype=np.int64)
    sp_depth = np.zeros(contigs.n_species, np.int64)
    sp_cov = np.zeros(contigs.n_species, np.int64)
    sp_reads = np.zeros(contigs.n_species, np.int64)
    for k in range(contigs.n_contigs):
        s = int(contigs.species[k])
        d = depth[off[k]:off[k + 1]]
        sp_depth[s] += int(d.sum())
        sp_cov[s] += int((d > 0).sum())
        sp_reads[s] += int(contigs.read_begin[k + 1] - contigs.read_begin[k])
    np.testing.assert_array_equal(stats[:, abi.STAT_TOTAL_DEPTH], sp_depth)
    np.testing.assert_array_equal(stats[:, abi.STAT_COVERED_BASES], sp_cov)
    np.testing.assert_array_equal(stats[:, abi.STAT_ALIGNED_READS], sp_reads)
    assert (stats[:, abi.STAT_MAPPED_READS] > 0).all() and (stats[:, abi.STAT_MAPPED_READS] < sp_reads).all()


def _subset(reads, sel):
    """Sub-select records of a ReadsSoA (fixed-stride payloads not assumed)."""
    def gather(data, off, per):
        lens = (off[1:] - off[:-1])[sel]
        new_off = np.zeros(sel.size + 1, dtype=np.int64)
        np.cumsum(lens, out=new_off[1:])
        starts = off[:-1][sel]
        ix = np.repeat(starts - new_off[:-1], lens) + np.arange(new_off[-1])
        return data[ix], new_off
    seq4, seq_off = gather(reads.seq4, reads.seq_off, None)
    qual, qual_off = gather(reads.qual, reads.qual_off, None)
    cigar, cigar_off = gather(reads.cigar, reads.cigar_off, None)
    return abi.ReadsSoA(pos=reads.pos[sel], mapq=reads.mapq[sel], flag=reads.flag[sel], nm=reads.nm[sel],
                        l_seq=reads.l_seq[sel], seq_off=seq_off, qual_off=qual_off, cigar_off=cigar_off,
                        seq4=seq4, qual=qual, cigar=cigar)


def _random_cigar(rng, l):
    """A CIGAR for a query of l bases drawn from a grammar that covers what the packer must tell apart: regular
    (clips at the ends around M/=/X/I/D/N), and irregular in every way it knows (P ops, several clips at one end, a clip
    in the middle, hard clips, zero-length ops, a query length shorter than l_seq, more segments than it serves)."""
    kind = rng.random()
    ops = []
    q = 0

    def match(n):
        nonlocal q
        ops.append((rng.choice([0, 0, 0, 7, 8]), n))
        q += n
    lead = rng.randint(0, l // 5) if rng.random() < 0.3 else 0
    trail = rng.randint(0, l // 5) if rng.random() < 0.3 else 0
    if rng.random() < 0.1:
        ops.append((5, rng.randint(1, 9)))                       # H
    if lead:
        if kind > 0.97:
            a = rng.randint(0, lead)
            ops += [(4, a), (4, lead - a)]                       # two S at one end (maybe zero-length)
        else:
            ops.append((4, lead))
        q += lead
    body = l - lead - trail
    n_events = rng.choice([0, 0, 0, 1, 1, 2, 3, 8]) if kind < 0.9 else rng.choice([1, 2])
    cuts = sorted(rng.sample(range(1, body), min(n_events, max(0, body - 1)))) if body > 1 else []
    prev = 0
    for c in cuts + [body]:
        if c - prev > 0:
            match(c - prev)
        prev = c
        if c != body:
            ev = rng.random()
            if ev < 0.35:
                ins = min(rng.randint(1, 4), body - c - 1) if body - c > 1 else 0
                if ins > 0:
                    ops.append((1, ins)); q += ins; prev = c + ins
            elif ev < 0.7:
                ops.append((2, rng.randint(1, 6)))
            elif ev < 0.85:
                ops.append((3, rng.choice([1, 50, 4000, 9000])))     # N skip, possibly across tiles
            elif ev < 0.92:
                ops.append((6, rng.randint(1, 3)))                    # P
            elif ev < 0.96:
                ops.append((4, 0) if rng.random() < 0.5 else (5, 2))  # a clip in the middle (zero-length S, or H)
            else:
                ops.append((2, 0))                                    # zero-length D
    # the walk above may have consumed fewer query bases than planned when insertions were clipped
    if trail:
        ops.append((4, trail)); q += trail
    if rng.random() < 0.1:
        ops.append((5, rng.randint(1, 9)))
    if q < l:
        if kind > 0.94 and kind <= 0.97:
            pass                                                      # a CIGAR that covers fewer bases than l_seq
        else:
            ops.insert(len(ops) - (1 if trail else 0) - (1 if ops and ops[-1][0] == 5 else 0), (0, l - q))
    return [(op, n) for op, n in ops], sum(n for op, n in ops if op in (0, 1, 4, 7, 8))


@pytest.mark.parametrize("seed", [99, 7, 2026])
def test_random_cigar_grammar_matches_oracle(hip_ctx, seed):
    import random
    rng = random.Random(seed)
    L = 30000
    reads = []
    for _ in range(6000):
        l = rng.choice([150, 150, 150, 100, 60, 33, rng.randint(20, 400)])
        while True:
            cigar, qlen = _random_cigar(rng, l)
            if qlen <= l and any(op in (0, 7, 8) and n > 0 for op, n in cigar):
                break
        pos = rng.choice([rng.randint(0, L - 1), rng.randint(4000, 4200), rng.randint(L - 300, L - 1), rng.randint(8100, 8250)])
        seq = "".join(rng.choice("ACGTACGTACGTN") for _ in range(l))
        qual = [rng.choice([40, 38, 35, 31, 30, 29, 12, 2]) for _ in range(l)]
        reads.append(dict(pos=pos, cigar=cigar, seq=seq, qual=qual, nm=rng.choice([0, 1, 2, 5, 9, 30, 1023, 1024, 3000]),
                          mapq=rng.choice([42, 42, 30, 20, 19, 3])))
    reads.sort(key=lambda r: r['pos'])
    soa = H.reads_from_dicts(reads)
    ref = "".join(rng.choice("ACGTacgtN") for _ in range(L))
    contig = H.single_contig(L, len(reads), ref)
    for args in (dict(abi.DEFAULT_ARGS), dict(abi.DEFAULT_ARGS, baseq=0, mapid=80.0, aln_cov=0.3, readq=10, mapq=0)):
        counts, stats = _assert_same(hip_ctx, abi.Thresholds.from_args(args), contig, soa)
        assert int(stats[0, abi.STAT_MAPPED_READS]) > 500
