"""Inputs the reference accepts without limits (pysam streams whatever the BAM holds, midas/run/snps.py:187-199) and the fast
device paths do not take: reads longer than 1024 bases, CIGARs of more than 65 534 ops, NM above 65 534 -- such a batch runs on
the long path (pileup_long.hip) and gives the oracle's table -- and a base-quality threshold above 62 on a batch of the packed
path that holds qualities above 62 (the packed payload keeps six bits: that run goes through the direct kernel)."""
import numpy as np
import pytest

from midas_amd import abi, synth
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _same(ctx, thr, contigs, reads, want_path=None):
    st, er, oc, oa, os_ = c_oracle.pileup(thr, contigs, reads)
    assert st == 0, (st, er)
    b = ctx.batch(contigs, reads)
    if want_path is not None:
        assert b.info().path == want_path
    b.run(thr)
    counts, allele, stats = b.fetch()
    b.close()
    assert np.array_equal(counts, oc) and np.array_equal(allele, oa) and np.array_equal(stats, os_)
    return counts


def test_five_kilobase_reads_take_the_long_path(hip_ctx, thr_default):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=60000, n_reads=400, read_len=5000, seed=51)
    assert int(reads.l_seq.max()) == 5000
    counts = _same(hip_ctx, thr_default, contigs, reads, abi.PATH_LONG)
    assert counts.sum() > 0
    b = hip_ctx.batch(contigs, reads)
    for path in (abi.PATH_DIRECT, abi.PATH_PACKED):
        with pytest.raises(abi.MidasSnpsError) as ei:
            b.select_path(path)
        assert ei.value.status == abi.ERR_UNSUPPORTED
    b.select_path(abi.PATH_AUTO)
    assert b.info().path == abi.PATH_LONG
    b.close()
    # the one-shot entry point, and loose thresholds
    loose = abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, baseq=0, readq=0, mapq=0, mapid=0.0, aln_cov=0.0))
    st, _, oc, oa, os_ = c_oracle.pileup(loose, contigs, reads)
    counts, allele, stats = hip_ctx.pileup(loose, contigs, reads)
    assert st == 0 and np.array_equal(counts, oc) and np.array_equal(stats, os_)


def test_one_long_read_among_short_ones_and_huge_nm_and_cigar(hip_ctx):
    rng = np.random.default_rng(8)
    L = 50000

    def rs(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    reads = [dict(pos=int(p), cigar="150M", seq=rs(150), nm=int(rng.integers(0, 4))) for p in rng.integers(0, L - 200, size=3000)]
    reads.append(dict(pos=1000, cigar="700M1D600M2I700M", seq=rs(2002), nm=3))                       # l_seq 2002
    reads.append(dict(pos=20000, cigar="100M", seq=rs(100), nm=70000))                               # NM beyond 16 bits: pid < mapid, dropped
    many = [(0, 1), (2, 1)] * 33000 + [(0, 1)]                                                       # 66 001 ops: 1M1D ... 1M
    reads.append(dict(pos=5000, cigar=many, seq=rs(33001), nm=33000))
    reads.sort(key=lambda r: r["pos"])
    soa = H.reads_from_dicts(reads)
    contig = H.single_contig(L, soa.n_reads, rs(L))
    for args in (dict(abi.DEFAULT_ARGS), dict(abi.DEFAULT_ARGS, mapid=0.0, aln_cov=0.0)):
        _same(hip_ctx, abi.Thresholds.from_args(args), contig, soa, abi.PATH_LONG)


def test_errors_of_long_reads_are_the_oracles(hip_ctx, thr_default):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=40000, n_reads=300, read_len=3000, seed=52)
    d = {k: v.copy() for k, v in reads.as_dict().items()}
    d['nm'][200] = -1
    d['nm'][37] = -1
    bad = abi.ReadsSoA(**d)
    st, er, *_ = c_oracle.pileup(thr_default, contigs, bad)
    with pytest.raises(abi.MidasSnpsError) as ei:
        hip_ctx.pileup(thr_default, contigs, bad)
    assert (ei.value.status, ei.value.read_index) == (st, er) == (abi.ERR_READ_NO_NM, 37)


def test_baseq_above_62_on_a_hot_spot_batch_of_the_packed_path(hip_ctx):
    """BAM allows qualities up to 93.  A coverage hot spot makes `auto` choose the packed path, whose payload keeps six bits of
    a quality; a threshold of 70 on such reads is run by the direct kernel instead of being refused, and the batch stays
    packed for the thresholds the payload can serve."""
    hc, hr = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=3000, n_reads=60000, seed=24)
    d = {k: v.copy() for k, v in hr.as_dict().items()}
    rng = np.random.default_rng(2)
    d['qual'] = rng.choice(np.array([93, 80, 71, 70, 69, 63, 62, 40, 30, 2], dtype=np.uint8), size=d['qual'].size)
    reads = abi.ReadsSoA(**d)
    b = hip_ctx.batch(hc, reads)
    assert b.info().path == abi.PATH_PACKED
    for baseq in (70, 30, 63, 93, 94, 62):
        thr = abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, baseq=baseq, readq=0))
        st, _, oc, oa, os_ = c_oracle.pileup(thr, hc, reads)
        assert st == 0
        b.run(thr)
        counts, allele, stats = b.fetch()
        assert np.array_equal(counts, oc) and np.array_equal(stats, os_), baseq
        assert b.info().path == abi.PATH_PACKED
    b.close()
