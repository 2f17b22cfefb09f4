"""ONE pass from the BAM's bytes to the pileup kernel's input (midas_bam_load_resident, midas_snps_batch_create_resident): the
device decoder writes the direct layout itself -- a record's [cigar][seq][qual] run copied once, its 16-byte record beside it --
and a batch takes the handle's records where they lie.  Held to the host decode column by column, and to the C oracle / the
batch over the caller's arrays count by count: whole files, runs of records from the middle of a handle, the packed and the long
path of such a batch (which cut their columns out of the handle's stream on first use), a rank's ranges, reads that raise."""
import os

import numpy as np
import pytest

from midas_amd import abi, bam, synth
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with abi.Context(0) as c:
        yield c


THR = abi.Thresholds.from_args(abi.DEFAULT_ARGS)


def _bam_of(tmp_path, contigs, reads, name="s.bam", refid=None):
    path = str(tmp_path / name)
    if refid is None:
        refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
    return path


def _run(batch, path=None):
    if path is not None:
        batch.select_path(path)
    batch.run(THR)
    return batch.fetch()


def test_resident_decode_holds_the_columns_of_the_host_decode(ctx, tmp_path):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=60000, seed=71, var_len=True)
    rng = np.random.default_rng(3)
    nm = reads.nm.copy()
    nm[rng.integers(0, reads.n_reads, 50)] = 300          # NM:i
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    refid[rng.integers(0, reads.n_reads, 500)] = -1       # unmapped records in between: not kept
    path = _bam_of(tmp_path, contigs, abi.ReadsSoA(**{**reads.as_dict(), "nm": nm}), refid=refid)
    for p in (path, os.path.join(H.GOLDEN, "spec_fixture.bam")):
        names_h, lens_h, refid_h, host = abi.read_bam(p)
        names_r, lens_r, refid_r, res = abi.read_bam(p, ctx, resident=True)
        assert isinstance(res, abi.ResidentReads) and names_h == names_r and lens_h == lens_r
        assert res.n_reads == host.n_reads and res.l_seq_total == int(host.l_seq.sum())
        np.testing.assert_array_equal(refid_h, refid_r)
        down = ctx.fetch_payload(res)
        np.testing.assert_array_equal(refid_h, refid_r)     # (the refID view outlives the columns' arrival)
        for k in abi._SOA_DTYPES:
            np.testing.assert_array_equal(getattr(host, k), getattr(down, k), err_msg=k)


def test_a_batch_over_resident_records_counts_what_the_oracle_counts(ctx, tmp_path):
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=3, contig_len=30011, n_reads=90000, seed=72, var_len=True,
                                        lowercase_frac=0.05)
    path = _bam_of(tmp_path, contigs, reads)
    st, _, oc, oa, os_ = c_oracle.pileup(THR, contigs, reads)
    assert st == 0
    _, _, refid, res = abi.read_bam(path, ctx, resident=True)
    sub, read_begin = bam.group_by_contig(contigs.ids, refid, res, contigs.ids)
    assert sub is res
    np.testing.assert_array_equal(read_begin, contigs.read_begin)
    b = ctx.batch(contigs, res)
    info = b.info()
    assert info.path == abi.PATH_DIRECT and info.n_reads == reads.n_reads
    alg = sum((int(l) + 1) // 2 + int(l) + 4 * int(c) + 16 for l, c in zip(reads.l_seq, np.diff(reads.cigar_off))) + 17 * contigs.n_sites
    assert info.algorithmic_bytes == alg
    for want, got in zip((oc, oa, os_), _run(b)):
        np.testing.assert_array_equal(want, got)
    # the other two paths of the same batch: their SEQ / QUAL / CIGAR columns are cut out of the handle's stream on first use
    for p in (abi.PATH_PACKED, abi.PATH_LONG, abi.PATH_DIRECT):
        for want, got in zip((oc, oa, os_), _run(b, p)):
            np.testing.assert_array_equal(want, got, err_msg=abi.PATH_NAMES[p])
    b.close()


def test_runs_of_a_resident_handle_make_batches_of_their_own(ctx, tmp_path):
    """Batches over records [first, first + n) of one handle -- what a rank does when its contigs go up in several batches."""
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=4, contig_len=20000, n_reads=50000, seed=73, var_len=True)
    path = _bam_of(tmp_path, contigs, reads)
    _, _, oc, oa, os_ = c_oracle.pileup(THR, contigs, reads)
    _, _, refid, res = abi.read_bam(path, ctx, resident=True)
    off = contigs.site_offsets()
    stats = np.zeros_like(os_)
    for lo, hi in ((0, 3), (3, 4), (4, 8)):
        ids = contigs.ids[lo:hi]
        sub, rb = bam.group_by_contig(contigs.ids, refid, res, ids)
        assert isinstance(sub, abi.ResidentReads) and sub.first == int(contigs.read_begin[lo]) and sub.n_reads == int(rb[-1])
        table = abi.ContigTable(length=contigs.length[lo:hi], species=contigs.species[lo:hi], read_begin=rb,
                                ref=contigs.ref[off[lo]:off[hi]], n_species=contigs.n_species)
        for path_ in (abi.PATH_DIRECT, abi.PATH_PACKED):
            b = ctx.batch(table, sub)
            counts, allele, st = _run(b, path_)
            b.close()
            np.testing.assert_array_equal(counts, oc[off[lo]:off[hi]])
            np.testing.assert_array_equal(allele, oa[off[lo]:off[hi]])
        stats += st
    np.testing.assert_array_equal(stats, os_)
    # records that are not one run (a contig in the middle left out): the host regroups them, from the columns
    ids = [contigs.ids[0], contigs.ids[2]]
    sub, rb = bam.group_by_contig(contigs.ids, refid, res, ids, fetch=ctx.fetch_payload)
    assert isinstance(sub, abi.ReadsSoA) and sub.n_reads == int(rb[-1])


def test_reads_that_raise_are_reported_from_resident_batches_too(ctx, tmp_path):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=4000, seed=74)
    nm = reads.nm.copy()
    nm[1234] = -1
    path = _bam_of(tmp_path, contigs, abi.ReadsSoA(**{**reads.as_dict(), "nm": nm}))
    _, _, refid, res = abi.read_bam(path, ctx, resident=True)
    b = ctx.batch(contigs, res)
    with pytest.raises(abi.MidasSnpsError) as ei:
        _run(b)
    assert ei.value.status == abi.ERR_READ_NO_NM and ei.value.read_index == 1234
    b.close()


def test_long_reads_in_a_resident_handle_take_the_long_path(ctx, tmp_path):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=30000, n_reads=600, read_len=1500, seed=75)
    path = _bam_of(tmp_path, contigs, reads)
    _, _, oc, oa, os_ = c_oracle.pileup(THR, contigs, reads)
    _, _, refid, res = abi.read_bam(path, ctx, resident=True)
    b = ctx.batch(contigs, res)
    assert b.info().path == abi.PATH_LONG
    for want, got in zip((oc, oa, os_), _run(b)):
        np.testing.assert_array_equal(want, got)
    b.close()


def test_a_ranks_ranges_stay_resident(ctx, tmp_path):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=60000, seed=76, var_len=True)
    path = _bam_of(tmp_path, contigs, reads)
    sl = abi.BamSlice(path, 1, 3)
    mid = int(sl.ref_first[sl.ref_first >= 0][-1])
    for ranges in ([(sl.first, sl.end)], [(sl.first, mid), (mid, sl.end)], []):
        want_refid, want = sl.load_ranges(ranges)
        got_refid, got = abi.BamSlice(path, 1, 3).load_ranges(ranges, ctx, resident=True)
        assert isinstance(got, abi.ResidentReads) or not ranges
        np.testing.assert_array_equal(want_refid, got_refid)
        down = ctx.fetch_payload(got) if ranges else got
        for k in abi._SOA_DTYPES:
            np.testing.assert_array_equal(getattr(want, k), getattr(down, k), err_msg=k)
    with pytest.raises(abi.MidasSnpsError) as ei:
        abi.BamSlice(path, 1, 3).load_ranges([(sl.first, sl.end - 7)], ctx, resident=True)
    assert ei.value.status == abi.ERR_BAD_LAYOUT
