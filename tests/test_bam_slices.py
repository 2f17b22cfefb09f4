"""Rank-local BAM decode (midas_bam_open_slice / _slice_facts / _load_ranges, host-only): a rank walks its share of the
file's bytes, guesses the first record boundary of its slice, and the guess is only trusted when the neighbouring slice's
walk ends on it.  Here: for 1..40 slices of a multi-block BAM the slices chain exactly, their per-reference counts add up to
the whole file's, and range loads return exactly the records of the chosen references -- equal to the full decode."""
import os

import numpy as np
import pytest

from midas_amd import abi, bam, synth


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=4, contig_len=30011, n_reads=60000, seed=7, var_len=True)
    path = str(tmp_path_factory.mktemp("bam") / "genomes.bam")
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    # leave contig 5 without reads and add unmapped records (refID -1) behind the last contig, as a sorted BAM has them
    keep = refid != 5
    from tests.test_gpu_parity import _subset
    reads = _subset(reads, np.nonzero(keep)[0])
    refid = refid[keep]
    n_un = 50
    tail = _subset(reads, np.arange(n_un))
    both = abi.ReadsSoA(**{k: np.concatenate([getattr(reads, k), getattr(tail, k)]) for k in ('pos', 'mapq', 'flag', 'nm', 'l_seq')},
                        seq_off=np.concatenate([reads.seq_off, reads.seq_off[-1] + tail.seq_off[1:]]),
                        qual_off=np.concatenate([reads.qual_off, reads.qual_off[-1] + tail.qual_off[1:]]),
                        cigar_off=np.concatenate([reads.cigar_off, reads.cigar_off[-1] + tail.cigar_off[1:]]),
                        seq4=np.concatenate([reads.seq4, tail.seq4]), qual=np.concatenate([reads.qual, tail.qual]),
                        cigar=np.concatenate([reads.cigar, tail.cigar]))
    rid = np.concatenate([refid, np.full(n_un, -1, np.int32)])
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], rid, both)
    return path, contigs


def _global_first(slices, n_ref):
    first = np.full(n_ref, -1, np.int64)
    for s in slices:
        m = (s.ref_first >= 0) & ((first < 0) | (s.ref_first < first))
        first[m] = s.ref_first[m]
    return first


@pytest.mark.parametrize("n_slices", [1, 2, 3, 8, 40])
def test_slices_chain_and_add_up(sample, n_slices):
    path, contigs = sample
    names, lens, rid, full = abi.read_bam(path)          # the whole-file decoder drops refID -1 records
    assert os.path.getsize(path) > 20 * 65536 // 4        # several BGZF blocks per slice even at 40 slices? at least many blocks
    sl = [abi.BamSlice(path, k, n_slices) for k in range(n_slices)]
    assert sl[0].first == sl[0].rec_begin
    for k in range(n_slices - 1):
        assert sl[k].end == sl[k + 1].first               # every guessed boundary is confirmed by the walk before it
    assert sl[-1].end == sl[-1].total
    assert all(s.sorted == 1 for s in sl)
    assert np.array_equal(sum(s.ref_reads for s in sl), np.bincount(rid, minlength=len(names)))
    assert np.array_equal(sum(s.ref_bases for s in sl), np.bincount(rid, weights=full.l_seq, minlength=len(names)).astype(np.int64))
    first = _global_first(sl, len(names))
    assert first[5] == -1 and (np.delete(first, 5) >= 0).all()
    assert (np.diff(first[first >= 0]) > 0).all()         # coordinate-sorted: references start one after the other
    # records of references 1, 2 (adjacent: one merged range), 7 and 11 (the last: its range runs to the end of the file,
    # unmapped records included -- the decoder drops those)
    have = [i for i in range(len(names)) if first[i] >= 0]

    def rng(c):
        k = have.index(c)
        return (int(first[c]), int(first[have[k + 1]]) if k + 1 < len(have) else sl[0].total)
    want = [1, 2, 7, 11]
    got_rid, got = sl[n_slices // 2].load_ranges([rng(1)[0:1] + rng(2)[1:2], rng(7), rng(11)])
    sel = np.isin(rid, want)
    assert np.array_equal(got_rid, rid[sel])
    idx = np.nonzero(sel)[0]
    assert np.array_equal(got.pos, full.pos[sel]) and np.array_equal(got.nm, full.nm[sel]) and np.array_equal(got.mapq, full.mapq[sel])
    assert np.array_equal(got.l_seq, full.l_seq[sel]) and np.array_equal(got.flag, full.flag[sel])
    for name, off in (("qual", "qual_off"), ("seq4", "seq_off"), ("cigar", "cigar_off")):
        o = getattr(full, off)
        want_bytes = np.concatenate([getattr(full, name)[o[i]:o[i + 1]] for i in idx])
        assert np.array_equal(getattr(got, name), want_bytes), name


def test_an_unsorted_file_says_so(tmp_path):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=3, contig_len=20000, n_reads=9000, seed=3)
    refid = np.repeat(np.arange(3, dtype=np.int32), np.diff(contigs.read_begin))
    refid = refid[::-1].copy()                             # references run backwards: not what `samtools sort` writes
    path = str(tmp_path / "u.bam")
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
    sl = [abi.BamSlice(path, k, 3) for k in range(3)]
    # inside a slice or across slices, the references go backwards somewhere: the host must fall back to a whole decode
    backwards = any(s.sorted == 0 for s in sl) or any(a.last_ref > b.first_ref for a, b in zip(sl, sl[1:]) if a.last_ref >= 0 and b.first_ref >= 0)
    assert backwards


def test_range_that_cuts_a_record_is_an_error(sample):
    path, _ = sample
    s = abi.BamSlice(path, 0, 1)
    with pytest.raises(abi.MidasSnpsError) as ei:
        s.load_ranges([(s.rec_begin, s.rec_begin + 50)])
    assert ei.value.status == abi.ERR_BAD_LAYOUT
    with pytest.raises(abi.MidasSnpsError):
        s.load_ranges([(s.rec_begin - 4, s.rec_begin)])


def test_local_block_tables_of_the_ranks_chain_and_find_the_shares_of_the_full_walk(tmp_path):
    """midas_bam_open_share_local / _share_locate: every rank walks the BGZF chain over its own 1 / N of the file only.  The
    walks must chain (rank 0 from 0, each ending where the next begins, the last at the file's end), and with the bases that
    follows from them every rank finds the share -- first record, total, header -- that the walk of the whole file finds, and
    decodes the same records from it (the table growing along the chain where a share reaches into the next rank's bytes)."""
    from midas_amd import abi, synth
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=5, contig_len=30000, n_reads=60000, seed=33, var_len=True)
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "s.bam")
    abi.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
    size = os.path.getsize(path)
    for n in (1, 2, 3, 5, 8, 64):
        local = [abi.BamShare.open_local(path, k, n) for k in range(n)]
        walks = np.array([s.walk for s in local])
        assert walks[0, 0] == 0 and (walks[:-1, 1] == walks[1:, 0]).all() and walks[-1, 1] == size and (walks[:, 3] == size).all()
        full = [abi.BamShare(path, k, n) for k in range(n)]
        assert int(walks[:, 2].sum()) == full[0].total
        for k in range(n):
            local[k].locate(int(walks[:k, 2].sum()), int(walks[:, 2].sum()))
            assert (local[k].first, local[k].total, local[k].rec_begin) == (full[k].first, full[k].total, full[k].rec_begin), (n, k)
            assert local[k].ref_names == full[k].ref_names and local[k].ref_lens == full[k].ref_lens
        firsts = [s.first for s in full] + [full[0].total]
        if all(f >= 0 for f in firsts) and all(a <= b for a, b in zip(firsts[:-1], firsts[1:])):
            for k in range(n):
                if firsts[k] == firsts[k + 1]:
                    continue
                want_refid, want = full[k].load_ranges([(firsts[k], firsts[k + 1])])
                got_refid, got = local[k].load_ranges([(firsts[k], firsts[k + 1])])
                np.testing.assert_array_equal(want_refid, got_refid)
                for c in abi._SOA_DTYPES:
                    np.testing.assert_array_equal(getattr(want, c), getattr(got, c), err_msg=c)
        # a range in front of a rank's own blocks is refused, not misread
        if n >= 3 and firsts[1] > firsts[0] >= 0:
            with pytest.raises(abi.MidasSnpsError) as ei:
                abi.BamShare.open_local(path, n - 1, n).locate(int(walks[:n - 1, 2].sum()), int(walks[:, 2].sum())).load_ranges([(firsts[0], firsts[1])])
            assert ei.value.status == abi.ERR_INVALID_ARG
        for s in local + full:
            s.close()
