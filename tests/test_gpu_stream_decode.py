"""The streamed device decode (snps_abi.hip device_decode_stream): a BAM of several device-fills of BGZF blocks decoded resident
GROUP BY GROUP -- upload of group g + 1 under the kernels of group g, the columns and the direct layout of every group written
behind those of the groups before it -- instead of in one arena of ~2.3 x the file's inflated bytes.  The reference streams the
file record by record behind pysam.AlignmentFile (midas/run/snps.py:186-199): however the device cuts the file up, the records,
their order and every count must be those of the one-arena decode and of the host decode.

The groups are forced small here (MIDAS_SNPS_DECODE_GROUP_BLOCKS) so that a BAM of a few hundred blocks is many groups: records
that straddle group borders at every border, one / two / three slots, the result's arrays grown after a first estimate that is
short, unmapped records, a damaged block in a late group, a record longer than what a group keeps behind its end (the decode
falls back to one arena), ranges of a rank."""
import os

import numpy as np
import pytest

from midas_amd import abi, bam, synth
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

THR = abi.Thresholds.from_args(abi.DEFAULT_ARGS)


@pytest.fixture(scope="module")
def ctx():
    with abi.Context(0) as c:
        yield c


def _bam_of(tmp_path, contigs, reads, name="s.bam", refid=None, names=None):
    path = str(tmp_path / name)
    if refid is None:
        refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads, **({"read_names": names} if names else {}))
    return path


def _streamed(capfd):
    """how many groups the last decode went through (0: one arena), read from the library's trace"""
    err = capfd.readouterr().err
    for line in err.splitlines():
        if line.startswith("[device decode] streamed:") and "groups of" in line:
            return int(line.split("streamed:")[1].split("groups")[0])
    return 0


@pytest.mark.parametrize("group_blocks,slots", [(7, 1), (7, 2), (23, 2), (16, 3), (60, 2)])
def test_streamed_decode_is_the_one_arena_decode(ctx, tmp_path, monkeypatch, capfd, group_blocks, slots):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=60000, seed=171, var_len=True)
    rng = np.random.default_rng(5)
    nm = reads.nm.copy()
    nm[rng.integers(0, reads.n_reads, 50)] = 300
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    refid[rng.integers(0, reads.n_reads, 500)] = -1         # unmapped records in between: walked over, not kept
    names = ["r%d%s" % (i, "x" * (i % 19)) for i in range(reads.n_reads)]        # record sizes differ read to read
    path = _bam_of(tmp_path, contigs, abi.ReadsSoA(**{**reads.as_dict(), "nm": nm}), refid=refid, names=names)
    names_h, lens_h, refid_h, host = abi.read_bam(path)
    monkeypatch.setenv("MIDAS_SNPS_TRACE", "1")
    monkeypatch.setenv("MIDAS_SNPS_DECODE_STREAM", "0")
    _, _, refid_a, arena = abi.read_bam(path, ctx, resident=True)
    assert _streamed(capfd) == 0
    monkeypatch.setenv("MIDAS_SNPS_DECODE_STREAM", "1")
    monkeypatch.setenv("MIDAS_SNPS_DECODE_GROUP_BLOCKS", str(group_blocks))
    monkeypatch.setenv("MIDAS_SNPS_DECODE_SLOTS", str(slots))
    names_s, lens_s, refid_s, res = abi.read_bam(path, ctx, resident=True)
    assert _streamed(capfd) >= 3
    assert names_s == names_h and lens_s == lens_h
    assert res.n_reads == host.n_reads == arena.n_reads and res.l_seq_total == arena.l_seq_total
    np.testing.assert_array_equal(refid_h, refid_s)
    np.testing.assert_array_equal(refid_a, refid_s)
    # the same pileup from the two resident layouts (the kernel reads the records + payload the decode wrote) ...
    kept = refid_h >= 0
    assert kept.all()
    counts = []
    for r in (arena, res):
        sub, rb = bam.group_by_contig(contigs.ids, refid_s, r, contigs.ids)
        assert sub is r
        table = abi.ContigTable(length=contigs.length, species=contigs.species, read_begin=rb, ref=contigs.ref, n_species=contigs.n_species)
        b = ctx.batch(table, r)
        assert b.info().path == abi.PATH_DIRECT
        b.run(THR)
        counts.append(b.fetch())
        # ... and from the other paths of the streamed handle's batch: their SEQ / QUAL / CIGAR columns are cut out of the direct
        # layout on first use (no inflated stream is kept)
        if r is res:
            for p in (abi.PATH_PACKED, abi.PATH_LONG):
                b.select_path(p)
                b.run(THR)
                for want, got in zip(counts[0], b.fetch()):
                    np.testing.assert_array_equal(want, got, err_msg=abi.PATH_NAMES[p])
        b.close()
    for want, got in zip(counts[0], counts[1]):
        np.testing.assert_array_equal(want, got)
    assert int(counts[0][0].sum()) > 0
    # every column, brought down (cut out of the direct layout): the host decode's
    down = ctx.fetch_payload(res)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(host, k), getattr(down, k), err_msg=k)


@pytest.mark.parametrize("order", ["long_reads_first", "short_reads_first"])
def test_streamed_decode_against_the_oracle_with_a_short_first_estimate(ctx, tmp_path, monkeypatch, capfd, order):
    """The result's arrays are sized from the FIRST group's records and payload bytes per inflated byte.  A file whose first part
    holds long reads (few records per byte) and whose rest holds short ones proves the record arrays short; the other way round
    (short reads carry more header per payload byte) the payload.  Either way the arrays are grown on the device and the
    counts are the oracle's."""
    one = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=50000, n_reads=4000, read_len=900, seed=181)
    two = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=50000, n_reads=60000, read_len=60, seed=182)
    (a_contigs, a_reads), (b_contigs, b_reads) = (one, two) if order == "long_reads_first" else (two, one)
    na, nb = a_contigs.n_contigs, b_contigs.n_contigs
    ids = ["a_%d" % i for i in range(na)] + ["b_%d" % i for i in range(nb)]
    contigs = abi.ContigTable(length=np.concatenate([a_contigs.length, b_contigs.length]),
                              species=np.array([0] * na + [1] * nb, dtype=np.int32),
                              read_begin=np.concatenate([a_contigs.read_begin, a_reads.n_reads + b_contigs.read_begin[1:]]),
                              ref=np.concatenate([a_contigs.ref, b_contigs.ref]), n_species=2, ids=ids, species_ids=["sp_a", "sp_b"])
    parts = {}
    for k in abi._SOA_DTYPES:
        x, y = getattr(a_reads, k), getattr(b_reads, k)
        if k in ("seq_off", "qual_off", "cigar_off"):
            parts[k] = np.concatenate([x, y[1:] + x[-1]])
        else:
            parts[k] = np.concatenate([x, y])
    reads = abi.ReadsSoA(**parts)
    refid = np.repeat(np.arange(na + nb, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "two.bam")
    bam.write_bam(path, ids, [int(x) for x in contigs.length], refid, reads)
    st, _, oc, oa, os_ = c_oracle.pileup(THR, contigs, reads)
    assert st == 0
    monkeypatch.setenv("MIDAS_SNPS_TRACE", "1")
    monkeypatch.setenv("MIDAS_SNPS_DECODE_GROUP_BLOCKS", "12")
    _, _, refid_s, res = abi.read_bam(path, ctx, resident=True)
    err = capfd.readouterr().err
    line = [ln for ln in err.splitlines() if ln.startswith("[device decode] streamed:") and "records in" in ln]
    assert line and "(0 times)" not in line[0], line      # the arrays were grown
    np.testing.assert_array_equal(refid_s, refid)
    b = ctx.batch(contigs, res)
    b.run(THR)
    for want, got in zip((oc, oa, os_), b.fetch()):
        np.testing.assert_array_equal(want, got)
    b.close()
    down = ctx.fetch_payload(res)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(reads, k), getattr(down, k), err_msg=k)


def test_a_damaged_block_in_a_late_group_is_named(ctx, tmp_path, monkeypatch):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=30000, n_reads=40000, seed=191)
    path = _bam_of(tmp_path, contigs, reads)
    blob = bytearray(open(path, "rb").read())
    # the block table: walk the file (BSIZE in the BC field), damage a byte in the middle of the DEFLATE stream of block 100
    starts, p = [], 0
    while p < len(blob):
        starts.append(p)
        p += int.from_bytes(blob[p + 16:p + 18], "little") + 1
    assert len(starts) > 120
    for which in (100,):
        bad = bytearray(blob)
        mid = (starts[which] + starts[which + 1]) // 2
        bad[mid] ^= 0x5A
        q = str(tmp_path / ("bad%d.bam" % which))
        open(q, "wb").write(bad)
        messages = []
        for stream in ("0", "1"):
            monkeypatch.setenv("MIDAS_SNPS_DECODE_STREAM", stream)
            monkeypatch.setenv("MIDAS_SNPS_DECODE_GROUP_BLOCKS", "16")
            with pytest.raises(abi.MidasSnpsError) as e:
                abi.read_bam(q, ctx, resident=True)
            assert e.value.status == abi.ERR_BAD_LAYOUT
            messages.append(e.value.message)
        assert messages[0] == messages[1], messages         # (the same block, named the same way)


def test_a_record_longer_than_a_groups_tail_goes_to_one_arena(ctx, tmp_path, monkeypatch, capfd):
    rng = np.random.default_rng(9)
    L = 900000

    def rs(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    reads = [dict(pos=int(p), cigar="100M", seq=rs(100), nm=0) for p in rng.integers(0, L - 200, size=20000)]
    reads.append(dict(pos=1000, cigar="700000M", seq=rs(700000), nm=0))          # ~1 MB of record: 17 blocks
    reads.sort(key=lambda r: r["pos"])
    soa = H.reads_from_dicts(reads)
    contigs = H.single_contig(L, soa.n_reads)
    path = str(tmp_path / "long.bam")
    bam.write_bam(path, ["c0"], [L], np.zeros(soa.n_reads, dtype=np.int32), soa)
    monkeypatch.setenv("MIDAS_SNPS_TRACE", "1")
    monkeypatch.setenv("MIDAS_SNPS_DECODE_GROUP_BLOCKS", "4")
    _, _, refid, res = abi.read_bam(path, ctx, resident=True)
    err = capfd.readouterr().err
    assert "streamed decode gave up" in err
    assert res.n_reads == soa.n_reads
    down = ctx.fetch_payload(res)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(soa, k), getattr(down, k), err_msg=k)


def test_a_ranks_range_is_streamed_too(ctx, tmp_path, monkeypatch, capfd):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=4, contig_len=25000, n_reads=80000, seed=201, var_len=True)
    path = _bam_of(tmp_path, contigs, reads)
    _, _, refid_h, host = abi.read_bam(path)
    monkeypatch.setenv("MIDAS_SNPS_TRACE", "1")
    monkeypatch.setenv("MIDAS_SNPS_DECODE_GROUP_BLOCKS", "9")
    for n in (2, 3):
        got_refid, got_pos, total = [], [], 0
        shares = [abi.BamSlice(path, k, n) for k in range(n)]
        for k, s in enumerate(shares):
            lo, hi = s.first, s.end
            if lo < 0 or lo >= hi:
                continue
            refid, rr = s.load_ranges([(lo, hi)], ctx, resident=True)
            assert _streamed(capfd) >= 2
            cols = ctx.fetch_payload(rr)
            got_refid.append(np.array(refid))
            got_pos.append(np.array(cols.pos))
            total += rr.n_reads
            s.close()
        assert total == host.n_reads
        np.testing.assert_array_equal(np.concatenate(got_refid), refid_h)
        np.testing.assert_array_equal(np.concatenate(got_pos), host.pos)
