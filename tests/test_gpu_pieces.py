"""Pieces of long contigs on the device (midas_snps_contigs.origin): both paths give the whole contig's tallies, counters and
table bytes when the contig is cut into pieces -- the work item that lets one long contig spread over several GPUs."""
import numpy as np
import pytest

from midas_amd import abi, pieces, synth
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

PATHS = [abi.PATH_DIRECT, abi.PATH_PACKED]


@pytest.fixture(scope="module", params=PATHS, ids=[abi.PATH_NAMES[p] for p in PATHS])
def path_ctx(request):
    ctx = abi.Context(0)
    ctx.set_default_path(request.param)
    yield ctx
    ctx.close()


THR = abi.Thresholds(mapid=94.0, mapq=20, baseq=30, readq=20, aln_cov=0.75)


@pytest.mark.parametrize("piece_len", [65536, 262144])
def test_pieces_equal_the_whole_contig(path_ctx, piece_len, tmp_path):
    table, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=1000000, n_reads=200000, seed=synth.BASE_SEED + 31)
    st, _, oc, oa, os_ = c_oracle.pileup(THR, table, reads)
    assert st == 0
    pt, pr, entries = pieces.split_table(table, reads, piece_len)
    path_ctx.set_row_coder(abi.ROWS_HOST)
    b = path_ctx.batch(pt, pr)
    b.run(THR)
    counts, allele, stats = b.fetch()
    np.testing.assert_array_equal(counts, oc)
    np.testing.assert_array_equal(allele, oa)
    np.testing.assert_array_equal(stats, os_)
    # the rows of the pieces, written from the device results, are the whole contigs' rows byte for byte
    ids = [pt.ids[k] for k in range(pt.n_contigs)]
    part = str(tmp_path / "pieces.gz")
    b.write_part(part, list(range(pt.n_contigs)), ids, header=True, gz_level=4, threads=4)
    # ... and so are the members of the device's row coder: the pieces' file equals the whole contigs' file from the same coder
    path_ctx.set_row_coder(abi.ROWS_DEVICE)
    dev_part = str(tmp_path / "pieces_dev.gz")
    b.write_part(dev_part, list(range(pt.n_contigs)), ids, header=True, gz_level=4, threads=4)
    b.close()
    bw = path_ctx.batch(table, reads)
    bw.run(THR)
    dev_whole = str(tmp_path / "whole_dev.gz")
    bw.write_part(dev_whole, list(range(table.n_contigs)), table.ids, header=True, gz_level=4, threads=4)
    bw.close()
    assert open(dev_part, "rb").read() == open(dev_whole, "rb").read()
    whole = str(tmp_path / "whole.gz")
    off = table.site_offsets()
    abi.write_table(whole, table.ids, [oa[off[k]:off[k + 1]] for k in range(table.n_contigs)],
                    [oc[off[k]:off[k + 1]] for k in range(table.n_contigs)], gz_level=4, threads=4)
    assert open(part, "rb").read() == open(whole, "rb").read()


def test_pieces_with_long_deletions_and_clips(path_ctx):
    """Reads whose reference span is far longer than their sequence (spliced-style N ops, deletions) straddling piece
    borders: the halo is the longest span, and a read can reach across more than one border."""
    rng = np.random.default_rng(77)
    L = 4 * 65536
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    rd = []
    for _ in range(3000):
        gap = int(rng.integers(1, 70000)) if rng.random() < 0.2 else int(rng.integers(1, 30))
        op = "N" if rng.random() < 0.5 else "D"
        a, c = int(rng.integers(5, 60)), int(rng.integers(5, 60))
        s = int(rng.integers(0, 8))
        cigar = ("%dS" % s if s else "") + "%dM%d%s%dM" % (a, gap, op, c)
        n = s + a + c
        pos = int(rng.integers(0, L - 10))
        seq = "".join(rng.choice(list("ACGT"), n))
        rd.append(dict(pos=pos, cigar=cigar, seq=seq, qual=[int(x) for x in rng.integers(20, 41, n)], nm=int(rng.integers(0, 3)), mapq=40))
    rd.sort(key=lambda r: r["pos"])
    reads = H.reads_from_dicts(rd)
    table = abi.ContigTable(length=[L], species=[0], read_begin=[0, len(rd)], ref=ref, n_species=1, ids=["c"], species_ids=["s"])
    thr = abi.Thresholds(mapid=0.0, mapq=0, baseq=25, readq=0, aln_cov=0.0)
    st, _, oc, oa, os_ = c_oracle.pileup(thr, table, reads)
    assert st == 0
    pt, pr, _ = pieces.split_table(table, reads, 65536)
    assert pt.n_contigs == 4
    st2, _, oc2, _, os2 = c_oracle.pileup(thr, pt, pr)
    assert st2 == 0 and np.array_equal(oc, oc2) and np.array_equal(os_, os2)
    counts, allele, stats = path_ctx.pileup(thr, pt, pr)
    np.testing.assert_array_equal(counts, oc)
    np.testing.assert_array_equal(allele, oa)
    np.testing.assert_array_equal(stats, os_)


def test_overrun_in_the_next_piece_is_reported_there(path_ctx):
    L = 2 * 65536
    ref = np.frombuffer(b"ACGT" * (L // 4), np.uint8)
    rd = H.reads_from_dicts([dict(pos=10, cigar="30M", seq="ACGTA" * 6, qual=[40] * 30, nm=0, mapq=40),
                             dict(pos=65536 - 20, cigar="60M", seq="ACGTA" * 6, qual=[40] * 30, nm=0, mapq=40)])
    table = abi.ContigTable(length=[L], species=[0], read_begin=[0, 2], ref=ref, n_species=1, ids=["c"], species_ids=["s"])
    thr = abi.Thresholds(mapid=0.0, mapq=0, baseq=0, readq=0, aln_cov=0.0)
    pt, pr, _ = pieces.split_table(table, rd, 65536)
    assert pr.n_reads == 3
    with pytest.raises(abi.MidasSnpsError) as ei:
        path_ctx.pileup(thr, pt, pr)
    assert ei.value.status == abi.ERR_READ_CIGAR_OVERRUN and ei.value.read_index == 2
