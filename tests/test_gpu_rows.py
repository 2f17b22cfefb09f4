"""The device's row coder (rows_deflate.hip): <species>.snps.gz formatted and deflated by a kernel, one workgroup per gzip
member.  The files must inflate -- with zlib, which also checks each member's CRC-32 and ISIZE -- to exactly the text the
host's formatter writes (midas/run/snps.py:201-210), be readable by this library's own table reader (member size and row
count in the gzip extra field), and come out smaller than the counts they replace on the link."""
import gzip
import os

import numpy as np
import pytest

from midas_amd import abi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

THR = abi.Thresholds(mapid=94.0, mapq=20, baseq=30, readq=20, aln_cov=0.75)


def _both(ctx, table, reads, names, tmp_path, tag, thr=THR, picks=None):
    b = ctx.batch(table, reads)
    b.run(thr)
    counts, allele, _ = b.fetch()
    off = table.site_offsets()
    pick = list(range(table.n_contigs)) if picks is None else picks
    dev, host = str(tmp_path / (tag + "_dev.gz")), str(tmp_path / (tag + "_host.gz"))
    ctx.set_row_coder(abi.ROWS_DEVICE)
    b.write_part(dev, pick, [names[c] for c in pick], header=True, gz_level=4, threads=4)
    b.close()
    abi.write_table(host, [names[c] for c in pick], [allele[off[c]:off[c + 1]] for c in pick],
                    [counts[off[c]:off[c + 1]] for c in pick], gz_level=4, threads=4)
    a, h = gzip.open(dev, "rb").read(), gzip.open(host, "rb").read()
    if a != h:
        la, lh = a.split(b"\n"), h.split(b"\n")
        bad = next((i for i in range(min(len(la), len(lh))) if la[i] != lh[i]), min(len(la), len(lh)))
        raise AssertionError("%s: text differs at line %d of %d/%d: %r vs %r" % (tag, bad, len(la), len(lh), la[bad:bad + 2], lh[bad:bad + 2]))
    return dev, host, counts


@pytest.fixture(scope="module")
def ctx():
    with abi.Context(0) as c:
        yield c


def test_a_20x_genome(ctx, tmp_path):
    table, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=250000, n_reads=200000, seed=synth.BASE_SEED + 41)
    dev, host, counts = _both(ctx, table, reads, table.ids, tmp_path, "cov20")
    n = table.n_sites
    assert open(dev, "rb").read() != open(host, "rb").read()         # (it WAS the device's coder, not the fallback)
    # the coders choose matches differently but play in the same league; and what crossed the link is a fraction of 17 B a site
    assert os.path.getsize(dev) < 1.25 * os.path.getsize(host)
    assert os.path.getsize(dev) < 6.5 * n
    # this library's own reader (merge_midas.py snps reads these tables member-parallel through the extra field)
    c2 = abi.read_snps_table(dev, want_keys=False)[0]
    np.testing.assert_array_equal(np.asarray(c2).reshape(-1, 4), counts)


@pytest.mark.parametrize("n_sites", [1, 2, 63, 64, 65, 16383, 16384, 16385, 32768, 40001])
def test_member_edges(ctx, tmp_path, n_sites):
    rng = np.random.default_rng(n_sites)
    ref = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n_sites, p=[0.24, 0.24, 0.24, 0.24, 0.04])
    rd = []
    for _ in range(min(400, 4 * n_sites)):
        L = int(rng.integers(1, min(60, n_sites) + 1))
        pos = int(rng.integers(0, n_sites - L + 1))
        rd.append(dict(pos=pos, cigar="%dM" % L, seq="".join(rng.choice(list("ACGT"), L)), qual=[40] * L, nm=0, mapq=40))
    rd.sort(key=lambda r: r["pos"])
    reads = H.reads_from_dicts(rd)
    table = abi.ContigTable(length=[n_sites], species=[0], read_begin=[0, len(rd)], ref=ref, n_species=1, ids=["c"], species_ids=["s"])
    thr = abi.Thresholds(mapid=0.0, mapq=0, baseq=0, readq=0, aln_cov=0.0)
    _both(ctx, table, reads, ["k"], tmp_path, "edge%d" % n_sites, thr)


def test_deep_and_empty_contigs_and_long_names(ctx, tmp_path):
    """Depths of several thousand (five-digit counts, tails that match nothing), contigs no read touches (every tail is one of
    five), ids up to the coder's limit, one beyond it (that part falls back to the host's formatter), and positions that run
    from one digit count into the next."""
    table, reads = synth.make_dataset(n_species=1, contigs_per_species=4, contig_len=3000, n_reads=120000, seed=synth.BASE_SEED + 43)
    rb = table.read_begin
    # contig 2 loses its reads: move them to contig 0's range by cutting the table differently is not possible; drop them
    keep = np.concatenate([np.arange(rb[0], rb[2]), np.arange(rb[3], rb[4])])
    from tests.test_gpu_parity import _subset
    sub = _subset(reads, keep)
    table = abi.ContigTable(length=table.length, species=table.species, read_begin=[0, rb[1], rb[2], rb[2], rb[2] + rb[4] - rb[3]],
                            ref=table.ref, n_species=1, ids=table.ids, species_ids=table.species_ids)
    names = ["a", "x" * 192, "Species_00001_contig_with_a_rather_long_name|and:odd=chars", "y" * 40]
    _both(ctx, table, sub, names, tmp_path, "deep")
    _both(ctx, table, sub, ["a", "x" * 193, "b", "c"], tmp_path, "toolong")          # (host fallback: same text, and no error)


def test_pieces_number_their_rows_from_the_origin(ctx, tmp_path):
    from midas_amd import pieces
    table, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=300000, n_reads=60000, seed=synth.BASE_SEED + 47)
    pt, pr, _ = pieces.split_table(table, reads, 65536)
    ctx.set_row_coder(abi.ROWS_DEVICE)
    b = ctx.batch(pt, pr)
    b.run(THR)
    dev = str(tmp_path / "pieces_dev.gz")
    b.write_part(dev, list(range(pt.n_contigs)), pt.ids, header=True, gz_level=4, threads=4)
    ctx.set_row_coder(abi.ROWS_HOST)
    host = str(tmp_path / "pieces_host.gz")
    b.write_part(host, list(range(pt.n_contigs)), pt.ids, header=True, gz_level=4, threads=4)
    b.close()
    ctx.set_row_coder(abi.ROWS_DEVICE)
    assert gzip.open(dev, "rb").read() == gzip.open(host, "rb").read()
