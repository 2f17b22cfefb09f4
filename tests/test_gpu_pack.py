"""The device packer (pack_reads.hip) against its host mirror (pack.cpp), bit for bit, through the C-ABI.

batch_create uploads the caller's BAM-native arrays and packs them on the GPU: CIGAR -> match segments cut at tile
boundaries, floor(mean quality) by wave reduction, 4-bit SEQ -> call codes with the non-ACGT mask folded into the
quality bytes, tile order + bank-phase dealing.  The host mirror computes the same layout on the CPU (its own tests pin
it against readable expectations: tests/test_abi_library.py); here every record, payload byte, input-order index and index
key of the two must be equal -- and stay equal when the pack is re-run on the resident arrays.
"""
import random

import numpy as np
import pytest

from midas_amd import abi, synth
from tests import helpers as H
from tests import mirror
from tests.test_gpu_parity import _random_cigar

pytestmark = pytest.mark.gpu


def _same_layout(ctx, contigs, reads, repack=0):
    want = mirror.pack_reads_tiled(reads, contigs)
    b = ctx.batch(contigs, reads)
    try:
        for rnd in range(repack + 1):
            if rnd:
                b.pack()
            got = b.fetch_packed()
            for name, g, w in zip(("records", "payload", "input-order map", "index keys"), got, want):
                assert g.shape == w.shape, "%s: device %s vs host mirror %s" % (name, g.shape, w.shape)
                if not np.array_equal(g, w):
                    bad = np.nonzero(g.reshape(-1) != w.reshape(-1))[0]
                    raise AssertionError("%s differ at %d positions (round %d), first flat index %d: device %s host %s"
                                         % (name, bad.size, rnd, bad[0], g.reshape(-1)[bad[:8]].tolist(),
                                            w.reshape(-1)[bad[:8]].tolist()))
    finally:
        b.close()
    return want


@pytest.mark.parametrize("kw", [
    dict(n_species=2, contigs_per_species=3, contig_len=30011, n_reads=40000, seed=7, var_len=True, lowercase_frac=0.1),
    dict(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=120000, seed=9),                    # ~2000x hot spot
    dict(n_species=1, contigs_per_species=2, contig_len=5000, n_reads=4000, read_len=16, seed=43, var_len=False),
    dict(n_species=1, contigs_per_species=2, contig_len=5000, n_reads=4000, read_len=33, seed=44, var_len=False),
    dict(n_species=2, contigs_per_species=2, contig_len=9000, n_reads=5000, read_len=125, seed=45, var_len=False),   # 32 bases / lane
    dict(n_species=2, contigs_per_species=2, contig_len=9000, n_reads=5000, read_len=250, seed=46, var_len=False),
    dict(n_species=1, contigs_per_species=3, contig_len=20000, n_reads=3000, read_len=1000, seed=41, var_len=True),
], ids=["ragged", "hotspot", "l16", "l33", "l125", "l250", "l1000"])
def test_synthetic_datasets(hip_ctx, kw):
    contigs, reads = synth.make_dataset(**kw)
    rec, blob, orig, key = _same_layout(hip_ctx, contigs, reads, repack=1)
    assert rec.shape[0] - 1 >= reads.n_reads


@pytest.mark.parametrize("seed", [99, 2026])
def test_random_cigar_grammar(hip_ctx, seed):
    """Everything the packer must tell apart: regular CIGARs (served as segments) and irregular ones in every way it
    knows (kept as they are), reads at the tile and contig borders, odd and even first bases, NM up to and past 1023."""
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
        seq = "".join(rng.choice("ACGTACGTACGTNRY=") for _ in range(l))
        qual = [rng.choice([40, 38, 35, 31, 30, 29, 12, 2, 255]) for _ in range(l)]
        reads.append(dict(pos=pos, cigar=cigar, seq=seq, qual=qual, nm=rng.choice([0, 1, 2, 5, 9, 30, 1023, 1024, 3000, None]),
                          mapq=rng.choice([42, 42, 30, 20, 19, 3])))
    reads.sort(key=lambda r: r['pos'])
    soa = H.reads_from_dicts(reads)
    ref = "".join(rng.choice("ACGTacgtN") for _ in range(L))
    _same_layout(hip_ctx, H.single_contig(L, len(reads), ref), soa)


def test_unsorted_input_and_long_skips(hip_ctx):
    rng = np.random.default_rng(3)
    L = 40000

    def rs(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    reads = [dict(pos=4000, cigar="50M9000N50M", seq=rs(100), nm=0), dict(pos=4090, cigar="10M2D10M", seq=rs(20), nm=2),
             dict(pos=4095, cigar="1M", seq="G"), dict(pos=4096, cigar="1M", seq="T"),
             dict(pos=8100, cigar="5H20S100M3I27M10S2H", seq=rs(160), nm=3), dict(pos=8190, cigar="4M1P4M", seq=rs(8)),
             dict(pos=39990, cigar="30M", seq=rs(30)), dict(pos=39999, cigar="5S30M", seq=rs(35)),
             dict(pos=100, cigar="10M35000N10M", seq=rs(20), nm=0), dict(pos=0, cigar="20M", seq=rs(20), qual="absent")]
    for _ in range(400):
        l = int(rng.integers(30, 200))
        a = int(rng.integers(5, l - 10))
        reads.append(dict(pos=int(rng.integers(0, L - 200)), cigar="%dM%dD%dM" % (a, int(rng.integers(1, 9000)), l - a), seq=rs(l), nm=0))
    soa = H.reads_from_dicts(reads)      # NOT sorted by position
    _same_layout(hip_ctx, H.single_contig(L, soa.n_reads, ref=rs(L)), soa, repack=1)


def test_empty_batches(hip_ctx):
    contigs, _ = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=10, seed=5)
    empty = abi.ContigTable(length=contigs.length, species=contigs.species, read_begin=[0, 0, 0], ref=contigs.ref, n_species=1)
    rec, blob, orig, key = _same_layout(hip_ctx, empty, abi.ReadsSoA.empty(), repack=1)
    assert rec.shape[0] == 1 and blob.size == 0


def test_c2_full_size_and_results_survive_a_repack(hip_ctx, thr_default):
    contigs, reads = synth.make_dataset(**synth.CONFIGS['c2'])
    want = mirror.pack_reads_tiled(reads, contigs)
    b = hip_ctx.batch(contigs, reads)
    try:
        b.run(thr_default)
        c1, a1, s1 = b.fetch()
        b.pack()
        got = b.fetch_packed()
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
        b.run(thr_default)
        c2, a2, s2 = b.fetch()
        np.testing.assert_array_equal(c1, c2)
        np.testing.assert_array_equal(s1, s2)
    finally:
        b.close()


def test_malformed_reads_are_statuses_from_the_device(hip_ctx, thr_default):
    ok = H.reads_from_dicts([dict(pos=0, cigar="4M", seq="ACGT"), dict(pos=2, cigar="4M", seq="ACGT"), dict(pos=3, cigar="4M", seq="ACGT")])
    contig = H.single_contig(100, 3)
    bad = abi.ReadsSoA(**{**ok.as_dict(), 'qual_off': np.array([0, 4, 6, 10], dtype=np.int64)})   # read 1 has 2 quality bytes
    with pytest.raises(abi.MidasSnpsError) as ei:
        hip_ctx.pileup(thr_default, contig, bad)
    assert ei.value.status == abi.ERR_BAD_LAYOUT and ei.value.read_index == 1
    bad = abi.ReadsSoA(**{**ok.as_dict(), 'cigar_off': np.array([0, 1, 5, 3], dtype=np.int64)})    # offsets run past the array
    with pytest.raises(abi.MidasSnpsError) as ei:
        hip_ctx.pileup(thr_default, contig, bad)
    assert ei.value.status == abi.ERR_BAD_LAYOUT
    # a read beyond the packed layout's limits is not malformed: the batch runs (on the long path, round 5) and only the packed
    # and the direct path decline it
    long_read = H.reads_from_dicts([dict(pos=0, cigar="4M", seq="ACGT"), dict(pos=0, cigar="1025M", seq="A" * 1025)])
    table = H.single_contig(5000, 2)
    from oracle import c_oracle
    st, _, oc, oa, os_ = c_oracle.pileup(thr_default, table, long_read)
    counts, allele, stats = hip_ctx.pileup(thr_default, table, long_read)
    assert st == 0 and np.array_equal(counts, oc) and np.array_equal(stats, os_)
    b = hip_ctx.batch(table, long_read)
    assert b.info().path == abi.PATH_LONG
    with pytest.raises(abi.MidasSnpsError) as ei:
        b.select_path(abi.PATH_PACKED)
    assert ei.value.status == abi.ERR_UNSUPPORTED
    b.close()
