"""CPU tests of the host side of the path: BAM decode, row formatter + gzip writer, summary text, contig
ordering, CLI argument surface.  None of them needs a GPU; none of the product code under test touches oracle/."""
import gzip
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from midas_amd import abi, bam, fasta, synth, utility
from midas_amd.run import snps as msnps
from oracle import pileup_oracle as po
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bam_round_trip_through_native_decoder(tmp_path):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=5000, n_reads=3000,
                                        seed=3, var_len=True)
    reads.nm[5] = -1          # a record without NM
    reads.nm[6] = 300         # NM that needs the 'i' aux type
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "t.bam")
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
    names, lens, rid, got = abi.read_bam(path)
    assert names == contigs.ids and lens == [int(x) for x in contigs.length]
    np.testing.assert_array_equal(rid, refid)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(got, k), getattr(reads, k), err_msg=k)


def _rewrite_bgzf(src, dst, edit):
    """inflate a BAM, let `edit` change the byte stream, write it back as BGZF blocks"""
    import gzip
    raw = bytearray(gzip.open(src, 'rb').read())
    raw = edit(raw)
    with open(dst, 'wb') as f:
        for lo in range(0, len(raw), 60000):
            f.write(bam._bgzf_block(bytes(raw[lo:lo + 60000])))
        f.write(bam._bgzf_block(b""))


def test_large_bam_goes_through_the_parallel_record_walk(tmp_path):
    """Above a few MB of inflated records the decoder finds record boundaries inside the stream and walks the pieces on
    all cores, stitched in order (hostio.cpp walk_records): the result must be the serial walk's, i.e. what was written --
    also when bytes that look like a run of records sit inside a read's qualities, and when the stream is cut short."""
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=50000, seed=21, var_len=True)
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    names = ["r%d%s" % (i, "x" * (i % 23)) for i in range(reads.n_reads)]       # record sizes differ read to read
    path = str(tmp_path / "big.bam")
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads, read_names=names)
    _, _, rid, got = abi.read_bam(path)
    np.testing.assert_array_equal(rid, refid)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(got, k), getattr(reads, k), err_msg=k)

    cut = str(tmp_path / "cut.bam")
    _rewrite_bgzf(path, cut, lambda raw: raw[:len(raw) - 11])
    with pytest.raises(abi.MidasSnpsError) as e:
        abi.read_bam(cut)
    assert "truncated or malformed alignment record" in e.value.message


def test_block_table_walked_in_pieces_is_the_serial_walk(tmp_path, monkeypatch):
    """A large BAM's BGZF block table is walked by several threads, each from a GUESSED block start (hostio.cpp bgzf_walk_file):
    believed only if every piece's walk ends on the next piece's guess.  Forced on for a small file here: the decode must be
    what was written -- also when the guess lands in a block whose payload holds bytes that look like a block header (a stored
    block carrying a copy of a real header: the guess starts a chain there that does not hit the next piece's start, and one
    thread walks the file again) -- and a truncated file is refused as before."""
    import struct
    import zlib
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=30000, seed=22, var_len=True)
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "p.bam")
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
    want = abi.read_bam(path)
    monkeypatch.setenv("MIDAS_SNPS_PARALLEL_WALK_MIN", "1000")
    for use in (abi.read_bam,):
        got = use(path)
        assert got[0] == want[0] and got[1] == want[1]
        np.testing.assert_array_equal(got[2], want[2])
        for k in abi._SOA_DTYPES:
            np.testing.assert_array_equal(getattr(got[3], k), getattr(want[3], k), err_msg=k)

    # the same records in STORED blocks whose bytes are full of real-looking headers: every guess is at risk
    raw = gzip.open(path, 'rb').read()
    fake = bam._bgzf_block(b"x" * 300)[:18]
    stuffed = str(tmp_path / "stuffed.bam")
    with open(stuffed, 'wb') as f:
        for lo in range(0, len(raw), 20000):
            piece = raw[lo:lo + 20000]
            body = b"\x01" + struct.pack('<HH', len(piece), len(piece) ^ 0xFFFF) + piece           # one stored DEFLATE block
            bsize = 18 + len(body) + 8
            f.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack('<H', 6) + b"BC" + struct.pack('<HH', 2, bsize - 1) + body
                    + struct.pack('<II', zlib.crc32(piece), len(piece)))
        f.write(bam._bgzf_block(b""))
    got = abi.read_bam(stuffed)
    np.testing.assert_array_equal(got[2], want[2])
    np.testing.assert_array_equal(got[3].seq4, want[3].seq4)
    # ... and with decoy headers INSIDE the stored bytes (the qualities of a read replaced by copies of a header)
    rawb = bytearray(raw)
    at = len(rawb) // 2
    for k in range(0, 4000, 18):
        rawb[at + k:at + k + 18] = fake
    decoy = str(tmp_path / "decoy.bam")
    with open(decoy, 'wb') as f:
        for lo in range(0, len(rawb), 20000):
            piece = bytes(rawb[lo:lo + 20000])
            body = b"\x01" + struct.pack('<HH', len(piece), len(piece) ^ 0xFFFF) + piece
            bsize = 18 + len(body) + 8
            f.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack('<H', 6) + b"BC" + struct.pack('<HH', 2, bsize - 1) + body
                    + struct.pack('<II', zlib.crc32(piece), len(piece)))
        f.write(bam._bgzf_block(b""))
    monkeypatch.delenv("MIDAS_SNPS_PARALLEL_WALK_MIN")
    try:
        serial = abi.read_bam(decoy)
        serial_err = None
    except abi.MidasSnpsError as e:
        serial, serial_err = None, e.message
    monkeypatch.setenv("MIDAS_SNPS_PARALLEL_WALK_MIN", "1000")
    try:
        pieces = abi.read_bam(decoy)
        pieces_err = None
    except abi.MidasSnpsError as e:
        pieces, pieces_err = None, e.message
    assert (serial is None) == (pieces is None) and serial_err == pieces_err      # (the records are damaged: both refuse, the same way -- or both read)
    if serial is not None:
        np.testing.assert_array_equal(pieces[2], serial[2])

    cut = str(tmp_path / "cutfile.bam")
    blob = open(path, 'rb').read()
    open(cut, 'wb').write(blob[:len(blob) - 40])
    with pytest.raises(abi.MidasSnpsError):
        abi.read_bam(cut)


def test_record_walk_is_not_fooled_by_bytes_that_look_like_records(tmp_path):
    """White box: walk_records cuts the inflated stream into min(4 x threads, MiB) pieces and scans forward from every cut
    for eight plausible records in a row.  Here the QUAL bytes of the reads that straddle those cuts spell eight
    well-formed tiny records, so the scans find them before the next true record; the stitching must notice that the
    true chain does not arrive there and walk those pieces again."""
    import gzip
    import os
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=3, contig_len=60000, n_reads=30000, read_len=320,
                                        seed=22, var_len=False)
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "plain.bam")
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
    raw = gzip.open(path, 'rb').read()
    # header: magic, l_text, text, n_ref, then per reference l_name, name, l_ref
    p = 8 + int(np.frombuffer(raw, '<i4', 1, 4)[0])
    n_ref = int(np.frombuffer(raw, '<i4', 1, p)[0]); p += 4
    for _ in range(n_ref):
        p += 4 + int(np.frombuffer(raw, '<i4', 1, p)[0]) + 4
    rec_begin, total = p, len(raw)
    starts = []
    while p < total:
        starts.append(p)
        p += 4 + int(np.frombuffer(raw, '<i4', 1, p)[0])
    starts = np.array(starts)
    from midas_amd import utility
    nt = min(utility.cpu_budget(), 128)            # the library's own thread budget (hostio.cpp hw_threads)
    span = total - rec_begin
    n_pieces = min(nt * 4, span >> 20)
    assert n_pieces >= 8
    per = span // n_pieces
    fake = bytearray()
    for k in range(8):
        rec = bytearray(33)
        rec[0:4] = np.int32(0).tobytes(); rec[4:8] = np.int32(5 + k).tobytes()       # refID 0, pos
        rec[8] = 1; rec[9] = 30                                                      # l_read_name 1 (just the NUL), mapq
        rec[20:24] = np.int32(-1).tobytes(); rec[24:28] = np.int32(-1).tobytes()     # next refID, next pos
        fake += np.int32(len(rec)).tobytes() + rec
    placed = 0
    qual = reads.qual.copy()
    for k in range(1, n_pieces):
        u = rec_begin + k * per
        i = int(np.searchsorted(starts, u, side='right')) - 1      # the record this cut falls into
        qual_at = int(starts[i]) + 4 + 32 + raw[int(starts[i]) + 12] + 4 * int(np.frombuffer(raw, '<u2', 1, int(starts[i]) + 16)[0]) + 160
        if u <= qual_at:                                           # the scan from u reaches this read's QUAL first
            q0 = int(reads.qual_off[i])
            qual[q0:q0 + len(fake)] = np.frombuffer(bytes(fake), np.uint8)
            placed += 1
    assert placed >= 2
    decoyed = abi.ReadsSoA(**{**reads.as_dict(), 'qual': qual})
    path2 = str(tmp_path / "decoy.bam")
    bam.write_bam(path2, contigs.ids, [int(x) for x in contigs.length], refid, decoyed)
    assert len(gzip.open(path2, 'rb').read()) == total
    _, _, rid, got = abi.read_bam(path2)
    np.testing.assert_array_equal(rid, refid)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(got, k), getattr(decoyed, k), err_msg=k)


def test_bam_decoder_rejects_garbage(tmp_path):
    p = tmp_path / "bad.bam"
    p.write_bytes(b"this is not a bam file at all")
    with pytest.raises(abi.MidasSnpsError):
        abi.read_bam(str(p))
    with pytest.raises(abi.MidasSnpsError):
        abi.read_bam(str(tmp_path / "missing.bam"))


def test_a_damaged_bgzf_block_is_refused(tmp_path):
    """htslib (behind pysam.AlignmentFile, midas/run/snps.py:186) checks every block's CRC-32.  A BAM of stored (level 0)
    DEFLATE blocks with one payload byte flipped inflates to the right sizes -- only the sums know: the whole-file decode and
    the slices' both name the block."""
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=40000, n_reads=4000, seed=9)
    path = str(tmp_path / "stored.bam")
    refid = np.repeat(np.arange(2, dtype=np.int32), np.diff(contigs.read_begin))
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads, level=0)
    abi.read_bam(path)
    raw = bytearray(open(path, "rb").read())
    p, blocks = 0, []
    while p < len(raw):
        blocks.append(p)
        p += int.from_bytes(raw[p + 16:p + 18], "little") + 1
    at = blocks[2]
    raw[at + 18 + 5 + 1000] ^= 0x10          # (18 bytes of BGZF header, 5 of the stored block's)
    open(path, "wb").write(bytes(raw))
    with pytest.raises(abi.MidasSnpsError) as ei:
        abi.read_bam(path)
    assert ei.value.status == abi.ERR_BAD_LAYOUT and "offset %d" % at in ei.value.message and "CRC" in ei.value.message
    with pytest.raises(abi.MidasSnpsError) as ei:
        s = abi.BamSlice(path, 0, 1)
        s.load_ranges([(s.rec_begin, s.total)])
    assert ei.value.status == abi.ERR_BAD_LAYOUT


def test_group_by_contig_regroups_and_drops_foreign_contigs():
    reads = H.reads_from_dicts([dict(pos=5, cigar="4M", seq="ACGT"), dict(pos=1, cigar="4M", seq="CCCC"),
                                dict(pos=2, cigar="4M", seq="GGGG"), dict(pos=9, cigar="4M", seq="TTTT")])
    refid = np.array([2, 0, 1, 2], dtype=np.int32)
    sub, rb = bam.group_by_contig(["a", "b", "c"], refid, reads, ["c", "a"])   # table order c, a; b is foreign
    assert rb.tolist() == [0, 2, 3]
    assert sub.pos.tolist() == [5, 9, 1]


def test_fasta_parser_matches_biopython_conventions():
    text = ">c1 description here\nacgt\nNNAC\n\n>c2\nGG\n>c3\tx\n"
    recs = list(fasta.parse(io.StringIO(text)))
    assert recs == [("c1", "acgtNNAC"), ("c2", "GG"), ("c3", "")]
    # text before the first header is not a record; '>' only starts a record at the start of a line; CR LF files (text
    # mode has turned them into LF by the time the parser sees them) and blanks inside sequence lines; an empty header
    text = "stray line\n>a\nAC>GT \tN\n\n>\nTT\n> \nGG\n>b x y\n"
    assert list(fasta.parse(io.StringIO(text))) == [("a", "AC>GTN"), ("", "TT"), ("", "GG"), ("b", "")]
    for t in (text, text.replace("\n", "\r\n"), ">c1 description here\nacgt\nNNAC\n\n>c2\nGG\n>c3\tx\n", "", "no header at all\n"):
        assert [(i, s.decode()) for i, s in fasta.parse_bytes(t.encode())] == list(fasta.parse(io.StringIO(t.replace("\r\n", "\n"))))
    assert list(fasta.parse(io.StringIO(""))) == [] and list(fasta.parse(io.StringIO("no header at all\n"))) == []
    import random
    rng = random.Random(4)
    for _ in range(50):       # against the line-by-line reading of the same text
        lines = []
        for _ in range(rng.randint(0, 12)):
            kind = rng.random()
            if kind < 0.3:
                lines.append(">" + "".join(rng.choice("ab \t|.") for _ in range(rng.randint(0, 6))))
            else:
                lines.append("".join(rng.choice("ACGTn >\t") for _ in range(rng.randint(0, 9))))
        text = "\n".join(lines) + ("\n" if rng.random() < 0.7 else "")
        want, rid, chunks = [], None, []
        for line in io.StringIO(text):
            if line.startswith(">"):
                if rid is not None:
                    want.append((rid, "".join(chunks)))
                h = line[1:].strip()
                rid, chunks = (h.split()[0] if h else ""), []
            elif rid is not None:
                chunks.append("".join(line.split()))
        if rid is not None:
            want.append((rid, "".join(chunks)))
        assert list(fasta.parse(io.StringIO(text))) == want, repr(text)


def _oracle_text(contigs, reads, args):
    """<species>.snps text per species, from the pysam-shaped oracle."""
    alns = po.alns_from_soa(reads.as_dict())
    off = contigs.site_offsets()
    oc, by = {}, {}
    for k, cid in enumerate(contigs.ids):
        seq = bytes(contigs.ref[off[k]:off[k + 1]]).decode().upper()
        oc[cid] = po.OContig(id=cid, seq=seq, species_id=contigs.species_ids[contigs.species[k]])
        by[cid] = alns[int(contigs.read_begin[k]):int(contigs.read_begin[k + 1])]
    return {sp: po.species_pileup(args, sp, oc, by) for sp in contigs.species_ids}


def test_row_writer_reproduces_reference_text(tmp_path):
    """Counts from the C oracle -> native formatter -> must equal, byte for byte after gunzip, the text the
    pysam-shaped oracle emits with the reference's own loop (midas/run/snps.py:201-210)."""
    from oracle import c_oracle
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=3, contig_len=3000, n_reads=900, seed=2,
                                        lowercase_frac=0.2)
    contigs.ids = ["c_10", "C_1", "a|b"]                    # sorted() order != table order
    args = dict(abi.DEFAULT_ARGS)
    thr = abi.Thresholds.from_args(args)
    st, _, counts, allele, stats = c_oracle.pileup(thr, contigs, reads)
    assert st == 0
    out = str(tmp_path / "sp.snps.gz")
    off = contigs.site_offsets()
    first = True
    for cid in sorted(contigs.ids):
        k = contigs.ids.index(cid)
        abi.write_rows(out, not first, cid, allele[off[k]:off[k + 1]], counts[off[k]:off[k + 1]], threads=3)
        first = False
    got = gzip.open(out, "rt").read()
    exp, exp_stats = _oracle_text(contigs, reads, args)[contigs.species_ids[0]]
    assert got == exp
    # the one-call table writer (what the host uses) produces the same text
    out2 = str(tmp_path / "sp2.snps.gz")
    ks = [contigs.ids.index(cid) for cid in sorted(contigs.ids)]
    abi.write_table(out2, [contigs.ids[k] for k in ks], [allele[off[k]:off[k + 1]] for k in ks],
                    [counts[off[k]:off[k + 1]] for k in ks], threads=5)
    assert gzip.open(out2, "rt").read() == exp
    abi.write_table(out2, [], [], [])            # a species without contigs: header only
    assert gzip.open(out2, "rt").read() == exp.splitlines(keepends=True)[0]
    assert [l.split("\t")[0] for l in got.splitlines()[1::3000]] == ["C_1", "a|b", "c_10"]
    assert exp_stats['total_depth'] == int(stats[0, abi.STAT_TOTAL_DEPTH])


def test_row_writer_large_counts_and_many_members(tmp_path):
    n = 200000   # 13 gzip members of 16 384 rows
    counts = np.zeros((n, 4), dtype=np.uint32)
    counts[:, 0] = np.arange(n)
    counts[7] = [4000000000, 4000000000, 4000000000, 4000000000]   # depth needs 64-bit
    allele = np.frombuffer((b"ACGTN" * (n // 5 + 1))[:n], dtype=np.uint8)
    out = str(tmp_path / "big.snps.gz")
    abi.write_rows(out, False, "contig|1", allele, counts, gz_level=1)
    lines = gzip.open(out, "rt").read().splitlines()
    assert len(lines) == n + 1 and lines[0].startswith("ref_id\tref_pos")
    assert lines[8] == "contig|1\t8\tG\t16000000000\t4000000000\t4000000000\t4000000000\t4000000000"
    assert lines[n] == "contig|1\t%d\t%s\t%d\t%d\t0\t0\t0" % (n, "ACGTN"[(n - 1) % 5], n - 1, n - 1)


def test_summary_text_and_derived_floats(tmp_path):
    """snps_summary + the derived floats of pysam_pileup (midas/run/snps.py:231-241, 247-262): repr() floats,
    untouched int 0 for an uncovered species."""
    a, b = msnps.Species("sp_a"), msnps.Species("sp_b")
    a.genome_length, a.covered_bases, a.total_depth, a.aligned_reads, a.mapped_reads = 15, 14, 33, 9, 7
    a.fraction_covered = a.covered_bases / float(a.genome_length)
    a.mean_coverage = a.total_depth / float(a.covered_bases)
    b.genome_length = 20
    b.fraction_covered = b.covered_bases / float(b.genome_length)
    args = {'outdir': str(tmp_path)}
    os.makedirs(tmp_path / "snps")
    msnps.snps_summary(args, {"sp_a": a, "sp_b": b})
    got = open(tmp_path / "snps" / "summary.txt").read()
    exp = po.snps_summary_text({
        "sp_a": dict(genome_length=15, covered_bases=14, total_depth=33, aligned_reads=9, mapped_reads=7),
        "sp_b": dict(genome_length=20, covered_bases=0, total_depth=0, aligned_reads=0, mapped_reads=0)})
    assert got == exp
    assert "0.9333333333333333" in got and got.splitlines()[2] == "sp_b\t20\t0\t0.0\t0\t0\t0"


def test_initialize_contigs_uppercases_and_keys_by_contig_id(tmp_path):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=400, n_reads=10, seed=1,
                                        lowercase_frac=0.5)
    synth.write_sample(str(tmp_path / "out"), str(tmp_path / "db"), contigs, reads, gz_fasta=True)
    args = {'outdir': str(tmp_path / "out"), 'db': str(tmp_path / "db"), 'build_db': False}
    species = msnps.initialize_species(args)
    assert sorted(species) == sorted(contigs.species_ids)
    cs = msnps.initialize_contigs(species)
    assert sorted(cs) == sorted(contigs.ids)
    off = contigs.site_offsets()
    c0 = cs[contigs.ids[0]]
    assert c0.seq == bytes(contigs.ref[off[0]:off[1]]).decode().upper() and c0.length == 400
    assert c0.species_id == contigs.species_ids[0]


def _cli(*argv, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py")] + list(argv),
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def test_cli_help_and_unknown_commands():
    assert _cli("-h").returncode == 0
    assert _cli("snps", "-h").returncode == 0
    r = _cli("bogus")
    assert r.returncode == 1 and "Unrecognized command" in r.stderr
    r = _cli("species", "x")
    assert r.returncode == 1 and "not part of this build" in r.stderr


def test_cli_argument_checks_follow_the_reference(tmp_path):
    r = _cli("snps", str(tmp_path / "o"), "--pileup", env={"MIDAS_DB": ""})
    assert r.returncode == 1 and "reference database" in r.stderr.lower()
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=400, n_reads=5, seed=1)
    synth.write_sample(str(tmp_path / "out"), str(tmp_path / "db"), contigs, reads)
    db = str(tmp_path / "db")
    r = _cli("snps", str(tmp_path / "empty"), "--pileup", "-d", db)
    assert r.returncode == 1 and "no alignments were found" in r.stderr
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--mapid", "0")
    assert r.returncode == 1 and "between 1 and 100" in r.stderr
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--aln_cov", "1.5")
    assert r.returncode == 1 and "ALN_COV" in r.stderr
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--species_id", "Nope_1")
    assert r.returncode == 1 and "not found" in r.stderr
    # dead flags of the reference are accepted
    r = _cli("snps", str(tmp_path / "out"), "--pileup", "-d", db, "--discard", "--baq", "--adjust_mq", "--baseq", "101")
    assert r.returncode == 1 and "BASEQ" in r.stderr


def test_cpu_budget_is_the_quota_or_the_hardware(monkeypatch, tmp_path):
    """utility.cpu_budget: hardware threads, or the cgroup's CPU quota when that is smaller (the native library applies
    the same rule to its thread counts)."""
    import builtins
    from midas_amd import utility
    n = utility.cpu_budget()
    assert 1 <= n <= (os.cpu_count() or 1)
    real_open = builtins.open

    def fake(limit):
        def opener(path, *a, **k):
            if path == '/sys/fs/cgroup/cpu.max':
                return io.StringIO(limit)
            return real_open(path, *a, **k)
        return opener
    monkeypatch.setattr(os, 'cpu_count', lambda: 64)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(64)), raising=False)
    monkeypatch.setattr(builtins, 'open', fake("1600000 100000\n"))
    assert utility.cpu_budget() == 16
    monkeypatch.setattr(builtins, 'open', fake("150000 100000\n"))
    assert utility.cpu_budget() == 2                      # a fraction of a CPU counts as one more thread
    monkeypatch.setattr(builtins, 'open', fake("max 100000\n"))
    assert utility.cpu_budget() == 64
    monkeypatch.setattr(builtins, 'open', fake("99900000 100000\n"))
    assert utility.cpu_budget() == 64                     # a quota above the hardware changes nothing
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')           # torchrun: eight ranks share the node
    assert utility.cpu_budget() == 8
    monkeypatch.setattr(builtins, 'open', fake("400000 100000\n"))
    assert utility.cpu_budget() == 1


def test_native_cpu_budget_follows_the_same_rule_as_the_python_one():
    """midas::cpu_budget (workers.h) and utility.cpu_budget: hardware threads, affinity mask, cgroup quota, LOCAL_WORLD_SIZE --
    the same number in this process (the native one is computed once per process)."""
    import os
    lib = abi.load_library()
    assert lib.midas_snps_cpu_budget() == utility.cpu_budget()
    assert 1 <= lib.midas_snps_cpu_budget() <= (os.cpu_count() or 1)


def test_a_throwing_region_leaves_the_worker_pool_usable(tmp_path):
    """A parallel region that fails (here: a table that cannot be read) must not leave the pool marked as taken: the next
    region runs on the pool again and gives the right answer."""
    rng = np.random.default_rng(3)
    n = 40000
    counts = rng.integers(0, 50, size=(n, 4)).astype(np.uint32)
    allele = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    good = str(tmp_path / "t.snps.gz")
    abi.write_rows(good, False, "c1", allele, counts, gz_level=4, threads=4)
    with pytest.raises(abi.MidasSnpsError):
        abi.read_snps_table(str(tmp_path / "missing.snps.gz"))
    for _ in range(3):
        c2, keys, _ = abi.read_snps_table(good)
        assert np.array_equal(c2, counts)
    assert abi.count_snps_rows(good) == n


def test_decoder_reads_a_bam_assembled_from_the_spec_tables():
    """tests/golden/spec_fixture.bam was put together byte by byte from the SAM/BAM specification by
    tests/golden/make_bam_fixture.py (struct + zlib only, nothing of midas_amd): records straddling BGZF blocks, NM in every
    integer width behind B / Z / H / A / f tags, a record without NM, one without SEQ, QUAL absent, an unmapped record, the
    EOF block.  The decoder must hand back exactly the literals the generator wrote down."""
    import json
    with open(os.path.join(H.GOLDEN, "spec_fixture.json")) as f:
        exp = json.load(f)
    names, lens, refid, reads = abi.read_bam(os.path.join(H.GOLDEN, "spec_fixture.bam"))
    assert [[n, l] for n, l in zip(names, lens)] == exp["refs"]
    recs = exp["records"]
    assert reads.n_reads == len(recs) == 9                      # the unmapped record (refID -1) is not part of any contig's fetch
    for i, r in enumerate(recs):
        assert int(refid[i]) == r["refid"] and int(reads.pos[i]) == r["pos"] and int(reads.mapq[i]) == r["mapq"]
        assert int(reads.flag[i]) == r["flag"] and int(reads.l_seq[i]) == len(r["seq"])
        nm = 2147483647 if r["nm"] == "overflow" else r["nm"]   # an NM above int32 (aux type I) is carried as INT32_MAX
        assert int(reads.nm[i]) == nm
        cg = reads.cigar[reads.cigar_off[i]:reads.cigar_off[i + 1]].tolist()
        assert cg == r["cigar"]
        q = reads.qual[reads.qual_off[i]:reads.qual_off[i + 1]].tolist()
        assert q == r["qual"]
        s4 = reads.seq4[reads.seq_off[i]:reads.seq_off[i + 1]]
        seq = "".join(H.NT16[(int(s4[j >> 1]) >> (0 if j & 1 else 4)) & 15] for j in range(len(r["seq"])))
        assert seq == r["seq"]
    # ... and the rank-local slice walk sees the same records (one slice: the whole file)
    sl = abi.BamSlice(os.path.join(H.GOLDEN, "spec_fixture.bam"), 0, 1)
    try:
        assert int(sl.ref_reads.sum()) == 9
    finally:
        sl.close()


def test_a_record_of_no_reference_is_refused_by_the_host_decoders(tmp_path):
    """pysam / htslib refuse a record whose refID is not in the header (midas/run/snps.py:186); so do the host's whole-file walk
    and its slice walk -- neither hands an index past the header's references on (the device twins: tests/test_gpu_inflate.py)."""
    import numpy as np
    from midas_amd import abi, bam, synth
    from tests.test_gpu_inflate import _point_a_record_at_no_reference
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=40000, n_reads=4000, seed=11)
    path = str(tmp_path / "noref.bam")
    refid = np.repeat(np.arange(2, dtype=np.int32), np.diff(contigs.read_begin))
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads, level=0)
    abi.read_bam(path)
    _point_a_record_at_no_reference(path, 2)
    for f in (lambda: abi.read_bam(path), lambda: abi.BamSlice(path, 0, 1)):
        with pytest.raises(abi.MidasSnpsError) as ei:
            f()
        assert ei.value.status == abi.ERR_BAD_LAYOUT
