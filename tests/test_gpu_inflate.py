"""BGZF blocks inflated on the device (bgzf_inflate.hip: one lane per DEFLATE stream decodes it into tokens, one wavefront per
stream lays the bytes out), held to zlib:
stored, fixed and dynamic blocks, streams of several blocks, empty streams, the longest codes and distances, matches that
overlap their own output, every compression level, and corrupt streams (a status, never a fault).  Then the BAM decoder with
the device as its inflater against the same decoder on the host's threads."""
import os
import zlib

import numpy as np
import pytest

from midas_amd import abi, bam, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with abi.Context(0) as c:
        yield c


def _raw(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def _payloads():
    rng = np.random.default_rng(7)
    text = b"".join(b"read_%07d\t%d\tACGTTGCA%s\n" % (i, int(rng.integers(0, 1 << 20)), bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 40)))
                    for i in range(1200))
    out = {
        "empty": b"",
        "one_byte": b"x",
        "zeros_64k": bytes(65280),
        "run_of_one_symbol": b"a" * 5000,
        "period_3": b"abc" * 9000,
        "random_64k": bytes(rng.integers(0, 256, 65280, dtype=np.uint8)),
        "text": text[:65000],
        "two_symbols": bytes(rng.choice(np.frombuffer(b"ab", np.uint8), 30000)),
        "skewed": bytes(np.minimum(rng.geometric(0.08, 60000), 255).astype(np.uint8)),       # long codes for the rare bytes
        "far_matches": bytes(rng.integers(0, 256, 32768, dtype=np.uint8)) * 2,                 # distance 32768 (one shy of 64 KiB)
        "bam_like": bytes(rng.integers(0, 16, 30000, dtype=np.uint8)) + bytes(rng.integers(20, 42, 30000, dtype=np.uint8)),
    }
    # as many matches as a stream can hold (one per 3-4 bytes: tokens of three bytes from a small dictionary, in random order):
    # more than the decoder's first-pass room for them, so these streams go through its second pass
    tokens = rng.integers(0, 256, (400, 3), dtype=np.uint8)
    out["dense_matches"] = tokens.tobytes() + tokens[rng.integers(0, 400, 20000)].tobytes()
    out["far_matches"] = out["far_matches"][:65280]
    return out


def _inflate_all(ctx, streams, sizes, crc=None):
    cpos, at = [], 0
    blob = bytearray()
    for s in streams:
        cpos.append(len(blob))
        blob += s
    upos = np.concatenate([[0], np.cumsum(sizes)])[:-1] if sizes else np.zeros(0, np.int64)
    return ctx.inflate_blocks(bytes(blob), cpos, [len(s) for s in streams], upos, sizes, int(sum(sizes)), crc=crc), upos


def test_streams_of_every_kind_against_zlib(ctx):
    cases, streams, plain = [], [], []
    for name, data in _payloads().items():
        for level in (0, 1, 6, 9):
            for strat, sname in ((zlib.Z_DEFAULT_STRATEGY, "default"), (zlib.Z_FIXED, "fixed"), (zlib.Z_HUFFMAN_ONLY, "huffman"), (zlib.Z_RLE, "rle")):
                if level == 0 and sname != "default":
                    continue
                cases.append("%s/level%d/%s" % (name, level, sname))
                streams.append(_raw(data, level, strat))
                plain.append(data)
    # streams of several DEFLATE blocks: full flushes in the middle, a stored block between two dynamic ones
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [bytes(np.random.default_rng(k).integers(60, 70, 9000, dtype=np.uint8)) for k in range(4)]
    s = b"".join(c.compress(p) + c.flush(zlib.Z_FULL_FLUSH) for p in parts[:3]) + c.compress(parts[3]) + c.flush()
    cases.append("several_blocks"); streams.append(s); plain.append(b"".join(parts))
    out, upos = _inflate_all(ctx, streams, [len(p) for p in plain], crc=[zlib.crc32(p) for p in plain])     # (sizes AND sums)
    for name, p, u in zip(cases, plain, upos):
        got = bytes(out[int(u):int(u) + len(p)])
        assert got == p, "%s: first difference at %d" % (name, next((i for i in range(len(p)) if got[i] != p[i]), -1))
    assert len(cases) > 120


def test_streams_whose_table_headers_fall_at_different_steps(ctx):
    """The decoder reads table headers at the lanes' common step, by all lanes of a wavefront that want one together
    (bgzf_inflate.hip, namespace w64).  zlib's own streams of a BAM reach their headers at the same symbol; these do not: every
    stream is a different number of DEFLATE blocks of different lengths -- dynamic, fixed and stored ones, empty stored blocks
    (a sync flush) and a block that ends on the stream's last byte -- and the streams of one wavefront are of every size from a few
    bytes to a full BGZF block.  Also the counts that decide whether a rare kind of literal run reaches the tokens: runs of 511
    literals and more (an escape token) between matches."""
    rng = np.random.default_rng(11)
    streams, plain = [], []
    for k in range(200):
        parts, c = [], zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15, 8,
                                        [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED][k % 5])
        s, budget = b"", int(rng.integers(40, 65280))
        while budget > 0:
            n = int(min(budget, rng.integers(1, 9000)))
            kind = int(rng.integers(0, 4))
            if kind == 0:   # incompressible: literals only (long literal runs), or stored by zlib's choice
                piece = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            elif kind == 1:  # few symbols: short codes, many matches
                piece = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
            elif kind == 2:  # a repeat of what came before (far matches) with fresh bytes in between
                src = b"".join(parts) or b"seed"
                piece = (src[-min(len(src), 3000):] + bytes(rng.integers(30, 45, 700, dtype=np.uint8)))[:n]
            else:            # skewed bytes: long codes
                piece = bytes(np.minimum(rng.geometric(0.05, n), 255).astype(np.uint8))
            parts.append(piece)
            budget -= len(piece)
            s += c.compress(piece)
            if budget > 0:
                s += c.flush([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK][int(rng.integers(0, 3))])
        s += c.flush()
        streams.append(s)
        plain.append(b"".join(parts))
    out, upos = _inflate_all(ctx, streams, [len(p) for p in plain], crc=[zlib.crc32(p) for p in plain])
    for k, (p, u) in enumerate(zip(plain, upos)):
        got = bytes(out[int(u):int(u) + len(p)])
        assert got == p, "stream %d: first difference at %d of %d" % (k, next((i for i in range(len(p)) if got[i] != p[i]), -1), len(p))


def test_corrupt_streams_are_a_status(ctx):
    data = _payloads()["text"]
    good = _raw(data)
    rng = np.random.default_rng(3)
    bad = []
    for k in range(40):                       # flipped bits, truncations, garbage
        b = bytearray(good)
        if k % 3 == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k % 3 == 1:
            b = b[:int(rng.integers(1, len(b) - 1))]
        else:
            b = bytearray(rng.integers(0, 256, int(rng.integers(10, 3000)), dtype=np.uint8).tobytes())
        bad.append(bytes(b))
    for b in bad:
        try:                                   # what zlib makes of it: a complete stream of the expected size, or not
            d = zlib.decompressobj(-15)
            ref = d.decompress(b)
            valid = d.eof and len(ref) == len(data)
        except zlib.error:
            valid = False
        try:
            out = ctx.inflate_blocks(b, [0], [len(b)], [0], [len(data)], len(data))
            assert valid and bytes(out) == ref     # (a flipped bit inside a literal's code is still a valid stream ...)
        except abi.MidasSnpsError as e:
            assert not valid and e.status == abi.ERR_BAD_LAYOUT and e.read_index == 0
        if valid and ref != data:                  # (... which the CRC-32 of the original bytes refuses)
            with pytest.raises(abi.MidasSnpsError) as ei:
                ctx.inflate_blocks(b, [0], [len(b)], [0], [len(data)], len(data), crc=[zlib.crc32(data)])
            assert ei.value.status == abi.ERR_BAD_LAYOUT and ei.value.read_index == 0 and "CRC" in ei.value.message
    assert bytes(ctx.inflate_blocks(good, [0], [len(good)], [0], [len(data)], len(data), crc=[zlib.crc32(data)])) == data
    # the wrong size is an error too
    with pytest.raises(abi.MidasSnpsError):
        ctx.inflate_blocks(good, [0], [len(good)], [0], [len(data) - 1], len(data) - 1)
    with pytest.raises(abi.MidasSnpsError):
        ctx.inflate_blocks(good, [0], [len(good)], [0], [len(data) + 1], len(data) + 1)


def test_bam_decoded_with_the_device_inflater_equals_the_host_decode(ctx, tmp_path):
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=60000, n_reads=150000, seed=synth.BASE_SEED + 61, var_len=True)
    out, db = str(tmp_path / "s"), str(tmp_path / "db")
    synth.write_sample(out, db, contigs, reads)
    path = os.path.join(out, "snps", "temp", "genomes.bam")
    names_h, lens_h, refid_h, host = abi.read_bam(path)
    names_d, lens_d, refid_d, dev = abi.read_bam(path, ctx)
    assert names_h == names_d and lens_h == lens_d
    np.testing.assert_array_equal(refid_h, refid_d)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(host, k), getattr(dev, k), err_msg=k)
    # a rank's slice: walked on the device, the same facts; its ranges decoded there, the same records (payload left there)
    for k_slice, n_slices in ((0, 1), (0, 3), (1, 3), (2, 3), (5, 7)):
        sl = abi.BamSlice(path, k_slice, n_slices)
        sd = abi.BamSlice(path, k_slice, n_slices, ctx=ctx)
        for f in ("first", "end", "sorted", "first_ref", "last_ref", "rec_begin", "total"):
            assert getattr(sl, f) == getattr(sd, f), (k_slice, n_slices, f)
        for f in ("ref_reads", "ref_bases", "ref_first"):
            np.testing.assert_array_equal(getattr(sl, f), getattr(sd, f), err_msg=f)
        for x, y in zip(sl.marks(), sd.marks()):
            np.testing.assert_array_equal(x, y)
    sl = abi.BamSlice(path, 1, 3)
    mid = int(sl.ref_first[sl.ref_first >= 0][-1])          # (a reference's first record: an exact boundary inside the slice)
    for ranges in ([(sl.first, sl.end)], [(sl.first, mid), (mid, sl.end)], [(mid, sl.end)], []):
        a = sl.load_ranges(ranges)
        b = abi.BamSlice(path, 1, 3).load_ranges(ranges, ctx)
        assert b[1].device is not None or not ranges
        np.testing.assert_array_equal(a[0], b[0])
        down = ctx.fetch_payload(b[1])
        for k in abi._SOA_DTYPES:
            np.testing.assert_array_equal(getattr(a[1], k), getattr(down, k), err_msg=k)
    with pytest.raises(abi.MidasSnpsError) as ei:                  # a range that ends inside a record
        abi.BamSlice(path, 1, 3).load_ranges([(sl.first, sl.end - 7)], ctx)
    assert ei.value.status == abi.ERR_BAD_LAYOUT
    # the payload columns cut on the device and left there: the same bytes, and a batch takes them where they are
    names_p, lens_p, refid_p, on_dev = abi.read_bam(path, ctx, payload_on_device=True)
    assert on_dev.device is not None and on_dev.seq4.size == 0 and names_p == names_h
    np.testing.assert_array_equal(refid_h, refid_p)
    down = ctx.fetch_payload(on_dev)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(host, k), getattr(down, k), err_msg=k)
    table = abi.ContigTable(length=contigs.length, species=contigs.species, read_begin=contigs.read_begin, ref=contigs.ref,
                            n_species=contigs.n_species)
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    want = ctx.pileup(thr, table, host)
    got = ctx.pileup(thr, table, on_dev)
    for a_, b_ in zip(want, got):
        np.testing.assert_array_equal(a_, b_)
    # a truncated file
    data = open(path, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(data[:len(data) // 2])
    with pytest.raises(abi.MidasSnpsError):
        abi.read_bam(cut, ctx)


def _flip_a_stored_byte(path):
    """A BAM written with stored (level 0) DEFLATE blocks, one payload byte of its third block flipped: every block still
    inflates to the size its footer states -- only the CRC-32 knows.  Returns the block's file offset."""
    raw = bytearray(open(path, "rb").read())
    p, blocks = 0, []
    while p < len(raw):
        blocks.append(p)
        p += int.from_bytes(raw[p + 16:p + 18], "little") + 1
    at = blocks[2]
    raw[at + 18 + 5 + 1000] ^= 0x10          # (18 bytes of BGZF header, 5 of the stored block's)
    open(path, "wb").write(bytes(raw))
    return at


def test_a_damaged_bgzf_block_is_refused_by_both_decoders(ctx, tmp_path):
    """htslib (behind pysam.AlignmentFile, midas/run/snps.py:186) checks every block's CRC-32; so do the host's threads and
    the device's kernels -- the whole-file decode, the decode that leaves the payload on the device, and the slices'."""
    from midas_amd import bam
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=40000, n_reads=4000, seed=9)
    path = str(tmp_path / "stored.bam")
    refid = np.repeat(np.arange(2, dtype=np.int32), np.diff(contigs.read_begin))
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads, level=0)
    abi.read_bam(path)                                                    # (undamaged: fine)
    at = _flip_a_stored_byte(path)
    for kw in (dict(), dict(ctx=ctx), dict(ctx=ctx, payload_on_device=True)):
        with pytest.raises(abi.MidasSnpsError) as ei:
            abi.read_bam(path, **kw)
        assert ei.value.status == abi.ERR_BAD_LAYOUT and "offset %d" % at in ei.value.message, (kw, ei.value.message)


def _same_decode(ctx, path):
    names_h, lens_h, refid_h, host = abi.read_bam(path)
    names_d, lens_d, refid_d, on_dev = abi.read_bam(path, ctx, payload_on_device=True)
    assert names_h == names_d and lens_h == lens_d and on_dev.device is not None
    np.testing.assert_array_equal(refid_h, refid_d)
    down = ctx.fetch_payload(on_dev)
    for k in abi._SOA_DTYPES:
        np.testing.assert_array_equal(getattr(host, k), getattr(down, k), err_msg=k)
    return host


def test_the_device_record_walk_is_the_hosts(ctx, tmp_path):
    """midas_bam_load_device finds the records, decodes their columns and NM and cuts SEQ / QUAL / CIGAR on the device
    (bam_walk.hip); the host's walk (hostio.cpp) is the yardstick.  The spec-assembled fixture; records that span several BGZF
    blocks and several walk chunks (1 000 bp reads behind 150 bp ones); records without NM, without SEQ, with NM of every
    integer width; unmapped records (refID -1) between mapped ones; and QUAL bytes that spell well-formed records exactly where
    the walk's chunks begin, so that guesses are wrong and the stitching has to walk those chunks again."""
    from midas_amd import bam
    from tests import helpers as H
    _same_decode(ctx, os.path.join(H.GOLDEN, "spec_fixture.bam"))
    rng = np.random.default_rng(5)
    long_c, long_r = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=50000, n_reads=3000, read_len=1000, seed=3)
    short_c, short_r = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=50000, n_reads=20000, read_len=150, seed=4, var_len=True)
    d = {k: np.concatenate([getattr(short_r, k), getattr(long_r, k)]) for k in ("pos", "mapq", "flag", "nm", "l_seq", "seq4", "qual", "cigar")}
    for k in ("seq_off", "qual_off", "cigar_off"):
        a, b = getattr(short_r, k), getattr(long_r, k)
        d[k] = np.concatenate([a, b[1:] + a[-1]])
    n = d["pos"].size
    d["nm"] = d["nm"].copy()
    d["nm"][rng.integers(0, n, 200)] = -1                   # no NM tag
    d["nm"][rng.integers(0, n, 200)] = 300                  # NM:i (four bytes)
    reads = abi.ReadsSoA(**d)
    refid = rng.integers(-1, 2, n).astype(np.int32)          # unmapped records in between; not sorted (the decoder does not care)
    path = str(tmp_path / "mixed.bam")
    bam.write_bam(path, short_c.ids, [int(x) for x in short_c.length], refid, reads)
    host = _same_decode(ctx, path)
    assert host.n_reads == int((refid >= 0).sum()) and (host.nm == -1).any() and (host.l_seq == 1000).any()
    # decoys: a well-formed tiny record chain in the QUAL of every read that covers the start of a 32 KiB walk chunk
    import gzip
    c3, r3 = synth.make_dataset(n_species=1, contigs_per_species=3, contig_len=60000, n_reads=30000, read_len=320, seed=22, var_len=False)
    rid3 = np.repeat(np.arange(c3.n_contigs, dtype=np.int32), np.diff(c3.read_begin))
    plain = str(tmp_path / "plain.bam")
    bam.write_bam(plain, c3.ids, [int(x) for x in c3.length], rid3, r3)
    raw = gzip.open(plain, "rb").read()
    q = 8 + int(np.frombuffer(raw, "<i4", 1, 4)[0])
    n_ref = int(np.frombuffer(raw, "<i4", 1, q)[0]); q += 4
    for _ in range(n_ref):
        q += 4 + int(np.frombuffer(raw, "<i4", 1, q)[0]) + 4
    starts = []
    while q < len(raw):
        starts.append(q)
        q += 4 + int(np.frombuffer(raw, "<i4", 1, q)[0])
    starts = np.array(starts)
    fake = bytearray()
    for k in range(6):
        rec = bytearray(33)
        rec[0:4] = np.int32(0).tobytes(); rec[4:8] = np.int32(5 + k).tobytes()
        rec[8] = 1; rec[9] = 30
        rec[20:24] = np.int32(-1).tobytes(); rec[24:28] = np.int32(-1).tobytes()
        fake += np.int32(len(rec)).tobytes() + rec
    qual, placed = r3.qual.copy(), 0
    for u in range(32768 * (starts[0] // 32768 + 1), len(raw), 32768):
        i = int(np.searchsorted(starts, u, side="right")) - 1
        qual_at = int(starts[i]) + 4 + 32 + raw[int(starts[i]) + 12] + 4 * int(np.frombuffer(raw, "<u2", 1, int(starts[i]) + 16)[0]) + 160
        if u <= qual_at and qual_at + len(fake) <= (starts[i + 1] if i + 1 < starts.size else len(raw)):
            q0 = int(r3.qual_off[i])
            qual[q0:q0 + len(fake)] = np.frombuffer(bytes(fake), np.uint8)
            placed += 1
    assert placed >= 20
    decoy = str(tmp_path / "decoy.bam")
    bam.write_bam(decoy, c3.ids, [int(x) for x in c3.length], rid3, abi.ReadsSoA(**{**r3.as_dict(), "qual": qual}))
    got = _same_decode(ctx, decoy)
    assert got.n_reads == r3.n_reads


def _point_a_record_at_no_reference(path, n_ref):
    """A BAM of stored (level 0) blocks: the refID of the first record of its third block is set to n_ref + 3 and the block's
    CRC-32 recomputed -- every block is intact, one record names a reference the header does not have."""
    raw = bytearray(open(path, "rb").read())
    p, blocks = 0, []
    while p < len(raw):
        blocks.append(p)
        p += int.from_bytes(raw[p + 16:p + 18], "little") + 1
    # the inflated stream, to find a record start inside the third block
    sizes = [int.from_bytes(raw[b + 18 + 1:b + 18 + 3], "little") for b in blocks]      # LEN of the stored block
    stream = b"".join(bytes(raw[b + 23:b + 23 + n]) for b, n in zip(blocks, sizes))
    q = 8 + int.from_bytes(stream[4:8], "little")
    nr = int.from_bytes(stream[q:q + 4], "little"); q += 4
    for _ in range(nr):
        q += 4 + int.from_bytes(stream[q:q + 4], "little") + 4
    lo = sum(sizes[:2])
    while q < lo:
        q += 4 + int.from_bytes(stream[q:q + 4], "little")
    assert q + 8 <= lo + sizes[2]
    at = blocks[2] + 23 + (q - lo) + 4                        # the record's refID in the file
    raw[at:at + 4] = int(n_ref + 3).to_bytes(4, "little")
    data = bytes(raw[blocks[2] + 23:blocks[2] + 23 + sizes[2]])
    foot = (blocks[3] if len(blocks) > 3 else len(raw)) - 8      # (the block ends with CRC-32 and ISIZE)
    raw[foot:foot + 4] = (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little")
    open(path, "wb").write(bytes(raw))


def test_a_record_of_no_reference_is_refused_by_every_decoder(ctx, tmp_path):
    """pysam / htslib refuse a record whose refID is not in the header (midas/run/snps.py:186 would raise on fetch); the host's
    walk does (plausible_record), and so do the device's whole-file decode, its payload-on-device form and a slice walked on
    the device -- none of them indexes a per-reference table with it."""
    from midas_amd import bam
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=40000, n_reads=4000, seed=11)
    path = str(tmp_path / "noref.bam")
    refid = np.repeat(np.arange(2, dtype=np.int32), np.diff(contigs.read_begin))
    bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads, level=0)
    abi.read_bam(path, ctx)                                               # (as written: fine)
    _point_a_record_at_no_reference(path, 2)
    for kw in (dict(), dict(ctx=ctx), dict(ctx=ctx, payload_on_device=True)):
        with pytest.raises(abi.MidasSnpsError) as ei:
            abi.read_bam(path, **kw)
        assert ei.value.status == abi.ERR_BAD_LAYOUT, (kw, ei.value.message)
    for kw in (dict(), dict(ctx=ctx)):
        with pytest.raises(abi.MidasSnpsError) as ei:
            abi.BamSlice(path, 0, 1, **kw)
        assert ei.value.status == abi.ERR_BAD_LAYOUT, (kw, ei.value.message)
