"""BAM helpers.  Reading goes through the native decoder (abi.read_bam); the small pure-Python writer below
exists because no samtools/bowtie2/pysam is available in the build image: tests and the synthetic benchmark
need a way to put `snps/temp/genomes.bam` on disk in the exact format a coordinate-sorted bowtie2 BAM has."""

import struct
import zlib

import numpy as np

_EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data: bytes, level: int = 6) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord('B'), ord('C'), 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def write_bam(path, ref_names, ref_lengths, refid, reads, read_names=None, header_text=None, level=6):
    """Write a BAM from SoA records (ReadsSoA) + a refID per record.  Records are written in the given order."""
    if header_text is None:
        header_text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join(
            "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(ref_names, ref_lengths))
    ht = header_text.encode()
    out = bytearray(b"BAM\1" + struct.pack("<i", len(ht)) + ht + struct.pack("<i", len(ref_names)))
    for n, l in zip(ref_names, ref_lengths):
        nb = n.encode() + b"\0"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    seq4 = reads.seq4.tobytes()
    qual = reads.qual.tobytes()
    cigar = reads.cigar.astype("<u4").tobytes()
    chunks = [bytes(out)]
    for i in range(reads.n_reads):
        name = (read_names[i] if read_names is not None else "r%d" % i).encode() + b"\0"
        l = int(reads.l_seq[i])
        c0, c1 = int(reads.cigar_off[i]), int(reads.cigar_off[i + 1])
        n_cig = c1 - c0
        pos = int(reads.pos[i])
        nm = int(reads.nm[i])
        aux = b"" if nm < 0 else (b"NMC" + struct.pack("<B", nm) if nm < 256 else b"NMi" + struct.pack("<i", nm))
        aux += b"YTZUU\0"
        body = struct.pack("<iiBBHHHIiii", int(refid[i]), pos, len(name), int(reads.mapq[i]),
                           4680, n_cig, int(reads.flag[i]), l, -1, -1, 0)
        body += name + cigar[4 * c0:4 * c1]
        body += seq4[int(reads.seq_off[i]):int(reads.seq_off[i]) + (l + 1) // 2]
        body += qual[int(reads.qual_off[i]):int(reads.qual_off[i]) + l] + aux
        chunks.append(struct.pack("<i", len(body)) + body)
    stream = b"".join(chunks)
    with open(path, "wb") as f:
        for o in range(0, len(stream), 0xff00):
            f.write(_bgzf_block(stream[o:o + 0xff00], level))
        f.write(_EOF_BLOCK)


def read_header(path):
    """(reference names, reference lengths) of a BAM's header, from its first BGZF blocks alone (gzip members: the standard
    library reads them) -- what a rank needs before it decides how the file is dealt, without mapping or walking the file."""
    import gzip
    with gzip.open(path, 'rb') as f:
        def need(n):
            out = b""
            while len(out) < n:
                piece = f.read(n - len(out))
                if not piece:
                    raise ValueError("%s: truncated BAM header" % path)
                out += piece
            return out
        head = need(12)
        if head[:4] != b"BAM\1":
            raise ValueError("%s: missing BAM magic" % path)
        l_text, = struct.unpack("<i", head[4:8])
        rest = head[8:]                       # (the first four bytes of the text, or of n_ref when there is none)
        body = rest + need(l_text + 4 - len(rest)) if l_text + 4 > len(rest) else rest
        n_ref, = struct.unpack("<i", body[l_text:l_text + 4])
        names, lens = [], []
        for _ in range(n_ref):
            l_name, = struct.unpack("<i", need(4))
            names.append(need(l_name)[:-1].decode('latin-1'))
            lens.append(struct.unpack("<i", need(4))[0])
    return names, lens


def group_by_contig(ref_names, refid, reads, contig_ids, fetch=None):
    """Records of a decoded BAM -> (ReadsSoA in contig-table order, read_begin) for the given contig ids.
    A coordinate-sorted BAM is already grouped by refID; anything else -- records of contigs that are not wanted among
    them (species.txt edited after the alignment, a BAM from elsewhere), records out of refID order -- is stably regrouped.
    `fetch`: brings SEQ / QUAL / CIGAR down when they were left on the device (Context.fetch_payload): the regroup slices
    them, which it can only do in host memory."""
    from .abi import ReadsSoA
    index_of = {n: i for i, n in enumerate(ref_names)}
    want = np.array([index_of.get(c, -1) for c in contig_ids], dtype=np.int64)
    # the usual case costs two binary searches per contig: records already in refID order, the wanted contigs in header
    # order, and no record of any other contig among them
    n = int(refid.shape[0])
    if want.size and (want >= 0).all() and (want.size == 1 or (want[1:] > want[:-1]).all()) and \
            (n < 2 or bool((refid[1:] >= refid[:-1]).all())):
        w = want.astype(refid.dtype)      # (a wider needle type would make searchsorted convert all of refid)
        lo = np.searchsorted(refid, w, side='left')
        hi = np.searchsorted(refid, w, side='right')
        if int((hi - lo).sum()) == n:
            read_begin = np.zeros(len(contig_ids) + 1, dtype=np.int64)
            np.cumsum(hi - lo, out=read_begin[1:])
            return reads, read_begin
        # records that stayed on the device in the kernel's own layout (abi.ResidentReads): ONE run of them -- the wanted
        # contigs next to each other in the file, as the contigs of a batch are -- is taken where it lies
        if hasattr(reads, 'l_seq_total') and int((hi - lo).sum()) == int(hi[-1] - lo[0]):
            from .abi import ResidentReads
            read_begin = np.zeros(len(contig_ids) + 1, dtype=np.int64)
            np.cumsum(hi - lo, out=read_begin[1:])
            run = int(hi[-1] - lo[0])
            share = reads.l_seq_total * run // max(1, reads.n_reads)
            return ResidentReads(reads._lib, reads._h, reads.owner, run, share, first=reads.first + int(lo[0])), read_begin
    if getattr(reads, 'device', None) is not None:
        if fetch is None:
            raise ValueError("the reads' SEQ / QUAL / CIGAR are on the device: fetch them (Context.fetch_payload) before regrouping")
        reads = fetch(reads)
    rank = np.full(len(ref_names) + 1, -1, dtype=np.int64)
    for k, r in enumerate(want):
        if r >= 0:
            rank[r] = k
    key = rank[refid]
    keep = key >= 0
    order = np.nonzero(keep)[0]
    k = key[order]
    if k.size and np.any(k[1:] < k[:-1]):
        order = order[np.argsort(k, kind='stable')]
        k = key[order]
    read_begin = np.zeros(len(contig_ids) + 1, dtype=np.int64)
    np.cumsum(np.bincount(k, minlength=len(contig_ids)), out=read_begin[1:])
    if order.size == reads.n_reads and (order.size == 0 or np.all(order == np.arange(order.size))):
        return reads, read_begin

    def gather(data, off):
        lens = (off[1:] - off[:-1])[order]
        new_off = np.zeros(order.size + 1, dtype=np.int64)
        np.cumsum(lens, out=new_off[1:])
        ix = np.repeat(off[:-1][order] - new_off[:-1], lens) + np.arange(new_off[-1])
        return data[ix], new_off
    seq4, seq_off = gather(reads.seq4, reads.seq_off)
    qual, qual_off = gather(reads.qual, reads.qual_off)
    cigar, cigar_off = gather(reads.cigar, reads.cigar_off)
    sub = ReadsSoA(pos=reads.pos[order], mapq=reads.mapq[order], flag=reads.flag[order], nm=reads.nm[order],
                   l_seq=reads.l_seq[order], seq_off=seq_off, qual_off=qual_off, cigar_off=cigar_off,
                   seq4=seq4, qual=qual, cigar=cigar)
    return sub, read_begin
