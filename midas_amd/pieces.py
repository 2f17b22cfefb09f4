"""Pieces of long contigs: the unit of work when one contig is too long for one rank's share.

The reference calls ``count_coverage`` once per contig (midas/run/snps.py:194-199), so a whole contig is the natural work
item -- until a single 20 Mb chromosome is most of a sample and seven of eight GPUs idle.  A contig longer than the split
length is therefore cut at multiples of ``abi.ROWS_PER_MEMBER`` rows into pieces; a piece is an entry of the contig table
with ``origin`` set (include/midas_snps.h, midas_snps_contigs.origin).  A piece needs the reads that start in it and, in
front, the reads that start before it and reach in (the halo: positions below the piece's start by at most the longest
reference span of any read of that contig); positions are stored relative to the piece.  The rows of the pieces,
concatenated, are byte for byte the rows of the whole contig, and the read counters add up because every read is counted
by exactly the piece its start lies in.
"""
import numpy as np

from . import abi

MARK_SPAN = 65536          # MIDAS_BAM_MARK_SPAN: the grid of the decoder's position -> record offset marks


def cut(length: int, piece_len: int):
    """[(lo, hi)] covering [0, length): every piece but the last is piece_len long (a multiple of MARK_SPAN)."""
    if piece_len <= 0 or length <= piece_len:
        return [(0, int(length))]
    assert piece_len % MARK_SPAN == 0 and MARK_SPAN % abi.ROWS_PER_MEMBER == 0
    return [(lo, min(lo + piece_len, int(length))) for lo in range(0, int(length), piece_len)]


def piece_length(split_len: int) -> int:
    """The split length rounded up to the marks' grid (0: never split)."""
    return 0 if split_len <= 0 else -(-int(split_len) // MARK_SPAN) * MARK_SPAN


def gather(reads: "abi.ReadsSoA", read_begin, pieces):
    """The reads of the given pieces.

    reads / read_begin: records grouped by contig, positions non-decreasing inside a contig (read_begin[k] ..
    read_begin[k + 1] are contig k's).  pieces: [(k, lo, hi, last, halo)] -- piece [lo, hi) of contig k; `last`: it is the
    contig's last piece (it also takes the reads that start behind the contig's end, as the whole contig would); halo:
    how far in front of lo a read that reaches into the piece can start.
    -> (ReadsSoA with positions relative to each piece's lo, piece_read_begin int64 [len(pieces) + 1])"""
    seg = []
    for k, lo, hi, last, halo in pieces:
        a0, a1 = int(read_begin[k]), int(read_begin[k + 1])
        p = reads.pos[a0:a1]
        a = a0 if lo == 0 else a0 + int(np.searchsorted(p, lo - halo, side='left'))
        b = a1 if last else a0 + int(np.searchsorted(p, hi, side='left'))
        seg.append((a, max(a, b), lo))
    begin = np.zeros(len(pieces) + 1, np.int64)
    np.cumsum([b - a for a, b, _ in seg], out=begin[1:])

    def col(x):
        return np.concatenate([x[a:b] for a, b, _ in seg]) if seg else x[:0]

    def payload(data, off):
        parts = [data[int(off[a]):int(off[b])] for a, b, _ in seg]
        lens = col(off[1:] - off[:-1])
        new_off = np.zeros(lens.size + 1, np.int64)
        np.cumsum(lens, out=new_off[1:])
        return (np.concatenate(parts) if parts else data[:0]), new_off

    pos = col(reads.pos).copy()
    for (a, b, lo), s in zip(seg, begin[:-1]):
        if lo:
            pos[int(s):int(s) + (b - a)] -= lo
    seq4, seq_off = payload(reads.seq4, reads.seq_off)
    qual, qual_off = payload(reads.qual, reads.qual_off)
    cigar, cigar_off = payload(reads.cigar, reads.cigar_off)
    sub = abi.ReadsSoA(pos=pos, mapq=col(reads.mapq), flag=col(reads.flag), nm=col(reads.nm), l_seq=col(reads.l_seq),
                       seq_off=seq_off, qual_off=qual_off, cigar_off=cigar_off, seq4=seq4, qual=qual, cigar=cigar)
    return sub, begin


def reference_span(reads: "abi.ReadsSoA") -> int:
    """The longest stretch of reference any of the reads covers (M, D, N, =, X lengths summed): the halo a piece needs."""
    if reads.n_reads == 0:
        return 0
    cg = np.asarray(reads.cigar)
    op = cg & 15
    ln = np.where((op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8), cg >> 4, 0).astype(np.int64)
    cs = np.zeros(ln.size + 1, np.int64)
    np.cumsum(ln, out=cs[1:])
    off = np.asarray(reads.cigar_off)
    return int((cs[off[1:]] - cs[off[:-1]]).max())


def split_table(table: "abi.ContigTable", reads: "abi.ReadsSoA", piece_len: int, halo=None):
    """A whole-contig table (reads position-sorted inside every contig) -> the same work as a table of pieces.
    -> (ContigTable with origin, ReadsSoA, [(contig index, lo, hi)] per entry)"""
    piece_len = piece_length(piece_len)
    if halo is None:
        halo = reference_span(reads)
    off = table.site_offsets()
    entries, plan, refs = [], [], []
    for k in range(table.n_contigs):
        cuts = cut(int(table.length[k]), piece_len)
        for j, (lo, hi) in enumerate(cuts):
            entries.append((k, lo, hi))
            plan.append((k, lo, hi, j + 1 == len(cuts), halo))
            refs.append(table.ref[int(off[k]) + lo:int(off[k]) + hi])
    sub, begin = gather(reads, table.read_begin, plan)
    ids = [table.ids[k] for k, _, _ in entries] if table.ids else []
    out = abi.ContigTable(length=[hi - lo for _, lo, hi in entries], species=[int(table.species[k]) for k, _, _ in entries],
                          read_begin=begin, ref=np.concatenate(refs) if refs else table.ref[:0], n_species=table.n_species,
                          ids=ids, species_ids=list(table.species_ids), origin=[lo for _, lo, _ in entries])
    return out, sub, entries
