"""CPU oracle for the MIDAS `run_midas.py snps` pileup hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``midas_amd/`` may import this module:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

PARITY UNPINNED for pysam, PINNED for the reference's own code.  The per-base
arithmetic of this path lives in a third-party dependency that is absent from
/root/reference and from this image: ``pysam >= 0.8.1`` (unpinned: reference
``setup.py:15``), i.e. ``AlignmentFile.count_coverage`` and the
``AlignedSegment`` accessors, wrapping htslib.  The reference's own tests
(``test/test_midas.py:98-102``) assert exit codes only, so nothing pins pysam's
behaviour: the [EXT] functions below *restate* the published pysam / SAM-spec
semantics at the reference's call sites and are checked against hand-derived
known-answer cases (``tests/golden/kat_cases.json``).  Everything the
reference's OWN code decides around that one pysam call IS pinned against the
reference itself: ``tests/golden/make_keep_read_vectors.py`` and
``make_emit_vectors.py`` execute the reference's ``keep_read``,
``species_pileup`` and ``snps_summary`` (text taken from /root/reference at
generation time, in the build container) and commit inputs + outputs as data;
``tests/test_keep_read_golden.py`` and ``test_emit_golden.py`` hold this oracle
-- and, on the GPU box, the product through the C-ABI -- to them.

What is restated, and from where (paths relative to /root/reference):

* ``keep_read``               <- midas/run/snps.py:141-162
* ``count_coverage``          <- call site midas/run/snps.py:194-199 (pysam
                                 ``count_coverage`` with a callable
                                 ``read_callback``: no flag filter at all)
* ``species_pileup``          <- midas/run/snps.py:164-216 (contig order, row
                                 format, per-species counters)
* ``fold_species_stats``      <- midas/run/snps.py:231-241
* ``snps_summary_text``       <- midas/run/snps.py:247-262
* pysam ``query_alignment_start/end`` (soft-clip trimming, the backward walk
  stops at cigar index 1), ``get_aligned_pairs(matches_only=True)`` and the BAM
  4-bit base code ``=ACMGRSVTWYHKDBN`` are [EXT] facts restated from the pysam
  sources / SAM v1 spec.

The code is deliberately "pysam shaped" (one Python object per read, a Python
callback per read, a Python loop per site) so that timing it gives an honest
order of magnitude for what the reference costs per site.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

# BAM CIGAR operation codes (SAM v1 spec 4.2): MIDNSHP=XB
BAM_CMATCH, BAM_CINS, BAM_CDEL, BAM_CREF_SKIP, BAM_CSOFT_CLIP, BAM_CHARD_CLIP, \
    BAM_CPAD, BAM_CEQUAL, BAM_CDIFF, BAM_CBACK = range(10)
CIGAR_CHARS = "MIDNSHP=XB"
# BAM 4-bit base codes
SEQ_NT16 = "=ACMGRSVTWYHKDBN"


class PileupError(Exception):
    """The reference would raise inside the worker at this read.

    ``kind`` names the Python exception pysam/MIDAS would have produced; the
    HIP library reports the same situations as MIDAS_SNPS_ERR_* status codes.
    """

    def __init__(self, kind: str, read_index: int, msg: str = ""):
        super().__init__("%s at read %d %s" % (kind, read_index, msg))
        self.kind = kind
        self.read_index = read_index


# Error kinds, in the numbering the C-ABI uses (include/midas_snps.h).
ERR_NO_SEQ = 1        # len(None): TypeError  (l_qseq == 0)
ERR_NO_NM = 2         # dict(aln.tags)['NM']: KeyError
ERR_ZERO_ALIGN = 3    # /float(0): ZeroDivisionError
ERR_NO_QUAL = 4       # np.mean(None): TypeError
ERR_CIGAR_OVERRUN = 5  # seq[qpos]: IndexError (kept read only)
ERR_BAD_CIGAR_OP = 6  # op code > 8 (not representable in pysam's walk)
ERR_KIND_NAMES = {
    ERR_NO_SEQ: "TypeError(no SEQ)",
    ERR_NO_NM: "KeyError('NM')",
    ERR_ZERO_ALIGN: "ZeroDivisionError(align_len==0)",
    ERR_NO_QUAL: "TypeError(no QUAL)",
    ERR_CIGAR_OVERRUN: "IndexError(CIGAR longer than SEQ)",
    ERR_BAD_CIGAR_OP: "ValueError(bad CIGAR op)",
}


@dataclass
class Aln:
    """One BAM record, as far as this path looks at it."""
    pos: int                               # 0-based leftmost reference position
    mapq: int
    flag: int
    cigar: List[Tuple[int, int]]           # [(op, length)]
    seq: Optional[str]                     # stored query incl. soft clips; None if absent
    qual: Optional[Sequence[int]]          # phred ints, len == len(seq); None if absent (0xFF)
    nm: Optional[int]                      # NM aux tag; None if absent
    index: int = -1                        # position in the input, for error reports


def query_alignment_start(aln: Aln) -> int:
    """[EXT] pysam getQueryStart: sum leading S, hopping over H, stop at anything else."""
    start = 0
    for op, ln in aln.cigar:
        if op == BAM_CHARD_CLIP:
            continue
        elif op == BAM_CSOFT_CLIP:
            start += ln
        else:
            break
    return start


def query_alignment_end(aln: Aln) -> int:
    """[EXT] pysam getQueryEnd: l_qseq minus trailing S; the backward walk covers
    cigar indices n-1 .. 1 only (index 0 is never inspected)."""
    end = len(aln.seq)
    for k in range(len(aln.cigar) - 1, 0, -1):
        op, ln = aln.cigar[k]
        if op == BAM_CHARD_CLIP:
            continue
        elif op == BAM_CSOFT_CLIP:
            end -= ln
        else:
            break
    return end


def keep_read(aln: Aln, args: dict, aln_stats: dict) -> bool:
    """midas/run/snps.py:141-162, statement for statement."""
    aln_stats['aligned_reads'] += 1
    # align and query length -- len(aln.query_alignment_sequence), aln.query_length
    if aln.seq is None or len(aln.seq) == 0:
        raise PileupError("TypeError", aln.index, "(no SEQ)")
    align_len = max(0, query_alignment_end(aln) - query_alignment_start(aln))
    query_len = len(aln.seq)
    # min pid filter
    if aln.nm is None:
        raise PileupError("KeyError", aln.index, "(no NM tag)")
    if align_len == 0:
        raise PileupError("ZeroDivisionError", aln.index, "(align_len == 0)")
    if 100 * (align_len - aln.nm) / float(align_len) < args['mapid']:
        return False
    # min read quality filter -- np.mean(aln.query_qualities): fp64 sum / n
    if aln.qual is None:
        raise PileupError("TypeError", aln.index, "(no QUAL)")
    elif sum(aln.qual) / float(len(aln.qual)) < args['readq']:
        return False
    # min map quality filter
    elif aln.mapq < args['mapq']:
        return False
    # min aln cov filter
    elif align_len / float(query_len) < args['aln_cov']:
        return False
    else:
        aln_stats['mapped_reads'] += 1
        return True


PAD_ADVANCES_QUERY = False      # set_pad_rule: what BAM_CPAD does to the query position (include/midas_snps.h, midas_snps_set_pad_rule)


def set_pad_rule(pysam_rule: bool):
    global PAD_ADVANCES_QUERY
    PAD_ADVANCES_QUERY = bool(pysam_rule)


def get_aligned_pairs_matches_only(aln: Aln) -> List[Tuple[int, int]]:
    """[EXT] pysam AlignedSegment.get_aligned_pairs(matches_only=True).

    M,=,X advance both; I,S advance the query; D,N advance the reference;
    H,P advance neither (SAM v1 spec).  qpos indexes the *stored* query, so the
    first aligned base of `3S7M` has qpos 3.
    """
    pairs = []
    qpos = 0
    rpos = aln.pos
    for op, ln in aln.cigar:
        if op in (BAM_CMATCH, BAM_CEQUAL, BAM_CDIFF):
            for i in range(ln):
                pairs.append((qpos + i, rpos + i))
            qpos += ln
            rpos += ln
        elif op in (BAM_CINS, BAM_CSOFT_CLIP) or (op == BAM_CPAD and PAD_ADVANCES_QUERY):
            qpos += ln
        elif op in (BAM_CDEL, BAM_CREF_SKIP):
            rpos += ln
        else:
            # H, P, and any code pysam's if/elif chain does not name (B = 9): no effect
            pass
    return pairs


def count_coverage(reads: Iterable[Aln], length: int, quality_threshold: int,
                   read_callback) -> List[List[int]]:
    """[EXT] pysam AlignmentFile.count_coverage(contig, 0, length, quality_threshold,
    read_callback=<callable>) over the records `fetch(contig, 0, length)` yields."""
    counts = [[0] * length for _ in range(4)]
    threshold = quality_threshold or 0
    for read in reads:
        if not read_callback(read):
            continue
        seq = read.seq
        if seq is None:
            continue
        quality = read.qual
        for qpos, refpos in get_aligned_pairs_matches_only(read):
            if 0 <= refpos < length:
                if qpos >= len(seq):
                    raise PileupError("IndexError", read.index, "(CIGAR longer than SEQ)")
                if (threshold and quality is not None and quality[qpos] >= threshold) or not threshold:
                    b = seq[qpos]
                    if b == 'A':
                        counts[0][refpos] += 1
                    elif b == 'C':
                        counts[1][refpos] += 1
                    elif b == 'G':
                        counts[2][refpos] += 1
                    elif b == 'T':
                        counts[3][refpos] += 1
    return counts


SNPS_HEADER = ['ref_id', 'ref_pos', 'ref_allele', 'depth', 'count_a', 'count_c', 'count_g', 'count_t']


@dataclass
class OContig:
    id: str
    seq: str            # already upper-cased (midas/run/snps.py:62)
    species_id: str

    @property
    def length(self) -> int:
        return len(self.seq)


def species_pileup(args: dict, species_id: str, contigs: Dict[str, OContig],
                   reads_by_contig: Dict[str, List[Aln]]):
    """midas/run/snps.py:164-216 without the file handle: returns
    (text of <species>.snps, aln_stats)."""
    aln_stats = {'genome_length': 0, 'total_depth': 0, 'covered_bases': 0,
                 'aligned_reads': 0, 'mapped_reads': 0}
    lines = ['\t'.join(SNPS_HEADER) + '\n']
    for contig_id in sorted(list(contigs.keys())):
        contig = contigs[contig_id]
        if contig.species_id != species_id:
            continue
        if contig.length == 0:
            raise PileupError("ValueError", -1, "(interval of size 0: contig %s)" % contig_id)
        counts = count_coverage(reads_by_contig.get(contig_id, []), contig.length, args['baseq'],
                                lambda r: keep_read(r, args, aln_stats))
        for i in range(0, contig.length):
            ref_pos = i + 1
            ref_allele = contig.seq[i]
            depth = sum([counts[_][i] for _ in range(4)])
            row = [contig.id, ref_pos, ref_allele, depth,
                   counts[0][i], counts[1][i], counts[2][i], counts[3][i]]
            lines.append('\t'.join([str(_) for _ in row]) + '\n')
            aln_stats['genome_length'] += 1
            aln_stats['total_depth'] += depth
            if depth > 0:
                aln_stats['covered_bases'] += 1
    return ''.join(lines), aln_stats


def fold_species_stats(stats: dict) -> dict:
    """midas/run/snps.py:231-241 -- the derived floats (ints 0 stay ints)."""
    out = dict(stats)
    out['fraction_covered'] = 0
    out['mean_coverage'] = 0
    if out['genome_length'] > 0:
        out['fraction_covered'] = out['covered_bases'] / float(out['genome_length'])
    if out['covered_bases'] > 0:
        out['mean_coverage'] = out['total_depth'] / float(out['covered_bases'])
    return out


SUMMARY_FIELDS = ['species_id', 'genome_length', 'covered_bases', 'fraction_covered',
                  'mean_coverage', 'aligned_reads', 'mapped_reads']


def snps_summary_text(species_stats: Dict[str, dict]) -> str:
    """midas/run/snps.py:247-262."""
    out = '\t'.join(SUMMARY_FIELDS) + '\n'
    for sp_id, st in species_stats.items():
        st = fold_species_stats(st)
        out += sp_id + '\t'
        out += str(st['genome_length']) + '\t'
        out += str(st['covered_bases']) + '\t'
        out += str(st['fraction_covered']) + '\t'
        out += str(st['mean_coverage']) + '\t'
        out += str(st['aligned_reads']) + '\t'
        out += str(st['mapped_reads']) + '\n'
    return out


# ---------------------------------------------------------------------------
# Bridges between the object form above and the SoA form the C-ABI takes.
# ---------------------------------------------------------------------------

def decode_seq4(buf: bytes, l_seq: int) -> str:
    """BAM 4-bit packed SEQ -> str (high nibble first)."""
    out = []
    for i in range(l_seq):
        b = buf[i >> 1]
        out.append(SEQ_NT16[(b >> 4) if (i & 1) == 0 else (b & 0xF)])
    return ''.join(out)


def encode_seq4(seq: str) -> bytes:
    codes = [SEQ_NT16.index(c) if c in SEQ_NT16 else 15 for c in seq.upper()]
    if len(codes) & 1:
        codes.append(0)
    return bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))


def alns_from_soa(soa: dict, lo: int = 0, hi: Optional[int] = None) -> List[Aln]:
    """SoA dict (numpy arrays named as in include/midas_snps.h) -> [Aln]."""
    n = len(soa['pos'])
    hi = n if hi is None else hi
    out = []
    seq4 = bytes(soa['seq4'])
    qual = bytes(soa['qual'])
    for i in range(lo, hi):
        l = int(soa['l_seq'][i])
        c0, c1 = int(soa['cigar_off'][i]), int(soa['cigar_off'][i + 1])
        cig = [(int(v) & 0xF, int(v) >> 4) for v in soa['cigar'][c0:c1]]
        s0 = int(soa['seq_off'][i])
        q0 = int(soa['qual_off'][i])
        if l == 0:
            seq = None
            q = None
        else:
            seq = decode_seq4(seq4[s0:s0 + ((l + 1) >> 1)], l)
            qb = qual[q0:q0 + l]
            q = None if qb[0] == 0xFF else list(qb)
        nm = int(soa['nm'][i])
        out.append(Aln(pos=int(soa['pos'][i]), mapq=int(soa['mapq'][i]), flag=int(soa['flag'][i]),
                       cigar=cig, seq=seq, qual=q, nm=None if nm < 0 else nm, index=i))
    return out
