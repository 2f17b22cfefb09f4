"""Seeded synthetic inputs for the pileup path (SURVEY.md 8d "Synthetic generator").

There is no aligner and no MIDAS DB in the build image, so benchmark and parity
inputs are generated directly in the form a coordinate-sorted bowtie2 BAM decodes
to: per-contig, pos-sorted records with BAM-native SEQ/QUAL/CIGAR encodings.

Distributions (thresholds bite at the CLI defaults):
  reference   i.i.d. uniform ACGT, 0.01 % of sites turned to N in short runs
  reads       start uniform over valid positions, L stored bases, forward strand
  edits       1 % substitutions; 5 % of reads carry one I or D of length 1-3;
              5 % carry a soft clip of 1-20 at one end; 0.1 % of read bases are N
  NM          mismatches + inserted + deleted bases, plus an excess on 2 % of reads
  QUAL        per base {70 %: 37-41, 20 %: 30-36, 10 %: 2-29}; 1 % of reads all 2-19
  MAPQ        {80 %: 42, 10 %: 20-41, 10 %: 0-19}
"""

from __future__ import annotations

import numpy as np

from .abi import ContigTable, ReadsSoA

BASE_SEED = 20260927
_NT16 = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
_CODE_OF_ACGT = np.array([1, 2, 4, 8], dtype=np.uint8)   # BAM 4-bit codes of A,C,G,T
_ASCII_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)


def make_reference(rng, n_contigs: int, contig_len: int, n_frac: float = 1e-4, lowercase_frac: float = 0.0):
    """-> (ref ascii uint8 [n_contigs*contig_len], ref2bit int8 with -1 at N)"""
    g = n_contigs * contig_len
    two = rng.integers(0, 4, size=g, dtype=np.int8)
    ref = _ASCII_ACGT[two]
    n_runs = int(g * n_frac / 5) if n_frac > 0 else 0
    if n_runs:
        starts = rng.integers(0, max(1, g - 10), size=n_runs)
        lens = rng.integers(1, 10, size=n_runs)
        for s, ln in zip(starts, lens):
            ref[s:s + ln] = ord('N')
            two[s:s + ln] = -1
    if lowercase_frac > 0:
        lc = rng.random(g) < lowercase_frac
        ref = np.where(lc, ref | 0x20, ref).astype(np.uint8)
    return ref, two


def make_dataset(n_species: int = 1, contigs_per_species: int = 4, contig_len: int = 20000,
                 n_reads: int = 2000, read_len: int = 150, seed: int = BASE_SEED,
                 var_len: bool = False, chunk: int = 1 << 17, lowercase_frac: float = 0.0,
                 plain_only: bool = False):
    """-> (ContigTable, ReadsSoA).  Contig ids are '<species>_c<k>' (k unpadded, so Python's
    sorted() order differs from table order, as the reference's emit loop must be fed)."""
    rng = np.random.default_rng(seed)
    n_contigs = n_species * contigs_per_species
    ref, two = make_reference(rng, n_contigs, contig_len, lowercase_frac=lowercase_frac)
    species_ids = ["Species_%05d" % (s + 1) for s in range(n_species)]
    ids, species = [], []
    for s in range(n_species):
        for k in range(contigs_per_species):
            ids.append("%s_c%d" % (species_ids[s], k + 1))
            species.append(s)
    length = np.full(n_contigs, contig_len, dtype=np.int64)
    L = read_len
    if contig_len < L + 8:
        raise ValueError("contig_len must exceed read_len + 8")

    # reads per contig, then sorted starts
    per = rng.multinomial(n_reads, np.full(n_contigs, 1.0 / n_contigs))
    read_begin = np.zeros(n_contigs + 1, dtype=np.int64)
    np.cumsum(per, out=read_begin[1:])
    contig_of = np.repeat(np.arange(n_contigs, dtype=np.int64), per)

    # per-read shape
    u = rng.random(n_reads)
    kind = np.zeros(n_reads, dtype=np.int8)            # 0 plain, 1 ins, 2 del, 3 lead clip, 4 trail clip
    kind[u < 0.025] = 1
    kind[(u >= 0.025) & (u < 0.05)] = 2
    kind[(u >= 0.05) & (u < 0.075)] = 3
    kind[(u >= 0.075) & (u < 0.10)] = 4
    if plain_only:          # every read a single full-length match (kernel fast path only; developer measurements)
        kind[:] = 0
    lens = np.full(n_reads, L, dtype=np.int32)
    if var_len:
        trim = rng.random(n_reads) < 0.3
        lens[trim] = rng.integers(max(20, L // 3), L + 1, size=int(trim.sum()))
    ev_len = np.zeros(n_reads, dtype=np.int32)
    ev_len[(kind == 1) | (kind == 2)] = rng.integers(1, 4, size=int(((kind == 1) | (kind == 2)).sum()))
    ev_len[kind >= 3] = rng.integers(1, 21, size=int((kind >= 3).sum()))
    ev_len = np.minimum(ev_len, lens // 4)
    kind[ev_len == 0] = 0
    # indel offset a: bases before the event
    a = (rng.random(n_reads) * (lens - ev_len - 20)).astype(np.int32) + 10
    a = np.clip(a, 1, np.maximum(lens - ev_len - 1, 1))
    ref_span = lens.astype(np.int64).copy()
    ref_span[kind == 1] -= ev_len[kind == 1]
    ref_span[kind == 2] += ev_len[kind == 2]
    ref_span[kind >= 3] -= ev_len[kind >= 3]
    pos = (rng.random(n_reads) * (contig_len - ref_span)).astype(np.int64)
    # sort by (contig, pos); keep everything else aligned to the same permutation
    order = np.lexsort((pos, contig_of))
    pos, kind, lens, ev_len, a, ref_span = pos[order], kind[order], lens[order], ev_len[order], a[order], ref_span[order]

    # CIGARs
    n_cig = np.ones(n_reads, dtype=np.int64)
    n_cig[(kind == 1) | (kind == 2)] = 3
    n_cig[kind >= 3] = 2
    cigar_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(n_cig, out=cigar_off[1:])
    cigar = np.zeros(int(cigar_off[-1]), dtype=np.uint32)
    c0 = cigar_off[:-1]
    m = kind == 0
    cigar[c0[m]] = (lens[m].astype(np.uint32) << 4) | OP_M
    for kd, op in ((1, OP_I), (2, OP_D)):
        m = kind == kd
        rest = lens[m] - a[m] - (ev_len[m] if kd == 1 else 0)
        cigar[c0[m]] = (a[m].astype(np.uint32) << 4) | OP_M
        cigar[c0[m] + 1] = (ev_len[m].astype(np.uint32) << 4) | op
        cigar[c0[m] + 2] = (rest.astype(np.uint32) << 4) | OP_M
    m = kind == 3
    cigar[c0[m]] = (ev_len[m].astype(np.uint32) << 4) | OP_S
    cigar[c0[m] + 1] = ((lens[m] - ev_len[m]).astype(np.uint32) << 4) | OP_M
    m = kind == 4
    cigar[c0[m]] = ((lens[m] - ev_len[m]).astype(np.uint32) << 4) | OP_M
    cigar[c0[m] + 1] = (ev_len[m].astype(np.uint32) << 4) | OP_S

    seq_bytes = (lens.astype(np.int64) + 1) >> 1
    seq_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(seq_bytes, out=seq_off[1:])
    qual_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(lens.astype(np.int64), out=qual_off[1:])
    seq4 = np.zeros(int(seq_off[-1]), dtype=np.uint8)
    qual = np.zeros(int(qual_off[-1]), dtype=np.uint8)
    nm = np.zeros(n_reads, dtype=np.int32)

    contig_site0 = contig_of[order] * contig_len
    j = np.arange(L, dtype=np.int64)[None, :]
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        k_, l_, e_, a_, p_ = kind[lo:hi, None], lens[lo:hi, None].astype(np.int64), \
            ev_len[lo:hi, None].astype(np.int64), a[lo:hi, None].astype(np.int64), pos[lo:hi, None]
        # reference offset (relative to pos) of query base j, or -1 when the base is not aligned
        rel = np.broadcast_to(j, (hi - lo, L)).copy()
        ins = k_ == 1
        rel = np.where(ins & (j >= a_ + e_), j - e_, rel)
        rel = np.where(ins & (j >= a_) & (j < a_ + e_), -1, rel)
        rel = np.where((k_ == 2) & (j >= a_), j + e_, rel)
        rel = np.where(k_ == 3, np.where(j < e_, -1, j - e_), rel)
        rel = np.where((k_ == 4) & (j >= l_ - e_), -1, rel)
        inread = j < l_
        aligned = (rel >= 0) & inread
        site = contig_site0[lo:hi, None] + p_ + np.where(aligned, rel, 0)
        rb = two[site].astype(np.int16)                      # -1 at reference N
        rnd = rng.integers(0, 4, size=rb.shape, dtype=np.int16)
        sub = rng.random(rb.shape) < 0.01
        shift = rng.integers(1, 4, size=rb.shape, dtype=np.int16)
        base = np.where(aligned & (rb >= 0), np.where(sub, (rb + shift) & 3, rb), rnd)
        mism = aligned & ((rb < 0) | sub)
        code = _CODE_OF_ACGT[base]
        isn = rng.random(rb.shape) < 0.001
        code = np.where(isn, 15, code).astype(np.uint8)
        mism |= aligned & isn
        code = np.where(inread, code, 0).astype(np.uint8)
        nm_c = mism.sum(axis=1).astype(np.int32)
        kk = kind[lo:hi]
        nm_c += np.where((kk == 1) | (kk == 2), ev_len[lo:hi], 0)
        nm[lo:hi] = nm_c
        # qualities
        cat = rng.random(rb.shape)
        q = np.where(cat < 0.7, rng.integers(37, 42, size=rb.shape),
                     np.where(cat < 0.9, rng.integers(30, 37, size=rb.shape), rng.integers(2, 30, size=rb.shape)))
        bad = rng.random(hi - lo) < 0.01
        q = np.where(bad[:, None], rng.integers(2, 20, size=rb.shape), q).astype(np.uint8)
        # scatter into the ragged arrays
        flat_keep = inread.ravel()
        qual[qual_off[lo]:qual_off[hi]] = q.ravel()[flat_keep]
        Lp = L + (L & 1)
        if Lp != L:
            code = np.concatenate([code, np.zeros((hi - lo, 1), dtype=np.uint8)], axis=1)
        packed = (code[:, 0::2] << 4) | code[:, 1::2]
        nb = np.arange(Lp // 2)[None, :] < seq_bytes[lo:hi, None]
        seq4[seq_off[lo]:seq_off[hi]] = packed.ravel()[nb.ravel()]
    excess = rng.random(n_reads) < 0.02
    nm[excess] += rng.integers(8, 20, size=int(excess.sum())).astype(np.int32)

    mq_cat = rng.random(n_reads)
    mapq = np.where(mq_cat < 0.8, 42, np.where(mq_cat < 0.9, rng.integers(20, 42, size=n_reads),
                                               rng.integers(0, 20, size=n_reads))).astype(np.uint8)
    flag = np.where(rng.random(n_reads) < 0.5, 16, 0).astype(np.uint16)

    reads = ReadsSoA(pos=pos.astype(np.int32), mapq=mapq, flag=flag, nm=nm, l_seq=lens,
                     seq_off=seq_off, qual_off=qual_off, cigar_off=cigar_off, seq4=seq4, qual=qual, cigar=cigar)
    contigs = ContigTable(length=length, species=np.array(species, dtype=np.int32), read_begin=read_begin,
                          ref=ref, n_species=n_species, ids=ids, species_ids=species_ids)
    return contigs, reads


# The workloads BASELINE.json / SURVEY.md 8d name.  C2 is the one the headline metric is quoted on.
CONFIGS = {
    'tiny': dict(n_species=1, contigs_per_species=4, contig_len=20000, n_reads=2000, seed=BASE_SEED + 1),
    'c2': dict(n_species=1, contigs_per_species=60, contig_len=250000, n_reads=1000000, seed=BASE_SEED + 2),
    'c3': dict(n_species=20, contigs_per_species=16, contig_len=250000, n_reads=10666667, seed=BASE_SEED + 3),
    # one rank's share of C4 (100 species x 4 Mb, 80 M reads over 8 GPUs)
    'c4_rank': dict(n_species=13, contigs_per_species=16, contig_len=250000, n_reads=10400000, seed=BASE_SEED + 4),
    # developer measurements: configs[2] with every read a full-length match (no clips, no indels)
    'c3_plain': dict(n_species=20, contigs_per_species=16, contig_len=250000, n_reads=10666667, seed=BASE_SEED + 3, plain_only=True),
}


# ---- BASELINE.json configs[3]: 100 species x 16 contigs x 250 kb = 400 Mb, 80 M aligned 150 bp reads (30x on average), dealt
# to the ranks CONTIG BY CONTIG with the product's own partitioner (dist.shard_items, the weights of run/snps.py).
C4 = dict(n_species=100, contigs_per_species=16, contig_len=250000, total_reads=80_000_000, read_len=150, seed=BASE_SEED + 5)


def c4_items(n_species=None, contigs_per_species=None, contig_len=None, total_reads=None, read_len=None, seed=None, sigma=0.5):
    """The contigs of the configs[3] sample: [(species, contig_in_species, n_reads)], abundances log-normal over the species
    (a metagenome is not flat), reads spread evenly over a species' contigs.  Deterministic in `seed`."""
    a = dict(C4)
    for k, v in dict(n_species=n_species, contigs_per_species=contigs_per_species, contig_len=contig_len, total_reads=total_reads,
                     read_len=read_len, seed=seed).items():
        if v is not None:
            a[k] = v
    rng = np.random.default_rng(a['seed'])
    ab = rng.lognormal(0.0, sigma, a['n_species'])
    per_contig = np.maximum(1, np.floor(ab / ab.sum() * a['total_reads'] / a['contigs_per_species'])).astype(np.int64)
    items = [(s, k, int(per_contig[s])) for s in range(a['n_species']) for k in range(a['contigs_per_species'])]
    return items, a


def c4_weights(items, a):
    """The LPT weights midas_amd/run/snps.py uses: 1.6 B per aligned base + 17 B per site (SURVEY 8d's figures)."""
    return {(s, k): 1.6 * a['read_len'] * n + 17.0 * a['contig_len'] for s, k, n in items}


def _take_reads(reads: ReadsSoA, sel: np.ndarray) -> dict:
    """Rows `sel` (increasing) of a ReadsSoA as a dict of arrays with rebased offsets."""
    def gather(data, off):
        lens = (off[1:] - off[:-1])[sel]
        new_off = np.zeros(sel.size + 1, dtype=np.int64)
        np.cumsum(lens, out=new_off[1:])
        ix = np.repeat(off[:-1][sel] - new_off[:-1], lens) + np.arange(int(new_off[-1]), dtype=np.int64)
        return data[ix], new_off
    d = {k: getattr(reads, k)[sel] for k in ('pos', 'mapq', 'flag', 'nm', 'l_seq')}
    n, l0 = reads.n_reads, int(reads.l_seq[0]) if reads.n_reads else 0
    if n and reads.qual.size == n * l0 and reads.seq4.size == n * ((l0 + 1) // 2):      # reads of one length: whole rows
        d['seq4'] = reads.seq4.reshape(n, (l0 + 1) // 2)[sel].reshape(-1)
        d['qual'] = reads.qual.reshape(n, l0)[sel].reshape(-1)
        d['seq_off'] = np.arange(sel.size + 1, dtype=np.int64) * ((l0 + 1) // 2)
        d['qual_off'] = np.arange(sel.size + 1, dtype=np.int64) * l0
    else:
        d['seq4'], d['seq_off'] = gather(reads.seq4, reads.seq_off)
        d['qual'], d['qual_off'] = gather(reads.qual, reads.qual_off)
    d['cigar'], d['cigar_off'] = gather(reads.cigar, reads.cigar_off)
    return d


def c4_share(rank: int, world: int, **kw):
    """-> (ContigTable, ReadsSoA, facts) of the contigs rank `rank` of `world` owns.  Every contig table carries all the
    species (n_species rows of counters on every rank, as the summary all-gather wants them).  A contig draws its reads -- its
    own random subset, in position order -- from a pool of reads
    generated against it: sixteen times cheaper to generate than sixteen contigs, and the partitioner, the pileup, the
    per-species counters and the concatenation of the parts do not care."""
    from . import dist
    items, a = c4_items(**kw)
    w = c4_weights(items, a)
    owner = dist.shard_items(w, world)
    mine = [(s, k, n) for s, k, n in items if owner[(s, k)] == rank]
    loads = [sum(w[(s, k)] for s, k, n in items if owner[(s, k)] == r) for r in range(world)]
    lengths, species, ids, refs, parts, rb = [], [], [], [], [], [0]
    n_pools = 8
    pool_reads = int(max(n for _, _, n in items) * 1.25) + 16
    for s, k, n in mine:
        key = (a['seed'], a['contig_len'], a['read_len'], pool_reads, s % n_pools)
        if key not in _C4_POOLS:      # (the partitioner deals a species' contigs to all the ranks: every rank needs every pool;
            _C4_POOLS[key] = make_dataset(n_species=1, contigs_per_species=1, contig_len=a['contig_len'], n_reads=pool_reads,   # kept for
                                          read_len=a['read_len'], seed=a['seed'] + 7919 * (s % n_pools + 1))              # the process)
        pc, pr = _C4_POOLS[key]
        sel = np.sort(np.random.default_rng(a['seed'] + 104729 * (s + 1) + k).choice(pr.n_reads, size=n, replace=False))
        parts.append(_take_reads(pr, sel))
        lengths.append(a['contig_len']); species.append(s); ids.append("Species_%05d_c%d" % (s + 1, k + 1)); refs.append(pc.ref)
        rb.append(rb[-1] + n)
    def cat(key, off_key=None):
        if not parts:
            return np.zeros(0, dtype=_DT[key])
        return np.concatenate([p[key] for p in parts])
    def cat_off(key):
        out = [np.zeros(1, dtype=np.int64)]
        base = 0
        for p in parts:
            out.append(p[key][1:] + base)
            base += int(p[key][-1])
        return np.concatenate(out)
    reads = ReadsSoA(pos=cat('pos'), mapq=cat('mapq'), flag=cat('flag'), nm=cat('nm'), l_seq=cat('l_seq'),
                     seq_off=cat_off('seq_off'), qual_off=cat_off('qual_off'), cigar_off=cat_off('cigar_off'),
                     seq4=cat('seq4'), qual=cat('qual'), cigar=cat('cigar'))
    contigs = ContigTable(length=np.array(lengths, dtype=np.int64), species=np.array(species, dtype=np.int32),
                          read_begin=np.array(rb, dtype=np.int64), ref=np.concatenate(refs) if refs else np.zeros(0, np.uint8),
                          n_species=a['n_species'], ids=ids, species_ids=["Species_%05d" % (s + 1) for s in range(a['n_species'])])
    facts = dict(n_items=len(items), my_items=len(mine), loads=loads, imbalance=max(loads) / (sum(loads) / world),
                 total_sites=a['n_species'] * a['contigs_per_species'] * a['contig_len'], total_reads=sum(n for _, _, n in items))
    return contigs, reads, facts


_C4_POOLS = {}
_DT = dict(pos=np.int32, mapq=np.uint8, flag=np.uint16, nm=np.int32, l_seq=np.int32, seq4=np.uint8, qual=np.uint8, cigar=np.uint32)


def write_sample(outdir, db_dir, contigs, reads, line_width=60, gz_fasta=False):
    """Lay a synthetic dataset out on disk the way `run_midas.py snps --build_db --align` would leave it:
    a minimal MIDAS DB (rep_genomes/<sp>/genome.fna + the files utility.check_database wants) and
    <outdir>/snps/{species.txt,temp/genomes.fa,temp/genomes.bam}.  BAM @SQ order = contig-table order."""
    import gzip
    import os
    from . import bam
    os.makedirs(db_dir, exist_ok=True)
    for d in ("marker_genes", "pan_genomes", "rep_genomes"):
        os.makedirs(os.path.join(db_dir, d), exist_ok=True)
    write_db_tables(db_dir, contigs.species_ids)
    off = contigs.site_offsets()
    for si, sp in enumerate(contigs.species_ids):
        d = os.path.join(db_dir, "rep_genomes", sp)
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "genome.fna" + (".gz" if gz_fasta else ""))
        opener = (lambda p: gzip.open(p, "wt")) if gz_fasta else (lambda p: open(p, "w"))
        with opener(path) as h:
            for k, cid in enumerate(contigs.ids):
                if contigs.species[k] != si:
                    continue
                seq = bytes(contigs.ref[off[k]:off[k + 1]]).decode()
                h.write(">%s synthetic contig %d\n" % (cid, k))
                for o in range(0, len(seq), line_width):
                    h.write(seq[o:o + line_width] + "\n")
    os.makedirs(os.path.join(outdir, "snps", "temp"), exist_ok=True)
    os.makedirs(os.path.join(outdir, "snps", "output"), exist_ok=True)
    with open(os.path.join(outdir, "snps", "species.txt"), "w") as h:
        h.write("".join(s + "\n" for s in contigs.species_ids))
    with open(os.path.join(outdir, "snps", "temp", "genomes.fa"), "w") as h:
        for k, cid in enumerate(contigs.ids):
            h.write(">%s\n%s\n" % (cid, bytes(contigs.ref[off[k]:off[k + 1]]).decode().upper()))
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    from . import abi as _abi       # (the native writer: the bytes of bam.write_bam, by all cores -- a configs[3] BAM is 9 GB)
    _abi.write_bam(os.path.join(outdir, "snps", "temp", "genomes.bam"), contigs.ids,
                   [int(x) for x in contigs.length], refid, reads)


def write_db_tables(db_dir, species_ids):
    """species_info.txt (species_id, rep_genome) and genome_info.txt (genome_id, ...) the way merge_midas.py reads
    them (midas/merge/merge.py:88-102); the representative genome of species X is named X.rep."""
    import os
    os.makedirs(db_dir, exist_ok=True)
    with open(os.path.join(db_dir, "species_info.txt"), "w") as h:
        h.write("species_id\trep_genome\tcount_genomes\n")
        h.write("".join("%s\t%s.rep\t1\n" % (s, s) for s in species_ids))
    with open(os.path.join(db_dir, "genome_info.txt"), "w") as h:
        h.write("genome_id\tspecies_id\trep_genome\n")
        h.write("".join("%s.rep\t%s\t1\n" % (s, s) for s in species_ids))


def make_genes(rng, contig_ids, contig_lengths, mean_gene=900, mean_gap=150):
    """Non-overlapping genes over each contig, ~85% coding, both strands, lengths mostly multiples of 3 (a few are
    not, and a few are non-CDS, to hit the reference's annotate() branches).  -> list of feature rows."""
    genes = []
    for cid, length in zip(contig_ids, contig_lengths):
        pos = 1 + int(rng.integers(0, mean_gap))
        k = 0
        while True:
            glen = 3 * int(rng.integers(mean_gene // 6, mean_gene // 2))
            if rng.random() < 0.05:
                glen += 1
            if pos + glen - 1 > length:
                break
            k += 1
            kind = "CDS" if rng.random() < 0.9 else ("tRNA" if rng.random() < 0.5 else "rRNA")
            genes.append(dict(gene_id="%s_g%d" % (cid, k), scaffold_id=cid, start=pos, end=pos + glen - 1,
                              strand="+" if rng.random() < 0.5 else "-", gene_type=kind))
            pos += glen + int(rng.integers(0, 2 * mean_gap))
    return genes


def write_features(db_dir, species_id, genes):
    import os
    d = os.path.join(db_dir, "rep_genomes", species_id)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "genome.features"), "w") as h:
        h.write("gene_id\tscaffold_id\tstart\tend\tstrand\tgene_type\n")
        for g in genes:
            h.write("%s\t%s\t%d\t%d\t%s\t%s\n" % (g["gene_id"], g["scaffold_id"], g["start"], g["end"], g["strand"],
                                                  g["gene_type"]))


def make_merge_dataset(root, n_samples=4, n_sites=5000, n_contigs=3, species_id="sp1", seed=0, mean_depth=12.0,
                       snp_rate=0.03, zero_depth_sample=False):
    """Inputs of `merge_midas.py snps` without running the pileup: a DB (genome.fna, genome.features, info tables)
    and n_samples run_midas-style sample dirs holding snps/output/<species>.snps.gz + snps/summary.txt.

    Sites are a mix of uncovered, monomorphic, bi/tri/quad-allelic, tied and very deep ones, so every branch of the
    reference's call_alleles / compute_prevalence / flag is exercised.  Returns a dict describing what was written.
    """
    import gzip
    import os
    rng = np.random.default_rng(seed)
    db = os.path.join(root, "db")
    for d in ("marker_genes", "pan_genomes", "rep_genomes"):
        os.makedirs(os.path.join(db, d), exist_ok=True)
    write_db_tables(db, [species_id])
    lens = [n_sites // n_contigs + (1 if k < n_sites % n_contigs else 0) for k in range(n_contigs)]
    ids = sorted("%s_contig_%d" % (species_id, k + 1) for k in range(n_contigs))
    seqs = ["".join(rng.choice(list("ACGT"), size=l)) for l in lens]
    gdir = os.path.join(db, "rep_genomes", species_id)
    os.makedirs(gdir, exist_ok=True)
    with open(os.path.join(gdir, "genome.fna"), "w") as h:
        for cid, seq in zip(ids, seqs):
            h.write(">%s\n" % cid)
            for o in range(0, len(seq), 70):
                h.write(seq[o:o + 70] + "\n")
    write_features(db, species_id, make_genes(rng, ids, lens, mean_gene=240, mean_gap=60))
    ref = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    ref_idx = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), ref)
    n = len(ref)
    kind = rng.random(n)
    alt = (ref_idx + rng.integers(1, 4, n)) % 4
    alt2 = (alt + rng.integers(1, 4, n)) % 4
    samples, counts_all = [], []
    for s in range(n_samples):
        depth = rng.poisson(mean_depth, n).astype(np.int64)
        depth[rng.random(n) < 0.04] = 0
        deep = rng.random(n) < 0.01
        depth[deep] *= 5
        c = np.zeros((n, 4), np.int64)
        af = np.where(kind < snp_rate, rng.random(n), 0.0)                 # polymorphic across samples
        na = rng.binomial(depth, af)
        np.add.at(c, (np.arange(n), ref_idx), depth - na)
        np.add.at(c, (np.arange(n), alt), na)
        third = (kind < snp_rate / 3) & (depth > 0)
        n3 = np.where(third, rng.integers(0, 3, n), 0)
        np.add.at(c, (np.arange(n), alt2), n3)
        fourth = (kind < snp_rate / 6) & (depth > 0)                       # a fourth allele on a few sites
        alt3 = 6 - ref_idx - alt - alt2
        ok4 = fourth & (alt3 >= 0) & (alt3 < 4) & (alt3 != ref_idx) & (alt3 != alt) & (alt3 != alt2)
        np.add.at(c, (np.arange(n)[ok4], alt3[ok4]), rng.integers(1, 4, int(ok4.sum())))
        c[(kind > 0.5) & (kind < 0.51)] = 0                                 # sites no sample covers
        tied = (kind > 0.995)                                               # exact ties between two alleles
        c[tied] = 0
        c[tied, ref_idx[tied]] = 3
        c[tied, alt[tied]] = 3
        if zero_depth_sample and s == n_samples - 1:
            c[:] = 0
        sdir = os.path.join(root, "samples", "sample_%d" % (s + 1))
        os.makedirs(os.path.join(sdir, "snps", "output"), exist_ok=True)
        tot = c.sum(1)
        with gzip.open(os.path.join(sdir, "snps", "output", "%s.snps.gz" % species_id), "wt", compresslevel=1) as h:
            h.write("ref_id\tref_pos\tref_allele\tdepth\tcount_a\tcount_c\tcount_g\tcount_t\n")
            row = 0
            for cid, seq in zip(ids, seqs):
                for p, b in enumerate(seq):
                    h.write("%s\t%d\t%s\t%d\t%d\t%d\t%d\t%d\n" % (cid, p + 1, b, tot[row], c[row, 0], c[row, 1],
                                                                  c[row, 2], c[row, 3]))
                    row += 1
        cov = int((tot > 0).sum())
        mean_cov = float(tot.sum()) / cov if cov else 0.0
        with open(os.path.join(sdir, "snps", "summary.txt"), "w") as h:
            h.write("species_id\tgenome_length\tcovered_bases\tfraction_covered\tmean_coverage\taligned_reads\tmapped_reads\n")
            h.write("%s\t%d\t%d\t%s\t%s\t%d\t%d\n" % (species_id, n, cov, cov / float(n), mean_cov, 1000 + s, 900 + s))
        samples.append(sdir)
        counts_all.append(c)
    keys = ["%s|%d|%s" % (cid, p + 1, b) for cid, seq in zip(ids, seqs) for p, b in enumerate(seq)]
    return dict(db=db, samples=samples, species_id=species_id, keys=keys, counts=counts_all, contig_ids=ids,
                contig_seqs=seqs)


def take_reads(reads, order):
    """The reads `order` picks, in that order (ragged columns regathered)."""
    order = np.asarray(order, dtype=np.int64)

    def ragged(data, off):
        n = (off[1:] - off[:-1])[order]
        new_off = np.zeros(order.size + 1, dtype=np.int64)
        np.cumsum(n, out=new_off[1:])
        src = np.repeat(off[:-1][order] - new_off[:-1], n) + np.arange(new_off[-1])
        return data[src], new_off
    seq4, seq_off = ragged(reads.seq4, reads.seq_off)
    qual, qual_off = ragged(reads.qual, reads.qual_off)
    cigar, cigar_off = ragged(reads.cigar, reads.cigar_off)
    return ReadsSoA(pos=reads.pos[order], mapq=reads.mapq[order], flag=reads.flag[order], nm=reads.nm[order],
                    l_seq=reads.l_seq[order], seq_off=seq_off, qual_off=qual_off, cigar_off=cigar_off,
                    seq4=seq4, qual=qual, cigar=cigar)


def concat_reads(parts):
    def offsets(name):
        sizes = np.concatenate([np.diff(getattr(p, name)) for p in parts]) if parts else np.zeros(0, np.int64)
        off = np.zeros(sizes.size + 1, dtype=np.int64)
        np.cumsum(sizes, out=off[1:])
        return off
    cat = lambda name: np.concatenate([getattr(p, name) for p in parts])
    return ReadsSoA(pos=cat('pos'), mapq=cat('mapq'), flag=cat('flag'), nm=cat('nm'), l_seq=cat('l_seq'),
                    seq_off=offsets('seq_off'), qual_off=offsets('qual_off'), cigar_off=offsets('cigar_off'),
                    seq4=cat('seq4'), qual=cat('qual'), cigar=cat('cigar'))


N_MARKER_FAMILIES = 15


def make_pangenome_dataset(n_species=3, genes_per_species=120, n_reads=20000, read_len=150, seed=BASE_SEED + 9,
                           gene_lengths=(420, 900, 1500, 3000), var_len=True, silent_fraction=0.15):
    """What `run_midas.py genes --build_db --align` leaves for --call_genes, in memory: pangenomes of n_species species
    (centroid genes of a few lengths; ids sort differently from pangenome order), 15 marker families per species (one
    family with two genes), reads over the genes in aligner (= random) order, a share of the genes without reads.
    -> dict(species_ids, gene_ids, gene_species, gene_seq, marker, refid, reads)."""
    rng = np.random.default_rng(seed)
    per_class = max(1, genes_per_species // len(gene_lengths))
    species_ids = ["Species_%05d" % (s + 1) for s in range(n_species)]
    gene_ids, gene_species, gene_seq = [], [], []
    first = {}                                        # (species, class) -> global index of its first gene
    sets = []
    for c, glen in enumerate(gene_lengths):
        contigs, reads = make_dataset(n_species=n_species, contigs_per_species=per_class, contig_len=int(glen),
                                      n_reads=max(1, n_reads // len(gene_lengths)), read_len=min(read_len, int(glen) - 9),
                                      seed=seed + 31 * (c + 1), var_len=var_len)
        sets.append((contigs, reads))
    for s in range(n_species):
        for c, (contigs, _) in enumerate(sets):
            first[(s, c)] = len(gene_ids)
            off = contigs.site_offsets()
            for k in range(per_class):
                j = s * per_class + k
                gene_ids.append("%s.peg.%d" % (species_ids[s][-5:], len(gene_ids) - first[(s, 0)] + 1))
                gene_species.append(species_ids[s])
                gene_seq.append(bytes(contigs.ref[off[j]:off[j + 1]]).decode())
    parts, refids = [], []
    for c, (contigs, reads) in enumerate(sets):
        local = np.repeat(np.arange(contigs.n_contigs, dtype=np.int64), np.diff(contigs.read_begin))
        base = np.array([first[(j // per_class, c)] + j % per_class for j in range(contigs.n_contigs)], dtype=np.int64)
        parts.append(reads)
        refids.append(base[local])
    reads, refid = concat_reads(parts), np.concatenate(refids)
    silent = rng.random(len(gene_ids)) < silent_fraction
    order = rng.permutation(refid.size)
    order = order[~silent[refid[order]]]
    reads, refid = take_reads(reads, order), refid[order].astype(np.int32)
    marker = {}
    for s in range(n_species):
        mine = [g for g, sp in zip(gene_ids, gene_species) if sp == species_ids[s]]
        picks = rng.choice(len(mine), size=min(len(mine), N_MARKER_FAMILIES + 1), replace=False)
        for m, g in enumerate(picks):
            marker[mine[int(g)]] = "B%06d" % (min(m, N_MARKER_FAMILIES - 1) + 1)
    return dict(species_ids=species_ids, gene_ids=gene_ids, gene_species=gene_species, gene_seq=gene_seq,
                marker=marker, refid=refid, reads=reads)


def write_pangenome_sample(outdir, db_dir, ds, line_width=70, gz=True):
    """Lay a make_pangenome_dataset() out on disk: the database side (pan_genomes/<species>/centroids.ffn[.gz],
    marker_genes/phyeco.map with rows of other species mixed in, the files utility.check_database wants) and the
    sample side (<outdir>/genes/{species.txt, temp/pangenomes.fa, temp/pangenomes.bam}); BAM @SQ order = pangenome
    order, records in aligner order (not sorted), as `samtools view -b` of bowtie2's output is."""
    import gzip
    import os
    from . import bam
    for d in ("marker_genes", "pan_genomes", "rep_genomes"):
        os.makedirs(os.path.join(db_dir, d), exist_ok=True)
    write_db_tables(db_dir, ds['species_ids'])
    for sp in ds['species_ids']:
        d = os.path.join(db_dir, "pan_genomes", sp)
        os.makedirs(d, exist_ok=True)
        opener = (lambda p: gzip.open(p + ".gz", "wt")) if gz else (lambda p: open(p, "w"))
        with opener(os.path.join(d, "centroids.ffn")) as h:
            for gid, gsp, seq in zip(ds['gene_ids'], ds['gene_species'], ds['gene_seq']):
                if gsp == sp:
                    h.write(">%s\n" % gid)
                    for o in range(0, len(seq), line_width):
                        h.write(seq[o:o + line_width] + "\n")
    with open(os.path.join(db_dir, "marker_genes", "phyeco.map"), "w") as h:
        h.write("species_id\tgenome_id\tgene_id\tgene_length\tmarker_id\n")
        h.write("Species_99999\tSpecies_99999.rep\t99999.peg.1\t800\tB000001\n")
        for gid, gsp, seq in zip(ds['gene_ids'], ds['gene_species'], ds['gene_seq']):
            if gid in ds['marker']:
                h.write("%s\t%s.rep\t%s\t%d\t%s\n" % (gsp, gsp, gid, len(seq), ds['marker'][gid]))
    os.makedirs(os.path.join(outdir, "genes", "temp"), exist_ok=True)
    os.makedirs(os.path.join(outdir, "genes", "output"), exist_ok=True)
    with open(os.path.join(outdir, "genes", "species.txt"), "w") as h:
        h.write("".join(s + "\n" for s in ds['species_ids']))
    with open(os.path.join(outdir, "genes", "temp", "pangenomes.fa"), "w") as h:
        for gid, seq in zip(ds['gene_ids'], ds['gene_seq']):
            h.write(">%s\n%s\n" % (gid, seq.upper()))
    bam.write_bam(os.path.join(outdir, "genes", "temp", "pangenomes.bam"), ds['gene_ids'],
                  [len(s) for s in ds['gene_seq']], ds['refid'], ds['reads'])
