"""MI355X-native `run_midas.py genes` pipeline: the host side (SURVEY.md 8f "next" #4).

Same steps, files and numbers as /root/reference/midas/run/genes.py -- pangenome database from the species' centroid
genes, bowtie2 alignment, then per gene: aligned reads, reads passing the read filter, depth, copy number relative to
the median depth of the species' marker genes -- with the pass over the BAM (count_mapped_bp, :165-180) done by the
device through the C-ABI (midas_genes_count): every read of a gene is filtered and its aligned length / gene length
added to the gene's depth in BAM order, so the fp64 sums are the reference's bit for bit.  No CPU fallback for it.
"""

import csv
import os
import shutil
import subprocess
import sys
from collections import defaultdict
from time import time

import numpy as np

from midas_amd import abi, bam, dist, fasta, utility
from midas_amd.run.snps import select_species


class Species:
    """A species of the sample: its pangenome files and, after the pass, its summary numbers."""

    def __init__(self, id):
        self.id = id
        self.paths = {}
        self.genes = []                      # depth of every gene, in pangenome order
        self.pangenome_size = 0
        self.aligned_reads = 0
        self.mapped_reads = 0
        self.markers = defaultdict(float)    # marker family -> summed depth of its genes
        self.covered_genes = 0
        self.mean_coverage = 0
        self.fraction_covered = 0
        self.marker_coverage = 0

    def init_ref_db(self, ref_db):
        """centroids.ffn / cluster_info.txt / gene_info.txt of the species, plain or .gz (the .gz wins when both exist)."""
        self.dir = os.path.join(ref_db, 'pan_genomes', self.id)
        for name in ('centroids.ffn', 'cluster_info.txt', 'gene_info.txt'):
            for suffix in ('', '.gz'):
                candidate = os.path.join(self.dir, name + suffix)
                if os.path.isfile(candidate):
                    self.paths[name] = candidate


class Gene:
    """A centroid gene of a pangenome."""
    __slots__ = ('id', 'species_id', 'length', 'aligned_reads', 'mapped_reads', 'depth', 'copies', 'marker_id')

    def __init__(self, id, species_id=None, length=0):
        self.id = id
        self.species_id = species_id
        self.length = length
        self.aligned_reads = self.mapped_reads = 0
        self.depth = self.copies = 0.0
        self.marker_id = None


def initialize_species(args):
    """{species_id: Species}: chosen now (and written to genes/species.txt) when the database is being built, else read
    back from that file (genes.py:33-50)."""
    listing = os.path.join(args['outdir'], 'genes', 'species.txt')
    if args['build_db']:
        ids = select_species(args, 'pan_genomes')
        with open(listing, 'w') as handle:
            handle.writelines(i + '\n' for i in ids)
    elif os.path.isfile(listing):
        with open(listing) as handle:
            ids = [line.rstrip() for line in handle]
    else:
        ids = []
    species = {i: Species(i) for i in ids}
    for sp in species.values():
        sp.init_ref_db(args['db'])
    return species


def _centroids(sp):
    if 'centroids.ffn' not in sp.paths:
        sys.exit("\nError: Could not locate the pangenome of species: %s\n" % sp.id)
    with utility.iopen(sp.paths['centroids.ffn']) as handle:
        for rec_id, rec_seq in fasta.parse(handle):
            yield rec_id, rec_seq


def initialize_genes(args, species):
    """{gene_id: Gene} in pangenome order (species by species, genes in FASTA order), with lengths and -- from
    marker_genes/phyeco.map -- the marker family of the genes that are universal single-copy markers (genes.py:63-85)."""
    genes = {}
    for sp in species.values():
        for gid, seq in _centroids(sp):
            genes[gid] = Gene(gid, sp.id, len(seq))
            sp.pangenome_size += 1
    with utility.iopen(os.path.join(args['db'], 'marker_genes', 'phyeco.map')) as handle:
        for row in csv.DictReader(handle, delimiter='\t'):
            gene = genes.get(row['gene_id'])
            if gene is not None:
                gene.marker_id = row['marker_id']
    return genes


def _shell(args, command):
    args['log'].write('command: ' + command + '\n')
    process = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    utility.check_exit_code(process, command)


def build_pangenome_db(args, species):
    """genes/temp/pangenomes.fa (+ .map: gene -> species) from the centroid genes, then `bowtie2-build` (genes.py:87-116)."""
    temp = os.path.join(args['outdir'], 'genes', 'temp')
    n_seqs = n_bases = 0
    with open(os.path.join(temp, 'pangenomes.fa'), 'w') as fa, open(os.path.join(temp, 'pangenomes.map'), 'w') as mp:
        for sp in species.values():
            for gid, seq in _centroids(sp):
                fa.write('>%s\n%s\n' % (gid, seq.upper()))
                mp.write('%s\t%s\n' % (gid, sp.id))
                n_seqs += 1
                n_bases += len(seq)
    print("  total species: %s\n  total genes: %s\n  total base-pairs: %s" % (len(species), n_seqs, n_bases))
    if not args.get('bowtie2-build'):
        sys.exit("\nError: bowtie2-build not found on PATH (needed for --build_db; the aligner is not part of this build)\n")
    _shell(args, ' '.join(str(x) for x in (args['bowtie2-build'], '--threads', args['threads'],
                                           os.path.join(temp, 'pangenomes.fa'), os.path.join(temp, 'pangenomes'))) + ' ')


def pangenome_align(args):
    """bowtie2 (no unaligned reads) | samtools view -b > genes/temp/pangenomes.bam, unsorted, with the reference's
    switches (genes.py:118-146)."""
    if not args.get('bowtie2') or not args.get('samtools'):
        sys.exit("\nError: bowtie2 / samtools not found on PATH (needed for --align; the aligner is not part of this build)\n")
    temp = os.path.join(args['outdir'], 'genes', 'temp')
    bt2 = [args['bowtie2'], '--no-unal', '-x', os.path.join(temp, 'pangenomes')]
    if args['max_reads']:
        bt2 += ['-u', args['max_reads']]
    if args['trim']:
        bt2 += ['--trim3', args['trim']]
    bt2 += ['--%s%s' % (args['speed'], '-local' if args['mode'] == 'local' else ''), '--threads', args['threads'],
            '-f' if args['file_type'] == 'fasta' else '-q']
    if args['m2']:
        bt2 += ['-1', args['m1'], '-2', args['m2']]
    elif args['interleaved']:
        bt2 += ['--interleaved', args['m1']]
    else:
        bt2 += ['-U', args['m1']]
    view = [args['samtools'], 'view', '--threads', args['threads'], '-b', '-', '>', os.path.join(temp, 'pangenomes.bam')]
    _shell(args, ' '.join(str(x) for x in bt2) + ' | ' + ' '.join(str(x) for x in view))
    print("  finished aligning")


def fold_counts(species, genes, gene_ids, aligned, mapped, depth):
    """The per-gene device results into Gene / Species, then the species summaries of genes.py:182-199: covered genes,
    mean depth over the covered ones (np.mean over the depths in pangenome order, like the reference), fraction covered."""
    for k, gid in enumerate(gene_ids):
        gene = genes[gid]
        gene.aligned_reads, gene.mapped_reads, gene.depth = int(aligned[k]), int(mapped[k]), float(depth[k])
        sp = species[gene.species_id]
        sp.aligned_reads += gene.aligned_reads
        sp.mapped_reads += gene.mapped_reads
    for gene in genes.values():
        species[gene.species_id].genes.append(gene.depth)
    for sp in species.values():
        covered = [d for d in sp.genes if d > 0]
        sp.covered_genes = len(covered)
        sp.mean_coverage = np.mean(covered) if covered else 0
        sp.fraction_covered = sp.covered_genes / float(sp.pangenome_size)


def _missing_gene_check(ref_names, refid, genes):
    """The reference: KeyError in genes[bamfile.getrname(...)] at the first read of a gene the database does not have."""
    if all(n in genes for n in ref_names):
        return
    used = set(np.unique(refid).tolist())
    bad = [n for i, n in enumerate(ref_names) if n not in genes and i in used]
    if bad:
        sys.exit("\nError: gene '%s' of the BAM header is not in the pangenome database\n" % bad[0])


def _thresholds(args):
    return abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, **{k: args[k] for k in ('mapid', 'readq', 'mapq', 'aln_cov')}))


def _slices_chain(bam_path, rank, ws):
    """Every rank walks its share of the (unsorted) BAM's bytes; the slices are accepted if they chain: slice 0 starts at the
    header's end, every slice ends where the next one starts (a guessed record boundary is confirmed by a walk that began at
    an exact one), the last ends with the file.  -> this rank's slice, or None (every rank then decodes the whole file)."""
    error, sl = None, None
    try:
        sl = abi.BamSlice(bam_path, rank, ws)
    except abi.MidasSnpsError as e:
        error = "\nError: could not read %s\n%s\n" % (bam_path, e.message)
    dist.agree_or_exit(error)
    head = dist.all_gather_i64(np.array([sl.first, sl.end, sl.rec_begin, sl.total, len(sl.ref_names)], np.int64))
    ok = bool((head[:, 4] == head[0, 4]).all() and head[0, 0] == head[0, 2] and head[-1, 1] == head[0, 3])
    for r in range(ws - 1):
        ok = ok and head[r, 1] == head[r + 1, 0]
    if not ok:
        sl.close()
        return None
    return sl


def _count_below_the_species(args, species, genes, ctx, mine, owner, sl):
    """N ranks, genes.py:165-199 below the species: a rank decodes ITS slice of the unsorted BAM and turns every read into a
    term (midas_genes_terms); the (gene, term) pairs go to the rank that owns the gene's species -- one all-to-all, and since the
    slices are in file order the pairs a rank receives, taken by source rank, are too -- and the owner adds a gene's terms up
    in that order (midas_genes_sum): the reference's running fp64 sum, bit for bit, with no rank decoding the whole file."""
    rank, ws = dist.world()
    bam_path = os.path.join(args['outdir'], 'genes', 'temp', 'pangenomes.bam')
    error, refid, reads, term = None, None, None, None
    ref_names = sl.ref_names
    try:
        refid, reads = sl.load_ranges([(sl.first, sl.end)])
        _missing_gene_check(ref_names, refid, genes)
    except abi.MidasSnpsError as e:
        error = "\nError: could not read %s\n%s\n" % (bam_path, e.message)
    except SystemExit as e:
        error = dist.exit_message(e)
    dist.agree_or_exit(error)
    n_local = int(reads.n_reads)
    per_rank = dist.all_gather_i64([n_local])[:, 0]
    base = int(per_rank[:rank].sum())                 # this slice's first read in the BAM
    if rank == 0:
        line = "rank-local BAM decode (genes): %d slices chained, %d records; records decoded per rank: %s" % (
            ws, int(per_rank.sum()), ' '.join(str(int(x)) for x in per_rank))
        print("  " + line)
        if args.get('log') is not None:
            args['log'].write(line + "\n")
    lengths = np.array([genes[n].length if n in genes else sl.ref_lens[i] for i, n in enumerate(ref_names)], dtype=np.int64)
    try:
        term = ctx.genes_terms(_thresholds(args), reads, refid, lengths)
    except abi.MidasSnpsError as e:
        where = " [read %d of the BAM]" % (e.read_index + base) if e.read_index >= 0 else ""
        error = "\nError: %s%s\n" % (e.message, where)
    dist.agree_or_exit(error)
    # gene (header index) -> the rank that owns its species; -1: not in the database (no read is on such a gene, checked above)
    ref_owner = np.array([owner[genes[n].species_id] if n in genes else -1 for n in ref_names], dtype=np.int64)
    # (every record of a slice is on a gene of the header and of the database: _missing_gene_check and midas_genes_terms said so
    # above -- a negative index here would silently wrap)
    assert n_local == 0 or (int(refid.min()) >= 0 and int(ref_owner[refid].min()) >= 0), "record on no gene of the database"
    dest = ref_owner[refid] if n_local else np.zeros(0, np.int64)
    order = np.argsort(dest, kind='stable')           # (stable: file order survives inside a destination)
    cuts = np.searchsorted(dest[order], np.arange(ws + 1))
    g_parts = [refid[order[cuts[r]:cuts[r + 1]]].astype(np.int32) for r in range(ws)]
    t_parts = [term[order[cuts[r]:cuts[r + 1]]] for r in range(ws)]
    got_g = np.concatenate(dist.all_to_all_v(g_parts))
    got_t = np.concatenate(dist.all_to_all_v(t_parts))
    # this rank's genes, renumbered
    gene_ids = [n for n in ref_names if n in genes and genes[n].species_id in mine]
    local = np.full(len(ref_names), -1, dtype=np.int64)
    local[[i for i, n in enumerate(ref_names) if n in genes and genes[n].species_id in mine]] = np.arange(len(gene_ids))
    try:
        aligned, mapped, depth = ctx.genes_sum(local[got_g], got_t, len(gene_ids))
    except abi.MidasSnpsError as e:
        error = "\nError: %s\n" % e.message
    dist.agree_or_exit(error)
    fold_counts(species, genes, gene_ids, aligned, mapped, depth)
    # the sample's totals (midas/run/genes.py:196-197 prints them once): every rank holds its own species' share
    totals = dist.all_gather_i64([sum(sp.aligned_reads for sp in species.values() if sp.id in mine),
                                  sum(sp.mapped_reads for sp in species.values() if sp.id in mine)]).sum(axis=0)
    if rank == 0:
        print("  total aligned reads: %s" % int(totals[0]))
        print("  total mapped reads: %s" % int(totals[1]))
    return 0.0


def count_mapped_bp(args, species, genes, ctx, mine=None, owner=None):
    """genes.py:165-199 with the BAM pass on the device: native BAM decode, one midas_genes_count call.  `mine` (N > 1):
    the species this rank owns -- only reads on their genes are counted here; with `owner` (species -> rank) and a BAM whose
    slices chain, every rank decodes only its slice (_count_below_the_species)."""
    bam_path = os.path.join(args['outdir'], 'genes', 'temp', 'pangenomes.bam')
    if mine is not None and owner is not None and hasattr(ctx, 'genes_terms'):
        sl = _slices_chain(bam_path, *dist.world())
        if sl is not None:
            try:
                return _count_below_the_species(args, species, genes, ctx, mine, owner, sl)
            finally:
                sl.close()
    try:
        ref_names, ref_lens, refid, reads = abi.read_bam(bam_path)
    except abi.MidasSnpsError as e:
        sys.exit("\nError: could not read %s\n%s\n" % (bam_path, e.message))
    _missing_gene_check(ref_names, refid, genes)
    gene_ids = list(ref_names)
    if mine is not None:     # this rank's genes, and the reads on them (BAM order inside a gene is kept)
        gene_ids = [n for n in ref_names if n in genes and genes[n].species_id in mine]
        reads, begin = bam.group_by_contig(ref_names, refid, reads, gene_ids)
        refid = np.repeat(np.arange(len(gene_ids), dtype=np.int32), np.diff(begin))
        ref_lens = [genes[n].length for n in gene_ids]
    lengths = np.array([genes[n].length if n in genes else ref_lens[i] for i, n in enumerate(gene_ids)], dtype=np.int64)
    try:
        aligned, mapped, depth, ms = ctx.genes_count(_thresholds(args), reads, refid, lengths)
    except abi.MidasSnpsError as e:
        where = " [read %d of the BAM]" % e.read_index if e.read_index >= 0 else ""
        sys.exit("\nError: %s%s\n" % (e.message, where))
    known = [k for k, n in enumerate(gene_ids) if n in genes]
    fold_counts(species, genes, [gene_ids[k] for k in known], aligned[known], mapped[known], depth[known])
    print("  total aligned reads: %s" % sum(sp.aligned_reads for sp in species.values()))
    print("  total mapped reads: %s" % sum(sp.mapped_reads for sp in species.values()))
    return ms


def normalize(args, species, genes):
    """Copy number of a gene = its depth over the median depth of the species' marker families (genes.py:201-215)."""
    for gene in genes.values():
        if gene.marker_id is not None:
            species[gene.species_id].markers[gene.marker_id] += gene.depth
    for sp in species.values():
        sp.marker_coverage = np.median(list(sp.markers.values()))      # nan (with numpy's warning) when there are none
    for gene in genes.values():
        mc = species[gene.species_id].marker_coverage
        if mc > 0:
            gene.copies = gene.depth / mc


SUMMARY_FIELDS = ('pangenome_size', 'covered_genes', 'fraction_covered', 'mean_coverage', 'marker_coverage', 'aligned_reads',
                  'mapped_reads')


def write_results(args, species, genes, mine=None):
    """genes/output/<species>.genes.gz (genes in sorted id order) and genes/summary.txt (genes.py:217-244).  N > 1: a rank
    writes the tables of its own species (`mine`), the summary rows are all-gathered and rank 0 writes summary.txt."""
    if mine is not None:
        return _write_results_sharded(args, species, genes, mine)
    handles = {}
    for sp in species.values():
        handles[sp.id] = utility.iopen(os.path.join(args['outdir'], 'genes', 'output', '%s.genes.gz' % sp.id), 'w')
        handles[sp.id].write('gene_id\tcount_reads\tcoverage\tcopy_number\n')
    for gid in sorted(genes):
        gene = genes[gid]
        handles[gene.species_id].write('%s\t%s\t%s\t%s\n' % (gene.id, gene.mapped_reads, gene.depth, gene.copies))
    for handle in handles.values():
        handle.close()
    with open(os.path.join(args['outdir'], 'genes', 'summary.txt'), 'w') as handle:
        handle.write('\t'.join(('species_id',) + SUMMARY_FIELDS) + '\n')
        for sp in species.values():
            handle.write('\t'.join([sp.id] + [str(getattr(sp, f)) for f in SUMMARY_FIELDS]) + '\n')


def _write_results_sharded(args, species, genes, mine):
    handles = {}
    for sp in species.values():
        if sp.id in mine:
            handles[sp.id] = utility.iopen(os.path.join(args['outdir'], 'genes', 'output', '%s.genes.gz' % sp.id), 'w')
            handles[sp.id].write('gene_id\tcount_reads\tcoverage\tcopy_number\n')
    for gid in sorted(genes):
        gene = genes[gid]
        if gene.species_id in mine:
            handles[gene.species_id].write('%s\t%s\t%s\t%s\n' % (gene.id, gene.mapped_reads, gene.depth, gene.copies))
    for handle in handles.values():
        handle.close()
    # one row per species, zero outside this rank's: the owner's numbers survive the sum over ranks
    rows = np.zeros((len(species), len(SUMMARY_FIELDS)), dtype=np.float64)
    for k, sp in enumerate(species.values()):
        if sp.id in mine:
            rows[k] = [float(getattr(sp, f)) for f in SUMMARY_FIELDS]
    rows = dist.all_gather_rows_f64(rows)
    if dist.world()[0] != 0:
        return
    ints = ('pangenome_size', 'covered_genes', 'aligned_reads', 'mapped_reads')
    with open(os.path.join(args['outdir'], 'genes', 'summary.txt'), 'w') as handle:
        handle.write('\t'.join(('species_id',) + SUMMARY_FIELDS) + '\n')
        for k, sp in enumerate(species.values()):
            cells = []
            for f, v in zip(SUMMARY_FIELDS, rows[k]):
                if f in ints or (f == 'mean_coverage' and rows[k][1] == 0):   # the reference's mean of no genes is the int 0
                    cells.append(str(int(v)))
                else:
                    cells.append(str(np.float64(v)) if f != 'fraction_covered' else str(float(v)))
            handle.write('\t'.join([sp.id] + cells) + '\n')


def pangenome_coverage(args, species, genes, make_context=None):
    """N > 1 (torchrun): species are dealt to the ranks by pangenome size; a rank counts the reads on its species' genes,
    writes their tables, and one all-gather of the summary rows lets rank 0 write summary.txt."""
    rank, ws = dist.world()
    mine, owner = None, None
    if ws > 1:
        owner = dist.shard_species({sp.id: float(sp.pangenome_size) for sp in species.values()}, ws)
        mine = {sp for sp, r in owner.items() if r == rank}
    make_context = make_context or (lambda: abi.Context(int(os.environ.get("LOCAL_RANK", "0"))))
    error, ms = None, 0.0
    try:
        with make_context() as ctx:
            line = dist.attach_context(ctx)      # (N ranks: the RCCL communicator of their devices, for the all-to-all of the pairs)
            if line and rank == 0 and args.get('log') is not None:
                args['log'].write(line + "\n")
            try:
                ms = count_mapped_bp(args, species, genes, ctx, mine, owner)
            finally:
                dist.detach_context()
    except abi.MidasSnpsError as e:
        error = "\nError: %s\n" % e.message
    except SystemExit as e:
        error = dist.exit_message(e)
    except Exception as e:      # (an OSError from the decoder, a MemoryError ...: the other ranks must not wait for this one)
        error = "\nError: %s: %s\n" % (type(e).__name__, e)
    dist.agree_or_exit(error)       # every rank leaves with the failing one, in front of the summary all-gather
    normalize(args, species, genes)
    write_results(args, species, genes, mine)
    return ms


def remove_tmp(args):
    shutil.rmtree(os.path.join(args['outdir'], 'genes', 'temp'))


def run_pipeline(args):
    def timed(title, log_title, fn, *a):
        start = time()
        print("\n" + title)
        if log_title:
            args['log'].write("\n" + log_title + "\n")
        out = fn(*a)
        print("  %s minutes" % round((time() - start) / 60, 2))
        print("  %s Gb maximum memory" % utility.max_mem_usage())
        return out

    # (N ranks meet in the sample's temp directory: no process group, no torch -- midas_amd/dist.py)
    rank, ws = dist.init_from_env(rendezvous_dir=os.path.join(args['outdir'], 'genes', 'temp'))
    species = timed("Reading reference data", None, initialize_species, args)
    genes = initialize_genes(args, species)
    if args['build_db'] and rank == 0:
        timed("Building pangenome database", "Building pangenome database", build_pangenome_db, args, species)
    if args['align'] and rank == 0:
        args['file_type'] = utility.auto_detect_file_type(args['m1'])
        timed("Aligning reads to pangenomes", "Aligning reads to pangenomes", pangenome_align, args)
    dist.barrier()
    if args['cov']:
        timed("Computing coverage of pangenomes", "Computing coverage of pangenomes", pangenome_coverage, args, species, genes)
    dist.barrier()
    dist.finalize()
    if args['remove_temp'] and rank == 0:
        remove_tmp(args)
