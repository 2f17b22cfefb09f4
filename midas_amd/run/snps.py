"""MI355X-native `run_midas.py snps` pipeline: the host side.

Mirrors /root/reference/midas/run/snps.py name for name (Species, Contig, initialize_species,
initialize_contigs, build_genome_db, genome_align, index_bam, keep_read, species_pileup, pysam_pileup,
snps_summary, remove_tmp, run_pipeline) so that scripts/run_midas.py can call it exactly as the
reference's CLI calls the original, and so that the tests read like the reference's own.

What is different underneath:
  * pysam_pileup / species_pileup do not walk the BAM with pysam callbacks; they decode the BAM once
    (native reader), hand the records to the HIP library through the C-ABI (include/midas_snps.h) and
    let the native formatter write <species>.snps.gz.  There is NO CPU fallback: without the library
    or a gfx950 GPU the stage exits with an error.
  * index_bam does not run `samtools index`: the device builds its own per-tile read index.
  * the pileup is one task per *rank* (torch.distributed, one process per GPU), species sharded over
    ranks, not one mp.Pool task per species (which is also what breaks the reference on python3:
    args['log'] is not picklable).
"""

import os
import shutil
import subprocess
import sys
from time import time

import numpy as np

from midas_amd import abi, bam, dist, fasta, utility


class Species:
    """A species of the sample: id, where its representative genome lives, and (after the pileup) its counters."""
    __slots__ = ('id', 'paths', 'aligned_reads', 'mapped_reads', 'genome_length', 'covered_bases', 'total_depth',
                 'fraction_covered', 'mean_coverage')

    def __init__(self, id):
        self.id = id
        self.paths = {}
        self.aligned_reads = self.mapped_reads = self.genome_length = self.covered_bases = self.total_depth = 0
        self.fraction_covered = self.mean_coverage = 0

    def fetch_paths(self, ref_db):
        """genome.fna / genome.features of the species, plain or .gz (the .gz wins when both exist, as in the reference)."""
        base = os.path.join(ref_db, 'rep_genomes', self.id)
        for kind in ('fna', 'features'):
            for suffix in ('', '.gz'):
                candidate = os.path.join(base, 'genome.%s%s' % (kind, suffix))
                if os.path.isfile(candidate):
                    self.paths[kind] = candidate


class Contig:
    """One FASTA record of a representative genome: id, upper-cased sequence, length, owning species."""
    __slots__ = ('id', 'seq', 'length', 'species_id')

    def __init__(self, id, seq='', species_id=None):
        self.id = id
        self.seq = seq
        self.length = len(seq)
        self.species_id = species_id


def select_species(args, per_species='rep_genomes'):
    """Only --species_id can be honoured here: --species_cov / --species_topn read the abundance profile written by
    `run_midas.py species` (midas/run/species.py:191-227), a pipeline outside this build."""
    wanted = args.get('species_id')
    if not wanted:
        sys.exit("\nError: this build only selects species with --species_id "
                 "(--species_cov/--species_topn need `run_midas.py species`, which is out of scope)\n")
    for sp in wanted:
        if not os.path.isdir(os.path.join(args['db'], per_species, sp)):
            sys.exit("\nError: Species id not found in database: %s\n" % sp)
    return list(wanted)


def initialize_species(args):
    """{species_id: Species}: chosen now and recorded in snps/species.txt when the database is being built, else
    read back from that file (midas/run/snps.py:38-53)."""
    listing = os.path.join(args['outdir'], 'snps', 'species.txt')
    if args['build_db']:
        ids = select_species(args)
        with open(listing, 'w') as handle:
            handle.writelines(i + '\n' for i in ids)
    elif os.path.isfile(listing):
        with open(listing) as handle:
            ids = [line.rstrip() for line in handle]
    else:
        ids = []
    species = {i: Species(i) for i in ids}
    for sp in species.values():
        sp.fetch_paths(ref_db=args['db'])
    return species


def _records(sp):
    if 'fna' not in sp.paths:
        sys.exit("\nError: Could not locate the representative genome of species: %s\n" % sp.id)
    with utility.iopen(sp.paths['fna']) as handle:
        for rec_id, rec_seq in fasta.parse(handle):
            yield rec_id, rec_seq.upper()


def initialize_contigs(species):
    """{contig_id: Contig} over every species' representative genome, sequences upper-cased (midas/run/snps.py:55-67)."""
    return {rid: Contig(rid, seq, sp.id) for sp in species.values() for rid, seq in _records(sp)}


def _shell(args, stages):
    """Run `stage | stage | ...` through the shell, logging the command; a failing pipeline ends the run."""
    command = ' | '.join(' '.join(str(x) for x in stage) for stage in stages) + ' '
    args['log'].write('command: ' + command + '\n')
    process = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    utility.check_exit_code(process, command)


def build_genome_db(args, species):
    """snps/temp/genomes.fa = every selected genome, then `bowtie2-build` on it (midas/run/snps.py:69-95)."""
    temp = os.path.join(args['outdir'], 'snps', 'temp')
    n_seqs = n_bases = 0
    with open(os.path.join(temp, 'genomes.fa'), 'w') as out:
        for sp in species.values():
            for rid, seq in _records(sp):
                out.write('>%s\n%s\n' % (rid, seq))
                n_seqs += 1
                n_bases += len(seq)
    print("  total genomes: %s\n  total contigs: %s\n  total base-pairs: %s" % (len(species), n_seqs, n_bases))
    if not args.get('bowtie2-build'):
        sys.exit("\nError: bowtie2-build not found on PATH (needed for --build_db; the aligner is not part of this build)\n")
    _shell(args, [[args['bowtie2-build'], '--threads', args['threads'], os.path.join(temp, 'genomes.fa'),
                   os.path.join(temp, 'genomes')]])


def genome_align(args):
    """bowtie2 (no unaligned reads) | samtools view -b | samtools sort -> snps/temp/genomes.bam, with the same
    switches the reference passes (midas/run/snps.py:97-128)."""
    if not args.get('bowtie2') or not args.get('samtools'):
        sys.exit("\nError: bowtie2 / samtools not found on PATH (needed for --align; the aligner is not part of this build)\n")
    temp = os.path.join(args['outdir'], 'snps', 'temp')
    bt2 = [args['bowtie2'], '--no-unal', '-x', os.path.join(temp, 'genomes')]
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
    view = [args['samtools'], 'view', '-b', '-', '--threads', args['threads']]
    sort = [args['samtools'], 'sort', '-', '--threads', args['threads'], '-o', os.path.join(temp, 'genomes.bam')]
    _shell(args, [bt2, view, sort])
    print("  finished aligning")


def index_bam(args):
    """The reference runs `samtools index` here because pysam fetches through the .bai (midas/run/snps.py:130-139).
    The device builds its own per-tile read index at the start of every pileup pass (index_reads_kernel), so the
    host has nothing to do; the step is kept so that logs and callers look the same."""
    start = time()
    print("\nIndexing bamfile")
    args['log'].write("\nIndexing bamfile\ncommand: (none) per-tile read index is built on the GPU by index_reads_kernel\n")
    print("  %s minutes" % round((time() - start) / 60, 2))
    print("  %s Gb maximum memory" % utility.max_mem_usage())


def keep_read(aln_len_minus_nm, align_len, qual_sum, query_len, mapq, args):
    """The read filter of midas/run/snps.py:141-162 on already-extracted numbers, tests in the reference's order:
    identity, mean quality, mapping quality, aligned fraction.  Documentation and host-side spot checks only: the
    pileup evaluates exactly this on the GPU (pileup_tiles.hip)."""
    tests = (100 * aln_len_minus_nm / float(align_len) < args['mapid'],
             qual_sum / float(query_len) < args['readq'],
             mapq < args['mapq'],
             align_len / float(query_len) < args['aln_cov'])
    return not any(tests)


_ERR_TEXT = {
    abi.ERR_READ_NO_SEQ: "an alignment has no SEQ (the reference raises TypeError in keep_read)",
    abi.ERR_READ_NO_NM: "an alignment has no NM tag (the reference raises KeyError: 'NM' in keep_read)",
    abi.ERR_READ_ZERO_ALIGN: "an alignment has aligned length 0 (the reference raises ZeroDivisionError in keep_read)",
    abi.ERR_READ_NO_QUAL: "an alignment has no QUAL (the reference raises TypeError in np.mean)",
    abi.ERR_READ_CIGAR_OVERRUN: "an alignment's CIGAR is longer than its SEQ (the reference raises IndexError)",
}


def _exit_on(e):
    if isinstance(e, abi.MidasSnpsError):
        msg = _ERR_TEXT.get(e.status, e.message)
        where = " [read %d of the species' records]" % e.read_index if e.read_index >= 0 else ""
        sys.exit("\nError: %s%s\n%s\n" % (msg, where, e.message))
    raise e


def _contig_table(species_ids, contigs, ref_names, ref_lens, refid, reads):
    """(ContigTable, ReadsSoA) for the given species: contigs in BAM header order, reads regrouped to match."""
    sp_index = {s: i for i, s in enumerate(species_ids)}
    order = {n: i for i, n in enumerate(ref_names)}
    mine = [c for c in contigs.values() if c.species_id in sp_index]
    missing = [c.id for c in mine if c.id not in order]
    # a contig that is not in the BAM header: pysam would raise on count_coverage(contig.id, ...)
    if missing:
        sys.exit("\nError: contig '%s' is not in the BAM header (was the genome database rebuilt after alignment?)\n" % missing[0])
    mine.sort(key=lambda c: order[c.id])
    for c in mine:
        if ref_lens[order[c.id]] != c.length:
            sys.exit("\nError: contig '%s' has length %d in the BAM header but %d in the FASTA\n"
                     % (c.id, ref_lens[order[c.id]], c.length))
    ids = [c.id for c in mine]
    sub, read_begin = bam.group_by_contig(ref_names, refid, reads, ids)
    ref = np.frombuffer(''.join(c.seq for c in mine).encode('latin-1'), dtype=np.uint8)
    table = abi.ContigTable(length=[c.length for c in mine], species=[sp_index[c.species_id] for c in mine],
                            read_begin=read_begin, ref=ref, n_species=len(species_ids), ids=ids,
                            species_ids=list(species_ids))
    return table, sub


def _write_species(args, species_id, table, counts, allele):
    """<outdir>/snps/output/<species>.snps.gz -- header + rows in sorted(contig id) order
    (midas/run/snps.py:179-182, 187-192, 201-210), formatted and gzipped by the native writer."""
    out_path = '%s/snps/output/%s.snps.gz' % (args['outdir'], species_id)
    off = table.site_offsets()
    sp = table.species_ids.index(species_id)
    ks = [table.ids.index(cid) for cid in sorted(table.ids)]
    ks = [k for k in ks if table.species[k] == sp]         # a species without contigs still gets its header-only file
    abi.write_table(out_path, [table.ids[k] for k in ks], [allele[off[k]:off[k + 1]] for k in ks],
                    [counts[off[k]:off[k + 1]] for k in ks], gz_level=int(args.get('gz_level', 6)),
                    threads=int(args.get('threads', 1) or 1))


def _pileup_species_set(args, species_ids, contigs, decoded, ctx):
    """count_coverage + keep_read + emit for a set of species on one GPU -> {species_id: aln_stats}"""
    ref_names, ref_lens, refid, reads = decoded
    table, sub = _contig_table(species_ids, contigs, ref_names, ref_lens, refid, reads)
    thr = abi.Thresholds.from_args(args)
    try:
        counts, allele, stats = ctx.pileup(thr, table, sub)
    except abi.MidasSnpsError as e:
        _exit_on(e)
    genome_length = np.bincount(table.species, weights=table.length, minlength=len(species_ids)).astype(np.int64)
    out = {}
    for i, sp in enumerate(species_ids):
        _write_species(args, sp, table, counts, allele)
        out[sp] = {'genome_length': int(genome_length[i]),
                   'total_depth': int(stats[i, abi.STAT_TOTAL_DEPTH]),
                   'covered_bases': int(stats[i, abi.STAT_COVERED_BASES]),
                   'aligned_reads': int(stats[i, abi.STAT_ALIGNED_READS]),
                   'mapped_reads': int(stats[i, abi.STAT_MAPPED_READS])}
    return out


def species_pileup(args, species_id, contigs):
    """midas/run/snps.py:164-216 for ONE species on GPU 0: writes <species>.snps.gz, returns (species_id, aln_stats)."""
    bampath = '%s/snps/temp/genomes.bam' % args['outdir']
    try:
        decoded = abi.read_bam(bampath)
        with abi.Context(int(os.environ.get("LOCAL_RANK", "0"))) as ctx:
            stats = _pileup_species_set(args, [species_id], contigs, decoded, ctx)
    except abi.MidasSnpsError as e:
        _exit_on(e)
    return (species_id, stats[species_id])


def pysam_pileup(args, species, contigs):
    """midas/run/snps.py:219-244.  Name kept for drop-in; there is no pysam underneath."""
    start = time()
    rank, ws = dist.world()
    if rank == 0:
        print("\nCounting alleles")
        args['log'].write("\nCounting alleles\n")

    bampath = '%s/snps/temp/genomes.bam' % args['outdir']
    try:
        decoded = abi.read_bam(bampath)
    except abi.MidasSnpsError as e:
        sys.exit("\nError: could not read %s\n%s\n" % (bampath, e.message))
    ref_names, ref_lens, refid, reads = decoded

    # species -> rank, by aligned reads + genome length (every rank computes the same assignment)
    all_ids = sorted(species)
    reads_per_ref = np.bincount(refid, minlength=len(ref_names)) if refid.size else np.zeros(len(ref_names), np.int64)
    ref_index = {n: i for i, n in enumerate(ref_names)}
    weight = {sp: 0.0 for sp in all_ids}
    for c in contigs.values():
        if c.species_id in weight:
            weight[c.species_id] += 150.0 * float(reads_per_ref[ref_index[c.id]] if c.id in ref_index else 0) + c.length
    owner = dist.shard_species(weight, ws)
    mine = [sp for sp in all_ids if owner[sp] == rank]

    local = {}
    if mine:
        try:
            with abi.Context(int(os.environ.get("LOCAL_RANK", "0"))) as ctx:
                local = _pileup_species_set(args, mine, contigs, decoded, ctx)
        except abi.MidasSnpsError as e:
            _exit_on(e)

    # one all-gather of the per-species summary rows; per-site output stays on its rank
    rows = np.zeros((len(all_ids), 5), dtype=np.int64)
    for i, sp in enumerate(all_ids):
        if sp in local:
            st = local[sp]
            rows[i] = [st['genome_length'], st['covered_bases'], st['total_depth'], st['aligned_reads'], st['mapped_reads']]
    rows = dist.all_gather_summary(rows)

    # update alignment stats for species objects -- midas/run/snps.py:230-241
    for i, species_id in enumerate(all_ids):
        sp = species[species_id]
        sp.genome_length = int(rows[i, 0])
        sp.covered_bases = int(rows[i, 1])
        sp.total_depth = int(rows[i, 2])
        sp.aligned_reads = int(rows[i, 3])
        sp.mapped_reads = int(rows[i, 4])
        if sp.genome_length > 0:
            sp.fraction_covered = sp.covered_bases / float(sp.genome_length)
        if sp.covered_bases > 0:
            sp.mean_coverage = sp.total_depth / float(sp.covered_bases)

    if rank == 0:
        print("  %s minutes" % round((time() - start) / 60, 2))
        print("  %s Gb maximum memory" % utility.max_mem_usage())


def snps_summary(args, species):
    """Get summary of mapping statistics -- midas/run/snps.py:247-262"""
    fields = ['species_id', 'genome_length', 'covered_bases', 'fraction_covered', 'mean_coverage', 'aligned_reads', 'mapped_reads']
    outfile = open(args['outdir'] + '/snps/summary.txt', 'w')
    outfile.write('\t'.join(fields) + '\n')
    for sp in species.values():
        outfile.write(sp.id + '\t')
        outfile.write(str(sp.genome_length) + '\t')
        outfile.write(str(sp.covered_bases) + '\t')
        outfile.write(str(sp.fraction_covered) + '\t')
        outfile.write(str(sp.mean_coverage) + '\t')
        outfile.write(str(sp.aligned_reads) + '\t')
        outfile.write(str(sp.mapped_reads) + '\n')
    outfile.close()


def remove_tmp(args):
    """Remove specified temporary files -- midas/run/snps.py:264-266"""
    shutil.rmtree('/'.join([args['outdir'], 'snps/temp']))


def run_pipeline(args):
    """Run entire pipeline -- midas/run/snps.py:268-305"""
    rank, ws = dist.init_from_env()

    print("\nReading reference data")
    start = time()
    species = initialize_species(args)
    contigs = initialize_contigs(species)
    print("  %s minutes" % round((time() - start) / 60, 2))
    print("  %s Gb maximum memory" % utility.max_mem_usage())

    if args['build_db'] and rank == 0:
        print("\nBuilding database of representative genomes")
        args['log'].write("\nBuilding database of representative genomes\n")
        start = time()
        build_genome_db(args, species)
        print("  %s minutes" % round((time() - start) / 60, 2))
        print("  %s Gb maximum memory" % utility.max_mem_usage())

    if args['align'] and rank == 0:
        args['file_type'] = utility.auto_detect_file_type(args['m1'])
        print("\nMapping reads to representative genomes")
        args['log'].write("\nMapping reads to representative genomes\n")
        start = time()
        genome_align(args)
        print("  %s minutes" % round((time() - start) / 60, 2))
        print("  %s Gb maximum memory" % utility.max_mem_usage())
    dist.barrier()

    if args['call']:
        if rank == 0:
            index_bam(args)
        pysam_pileup(args, species, contigs)
        if rank == 0:
            snps_summary(args, species)
    dist.barrier()

    if args['remove_temp'] and rank == 0:
        remove_tmp(args)
