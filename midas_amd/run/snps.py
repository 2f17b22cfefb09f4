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
    """Base class for species -- midas/run/snps.py:12-31"""
    def __init__(self, id):
        self.id = id
        self.paths = {}
        self.aligned_reads = 0
        self.mapped_reads = 0
        self.genome_length = 0
        self.covered_bases = 0
        self.total_depth = 0
        self.fraction_covered = 0
        self.mean_coverage = 0

    def fetch_paths(self, ref_db):
        indir = '%s/rep_genomes/%s' % (ref_db, self.id)
        for ext in ['', '.gz']:
            for type in ['fna', 'features']:
                path = '%s/genome.%s%s' % (indir, type, ext)
                if os.path.isfile(path):
                    self.paths[type] = path


class Contig:
    """Base class for contig -- midas/run/snps.py:33-36"""
    def __init__(self, id):
        self.id = id


def select_species(args):
    """The slice of midas/run/species.py:191-227 this path can honour without a species profile:
    --species_id (checked against the database).  --species_cov / --species_topn need the output of
    `run_midas.py species`, which is outside this build."""
    if not args.get('species_id'):
        sys.exit("\nError: this build only selects species with --species_id "
                 "(--species_cov/--species_topn need `run_midas.py species`, which is out of scope)\n")
    ids = []
    for sp in args['species_id']:
        if not os.path.isdir('%s/rep_genomes/%s' % (args['db'], sp)):
            sys.exit("\nError: Species id not found in database: %s\n" % sp)
        ids.append(sp)
    return ids


def initialize_species(args):
    """midas/run/snps.py:38-53"""
    species = {}
    splist = '%s/snps/species.txt' % args['outdir']
    if args['build_db']:
        with open(splist, 'w') as outfile:
            for id in select_species(args):
                species[id] = Species(id)
                outfile.write(id + '\n')
    elif os.path.isfile(splist):
        for line in open(splist):
            id = line.rstrip()
            species[id] = Species(id)
    for sp in species.values():
        sp.fetch_paths(ref_db=args['db'])
    return species


def initialize_contigs(species):
    """midas/run/snps.py:55-67 (Bio.SeqIO replaced by midas_amd.fasta; same id / upper-cased seq)"""
    contigs = {}
    for sp in species.values():
        if 'fna' not in sp.paths:
            sys.exit("\nError: Could not locate the representative genome of species: %s\n" % sp.id)
        infile = utility.iopen(sp.paths['fna'])
        for rec_id, rec_seq in fasta.parse(infile):
            contig = Contig(rec_id)
            contig.id = rec_id
            contig.seq = rec_seq.upper()
            contig.length = len(contig.seq)
            contig.species_id = sp.id
            contigs[contig.id] = contig
        infile.close()
    return contigs


def build_genome_db(args, species):
    """Build FASTA and BT2 database of representative genomes -- midas/run/snps.py:69-95"""
    outfile = open('/'.join([args['outdir'], 'snps/temp/genomes.fa']), 'w')
    db_stats = {'total_length': 0, 'total_seqs': 0, 'species': 0}
    for sp in species.values():
        db_stats['species'] += 1
        infile = utility.iopen(sp.paths['fna'])
        for rec_id, rec_seq in fasta.parse(infile):
            outfile.write('>%s\n%s\n' % (rec_id, rec_seq.upper()))
            db_stats['total_length'] += len(rec_seq)
            db_stats['total_seqs'] += 1
        infile.close()
    outfile.close()
    print("  total genomes: %s" % db_stats['species'])
    print("  total contigs: %s" % db_stats['total_seqs'])
    print("  total base-pairs: %s" % db_stats['total_length'])
    if not args.get('bowtie2-build'):
        sys.exit("\nError: bowtie2-build not found on PATH (needed for --build_db; the aligner is not part of this build)\n")
    command = '%s ' % args['bowtie2-build']
    command += '--threads %s ' % args['threads']
    command += '%s/snps/temp/genomes.fa ' % args['outdir']
    command += '%s/snps/temp/genomes ' % args['outdir']
    args['log'].write('command: ' + command + '\n')
    process = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    utility.check_exit_code(process, command)


def genome_align(args):
    """Use Bowtie2 to map reads to representative genomes -- midas/run/snps.py:97-128"""
    if not args.get('bowtie2') or not args.get('samtools'):
        sys.exit("\nError: bowtie2 / samtools not found on PATH (needed for --align; the aligner is not part of this build)\n")
    bam_path = os.path.join(args['outdir'], 'snps/temp/genomes.bam')
    command = '%s --no-unal ' % args['bowtie2']
    command += '-x %s ' % '/'.join([args['outdir'], 'snps/temp/genomes'])
    if args['max_reads']: command += '-u %s ' % args['max_reads']
    if args['trim']: command += '--trim3 %s ' % args['trim']
    command += '--%s' % args['speed']
    command += '-local ' if args['mode'] == 'local' else ' '
    command += '--threads %s ' % args['threads']
    command += '-f ' if args['file_type'] == 'fasta' else '-q '
    if args['m2']:
        command += '-1 %s -2 %s ' % (args['m1'], args['m2'])
    elif args['interleaved']:
        command += '--interleaved %s ' % args['m1']
    else:
        command += '-U %s ' % args['m1']
    command += '| %s view -b - ' % args['samtools']
    command += '--threads %s ' % args['threads']
    command += '| %s sort - ' % args['samtools']
    command += '--threads %s ' % args['threads']
    command += '-o %s ' % bam_path
    args['log'].write('command: ' + command + '\n')
    process = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    utility.check_exit_code(process, command)
    print("  finished aligning")


def index_bam(args):
    """midas/run/snps.py:130-139.  The reference shells out to `samtools index` because pysam's
    count_coverage fetches through the .bai; here the index is the per-tile read range table the device
    builds at the start of every pileup pass, so there is nothing to do on the host."""
    start = time()
    print("\nIndexing bamfile")
    args['log'].write("\nIndexing bamfile\n")
    args['log'].write('command: (none) per-tile read index is built on the GPU by index_reads_kernel\n')
    print("  %s minutes" % round((time() - start) / 60, 2))
    print("  %s Gb maximum memory" % utility.max_mem_usage())


def keep_read(aln_len_minus_nm, align_len, qual_sum, query_len, mapq, args):
    """The predicate of midas/run/snps.py:141-162 on already-extracted numbers.  Documentation and host-side
    spot checks only: the pileup evaluates exactly this on the GPU (pileup_tiles.hip, `keep_read` block)."""
    if 100 * aln_len_minus_nm / float(align_len) < args['mapid']:
        return False
    elif qual_sum / float(query_len) < args['readq']:
        return False
    elif mapq < args['mapq']:
        return False
    elif align_len / float(query_len) < args['aln_cov']:
        return False
    return True


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
