#!/usr/bin/env python
"""run_midas.py snps | genes -- per-sample pileup + allele counting, and pangenome gene coverage, on MI355X.

Drop-in for the `snps` command of the reference's scripts/run_midas.py: same positional arguments, option
names, defaults and output layout (<outdir>/snps/{output/<species>.snps.gz, species.txt, summary.txt, log.txt,
readme.txt, temp/}), so existing command lines keep working.  What the options mean is the reference's
(scripts/run_midas.py:338-430); how this file is written is not.  `genes` (gene coverage over the species'
pangenomes, reference options :205-336) is served the same way; `species` is another pipeline and not part of
this build.

Multi-GPU: start it under `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1`; species
are dealt to the ranks, each rank writes the tables of its species, rank 0 writes summary.txt.
"""

import argparse
import os
import platform
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from midas_amd import utility  # noqa: E402

RANK = int(os.environ.get('RANK', '0'))

# (group title, [(flags, argparse keywords)])
OPTION_GROUPS = [
    ("Stages (any subset; none given = all three)", [
        (['--build_db'], dict(action='store_true', help="concatenate the representative genomes and index them with bowtie2-build")),
        (['--align'], dict(action='store_true', help="map the reads with bowtie2 | samtools view | samtools sort")),
        (['--pileup'], dict(action='store_true', dest='call', help="count A/C/G/T per genomic site (this is the GPU stage)")),
    ]),
    ("Which species (for --build_db)", [
        (['-d'], dict(dest='db', default=os.environ.get('MIDAS_DB'), help="MIDAS reference database (default: $MIDAS_DB)")),
        (['--species_cov'], dict(type=float, metavar='FLOAT', help="species whose genome coverage exceeds this (3.0); needs `run_midas.py species` output")),
        (['--species_topn'], dict(type=int, metavar='INT', help="the N most abundant species; needs `run_midas.py species` output")),
        (['--species_id'], dict(metavar='ID[,ID...]', help="these species, comma separated")),
    ]),
    ("Reads and aligner (for --align)", [
        (['-1'], dict(dest='m1', help="FASTA/FASTQ of unpaired reads or of the first mates (.gz / .bz2 accepted)")),
        (['-2'], dict(dest='m2', help="FASTA/FASTQ of the second mates")),
        (['--interleaved'], dict(action='store_true', help="-1 holds both mates, interleaved")),
        (['-s'], dict(dest='speed', default='very-sensitive', choices=['very-fast', 'fast', 'sensitive', 'very-sensitive'],
                      help="bowtie2 preset (very-sensitive)")),
        (['-n'], dict(dest='max_reads', type=int, help="use only the first N reads (all)")),
        (['-m'], dict(dest='mode', default='global', choices=['local', 'global'], help="end-to-end or local alignment (global)")),
        (['-t'], dict(dest='threads', default=1, help="CPU threads for the aligner and the table writer (1)")),
    ]),
    ("Read and base filters (for --pileup)", [
        (['--mapid'], dict(type=float, default=94.0, metavar='FLOAT', help="drop reads below this percent identity (94.0)")),
        (['--mapq'], dict(type=int, default=20, metavar='INT', help="drop reads below this mapping quality (20)")),
        (['--baseq'], dict(type=int, default=30, metavar='INT', help="ignore bases below this quality (30)")),
        (['--readq'], dict(type=int, default=20, metavar='INT', help="drop reads whose mean quality is below this (20)")),
        (['--aln_cov'], dict(type=float, default=0.75, metavar='FLOAT', help="drop reads aligned over less than this fraction of their length (0.75)")),
        (['--trim'], dict(type=int, default=0, metavar='INT', help="bases trimmed off the 3' end before alignment (0)")),
        (['--discard'], dict(action='store_true', help="accepted for compatibility; has no effect (as in the reference)")),
        (['--baq'], dict(action='store_true', help="accepted for compatibility; has no effect (as in the reference)")),
        (['--adjust_mq'], dict(action='store_true', help="accepted for compatibility; has no effect (as in the reference)")),
    ]),
]


GENES_OPTION_GROUPS = [
    ("Stages (any subset; none given = all three)", [
        (['--build_db'], dict(action='store_true', help="concatenate the species' centroid genes and index them with bowtie2-build")),
        (['--align'], dict(action='store_true', help="map the reads with bowtie2 | samtools view")),
        (['--call_genes'], dict(action='store_true', dest='cov', help="reads, depth and copy number per gene (this is the GPU stage)")),
    ]),
    OPTION_GROUPS[1],
    ("Reads and aligner (for --align)", [
        (['-1'], dict(dest='m1', required=True, help="FASTA/FASTQ of unpaired reads or of the first mates (.gz / .bz2 accepted)")),
        (['-2'], dict(dest='m2', help="FASTA/FASTQ of the second mates")),
        (['--interleaved'], dict(action='store_true', help="-1 holds both mates, interleaved")),
        (['-s'], dict(dest='speed', default='very-sensitive', choices=['very-fast', 'fast', 'sensitive', 'very-sensitive'],
                      help="bowtie2 preset (very-sensitive)")),
        (['-m'], dict(dest='mode', default='local', choices=['local', 'global'], help="local or end-to-end alignment (local)")),
        (['-n'], dict(dest='max_reads', type=int, help="use only the first N reads (all)")),
        (['-t'], dict(dest='threads', default=1, help="CPU threads for the aligner (1)")),
    ]),
    ("Read filters (for --call_genes)", [
        (['--readq'], dict(type=int, default=20, metavar='INT', help="drop reads whose mean quality is below this (20)")),
        (['--mapid'], dict(type=float, default=94.0, metavar='FLOAT', help="drop reads below this percent identity (94.0)")),
        (['--mapq'], dict(type=int, default=0, metavar='INT', help=argparse.SUPPRESS)),
        (['--aln_cov'], dict(type=float, default=0.75, metavar='FLOAT', help="drop reads aligned over less than this fraction of their length (0.75)")),
        (['--trim'], dict(type=int, default=0, metavar='INT', help="bases trimmed off the 3' end before alignment (0)")),
    ]),
]


def die(message):
    sys.exit("\nError: %s\n" % message)


def get_program():
    """First positional word: `snps` or `genes`."""
    word = sys.argv[1] if len(sys.argv) > 1 else '-h'
    if word in ('-h', '--help'):
        print("run_midas.py <command> [options]\n\n"
              "  snps   count alleles at every site of the representative genomes of a sample's abundant species\n"
              "         (pileup on the MI355X); `run_midas.py snps -h` lists the options\n"
              "  genes  reads, depth and copy number of every gene of the species' pangenomes (read filter and per-gene\n"
              "         sums on the MI355X); `run_midas.py genes -h` lists the options\n\n"
              "species is not part of this build.")
        sys.exit(0)
    if word == 'species':
        die("'%s' is not part of this build (only the snps and genes paths are)" % word)
    if word not in ('snps', 'genes'):
        die("Unrecognized command: '%s'" % word)
    return word


def build_genes_parser():
    parser = argparse.ArgumentParser(
        prog='run_midas.py genes', formatter_class=argparse.RawTextHelpFormatter,
        description="Map a metagenome to the pangenomes (centroid genes) of its abundant species and report, per gene, the\n"
                    "reads that pass the filter, the depth and the copy number relative to the species' marker genes.\n"
                    "Stages: --build_db, --align, --call_genes (the last one runs on the GPU).",
        epilog="examples:\n"
               "  run_midas.py genes OUT -1 reads_1.fq.gz -2 reads_2.fq.gz\n"
               "  run_midas.py genes OUT -1 reads.fq.gz --call_genes --mapid 96")
    parser.add_argument('program', help=argparse.SUPPRESS)
    parser.add_argument('outdir', help="sample directory (its name is the sample id)")
    parser.add_argument('--remove_temp', action='store_true', help="delete <outdir>/genes/temp when done")
    for title, options in GENES_OPTION_GROUPS:
        group = parser.add_argument_group(title)
        for flags, kw in options:
            group.add_argument(*flags, **kw)
    return parser


def build_parser():
    parser = argparse.ArgumentParser(
        prog='run_midas.py snps', formatter_class=argparse.RawTextHelpFormatter,
        description="Map a metagenome to the representative genomes of its abundant species and count the four alleles\n"
                    "at every genomic site.  Stages: --build_db, --align, --pileup (the last one runs on the GPU).\n"
                    "Afterwards: merge_midas.py snps.",
        epilog="examples:\n"
               "  run_midas.py snps OUT -1 reads_1.fq.gz -2 reads_2.fq.gz\n"
               "  run_midas.py snps OUT --pileup --mapid 95 --baseq 35\n"
               "  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_midas.py snps OUT --pileup")
    parser.add_argument('program', help=argparse.SUPPRESS)
    parser.add_argument('outdir', help="sample directory (its name is the sample id)")
    parser.add_argument('--remove_temp', action='store_true', help="delete <outdir>/snps/temp when done")
    parser.add_argument('--split_length', type=int, default=8 << 20, metavar='INT',
                        help="under torchrun, contigs longer than this are dealt to the GPUs in pieces (0: never; default 8 Mb)")
    parser.add_argument('--max_batch_reads', type=int, default=0, metavar='INT',
                        help="a GPU's contigs go to the device in batches of at most this many reads (0: the built-in limit, 2^30; "
                             "nearly every job is one batch -- the tables do not depend on it)")
    parser.add_argument('--device_inflate', choices=('auto', 'on', 'off'), default='auto',
                        help="inflate the BAM's blocks on the GPU instead of with the host's threads; auto (default): for one rank "
                             "and a BAM of at least 0.5 GB whose decode fits the device's memory (twelve times the file), and for a "
                             "rank that has few CPUs; in auto a device that cannot (out of memory, HIP error) hands over to the host's "
                             "threads, with 'on' that is an error")
    parser.add_argument('--pad_rule', choices=('pysam', 'spec'), default='pysam',
                        help="what the CIGAR op P does to the query position in the pileup: pysam (default) = it advances, as in\n"
                             "get_aligned_pairs of the pysam releases MIDAS runs on; spec = nothing, the SAM specification's rule.\n"
                             "bowtie2 never writes P: the choice only matters for BAMs from elsewhere")
    for title, options in OPTION_GROUPS:
        group = parser.add_argument_group(title)
        for flags, kw in options:
            group.add_argument(*flags, **kw)
    return parser


def get_arguments(program):
    args = vars((build_genes_parser() if program == 'genes' else build_parser()).parse_args())
    if args['species_id']:
        args['species_id'] = args['species_id'].split(',')
    # the aligner stages shell out like the reference does, but to whatever is on PATH: no binaries ship here
    for tool in ('bowtie2-build', 'bowtie2', 'samtools'):
        args[tool] = utility.find_executable(tool)
    return args


def check_arguments(program, args):
    """The reference's sanity checks (scripts/run_midas.py:556-628 snps, :484-554 genes): same conditions, same exits."""
    if platform.system() not in ('Linux', 'Darwin'):
        die("Operating system '%s' not supported" % platform.system())
    if args['m1']:
        args['file_type'] = utility.auto_detect_file_type(args['m1'])
    utility.check_database(args)
    per_species = 'rep_genomes' if program == 'snps' else 'pan_genomes'
    for sp in args['species_id'] or []:
        if not os.path.isdir(os.path.join(args['db'], per_species, sp)):
            die("the specified species_id '%s' was not found in the database" % sp)
    os.makedirs(os.path.join(args['outdir'], program), exist_ok=True)
    last = 'call' if program == 'snps' else 'cov'       # the stage after alignment
    if not (args['build_db'] or args['align'] or args[last]):
        args['build_db'] = args['align'] = args[last] = True
    if not (args['species_id'] or args['species_topn'] or args['species_cov']):
        args['species_cov'] = 3.0
    temp = os.path.join(args['outdir'], program, 'temp')
    fa_name, bam_name, flag = ('genomes.fa', 'genomes.bam', '--pileup') if program == 'snps' else \
        ('pangenomes.fa', 'pangenomes.bam', '--call_genes')
    profile = os.path.join(args['outdir'], 'species', 'species_profile.txt')
    if args['build_db'] and (args['species_topn'] or args['species_cov']) and not os.path.isfile(profile):
        die("Could not find species abundance profile: %s\n"
            "--species_topn / --species_cov need the output of `run_midas.py species`; use --species_id otherwise" % profile)
    have_fa, have_bam = (os.path.isfile(os.path.join(temp, f)) for f in (fa_name, bam_name))
    if args['align'] and not args['build_db'] and not have_fa:
        die("You've specified --align, but no database has been built\nTry running with --build_db")
    if args[last] and not args['align'] and not have_bam:
        die("You've specified %s, but no alignments were found\nTry running with --align" % flag)
    if program == 'snps' and args['call'] and not args['build_db'] and not have_fa:
        die("You've specified --pileup, but no genome database was found\nTry running with --build_db")
    if args['align'] and not args['m1']:
        die("To align reads, you must specify path to input FASTA/FASTQ")
    if args['m2'] and not args['m1']:
        die("Must specify -1 and -2 if aligning paired end reads")
    if args['m2'] and args['interleaved']:
        die("Cannot specify --interleaved together with -2")
    for key in ('m1', 'm2'):
        if args[key]:
            if not os.path.isfile(args[key]):
                die("Input file does not exist: '%s'" % args[key])
            utility.check_compression(args[key])
    for key, lo, hi in (('mapid', 1, 100), ('mapq', 0, 100), ('baseq', 0, 100), ('aln_cov', 0, 1)):
        if key in args and not lo <= args[key] <= hi:
            die("%s must be between %s and %s" % (key.upper(), lo, hi))


def create_directories(program, args):
    for sub in ('', 'output', 'temp'):
        os.makedirs(os.path.join(args['outdir'], program, sub), exist_ok=True)


def open_log(program, args):
    """Rank 0 owns log.txt; the other ranks log to the null device."""
    args['log'] = open(os.path.join(args['outdir'], program, 'log.txt') if RANK == 0 else os.devnull, 'w')


def print_arguments(program, args):
    last = ('pileup', args.get('call')) if program == 'snps' else ('call_genes', args.get('cov'))
    stages = [name for name, on in (('build_db', args['build_db']), ('align', args['align']), last) if on]
    shown = [('command', ' '.join(sys.argv)), ('database', args['db']), ('output directory', args['outdir']),
             ('stages', ', '.join(stages)), ('remove temp', args['remove_temp'])]
    if args['build_db']:
        shown += [('species_id', args['species_id']), ('species_topn', args['species_topn']), ('species_cov', args['species_cov'])]
    if args['align']:
        shown += [('reads', ' '.join(x for x in (args['m1'], args['m2']) if x) + (' (interleaved)' if args['interleaved'] else '')),
                  ('bowtie2', '--%s%s' % (args['speed'], '-local' if args['mode'] == 'local' else '')),
                  ('max reads', args['max_reads'] or 'all'), ('threads', args['threads'])]
    if last[1]:
        shown += [(k, args[k]) for k in ('mapid', 'mapq', 'baseq', 'readq', 'aln_cov', 'trim') if k in args]
    text = "=== run_midas.py %s (MI355X) ===\n" % program + ''.join("%-18s %s\n" % (k + ':', v) for k, v in shown) + "===\n"
    args['log'].write(text)
    if RANK == 0:
        sys.stdout.write(text)


README = """run_midas.py snps -- files in this directory

output/<species_id>.snps.gz   one row per site of the species' representative genome, tab separated, gzip:
                                ref_id      contig id
                                ref_pos     1-based position on the contig
                                ref_allele  reference base (upper case)
                                depth       count_a + count_c + count_g + count_t
                                count_a/c/g/t  reads supporting each allele at the site
species.txt                   species whose genomes are in temp/genomes.fa
summary.txt                   per species: genome_length, covered_bases, fraction_covered, mean_coverage,
                              aligned_reads, mapped_reads
log.txt                       parameters and external commands of this run
temp/                         genomes.fa, bowtie2 index, genomes.bam (deleted by --remove_temp)

Reads counted: identity >= --mapid, mean quality >= --readq, mapping quality >= --mapq, aligned fraction
>= --aln_cov; bases counted: A/C/G/T with quality >= --baseq.  mean_coverage is over covered sites only.
Next step: merge_midas.py snps.
"""


GENES_README = """run_midas.py genes -- files in this directory

output/<species_id>.genes.gz  one row per gene of the species' pangenome, tab separated, gzip:
                                gene_id       centroid gene id
                                count_reads   reads that passed the filter on this gene
                                coverage      sum over those reads of aligned bases / gene length
                                copy_number   coverage / median coverage of the species' 15 marker genes
species.txt                   species whose pangenomes are in temp/pangenomes.fa
summary.txt                   per species: pangenome_size, covered_genes, fraction_covered, mean_coverage (over covered
                              genes), marker_coverage, aligned_reads, mapped_reads
log.txt                       parameters and external commands of this run
temp/                         pangenomes.fa, bowtie2 index, pangenomes.bam (deleted by --remove_temp)

Reads counted: identity >= --mapid, mean quality >= --readq, mapping quality >= --mapq, aligned fraction >= --aln_cov.
Next step: merge_midas.py genes.
"""


def write_readme(program, args):
    if RANK == 0:
        with open(os.path.join(args['outdir'], program, 'readme.txt'), 'w') as handle:
            handle.write(README if program == 'snps' else GENES_README)


def run_program(program, args):
    if program == 'genes':
        from midas_amd.run import genes
        genes.run_pipeline(args)
    else:
        from midas_amd.run import snps
        snps.run_pipeline(args)


if __name__ == '__main__':
    program = get_program()
    args = get_arguments(program)
    check_arguments(program, args)
    create_directories(program, args)
    open_log(program, args)
    print_arguments(program, args)
    write_readme(program, args)
    run_program(program, args)
    # Done: every output file is written and closed; the handles (the BAM's mapping, the device arena, the context) were released
    # where their owners went out of scope.  What is left is the interpreter's and the HIP runtime's own teardown.
    # MIDAS_SNPS_EXIT=fast skips it (os._exit once the log is closed) -- measured at configs[3]: no faster, the kernel then
    # reclaims what the runtime would have freed (profiles/r06_cli_stage_c4.txt) -- so the ordinary exit is the default.
    if os.environ.get('MIDAS_SNPS_TRACE'):
        import time as _t
        _t0 = _t.time()
        import gc
        gc.collect()
        sys.stderr.write("[stage] %-44s %9.3f ms\n" % ("run_pipeline returned: cycles collected", (_t.time() - _t0) * 1e3))
    if os.environ.get('MIDAS_SNPS_EXIT') == 'fast':
        args['log'].close()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
