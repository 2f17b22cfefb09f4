#!/usr/bin/env python
"""merge_midas.py -- the `snps` command (multi-sample SNP calling) on MI355X.

Keeps the reference's command line for `merge_midas.py snps` (scripts/merge_midas.py:148-281 arguments and presets,
:283-332 checks) and output layout; `species` and `genes` merges are not part of this build.
"""

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def get_program():
    if len(sys.argv) == 1 or sys.argv[1] in ['-h', '--help']:
        print('Description: merge MIDAS results across metagenomic samples')
        print('')
        print('Usage: merge_midas.py <command> [options]')
        print('')
        print('Commands:')
        print('\tsnps\t perform multi-sample SNP calling and build SNP matrix for each species (MI355X)')
        print('')
        print('Note: use merge_midas.py <command> -h to view usage for a specific command')
        quit()
    elif sys.argv[1] in ['species', 'genes']:
        sys.exit("\nError: '%s' is not part of this build (only the snps path is)\n" % sys.argv[1])
    elif sys.argv[1] != 'snps':
        sys.exit("\nError: Unrecognized command: '%s'\n" % sys.argv[1])
    return sys.argv[1]


def snps_arguments():
    parser = argparse.ArgumentParser(
        formatter_class=argparse.RawTextHelpFormatter, usage=argparse.SUPPRESS,
        description="""
Description: perform multi-sample core-genome SNP calling

Usage: merge_midas.py snps <outdir> [options]
""",
        epilog="""Examples:
1) Call SNPs for all species. Provide list of paths to sample directories:
merge_midas.py snps /path/to/outdir -i sample_1,sample_2 -t list

2) Merge results for all sites in the core genome, including those that aren't SNPs:
merge_midas.py snps /path/to/outdir -i /path/to/samples -t dir --core_sites
""")
    parser.add_argument('program', help=argparse.SUPPRESS)
    parser.add_argument('outdir', type=str, help="Directory for output files. \nA subdirectory will be created for each species_id")
    parser.add_argument('--threads', type=int, default=1, metavar='INT', help="Number of CPUs to use (1)")
    io = parser.add_argument_group('Input/Output')
    io.add_argument('-i', type=str, dest='input', required=True, help="Input to sample directories output by run_midas.py; see '-t' for details")
    io.add_argument('-t', choices=['list', 'file', 'dir'], dest='intype', required=True, metavar="INPUT_TYPE",
                    help="list: comma-separated list; dir: directory containing all samples; file: file of paths")
    io.add_argument('-d', type=str, dest='db', default=os.environ['MIDAS_DB'] if 'MIDAS_DB' in os.environ else None,
                    help="Path to reference database (default: MIDAS_DB)")
    snps = parser.add_argument_group("Presets")
    snps.add_argument('--core_snps', action='store_true', help="Same as: --snp_type bi --site_depth 1 --site_ratio 2.0 --site_prev 0.95 (default)")
    snps.add_argument('--core_sites', action='store_true', help="Same as: --snp_type any --site_depth 1 --site_ratio 2.0 --site_prev 0.95")
    snps.add_argument('--all_snps', action='store_true', help="Same as: --snp_type bi --site_prev 0.0")
    snps.add_argument('--all_sites', action='store_true', help="Same as: --snp_type any --site_prev 0.0")
    species = parser.add_argument_group("Species filters (select subset of species from INPUT)")
    species.add_argument('--min_samples', type=int, default=1, metavar='INT', help="All species with >= MIN_SAMPLES (1)")
    species.add_argument('--species_id', dest='species_id', type=str, metavar='CHAR', help="Comma-separated list of species ids")
    species.add_argument('--max_species', type=int, metavar='INT', help="Maximum number of species to call SNPs for")
    sample = parser.add_argument_group("Sample filters (select subset of samples from INPUT)")
    sample.add_argument('--sample_depth', dest='sample_depth', type=float, default=5.0, metavar='FLOAT', help="Minimum average read depth per sample (5.0)")
    sample.add_argument('--fract_cov', dest='fract_cov', type=float, default=0.4, metavar='FLOAT', help="Fraction of reference sites covered by at least 1 read (0.4)")
    sample.add_argument('--max_samples', type=int, metavar='INT', help="Maximum number of samples to process")
    sample.add_argument('--all_samples', default=False, action='store_true', help="Include all samples in output")
    snps = parser.add_argument_group("Site filters (select subset of genomic sites from INPUT)")
    snps.add_argument('--snp_type', choices=['any', 'mono', 'bi', 'tri', 'quad'], nargs='+', default=['bi'], metavar="",
                      help="mono/bi/tri/quad: keep sites with 1/2/3/4 alleles > ALLELE_FREQ; any: keep regardless")
    snps.add_argument('--allele_freq', type=float, default=0.01, metavar='FLOAT', help="Minimum frequency for calling an allele present (0.01)")
    snps.add_argument('--site_depth', type=int, default=1, metavar='INT', help="Minimum number of reads mapped to genomic site (1)")
    snps.add_argument('--site_ratio', type=float, default=2.0, metavar='FLOAT', help="Maximum ratio of site depth to genome depth (2.0)")
    snps.add_argument('--site_prev', type=float, default=0.95, metavar='FLOAT', help="Minimum fraction of samples where the site passes (0.95)")
    snps.add_argument('--max_sites', type=int, default=float('Inf'), metavar='INT', help="Maximum number of sites to include in output (use all)")
    args = vars(parser.parse_args())
    return add_snp_presets(args)


def add_snp_presets(args):
    """scripts/merge_midas.py:259-281"""
    if args['all_samples']:
        args['sample_depth'] = 0.0
        args['fract_cov'] = 0.0
    if args['all_sites']:
        args['site_prev'] = 0.0
        args['snp_type'] = ['any']
    if args['all_snps']:
        args['site_prev'] = 0.0
        args['snp_type'] = ['bi']
    if args['core_sites']:
        args['site_depth'] = 1
        args['site_ratio'] = 2.0
        args['site_prev'] = 0.95
        args['snp_type'] = ['any']
    if args['core_snps']:
        args['site_depth'] = 1
        args['site_ratio'] = 2.0
        args['site_prev'] = 0.95
        args['snp_type'] = ['bi']
    return args


def check_arguments(args):
    """scripts/merge_midas.py:283-332 (the parts that apply to snps)"""
    if not os.path.isdir(args['outdir']):
        os.makedirs(args['outdir'], exist_ok=True)
    if args['db'] is None:
        sys.exit("\nError: No reference database specified\nUse the flag -d to specify a database,\nOr set the MIDAS_DB environmental variable: export MIDAS_DB=/path/to/midas/db\n")
    if not os.path.isdir(args['db']):
        sys.exit("\nError: Specified reference database does not exist: %s\n" % args['db'])
    if args['intype'] == 'dir':
        if not os.path.isdir(args['input']):
            sys.exit("\nError: Specified input directory '%s' does not exist\n" % args['input'])
        args['indirs'] = [os.path.join(args['input'], d) for d in sorted(os.listdir(args['input']))]
    elif args['intype'] == 'file':
        if not os.path.isfile(args['input']):
            sys.exit("\nError: Specified input file '%s' does not exist\n" % args['input'])
        args['indirs'] = [line.rstrip().rstrip('/') for line in open(args['input']) if line.strip()]
    else:
        args['indirs'] = args['input'].split(',')
    for d in args['indirs']:
        if not os.path.isdir(d):
            sys.exit("\nError: Specified input directory '%s' does not exist\n" % d)
    if args['site_depth'] < 0:
        sys.exit("\nError: --site_depth must be >=0\n")
    if args['allele_freq'] <= 0.0 or args['allele_freq'] >= 0.5:
        sys.exit("\nError: --allele_freq must be > 0.0 and < 0.5\n")
    if args['site_prev'] < 0 or args['site_prev'] > 1:
        sys.exit("\nError: --site_prev must be between 0 and 1\n")
    if args['max_sites'] != float('Inf') and args['max_sites'] < 0:
        sys.exit("\nError: --max_sites must be >= 0\n")


if __name__ == '__main__':
    program = get_program()
    args = snps_arguments()
    check_arguments(args)
    from midas_amd.merge import snps
    snps.run_pipeline(args)
