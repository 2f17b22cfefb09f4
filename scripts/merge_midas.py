#!/usr/bin/env python
"""merge_midas.py snps -- multi-sample SNP calling with the per-site arithmetic on MI355X.

Drop-in for the `snps` command of the reference's scripts/merge_midas.py: same positional arguments, option names,
defaults, presets and output files (<outdir>/<species>/snps_{info,freq,depth,summary}.txt, readme.txt).  The option
semantics are the reference's (scripts/merge_midas.py:148-281); `species` and `genes` merges are not part of this
build.  Under torch.distributed.run the species are dealt to the ranks (one GPU each).
"""

import argparse
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

# preset flag -> the site filters it pins (applied in this order, later ones win, as in the reference)
PRESETS = [
    ('all_sites', dict(site_prev=0.0, snp_type=['any'])),
    ('all_snps', dict(site_prev=0.0, snp_type=['bi'])),
    ('core_sites', dict(site_depth=1, site_ratio=2.0, site_prev=0.95, snp_type=['any'])),
    ('core_snps', dict(site_depth=1, site_ratio=2.0, site_prev=0.95, snp_type=['bi'])),
]

OPTION_GROUPS = [
    ("Samples", [
        (['-i'], dict(dest='input', required=True, help="run_midas.py output directories; how to read this is set by -t")),
        (['-t'], dict(dest='intype', required=True, choices=['list', 'file', 'dir'], metavar='list|file|dir',
                      help="list: comma separated paths; file: one path per line; dir: every sub-directory")),
        (['-d'], dict(dest='db', default=os.environ.get('MIDAS_DB'), help="MIDAS reference database (default: $MIDAS_DB)")),
    ]),
    ("Presets", [
        (['--core_snps'], dict(action='store_true', help="bi-allelic sites present in >= 95 %% of the samples (the default behaviour)")),
        (['--core_sites'], dict(action='store_true', help="like --core_snps but keeps every site, variable or not")),
        (['--all_snps'], dict(action='store_true', help="bi-allelic sites, whatever their prevalence")),
        (['--all_sites'], dict(action='store_true', help="every site")),
    ]),
    ("Species", [
        (['--min_samples'], dict(type=int, default=1, metavar='INT', help="species found in at least this many samples (1)")),
        (['--species_id'], dict(metavar='ID[,ID...]', help="only these species")),
        (['--max_species'], dict(type=int, metavar='INT', help="at most this many species, most prevalent first")),
    ]),
    ("Sample filters, per species", [
        (['--sample_depth'], dict(type=float, default=5.0, metavar='FLOAT', help="minimum mean_coverage of the sample (5.0)")),
        (['--fract_cov'], dict(type=float, default=0.4, metavar='FLOAT', help="minimum fraction_covered of the sample (0.4)")),
        (['--max_samples'], dict(type=int, metavar='INT', help="use at most this many samples")),
        (['--all_samples'], dict(action='store_true', help="no sample filters (sample_depth = fract_cov = 0)")),
    ]),
    ("Site filters", [
        (['--snp_type'], dict(nargs='+', default=['bi'], choices=['any', 'mono', 'bi', 'tri', 'quad'], metavar='TYPE',
                              help="keep sites with 1/2/3/4 alleles at or above --allele_freq: mono bi tri quad, or any (bi)")),
        (['--allele_freq'], dict(type=float, default=0.01, metavar='FLOAT', help="pooled frequency at which an allele counts as present (0.01)")),
        (['--site_depth'], dict(type=int, default=1, metavar='INT', help="a sample passes at a site with at least this depth (1)")),
        (['--site_ratio'], dict(type=float, default=2.0, metavar='FLOAT', help="... and at most this depth / mean_coverage (2.0)")),
        (['--site_prev'], dict(type=float, default=0.95, metavar='FLOAT', help="keep sites where at least this fraction of samples passes (0.95)")),
        (['--max_sites'], dict(type=int, default=float('Inf'), metavar='INT', help="read only the first N sites of every table (all)")),
    ]),
]


def die(message):
    sys.exit("\nError: %s\n" % message)


def get_program():
    word = sys.argv[1] if len(sys.argv) > 1 else '-h'
    if word in ('-h', '--help'):
        print("merge_midas.py <command> [options]\n\n"
              "  snps   pool the per-sample allele counts of a species, call alleles and SNP types, write the\n"
              "         freq/depth/info matrices (site arithmetic on the MI355X); `merge_midas.py snps -h` for options\n\n"
              "species and genes merges are not part of this build.")
        sys.exit(0)
    if word in ('species', 'genes'):
        die("'%s' is not part of this build (only the snps path is)" % word)
    if word != 'snps':
        die("Unrecognized command: '%s'" % word)
    return word


def add_snp_presets(args):
    """--all_samples and the four presets overwrite the individual filters (scripts/merge_midas.py:259-281)."""
    if args['all_samples']:
        args['sample_depth'] = args['fract_cov'] = 0.0
    for flag, pinned in PRESETS:
        if args[flag]:
            args.update({k: (list(v) if isinstance(v, list) else v) for k, v in pinned.items()})
    return args


def snps_arguments():
    parser = argparse.ArgumentParser(
        prog='merge_midas.py snps', formatter_class=argparse.RawTextHelpFormatter,
        description="Multi-sample SNP calling over the core genome of each species.",
        epilog="examples:\n"
               "  merge_midas.py snps OUT -i sample_1,sample_2 -t list\n"
               "  merge_midas.py snps OUT -i /path/to/samples -t dir --core_sites")
    parser.add_argument('program', help=argparse.SUPPRESS)
    parser.add_argument('outdir', help="output directory; one sub-directory per species")
    parser.add_argument('--threads', type=int, default=1, metavar='INT', help="CPU threads for reading and writing tables (1)")
    for title, options in OPTION_GROUPS:
        group = parser.add_argument_group(title)
        for flags, kw in options:
            group.add_argument(*flags, **kw)
    return add_snp_presets(vars(parser.parse_args()))


def check_arguments(args):
    """scripts/merge_midas.py:283-332, the parts that apply to snps."""
    os.makedirs(args['outdir'], exist_ok=True)
    if args['db'] is None:
        die("No reference database specified\nUse the flag -d to specify a database,\n"
            "Or set the MIDAS_DB environmental variable: export MIDAS_DB=/path/to/midas/db")
    if not os.path.isdir(args['db']):
        die("Specified reference database does not exist: %s" % args['db'])
    kind, source = args['intype'], args['input']
    if kind == 'dir':
        if not os.path.isdir(source):
            die("Specified input directory '%s' does not exist" % source)
        # every entry is taken, as in the reference (a stray README or tarball among the samples is dropped later,
        # when it turns out to have no snps/summary.txt); sorted, so that the sample columns do not depend on the
        # file system's listing order
        args['indirs'] = [os.path.join(source, d) for d in sorted(os.listdir(source))]
    else:
        if kind == 'file':
            if not os.path.isfile(source):
                die("Specified input file '%s' does not exist" % source)
            with open(source) as handle:
                args['indirs'] = [line.strip().rstrip('/') for line in handle if line.strip()]
        else:
            args['indirs'] = source.split(',')
        for d in args['indirs']:      # only listed directories are checked (scripts/merge_midas.py:320-331)
            if not os.path.isdir(d):
                die("Specified input directory '%s' does not exist" % d)
    if args['site_depth'] < 0:
        die("--site_depth must be >=0")
    for name in ('allele_freq', 'fract_cov', 'site_prev'):      # scripts/merge_midas.py:291-293
        if args.get(name) and not 0.0 <= args[name] <= 1.0:
            die("--%s must be between 0.0 and 1.0" % name)
    if args['max_sites'] != float('Inf') and args['max_sites'] < 0:
        die("--max_sites must be >= 0")


if __name__ == '__main__':
    get_program()
    args = snps_arguments()
    check_arguments(args)
    from midas_amd.merge import snps
    snps.run_pipeline(args)
