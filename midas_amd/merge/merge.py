"""Which (species, sample) pairs a merge works on -- the selection rules of /root/reference/midas/merge/merge.py,
restricted to what `merge_midas.py snps` needs.  Pure host logic, no device work.

Rules kept (reference line numbers in brackets):
  * a sample directory counts only if <dir>/snps/summary.txt exists [59-86];
  * a pair is dropped when --species_id excludes it, when the species already has --max_samples samples, when the
    sample's mean_coverage < --sample_depth or its fraction_covered < --fract_cov [104-119];
  * species are ranked by number of retained samples (stable), those under --min_samples are dropped and at most
    --max_species are kept [121-156];
  * species_info.txt must carry species_id + rep_genome and genome_info.txt genome_id [10-16, 88-102].
"""

import csv
import os

SUMMARY_FIELDS = ('genome_length', 'covered_bases', 'fraction_covered', 'mean_coverage', 'aligned_reads',
                  'mapped_reads')


def _keyed_table(path, key):
    with open(path) as handle:
        return {row[key]: row for row in csv.DictReader(handle, delimiter='\t')}


def read_species_info(db):
    return _keyed_table(os.path.join(db, 'species_info.txt'), 'species_id')


def read_genome_info(db):
    return _keyed_table(os.path.join(db, 'genome_info.txt'), 'genome_id')


class Sample:
    """One run_midas.py output directory; .info maps species_id -> its summary.txt row (None if no summary)."""

    def __init__(self, dir, data_type):
        self.dir = dir
        self.id = os.path.basename(dir)
        summary = os.path.join(dir, data_type, 'summary.txt')
        self.info = _keyed_table(summary, 'species_id') if os.path.isfile(summary) else None


class Species:
    """A species with the samples that passed the pair filters, in input order."""

    def __init__(self, id, species_info, genome_info):
        self.id = id
        self.info = species_info[id]                       # KeyError for an unknown species, as in the reference
        self.genome_info = genome_info[self.info['rep_genome']]
        self.samples = []
        self.sample_depth = []

    def fetch_sample_depth(self):
        self.sample_depth = [float(s.info[self.id]['mean_coverage']) for s in self.samples]

    def write_sample_info(self, dtype, outdir):
        """<outdir>/<species>/<dtype>_summary.txt: the per-sample summary rows, copied through as text."""
        with open(os.path.join(outdir, self.id, '%s_summary.txt' % dtype), 'w') as out:
            out.write('\t'.join(('sample_id',) + SUMMARY_FIELDS) + '\n')
            for s in self.samples:
                row = s.info[self.id]
                out.write('\t'.join([s.id] + [str(row[f]) for f in SUMMARY_FIELDS]) + '\n')


def init_samples(indirs, data_type):
    return [s for s in (Sample(d, data_type) for d in indirs) if s.info is not None]


def filter_sample_species(sample, species, species_id, args, dtype):
    """True when the pair must be dropped."""
    row = sample.info[species_id]
    wanted = args['species_id'].split(',') if args['species_id'] else None
    if wanted is not None and species_id not in wanted:
        return True
    if args['max_samples'] and species_id in species and len(species[species_id].samples) >= args['max_samples']:
        return True
    if float(row['mean_coverage']) < args['sample_depth']:
        return True
    return dtype == 'snps' and float(row['fraction_covered']) < args['fract_cov']


def init_species(samples, args, dtype):
    species = {}
    species_info, genome_info = read_species_info(args['db']), read_genome_info(args['db'])
    for sample in samples:
        for species_id in sample.info:
            sp = species.get(species_id)
            if sp is None:
                sp = species[species_id] = Species(species_id, species_info, genome_info)
            if not filter_sample_species(sample, species, species_id, args, dtype):
                sp.samples.append(sample)
    return list(species.values())


def sort_species(species):
    return sorted(species, key=lambda sp: len(sp.samples), reverse=True)    # stable, like the reference's sort


def filter_species(species, args):
    keep = []
    for sp in sort_species(species):
        sp.nsamples = len(sp.samples)
        if sp.nsamples < int(args['min_samples']) or (args['max_species'] and len(keep) >= args['max_species']):
            continue
        sp.fetch_sample_depth()
        sp.outdir = args['outdir'] + '/' + sp.id
        os.makedirs(sp.outdir, exist_ok=True)
        keep.append(sp)
    return keep


def select_species(args, dtype):
    return filter_species(init_species(init_samples(args['indirs'], dtype), args, dtype), args)


def species_for_rank(species_list, rank, world_size):
    """Species are independent units of a merge: rank r of N takes species r, r + N, ... of the (prevalence-sorted)
    list and writes their output directories itself; no data-path collective is needed."""
    return [sp for k, sp in enumerate(species_list) if k % world_size == rank]
