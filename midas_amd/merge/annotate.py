"""Gene annotation of the sites a merge keeps -- host side of `merge_midas.py snps`.

Behaviour follows /root/reference/midas/merge/snps.py:116-173 (GenomicSite.annotate / fetch_ref_codon) and
midas/utility.py:244-332 (read_genes, get_gene_seq, translate, index_replace): sites arrive in table order, genes are
sorted by (scaffold_id, start, -end), a cursor skips genes that end before the site, and the FIRST gene that
contains the site wins.  Only surviving sites come through here (a few per cent of the table for the default
--core_snps preset), so this stays on the host.
"""

import os
import sys

from midas_amd import fasta, utility

_PAIR = {'A': 'T', 'T': 'A', 'G': 'C', 'C': 'G'}
_ORDER = 'TCAG'
_AMINO = 'FFLLSSSSYY__CC_WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG'    # standard code; '_' = stop, as the reference prints it
CODONTABLE = {x + y + z: _AMINO[16 * i + 4 * j + k]
              for i, x in enumerate(_ORDER) for j, y in enumerate(_ORDER) for k, z in enumerate(_ORDER)}


_COMPLEMENT = str.maketrans(_PAIR)      # A<->T, C<->G; every other character stays what it is


def rev_comp(seq):
    return seq[::-1].translate(_COMPLEMENT)


def _db_file(db, species_id, name):
    base = '%s/rep_genomes/%s/%s' % (db, species_id, name)
    for path in (base, base + '.gz'):
        if os.path.exists(path):
            return path
    sys.exit("\nError: rep genome for %s not found\n" % species_id)


def read_genome(db, species_id):
    """{contig id: upper-cased sequence} (utility.py:272-286)"""
    with utility.iopen(_db_file(db, species_id, 'genome.fna')) as handle:
        return {rid: seq.upper() for rid, seq in fasta.parse(handle)}


def parse_file(inpath):
    """Rows of a headed tab-delimited file as dicts; rows whose width differs from the header's are skipped
    (utility.py:208-216)."""
    with utility.iopen(inpath) as handle:
        fields = next(handle).rstrip('\n').split('\t')
        for line in handle:
            values = line.rstrip('\n').split('\t')
            if len(values) == len(fields):
                yield dict(zip(fields, values))


class GeneCursor:
    """The sorted CDS genes of one representative genome plus the reference's forward-only cursor."""

    def __init__(self, genes):
        self.genes = sorted(genes, key=lambda g: (g['scaffold_id'], g['start'], -g['end']))
        self.index = 0

    @classmethod
    def from_db(cls, species_id, db):
        """utility.py:244-270: genome.features rows with gene_type CDS (or no gene_type column at all)."""
        genome = read_genome(db, species_id)
        genes = []
        for g in parse_file(_db_file(db, species_id, 'genome.features')):
            if 'gene_type' in g and g['gene_type'] != 'CDS':
                continue
            g['start'], g['end'] = int(g['start']), int(g['end'])
            seq = genome[g['scaffold_id']][g['start'] - 1:g['end']]
            g['seq'] = rev_comp(seq) if g['strand'] == '-' else seq
            genes.append(g)
        return cls(genes)

    def lookup(self, ref_id, ref_pos):
        """-> (locus_type, gene_id, site_type, amino_acids); None where the reference prints NA."""
        genes = self.genes
        while self.index < len(genes):
            g = genes[self.index]
            sid = g['scaffold_id']
            if ref_id < sid or (ref_id == sid and ref_pos < g['start']):
                break                                        # before the next gene: intergenic
            if ref_id > sid or (ref_id == sid and ref_pos > g['end']):
                self.index += 1                              # this gene is behind us for good
                continue
            kind = g['gene_type']                            # KeyError without the column, as in the reference
            if kind != 'CDS' or len(g['seq']) % 3:
                return kind, g['gene_id'], None, None
            plus = g['strand'] == '+'
            gpos = ref_pos - g['start'] if plus else g['end'] - ref_pos
            cpos = gpos % 3
            codon = g['seq'][gpos - cpos:gpos - cpos + 3]
            if any(b not in _PAIR for b in codon):
                return kind, g['gene_id'], None, None
            aas = [CODONTABLE[codon[:cpos] + (a if plus else _PAIR[a]) + codon[cpos + 1:]] for a in 'ACGT']
            return kind, g['gene_id'], '%dD' % (5 - len(set(aas))), ','.join(aas)
        return 'IGR', None, None, None
