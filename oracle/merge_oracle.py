"""CPU oracle for `merge_midas.py snps` (SURVEY.md 8f "next" #1): the per-site cross-sample arithmetic.

TEST INFRASTRUCTURE ONLY -- never imported from midas_amd/ or scripts/.

PINNED AGAINST THE REFERENCE ITSELF: the reference's tests hold no values for this path (test/test_midas.py:116-120
asserts an exit code only), but its arithmetic is plain Python, so tests/golden/make_merge_vectors.py executes the
reference's own GenomicSite class and codon helpers (in the build container, from /root/reference) on seeded inputs
and commits inputs + outputs as tests/golden/merge_vectors.json (4 800 site evaluations: every snp_type / flag /
annotation branch, floats compared bit for bit, the three output texts byte for byte); tests/test_merge_golden.py
holds this oracle, the product's annotation code and -- on the GPU box -- midas_merge_sites to those vectors.  Hand
derived cases (tests/test_merge_oracle.py) cover the corner semantics by reasoning.  The restatement below keeps the
reference's operations in the reference's order (float division, stable `sorted(..., reverse=True)`, '{0:.3g}'); each
function cites the lines of /root/reference/midas/merge/snps.py or midas/utility.py it follows.
"""

ALLELES = 'ACGT'
SNP_TYPES_FROM_RAREST = ('quad', 'tri', 'bi', 'mono')


def parse_site(values):
    """GenomicSite.__init__ (merge/snps.py:15-36): 'ref_id|ref_pos|ref_allele' + one 'a,c,g,t' string per sample."""
    ref_id, ref_pos, ref_allele = values[0].rsplit('|', 2)
    counts = [[int(x) for x in v.split(',')] for v in values[1:]]
    return ref_id, int(ref_pos), ref_allele, counts


def pooled_counts(counts):
    """compute_pooled_counts (:38-43)"""
    return [sum(c[k] for c in counts) for k in range(4)]


def call_alleles(pooled, snp_freq):
    """call_alleles (:49-76) -> (major_index|None, minor_index|None, snp_type|None).

    The reference sorts (allele, freq) pairs by freq, descending, with Python's stable sort: equal frequencies keep
    A,C,G,T order.  The SNP type is found by walking from the rarest allele up until one reaches snp_freq (>=).
    """
    depth = sum(pooled)
    if depth == 0:
        return None, None, None
    freqs = [float(c) / depth for c in pooled]
    order = sorted(range(4), key=lambda k: freqs[k], reverse=True)    # stable, same as sorting the zipped pairs
    major = order[0] if freqs[order[0]] > 0 else None
    minor = order[1] if freqs[order[1]] > 0 else None
    snp_type = None
    for name, k in zip(SNP_TYPES_FROM_RAREST, order[::-1]):
        if freqs[k] >= snp_freq:
            snp_type = name
            break
    return major, minor, snp_type


def per_sample(counts, major, minor):
    """compute_per_sample_mafs (:78-91) -> (mafs, depths); depth counts the major and minor allele only."""
    if major is None:
        return [0.0] * len(counts), [0] * len(counts)
    if minor is None:
        return [0.0] * len(counts), [c[major] for c in counts]
    mafs, depths = [], []
    for c in counts:
        d = c[major] + c[minor]
        mafs.append(float(c[minor]) / d if d > 0 else 0.0)
        depths.append(d)
    return mafs, depths


def prevalence(mean_depths, depths, min_depth, max_ratio):
    """compute_prevalence (:93-104) -> (count_samples, prevalence).  ZeroDivisionError when a sample's
    mean_coverage is 0 and the site's depth passes the first test, as in the reference."""
    ok = 0
    for mean_depth, d in zip(mean_depths, depths):
        if d < min_depth:
            continue
        if d / mean_depth > max_ratio:
            continue
        ok += 1
    return ok, ok / float(len(depths))


def flag_reason(prev, snp_type, min_prev, snp_types):
    """flag (:106-114) -> None (keep) | 'min_prev' | 'snp_type'"""
    if prev < min_prev:
        return 'min_prev'
    if 'any' not in snp_types and snp_type not in snp_types:
        return 'snp_type'
    return None


# ---- annotation: merge/snps.py:116-173 + utility.py:244-332 ----------------------------------------
_COMP = {'A': 'T', 'T': 'A', 'G': 'C', 'C': 'G'}
_AA = 'FFLLSSSSYY__CC_WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG'     # standard code, TCAG order, stop = '_'
CODONTABLE = {a + b + c: _AA[16 * i + 4 * j + k] for i, a in enumerate('TCAG') for j, b in enumerate('TCAG')
              for k, c in enumerate('TCAG')}


def rev_comp(seq):
    """utility.py:303-305 (bases outside ACGT pass through)"""
    return ''.join(_COMP.get(b, b) for b in seq[::-1])


def gene_seq(gene, contig_seq):
    """utility.py:288-294"""
    s = contig_seq[gene['start'] - 1:gene['end']]
    return rev_comp(s) if gene['strand'] == '-' else s


def sort_genes(genes):
    """utility.py:264-268: by (scaffold_id, start, -end)"""
    return {'list': sorted(genes, key=lambda g: (g['scaffold_id'], g['start'], -g['end'])), 'index': 0}


def annotate_site(ref_id, ref_pos, genes):
    """annotate + fetch_ref_codon (:116-173) -> (locus_type, gene_id, site_type, amino_acids); advances the cursor
    genes['index'] past genes that end before the site (sites arrive in table order)."""
    lst = genes['list']
    while genes['index'] < len(lst):
        g = lst[genes['index']]
        if ref_id < g['scaffold_id'] or (ref_id == g['scaffold_id'] and ref_pos < g['start']):
            break
        if ref_id > g['scaffold_id'] or (ref_id == g['scaffold_id'] and ref_pos > g['end']):
            genes['index'] += 1
            continue
        if g['gene_type'] != 'CDS' or len(g['seq']) % 3 != 0:
            return g['gene_type'], g['gene_id'], None, None
        gpos = ref_pos - g['start'] if g['strand'] == '+' else g['end'] - ref_pos
        cpos = gpos % 3
        codon = g['seq'][gpos - cpos:gpos - cpos + 3]
        if not all(b in 'ATCG' for b in codon):
            return g['gene_type'], g['gene_id'], None, None
        aas = []
        for allele in ALLELES:
            b = allele if g['strand'] == '+' else _COMP[allele]
            aas.append(CODONTABLE[codon[:cpos] + b + codon[cpos + 1:]])
        return g['gene_type'], g['gene_id'], '%sD' % (4 - len(set(aas)) + 1), ','.join(aas)
    return 'IGR', None, None, None


def na(x):
    return 'NA' if x is None else x


def format_rows(site_id, ref_id, ref_pos, ref_allele, major, minor, count_samples, pooled, annot, snp_type, mafs,
                depths):
    """write (:176-201): the info / freq / depth lines of one site"""
    locus_type, gene_id, site_type, amino_acids = annot
    info = [site_id, ref_id, str(ref_pos), ref_allele, None if major is None else ALLELES[major],
            None if minor is None else ALLELES[minor], str(count_samples)] + [str(c) for c in pooled] + \
           [locus_type, gene_id, snp_type, site_type, amino_acids]
    return ('\t'.join(na(x) for x in info) + '\n',
            site_id + '\t' + '\t'.join('{0:.3g}'.format(f) for f in mafs) + '\n',
            site_id + '\t' + '\t'.join(str(d) for d in depths) + '\n')


def site_rows(site_id, values, mean_depths, args, genes):
    """One site through the whole chain; returns None when the site is flagged (build_sharded_tables :343-361)."""
    ref_id, ref_pos, ref_allele, counts = parse_site(values)
    pooled = pooled_counts(counts)
    major, minor, snp_type = call_alleles(pooled, args['allele_freq'])
    mafs, depths = per_sample(counts, major, minor)
    count_samples, prev = prevalence(mean_depths, depths, args['site_depth'], args['site_ratio'])
    if flag_reason(prev, snp_type, args['site_prev'], args['snp_type']) is not None:
        return None
    annot = annotate_site(ref_id, ref_pos, genes)
    return format_rows(str(site_id), ref_id, ref_pos, ref_allele, major, minor, count_samples, pooled, annot,
                       snp_type, mafs, depths)


def merge_species(site_keys, sample_counts, mean_depths, args, genes):
    """build_sharded_tables (:324-364) over in-memory inputs.

    site_keys:     ['ref_id|ref_pos|ref_allele', ...] of the first sample's table
    sample_counts: [per sample][per site] 'A,C,G,T' strings
    -> (info_lines, freq_lines, depth_lines) of the unflagged sites; site ids are 1-based table row numbers.
    """
    info, freq, depth = [], [], []
    genes = {'list': genes['list'], 'index': 0}
    for n, key in enumerate(site_keys):
        rows = site_rows(n + 1, [key] + [sc[n] for sc in sample_counts], mean_depths, args, genes)
        if rows is not None:
            info.append(rows[0]); freq.append(rows[1]); depth.append(rows[2])
    return info, freq, depth
