"""merge_midas.py snps: the oracle (restated GenomicSite arithmetic) against hand-derived cases."""
from oracle import merge_oracle as mo

NOGENES = {'list': [], 'index': 0}


def chain(counts, allele_freq=0.01, mean=None, min_depth=1, max_ratio=2.0):
    pooled = mo.pooled_counts(counts)
    major, minor, snp_type = mo.call_alleles(pooled, allele_freq)
    mafs, depths = mo.per_sample(counts, major, minor)
    cs, prev = mo.prevalence(mean or [1.0] * len(counts), depths, min_depth, max_ratio)
    return dict(pooled=pooled, major=major, minor=minor, snp_type=snp_type, mafs=mafs, depths=depths, cs=cs, prev=prev)


def test_tie_breaks_in_acgt_order():
    # pooled [5,0,0,5]: stable descending sort -> major A, minor T (verified in SURVEY 8f)
    assert mo.call_alleles([5, 0, 0, 5], 0.01) == (0, 3, 'bi')
    assert mo.call_alleles([0, 3, 3, 3], 0.01) == (1, 2, 'tri')
    assert mo.call_alleles([2, 2, 2, 2], 0.01) == (0, 1, 'quad')


def test_snp_type_thresholds_scan_from_the_rarest():
    # depth 1000: A 985, C 10, G 4, T 1 -> .985 .010 .004 .001; allele_freq 0.01 -> 'bi' (>= is inclusive)
    assert mo.call_alleles([985, 10, 4, 1], 0.01) == (0, 1, 'bi')
    assert mo.call_alleles([985, 10, 4, 1], 0.001)[2] == 'quad'
    assert mo.call_alleles([985, 10, 4, 1], 0.011)[2] == 'mono'
    assert mo.call_alleles([985, 10, 4, 1], 0.99)[2] is None      # nothing reaches the threshold


def test_codon_table_is_the_standard_code():
    assert len(mo.CODONTABLE) == 64
    for codon, aa in dict(ATG='M', TAA='_', TAG='_', TGA='_', TGG='W', GCA='A', AGA='R', CTG='L', ATA='I', TTT='F',
                          GGG='G', CAT='H', AAC='N', GAA='E').items():
        assert mo.CODONTABLE[codon] == aa


def test_zero_depth_site():
    r = chain([[0, 0, 0, 0], [0, 0, 0, 0]], mean=[10.0, 10.0])
    assert r['major'] is None and r['snp_type'] is None
    assert r['mafs'] == [0.0, 0.0] and r['depths'] == [0, 0]
    assert r['cs'] == 0 and r['prev'] == 0.0
    assert mo.flag_reason(r['prev'], r['snp_type'], 0.0, ['bi']) == 'snp_type'     # None is not in ['bi']
    assert mo.flag_reason(r['prev'], r['snp_type'], 0.0, ['any']) is None


def test_per_sample_depth_counts_major_plus_minor_only():
    # pooled: A 12, C 6, G 1 -> major A, minor C; sample 2 has a read on G that does not count toward its depth
    r = chain([[10, 2, 0, 0], [2, 4, 1, 0]], mean=[6.0, 2.0])
    assert (r['major'], r['minor'], r['snp_type']) == (0, 1, 'tri')               # G's 1/19 >= 0.01
    assert r['depths'] == [12, 6]
    assert r['mafs'] == [2 / 12.0, 4 / 6.0]
    # 12/6 = 2.0 is NOT > 2.0 (pass); 6/2 = 3.0 > 2.0 (fail)
    assert r['cs'] == 1 and r['prev'] == 0.5
    assert mo.flag_reason(0.5, 'tri', 0.5, ['tri']) is None
    assert mo.flag_reason(0.5, 'tri', 0.51, ['any']) == 'min_prev'
    assert mo.flag_reason(0.5, 'tri', 0.5, ['bi', 'mono']) == 'snp_type'


def test_monomorphic_site_rows():
    args = dict(allele_freq=0.01, site_depth=1, site_ratio=2.0, site_prev=0.0, snp_type=['any'])
    rows = mo.site_rows(1, ["contig_1|7|A", "0,0,7,0", "0,0,0,0"], [3.0, 3.0], args, dict(NOGENES))
    # 7/3 > 2 fails, 0 < 1 fails -> count_samples 0 ; minor NA ; mafs 0
    assert rows[0] == "1\tcontig_1\t7\tA\tG\tNA\t0\t0\t0\t7\t0\tIGR\tNA\tmono\tNA\tNA\n"
    assert rows[1] == "1\t0\t0\n" and rows[2] == "1\t7\t0\n"
    args['snp_type'] = ['bi']
    assert mo.site_rows(1, ["contig_1|7|A", "0,0,7,0", "0,0,0,0"], [3.0, 3.0], args, dict(NOGENES)) is None


def test_ref_id_may_contain_pipes():
    assert mo.parse_site(["a|b|c|12|N", "1,2,3,4"]) == ("a|b|c", 12, "N", [[1, 2, 3, 4]])


def test_annotation_of_a_plus_strand_codon():
    # gene at 4..12 (+): ATGGCATAA ; site 8 is codon 2 ('GCA') position 1 -> GAA GCA GGA GTA = E,A,G,V -> 1D
    gene = dict(gene_id='g1', scaffold_id='contig_1', start=4, end=12, strand='+', gene_type='CDS', seq='ATGGCATAA')
    genes = {'list': [gene], 'index': 0}
    assert mo.annotate_site('contig_1', 2, genes) == ('IGR', None, None, None)           # upstream
    assert mo.annotate_site('contig_1', 8, genes) == ('CDS', 'g1', '1D', 'E,A,G,V')
    assert mo.annotate_site('contig_1', 9, genes) == ('CDS', 'g1', '4D', 'A,A,A,A')      # GCN is four-fold
    assert mo.annotate_site('contig_1', 13, genes) == ('IGR', None, None, None) and genes['index'] == 1


def test_annotation_of_a_minus_strand_codon():
    # contig ...TTATGCCAT... ; gene (-) over 1..9 has seq revcomp('TTATGCCAT') = 'ATGGCATAA'
    contig = 'TTATGCCAT'
    gene = dict(gene_id='g2', scaffold_id='c', start=1, end=9, strand='-', gene_type='CDS')
    gene['seq'] = mo.gene_seq(gene, contig)
    assert gene['seq'] == 'ATGGCATAA'
    genes = {'list': [gene], 'index': 0}
    # ref_pos 5 -> gene_pos 4 -> codon 'GCA' position 1; reference-strand alleles A,C,G,T complement to T,G,C,A:
    # GTA GGA GCA GAA = V,G,A,E
    assert mo.annotate_site('c', 5, genes) == ('CDS', 'g2', '1D', 'V,G,A,E')
    # a gene whose length is not a multiple of 3, or a codon with an N, gets no site_type
    bad = dict(gene_id='g3', scaffold_id='c', start=1, end=8, strand='+', gene_type='CDS', seq='TTATGCCA')
    assert mo.annotate_site('c', 5, {'list': [bad], 'index': 0}) == ('CDS', 'g3', None, None)
    nn = dict(gene_id='g4', scaffold_id='c', start=1, end=9, strand='+', gene_type='CDS', seq='TTANGCCAT')
    assert mo.annotate_site('c', 5, {'list': [nn], 'index': 0}) == ('CDS', 'g4', None, None)


def test_three_significant_digit_formatting():
    args = dict(allele_freq=0.000001, site_depth=0, site_ratio=1e9, site_prev=0.0, snp_type=['any'])
    rows = mo.site_rows(1, ["c|1|A", "2,1,0,0", "100000,1,0,0"], [1.0, 1.0], args, dict(NOGENES))
    assert rows[1] == "1\t0.333\t1e-05\n"


def test_merge_species_numbers_sites_by_table_row():
    args = dict(allele_freq=0.01, site_depth=1, site_ratio=2.0, site_prev=0.95, snp_type=['bi'])
    keys = ["c|1|A", "c|2|C", "c|3|G"]
    s1 = ["5,0,0,0", "3,3,0,0", "0,0,0,0"]
    s2 = ["6,0,0,0", "2,4,0,0", "0,0,1,1"]
    info, freq, depth = mo.merge_species(keys, [s1, s2], [5.0, 5.0], args, dict(NOGENES))
    # row 1 mono (dropped), row 2 bi with both samples passing, row 3: sample 1 depth 0 -> prevalence 0.5 (dropped)
    assert [l.split('\t')[0] for l in info] == ['2']
    assert info[0] == "2\tc\t2\tC\tC\tA\t2\t5\t7\t0\t0\tIGR\tNA\tbi\tNA\tNA\n"
    assert freq == ["2\t0.5\t0.333\n"] and depth == ["2\t6\t6\n"]
