"""merge_midas.py snps, host side (no GPU): the native table reader, species/sample selection, annotation and CLI presets."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from midas_amd import abi, synth
from midas_amd.merge import annotate, merge
from oracle import merge_oracle as mo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("merge"))
    return synth.make_merge_dataset(root, n_samples=3, n_sites=4000, seed=5)


def test_table_reader_matches_what_was_written(dataset):
    path = os.path.join(dataset['samples'][1], 'snps', 'output', 'sp1.snps.gz')
    counts, keys, off = abi.read_snps_table(path)
    assert counts.shape == (4000, 4) and counts.dtype == np.uint32
    assert np.array_equal(counts, dataset['counts'][1].astype(np.uint32))
    got = [bytes(keys[off[i]:off[i + 1]]).decode() for i in range(len(off) - 1)]
    assert got == dataset['keys']
    # max_rows (args['max_sites']) and keyless reads
    c2, k2, o2 = abi.read_snps_table(path, 17, False)
    assert c2.shape == (17, 4) and k2 is None and o2 is None and np.array_equal(c2, counts[:17])
    c0, _, o0 = abi.read_snps_table(path, 0, True)
    assert c0.shape == (0, 4) and list(o0) == [0]


def test_table_reader_reads_tables_written_by_the_pileup_stage(tmp_path):
    # multi-member gzip as midas_snps_write_rows produces it, 16384 rows per member
    n = 70000
    rng = np.random.default_rng(1)
    allele = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n)
    counts = rng.integers(0, 50, (n, 4)).astype(np.uint32)
    path = str(tmp_path / "x.snps.gz")
    abi.write_rows(path, False, "ctg|with|pipes", allele, counts)
    got, keys, off = abi.read_snps_table(path)
    assert np.array_equal(got, counts)
    assert bytes(keys[off[0]:off[1]]).decode() == "ctg|with|pipes|1|%s" % chr(allele[0])
    assert bytes(keys[off[n - 1]:off[n]]).decode() == "ctg|with|pipes|%d|%s" % (n, chr(allele[n - 1]))


def test_table_reader_rejects_malformed_rows(tmp_path):
    p = str(tmp_path / "bad.snps.gz")
    with gzip.open(p, "wt") as h:
        h.write("ref_id\tref_pos\tref_allele\tdepth\tcount_a\tcount_c\tcount_g\tcount_t\n")
        h.write("c\t1\tA\t3\t1\t1\t1\t0\n")
        h.write("c\t2\tA\t3\t1\tx\t1\t0\n")
    with pytest.raises(abi.MidasSnpsError) as e:
        abi.read_snps_table(p)
    assert "row 2" in e.value.message
    with pytest.raises(abi.MidasSnpsError):
        abi.read_snps_table(str(tmp_path / "missing.snps.gz"))


def base_args(dataset, outdir, **kw):
    a = dict(outdir=outdir, db=dataset['db'], indirs=list(dataset['samples']), species_id=None, max_samples=None,
             sample_depth=5.0, fract_cov=0.4, min_samples=1, max_species=None, threads=1, max_sites=float('Inf'),
             **abi.DEFAULT_MERGE_ARGS)
    a.update(kw)
    return a


def test_select_species_applies_the_pair_filters(dataset, tmp_path):
    sp = merge.select_species(base_args(dataset, str(tmp_path)), 'snps')
    assert [s.id for s in sp] == ['sp1'] and [x.id for x in sp[0].samples] == ['sample_1', 'sample_2', 'sample_3']
    assert len(sp[0].sample_depth) == 3 and all(d > 5 for d in sp[0].sample_depth)
    assert os.path.isdir(str(tmp_path / 'sp1'))
    assert merge.select_species(base_args(dataset, str(tmp_path), sample_depth=1e6), 'snps') == []
    assert merge.select_species(base_args(dataset, str(tmp_path), species_id='other'), 'snps') == []
    assert merge.select_species(base_args(dataset, str(tmp_path), min_samples=4), 'snps') == []
    two = merge.select_species(base_args(dataset, str(tmp_path), max_samples=2), 'snps')
    assert [x.id for x in two[0].samples] == ['sample_1', 'sample_2']
    # a directory without snps/summary.txt is not a sample
    extra = base_args(dataset, str(tmp_path))
    extra['indirs'] = extra['indirs'] + [str(tmp_path)]
    assert len(merge.select_species(extra, 'snps')[0].samples) == 3
    sp[0].write_sample_info('snps', str(tmp_path))
    lines = open(str(tmp_path / 'sp1' / 'snps_summary.txt')).read().splitlines()
    assert lines[0].split('\t') == ['sample_id', 'genome_length', 'covered_bases', 'fraction_covered', 'mean_coverage',
                                    'aligned_reads', 'mapped_reads']
    assert lines[1].split('\t')[0] == 'sample_1' and len(lines) == 4


def oracle_genes(dataset):
    """The oracle's gene list built independently of midas_amd.merge.annotate (plain parsing of the DB files)."""
    genome = dict(zip(dataset['contig_ids'], dataset['contig_seqs']))
    genes = []
    with open(os.path.join(dataset['db'], 'rep_genomes', 'sp1', 'genome.features')) as h:
        fields = next(h).rstrip('\n').split('\t')
        for line in h:
            g = dict(zip(fields, line.rstrip('\n').split('\t')))
            if g['gene_type'] != 'CDS':
                continue
            g['start'], g['end'] = int(g['start']), int(g['end'])
            g['seq'] = mo.gene_seq(g, genome[g['scaffold_id']])
            genes.append(g)
    return mo.sort_genes(genes)


def test_gene_cursor_agrees_with_the_oracle_on_every_site(dataset):
    cur = annotate.GeneCursor.from_db('sp1', dataset['db'])
    genes = oracle_genes(dataset)
    assert len(cur.genes) == len(genes['list']) > 5
    kinds = set()
    for key in dataset['keys'][::3]:
        ref_id, pos, _ = key.rsplit('|', 2)
        a = cur.lookup(ref_id, int(pos))
        assert a == mo.annotate_site(ref_id, int(pos), genes)
        kinds.add((a[0], a[2]))
    assert ('IGR', None) in kinds and ('CDS', '1D') in kinds and ('CDS', '4D') in kinds and ('CDS', None) in kinds


def run_cli(*argv):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'merge_midas.py')] + list(argv),
                          capture_output=True, text=True, cwd=ROOT)


def test_cli_usage_and_argument_checks(dataset, tmp_path):
    r = run_cli()
    assert r.returncode == 0 and 'snps' in r.stdout
    assert run_cli('genes', str(tmp_path)).returncode != 0
    r = run_cli('snps', str(tmp_path / 'o'), '-i', str(tmp_path / 'nope'), '-t', 'dir', '-d', dataset['db'])
    assert r.returncode != 0 and 'does not exist' in r.stderr
    # the reference accepts any frequency in [0, 1] (scripts/merge_midas.py:291-293): 1.7 is an error, 0.7 is not
    r = run_cli('snps', str(tmp_path / 'o'), '-i', ','.join(dataset['samples']), '-t', 'list', '-d', dataset['db'],
                '--allele_freq', '1.7')
    assert r.returncode != 0 and '--allele_freq' in r.stderr
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    try:
        import merge_midas
    finally:
        sys.path.pop(0)
    base = dict(outdir=str(tmp_path / 'o2'), db=dataset['db'], site_depth=1, max_sites=float('Inf'), site_prev=0.95, fract_cov=0.4)
    merge_midas.check_arguments(dict(base, intype='list', input=','.join(dataset['samples']), allele_freq=0.7))
    merge_midas.check_arguments(dict(base, intype='list', input=','.join(dataset['samples']), allele_freq=0.0))
    # -t dir takes every entry of the directory, samples or not (they are dropped later for want of a summary.txt)
    d = tmp_path / 'samples_dir'
    d.mkdir()
    (d / 'README.txt').write_text('not a sample\n')
    a = dict(base, intype='dir', input=str(d), allele_freq=0.01)
    merge_midas.check_arguments(a)
    assert [os.path.basename(x) for x in a['indirs']] == ['README.txt']


def test_cli_presets():
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    try:
        import merge_midas
    finally:
        sys.path.pop(0)
    base = dict(all_samples=False, all_sites=False, all_snps=False, core_sites=False, core_snps=False,
                sample_depth=5.0, fract_cov=0.4, site_prev=0.3, snp_type=['tri'], site_depth=7, site_ratio=9.0)
    a = merge_midas.add_snp_presets(dict(base, all_sites=True))
    assert a['site_prev'] == 0.0 and a['snp_type'] == ['any'] and a['site_depth'] == 7
    a = merge_midas.add_snp_presets(dict(base, core_snps=True, all_samples=True))
    assert (a['site_depth'], a['site_ratio'], a['site_prev'], a['snp_type']) == (1, 2.0, 0.95, ['bi'])
    assert a['sample_depth'] == 0.0 and a['fract_cov'] == 0.0
    a = merge_midas.add_snp_presets(dict(base, core_sites=True))
    assert a['snp_type'] == ['any'] and a['site_prev'] == 0.95
    a = merge_midas.add_snp_presets(dict(base))
    assert a['snp_type'] == ['tri'] and a['site_prev'] == 0.3


def test_matrix_writer_formats_like_python(tmp_path):
    rng = np.random.default_rng(3)
    S, n = 5, 30000
    depth = rng.integers(0, 2000, (S, n)).astype(np.uint32)
    minor = (depth * rng.random((S, n))).astype(np.uint32)
    depth[:, 7] = 0
    minor[:, 7] = 0
    minor[0, 9], depth[0, 9] = 1, 100000          # 1e-05
    minor[1, 9], depth[1, 9] = 1, 3               # 0.333
    minor[2, 9], depth[2, 9] = 2, 3               # 0.667
    minor[3, 9], depth[3, 9] = 5, 5               # 1
    keep = np.sort(rng.choice(n, 12000, replace=False))
    keep[:3] = [7, 8, 9]
    keep = np.unique(keep)
    header = "site_id\ta\tb\tc\td\te\n"
    fp, dp = str(tmp_path / "freq.txt"), str(tmp_path / "depth.txt")
    abi.write_merge_matrix(fp, header, keep, depth, minor, threads=4)
    abi.write_merge_matrix(dp, header, keep, depth, None, threads=4)
    exp_f = [header] + ["%d\t%s\n" % (i + 1, "\t".join('{0:.3g}'.format(float(minor[s, i]) / depth[s, i] if depth[s, i] > 0 else 0.0)
                                                       for s in range(S))) for i in keep]
    exp_d = [header] + ["%d\t%s\n" % (i + 1, "\t".join(str(int(depth[s, i])) for s in range(S))) for i in keep]
    assert open(fp).read() == "".join(exp_f)
    assert open(dp).read() == "".join(exp_d)
    abi.write_merge_matrix(fp, header, np.zeros(0, np.int64), depth, minor)
    assert open(fp).read() == header


def test_table_reader_foreign_gzip_and_row_limits(tmp_path):
    # a single-member gzip as Python's gzip (the reference's writer) produces it: serial inflate, parallel parse
    n = 300000
    rng = np.random.default_rng(8)
    counts = rng.integers(0, 99, (n, 4))
    p = str(tmp_path / "ref_style.snps.gz")
    with gzip.open(p, "wt", compresslevel=1) as h:
        h.write("ref_id\tref_pos\tref_allele\tdepth\tcount_a\tcount_c\tcount_g\tcount_t\n")
        h.write("".join("ctg\t%d\tA\t%d\t%d\t%d\t%d\t%d\n" % (i + 1, counts[i].sum(), *counts[i]) for i in range(n)))
    got, keys, off = abi.read_snps_table(p)
    assert np.array_equal(got, counts.astype(np.uint32)) and len(off) == n + 1
    assert bytes(keys[off[n - 1]:off[n]]).decode() == "ctg|%d|A" % n
    got2, _, _ = abi.read_snps_table(p, 123457, False)
    assert np.array_equal(got2, counts[:123457].astype(np.uint32))
    # a malformed row beyond max_rows is never looked at (the reference stops reading at max_sites)
    q = str(tmp_path / "late_garbage.snps.gz")
    with gzip.open(q, "wt") as h:
        h.write("ref_id\tref_pos\tref_allele\tdepth\tcount_a\tcount_c\tcount_g\tcount_t\n")
        h.write("c\t1\tA\t3\t1\t1\t1\t0\nc\t2\tA\t3\t1\t1\t1\t0\nc\t3\tA\tgarbage\n")
    ok, _, _ = abi.read_snps_table(q, 2, False)
    assert ok.tolist() == [[1, 1, 1, 0], [1, 1, 1, 0]]
    with pytest.raises(abi.MidasSnpsError) as e:
        abi.read_snps_table(q, 3, False)
    assert "row 3" in e.value.message


def test_species_are_dealt_round_robin_to_ranks():
    sp = ["s%d" % i for i in range(7)]
    parts = [merge.species_for_rank(sp, r, 3) for r in range(3)]
    assert parts == [["s0", "s3", "s6"], ["s1", "s4"], ["s2", "s5"]]
    assert sorted(sum(parts, [])) == sp and merge.species_for_rank(sp, 0, 1) == sp


def test_row_ranges_of_a_table_equal_slices_of_the_whole(tmp_path):
    """midas_snps_table_open_range: a rank of a site-sharded merge reads rows [lo, hi) only -- from a table that says how
    many rows its gzip members hold (only those members are inflated) and from one that does not (read whole, then cut)."""
    import gzip
    rng = np.random.default_rng(2)
    lens = [20000, 7, 33000]
    ids = ["c_a", "c_b", "c_c"]
    n = sum(lens)
    counts = rng.integers(0, 50, size=(n, 4)).astype(np.uint32)
    allele = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n)
    own = str(tmp_path / "own.snps.gz")
    o = np.cumsum([0] + lens)
    abi.write_table(own, ids, [allele[o[k]:o[k + 1]] for k in range(3)], [counts[o[k]:o[k + 1]] for k in range(3)], gz_level=1, threads=3)
    assert abi.count_snps_rows(own) == n
    plain = str(tmp_path / "plain.snps.gz")
    with gzip.open(plain, "wb") as h:
        h.write(gzip.open(own, "rb").read())
    assert abi.count_snps_rows(plain) == -1
    full_c, full_k, full_o = abi.read_snps_table(own)
    assert np.array_equal(full_c, counts)
    for path in (own, plain):
        for lo, hi in [(0, n), (0, 1), (16383, 16385), (19999, 20008), (20007, 53007), (n - 1, n), (5000, 5000), (40000, n + 10)]:
            c, k, ko = abi.read_snps_table(path, hi, True, lo)
            e = min(hi, n)
            assert np.array_equal(c, counts[lo:e]), (path, lo, hi)
            assert ko[0] == 0 and len(ko) == e - lo + 1
            assert bytes(k) == bytes(full_k[full_o[lo]:full_o[e]])
            assert np.array_equal(ko, full_o[lo:e + 1] - full_o[lo])


def test_several_tables_in_one_region(tmp_path):
    """midas_snps_tableset_*: the count columns of several samples' tables, every gzip member one task of a single
    parallel region.  Equal to reading the tables one by one; the shortest table bounds the rows (the reference's zip
    over the files); row ranges cut members; a table that does not announce its rows sends the caller the other way;
    malformed rows are named."""
    import gzip
    rng = np.random.default_rng(5)
    ids = ["c_a", "c_b", "c_c"]
    paths, tables = [], []
    for s, lens in enumerate([[20000, 7, 33000], [20000, 7, 33000], [20000, 7, 21000]]):     # the third sample's table is shorter
        n = sum(lens)
        counts = rng.integers(0, 2**31 - 1 if s == 1 else 60, size=(n, 4)).astype(np.uint32)
        allele = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n)
        o = np.cumsum([0] + lens)
        p = str(tmp_path / ("s%d.snps.gz" % s))
        abi.write_table(p, ids, [allele[o[k]:o[k + 1]] for k in range(3)], [counts[o[k]:o[k + 1]] for k in range(3)],
                        gz_level=[4, 6, 1][s], threads=3)
        paths.append(p)
        tables.append(counts)
    shortest = min(t.shape[0] for t in tables)
    got = abi.read_snps_counts(paths)
    assert [g.shape for g in got] == [(shortest, 4)] * 3
    for g, t in zip(got, tables):
        assert np.array_equal(g, t[:shortest])
    for lo, hi in [(0, 1), (16383, 16385), (19999, 20008), (20007, 40000), (shortest - 1, shortest + 50), (300, 300), (41000, -1)]:
        got = abi.read_snps_counts(paths, lo, hi)
        e = shortest if hi < 0 else min(hi, shortest)
        for g, t in zip(got, tables):
            assert np.array_equal(g, t[lo:e]), (lo, hi)
    # a table as the reference writes it (one gzip member, no row counts): not for this reader
    plain = str(tmp_path / "plain.snps.gz")
    with gzip.open(plain, "wb") as h:
        h.write(gzip.open(paths[0], "rb").read())
    assert abi.read_snps_counts([paths[0], plain]) is None
    with pytest.raises(abi.MidasSnpsError):
        abi.read_snps_counts([paths[0], str(tmp_path / "missing.snps.gz")])
    # a row that is not a row, inside a member that announces its rows: rewrite one member's text
    raw = bytearray(open(paths[0], "rb").read())
    import zlib
    text = gzip.open(paths[0], "rb").read()
    lines = text.split(b"\n")
    lines[25001] = b"c_b\t1\tA\t3\t1\tx\t1\t0"
    bad = str(tmp_path / "bad.snps.gz")
    # same member structure: compress line groups the way the writer cut them (header, then <= 16384 rows per member and contig)
    def member(payload, rows):
        z = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = z.compress(payload) + z.flush()
        total = 28 + len(body) + 8
        head = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 16, 0]) + b"MS" + (4).to_bytes(2, "little") + total.to_bytes(4, "little") \
            + b"MR" + (4).to_bytes(2, "little") + rows.to_bytes(4, "little")
        return head + body + zlib.crc32(payload).to_bytes(4, "little") + (len(payload) & 0xFFFFFFFF).to_bytes(4, "little")
    with open(bad, "wb") as h:
        h.write(member(lines[0] + b"\n", 0))
        for lo in range(1, len(lines) - 1, 10000):
            chunk = lines[lo:min(lo + 10000, len(lines) - 1)]
            h.write(member(b"\n".join(chunk) + b"\n", len(chunk)))
    assert gzip.open(bad, "rb").read() == b"\n".join(lines)
    with pytest.raises(abi.MidasSnpsError) as e:
        abi.read_snps_counts([paths[1], bad])
    assert "row 25001" in e.value.message and "bad.snps.gz" in e.value.message
    ok = abi.read_snps_counts([paths[1], bad], 0, 25000)          # the bad row is outside what is read
    assert np.array_equal(ok[1], tables[0][:25000])
