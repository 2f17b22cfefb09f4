"""The merge oracle pinned against the REFERENCE ITSELF: tests/golden/merge_vectors.json holds inputs and the outputs of
the reference's own GenomicSite class and codon helpers (generated in the build container by
tests/golden/make_merge_vectors.py, which executes those definitions from /root/reference).  Every intermediate the
reference exposes is compared -- allele calls, SNP type, per-sample depth and MAF (bit-exact floats via repr),
count_samples, prevalence, flag reason -- and so are the three output texts."""
import json
import os

import pytest

from oracle import merge_oracle as mo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(HERE, "golden", "merge_vectors.json")) as h:
        return json.load(h)


def test_vectors_exercise_every_branch(vectors):
    seen_types, seen_flags, seen_loci, n_sites = set(), set(), set(), 0
    for g in vectors['groups']:
        for v in g['variants']:
            for s in v['sites']:
                seen_types.add(s[2])
                seen_flags.add(s[7])
                n_sites += 1
            for line in v['info'].splitlines():
                f = line.split('\t')
                seen_loci.add((f[11], f[14]))
    assert seen_types == {None, 'mono', 'bi', 'tri', 'quad'} and seen_flags == {'keep', 'min_prev', 'snp_type'}
    assert {('IGR', 'NA'), ('CDS', 'NA'), ('CDS', '1D'), ('CDS', '4D'), ('tRNA', 'NA')} <= seen_loci
    assert n_sites == 3 * 5 * 320


def test_oracle_matches_the_reference_site_by_site(vectors):
    alle = 'ACGT'
    for g in vectors['groups']:
        for v in g['variants']:
            args = v['args']
            for key, counts, ref in zip(g['keys'], g['counts'], v['sites']):
                pooled = mo.pooled_counts(counts)
                major, minor, snp_type = mo.call_alleles(pooled, args['allele_freq'])
                mafs, depths = mo.per_sample(counts, major, minor)
                cs, prev = mo.prevalence(g['mean_depths'], depths, args['site_depth'], args['site_ratio'])
                why = mo.flag_reason(prev, snp_type, args['site_prev'], args['snp_type'])
                got = [None if major is None else alle[major], None if minor is None else alle[minor], snp_type, depths,
                       [repr(float(x)) for x in mafs], cs, repr(prev), why or "keep"]
                assert got == ref, (key, counts, args)


def test_oracle_reproduces_the_reference_text(vectors):
    for g in vectors['groups']:
        tabs = [[",".join(str(x) for x in site[s]) for site in g['counts']] for s in range(g['n_samples'])]
        genes = {'list': g['genes'], 'index': 0}
        for v in g['variants']:
            info, freq, depth = mo.merge_species(g['keys'], tabs, g['mean_depths'], v['args'], genes)
            assert "".join(info) == v['info']
            assert "".join(freq) == v['freq']
            assert "".join(depth) == v['depth']
            assert len(info) == sum(1 for s in v['sites'] if s[7] == 'keep')


def test_product_gene_cursor_matches_the_reference_annotation(vectors):
    from midas_amd.merge import annotate
    for g in vectors['groups']:
        for v in g['variants']:
            cur = annotate.GeneCursor([dict(x) for x in g['genes']])
            for line in v['info'].splitlines():
                f = line.split('\t')
                got = cur.lookup(f[1], int(f[2]))
                assert ["NA" if x is None else x for x in got] == [f[11], f[12], f[14], f[15]], line


@pytest.mark.gpu
def test_device_merge_matches_the_reference_vectors(vectors):
    """midas_merge_sites (through the C-ABI) against the reference's own per-site results."""
    import numpy as np
    from midas_amd import abi
    names = [None, 'mono', 'bi', 'tri', 'quad']
    with abi.Context(0) as ctx:
        for g in vectors['groups']:
            S = g['n_samples']
            counts = [np.array([site[s] for site in g['counts']], dtype=np.uint32) for s in range(S)]
            for v in g['variants']:
                a = dict(v['args'])
                res = ctx.merge_sites(abi.MergeParams.from_args(a), counts, g['mean_depths'])
                for i, ref in enumerate(v['sites']):
                    mj, mn = int(res['major'][i]), int(res['minor'][i])
                    assert ("ACGT"[mj] if mj < 4 else None) == ref[0] and ("ACGT"[mn] if mn < 4 else None) == ref[1]
                    assert names[int(res['snp_type'][i])] == ref[2]
                    d = res['depth'][:, i].tolist()
                    assert d == ref[3]
                    m = res['minor_count'][:, i].tolist()
                    mafs = [repr(float(m[s]) / d[s] if (mn < 4 and d[s] > 0) else 0.0) for s in range(S)]
                    assert mafs == ref[4]
                    assert int(res['count_samples'][i]) == ref[5]
                    assert [None, 'min_prev', 'snp_type'][int(res['flag'][i])] == (None if ref[7] == 'keep' else ref[7])


def test_native_info_writer_matches_the_reference_text(vectors, tmp_path):
    """midas_merge_write_info (host C++: annotation + info lines) against the reference's own snps_info text."""
    import numpy as np
    from midas_amd import abi
    code = {None: 255, 'A': 0, 'C': 1, 'G': 2, 'T': 3}
    types = [None, 'mono', 'bi', 'tri', 'quad']
    for gi, g in enumerate(vectors['groups']):
        keys = "".join(g['keys']).encode()
        key_off = np.zeros(len(g['keys']) + 1, np.int64)
        key_off[1:] = np.cumsum([len(k.encode()) for k in g['keys']])
        pooled = np.array([[sum(site[s][a] for s in range(g['n_samples'])) for a in range(4)] for site in g['counts']], np.uint64)
        for vi, v in enumerate(g['variants']):
            calls = np.array([[code[s[0]], code[s[1]], types.index(s[2]), 0 if s[7] == 'keep' else 1] for s in v['sites']], np.uint8)
            res = dict(major=calls[:, 0], minor=calls[:, 1], snp_type=calls[:, 2], flag=calls[:, 3],
                       count_samples=np.array([s[5] for s in v['sites']], np.uint32), pooled=pooled)
            keep = np.nonzero(calls[:, 3] == 0)[0]
            out = str(tmp_path / ("info_%d_%d.txt" % (gi, vi)))
            abi.write_merge_info(out, "H\n", keep, keys, key_off, res, g['genes'], threads=3)
            assert open(out).read() == "H\n" + v['info']
