"""Developer soak (GPU box) for the two widened rows: midas_merge_sites and midas_genes_count on random shapes against
their oracles, bit for bit.  usage: python tools/soak_next.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midas_amd import abi, synth  # noqa: E402
from oracle import genes_oracle as go  # noqa: E402
from oracle import merge_oracle as mo  # noqa: E402
from oracle import pileup_oracle as po  # noqa: E402

SNP = [None, 'mono', 'bi', 'tri', 'quad']


def merge_case(ctx, rng):
    S = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 40, 70]))
    n = int(rng.integers(1, 1500))
    lam = float(rng.choice([0.2, 3, 12, 300]))
    counts = []
    for s in range(S):
        c = rng.poisson(lam, (n, 4)).astype(np.int64)
        c[rng.random((n, 4)) < 0.6] = 0
        if rng.random() < 0.3:
            c[rng.integers(0, n), rng.integers(0, 4)] = int(rng.integers(1, 2**31 - 1))
        counts.append(c.astype(np.uint32))
    mean = [float(rng.choice([0.5, 3.0, 12.25, 100.0])) for _ in range(S)]
    args = dict(abi.DEFAULT_MERGE_ARGS, site_depth=int(rng.choice([0, 1, 2, 5])), site_ratio=float(rng.choice([0.5, 2.0, 5.0])),
                site_prev=float(rng.choice([0.0, 0.3, 0.95, 1.0])), allele_freq=float(rng.choice([0.01, 0.2, 0.5])),
                snp_type=[['any'], ['bi'], ['mono', 'bi', 'tri', 'quad'], ['tri', 'quad']][int(rng.integers(0, 4))])
    got = ctx.merge_sites(abi.MergeParams.from_args(args), counts, mean)
    for i in range(n):
        c = [[int(x) for x in counts[s][i]] for s in range(S)]
        pooled = mo.pooled_counts(c)
        major, minor, st = mo.call_alleles(pooled, args['allele_freq'])
        mafs, depths = mo.per_sample(c, major, minor)
        cs, prev = mo.prevalence(mean, depths, args['site_depth'], args['site_ratio'])
        why = mo.flag_reason(prev, st, args['site_prev'], args['snp_type'])
        ok = (got['major'][i] == (255 if major is None else major) and got['minor'][i] == (255 if minor is None else minor)
              and got['snp_type'][i] == SNP.index(st) and got['count_samples'][i] == cs
              and got['flag'][i] == {None: 0, 'min_prev': 1, 'snp_type': 2}[why] and list(got['depth'][:, i]) == depths
              and list(got['pooled'][i]) == pooled
              and list(got['minor_count'][:, i]) == ([x[minor] for x in c] if minor is not None else [0] * S))
        if not ok:
            return "merge S=%d n=%d site %d args %s" % (S, n, i, args)
    return None


def genes_case(ctx, rng):
    ds = synth.make_pangenome_dataset(n_species=int(rng.integers(1, 4)), genes_per_species=int(rng.choice([4, 40, 200])),
                                      n_reads=int(rng.choice([0, 50, 3000, 30000])), read_len=int(rng.choice([50, 100, 150, 250])),
                                      seed=int(rng.integers(1, 1 << 30)), var_len=bool(rng.random() < 0.5),
                                      silent_fraction=float(rng.choice([0.0, 0.3])))
    reads, refid = ds['reads'], ds['refid']
    if reads.n_reads > 100 and rng.random() < 0.3:          # a hot gene (the wave kernel from 2049 reads on)
        refid = np.where(rng.random(refid.size) < 0.7, refid[0], refid).astype(np.int32)
    lengths = [len(s) for s in ds['gene_seq']]
    args = dict(mapid=float(rng.choice([1.0, 94.0, 98.5])), readq=int(rng.choice([0, 20, 35])), mapq=int(rng.choice([0, 10, 42])),
                aln_cov=float(rng.choice([0.0, 0.75, 1.0])))
    recs = []
    for aln, rid in zip(po.alns_from_soa(reads.as_dict()), refid):
        a = max(0, po.query_alignment_end(aln) - po.query_alignment_start(aln))
        recs.append((int(rid), a, len(aln.seq), aln.nm, aln.qual, aln.mapq))
    ea, em, ed, _ = go.count_mapped_bp(args, recs, ds['gene_ids'], ds['gene_species'], lengths)
    al, mp, dp, _ = ctx.genes_count(abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, **args)), reads, refid, lengths)
    if al.tolist() != ea or mp.tolist() != em or [repr(float(x)) for x in dp] != [repr(float(x)) for x in ed]:
        return "genes %d reads %d genes args %s" % (reads.n_reads, len(lengths), args)
    return None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    t_end = time.time() + budget
    n = {'merge': 0, 'genes': 0}
    bad = []
    with abi.Context(0) as ctx:
        while time.time() < t_end:
            for name, fn in (('merge', merge_case), ('genes', genes_case)):
                r = fn(ctx, rng)
                n[name] += 1
                if r:
                    bad.append(r)
                    print("MISMATCH", r, flush=True)
    print("soak_next: %s cases, %d mismatches" % (n, len(bad)))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
