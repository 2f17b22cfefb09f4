"""The read filter pinned against the REFERENCE'S OWN keep_read: tests/golden/keep_read_vectors.json holds records and the
outcomes (True / False / exception type, aligned/mapped counters) the reference's function produced for them
(tests/golden/make_keep_read_vectors.py executes it from /root/reference in the build container).  What a BAM record's
five pysam attributes are stays [EXT]; each record is realised here as `<L-a>S<a>M` (aligned length a, query length L)."""
import json
import os

import numpy as np
import pytest

from oracle import pileup_oracle as po
from tests import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))
EXC_STATUS = {"KeyError": (2,), "ZeroDivisionError": (3,), "TypeError": (1, 4)}


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(HERE, "golden", "keep_read_vectors.json")) as h:
        return json.load(h)


def as_read(case, pos):
    L, a = case['seq_len'], case['align_len']
    if L is None:                       # no SEQ stored; the CIGAR still describes the alignment
        return dict(pos=pos, cigar="10M", seq="", qual=[], nm=case['nm'], mapq=case['mapq'])
    cigar = "%dM" % a if a == L else ("%dS" % L if a == 0 else "%dS%dM" % (L - a, a))
    return dict(pos=pos, cigar=cigar, seq="A" * L, qual="absent" if case['quals'] is None else case['quals'],
                nm=case['nm'], mapq=case['mapq'])


def test_oracle_keep_read_matches_the_reference(vectors):
    for run in vectors['runs']:
        args = dict(run['thresholds'], baseq=0)
        stats = {'aligned_reads': 0, 'mapped_reads': 0}
        for case, ref in zip(vectors['cases'], run['results']):
            d = as_read(case, 0)
            aln = po.Aln(pos=0, mapq=d['mapq'], flag=0, cigar=H.parse_cigar(d['cigar']), seq=d['seq'] or None,
                         qual=None if d['qual'] == "absent" or d['seq'] == "" else d['qual'], nm=d['nm'])
            try:
                got = bool(po.keep_read(aln, args, stats))
            except po.PileupError as e:
                got = e.kind
            assert got == ref, (case['seq_len'], case['align_len'], case['nm'], case['mapq'], run['thresholds'])
        assert stats == run['aln_stats']


@pytest.mark.gpu
def test_device_keep_read_matches_the_reference(vectors):
    from midas_amd import abi
    cases = vectors['cases']
    with abi.Context(0) as ctx:
        for run in vectors['runs']:
            args = dict(abi.DEFAULT_ARGS, baseq=0, **run['thresholds'])    # baseq 0: a kept read shows at its sites
            thr = abi.Thresholds.from_args(args)
            plain = [(c, r) for c, r in zip(cases, run['results']) if isinstance(r, bool)]
            reads = H.reads_from_dicts([as_read(c, 300 * i) for i, (c, _) in enumerate(plain)])
            n = len(plain)
            contig = H.single_contig(300 * n + 300, n)
            counts, _, stats = ctx.pileup(thr, contig, reads)
            depth_at_start = counts[[300 * i for i in range(n)]].sum(axis=1)
            assert (depth_at_start > 0).tolist() == [r for _, r in plain]
            assert int(stats[0, abi.STAT_ALIGNED_READS]) == n
            assert int(stats[0, abi.STAT_MAPPED_READS]) == sum(r for _, r in plain)
            for c, r in zip(cases, run['results']):
                if isinstance(r, bool):
                    continue
                reads = H.reads_from_dicts([as_read(cases[0], 0), as_read(c, 400), as_read(cases[1], 800)])
                with pytest.raises(abi.MidasSnpsError) as e:
                    ctx.pileup(thr, H.single_contig(2000, 3), reads)
                assert e.value.status in EXC_STATUS[r] and e.value.read_index == 1, (c, r, e.value.status)
                assert e.value.status == (1 if c['seq_len'] is None else 4) if r == "TypeError" else True
