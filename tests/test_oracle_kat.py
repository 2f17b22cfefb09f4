"""The oracle (Python and C restatements) against the hand-derived known-answer cases.

The reference holds no golden vectors for this path (test/test_midas.py:98-102 asserts exit
codes only), so these hand-derived cases are what pins the oracle; the oracle in turn is what
the HIP path is compared with in the -m gpu tests.
"""
import numpy as np
import pytest

from oracle import c_oracle, pileup_oracle as po
from tests import helpers as H

CASES = H.load_kat_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_c_oracle_matches_hand_derived(case):
    contigs, reads, thr, _ = H.kat_inputs(case)
    c_oracle.set_pad_rule(H.kat_pysam_pad_rule(case))
    try:
        st, err_read, counts, allele, stats = c_oracle.pileup(thr, contigs, reads)
    finally:
        c_oracle.set_pad_rule(False)
    if "error" in case:
        assert st == case["error"]
        assert err_read == case.get("error_read", 0)
        return
    assert st == 0
    np.testing.assert_array_equal(counts, H.kat_expected_counts(case))
    np.testing.assert_array_equal(stats, H.kat_expected_stats(case))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_python_oracle_matches_hand_derived(case):
    contigs, reads, thr, args = H.kat_inputs(case)
    alns = po.alns_from_soa(reads.as_dict())
    stats = {'aligned_reads': 0, 'mapped_reads': 0}
    kind_of = {"TypeError": None, "KeyError": po.ERR_NO_NM, "ZeroDivisionError": po.ERR_ZERO_ALIGN,
               "IndexError": po.ERR_CIGAR_OVERRUN}
    po.set_pad_rule(H.kat_pysam_pad_rule(case))
    try:
        cc = po.count_coverage(alns, case["contig_len"], args['baseq'], lambda r: po.keep_read(r, args, stats))
    except po.PileupError as e:
        po.set_pad_rule(False)
        assert "error" in case, "unexpected %s" % e
        exp = case["error"]
        if e.kind == "TypeError":
            assert exp in (po.ERR_NO_SEQ, po.ERR_NO_QUAL)
        else:
            assert kind_of[e.kind] == exp
        assert e.read_index == case.get("error_read", 0)
        return
    po.set_pad_rule(False)
    assert "error" not in case
    got = np.array(cc, dtype=np.uint32).T
    np.testing.assert_array_equal(got, H.kat_expected_counts(case))
    assert stats['aligned_reads'] == case["aligned_reads"]
    assert stats['mapped_reads'] == case["mapped_reads"]
    depth = got.sum(axis=1)
    assert int(depth.sum()) == case["total_depth"]
    assert int((depth > 0).sum()) == case["covered_bases"]
