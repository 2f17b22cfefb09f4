"""tests/golden/check_kats_against_pysam.py -- the script that holds the hand-derived cases to REAL pysam wherever pysam can be
imported (it cannot here: /root/reference/setup.py:15 names it, the image has none).  What can be checked without it: the script
skips cleanly; the BAMs its own struct + zlib writer produces are the cases' records (read back by this repository's decoder);
and its whole flow -- write, index, count_coverage with the filter as read_callback, compare -- comes out with no difference
when a stand-in for pysam built on the Python oracle (under the pysam pad rule) sits in `sys.modules`."""
import importlib.util
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from midas_amd import abi
from oracle import pileup_oracle as po
from tests import helpers as H

SCRIPT = os.path.join(H.GOLDEN, "check_kats_against_pysam.py")


def _load():
    spec = importlib.util.spec_from_file_location("check_kats_against_pysam", SCRIPT)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_without_pysam_the_script_skips():
    try:
        import pysam  # noqa: F401
        pytest.skip("pysam is importable here: run the script itself")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, SCRIPT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout.startswith("SKIPPED"), r.stdout + r.stderr


def test_the_scripts_bam_writer_writes_the_cases_records(tmp_path):
    m = _load()
    for case in H.load_kat_cases():
        reads = sorted(case["reads"], key=lambda r: r["pos"])
        path = str(tmp_path / (case["name"] + ".bam"))
        m.write_bam(path, "contig_1", case["contig_len"], reads)
        names, lens, refid, got = abi.read_bam(path)
        want = H.reads_from_dicts(reads)
        assert names == ["contig_1"] and lens == [case["contig_len"]] and got.n_reads == want.n_reads
        for k in ("pos", "mapq", "nm", "l_seq", "seq4", "qual", "cigar", "seq_off", "qual_off", "cigar_off"):
            np.testing.assert_array_equal(getattr(got, k), getattr(want, k), err_msg="%s %s" % (case["name"], k))


class _Segment:
    """The five attributes keep_read reads, as pysam's AlignedSegment serves them ([EXT], oracle/pileup_oracle.py)."""

    def __init__(self, a):
        self.a = a
        self.query_length = 0 if a.seq is None else len(a.seq)
        self.query_alignment_sequence = None if a.seq is None else a.seq[po.query_alignment_start(a):max(po.query_alignment_start(a), po.query_alignment_end(a))]
        self.tags = [("AS", 0)] + ([("NM", a.nm)] if a.nm is not None else [])
        self.query_qualities = None if a.qual is None else np.array(a.qual, dtype=np.uint8)
        self.mapping_quality = a.mapq


def _pysam_double():
    mod = types.ModuleType("pysam")
    mod.__version__ = "stand-in (oracle)"
    mod.index = lambda path: open(path + ".bai", "wb").close()

    class AlignmentFile:
        def __init__(self, path, mode):
            _, _, _, reads = abi.read_bam(path)
            self.alns = po.alns_from_soa(reads.as_dict())

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def count_coverage(self, contig, start, end, quality_threshold, read_callback):
            counts = [[0] * end for _ in range(4)]
            for a in self.alns:
                if not read_callback(_Segment(a)):
                    continue
                if a.seq is None:
                    continue
                for qpos, refpos in po.get_aligned_pairs_matches_only(a):
                    if 0 <= refpos < end:
                        if qpos >= len(a.seq):
                            raise IndexError("string index out of range")
                        if (quality_threshold and a.qual is not None and a.qual[qpos] >= quality_threshold) or not quality_threshold:
                            if a.seq[qpos] in "ACGT":
                                counts["ACGT".index(a.seq[qpos])][refpos] += 1
            return counts
    mod.AlignmentFile = AlignmentFile
    return mod


def test_the_scripts_flow_with_a_stand_in_for_pysam(tmp_path, monkeypatch, capsys):
    m = _load()
    monkeypatch.setitem(sys.modules, "pysam", _pysam_double())
    monkeypatch.setattr(sys, "argv", [SCRIPT, "--keep", str(tmp_path), "--random", "150"])
    po.set_pad_rule(True)
    try:
        rc = m.main()
    finally:
        po.set_pad_rule(False)
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "0 difference(s)" in out and "follows the pysam rule" in out and "MISMATCH" not in out
    assert out.count(" ok ") >= 20
