"""The library's FASTA reader (midas_fasta_load, hostio.cpp) against midas_amd/fasta.py -- the restatement of
`Bio.SeqIO.parse(infile, 'fasta')` + `str(rec.seq).upper()` at midas/run/snps.py:59-62 that the host used to read the genomes
with: the same records (ids, sequences, order) for every oddity a FASTA file can hold, plain and gzip, many files at once."""
import gzip
import os

import numpy as np
import pytest

from midas_amd import abi, fasta
from midas_amd.run import snps as msnps


CASES = {
    "plain": b">c1 desc\nACGT\nacgt\n>c2\nNNNN\n",
    "no_trailing_newline": b">c1\nAC\nGT",
    "crlf": b">c1 x\r\nACGT\r\nTT\r\n>c2\r\nGG\r\n",
    "text_before_first_header": b"; a comment\nACGT\n>c1\nAC\n",
    "no_header_at_all": b"ACGT\nACGT\n",
    "empty": b"",
    "only_header": b">c1",
    "header_then_header": b">a\n>b\nAC\n>c\n",
    "blank_header": b">\nACGT\n> \t\nGG\n",
    "spaces_and_tabs_in_sequence": b">c1\nAC GT\tAC\x0bGT\x0cAA\n\n\nTT\n",
    "gt_inside_a_line": b">c1\nAC>GT\nA >x\n>c2\nT\n",
    "gt_after_space_at_line_start": b">c1\nAC\n >notaheader\nGG\n",
    "lower_and_iupac": b">c1\nacgtnryswkmbdhv-*\n",
    "high_bytes": b">c\xe9 d\nAC\xe9\xb5GT\n",
    "duplicate_ids": b">c1\nAA\n>c1\nCC\n",
    "long_lines": b">c1\n" + b"ACGT" * 50000 + b"\n>c2\n" + b"\n".join([b"TTGCA" * 12] * 3000) + b"\n",
    "leading_whitespace_in_header": b">  \tc1   more words\nAC\n",
    "newline_first": b"\n>c1\nAC\n",
}


def _expect(data):
    return [(rid, seq.upper()) for rid, seq in fasta.parse_bytes(data)]


def _native(paths):
    pool, recs = abi.read_fasta_files(paths, threads=3)
    return [(rid, fi, bytes(pool[at:at + n])) for rid, fi, at, n in recs], pool


@pytest.mark.parametrize("name", sorted(CASES))
def test_the_records_are_the_python_readers(tmp_path, name):
    data = CASES[name]
    plain, gz = str(tmp_path / "g.fna"), str(tmp_path / "g.fna.gz")
    open(plain, "wb").write(data)
    with gzip.open(gz, "wb") as h:
        h.write(data)
    want = _expect(data)
    for path in (plain, gz):
        got, _ = _native([path])
        assert [(r, s) for r, _, s in got] == want, path


def test_many_files_lie_back_to_back_in_file_order(tmp_path):
    rng = np.random.default_rng(5)
    paths, want = [], []
    for k in range(37):
        recs = []
        for c in range(int(rng.integers(0, 6))):
            seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNn", np.uint8), int(rng.integers(0, 30000))))
            width = int(rng.integers(1, 200))
            recs.append(b">f%d_c%d x\n" % (k, c) + b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + b"\n")
            want.append(("f%d_c%d" % (k, c), k, seq.upper()))
        data = b"".join(recs)
        path = str(tmp_path / ("g%02d.fna%s" % (k, ".gz" if k % 3 == 0 else "")))
        if path.endswith(".gz"):
            with gzip.open(path, "wb") as h:
                h.write(data)
        else:
            open(path, "wb").write(data)
        paths.append(path)
    got, pool = _native(paths)
    assert got == want
    assert pool.size == sum(len(s) for _, _, s in want) and not pool.flags.writeable
    assert _native([])[0] == []


def test_a_file_that_cannot_be_read_is_an_error(tmp_path):
    with pytest.raises(abi.MidasSnpsError):
        abi.read_fasta_files([str(tmp_path / "missing.fna")])
    bad = str(tmp_path / "bad.fna.gz")
    good = gzip.compress(b">c1\n" + b"ACGT" * 100000 + b"\n")
    open(bad, "wb").write(good[:len(good) // 2])
    with pytest.raises(abi.MidasSnpsError):
        abi.read_fasta_files([bad])


def test_initialize_contigs_reads_through_it(tmp_path):
    class Sp:
        def __init__(self, i, p):
            self.id, self.paths = i, {"fna": p}
    a, b = str(tmp_path / "a.fna"), str(tmp_path / "b.fna.gz")
    open(a, "wb").write(b">x1\nacgt\nNN\n>x2\nGG\n")
    with gzip.open(b, "wb") as h:
        h.write(b">y1 z\nTTtt\n")
    contigs = msnps.initialize_contigs({"A": Sp("A", a), "B": Sp("B", b)})
    assert sorted(contigs) == ["x1", "x2", "y1"]
    assert contigs["x1"].seq == "ACGTNN" and contigs["x1"].species_id == "A" and contigs["x1"].length == 6
    assert contigs["y1"].seq == "TTTT" and contigs["y1"].species_id == "B"
    assert bytes(contigs["x2"].seq_bytes) == b"GG" and contigs["x2"].pool is contigs["y1"].pool
    with pytest.raises(SystemExit):
        msnps.initialize_contigs({"C": Sp("C", str(tmp_path / "none.fna"))})
