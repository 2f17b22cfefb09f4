"""The row coder behind <species>.snps.gz (row_deflate.cpp; gz_level 1-5 of the writers, midas_snps_deflate_rows on its
own): whatever it is fed, a stock inflater must give the text back.  zlib is the judge here -- the coder shares no code
with it.  Covered: real table text, rows that match nothing, matches longer than 258 and at distance 32768 and beyond,
byte histograms skewed enough that Huffman's tree is deeper than the format's 15 (and the code length code's 7) bits,
single-symbol and single-row inputs, and the files the writers produce at both kinds of level."""
import gzip
import zlib

import numpy as np
import pytest

from midas_amd import abi


def inflate(raw):
    d = zlib.decompressobj(-15)
    text = d.decompress(raw) + d.flush()
    assert d.eof and d.unused_data == b""
    return text


def rows_of(lines, tail_at):
    """lines: list of bytes (each a row incl. newline); tail_at(row) -> offset of its tail inside the row"""
    text = b"".join(lines)
    begin = np.cumsum([0] + [len(x) for x in lines[:-1]]).astype(np.uint32)
    tails = np.array([b + tail_at(x) for b, x in zip(begin, lines)], dtype=np.uint32)
    return text, begin, tails


def table_rows(rng, n, ref_id=b"contig_000001", depth=20.0, snp=0.02, start=1):
    d = rng.poisson(depth, n)
    ref = rng.integers(0, 4, n)
    na = np.where(rng.random(n) < snp, rng.binomial(d, 0.3), 0)
    lines = []
    for i in range(n):
        c = [0, 0, 0, 0]
        c[ref[i]] = int(d[i] - na[i])
        c[(ref[i] + 1) % 4] += int(na[i])
        lines.append(b"%s\t%d\t%c\t%d\t%d\t%d\t%d\t%d\n" % (ref_id, start + i, b"ACGT"[ref[i]], sum(c), *c))
    return lines


def second_tab(row):
    return row.index(b"\t", row.index(b"\t") + 1)


def test_table_text_round_trips_and_is_small():
    rng = np.random.default_rng(1)
    lines = table_rows(rng, 20000)
    text, rb, tb = rows_of(lines, second_tab)
    raw = abi.deflate_rows(text, rb, tb)
    assert inflate(raw) == text
    assert len(raw) < len(zlib.compress(text, 6))        # the reason it exists: denser than zlib 6 on these tables
    assert len(raw) < 0.13 * len(text)


@pytest.mark.parametrize("case", ["one row", "one byte rows", "nothing matches", "huge counts", "long ids", "far matches",
                                  "tail at row start", "identical rows"])
def test_edge_shapes(case):
    rng = np.random.default_rng(7)
    if case == "one row":
        lines, tail = [b"c\t1\tA\t0\t0\t0\t0\t0\n"], second_tab
    elif case == "one byte rows":
        lines, tail = [b"\n"] * 50 + [b"x"] * 3, (lambda r: 0)
    elif case == "nothing matches":
        lines = [bytes(rng.integers(33, 127, rng.integers(5, 60)).astype(np.uint8)) + b"\n" for _ in range(3000)]
        tail = lambda r: len(r) // 2
    elif case == "huge counts":
        lines = [b"c1\t%d\tN\t%d\t%d\t%d\t%d\t%d\n" % (4294967295 - i, 4 * 4294967295 - i, 4294967295, 4294967295 - i, 4294967295, 4294967295)
                 for i in range(2000)]
        tail = second_tab
    elif case == "long ids":       # the row head repeats 700 bytes: matches go out in pieces of <= 258, none shorter than 3
        lines = table_rows(rng, 600, ref_id=b"k" * 259 + b"z" * 259 + b"_" * 182)
        tail = second_tab
    elif case == "far matches":    # the same tail 32768 bytes back exactly, one byte closer, one byte further
        filler = lambda n: bytes(rng.integers(97, 123, n - 1).astype(np.uint8)) + b"\n"
        t = b"\tA\t33\t33\t0\t0\t0\n"
        lines = [b"q" + t, filler(32768 - len(t) - 1), b"q" + t, filler(32767 - len(t) - 1), b"q" + t, filler(32769 - len(t) - 1), b"q" + t]
        tail = lambda r: 1 if r.startswith(b"q\t") else len(r) - 1
    elif case == "tail at row start":
        lines, tail = table_rows(rng, 500), (lambda r: 0)
    else:
        lines, tail = [b"same\t7\tA\t9\t9\t0\t0\t0\n"] * 4000, second_tab
    text, rb, tb = rows_of(lines, tail)
    assert inflate(abi.deflate_rows(text, rb, tb)) == text


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_code_lengths_stay_within_the_format(seed):
    """Byte counts 1, 1, 2, 4, ... 2^k: Huffman's tree for them is a comb deeper than 15 levels, and with all the code
    lengths 1..15 in use the code length code passes 7 bits as well -- both must be cut back to a complete code."""
    rng = np.random.default_rng(seed)
    syms = rng.permutation(np.arange(33, 127))[:22 + seed]
    parts = [bytes([syms[0]])]
    for k, s in enumerate(syms):
        parts.append(bytes([s]) * (1 << min(k, 17)))
    blob = np.frombuffer(b"".join(parts), np.uint8).copy()
    rng.shuffle(blob)
    data = blob.tobytes()
    lines = [data[i:i + 97] for i in range(0, len(data), 97)]
    text, rb, tb = rows_of(lines, lambda r: min(40, len(r) - 1))
    assert inflate(abi.deflate_rows(text, rb, tb)) == text


def test_random_row_structures():
    rng = np.random.default_rng(11)
    for trial in range(30):
        n = int(rng.integers(1, 400))
        alphabet = rng.integers(1, 255, int(rng.integers(1, 40))).astype(np.uint8)
        pool = [bytes(rng.choice(alphabet, int(rng.integers(1, 30)))) for _ in range(int(rng.integers(1, 12)))]
        lines = []
        for i in range(n):
            head = b"id" + str(1000 + i).encode()
            lines.append(head + pool[int(rng.integers(0, len(pool)))] + (b"\n" if rng.random() < 0.9 else b""))
        text, rb, tb = rows_of(lines, lambda r: min(len(r) - 1, 2 + len(str(1000))))
        assert inflate(abi.deflate_rows(text, rb, tb)) == text, trial


def test_bad_arguments():
    with pytest.raises(abi.MidasSnpsError):
        abi.deflate_rows(b"abc\n", [0, 5], [1, 6])          # a row that starts past the end
    with pytest.raises(abi.MidasSnpsError):
        abi.deflate_rows(b"abc\nabc\n", [0, 4], [5, 6])      # a tail outside its row


@pytest.mark.parametrize("level", [1, 4, 5, 6, 9, 0])
def test_writers_at_every_level_give_the_same_text(tmp_path, level):
    rng = np.random.default_rng(3)
    ids = ["NC_1", "second_contig.with.dots", "c3"]
    sizes = [16384 + 5, 1, 40000]                       # a member boundary inside a contig, a one-site contig
    alleles, counts = [], []
    for n in sizes:
        ref = rng.integers(0, 4, n)
        c = np.zeros((n, 4), np.uint32)
        c[np.arange(n), ref] = rng.poisson(15, n)
        c[rng.random(n) < 0.01] = rng.integers(0, 2**32 - 1, 4, dtype=np.uint64).astype(np.uint32)
        alleles.append(np.frombuffer(b"ACGTN", np.uint8)[np.where(rng.random(n) < 0.01, 4, ref)].copy())
        counts.append(c)
    path = str(tmp_path / ("t%d.snps.gz" % level))
    abi.write_table(path, ids, alleles, counts, gz_level=level, threads=3)
    want = ["ref_id\tref_pos\tref_allele\tdepth\tcount_a\tcount_c\tcount_g\tcount_t"]
    for cid, al, c in zip(ids, alleles, counts):
        for i in range(len(al)):
            r = [int(x) for x in c[i]]
            want.append("%s\t%d\t%s\t%d\t%d\t%d\t%d\t%d" % (cid, i + 1, chr(al[i]), sum(r), *r))
    assert gzip.open(path, 'rt').read() == "\n".join(want) + "\n"
    # the reader of merge_midas.py snps finds the members and their row counts whatever wrote them
    assert abi.count_snps_rows(path) == sum(sizes)


def test_length_limited_codes_are_complete(tmp_path):
    """tests/cpp/huffman_lengths_check.cpp: 12 000 frequency tables (flat, Fibonacci, powers of two, rare-among-heavy) x the
    three alphabets of a dynamic block; every code must be complete and within 15 / 15 / 7 bits."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "huffman_lengths_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(here, "cpp", "huffman_lengths_check.cpp"),
                    os.path.join(here, "..", "midas_amd", "csrc", "row_deflate.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok ")
