"""End to end on the GPU box: `run_midas.py snps <outdir> --pileup` on a synthetic sample directory, compared with
the text the reference's own loops produce (through the pysam-shaped oracle).  Reads like test/test_midas.py's
_07_RunSNPs (exit code 0) plus what that test never did: checking the output."""
import gzip
import os
import subprocess
import sys

import pytest

from midas_amd import abi, synth
from oracle import pileup_oracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_outputs(contigs, reads, args):
    alns = po.alns_from_soa(reads.as_dict())
    off = contigs.site_offsets()
    oc, by = {}, {}
    for k, cid in enumerate(contigs.ids):
        seq = bytes(contigs.ref[off[k]:off[k + 1]]).decode().upper()
        oc[cid] = po.OContig(id=cid, seq=seq, species_id=contigs.species_ids[contigs.species[k]])
        by[cid] = alns[int(contigs.read_begin[k]):int(contigs.read_begin[k + 1])]
    return {sp: po.species_pileup(args, sp, oc, by) for sp in contigs.species_ids}


@pytest.mark.parametrize("extra", [[], ["--baseq", "35", "--mapid", "96", "--mapq", "30", "--readq", "30", "--aln_cov", "0.9"]])
def test_run_midas_snps_pileup_matches_reference_text(tmp_path, extra):
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=4, contig_len=6000, n_reads=9000, seed=17,
                                        var_len=True, lowercase_frac=0.05)
    out, db = str(tmp_path / "sample"), str(tmp_path / "db")
    synth.write_sample(out, db, contigs, reads)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py"), "snps", out, "--pileup", "-d", db,
                        "-t", "4"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert "Counting alleles" in r.stdout
    args = dict(abi.DEFAULT_ARGS)
    for k, v in zip(extra[0::2], extra[1::2]):
        args[k[2:]] = float(v) if k in ("--mapid", "--aln_cov") else int(v)
    exp = _oracle_outputs(contigs, reads, args)
    rows = {}
    for line in open(os.path.join(out, "snps", "summary.txt")).read().splitlines()[1:]:
        f = line.split("\t")
        rows[f[0]] = f[1:]
    for sp in contigs.species_ids:
        got = gzip.open(os.path.join(out, "snps", "output", sp + ".snps.gz"), "rt").read()
        text, st = exp[sp]
        assert got == text, sp
        st = po.fold_species_stats(st)
        assert rows[sp] == [str(st[k]) for k in ('genome_length', 'covered_bases', 'fraction_covered',
                                                 'mean_coverage', 'aligned_reads', 'mapped_reads')]
    assert os.path.isfile(os.path.join(out, "snps", "readme.txt")) and os.path.isfile(os.path.join(out, "snps", "log.txt"))


@pytest.mark.parametrize("inflate", ["on", "off"])
def test_bam_with_records_of_a_species_that_is_not_selected(tmp_path, inflate):
    """species.txt edited after the alignment (or a BAM from elsewhere): the BAM holds records of contigs nobody asked for.
    pysam's count_coverage is called per wanted contig (midas/run/snps.py:187-199) and never sees them; neither may the
    decode that leaves SEQ / QUAL / CIGAR on the device (--device_inflate on, one rank) nor the host's."""
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=2, contig_len=5000, n_reads=6000, seed=23)
    out, db = str(tmp_path / "sample"), str(tmp_path / "db")
    synth.write_sample(out, db, contigs, reads)
    listing = os.path.join(out, "snps", "species.txt")
    kept = [l for l in open(listing).read().split() if l != contigs.species_ids[1]]
    open(listing, "w").write("".join(l + "\n" for l in kept))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py"), "snps", out, "--pileup", "-d", db,
                        "--device_inflate", inflate], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    exp = _oracle_outputs(contigs, reads, dict(abi.DEFAULT_ARGS))
    for sp in kept:
        assert gzip.open(os.path.join(out, "snps", "output", sp + ".snps.gz"), "rt").read() == exp[sp][0], sp
    assert not os.path.exists(os.path.join(out, "snps", "output", contigs.species_ids[1] + ".snps.gz"))


def test_the_cli_runs_the_dependencys_pad_rule_by_default(tmp_path):
    """A BAM with the CIGAR op P: the drop-in counts what pysam's get_aligned_pairs would (the query position advances over a
    pad: hand-derived case k04c), `--pad_rule spec` what the SAM specification says (P consumes nothing: the table of the
    same read without the pad)."""
    import numpy as np
    from tests import helpers as H
    cases = {c["name"]: c for c in H.load_kat_cases()}
    case = cases["k04c_pad_pysam_rule_reads_one_late"]
    contigs, reads, _, _ = H.kat_inputs(case)
    tables = {}
    for rule in (None, "spec"):
        out, db = str(tmp_path / ("sample_%s" % rule)), str(tmp_path / ("db_%s" % rule))
        synth.write_sample(out, db, contigs, reads)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py"), "snps", out, "--pileup", "-d", db] +
                           (["--pad_rule", rule] if rule else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        rows = gzip.open(os.path.join(out, "snps", "output", "sp.snps.gz"), "rt").read().splitlines()[1:]
        tables[rule] = np.array([[int(x) for x in row.split("\t")[4:8]] for row in rows], dtype=np.uint32)
    assert np.array_equal(tables[None], H.kat_expected_counts(case))
    st, _, spec_counts, _, _ = __import__("oracle.c_oracle", fromlist=["pileup"]).pileup(abi.Thresholds.from_args(abi.DEFAULT_ARGS), contigs, reads)
    assert st == 0 and np.array_equal(tables["spec"], spec_counts) and not np.array_equal(tables["spec"], tables[None])


def test_reference_exceptions_become_error_exits(tmp_path):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=3000, n_reads=200, seed=5)
    reads.nm[17] = -1     # bowtie2 writes NM on every aligned record; its absence is a KeyError in the reference
    out, db = str(tmp_path / "sample"), str(tmp_path / "db")
    synth.write_sample(out, db, contigs, reads)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py"), "snps", out, "--pileup", "-d", db],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1
    assert "NM" in r.stderr and "read 17" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("coder", ["host", "device"])
def test_rows_streamed_from_the_device_are_the_bytes_of_the_host_writer(tmp_path, coder):
    """midas_snps_batch_write_part with the HOST's formatter: the rows leave the device slab by slab through the pinned ring
    while the formatter works -- the file must be byte for byte what batch_fetch + midas_snps_write_part write, for whole
    tables, parts, contig subsets in any order, a contig longer than a ring slot, and at both kinds of gzip level.  With the
    DEVICE's row coder (levels 1-5) the file inflates to the same text (Python's gzip also checks every member's CRC-32 and
    ISIZE); at the zlib levels the host writes either way."""
    import gzip
    from midas_amd import abi, synth
    import numpy as np
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=30011, n_reads=30000, seed=17, var_len=True)
    long_contigs, long_reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=2_500_000, n_reads=20000, seed=18)
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    with abi.Context(0) as ctx:
        ctx.set_row_coder(abi.ROWS_HOST if coder == "host" else abi.ROWS_DEVICE)
        for tag, (tab, rd) in {"small": (contigs, reads), "long": (long_contigs, long_reads)}.items():
            b = ctx.batch(tab, rd)
            try:
                b.run(thr)
                counts, allele, _ = b.fetch()
                off = tab.site_offsets()
                nc = tab.n_contigs
                picks = [list(range(nc)), list(range(nc))[::-1], [nc - 1], [], [1, 0] if nc > 1 else [0]]
                for k, pick in enumerate(picks):
                    names = ["ctg_%s_%d" % (tag, c) for c in pick]
                    for header in (True, False):
                        for level in (4, 6):
                            a = str(tmp_path / ("dev_%s_%d_%d_%d.gz" % (tag, k, header, level)))
                            h = str(tmp_path / ("host_%s_%d_%d_%d.gz" % (tag, k, header, level)))
                            b.write_part(a, pick, names, header=header, gz_level=level, threads=7)
                            abi.write_table(h, names, [allele[off[c]:off[c + 1]] for c in pick],
                                            [counts[off[c]:off[c + 1]] for c in pick], gz_level=level, threads=5, header=header)
                            if coder == "host" or level == 6:
                                assert open(a, "rb").read() == open(h, "rb").read(), (tag, pick, header, level)
                            else:
                                assert gzip.open(a, "rb").read() == gzip.open(h, "rb").read(), (tag, pick, header, level)
                with pytest.raises(abi.MidasSnpsError):
                    b.write_part(str(tmp_path / "bad.gz"), [nc], ["x"])
            finally:
                b.close()


@pytest.mark.parametrize("inflate", ["on", "off"])
def test_a_share_dealt_to_the_device_in_several_batches_writes_the_same_files(tmp_path, inflate):
    """The reference streams contig by contig and has no size limit (midas/run/snps.py:187-199); a batch of the device has
    (2 * 10^9 reads, 32 GiB of payload, the device's memory), so a GPU's contigs go up in as many batches as it takes.
    Forced here with --max_batch_reads: the tables and the summary are byte for byte those of the one-batch run, species whose
    contigs fall into different batches included."""
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=4, contig_len=6000, n_reads=9000, seed=17,
                                        var_len=True, lowercase_frac=0.05)
    outs = []
    for name, extra in (("one", []), ("many", ["--max_batch_reads", "1700"])):
        out, db = str(tmp_path / name), str(tmp_path / ("db_" + name))
        synth.write_sample(out, db, contigs, reads)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py"), "snps", out, "--pileup", "-d", db,
                            "--device_inflate", inflate] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(out)
    log = open(os.path.join(outs[1], "snps", "log.txt")).read()
    assert "work items go to the device in" in log and "work items go to the device in" not in open(os.path.join(outs[0], "snps", "log.txt")).read()
    n_batches = int(log.split("work items go to the device in ")[1].split()[0])
    assert n_batches >= 4
    for sp in contigs.species_ids:
        a = open(os.path.join(outs[0], "snps", "output", sp + ".snps.gz"), "rb").read()
        b = open(os.path.join(outs[1], "snps", "output", sp + ".snps.gz"), "rb").read()
        assert a == b, sp
    assert open(os.path.join(outs[0], "snps", "summary.txt")).read() == open(os.path.join(outs[1], "snps", "summary.txt")).read()
    assert not [f for f in os.listdir(os.path.join(outs[1], "snps", "output")) if ".part" in f]
