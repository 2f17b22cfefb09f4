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


def test_reference_exceptions_become_error_exits(tmp_path):
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=3000, n_reads=200, seed=5)
    reads.nm[17] = -1     # bowtie2 writes NM on every aligned record; its absence is a KeyError in the reference
    out, db = str(tmp_path / "sample"), str(tmp_path / "db")
    synth.write_sample(out, db, contigs, reads)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_midas.py"), "snps", out, "--pileup", "-d", db],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1
    assert "NM" in r.stderr and "read 17" in r.stderr
