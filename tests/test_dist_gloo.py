"""The N > 1 path on CPU: world_size-2 gloo process group (the same code runs over RCCL/xGMI with backend nccl).
Covers the species -> rank assignment and the single all-gather of per-species summary rows."""
import os
import socket
import subprocess
import time
import sys

import numpy as np
import pytest

from midas_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from midas_amd import dist
rank, ws = dist.init_from_env("gloo")
assert ws == 2
weights = {"sp_%%02d" %% i: float((i * 7919) %% 13 + 1) for i in range(9)}
owner = dist.shard_species(weights, ws)
ids = sorted(weights)
rows = np.zeros((len(ids), 5), dtype=np.int64)
for i, sp in enumerate(ids):
    if owner[sp] == rank:
        rows[i] = [1000 + i, 10 * i, 100 * i, 7 * i + rank, i]
tot = dist.all_gather_summary(rows)
dist.barrier()
print(json.dumps({"rank": rank, "owner": owner, "tot": tot.tolist()}))
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_species_is_deterministic_and_balanced():
    w = {"a": 10.0, "b": 9.0, "c": 1.0, "d": 1.0, "e": 8.0}
    o1 = dist.shard_species(w, 2)
    o2 = dist.shard_species(dict(reversed(list(w.items()))), 2)
    assert o1 == o2
    load = [sum(w[s] for s in w if o1[s] == r) for r in range(2)]
    assert max(load) <= (4.0 / 3.0) * max(sum(w.values()) / 2.0, max(w.values()))   # the LPT guarantee
    assert dist.shard_species(w, 1) == {s: 0 for s in w}
    assert set(dist.shard_species({"x": 1.0}, 8).values()) == {0}


def test_all_gather_summary_single_process_is_identity():
    rows = np.arange(15, dtype=np.int64).reshape(3, 5)
    np.testing.assert_array_equal(dist.all_gather_summary(rows), rows)


def test_two_ranks_gloo_all_gather_of_summary_rows(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e
        outs.append(json.loads(o.strip().splitlines()[-1]))
    assert outs[0]["owner"] == outs[1]["owner"]                 # same assignment without talking
    assert set(outs[0]["owner"].values()) == {0, 1}
    assert outs[0]["tot"] == outs[1]["tot"]                     # every rank ends with every species' row
    ids = sorted(outs[0]["owner"])
    for i, sp in enumerate(ids):
        r = outs[0]["owner"][sp]
        assert outs[0]["tot"][i] == [1000 + i, 10 * i, 100 * i, 7 * i + r, i]


GENES_WORKER = r'''
import io, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from midas_amd import abi, dist
from midas_amd.run import genes as mgenes
from oracle import genes_oracle as go
from oracle import pileup_oracle as po


class OracleContext:
    """Stands in for the device in this CPU test: midas_genes_count's contract, computed by the oracle."""
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def genes_count(self, thr, reads, ref_id, gene_length):
        args = dict(mapid=thr.mapid, readq=thr.readq, mapq=thr.mapq, aln_cov=thr.aln_cov)
        recs = []
        for aln, rid in zip(po.alns_from_soa(reads.as_dict()), ref_id):
            a = max(0, po.query_alignment_end(aln) - po.query_alignment_start(aln))
            recs.append((int(rid), a, len(aln.seq), aln.nm, aln.qual, aln.mapq))
        n = len(gene_length)
        al, mp, dp, _ = go.count_mapped_bp(args, recs, list(range(n)), ["x"] * n, [int(x) for x in gene_length])
        return np.array(al, np.int64), np.array(mp, np.int64), np.array(dp, np.float64), 0.0
    # the two halves (midas_genes_terms / midas_genes_sum): what N ranks use below the species
    def genes_terms(self, thr, reads, ref_id, gene_length):
        out = np.zeros(int(reads.n_reads), np.float64)
        for i, (aln, rid) in enumerate(zip(po.alns_from_soa(reads.as_dict()), ref_id)):
            a = max(0, po.query_alignment_end(aln) - po.query_alignment_start(aln))
            if go.keep_read(a, len(aln.seq), aln.nm, aln.qual, aln.mapq, thr.mapid, thr.readq, thr.mapq, thr.aln_cov):
                out[i] = a / float(int(gene_length[int(rid)]))
        return out
    def genes_sum(self, gene, term, n_genes):
        al, mp, dp = [0] * n_genes, [0] * n_genes, [0.0] * n_genes
        for g, t in zip(np.asarray(gene).tolist(), np.asarray(term).tolist()):
            al[g] += 1
            mp[g] += t > 0
            dp[g] += t
        return np.array(al, np.int64), np.array(mp, np.int64), np.array(dp, np.float64)


out, db = sys.argv[1], sys.argv[2]
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    dist.init_from_env("gloo")
args = dict(outdir=out, db=db, build_db=False, align=False, cov=True, species_id=None, threads=1, log=io.StringIO(),
            mapid=94.0, readq=20, mapq=0, aln_cov=0.75)
species = mgenes.initialize_species(args)
genes = mgenes.initialize_genes(args, species)
real = os.environ.get("GENES_REAL_DEVICE") == "1"      # (tests/test_gpu_dist.py: the ranks share device 0)
mgenes.pangenome_coverage(args, species, genes, make_context=(lambda: abi.Context(0)) if real else OracleContext)
dist.barrier()
if dist.world()[0] == 0:
    sys.stderr.write("LOG " + args['log'].getvalue().replace("\n", " | ") + "\n")
'''


def _run_genes_workers(script, sample, db, n_ranks, extra_env=None):
    """The genes worker as n_ranks processes over gloo -> their stderr texts (rank 0's carries the run's log)."""
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env1.update(extra_env or {})
    port = _free_port()
    procs = []
    for k in range(n_ranks):
        env = dict(env1)
        if n_ranks > 1:
            env.update(RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), sample, db], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env))
    errs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e
        errs.append(e)
    return errs


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_gloo_genes_outputs_equal_the_single_process_ones(tmp_path, n_ranks):
    """run_midas.py genes with N = 2, 3: species dealt to the ranks; every rank decodes ITS slice of the unsorted BAM, the
    (gene, term) pairs travel to the gene's owner (one all-to-all) and are summed there in file order -- the files are the
    single-process files byte for byte, and the log shows that no rank decoded the whole BAM (the device calls are played
    by the oracle here; the GPU tests cover them)."""
    import gzip
    import re
    import shutil
    from midas_amd import synth
    ds = synth.make_pangenome_dataset(n_species=3, genes_per_species=20, n_reads=9000, seed=5)
    one, two, db = str(tmp_path / "one"), str(tmp_path / "two"), str(tmp_path / "db")
    synth.write_pangenome_sample(one, db, ds)
    shutil.copytree(one, two)
    script = tmp_path / "genes_worker.py"
    script.write_text(GENES_WORKER % {"root": ROOT})
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script), one, db], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env1, timeout=300)
    assert r.returncode == 0, r.stderr
    errs = _run_genes_workers(script, two, db, n_ranks)
    m = re.search(r"rank-local BAM decode \(genes\): (\d+) slices chained, (\d+) records; records decoded per rank: ([\d ]+)", errs[0])
    assert m, errs[0]
    per = [int(x) for x in m.group(3).split()]
    total = int(m.group(2))
    assert int(m.group(1)) == n_ranks and total == sum(per) > 6000 and max(per) < total * 0.7      # no rank decoded the whole file
    assert open(os.path.join(two, "genes", "summary.txt")).read() == open(os.path.join(one, "genes", "summary.txt")).read()
    for sp in ds['species_ids']:
        a = gzip.open(os.path.join(one, "genes", "output", sp + ".genes.gz"), "rt").read()
        b = gzip.open(os.path.join(two, "genes", "output", sp + ".genes.gz"), "rt").read()
        assert a == b and a.count("\n") == 21


SNPS_WORKER = r'''
import io, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from midas_amd import abi, dist
from midas_amd.run import snps as msnps
from oracle import c_oracle


class OracleContext:
    """Stands in for the device in this CPU test: midas_snps_pileup's contract, computed by the C oracle."""
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def pileup(self, thr, table, reads):
        st, bad, counts, allele, stats = c_oracle.pileup(thr, table, reads)
        if st != 0:
            raise abi.MidasSnpsError(st, "oracle status %%d" %% st, bad)
        return counts, allele, stats


out, db = sys.argv[1], sys.argv[2]
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    if os.environ.get("SNPS_TRANSPORT") == "native":      # the product's own: the ranks meet in the sample's temp directory, no torch
        dist.init_from_env(rendezvous_dir=os.path.join(out, "snps", "temp"))
        assert "torch" not in sys.modules
    else:
        dist.init_from_env("gloo")
rank, ws = dist.world()
args = dict(outdir=out, db=db, build_db=False, align=False, call=True, species_id=None, threads=2, log=io.StringIO(),
            mapid=94.0, readq=20, mapq=20, baseq=30, aln_cov=0.75, remove_temp=False)
if os.environ.get("SNPS_SPLIT_LENGTH"):
    args['split_length'] = int(os.environ["SNPS_SPLIT_LENGTH"])
if os.environ.get("SNPS_DEVICE_INFLATE"):
    args['device_inflate'] = os.environ["SNPS_DEVICE_INFLATE"]
species = msnps.initialize_species(args)
# (run_pipeline's way at N ranks: the genome files dealt to the ranks, DealtContigs; SNPS_ALL_GENOMES=1: every rank reads them all)
contigs = msnps.initialize_contigs(species) if os.environ.get("SNPS_ALL_GENOMES") or ws == 1 else msnps.ContigsInBackground(species, deal=(rank, ws))
if os.environ.get("SNPS_REAL_DEVICE"):       # (tests/test_gpu_dist.py: the ranks share GPU 0)
    os.environ["LOCAL_RANK"] = "0"
    msnps.pysam_pileup(args, species, contigs)
else:
    msnps.pysam_pileup(args, species, contigs, make_context=OracleContext)
if rank == 0:
    msnps.snps_summary(args, species)
    print("LOG:" + args['log'].getvalue().replace("\n", "|"))
dist.barrier()
dist.finalize()
if os.environ.get("SNPS_TRANSPORT") == "native":
    assert "torch" not in sys.modules
'''


def _run_snps_workers(tmp_path, script, outdir, db, n_ranks, transport=None):
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if transport:
        env1["SNPS_TRANSPORT"] = transport
    if n_ranks == 1:
        r = subprocess.run([sys.executable, str(script), outdir, db], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=env1, timeout=300)
        return [(r.returncode, r.stdout, r.stderr)]
    port = _free_port()
    procs = []
    for k in range(n_ranks):
        env = dict(env1, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), outdir, db], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env))
    res = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        res.append((p.returncode, o, e))
    return res


def test_two_ranks_snps_outputs_equal_single(tmp_path):
    """run_midas.py snps with N = 2 and 3: the contigs are dealt to the ranks, so even the ONE-species sample is split
    (its .snps.gz is the concatenation of the ranks' gzip parts in sorted-contig order) -- every <species>.snps.gz and
    summary.txt is byte for byte the single-process file.  (The device call is played by the C oracle here; the GPU
    tests cover it.)"""
    import shutil
    from midas_amd import synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    for tag, kw in (("one_species", dict(n_species=1, contigs_per_species=5, contig_len=20011, n_reads=6000, seed=11, var_len=True)),
                    ("three_species", dict(n_species=3, contigs_per_species=3, contig_len=17000, n_reads=9000, seed=12))):
        contigs, reads = synth.make_dataset(**kw)
        db = str(tmp_path / (tag + "_db"))
        one = str(tmp_path / (tag + "_n1"))
        synth.write_sample(one, db, contigs, reads)
        (rc, o, e), = _run_snps_workers(tmp_path, script, one, db, 1)
        assert rc == 0, e
        for n in (2, 3):
            many = str(tmp_path / ("%s_n%d" % (tag, n)))
            shutil.copytree(one, many, ignore=shutil.ignore_patterns("output"))
            os.makedirs(os.path.join(many, "snps", "output"))
            res = _run_snps_workers(tmp_path, script, many, db, n)
            for rc, o, e in res:
                assert rc == 0, e
            assert any("rank-local BAM decode: %d slices chained in ONE pass" % n in o for _, o, _ in res)   # every block inflated once, by its owner
            assert open(os.path.join(many, "snps", "summary.txt")).read() == open(os.path.join(one, "snps", "summary.txt")).read()
            files = sorted(os.listdir(os.path.join(one, "snps", "output")))
            assert sorted(os.listdir(os.path.join(many, "snps", "output"))) == files       # no part file left behind
            assert len(files) == contigs.n_species
            for f in files:
                a = open(os.path.join(one, "snps", "output", f), "rb").read()
                b = open(os.path.join(many, "snps", "output", f), "rb").read()
                assert a == b, "%s differs between 1 and %d ranks" % (f, n)


def test_native_transport_ranks_meet_in_the_samples_directory(tmp_path):
    """The product's own transport (midas_amd/dist.py): under RANK / WORLD_SIZE the ranks meet in <outdir>/snps/temp and never
    import torch -- the flags of agree_or_exit, the numbers of the rank-local decode plan and (with no device to form an RCCL
    communicator on: the device is played by the oracle here) the summary rows all travel through files there.  2 and 3 ranks:
    every table and summary.txt byte for byte the single process's, nothing left behind in the meeting place."""
    import shutil
    from midas_amd import synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=3, contig_len=17000, n_reads=9000, seed=12)
    db, one = str(tmp_path / "db"), str(tmp_path / "n1")
    synth.write_sample(one, db, contigs, reads)
    (rc, o, e), = _run_snps_workers(tmp_path, script, one, db, 1)
    assert rc == 0, e
    for n in (2, 3):
        many = str(tmp_path / ("n%d" % n))
        shutil.copytree(one, many, ignore=shutil.ignore_patterns("output"))
        os.makedirs(os.path.join(many, "snps", "output"))
        res = _run_snps_workers(tmp_path, script, many, db, n, transport="native")
        for rc, o, e in res:
            assert rc == 0, e
        assert any("rank-local BAM decode: %d slices chained" % n in o for _, o, _ in res)
        assert open(os.path.join(many, "snps", "summary.txt")).read() == open(os.path.join(one, "snps", "summary.txt")).read()
        for f in sorted(os.listdir(os.path.join(one, "snps", "output"))):
            assert open(os.path.join(one, "snps", "output", f), "rb").read() == open(os.path.join(many, "snps", "output", f), "rb").read(), f
        assert not [d for d in os.listdir(os.path.join(many, "snps", "temp")) if d.startswith("ranks.")]


def test_native_transport_a_failing_rank_takes_the_others_down(tmp_path):
    """agree_or_exit over the files: a rank whose stage fails leaves with its message, the others name it and leave too."""
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
from midas_amd import dist
rank, ws = dist.init_from_env(rendezvous_dir=sys.argv[1])
dist.agree_or_exit(None)
dist.agree_or_exit("\\nError: rank 1 could not read its slice\\n" if rank == 1 else None)
print("not reached")
''' % ROOT)
    meet = tmp_path / "meet"
    meet.mkdir()
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = [subprocess.Popen([sys.executable, str(script), str(meet)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(env1, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"))
             for k in range(3)]
    res = [(p.communicate(timeout=120), p.returncode) for p in procs]
    for k, ((o, e), rc) in enumerate(res):
        assert rc != 0 and "not reached" not in o
        assert ("could not read its slice" in e) if k == 1 else ("rank(s) [1] failed" in e), (k, e)


MEET_WORKER = '''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from midas_amd import dist
rank, ws = dist.init_from_env(rendezvous_dir=sys.argv[1])
dist.agree_or_exit(None)
got = dist.all_gather_i64([rank * 10 + 1, os.getpid()])
assert [int(x) for x in got[:, 0]] == [r * 10 + 1 for r in range(ws)], got
rows = np.zeros((3, 5), np.int64); rows[rank %% 3, :] = rank + 1
tot = dist.all_gather_summary(rows)
assert int(tot.sum()) == 5 * sum(r + 1 for r in range(ws)), tot
dist.barrier()
dist.finalize()
print("met %%d of %%d" %% (rank, ws))
'''


def _meet_env(k, n, port, extra=None):
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "MIDAS_RUN_ID")}
    env.update(RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.update(extra or {})
    return env


def test_native_transport_ignores_what_a_crashed_launch_left_behind(tmp_path):
    """The meeting place has the same name for every launch on one address and port, and a run that was killed leaves its files
    there: a stale list of ranks, a stale failure flag, a stale ncclUniqueId.  Nothing of it may reach the next launch -- a rank
    believes only what names its own living process (midas_amd/dist.py, _Native._meet)."""
    from midas_amd import dist
    script = tmp_path / "w.py"
    script.write_text(MEET_WORKER % ROOT)
    meet = tmp_path / "meet"
    env0 = _meet_env(0, 3, 29500)
    name = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from midas_amd import dist; print(dist._meeting_name())" % ROOT],
                          env=env0, stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
    assert name == "ranks.127.0.0.1_29500"
    place = meet / name
    stale = place / "gen.00deadbeef"
    stale.mkdir(parents=True)
    (place / "current").write_text("gen.00deadbeef\n0 999999 aaaa\n1 999998 bbbb\n2 999997 cccc\n")
    for r in range(3):
        (place / ("hello.%d.%d" % (r, 999999 - r))).write_text("12345 stale")
        (stale / ("0.%d" % r)).write_bytes(b"1")        # "this rank failed" in the first agree_or_exit of the dead run
        (stale / ("1.%d" % r)).write_bytes(b"\x07" * 16)
    procs = [subprocess.Popen([sys.executable, str(script), str(meet)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_meet_env(k, 3, 29500)) for k in range(3)]
    for k, p in enumerate(procs):
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0 and "met %d of 3" % k in o, (k, o, e)
    assert not os.path.exists(str(place))           # the last launch cleared the place, the dead one's generation included


def test_native_transport_ranks_of_different_parents_meet(tmp_path):
    """A launcher that starts every rank through a shell of its own gives the ranks different parent processes: the meeting
    place is named from what the launch tells all of them alike (MASTER_ADDR, MASTER_PORT, the run id), never from a rank's parent."""
    script = tmp_path / "w.py"
    script.write_text(MEET_WORKER % ROOT)
    meet = tmp_path / "meet"
    meet.mkdir()
    procs = [subprocess.Popen(["sh", "-c", "sleep 0.0%d; exec %s %s %s" % (k, sys.executable, script, meet)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=_meet_env(k, 2, 1234, {"MIDAS_RUN_ID": "job 7/a"})) for k in range(2)]
    wrapped = [subprocess.Popen(["sh", "-c", "%s %s %s; true" % (sys.executable, script, meet)], stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True, env=_meet_env(k, 2, 1235)) for k in range(2)]      # (the rank is a CHILD of its shell)
    for k, p in enumerate(procs + wrapped):
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0 and "met %d of 2" % (k % 2) in o, (k, o, e)


def test_native_transport_a_rank_that_meets_nobody_says_where_it_waited(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(MEET_WORKER % ROOT)
    meet = tmp_path / "meet"
    meet.mkdir()
    for k in (0, 1):
        t0 = time.time()
        r = subprocess.run([sys.executable, str(script), str(meet)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=_meet_env(k, 2, 777, {"MIDAS_MEET_TIMEOUT": "1.5"}), timeout=60)
        assert r.returncode != 0 and time.time() - t0 < 30
        assert "waited" in r.stderr and str(meet) in r.stderr and "MASTER_ADDR" in r.stderr, r.stderr


ATTACH_WORKER = '''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from midas_amd import abi, dist
rank, ws = dist.init_from_env(rendezvous_dir=sys.argv[1])
mode = sys.argv[2]
made = []
class FakeComm:                      # stands in for the RCCL binding: what attach() does around it is what is under test
    def __init__(self, ctx, ident, r, w):
        assert ident == b"I" * 128 and (r, w) == (rank, ws)
        if mode == "create" and r == 1:
            raise abi.MidasSnpsError(abi.ERR_HIP, "ncclCommInitRank: unhandled system error")
        made.append(self); self.open = True
    def close(self): self.open = False
    def all_gather(self, data): raise AssertionError("a communicator that did not come up everywhere must not be used")
    @staticmethod
    def device_key(ctx): return "0000:%%02x:00.0" %% rank
    @staticmethod
    def probe():
        if mode == "probe" and rank == 2:
            raise abi.MidasSnpsError(abi.ERR_UNSUPPORTED, "librccl.so not found: no such file")
        return 22203
    @staticmethod
    def unique_id(): return b"I" * 128
abi.Comm = FakeComm
class Ctx: _h = 1
line = dist.attach_context(Ctx())
assert dist._native.comm is None and all(not c.open for c in made), line
assert "stays on files" in line and ("rank 2: librccl.so not found" in line if mode == "probe" else "unhandled system error" in line), line
assert (mode == "probe" or rank == 1) == (not made), (mode, made)          # probe failure: NO rank entered ncclCommInitRank
rows = np.zeros((2, 5), np.int64); rows[rank %% 2] = rank + 1
assert int(dist.all_gather_summary(rows).sum()) == 5 * sum(r + 1 for r in range(ws))
dist.barrier(); dist.finalize()
print("fell back together")
'''


@pytest.mark.parametrize("mode", ["probe", "create"])
def test_native_transport_falls_back_to_files_together_when_rccl_fails_on_one_rank(tmp_path, mode):
    """One rank cannot load RCCL (found before anybody enters ncclCommInitRank), or its ncclCommInitRank fails: EVERY rank drops
    the communicator and the summary rows travel through the files -- nobody waits inside RCCL, nobody uses half a communicator."""
    script = tmp_path / "w.py"
    script.write_text(ATTACH_WORKER % ROOT)
    meet = tmp_path / "meet"
    meet.mkdir()
    procs = [subprocess.Popen([sys.executable, str(script), str(meet), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_meet_env(k, 3, 4242)) for k in range(3)]
    for k, p in enumerate(procs):
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0 and "fell back together" in o, (k, o, e)


def test_native_transport_a_rank_that_dies_without_a_word_is_noticed_at_once(tmp_path):
    """A rank killed between two exchanges (no sys.exit message, no flag): the ranks waiting for it know its process from the
    meeting and stop within a second of its death -- not at the collective's 48 h deadline."""
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys, time
sys.path.insert(0, %r)
from midas_amd import dist
rank, ws = dist.init_from_env(rendezvous_dir=sys.argv[1])
dist.agree_or_exit(None)
if rank == 1:
    os._exit(9)
dist.all_gather_i64([rank])
print("not reached")
''' % ROOT)
    meet = tmp_path / "meet"
    meet.mkdir()
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script), str(meet)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_meet_env(k, 3, 4343)) for k in range(3)]
    for k, p in enumerate(procs):
        o, e = p.communicate(timeout=60)
        assert p.returncode != 0 and "not reached" not in o
        assert k == 1 or "rank 1 (process %d) is gone" % procs[1].pid in e, (k, e)
    assert time.time() - t0 < 30


def test_a_launch_over_several_nodes_does_not_take_the_native_transport():
    """LOCAL_WORLD_SIZE below WORLD_SIZE: the ranks cannot meet in a directory -- init_from_env goes on to the torch process group."""
    code = ("import sys; sys.path.insert(0, %r)\nfrom midas_amd import dist\n"
            "import torch.distributed as td\n"
            "calls = []\ntd.init_process_group = lambda *a, **k: calls.append((a, k))\n"
            "dist.world = lambda: (0, 4)\n"
            "dist.init_from_env(rendezvous_dir='/nonexistent/never/made')\n"
            "assert dist._native is None and calls, calls\nprint('torch group asked for')\n") % ROOT
    env = _meet_env(0, 4, 5, {"LOCAL_WORLD_SIZE": "2"})
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "torch group asked for" in r.stdout, r.stderr


def test_one_long_contig_is_cut_into_pieces_across_ranks(tmp_path):
    """One 20 Mb contig (a finished chromosome) and nothing else: whole contigs as work items would leave every rank but one
    idle.  With 2 and 3 ranks the contig is dealt out in pieces; the table and the summary are byte for byte the single
    process's, and the log shows that no rank decoded the contig's records whole (each inflates its pieces' ranges only)."""
    import re
    import shutil
    from midas_amd import synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=20000000, n_reads=150000, seed=77, var_len=True)
    db, one = str(tmp_path / "db"), str(tmp_path / "n1")
    synth.write_sample(one, db, contigs, reads)
    (rc, o, e), = _run_snps_workers(tmp_path, script, one, db, 1)
    assert rc == 0, e
    os.environ["SNPS_SPLIT_LENGTH"] = str(3 << 20)
    try:
        for n in (2, 3):
            many = str(tmp_path / ("n%d" % n))
            shutil.copytree(one, many, ignore=shutil.ignore_patterns("output"))
            os.makedirs(os.path.join(many, "snps", "output"))
            res = _run_snps_workers(tmp_path, script, many, db, n)
            for rc, o, e in res:
                assert rc == 0, e
            log = "".join(o for _, o, _ in res)
            m = re.search(r"long contigs: 1 cut into pieces of (\d+) positions; records decoded per rank: ([\d ]+) of (\d+)", log)
            assert m, log
            assert int(m.group(1)) == 3 << 20
            per_rank, total = [int(x) for x in m.group(2).split()], int(m.group(3))
            assert len(per_rank) == n and total == reads.n_reads
            assert max(per_rank) < 0.75 * total and sum(per_rank) >= total          # (the sum holds every halo twice)
            assert sum(per_rank) < 1.1 * total
            assert open(os.path.join(many, "snps", "summary.txt")).read() == open(os.path.join(one, "snps", "summary.txt")).read()
            files = sorted(os.listdir(os.path.join(one, "snps", "output")))
            assert sorted(os.listdir(os.path.join(many, "snps", "output"))) == files
            for f in files:
                assert open(os.path.join(one, "snps", "output", f), "rb").read() == open(os.path.join(many, "snps", "output", f), "rb").read()
    finally:
        del os.environ["SNPS_SPLIT_LENGTH"]


def mixed_sample():
    """One species of a single 6 Mb contig and two of five 40 kb contigs each: pieces and whole contigs in one rank's table."""
    from midas_amd import abi, synth
    big, big_reads = synth.make_dataset(n_species=1, contigs_per_species=1, contig_len=6000000, n_reads=120000, seed=91, var_len=True)
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=5, contig_len=40000, n_reads=30000, seed=92)
    table = abi.ContigTable(length=np.concatenate([big.length, contigs.length]), species=[0] + [1 + int(s) for s in contigs.species],
                            read_begin=np.concatenate([big.read_begin, contigs.read_begin[1:] + big.read_begin[-1]]),
                            ref=np.concatenate([big.ref, contigs.ref]), n_species=3,
                            ids=["Big_00001_chromosome"] + list(contigs.ids), species_ids=["Big_00001"] + list(contigs.species_ids))
    return table, synth.concat_reads([big_reads, reads])


def test_pieces_and_whole_contigs_in_one_ranks_table(tmp_path):
    import shutil
    from midas_amd import synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    table, rd = mixed_sample()
    db, one, two = str(tmp_path / "db"), str(tmp_path / "n1"), str(tmp_path / "n2")
    synth.write_sample(one, db, table, rd)
    (rc, o, e), = _run_snps_workers(tmp_path, script, one, db, 1)
    assert rc == 0, e
    shutil.copytree(one, two, ignore=shutil.ignore_patterns("output"))
    os.makedirs(os.path.join(two, "snps", "output"))
    os.environ["SNPS_SPLIT_LENGTH"] = str(1 << 20)
    try:
        res = _run_snps_workers(tmp_path, script, two, db, 2)
    finally:
        del os.environ["SNPS_SPLIT_LENGTH"]
    assert all(rc == 0 for rc, _, _ in res), "\n".join(e[-1500:] for _, _, e in res)
    assert any("long contigs: 1 cut into pieces of 1048576 positions" in o for _, o, _ in res)
    for f in sorted(os.listdir(os.path.join(one, "snps", "output"))):
        assert open(os.path.join(one, "snps", "output", f), "rb").read() == open(os.path.join(two, "snps", "output", f), "rb").read()
    assert open(os.path.join(one, "snps", "summary.txt")).read() == open(os.path.join(two, "snps", "summary.txt")).read()


def test_two_ranks_on_a_bam_that_is_not_coordinate_sorted_fall_back_to_the_whole_decode(tmp_path):
    """Contigs written to the BAM in reverse order of the header: the slices cannot vouch for record ranges per contig, so
    every rank decodes the whole file (as one rank does) -- same files again."""
    import shutil
    from midas_amd import bam, synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=15000, n_reads=6000, seed=21)
    db, one, two = str(tmp_path / "db"), str(tmp_path / "n1"), str(tmp_path / "n2")
    synth.write_sample(one, db, contigs, reads)
    # rewrite the BAM with the records of the contigs in reverse contig order (within a contig still by position)
    from tests.test_gpu_parity import _subset
    rb = contigs.read_begin
    order = np.concatenate([np.arange(rb[c], rb[c + 1]) for c in reversed(range(contigs.n_contigs))])
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(rb))[order]
    bam.write_bam(os.path.join(one, "snps", "temp", "genomes.bam"), contigs.ids, [int(x) for x in contigs.length], refid,
                  _subset(reads, order))
    shutil.copytree(one, two)
    (rc, o, e), = _run_snps_workers(tmp_path, script, one, db, 1)
    assert rc == 0, e
    res = _run_snps_workers(tmp_path, script, two, db, 2)
    for rc, o, e in res:
        assert rc == 0, e
    assert not any("rank-local BAM decode" in o for _, o, _ in res)
    for f in sorted(os.listdir(os.path.join(one, "snps", "output"))):
        assert open(os.path.join(one, "snps", "output", f), "rb").read() == open(os.path.join(two, "snps", "output", f), "rb").read()
    assert open(os.path.join(one, "snps", "summary.txt")).read() == open(os.path.join(two, "snps", "summary.txt")).read()


def test_a_failing_rank_takes_every_rank_down_with_its_message(tmp_path):
    """A read the reference would raise on (no NM tag) sits on one rank's contig: both ranks exit non-zero, promptly, and
    the failing one says why -- nobody is left blocking in the all-gather."""
    from midas_amd import synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=4, contig_len=9000, n_reads=2000, seed=13)
    reads.nm[5] = -1
    db, out = str(tmp_path / "db"), str(tmp_path / "out")
    synth.write_sample(out, db, contigs, reads)
    res = _run_snps_workers(tmp_path, script, out, db, 2)
    assert all(rc != 0 for rc, _, _ in res)
    assert sum("NM tag" in e for _, _, e in res) == 1
    assert sum("stops with them" in e for _, _, e in res) == 1


MERGE_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "scripts"))
from midas_amd import abi, dist
from midas_amd.merge import snps as msnps
from tests.test_gpu_merge import oracle_fields
import merge_midas


class OracleContext:
    """Stands in for the device in this CPU test: midas_merge_sites' contract, computed by the restated reference."""
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def merge_sites(self, prm, counts, mean_depth):
        names = [t for t, bit in abi.SNP_TYPE_BITS.items() if prm.snp_types & bit]
        args = dict(allele_freq=prm.allele_freq, site_depth=prm.site_depth, site_ratio=prm.site_ratio, site_prev=prm.site_prev,
                    snp_type=names)
        out = oracle_fields([np.asarray(c) for c in counts], [float(x) for x in mean_depth], args)
        out['kernel_ms'] = 0.0
        return out


sys.argv = ['merge_midas.py'] + sys.argv[1:]
merge_midas.get_program()
args = merge_midas.snps_arguments()
merge_midas.check_arguments(args)
msnps.run_pipeline(args, make_context=OracleContext)
'''


def test_ranks_share_one_species_merge_by_site_range(tmp_path):
    """merge_midas.py snps with N = 2 and 3 on ONE species: the sample tables (written by this library's writer) say how
    many rows every gzip member holds, so rank r reads, merges and writes rows [n r / N, n (r+1) / N) only and the parts are
    concatenated -- snps_info / snps_freq / snps_depth are byte for byte the single-process files.  Tables that do not
    say (written like the reference writes them) make the species go whole to one rank: same files again."""
    import shutil
    from midas_amd import abi, synth
    script = tmp_path / "merge_worker.py"
    script.write_text(MERGE_WORKER % {"root": ROOT})
    ds = synth.make_merge_dataset(str(tmp_path / "ds"), n_samples=3, n_sites=50000, n_contigs=2, seed=5)
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    samples_dir = os.path.dirname(ds['samples'][0])

    def run(out, n_ranks):
        argv = [sys.executable, str(script), 'snps', out, '-i', samples_dir, '-t', 'dir', '-d', ds['db'], '--all_snps', '--threads', '2']
        if n_ranks == 1:
            r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env1, timeout=600)
            assert r.returncode == 0, r.stderr
            return r.stdout
        port = _free_port()
        procs = [subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  env=dict(env1, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                                           MASTER_PORT=str(port))) for k in range(n_ranks)]
        outs = []
        for p in procs:
            o, e = p.communicate(timeout=600)
            assert p.returncode == 0, e
            outs.append(o)
        return "".join(outs)

    def same(a, b):
        names = sorted(os.listdir(os.path.join(a, ds['species_id'])))
        assert sorted(os.listdir(os.path.join(b, ds['species_id']))) == names and 'snps_info.txt' in names
        for f in names:
            assert open(os.path.join(a, ds['species_id'], f), 'rb').read() == open(os.path.join(b, ds['species_id'], f), 'rb').read(), f

    # (1) tables as the reference writes them: one gzip stream, no row counts -> the species goes whole to one rank
    one = str(tmp_path / "ref_n1")
    run(one, 1)
    two = str(tmp_path / "ref_n2")
    assert "rows " not in run(two, 2)
    same(one, two)
    # (2) the same tables written by this library's writer (25 000-row contigs: two members each)
    off = 0
    for sdir, c in zip(ds['samples'], ds['counts']):
        alleles, counts = [], []
        off = 0
        for seq in ds['contig_seqs']:
            alleles.append(np.frombuffer(seq.encode(), np.uint8))
            counts.append(np.ascontiguousarray(c[off:off + len(seq)], np.uint32))
            off += len(seq)
        path = os.path.join(sdir, 'snps', 'output', ds['species_id'] + '.snps.gz')
        abi.write_table(path, ds['contig_ids'], alleles, counts, gz_level=1, threads=2)
        assert abi.count_snps_rows(path) == 50000
    for n in (2, 3):
        many = str(tmp_path / ("own_n%d" % n))
        log = run(many, n)
        assert log.count("rows ") == n          # every rank took its range of the one species
        same(one, many)
