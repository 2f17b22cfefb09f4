"""`run_midas.py snps` as 2 and 3 processes on ONE real GPU (the ranks talk over gloo and share device 0): the whole
multi-rank product path -- rank-local BAM decode, contigs and pieces of a long contig dealt to the ranks, the direct device
path, the device's row coder, parts concatenated -- against one process on the same GPU, byte for byte, and against the
CPU double of the device (the C oracle behind the host's formatter) in text."""
import gzip
import os
import shutil

import pytest

from tests.test_dist_gloo import GENES_WORKER, ROOT, SNPS_WORKER, _run_genes_workers, _run_snps_workers

pytestmark = pytest.mark.gpu


def test_ranks_sharing_one_gpu_write_the_single_process_files(tmp_path):
    from midas_amd import synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    from tests.test_dist_gloo import mixed_sample
    table, rd = mixed_sample()
    db, cpu = str(tmp_path / "db"), str(tmp_path / "cpu")
    synth.write_sample(cpu, db, table, rd)
    (rc, o, e), = _run_snps_workers(tmp_path, script, cpu, db, 1)          # the CPU double, one process
    assert rc == 0, e
    env = {"SNPS_REAL_DEVICE": "1", "SNPS_SPLIT_LENGTH": str(1 << 20)}
    os.environ.update(env)
    try:
        outs = {}
        # (11, 12: one and two ranks again with the BGZF blocks inflated on the device -- one rank also leaves SEQ / QUAL / CIGAR
        # there, two ranks inflate their slices' blocks)
        for n in (1, 2, 3, 11, 12):
            os.environ["SNPS_DEVICE_INFLATE"] = "on" if n > 10 else "off"
            tag, n = n, (n - 10 if n > 10 else n)
            d = str(tmp_path / ("gpu_n%d" % tag))
            shutil.copytree(cpu, d, ignore=shutil.ignore_patterns("output"))
            os.makedirs(os.path.join(d, "snps", "output"))
            res = _run_snps_workers(tmp_path, script, d, db, n)
            assert all(rc == 0 for rc, _, _ in res), "\n".join("rank %d: rc %d\n%s" % (k, rc, e[-1500:]) for k, (rc, _, e) in enumerate(res))
            if n > 1:
                assert any("long contigs: 1 cut into pieces" in o for _, o, _ in res)
            outs[tag] = d
    finally:
        for k in list(env) + ["SNPS_DEVICE_INFLATE"]:
            os.environ.pop(k, None)
    files = sorted(os.listdir(os.path.join(cpu, "snps", "output")))
    assert len(files) == table.n_species
    for n in (1, 2, 3, 11, 12):
        assert sorted(os.listdir(os.path.join(outs[n], "snps", "output"))) == files
        assert open(os.path.join(outs[n], "snps", "summary.txt")).read() == open(os.path.join(cpu, "snps", "summary.txt")).read()
        for f in files:
            got = open(os.path.join(outs[n], "snps", "output", f), "rb").read()
            assert got == open(os.path.join(outs[1], "snps", "output", f), "rb").read(), "%s: %d ranks vs 1" % (f, n)
    for f in files:          # device coder vs host coder: the same text
        assert gzip.open(os.path.join(outs[1], "snps", "output", f), "rb").read() == gzip.open(os.path.join(cpu, "snps", "output", f), "rb").read()


def test_device_payload_with_records_not_grouped_by_reference(tmp_path):
    """--device_inflate on for a BAM whose contigs are written in reverse order: the payload columns left on the device have to
    come down again for the regroup -- same files as with the host inflating."""
    import numpy as np
    from midas_amd import bam, synth
    from tests.test_gpu_parity import _subset
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=15000, n_reads=6000, seed=21)
    db, a, b = str(tmp_path / "db"), str(tmp_path / "host"), str(tmp_path / "dev")
    synth.write_sample(a, db, contigs, reads)
    rb = contigs.read_begin
    order = np.concatenate([np.arange(rb[c], rb[c + 1]) for c in reversed(range(contigs.n_contigs))])
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(rb))[order]
    bam.write_bam(os.path.join(a, "snps", "temp", "genomes.bam"), contigs.ids, [int(x) for x in contigs.length], refid, _subset(reads, order))
    shutil.copytree(a, b)
    os.environ["SNPS_REAL_DEVICE"] = "1"
    try:
        for d, how in ((a, "off"), (b, "on")):
            os.environ["SNPS_DEVICE_INFLATE"] = how
            (rc, o, e), = _run_snps_workers(tmp_path, script, d, db, 1)
            assert rc == 0, e[-1500:]
    finally:
        for k in ("SNPS_REAL_DEVICE", "SNPS_DEVICE_INFLATE"):
            os.environ.pop(k, None)
    for f in sorted(os.listdir(os.path.join(a, "snps", "output"))):
        assert open(os.path.join(a, "snps", "output", f), "rb").read() == open(os.path.join(b, "snps", "output", f), "rb").read()
    assert open(os.path.join(a, "snps", "summary.txt")).read() == open(os.path.join(b, "snps", "summary.txt")).read()


def test_genes_ranks_sharing_one_gpu_write_the_single_process_files(tmp_path):
    """`run_midas.py genes` below the species as 2 and 3 processes on one real GPU: every rank decodes its slice of the
    unsorted BAM, makes its reads' terms on the device (midas_genes_terms), the pairs travel to the genes' owners (gloo
    all-to-all) and are sorted and summed there (midas_genes_sum) -- the tables and the summary are those of one process on
    the device and of the CPU double (the oracle), byte for byte."""
    import re
    from midas_amd import synth
    ds = synth.make_pangenome_dataset(n_species=5, genes_per_species=60, n_reads=90000, seed=17)
    cpu, db = str(tmp_path / "cpu"), str(tmp_path / "db")
    synth.write_pangenome_sample(cpu, db, ds)
    script = tmp_path / "genes_worker.py"
    script.write_text(GENES_WORKER % {"root": ROOT})
    _run_genes_workers(script, cpu, db, 1)                                    # the CPU double, one process
    outs = {}
    for n in (1, 2, 3):
        d = str(tmp_path / ("gpu_n%d" % n))
        shutil.copytree(cpu, d, ignore=shutil.ignore_patterns("output", "summary.txt"))
        os.makedirs(os.path.join(d, "genes", "output"), exist_ok=True)
        errs = _run_genes_workers(script, d, db, n, {"GENES_REAL_DEVICE": "1"})
        if n > 1:
            m = re.search(r"rank-local BAM decode \(genes\): (\d+) slices chained, (\d+) records; records decoded per rank: ([\d ]+)", errs[0])
            assert m, errs[0][-2000:]
            per = [int(x) for x in m.group(3).split()]
            assert len(per) == n and max(per) < 0.7 * sum(per)
        outs[n] = d
    for n in (1, 2, 3):
        assert open(os.path.join(outs[n], "genes", "summary.txt")).read() == open(os.path.join(cpu, "genes", "summary.txt")).read()
        for sp in ds['species_ids']:
            a = gzip.open(os.path.join(cpu, "genes", "output", sp + ".genes.gz"), "rb").read()
            assert gzip.open(os.path.join(outs[n], "genes", "output", sp + ".genes.gz"), "rb").read() == a, "%s: %d ranks" % (sp, n)


def test_the_librarys_rccl_binding_on_a_one_rank_communicator():
    """midas_comm_* (comm.cpp): ncclGetUniqueId / ncclCommInitRank / ncclAllGather / grouped ncclSend + ncclRecv of librccl.so,
    loaded at run time, on the one GPU this box has (RCCL refuses two ranks on one device: the N-rank form is the driver's
    8-GPU run).  No torch anywhere in the process."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from midas_amd import abi
ctx = abi.Context(0)
ident = abi.Comm.unique_id()
assert len(ident) == 128 and any(ident)
comm = abi.Comm(ctx, ident, 0, 1)
rows = np.arange(100 * 5, dtype=np.int64).reshape(100, 5) * 3 - 7
got = comm.all_gather(rows.tobytes())
assert len(got) == 1 and np.array_equal(np.frombuffer(got[0], np.int64).reshape(100, 5), rows)
blob = np.random.default_rng(1).integers(0, 255, 1 << 20, dtype=np.uint8).tobytes()
back = comm.all_to_all_v([blob], [len(blob)])
assert back == [blob]
assert comm.all_to_all_v([b""], [0]) == [b""]
comm.close()
ctx.close()
assert "torch" not in sys.modules
print("rccl ok", abi.Comm.device_key(abi.Context(0)))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_native_transport_with_ranks_sharing_one_gpu(tmp_path):
    """The product's own transport on a real device: 2 and 3 ranks meet in the sample's temp directory, find that they share
    one GPU (the log says so: RCCL refuses that) and keep to the files for the summary rows too -- the tables and summary.txt
    are byte for byte the single process's, and torch is never imported (the worker asserts it)."""
    from midas_amd import synth
    script = tmp_path / "snps_worker.py"
    script.write_text(SNPS_WORKER % {"root": ROOT})
    contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=3, contig_len=17000, n_reads=9000, seed=12)
    db, one = str(tmp_path / "db"), str(tmp_path / "n1")
    synth.write_sample(one, db, contigs, reads)
    os.environ["SNPS_REAL_DEVICE"] = "1"
    try:
        (rc, o, e), = _run_snps_workers(tmp_path, script, one, db, 1)
        assert rc == 0, e[-1500:]
        for n in (2, 3):
            many = str(tmp_path / ("n%d" % n))
            shutil.copytree(one, many, ignore=shutil.ignore_patterns("output"))
            os.makedirs(os.path.join(many, "snps", "output"))
            res = _run_snps_workers(tmp_path, script, many, db, n, transport="native")
            assert all(rc == 0 for rc, _, _ in res), "\n".join("rank %d: rc %d\n%s" % (k, rc, e[-1500:]) for k, (rc, _, e) in enumerate(res))
            assert any("ranks share a device" in o for _, o, _ in res)
            assert open(os.path.join(many, "snps", "summary.txt")).read() == open(os.path.join(one, "snps", "summary.txt")).read()
            for f in sorted(os.listdir(os.path.join(one, "snps", "output"))):
                assert open(os.path.join(one, "snps", "output", f), "rb").read() == open(os.path.join(many, "snps", "output", f), "rb").read(), f
    finally:
        os.environ.pop("SNPS_REAL_DEVICE", None)


def _device_count():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return int(n.value) if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.skipif(_device_count() < 2, reason="one GPU on this box: RCCL refuses two ranks on one device (the first multi-GPU lease runs this)")
@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_the_products_transport_between_ranks_on_their_own_gpus(tmp_path, n_ranks):
    """The product's own exchange between REAL ranks -- one process per GPU, met in a directory, an RCCL communicator formed
    through the library's binding (midas_comm_*), the summary rows all-gathered and a ragged all-to-all over xGMI -- on the first
    box that has the devices.  No torch in the ranks."""
    import subprocess
    import sys
    if _device_count() < n_ranks:
        pytest.skip("%d GPUs here" % _device_count())
    script = tmp_path / "w.py"
    script.write_text(r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from midas_amd import abi, dist
rank, ws = dist.init_from_env(rendezvous_dir=sys.argv[1])
ctx = abi.Context(int(os.environ["LOCAL_RANK"]))
line = dist.attach_context(ctx)
assert dist._native.comm is not None and "RCCL communicator of %%d ranks" %% ws in line, line
rows = np.zeros((100, 5), np.int64)
rows[rank::ws] = np.arange(5) + 1000 * (rank + 1)
tot = dist.all_gather_summary(rows)
want = np.zeros((100, 5), np.int64)
for r in range(ws):
    want[r::ws] = np.arange(5) + 1000 * (r + 1)
assert np.array_equal(tot, want)
rng = np.random.default_rng(rank)
parts = [rng.integers(0, 1 << 40, 1000 * ((rank + d) %% 3) + d, dtype=np.int64) for d in range(ws)]
got = dist.all_to_all_v(parts)
for src in range(ws):
    theirs = np.random.default_rng(src)
    sent = [theirs.integers(0, 1 << 40, 1000 * ((src + d) %% 3) + d, dtype=np.int64) for d in range(ws)]
    assert np.array_equal(got[src], sent[rank]), (rank, src)
f = dist.all_gather_rows_f64(np.full((3, 2), rank + 0.5))
assert float(f.sum()) == 6 * sum(r + 0.5 for r in range(ws))
dist.detach_context()
ctx.close()
dist.barrier(); dist.finalize()
assert "torch" not in sys.modules
print("rccl between %%d ranks ok" %% ws)
''' % ROOT)
    meet = tmp_path / "meet"
    meet.mkdir()
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    procs = [subprocess.Popen([sys.executable, str(script), str(meet)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(env1, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT="29611",
                                       HSA_ENABLE_IPC_MODE_LEGACY="0")) for k in range(n_ranks)]
    for k, p in enumerate(procs):
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0 and "rccl between %d ranks ok" % n_ranks in o, (k, o, e[-3000:])
