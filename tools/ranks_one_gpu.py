"""Developer script (GPU box): the multi-rank PRODUCT path of `run_midas.py snps --pileup` as 1, 2 and 3 processes sharing ONE GPU
(the product's own transport: the ranks meet in the sample's temp directory and never import torch; they share a device, which
RCCL refuses, so even the summary rows go through the files -- RANKS_TRANSPORT=gloo runs the torch.distributed form beside it), each with the CPU budget of one rank of an 8-rank node
(LOCAL_WORLD_SIZE=8: a quota of 16 CPUs leaves 2 per rank) -- per-phase wall times of every rank: slice walk + exchange,
decode of the rank's record ranges, contig table / batch / pileup / rows + write, the summary all-gather, joining the parts.
usage: python tools/ranks_one_gpu.py [config] [workdir]"""
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from midas_amd import synth  # noqa: E402

WORKER = r'''
import io, os, sys, time
sys.path.insert(0, %(root)r)
t_start = time.perf_counter()
from midas_amd import abi, dist
from midas_amd.run import snps as msnps
T = {}
def timed(mod, name, label):
    f = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label] = T.get(label, 0.0) + time.perf_counter() - t
    setattr(mod, name, w)
timed(msnps, '_rank_local_plan', 'slice walk + exchange')
timed(msnps, '_pileup_contigs', 'table, batch, pileup, rows + write')
timed(dist, 'all_gather_summary', 'summary all-gather')
timed(msnps, '_join_parts', 'join parts')
timed(msnps, '_count_alleles', 'pileup stage (all of it)')
out, db = sys.argv[1], sys.argv[2]
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    if os.environ.get("RANKS_TRANSPORT", "native") == "native":
        dist.init_from_env(rendezvous_dir=os.path.join(out, "snps", "temp"))
    else:
        dist.init_from_env("gloo")
rank, ws = dist.world()
args = dict(outdir=out, db=db, build_db=False, align=False, call=True, species_id=None, threads=0, log=io.StringIO(),
            mapid=94.0, readq=20, mapq=20, baseq=30, aln_cov=0.75, remove_temp=False, device_inflate='auto')
t = time.perf_counter()
species = msnps.initialize_species(args)
contigs = msnps.ContigsInBackground(species)
os.environ["LOCAL_RANK"] = "0"          # every rank on GPU 0
msnps.pysam_pileup(args, species, contigs)
if rank == 0:
    msnps.snps_summary(args, species)
dist.barrier()
dist.finalize()
stage = T.pop('pileup stage (all of it)')
rest = stage - sum(T.values())
print("RANK %%d of %%d (cpu budget %%d): stage %%.3f s | %%s | decode of own ranges + the rest %%.3f s | imports + start %%.2f s | torch imported: %%s" %% (
    rank, ws, __import__('midas_amd.utility', fromlist=['x']).cpu_budget(), stage, " | ".join("%%s %%.3f s" %% kv for kv in T.items()), rest, t - t_start, "torch" in sys.modules), flush=True)
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/midas_ranks'
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    contigs, reads = synth.make_dataset(**synth.CONFIGS[cfg])
    out, db = os.path.join(work, 'sample'), os.path.join(work, 'db')
    synth.write_sample(out, db, contigs, reads)
    print("sample %s: %d sites, %d reads, BAM %.2f GB" % (cfg, contigs.n_sites, reads.n_reads, os.path.getsize(os.path.join(out, 'snps/temp/genomes.bam')) / 1e9), flush=True)
    script = os.path.join(work, 'worker.py')
    open(script, 'w').write(WORKER % {"root": os.path.abspath(ROOT)})
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    for n, lws, transport in ((1, None, "native"), (1, 8, "native"), (2, 8, "native"), (3, 8, "native"), (2, 8, "gloo"), (3, 8, "gloo"), (2, 8, "native")):
        shutil.rmtree(os.path.join(out, 'snps', 'output'), ignore_errors=True)
        os.makedirs(os.path.join(out, 'snps', 'output'))
        port = free_port()
        t = time.perf_counter()
        procs = []
        for k in range(n):
            env = dict(base, RANKS_TRANSPORT=transport)
            if lws:
                env["LOCAL_WORLD_SIZE"] = str(lws)
            if n > 1:
                env.update(RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, script, out, db], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
        res = [p.communicate(timeout=900) + (p.returncode,) for p in procs]
        dt = time.perf_counter() - t
        print("---- %d process(es) on one GPU, LOCAL_WORLD_SIZE=%s, transport %s: %.2f s wall (start of the first to exit of the last)" % (n, lws, transport, dt), flush=True)
        for o, e, rc in res:
            if rc != 0:
                print("  FAILED rc %d: %s" % (rc, e[-1500:]))
            for line in o.splitlines():
                if line.startswith("RANK"):
                    print("  " + line)
        sz = sum(os.path.getsize(os.path.join(out, 'snps/output', f)) for f in os.listdir(os.path.join(out, 'snps/output')))
        print("  tables: %d, %.0f MB" % (len(os.listdir(os.path.join(out, 'snps/output'))), sz / 1e6), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
