"""Developer script (GPU box with ONE GPU): what a rank of an N-GPU job does, measured rank by rank.

BASELINE.json configs[3] (or any config) through the files as N processes of the product path (`pysam_pileup` under the native
transport: the ranks meet in the sample's temp directory, never import torch), every rank with the CPU budget of one rank of an
N-rank node (LOCAL_WORLD_SIZE=N) -- but on ONE device, so the ranks take TURNS on it: a file lock around a rank's device-heavy
calls (its BAM share's decode, its contigs' pileup + rows).  A rank's own phases are then those of an uncontended GPU; the time
it spends waiting for its turn or for the other ranks at an exchange (who wait for THEIR turn) is measured and left out.
Prediction for N GPUs: max over ranks of (process start -> pipeline + its own phases).  What the prediction cannot hold: N
uploads sharing the host's memory system, N ranks writing their tables at once, RCCL's communicator start (the ranks share a
device here and stay on the files).
usage: python tools/predict_ranks.py [config=c4] [ranks=8] [workdir]"""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from midas_amd import synth  # noqa: E402

WORKER = r'''
import fcntl, io, os, sys, time
sys.path.insert(0, %(root)r)
t_start = time.perf_counter()
from midas_amd import abi, dist
from midas_amd.run import snps as msnps
t_import = time.perf_counter() - t_start
OWN, WAIT = {}, {}
lock = open(sys.argv[3], "w")
def wrap(owner, name, label, turn=False, waiting=False):
    f = getattr(owner, name)
    def w(*a, **k):
        if turn:
            t = time.perf_counter()
            fcntl.flock(lock, fcntl.LOCK_EX)
            WAIT["its turn on the device"] = WAIT.get("its turn on the device", 0.0) + time.perf_counter() - t
        t, w0 = time.perf_counter(), sum(WAIT.values())
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t
            if waiting:
                WAIT[label] = WAIT.get(label, 0.0) + dt
            else:       # (what the call waited for inside -- an exchange, another call's turn -- is not its own time)
                OWN[label] = OWN.get(label, 0.0) + dt - (sum(WAIT.values()) - w0)
            if turn:
                fcntl.flock(lock, fcntl.LOCK_UN)
    setattr(owner, name, w)
wrap(abi.BamSlice, 'load_ranges', "decode of its share (resident)", turn=True)
wrap(msnps, '_pileup_contigs', "contig table, batch, pileup, rows + write", turn=True)
wrap(msnps, '_one_pass_shares', "share offsets (host walk + exchange)")
wrap(msnps.DealtContigs, 'need', "genomes read late")
wrap(msnps.ContigsInBackground, 'wait', "genomes waited for")
wrap(msnps, '_header_lengths', "BAM header")
wrap(msnps, '_join_parts', "parts joined")
for name in ('agree_or_exit', 'all_gather_i64', 'all_gather_blob', 'all_gather_summary', 'barrier', 'attach_context'):
    wrap(dist, name, "the other ranks at an exchange", waiting=True)
out, db = sys.argv[1], sys.argv[2]
rank, ws = dist.init_from_env(rendezvous_dir=os.path.join(out, "snps", "temp"))
args = dict(outdir=out, db=db, build_db=False, align=False, call=True, species_id=None, threads=0, log=io.StringIO(),
            mapid=94.0, readq=20, mapq=20, baseq=30, aln_cov=0.75, remove_temp=False, device_inflate='on')
species = msnps.initialize_species(args)
contigs = msnps.ContigsInBackground(species, deal=(rank, ws))
os.environ["LOCAL_RANK"] = "0"          # every rank on GPU 0
t_pipe = time.perf_counter()
t = time.perf_counter()
fcntl.flock(lock, fcntl.LOCK_EX)        # (the context's creation touches the device too)
t_wait_ctx = time.perf_counter() - t
t = time.perf_counter()
ctx = abi.Context(0)
t_ctx = time.perf_counter() - t
fcntl.flock(lock, fcntl.LOCK_UN)
class Ready:
    def __enter__(self): return ctx
    def __exit__(self, *a): ctx.close(); return False
t = time.perf_counter()
msnps.pysam_pileup(args, species, contigs, make_context=Ready)
stage = time.perf_counter() - t
if rank == 0:
    msnps.snps_summary(args, species)
dist.barrier(); dist.finalize()
waits = sum(WAIT.values())
own = stage - waits
rest = own - sum(OWN.values())
print("RANK %%d of %%d (cpu budget %%d): start -> pipeline %%.3f s (imports %%.3f) | device context %%.3f s | OWN stage %%.3f s = %%s | the rest (plan, genomes waited for, tables of work items) %%.3f s || waited: %%s"
      %% (rank, ws, __import__('midas_amd.utility', fromlist=['x']).cpu_budget(), t_pipe - t_start, t_import, t_ctx, own,
          " + ".join("%%s %%.3f" %% kv for kv in OWN.items()), rest, ", ".join("%%s %%.3f s" %% kv for kv in WAIT.items())), flush=True)
print("PRED %%d %%.4f" %% (rank, (t_pipe - t_start) + t_ctx + own), flush=True)
'''


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    work = sys.argv[3] if len(sys.argv) > 3 else '/tmp/midas_predict'
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    if cfg == 'c4':
        contigs, reads, _ = synth.c4_share(0, 1)
    else:
        contigs, reads = synth.make_dataset(**synth.CONFIGS[cfg])
    out, db = os.path.join(work, 'sample'), os.path.join(work, 'db')
    synth.write_sample(out, db, contigs, reads)
    print("sample %s: %d species, %d sites, %d reads, BAM %.2f GB" % (cfg, contigs.n_species, contigs.n_sites, reads.n_reads,
                                                                   os.path.getsize(os.path.join(out, 'snps/temp/genomes.bam')) / 1e9), flush=True)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import vram_prelude
    vram_prelude.run()
    script = os.path.join(work, 'worker.py')
    open(script, 'w').write(WORKER % {"root": os.path.abspath(ROOT)})
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    for rep in range(2):
        shutil.rmtree(os.path.join(out, 'snps', 'output'), ignore_errors=True)
        os.makedirs(os.path.join(out, 'snps', 'output'))
        t = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, script, out, db, os.path.join(work, 'turn.lock')], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  env=dict(base, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                                           MASTER_PORT=str(29700 + rep))) for k in range(n)]
        res = [p.communicate(timeout=1500) + (p.returncode,) for p in procs]
        print("---- run %d: %d ranks taking turns on one GPU: %.2f s wall here (meaningless: the turns are serial)" % (rep + 1, n, time.perf_counter() - t), flush=True)
        pred = []
        for o, e, rc in res:
            if rc != 0:
                print("  FAILED rc %d: %s" % (rc, e[-1500:]))
            for line in o.splitlines():
                if line.startswith("RANK"):
                    print("  " + line)
                if line.startswith("PRED"):
                    pred.append(float(line.split()[2]))
        if pred:
            print("  PREDICTED wall of the job on %d GPUs: %.2f s = the slowest rank's start + context + own stage (mean %.2f s, fastest %.2f s)"
                  % (n, max(pred), sum(pred) / len(pred), min(pred)), flush=True)
        sz = sum(os.path.getsize(os.path.join(out, 'snps/output', f)) for f in os.listdir(os.path.join(out, 'snps/output')))
        print("  tables: %d, %.2f GB" % (len(os.listdir(os.path.join(out, 'snps/output'))), sz / 1e9), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
