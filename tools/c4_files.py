"""Developer script (GPU box): BASELINE.json configs[3] THROUGH THE FILES on one GPU -- the 100-species sample (400 Mb, 80 M aligned
reads) written as a sample directory (a ~9 GB genomes.bam), `scripts/run_midas.py snps OUT --pileup` run on it as a user would,
timed from outside, and every table + summary.txt checked against the oracle: the C oracle's counts (all cores, one task per
contig) formatted by the host's row writer give the expected text of every <species>.snps.gz; the CRC-32 of the decompressed
tables must agree, and summary.txt's rows with the oracle's per-species counters.
usage: python tools/c4_files.py [workdir] [--small]     (--small: 8 species, for a quick check of the script itself)"""
import gzip
import os
import shutil
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from midas_amd import abi, synth, utility  # noqa: E402
from oracle import c_oracle  # noqa: E402


def text_crc(path):
    crc, n = 0, 0
    with gzip.open(path, 'rb') as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            crc = zlib.crc32(b, crc)
            n += len(b)
    return crc, n


def main():
    work = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else '/tmp/midas_c4'
    kw = dict(n_species=8) if '--small' in sys.argv else {}
    shutil.rmtree(work, ignore_errors=True)
    t0 = time.time()
    contigs, reads, facts = synth.c4_share(0, 1, **kw)
    t1 = time.time()
    out, db = os.path.join(work, 'sample'), os.path.join(work, 'db')
    synth.write_sample(out, db, contigs, reads)
    bam = os.path.join(out, 'snps/temp/genomes.bam')
    print("sample: %d species, %d contigs, %d sites, %d reads; generated in %.0f s, written in %.0f s; BAM %.2f GB; host CPUs %d" % (
        contigs.n_species, contigs.n_contigs, contigs.n_sites, reads.n_reads, t1 - t0, time.time() - t1, os.path.getsize(bam) / 1e9,
        utility.cpu_budget()), flush=True)
    if os.environ.get('C4_PRELUDE', '1') != '0':
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import vram_prelude
        vram_prelude.run()
    runs = []
    for how in os.environ.get('C4_RUNS', 'auto,auto,auto,off').split(','):        # ('auto+KEY=VALUE': with that in the run's environment)
        shutil.rmtree(os.path.join(out, 'snps', 'output'), ignore_errors=True)
        time.sleep(float(os.environ.get('C4_PAUSE', '0')))       # (the driver wipes what the run before released: an allocation that lands on it waits -- tools/alloc_probe.py)
        env = dict(os.environ, MIDAS_SNPS_TRACE='1')
        for kv in how.split('+')[1:]:       # ('auto+MIDAS_SNPS_DECODE_STREAM=0': the run's environment)
            env[kv.split('=')[0]] = kv.split('=')[1]
        t = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'run_midas.py'), 'snps', out, '--pileup', '-d', db, '--device_inflate', how.split('+')[0]],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        dt = time.perf_counter() - t
        if r.returncode != 0:
            print("run_midas.py failed:", r.stderr[-3000:])
            sys.exit(1)
        sz = sum(os.path.getsize(os.path.join(out, 'snps/output', f)) for f in os.listdir(os.path.join(out, 'snps/output')))
        print("run_midas.py snps --pileup --device_inflate %-40s: %.2f s wall (process start to exit) -> %.3e sites/s end to end; %d tables, %.2f GB gz" % (
            how, dt, contigs.n_sites / dt, len(os.listdir(os.path.join(out, 'snps/output'))), sz / 1e9), flush=True)
        top = 0.0
        for line in r.stderr.splitlines():
            if line.startswith(('[device decode]', '[batch_create]', '[stage]', '[bam ', '[fasta]')):
                print("    " + line)
            if line.startswith('[stage] ') and not line.startswith('[stage]   '):        # (un-indented: the stage's own consecutive phases)
                top += float(line.split()[-2])
        print("    the stage's un-indented phases add up to %.3f s of the %.3f s wall (the rest: interpreter exit, the launcher)" % (top / 1e3, dt), flush=True)
        runs.append(dt)
        if how.startswith('auto'):
            keep = {f: text_crc(os.path.join(out, 'snps/output', f)) for f in sorted(os.listdir(os.path.join(out, 'snps/output')))} if len(runs) == 1 else keep
            if len(runs) > 1 and ('+' in how or os.environ.get('C4_VERIFY_ALL')):        # (a variant run: its text against the first run's, which is held to the oracle below)
                t = time.perf_counter()
                now = {f: text_crc(os.path.join(out, 'snps/output', f)) for f in sorted(os.listdir(os.path.join(out, 'snps/output')))}
                summ = open(os.path.join(out, 'snps', 'summary.txt')).read()
                print("    this run's %d tables: %s the first run's (CRC-32 + length of the decompressed rows, %.0f s); summary.txt %s" % (
                    len(now), "the same text as" if now == keep else "DIFFERENT FROM", time.perf_counter() - t,
                    "the same" if summ == first_summary else "DIFFERENT"), flush=True)
                if now != keep or summ != first_summary:
                    sys.exit(1)
            elif len(runs) == 1:
                first_summary = open(os.path.join(out, 'snps', 'summary.txt')).read()
    # ---- the oracle: counts by the C restatement (all cores), text by the host's row writer, CRC-32 of the text ----
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    t = time.perf_counter()
    c_oracle.build()
    st, counts, stats = c_oracle.pileup_parallel(thr, contigs, reads, utility.cpu_budget(), 'contig')
    assert st == 0
    print("oracle (C, %d threads): %.1f s" % (utility.cpu_budget(), time.perf_counter() - t), flush=True)
    off = contigs.site_offsets()
    allele = np.frombuffer(bytes(contigs.ref).upper(), np.uint8)
    bad = 0
    exp_dir = os.path.join(work, 'expected')
    os.makedirs(exp_dir, exist_ok=True)
    order = {}
    for k, cid in enumerate(contigs.ids):
        order.setdefault(int(contigs.species[k]), []).append(k)
    t = time.perf_counter()
    for si, sp in enumerate(contigs.species_ids):
        ks = sorted(order.get(si, []), key=lambda k: contigs.ids[k])        # sorted(contigs.keys()), midas/run/snps.py:187
        path = os.path.join(exp_dir, sp + '.snps.gz')
        abi.write_table(path, [contigs.ids[k] for k in ks], [allele[off[k]:off[k + 1]] for k in ks], [counts[off[k]:off[k + 1]] for k in ks])
        want = text_crc(path)
        got_last = text_crc(os.path.join(out, 'snps/output', sp + '.snps.gz'))       # (the last run: the host decode)
        os.remove(path)
        if want != keep[sp + '.snps.gz'] or want != got_last:
            bad += 1
            print("MISMATCH", sp, want, keep[sp + '.snps.gz'], got_last)
    print("tables: %d of %d agree with the oracle's text (CRC-32 + length of the decompressed rows; device decode and host decode); %.0f s" % (
        contigs.n_species - bad, contigs.n_species, time.perf_counter() - t), flush=True)
    rows = {l.split('\t')[0]: l.split('\t')[1:] for l in open(os.path.join(out, 'snps', 'summary.txt')).read().splitlines()[1:]}
    sbad = 0
    glen = np.bincount(contigs.species, weights=contigs.length, minlength=contigs.n_species).astype(np.int64)
    for si, sp in enumerate(contigs.species_ids):
        al, mp, cov, dep = (int(x) for x in stats[si])
        want = [str(int(glen[si])), str(cov), str(cov / float(glen[si])) if glen[si] else '0', str(dep / float(cov)) if cov else '0', str(al), str(mp)]
        if rows.get(sp) != want:
            sbad += 1
            print("SUMMARY MISMATCH", sp, rows.get(sp), want)
    print("summary.txt: %d of %d rows agree with the oracle's counters" % (contigs.n_species - sbad, contigs.n_species))
    print("RESULT: %s" % ("ok" if bad == 0 and sbad == 0 else "FAILED"))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
