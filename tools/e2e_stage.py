"""Developer script (GPU box): end-to-end timing of the pileup STAGE (SURVEY 8d second timing):
snps/temp/genomes.bam on disk -> <species>.snps.gz + summary.txt, phase by phase, through the same code
`run_midas.py snps --pileup` runs.  usage: python tools/e2e_stage.py [config] [workdir]   (E2E_REPS=n repeats the stage on the same files)"""
import gzip
import os
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from midas_amd import abi, synth, utility  # noqa: E402
from midas_amd.run import snps as msnps  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/midas_e2e'
shutil.rmtree(work, ignore_errors=True)
t0 = time.time()
contigs, reads = synth.make_dataset(**synth.CONFIGS[cfg])
out, db = os.path.join(work, 'sample'), os.path.join(work, 'db')
synth.write_sample(out, db, contigs, reads)
print("setup (generate + write FASTA/BAM, not part of the stage): %.1f s; BAM %.0f MB" % (
    time.time() - t0, os.path.getsize(os.path.join(out, 'snps/temp/genomes.bam')) / 1e6), flush=True)

def stage():
    args = dict(abi.DEFAULT_ARGS, outdir=out, db=db, build_db=False, threads=utility.cpu_budget(),
                log=open(os.devnull, 'w'))
    T = {}
    t = time.perf_counter(); species = msnps.initialize_species(args); cs = msnps.initialize_contigs(species); T['read FASTA'] = time.perf_counter() - t
    t = time.perf_counter(); decoded = abi.read_bam(os.path.join(out, 'snps/temp/genomes.bam')); T['BAM decode (native, parallel inflate)'] = time.perf_counter() - t
    ids = sorted(species)
    order, span = msnps._whole(msnps._species_contig_order(ids, cs), cs)
    items = [it for sp in ids for it in order[sp]]
    t = time.perf_counter(); table, sub, keys = msnps._contig_table(ids, items, span, cs, *decoded); T['contig table + regroup'] = time.perf_counter() - t
    with abi.Context(0) as ctx:
        thr = abi.Thresholds.from_args(args)
        t = time.perf_counter(); b = ctx.batch(table, sub); T['H2D raw reads + device pack (batch_create)'] = time.perf_counter() - t
        t = time.perf_counter(); b.run(thr); b.sync(); T['device pass (index + pileup kernels)'] = time.perf_counter() - t
        pos = {it: k for k, it in enumerate(keys)}
        off = table.site_offsets()
        t = time.perf_counter()
        _, _, stats = b.fetch(counts=False, allele=False)
        for sp in ids:      # what run/snps.py does: the rows leave the device slab by slab while the formatter works
            msnps._write_rows(args, '%s/snps/output/%s.snps.gz' % (out, sp), table, pos, order[sp], None, None, off, None, b)
        T['rows: D2H through the pinned ring + format + gzip (%d threads, level %d)' % (args['threads'], msnps.GZ_LEVEL)] = time.perf_counter() - t
        t = time.perf_counter(); counts, allele, _ = b.fetch(); T_fetch = time.perf_counter() - t
        t = time.perf_counter()
        for sp in ids:
            msnps._write_rows(args, '%s/snps/output/%s.host.snps.gz' % (out, sp), table, pos, order[sp], counts, allele, off, None)
        T_host = time.perf_counter() - t
        for sp in ids:
            a, h = ('%s/snps/output/%s%s.snps.gz' % (out, sp, x) for x in ('', '.host'))
            assert open(a, 'rb').read() == open(h, 'rb').read()
            os.remove(h)
        b.close()
        t = time.perf_counter(); ctx.pileup(thr, table, sub, pinned_slot=0); T2 = time.perf_counter() - t
        t = time.perf_counter(); ctx.pileup(thr, table, sub, pinned_slot=0); T3 = time.perf_counter() - t
    tot = sum(T.values())
    for k, v in T.items():
        print("  %-52s %8.3f s  %5.1f %%" % (k, v, 100 * v / tot))
    print("  %-52s %8.3f s  -> %.3e sites/s end to end (%d sites, %d reads)" % ("TOTAL pileup stage", tot, contigs.n_sites / tot, contigs.n_sites, reads.n_reads))
    print("  (the same rows the round-1 way: fetch into pageable arrays %.3f s, then format + gzip from them %.3f s; identical files)" % (T_fetch, T_host))
    print("  one-shot midas_snps_pileup into pinned results: first call %.1f ms (pins the buffers), second %.1f ms" % (T2 * 1e3, T3 * 1e3))
    sz = sum(os.path.getsize(os.path.join(out, 'snps/output', f)) for f in os.listdir(os.path.join(out, 'snps/output')))
    print("  output: %.0f MB gz" % (sz / 1e6))


for rep in range(int(os.environ.get('E2E_REPS', '1'))):
    print('---- run %d ----' % (rep + 1), flush=True)
    stage()
