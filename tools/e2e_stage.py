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

def stage(rep_check=True, device_decode=False):
    """device_decode: --device_inflate -- the BGZF blocks are inflated on the device and SEQ / QUAL / CIGAR stay there."""
    args = dict(abi.DEFAULT_ARGS, outdir=out, db=db, build_db=False, threads=utility.cpu_budget(),
                log=open(os.devnull, 'w'))
    T = {}
    # what run_pipeline does: the genomes are read on a thread of their own while the BAM is decoded
    t = time.perf_counter(); species = msnps.initialize_species(args); cs = msnps.ContigsInBackground(species)
    ctx = abi.Context(0)
    bam_path = os.path.join(out, 'snps/temp/genomes.bam')
    decoded = abi.read_bam(bam_path, ctx, resident=True) if device_decode else abi.read_bam(bam_path)
    T_bam = time.perf_counter() - t
    how = "decoded on the device into the kernel's own layout, everything resident" if device_decode else "native, parallel inflate"
    cs = cs.wait(); T['read FASTA (background thread) || BAM decode (%s; alone: %.3f s)' % (how, T_bam)] = time.perf_counter() - t
    ids = sorted(species)
    order, span = msnps._whole(msnps._species_contig_order(ids, cs), cs)
    items = [it for sp in ids for it in order[sp]]
    t = time.perf_counter(); table, sub, keys = msnps._contig_table(ids, items, span, cs, *decoded); T['contig table + regroup'] = time.perf_counter() - t
    with ctx:
        thr = abi.Thresholds.from_args(args)
        t = time.perf_counter(); b = ctx.batch(table, sub); T['H2D raw reads + device pack (batch_create)'] = time.perf_counter() - t
        t = time.perf_counter(); b.run(thr); b.sync(); T['device pass (index + pileup kernels)'] = time.perf_counter() - t
        pos = {it: k for k, it in enumerate(keys)}
        off = table.site_offsets()
        t = time.perf_counter()
        _, _, stats = b.fetch(counts=False, allele=False)
        # what run/snps.py does: the device formats and deflates the rows, the host frames and writes the members, a few tables side by side
        msnps._write_jobs(args, [('%s/snps/output/%s.snps.gz' % (out, sp), order[sp], None) for sp in ids], table, pos, None, None, off, b)
        T['rows: format + deflate on the device, D2H of the streams, write (level %d, %d files at a time)' % (msnps.GZ_LEVEL, msnps.WRITERS)] = time.perf_counter() - t
        # the same rows by the host's formatter (round 2's path): counts + alleles through the pinned ring, 16 threads format + deflate
        ctx.set_row_coder(abi.ROWS_HOST)
        t = time.perf_counter()
        msnps._write_jobs(args, [('%s/snps/output/%s.host.snps.gz' % (out, sp), order[sp], None) for sp in ids], table, pos, None, None, off, b)
        T_host = time.perf_counter() - t
        ctx.set_row_coder(abi.ROWS_DEVICE)
        sz_dev = sz_host = 0
        for sp in ids:
            a, h = ('%s/snps/output/%s%s.snps.gz' % (out, sp, x) for x in ('', '.host'))
            if rep_check:
                assert gzip.open(a, 'rb').read() == gzip.open(h, 'rb').read()
            sz_dev += os.path.getsize(a); sz_host += os.path.getsize(h)
            os.remove(h)
        b.close()
        T2 = T3 = 0.0
        if not device_decode:
            t = time.perf_counter(); ctx.pileup(thr, table, sub, pinned_slot=0); T2 = time.perf_counter() - t
            t = time.perf_counter(); ctx.pileup(thr, table, sub, pinned_slot=0); T3 = time.perf_counter() - t
    tot = sum(T.values())
    for k, v in T.items():
        print("  %-52s %8.3f s  %5.1f %%" % (k, v, 100 * v / tot))
    print("  %-52s %8.3f s  -> %.3e sites/s end to end (%d sites, %d reads)" % ("TOTAL pileup stage", tot, contigs.n_sites / tot, contigs.n_sites, reads.n_reads))
    print("  (the same rows by the host's formatter, %d threads: %.3f s; %d vs %d bytes, the same text%s)" % (args['threads'], T_host, sz_dev, sz_host, "" if rep_check else " (checked on run 1)"))
    if not device_decode:
        print("  one-shot midas_snps_pileup into pinned results: first call %.1f ms (pins the buffers), second %.1f ms" % (T2 * 1e3, T3 * 1e3))
    sz = sum(os.path.getsize(os.path.join(out, 'snps/output', f)) for f in os.listdir(os.path.join(out, 'snps/output')))
    print("  output: %.0f MB gz" % (sz / 1e6))


for rep in range(int(os.environ.get('E2E_REPS', '1'))):
    print('---- run %d ----' % (rep + 1), flush=True)
    stage(rep == 0)
    if os.environ.get('E2E_DEVICE_DECODE'):
        print('---- run %d, --device_inflate ----' % (rep + 1), flush=True)
        stage(False, True)
