"""Developer script (GPU box): `scripts/run_midas.py snps OUT --pileup` itself on a configs[k] sample -- interpreter start, imports,
the stage as the product chooses to run it (--device_inflate auto), the summary -- timed from outside, and again with
--device_inflate off.   usage: python tools/cli_stage.py [config] [workdir]"""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from midas_amd import synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/midas_cli'
shutil.rmtree(work, ignore_errors=True)
contigs, reads = synth.make_dataset(**synth.CONFIGS[cfg])
out, db = os.path.join(work, 'sample'), os.path.join(work, 'db')
synth.write_sample(out, db, contigs, reads)
print("sample: %d sites, %d reads, BAM %.0f MB" % (contigs.n_sites, reads.n_reads, os.path.getsize(os.path.join(out, 'snps/temp/genomes.bam')) / 1e6), flush=True)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vram_prelude  # noqa: E402
vram_prelude.run()
for how in ('auto', 'off', 'auto', 'off'):
    shutil.rmtree(os.path.join(out, 'snps', 'output'), ignore_errors=True)
    t = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'run_midas.py'), 'snps', out, '--pileup', '-d', db, '-t', '16',
                        '--device_inflate', how], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-2000:]
    sz = sum(os.path.getsize(os.path.join(out, 'snps/output', f)) for f in os.listdir(os.path.join(out, 'snps/output')))
    print("run_midas.py snps --pileup --device_inflate %-4s : %.2f s wall (process start to exit), %d tables, %.0f MB" % (
        how, dt, len(os.listdir(os.path.join(out, 'snps/output'))), sz / 1e6), flush=True)
