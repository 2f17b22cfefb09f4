"""Developer script (GPU box): the rows of one configs[k] batch written three times by the device's row coder and three times by the
host's formatter (Batch.write_part, one file), sizes and times, and the two files' inflated text compared.  MIDAS_SNPS_TRACE=1
prints the device call's phases.   usage: python tools/rows_probe.py [config]"""
import sys, os, time, gzip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from midas_amd import abi, synth
table, reads = synth.make_dataset(**synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
with abi.Context(0) as ctx:
    b = ctx.batch(table, reads); b.run(thr); b.sync()
    pick = list(range(table.n_contigs))
    for coder, name in ((abi.ROWS_DEVICE, "device"), (abi.ROWS_HOST, "host")):
        ctx.set_row_coder(coder)
        for rep in range(3):
            t = time.perf_counter()
            b.write_part("/tmp/rows_%s.gz" % name, pick, table.ids, header=True, gz_level=4, threads=16)
            dt = time.perf_counter() - t
            print("%s: %.3f s, %d bytes (%.2f B/row)" % (name, dt, os.path.getsize("/tmp/rows_%s.gz" % name), os.path.getsize("/tmp/rows_%s.gz" % name) / table.n_sites), flush=True)
    a = gzip.open("/tmp/rows_device.gz", "rb").read(); h = gzip.open("/tmp/rows_host.gz", "rb").read()
    print("text equal:", a == h, "bytes equal:", open("/tmp/rows_device.gz","rb").read() == open("/tmp/rows_host.gz","rb").read())
    b.close()
