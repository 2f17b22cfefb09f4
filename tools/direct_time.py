"""Developer script (GPU box): the direct path timed on a cached dataset (tools/direct_check.py makes the cache).  Usage: direct_time.py [config]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midas_amd import abi, synth  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
ctx = abi.Context(0)
thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
z = np.load(os.path.join(os.environ.get('DIRECT_CHECK_CACHE', '/tmp'), 'direct_check_%s.npz' % name), allow_pickle=True)
reads = abi.ReadsSoA(**{k[2:]: z[k] for k in z.files if k.startswith('r_')})
contigs = abi.ContigTable(length=z['c_length'], species=z['c_species'], read_begin=z['c_read_begin'], ref=z['c_ref'], n_species=int(z['c_n_species']))
b = ctx.batch(contigs, reads)
b.select_path(abi.PATH_DIRECT)
for rep in range(2):
    b.enable_timing(20)
    for _ in range(20):
        b.run(thr)
    b.sync()
tm = [b.timing(i) for i in range(20)]
print("direct: index %.4f ms pileup %.4f ms" % (np.mean([t['index_ms'] for t in tm]), np.mean([t['pileup_ms'] for t in tm])))
