import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from midas_amd import abi, bam, synth
THR = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=3, contig_len=40000, n_reads=60000, seed=171, var_len=True)
refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
path = "/tmp/dbg_s.bam"
bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
ctx = abi.Context(0)
for mode in sys.argv[1:]:
    os.environ["MIDAS_SNPS_DECODE_STREAM"] = "0" if mode == "arena" else "1"
    os.environ["MIDAS_SNPS_DECODE_GROUP_BLOCKS"] = "16"
    print("mode", mode, flush=True)
    _, _, rid, res = abi.read_bam(path, ctx, resident=True)
    b = ctx.batch(contigs, res)
    b.run(THR); c = b.fetch(); print(" direct ok", int(c[0].sum()), flush=True)
    down = ctx.fetch_payload(res) if "fetch" in mode else None
    print(" fetch ok", flush=True)
    b.select_path(abi.PATH_PACKED); print(" packed selected", flush=True)
    b.run(THR); c2 = b.fetch(); print(" packed ok", np.array_equal(c[0], c2[0]), flush=True)
    b.select_path(abi.PATH_LONG); b.run(THR); c3 = b.fetch(); print(" long ok", np.array_equal(c[0], c3[0]), flush=True)
    b.close()
