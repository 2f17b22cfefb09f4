"""Developer script (GPU box): PCIe-inclusive rate of the one-shot entry point (host SoA in, host tables out)."""
import sys, time
sys.path.insert(0, '.')
from midas_amd import abi, synth
contigs, reads = synth.make_dataset(**synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c2'])
thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
with abi.Context(0) as ctx:
    ctx.pileup(thr, contigs, reads)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.pileup(thr, contigs, reads); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("one-shot midas_snps_pileup (H2D raw + device pack + run + D2H into pageable arrays): %.1f ms -> %.3e sites/s" % (t * 1e3, contigs.n_sites / t))
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); ctx.pileup(thr, contigs, reads, pinned_slot=0); ts.append(time.perf_counter() - t0)
    print("one-shot into the context's pinned result buffers: first %.1f ms (pins), then %.1f ms -> %.3e sites/s" % (ts[0] * 1e3, min(ts[1:]) * 1e3, contigs.n_sites / min(ts[1:])))
    t0 = time.perf_counter(); b = ctx.batch(contigs, reads); t1 = time.perf_counter()
    b.run(thr); b.sync(); t2 = time.perf_counter(); b.fetch(); t3 = time.perf_counter()
    print("  batch_create (H2D raw + device pack) %.1f ms | run %.2f ms | fetch (D2H) %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
