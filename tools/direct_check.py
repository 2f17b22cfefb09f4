"""Developer script (GPU box): times the two device paths of a batch back to back on one dataset.
Usage: python tools/direct_check.py [config] [steps]      (config from midas_amd.synth.CONFIGS, default c3)
"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midas_amd import abi, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = abi.Context(0)
    print(ctx.device_info(), flush=True)
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    t0 = time.time()
    cache = os.path.join(os.environ.get('DIRECT_CHECK_CACHE', '/tmp'), 'direct_check_%s.npz' % name)
    if os.path.exists(cache):        # (several variants in one GPU call: generate once)
        z = np.load(cache, allow_pickle=True)
        reads = abi.ReadsSoA(**{k[2:]: z[k] for k in z.files if k.startswith('r_')})
        contigs = abi.ContigTable(length=z['c_length'], species=z['c_species'], read_begin=z['c_read_begin'], ref=z['c_ref'],
                                  n_species=int(z['c_n_species']))
    else:
        contigs, reads = synth.make_dataset(**synth.CONFIGS[name])
        try:
            np.savez(cache, c_length=contigs.length, c_species=contigs.species, c_read_begin=contigs.read_begin, c_ref=contigs.ref,
                     c_n_species=contigs.n_species, **{'r_' + k: v for k, v in reads.as_dict().items()})
        except Exception as e:
            print("cache not written:", e)
    print("dataset %s: %.1f s" % (name, time.time() - t0), flush=True)
    t0 = time.time()
    b = ctx.batch(contigs, reads)
    print("batch_create: %.3f s" % (time.time() - t0), flush=True)
    info = b.info()
    print("reads %d sites %d tiles %d | auto path %s | general reads %d reach %d | stream reads %d (%.3f x) max/tile %d | lanes %d x %d bases"
          % (info.n_reads, info.n_sites, info.n_tiles, abi.PATH_NAMES[info.path_auto], info.direct_general_reads,
             info.direct_reach, info.direct_stream_reads, info.direct_stream_reads / max(1, info.n_reads),
             info.direct_max_tile_reads, info.lanes_per_read, info.lane_bases), flush=True)
    res = {}
    only = os.environ.get('DIRECT_CHECK_PATHS')
    for path in ((abi.PATH_DIRECT, abi.PATH_PACKED, abi.PATH_DIRECT) if not only else tuple(int(x) for x in only.split(','))):
        b.select_path(path)
        b.enable_timing(steps)
        for _ in range(3):
            b.run(thr)
        b.sync()
        b.enable_timing(steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            b.run(thr)
        b.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
        tm = [b.timing(i) for i in range(steps)]
        idx = float(np.mean([t['index_ms'] for t in tm])); pil = float(np.mean([t['pileup_ms'] for t in tm]))
        alg = b.info().algorithmic_bytes
        print("%-7s index %.4f ms  pileup %.4f ms  run %.4f ms  wall/step %.4f ms | %.0f GB/s alg over the run = %.3f of 8 TB/s | %.3e sites/s"
              % (abi.PATH_NAMES[path], idx, pil, idx + pil, wall, alg / (idx + pil) / 1e6, alg / (idx + pil) / 1e6 / 8000.0,
                 info.n_sites / wall * 1e3), flush=True)
        res[path] = b.fetch()
    if abi.PATH_DIRECT in res and abi.PATH_PACKED in res:
        same = all(np.array_equal(x, y) for x, y in zip(res[abi.PATH_DIRECT], res[abi.PATH_PACKED]))
        print("direct == packed:", same, flush=True)
    b.close()
    ctx.close()


if __name__ == "__main__":
    main()
