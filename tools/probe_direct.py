"""Developer script (GPU box): where the waves of the direct pileup kernel spend their cycles.
Needs the probe variant: tools/build_variant.sh probe -DMIDAS_SNPS_DEBUG_BITS=256, then
MIDAS_SNPS_LIBRARY=midas_amd/lib/libmidas_snps_hip_probe.so python tools/probe_direct.py [config]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midas_amd import abi, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    ctx = abi.Context(0)
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    cache = os.path.join(os.environ.get('DIRECT_CHECK_CACHE', '/tmp'), 'direct_check_%s.npz' % name)
    if os.path.exists(cache):
        z = np.load(cache, allow_pickle=True)
        reads = abi.ReadsSoA(**{k[2:]: z[k] for k in z.files if k.startswith('r_')})
        contigs = abi.ContigTable(length=z['c_length'], species=z['c_species'], read_begin=z['c_read_begin'], ref=z['c_ref'],
                                  n_species=int(z['c_n_species']))
    else:
        contigs, reads = synth.make_dataset(**synth.CONFIGS[name])
    b = ctx.batch(contigs, reads)
    b.select_path(abi.PATH_DIRECT)
    b.enable_timing(8)
    for _ in range(8):
        b.run(thr)
    b.sync()
    tm = [b.timing(i) for i in range(8)]
    print("probe build: index %.4f ms pileup %.4f ms" % (np.mean([t['index_ms'] for t in tm]), np.mean([t['pileup_ms'] for t in tm])))
    lib = abi.load_library()
    n = 256 * 4 * 4 * 8          # CUs x workgroups per CU x waves per workgroup x words
    out = (C.c_ulonglong * n)()
    lib.midas_snps_debug_probe.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int64]
    st = lib.midas_snps_debug_probe(b._h, out, n)
    assert st == 0, st
    a = np.frombuffer(out, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    a = a[a[:, 6] > 0]
    tot = a[:, 6].sum()
    names = ["wait columns + issue", "wait bases", "work", "barrier 1 (DB: wait for a clean buffer)", "write-out + barrier 2 (DB: drain, all of it)", "iterations", "all", "DB: of the drain, waiting for the tile's last wave"]
    for k in (0, 2, 3, 4, 7):
        print("%-52s %6.1f %% of wave cycles   (%.0f cycles / iteration)" % (names[k], 100 * a[:, k].sum() / tot, a[:, k].sum() / a[:, 5].sum()))
    print("iterations per wave: mean %.1f  min %d  max %d ; cycles per wave: mean %.0f  min %.0f  max %.0f" % (
        a[:, 5].mean(), a[:, 5].min(), a[:, 5].max(), a[:, 6].mean(), a[:, 6].min(), a[:, 6].max()))
    # where the workgroups ran: HW_ID (cu_id bits 8-11, sh_id 12, se_id 13-15) and XCC_ID (bits 0-3) of every wave
    raw = np.frombuffer(out, dtype=np.uint64).reshape(-1, 8)
    raw = raw[raw[:, 6] > 0]
    hw = (raw[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((raw[:, 1] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    cu = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 15)
    cyc = raw[:, 6].astype(np.float64)
    n_w = 4
    wg_cyc = cyc.reshape(-1, n_w).max(axis=1)
    wg_cu = cu.reshape(-1, n_w)[:, 0]
    wg_xcc = xcc.reshape(-1, n_w)[:, 0]
    print("workgroups %d on %d distinct CUs; workgroups per CU: %s" % (wg_cyc.size, np.unique(wg_cu).size, np.bincount(np.unique(wg_cu, return_counts=True)[1]).tolist()))
    for x in range(8):
        m = wg_xcc == x
        if m.any():
            print("XCC %d: %3d workgroups, cycles mean %.0f min %.0f max %.0f" % (x, m.sum(), wg_cyc[m].mean(), wg_cyc[m].min(), wg_cyc[m].max()))
    within = (cyc.reshape(-1, n_w).max(axis=1) - cyc.reshape(-1, n_w).min(axis=1))
    print("spread inside a workgroup (max - min wave cycles): mean %.0f max %.0f" % (within.mean(), within.max()))
    order = np.argsort(wg_cyc)
    print("slowest workgroups (block, cu, cycles):", [(int(i), int(wg_cu[i]), int(wg_cyc[i])) for i in order[-6:]])
    print("fastest workgroups (block, cu, cycles):", [(int(i), int(wg_cu[i]), int(wg_cyc[i])) for i in order[:6]])
    b.close()
    ctx.close()


if __name__ == "__main__":
    main()
