#!/usr/bin/env python
"""bench.py -- genomic sites/sec of the MI355X SNP pileup (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path FROM THE BAM-NATIVE ARRAYS resident in HBM (pos, mapq, NM, l_seq, CSR offsets,
4-bit SEQ, QUAL, CIGAR -- what the BAM decoder hands over) to the per-site counts: the ranges pass (positions only, 4 bytes
per read: per tile the run of reads that can touch it -- the index_bam analogue), then the pileup kernel, which visits every
read ONCE: it fetches the read's columns, SEQ, QUAL and CIGAR where they are, decides the CIGAR's shape in registers, filters
(keep_read), reduces the mean quality, tallies A/C/G/T per site and emits counts, ref alleles and the per-species counters.
Nothing is sorted, packed, described or cached between steps.  With N > 1 a rank's K steps are its share of the job and the
job's one exchange -- the all-gather of every rank's per-species summary rows over RCCL -- follows them, inside the timed
region.

Workload (default): BASELINE.json configs[2], the largest single-GPU configuration -- 20 species, 80 Mb, 10 666 667 aligned
synthetic 150 bp reads (20x) per GPU (`--config c2` = configs[1]; `--config c4_rank` = one rank's share of configs[3]).
Multi-GPU: the line's own figures are weak scaling -- every rank owns its own configs[2]-sized set of species, so N = 1
agrees with the single-GPU line -- and, with no extra flag, the same line carries `configs3_strong`: BASELINE.json
configs[3] itself -- ONE sample of 100 species, 400 Mb, 80 M aligned reads -- dealt to the N ranks contig by contig with
the product's partitioner (midas_amd.dist.shard_items, the weights of midas_amd/run/snps.py): strong scaling, no data-path
collective, one RCCL all-gather of the summary rows, 400 M sites / the slowest rank (`--configs3` adds the block at N = 1,
`--config c4` makes configs[3] the line's own workload).

Secondary figures in the same JSON line: the step over a resident PACKED batch (`value_resident_packed`: the tile-ordered
records + one byte per base of pack_reads.hip, built once, outside the region), the device copy rate of this box, the CPU
baselines, and -- when rocprofv3 is on PATH -- the HBM traffic of the step's kernels measured on this box.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    "c2": "configs[1]: 1 species rep-genome (60 contigs x 250 kb = 15 Mb), 1M synthetic 150 bp reads at 10x per GPU",
    "c3": "configs[2]: 20 species (320 contigs x 250 kb = 80 Mb), 10 666 667 aligned synthetic 150 bp reads at 20x per GPU",
    "c4_rank": "one rank's share of configs[3]: 13 species (52 Mb), 10.4M aligned synthetic 150 bp reads at 30x",
    "c4": "configs[3]: 100 species (1600 contigs x 250 kb = 400 Mb), 80M aligned synthetic 150 bp reads (30x on average, "
          "log-normal abundances), contig-sharded over the ranks by midas_amd.dist.shard_items",
}
STEP_KERNELS = ["direct_ranges_kernel", "pileup_direct_kernel"]
CALIB_BYTES = 1 << 30
# how a step kernel's read bytes split over load widths (bytes per lane): weights of the calibrated FETCH_SIZE factors.
# ranges: pos[i], pos[i-1] dwords.  pileup, per 150 bp read: 241 B by 16-byte loads (QUAL 150, SEQ 75, first CIGAR ops 16),
# 13 B by dword / byte loads (pos, l_seq, NM, mapq), 16 B by 8-byte loads (SEQ / QUAL offsets), 12 B by a 12-byte load.
LOAD_MIX = {"direct_ranges_kernel": {4: 1.0}, "pileup_direct_kernel": {16: 241.0 / 282.0, 8: 28.0 / 282.0, 4: 13.0 / 282.0}}


def load_pmc_traffic(key, workload):
    """(HBM bytes per launch, where the figure comes from): the committed rocprofv3 --pmc passes of this same command."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        e = d.get(workload, {}).get(key)
        if not e:
            return None, None
        return float(e["hbm_bytes_per_launch"]), "committed profile: %s" % e.get("source", "profiles/pmc_traffic.json")
    except Exception:
        return None, None


def cpu_baseline(thr, contigs, reads, min_seconds):
    """The C oracle ("port" of the reference semantics) timed on this host: 1 core, then all cores with one task per
    contig, then the reference's own grain (one worker per species, midas/run/snps.py:225-228).
    Returns (1-core dict, all-cores dict, species-grain dict, outputs of the 1-core pass)."""
    from midas_amd import utility
    from oracle import c_oracle
    c_oracle.build()
    t0 = time.perf_counter()
    passes = 0
    out = None
    while True:
        out = c_oracle.pileup(thr, contigs, reads)
        passes += 1
        el = time.perf_counter() - t0
        if el >= min_seconds or passes >= 64:
            break
    sites = contigs.n_sites * passes
    one = {"value": sites / el, "unit": "sites/s", "cores": 1, "kind": "port",
           "sample": "%d pass(es) of the full workload (%d sites, %d reads) through oracle/pileup_oracle.c, %.1f s"
                     % (passes, contigs.n_sites, reads.n_reads, el),
           "host_hardware_threads": os.cpu_count(), "host_cpu_budget": utility.cpu_budget()}
    ncpu = utility.cpu_budget()
    more = []
    for grain, workers in (("contig", min(ncpu, contigs.n_contigs)), ("species", min(ncpu, contigs.n_species))):
        t0 = time.perf_counter()
        passes = 0
        ok = True
        while True:
            st, c, s = c_oracle.pileup_parallel(thr, contigs, reads, workers, grain)
            ok = ok and st == 0 and np.array_equal(c, out[2]) and np.array_equal(s, out[4])
            passes += 1
            el = time.perf_counter() - t0
            if el >= min_seconds / 2 or passes >= 64:
                break
        more.append({"value": contigs.n_sites * passes / el, "unit": "sites/s", "cores": int(workers), "kind": "port",
                     "grain": "one task per %s" % grain, "equals_1core_output": bool(ok),
                     "host_hardware_threads": os.cpu_count(), "host_cpu_budget": ncpu,
                     "sample": "%d pass(es) of the full workload over %d threads, %.1f s" % (passes, workers, el)})
    return one, more[0], more[1], out


def python_shaped_estimate(contigs, reads, args, max_sites=250000):
    """the pysam-shaped Python oracle (per-read callback, per-site Python emit + gzip-9
    text) on the first contig only -- an ESTIMATE of what the reference's own loop costs, never 'the reference'."""
    import gzip
    import io
    from oracle import pileup_oracle as po
    n = int(contigs.read_begin[1])
    length = min(int(contigs.length[0]), max_sites)
    alns = po.alns_from_soa(reads.as_dict(), 0, n)
    ref = bytes(contigs.ref[:length]).decode().upper()
    oc = {"c": po.OContig(id="c", seq=ref, species_id="s")}
    t0 = time.perf_counter()
    text, _ = po.species_pileup(args, "s", oc, {"c": alns})
    buf = io.BytesIO()
    with io.TextIOWrapper(gzip.GzipFile(fileobj=buf, mode="w")) as f:
        f.write(text)
    el = time.perf_counter() - t0
    return {"value": length / el, "unit": "sites/s", "cores": 1, "kind": "python-shaped estimate",
            "sample": "first contig (%d sites, %d reads), Python per-read/per-site loops + gzip-9" % (length, n)}


# ---- HBM traffic of the step's kernels on THIS box: rocprofv3 --pmc around a child that runs a few steps ----------------------
def save_dataset(path, contigs, reads):
    np.savez(path, c_length=contigs.length, c_species=contigs.species, c_read_begin=contigs.read_begin, c_ref=contigs.ref,
             c_n_species=contigs.n_species, **{"r_" + k: v for k, v in reads.as_dict().items()})


def load_dataset(path):
    from midas_amd import abi
    z = np.load(path)
    reads = abi.ReadsSoA(**{k[2:]: z[k] for k in z.files if k.startswith("r_")})
    contigs = abi.ContigTable(length=z["c_length"], species=z["c_species"], read_begin=z["c_read_begin"], ref=z["c_ref"],
                              n_species=int(z["c_n_species"]))
    return contigs, reads


def pmc_child(path, steps):
    """(run under rocprofv3 by measure_traffic) a few steps over the saved dataset, nothing else."""
    from midas_amd import abi
    contigs, reads = load_dataset(path)
    ctx = abi.Context(0)
    b = ctx.batch(contigs, reads)
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    for _ in range(steps):
        b.run(thr)
    b.sync()
    b.close()
    ctx.calibration_pass(CALIB_BYTES)      # known byte counts at 4 / 8 / 16 bytes per lane: the counters' factors
    ctx.close()


def measure_traffic(contigs, reads, steps=6):
    """FETCH_SIZE / WRITE_SIZE of the step's kernels, separate rocprofv3 --pmc passes with --kernel-trace only, corrected as
    MI355X_MICROARCH.md prescribes (KiB * 1024; FETCH_SIZE doubled on gfx950).  Returns a dict or None."""
    import sqlite3
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    tmp = tempfile.mkdtemp(prefix="midas_pmc_", dir="/tmp")
    try:
        data = os.path.join(tmp, "dataset.npz")
        save_dataset(data, contigs, reads)
        per = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", data, "--steps", str(steps)]
            res = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("_results.db")]
            if res.returncode != 0 or not dbs:
                return {"error": "rocprofv3 --pmc %s failed (rc %d)" % (counter, res.returncode)}
            cur = sqlite3.connect(dbs[0]).cursor()
            q = "select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name"
            rows_ = list(cur.execute(q, (counter,)))
            for name, v, n in rows_:
                for k in STEP_KERNELS + ["midas_calib_read4_kernel", "midas_calib_read8_kernel", "midas_calib_read16_kernel",
                                         "midas_calib_write16_kernel"]:
                    if k in name:
                        e = per.setdefault(k, {})
                        e[counter] = float(v)
                        e[counter + "_launches"] = int(n)
        # the counters' factors for this box and these access widths: known bytes / (counter KiB * 1024)
        calib = {}
        for w in (4, 8, 16):
            v = per.pop("midas_calib_read%d_kernel" % w, {}).get("FETCH_SIZE")
            calib["fetch_factor_%dB_per_lane" % w] = CALIB_BYTES / (v * 1024.0) if v else None
        v = per.pop("midas_calib_write16_kernel", {}).get("WRITE_SIZE")
        calib["write_factor_16B_per_lane"] = CALIB_BYTES / (v * 1024.0) if v else None
        total_r = total_w = 0.0
        for k, e in per.items():
            mix = LOAD_MIX.get(k, {16: 1.0})
            fr = sum(wt * (calib.get("fetch_factor_%dB_per_lane" % w) or 2.0) for w, wt in mix.items())
            fw = calib.get("write_factor_16B_per_lane") or 1.0
            rd = e.get("FETCH_SIZE", 0.0) * 1024.0 * fr
            wr = e.get("WRITE_SIZE", 0.0) * 1024.0 * fw
            e["fetch_factor"] = fr
            e["write_factor"] = fw
            e["hbm_read_bytes_corrected"] = rd
            e["hbm_write_bytes"] = wr
            total_r += rd
            total_w += wr
        return {"per_kernel": per, "hbm_bytes_per_step": total_r + total_w, "hbm_read_bytes": total_r, "hbm_write_bytes": total_w,
                "calibration": calib,
                "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes on this box; bytes = KiB * 1024 * "
                          "factor, the factor CALIBRATED in the same passes by kernels that move a known 1 GiB with 4-, 8- and "
                          "16-byte-per-lane loads / 16-byte stores (midas_snps_calibration_pass) and applied per kernel by the "
                          "byte mix of its load widths (bench.py LOAD_MIX); averages include the one ranges launch of batch_create"}
    except Exception as e:  # a courtesy measurement: never fail the bench on it
        return {"error": str(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _on_one_gpu():
    """MIDAS_BENCH_ONE_GPU=1 (tests/test_gpu_bench.py): the N > 1 code of this script run as N processes that SHARE device 0 and
    talk over gloo -- RCCL refuses two ranks on one device.  Everything but the transport is the path the driver launches."""
    return os.environ.get("MIDAS_BENCH_ONE_GPU") == "1"


def _all_gather(out, inp):
    import torch.distributed as dist
    if _on_one_gpu():          # gloo: through host copies
        i = inp.cpu().contiguous()
        o = out.cpu().reshape((out.numel() // i.numel() * i.shape[0],) + tuple(i.shape[1:]))     # (gloo wants the concatenated form)
        dist.all_gather_into_tensor(o, i)
        out.copy_(o.reshape(out.shape))
    else:
        dist.all_gather_into_tensor(out, inp)


def _all_reduce(t, op):
    import torch.distributed as dist
    if _on_one_gpu():
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


def configs3_strong(ctx, thr, rank, world, collective, steps):
    """BASELINE.json configs[3] as the product deals it: ONE 100-species sample (400 Mb, 80 M aligned reads), its contigs dealt
    to the ranks by midas_amd.dist.shard_items, every rank piling up its share, one all-gather of the summary rows per job.
    Strong scaling: value = 400 M sites / the slowest rank.  Returns the block rank 0 prints (None elsewhere)."""
    import torch
    import torch.distributed as dist
    from midas_amd import abi, synth
    contigs, reads, share = synth.c4_share(rank, world)
    batch = ctx.batch(contigs, reads)
    info = batch.info()
    n_sp = contigs.n_species
    rows = torch.zeros((n_sp, abi.NUM_STATS), dtype=torch.int64, device="cuda")
    gathered = torch.zeros((world, n_sp, abi.NUM_STATS), dtype=torch.int64, device="cuda")
    for _ in range(3):
        batch.run(thr)
    batch.sync()
    if collective:
        _all_gather(gathered, rows)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    batch.enable_timing(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run(thr)
    batch.stats_to_device(rows.data_ptr())
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                      # this rank's share of the job, steps times
    if collective:
        _all_gather(gathered, rows)     # the job's one exchange: every rank's per-species rows
        torch.cuda.synchronize()
    else:
        gathered[0].copy_(rows)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    batch.sync()
    tm = [batch.timing(i) for i in range(steps)]
    kern = float(np.mean([t["index_ms"] + t["pileup_ms"] for t in tm]))
    vals = torch.tensor([own, elapsed, kern, float(info.n_sites), float(info.n_reads), float(info.algorithmic_bytes)], dtype=torch.float64, device="cuda")
    allv = torch.zeros((world, vals.numel()), dtype=torch.float64, device="cuda")
    if collective:
        _all_gather(allv, vals)
    else:
        allv[0].copy_(vals)
    allv = allv.cpu().numpy()
    total = gathered.sum(dim=0).cpu().numpy()           # the fold snps_summary does: per-species counters over the ranks
    batch.close()
    if rank != 0:
        return None
    own_ms = (allv[:, 0] / steps * 1e3).tolist()
    kern_ms = allv[:, 2].tolist()
    job = float(allv[:, 1].max())
    sites = float(allv[:, 3].sum())
    return {
        "workload": WORKLOADS["c4"], "scaling": "strong", "steps": steps,
        "value": sites * steps / job, "unit": "sites/s", "ms_per_step": job / steps * 1e3,
        "total_sites": int(sites), "total_reads": int(allv[:, 4].sum()),
        "ranks": int(world), "collective": "RCCL all_gather_into_tensor of int64[n_species=%d][%d] per rank (torch.distributed "
                                           "backend nccl, world size %d)" % (n_sp, abi.NUM_STATS, dist.get_world_size() if collective else 1)
                                           if collective else "none (one rank, no --force-collective)",
        "per_rank_ms_per_step": own_ms, "per_rank_kernels_ms": kern_ms,
        "rank_time_max_over_mean": max(own_ms) / (sum(own_ms) / len(own_ms)),
        "partition": {"items": share["n_items"], "weight_max_over_mean": share["imbalance"]},
        "roofline_frac_slowest_rank": float((allv[:, 5] / (allv[:, 2] * 1e-3) / 1e9 / HBM_PEAK_GBPS).min()),
        "reads_counted_once": bool(int(total[:, abi.STAT_ALIGNED_READS].sum()) == int(allv[:, 4].sum())),
    }


def native_rccl(ctx, rank, world, n_species=100, timeout_s=90.0):
    """The PRODUCT's exchange (midas_amd/dist.py -> midas_comm_*, comm.cpp: librccl's ncclCommInitRank / ncclAllGather bound by the
    library itself, no process group) exercised beside the harness's torch.distributed: the per-species summary rows of every
    rank, all-gathered over xGMI, held to what torch's all-gather returns.  Time-bounded (a thread): a hang here must not cost
    the bench line.  Returns a dict on every rank."""
    import threading
    import torch
    import torch.distributed as dist
    from midas_amd import abi
    res = {}

    def work():
        try:
            torch.cuda.set_device(ctx.device)        # (the current device is per thread)
            ident = torch.zeros(128, dtype=torch.uint8, device="cuda:%d" % ctx.device)
            if rank == 0:
                ident.copy_(torch.tensor(list(abi.Comm.unique_id()), dtype=torch.uint8).to(ident.device))
            if world > 1:
                dist.broadcast(ident, 0)
            t0 = time.perf_counter()
            comm = abi.Comm(ctx, bytes(ident.cpu().tolist()), rank, world)
            init_ms = (time.perf_counter() - t0) * 1e3
            rows = (np.arange(n_species * 5, dtype=np.int64).reshape(n_species, 5) + 1) * (rank + 1)
            comm.all_gather(rows.tobytes())      # (channels are built on first use)
            t0 = time.perf_counter()
            got = comm.all_gather(rows.tobytes())
            ms = (time.perf_counter() - t0) * 1e3
            ok = all(np.array_equal(np.frombuffer(got[r], np.int64).reshape(n_species, 5), rows // (rank + 1) * (r + 1)) for r in range(world))
            back = comm.all_to_all_v([bytes([rank, r]) * 1000 for r in range(world)], [2000] * world)
            ok = ok and all(back[r] == bytes([r, rank]) * 1000 for r in range(world))
            comm.close()
            res.update(ok=bool(ok), comm_init_ms=init_ms, all_gather_ms=ms, ranks=world,
                       what="midas_comm_create (ncclCommInitRank) + midas_comm_all_gather of int64[%d][5] per rank + midas_comm_all_to_all_v, "
                            "host buffers in and out" % n_species)
        except Exception as e:      # noqa: BLE001
            res.update(ok=False, error="%s: %s" % (type(e).__name__, e))
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        res.update(ok=False, error="timed out after %d s" % timeout_s, hung=True)
    return res


def merge50(ctx, n_sites=2_000_000, n_samples=50, reps=3, cpu_sites=150_000):
    """BASELINE configs[4]: `merge_midas.py snps` across 50 samples -- the per-site cross-sample arithmetic of
    midas/merge/snps.py:13-114, 324-364 (pooled counts, major / minor allele, per-sample depth and minor-allele count, prevalence,
    the site filter) for one species' sites through midas_merge_sites (merge_sites.hip).  Not the graded metric: a block of its
    own with its own roofline and CPU leg (oracle/merge_oracle.py, the reference's loops restated, on a slice)."""
    from midas_amd import abi
    from oracle import merge_oracle as mo
    rng = np.random.default_rng(20260927 + 5)
    ref = rng.integers(0, 4, n_sites)
    alt = (ref + rng.integers(1, 4, n_sites)) % 4
    snp = rng.random(n_sites) < 0.03
    rows = np.arange(n_sites)
    counts = []
    for _ in range(n_samples):
        depth = rng.poisson(10.0, n_sites).astype(np.uint32)
        na = np.where(snp, rng.binomial(depth, 0.3), 0).astype(np.uint32)
        c = np.zeros((n_sites, 4), np.uint32)
        c[rows, ref] = depth - na
        c[rows, alt] += na
        counts.append(c)
    mean = [10.0] * n_samples
    margs = dict(abi.DEFAULT_MERGE_ARGS)
    prm = abi.MergeParams.from_args(margs)
    alg = n_sites * (24 * n_samples + 40)      # per (site, sample) 16 B of counts in, 8 B out (depth, minor count); 40 B of per-site outputs
    ms, res = [], None
    for _ in range(reps):
        res = ctx.merge_sites(prm, counts, mean)
        ms.append(res['kernel_ms'])
    k_ms = float(np.mean(ms[1:])) if len(ms) > 1 else float(ms[0])
    # CPU leg + parity on a slice: the reference's own per-site loops, restated
    sel = rng.choice(n_sites, size=cpu_sites, replace=False)
    t0, bad = time.perf_counter(), 0
    for i in sel:
        c = [[int(x) for x in counts[s][i]] for s in range(n_samples)]
        pooled = mo.pooled_counts(c)
        major, minor, st = mo.call_alleles(pooled, margs['allele_freq'])
        mafs, depths = mo.per_sample(c, major, minor)
        cs, prev = mo.prevalence(mean, depths, margs['site_depth'], margs['site_ratio'])
        why = mo.flag_reason(prev, st, margs['site_prev'], margs['snp_type'])
        ok = (res['major'][i] == (255 if major is None else major) and res['minor'][i] == (255 if minor is None else minor)
              and res['snp_type'][i] == [None, 'mono', 'bi', 'tri', 'quad'].index(st) and res['count_samples'][i] == cs
              and res['flag'][i] == {None: 0, 'min_prev': 1, 'snp_type': 2}[why] and list(res['depth'][:, i]) == depths
              and list(res['pooled'][i]) == pooled)
        bad += not ok
    cpu_s = time.perf_counter() - t0
    ach = alg / (k_ms * 1e-3) / 1e9
    return {"metric": "sites/sec, cross-sample SNP merge (midas_merge_sites)", "value": n_sites / (k_ms * 1e-3), "unit": "sites/s",
            "workload": "BASELINE configs[4]: %d samples x %d sites of one species (synthetic: Poisson depth 10, 3 %% of the sites "
                        "bi-allelic), merge_midas.py snps defaults" % (n_samples, n_sites),
            "kernel_ms": k_ms, "kernel_ms_runs": ms, "timing": "HIP events around merge_sites_kernel, inputs resident in HBM",
            "roofline": {"bound": "hbm", "kernel": "merge_sites_kernel", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": int(alg),
                         "algorithmic_bytes": "per site 16 B of counts in and 8 B out per sample + 40 B of per-site results",
                         "traffic": None},
            "cpu_baseline": {"value": cpu_sites / cpu_s, "unit": "sites/s", "cores": 1, "kind": "port",
                             "sample": "%d of the sites through oracle/merge_oracle.py (the reference's per-site loops, Python), %.1f s" % (cpu_sites, cpu_s)},
            "parity_vs_oracle": bad == 0, "sites_kept": int((res['flag'] == 0).sum())}


def bam_decode(ctx, contigs, reads, reps=3):
    """The stage's first half beside the graded line (rank 0, N = 1): the workload's own reads written as a BAM (the library's
    writer: zlib level 6 BGZF blocks, what samtools writes) and decoded back -- by the device (midas_bam_load_device: blocks up
    the link, Huffman decode + placement + CRC-32, record walk, columns and payload cut in kernels; SEQ / QUAL / CIGAR stay in
    HBM) and by the host's threads (midas_bam_open) -- wall time of the call, the file in the page cache.  Replaces
    `pysam.AlignmentFile(...)` + the record iteration behind count_coverage (midas/run/snps.py:186-199)."""
    import shutil
    import tempfile
    from midas_amd import abi
    work = tempfile.mkdtemp(prefix="midas_bench_bam_")
    try:
        path = os.path.join(work, "genomes.bam")
        refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
        ids = getattr(contigs, "ids", None) or ["c%d" % i for i in range(contigs.n_contigs)]
        t0 = time.perf_counter()
        abi.write_bam(path, ids, [int(x) for x in contigs.length], refid, reads)
        t_write = time.perf_counter() - t0
        size = os.path.getsize(path)
        dev, host, same = [], [], True
        for k in range(reps):
            t0 = time.perf_counter()
            d = abi.read_bam(path, ctx, payload_on_device=True)
            dev.append(time.perf_counter() - t0)
            if k == 0:       # the columns against the reads that were written (the payload fetched back from HBM)
                got = ctx.fetch_payload(d[3])
                same = bool(got.n_reads == reads.n_reads and np.array_equal(got.pos, reads.pos) and np.array_equal(got.nm, reads.nm)
                            and np.array_equal(got.l_seq, reads.l_seq) and np.array_equal(got.mapq, reads.mapq)
                            and np.array_equal(got.seq4, reads.seq4) and np.array_equal(got.qual, reads.qual)
                            and np.array_equal(got.cigar, reads.cigar))
                del got
            del d
        res = []
        for k in range(reps):       # ... and as the stage decodes it since round 6: everything resident, in the kernel's own layout
            t0 = time.perf_counter()
            d = abi.read_bam(path, ctx, resident=True)
            res.append(time.perf_counter() - t0)
            del d
        for k in range(2):
            t0 = time.perf_counter()
            d = abi.read_bam(path)
            host.append(time.perf_counter() - t0)
            del d
        best = min(dev)
        return {"metric": "BAM decoded to columns (whole call, file in the page cache)", "bam_bytes": int(size), "records": int(reads.n_reads),
                "device_decode_s": best, "device_decode_s_runs": dev, "device_resident_decode_s": min(res), "device_resident_decode_s_runs": res, "host_threads_decode_s": min(host), "host_threads": int(abi.load_library().midas_snps_cpu_budget()),
                "compressed_GBps": size / best / 1e9, "records_per_s": reads.n_reads / best,
                "columns_equal_what_was_written": same, "bam_write_s": t_write,
                "phases": "MIDAS_SNPS_TRACE=1 prints them; profiles/r06_e2e_stage_c3.txt"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def stage_e2e(contigs, reads, oracle_out, args, reps=3):
    """SURVEY 8(d)'s second timing beside the graded line (rank 0, N = 1): the END-TO-END pileup stage on this workload -- its BAM
    and FASTA files on disk (tmpfs when the box has one) -> every <species>.snps.gz + summary.txt through the code `run_midas.py
    snps --pileup` runs (midas_amd.run.snps.run_pipeline: index_bam, pysam_pileup, snps_summary; midas/run/snps.py:298-302),
    device decode as the product chooses it (--device_inflate auto).  Seconds of the whole call, the stage's own phases
    (midas_amd.run.snps.PHASES), and the files held to the oracle: summary.txt's counters, and one contig's rows of one species'
    table, text against text."""
    import contextlib
    import gzip
    import io
    import zlib
    from midas_amd import synth, utility
    from midas_amd.run import snps as msnps
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    work = tempfile.mkdtemp(prefix="midas_bench_stage_", dir=base)
    try:
        out, db = os.path.join(work, "sample"), os.path.join(work, "db")
        t0 = time.perf_counter()
        synth.write_sample(out, db, contigs, reads)
        t_setup = time.perf_counter() - t0
        bam_bytes = os.path.getsize(os.path.join(out, "snps", "temp", "genomes.bam"))
        runs, phases, species = [], None, None
        for k in range(reps):
            for f in os.listdir(os.path.join(out, "snps", "output")):
                os.remove(os.path.join(out, "snps", "output", f))
            sargs = dict(args, outdir=out, db=db, build_db=False, align=False, call=True, species_id=None, remove_temp=False,
                         threads=utility.cpu_budget(), log=io.StringIO(), device_inflate="auto")
            msnps.PHASES = []
            sink = io.StringIO()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(sink):
                msnps.run_pipeline(sargs)
            el = time.perf_counter() - t0
            ph = [(n, s) for n, s in msnps.PHASES if not n.strip().startswith("process start")]
            msnps.PHASES = None
            if not runs or el < min(runs):
                phases = ph
            runs.append(el)
        best = min(runs)
        # ---- the files against the oracle -----------------------------------------------------------------------------------
        st, _, oc, oa, os_ = oracle_out
        ok_summary = st == 0
        lines = open(os.path.join(out, "snps", "summary.txt")).read().split("\n")
        rows = {ln.split("\t")[0]: ln.split("\t") for ln in lines[1:] if ln}
        glen = np.bincount(contigs.species, weights=contigs.length, minlength=contigs.n_species).astype(np.int64)
        for i, sp in enumerate(contigs.species_ids):
            r = rows.get(sp)
            ok_summary = ok_summary and r is not None and [int(r[1]), int(r[2]), int(r[5]), int(r[6])] == \
                [int(glen[i]), int(os_[i, 2]), int(os_[i, 0]), int(os_[i, 1])] and \
                r[4] == (str(int(os_[i, 3]) / float(int(os_[i, 2]))) if os_[i, 2] else "0")
        sp0 = contigs.species_ids[0]
        mine = sorted(cid for cid, s in zip(contigs.ids, contigs.species) if s == 0)      # (the reference emits sorted(contig ids))
        k0 = contigs.ids.index(mine[0])
        off = contigs.site_offsets()
        c, al = oc[off[k0]:off[k0 + 1]].astype(np.int64), oa[off[k0]:off[k0 + 1]]
        want = ("ref_id\tref_pos\tref_allele\tdepth\tcount_a\tcount_c\tcount_g\tcount_t\n" + "".join(
            "%s\t%d\t%s\t%d\t%d\t%d\t%d\t%d\n" % (mine[0], i + 1, chr(al[i]), c[i].sum(), c[i, 0], c[i, 1], c[i, 2], c[i, 3])
            for i in range(c.shape[0]))).encode()
        text = gzip.open(os.path.join(out, "snps", "output", sp0 + ".snps.gz"), "rb").read()
        ok_rows = text[:len(want)] == want and text.count(b"\n") == int(glen[0]) + 1
        gz = sum(os.path.getsize(os.path.join(out, "snps", "output", f)) for f in os.listdir(os.path.join(out, "snps", "output")))
        tot = sum(s for n, s in phases if not n.startswith(" "))
        return {"metric": "genomic sites/sec, END-TO-END pileup stage (files in -> files out)", "value": contigs.n_sites / best, "unit": "sites/s",
                "seconds": best, "seconds_runs": runs,
                "what": "genomes.bam (%d MB, BGZF level 6) + %d genome.fna on %s -> %d <species>.snps.gz (%d MB) + summary.txt through "
                        "midas_amd.run.snps.run_pipeline (--pileup, --device_inflate auto, gzip level %d), in this process, files in the page cache"
                        % (bam_bytes // 1000000, contigs.n_species, "tmpfs" if base else "the temp directory's disk", contigs.n_species, gz // 1000000, msnps.GZ_LEVEL),
                "phases_s": [[n, round(s, 4)] for n, s in phases],
                "phases_note": "the stage's own laps, best run; indented names are parts of the line that follows them; the un-indented ones "
                               "add up to %.3f s of the %.3f s call" % (tot, best),
                "summary_equals_oracle": bool(ok_summary),
                "table_rows_equal_oracle": bool(ok_rows),
                "table_check": "species %s: header + every row of contig %s (%d rows) text against the C oracle's counts and alleles; line count = genome length + 1"
                               % (sp0, mine[0], c.shape[0]),
                "text_crc32_of_the_checked_table": "%08x" % (zlib.crc32(text) & 0xffffffff),
                "setup_s_not_timed": t_setup}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", help="workload (default c3 = BASELINE configs[2], the largest single-GPU "
                                                  "configuration; c2 = configs[1]; c4 = configs[3] sharded over the ranks)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 traffic measurement")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="untimed back-to-back steps in front of the timed region (the driver's GPU-busy sampler sees them)")
    ap.add_argument("--configs3", action="store_true",
                    help="add the configs3_strong block (BASELINE configs[3] dealt to the ranks) at N = 1 too; always on for N > 1")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the summary all-gather even with one rank (exercises the N>1 step on a 1-GPU box)")
    ap.add_argument("--no-merge50", action="store_true", help="skip the merge50 block (BASELINE configs[4], rank 0 at N = 1)")
    ap.add_argument("--no-stage", action="store_true", help="skip the stage_e2e block (the workload's files through run_pipeline; rank 0 at N = 1)")
    ap.add_argument("--no-bam-decode", action="store_true", help="skip the bam_decode block (the workload as a BAM, decoded on the device and by the host; rank 0 at N = 1)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.pmc_child:
        pmc_child(a.pmc_child, a.steps)
        return

    import torch
    import torch.distributed as dist
    from midas_amd import abi, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        sys.exit("--gpus (%d) != WORLD_SIZE (%d)" % (a.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    if _on_one_gpu():
        local_rank = 0
    torch.cuda.set_device(local_rank)
    collective = world > 1 or a.force_collective
    if collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if _on_one_gpu():
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    share = None
    if a.config == "c4":        # configs[3]: this rank's contigs of the one 400 Mb sample (strong scaling)
        contigs, reads, share = synth.c4_share(rank, world)
    else:
        cfg = dict(synth.CONFIGS[a.config])
        cfg["seed"] = cfg["seed"] + 1000 * rank       # every rank owns different species (weak scaling)
        contigs, reads = synth.make_dataset(**cfg)
    args = dict(abi.DEFAULT_ARGS)
    thr = abi.Thresholds.from_args(args)

    ctx = abi.Context(local_rank)
    # The step runs on a stream torch knows as its current one: torch's copies and the RCCL collectives order themselves
    # behind the kernels and the stats copy.  (Torch's DEFAULT stream has the null handle, which midas_snps_set_stream takes
    # for "the context's own stream" -- a stream torch does not wait for: the all-gather could read rows that were still
    # being written.  Found by tests/test_gpu_bench.py; hence a stream of our own, made current.)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    assert stream.cuda_stream != 0
    batch = ctx.batch(contigs, reads)            # uploads the BAM-native arrays as they are; picks the path
    info = batch.info()
    n_sp = contigs.n_species
    # N > 1: a rank's K steps are its share of the job; the job's one exchange -- the all-gather of every rank's summary
    # rows, what snps_summary needs to write summary.txt -- follows the last step, inside the timed region, exactly as
    # midas_amd/run/snps.py does it (all of a rank's contigs, then one all-gather).
    rows = torch.zeros((max(a.steps, a.warmup, 1), n_sp, abi.NUM_STATS), dtype=torch.int64, device="cuda")
    gathered = torch.zeros((world,) + tuple(rows.shape), dtype=torch.int64, device="cuda") if collective else None

    def job(n):
        for i in range(n):
            batch.run(thr)
            if collective:
                batch.stats_to_device(rows[i].data_ptr())
        if collective and n > 0:
            _all_gather(gathered, rows)

    job(a.warmup)
    batch.sync()
    # sustain: the same step, back to back, untimed -- the timed region below is a few dozen milliseconds, too short for a
    # once-a-second GPU-busy sampler to notice
    sustain_steps = 0
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < a.sustain_seconds:
        for _ in range(50):
            batch.run(thr)
        batch.sync()
        sustain_steps += 50
    torch.cuda.synchronize()
    batch.enable_timing(a.steps)                 # HIP events on the step's stream: before the index pass, between it and the
    if collective:      # RCCL builds its communicator and channels on first use: never inside the timed region
        _all_gather(gathered, rows)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    batch.sync()   # surfaces MIDAS_SNPS_ERR_READ_* of the last run, if any
    if collective and not torch.equal(gathered[rank], rows):      # the gathered table holds this rank's rows where they belong
        sys.exit("bench.py: all-gathered summary rows differ from this rank's rows")

    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    sites = torch.tensor([float(info.n_sites)], dtype=torch.float64, device="cuda")
    el_all = None
    if collective:
        el_all = torch.zeros(world, dtype=torch.float64, device="cuda")
        _all_gather(el_all, el)
        _all_reduce(el, dist.ReduceOp.MAX)
        _all_reduce(sites, dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_sites = float(sites.item())

    tm = [batch.timing(i) for i in range(a.steps)]
    index_ms = float(np.mean([t["index_ms"] for t in tm]))
    pile_ms = float(np.mean([t["pileup_ms"] for t in tm]))
    step_kernels_ms = index_ms + pile_ms
    path = abi.PATH_NAMES[info.path]

    # ---- secondary region: the step over a resident PACKED batch (the layout is built once, outside the region) ----------
    packed = None
    if world == 1:
        try:
            batch.select_path(abi.PATH_PACKED)
            kp = max(1, a.steps)
            for _ in range(3):
                batch.run(thr)
            batch.sync()
            batch.enable_timing(kp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(kp):
                batch.run(thr)
            torch.cuda.synchronize()
            el_p = time.perf_counter() - t0
            batch.sync()
            ptm = [batch.timing(i) for i in range(kp)]
            p_idx = float(np.mean([t["index_ms"] for t in ptm]))
            p_pil = float(np.mean([t["pileup_ms"] for t in ptm]))
            packed = {"value": info.n_sites * kp / el_p, "ms_per_step": el_p / kp * 1e3, "steps": kp,
                      "index_kernel_ms": p_idx, "pileup_tiles_kernel_ms": p_pil,
                      "pileup_tiles_kernel_achieved_GBps": info.algorithmic_bytes / (p_pil * 1e-3) / 1e9,
                      "note": "index + pileup over the packed records and one-byte-per-base payload of pack_reads.hip, which is "
                              "built ONCE outside this region (3.3 ms per pack on this workload): not the path's figure"}
        finally:
            batch.select_path(abi.PATH_AUTO)

    strong = None
    if (world > 1 or a.configs3) and a.config != "c4":
        # (N > 1: the block becomes the line's own figure below, so it runs exactly the K steps the line reports)
        strong = configs3_strong(ctx, thr, rank, world, collective, a.steps if world > 1 else max(1, min(a.steps, 50)))

    out = None
    if rank == 0:
        achieved = info.algorithmic_bytes / (step_kernels_ms * 1e-3) / 1e9
        pile_ach = info.algorithmic_bytes / (pile_ms * 1e-3) / 1e9
        traffic, traffic_src = load_pmc_traffic("direct_step" if path == "direct" else "pileup_tiles_kernel", a.config)
        out = {
            "metric": "genomic sites/sec pileup+allele-count",
            "value": total_sites * a.steps / elapsed,
            "unit": "sites/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if a.config == "c4" else "weak", "vs_baseline": None,
            "dtype": "u8/u32 integer tallies (fp64 only in the two keep_read ratio tests)",
            "data": "synthetic (seeded generator midas_amd/synth.py; SURVEY 8d distributions)",
            "config": {"workload": WORKLOADS.get(a.config, a.config),
                       "step": ("one 16-byte record per read (pos, l_seq | n_cigar, NM | mapq, payload offset) + the read's [cigar][seq][qual] "
                                "run in BAM's own order, resident in HBM -- the direct layout, a re-encoding of the BAM-native columns that "
                                "batch_create builds once per batch (roofline.layout_build_ms) and the device BAM decoder writes directly -- "
                                "-> per-site counts, alleles and per-species counters (ranges pass over the positions + pileup kernel, one "
                                "visit per read; path: %s)" % path) if path == "direct" else
                               "packed records + one byte per base resident in HBM -> per-site counts, alleles, counters (path: %s)" % path,
                       "sites_per_gpu": int(info.n_sites), "reads_per_gpu": int(info.n_reads),
                       "thresholds": args,
                       "parallelism": ("contig-sharded x%d (dist.shard_items), " % world if a.config == "c4" else
                                       "species-sharded x%d, " % world) + "one RCCL all-gather of the summary rows per job"
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm",
                         "kernel": "the step's kernels: direct_ranges_kernel (positions -> per-tile read ranges) + "
                                   "pileup_direct_kernel (one visit per read)" if path == "direct" else "index_reads_kernel + pileup_tiles_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(info.algorithmic_bytes),
                         "kernels_ms_avg": step_kernels_ms, "index_pass_ms_avg": index_ms, "pileup_kernel_ms_avg": pile_ms,
                         "layout_build_ms": info.layout_build_us / 1e3,
                         "layout_build_note": "device time of the three launches that re-encode the uploaded columns into records + payload, "
                                              "ONCE per batch, outside the timed step (nothing is decided in it: no CIGAR shape, no filter "
                                              "outcome); a batch over a device-decoded BAM has none -- the decoder writes the layout",
                         "timing": "HIP events on the step's stream around the index pass and around the pileup kernel, every "
                                   "timed step",
                         "dominant_kernel": {"name": "pileup_direct_kernel" if path == "direct" else "pileup_tiles_kernel",
                                             "ms_avg": pile_ms, "achieved": pile_ach, "frac": pile_ach / HBM_PEAK_GBPS}},
            "sustain": {"steps": sustain_steps, "seconds": a.sustain_seconds,
                        "note": "untimed back-to-back steps in front of the timed region (same work)"},
            "path": {"taken": path, "general_reads": int(info.direct_general_reads),
                     "reach": int(info.direct_reach), "stream_reads": int(info.direct_stream_reads),
                     "lanes_per_read": int(info.lanes_per_read), "lane_bases": int(info.lane_bases)},
        }
        if share is not None:
            out["config"]["sharding"] = {"items": share["n_items"], "imbalance_max_over_mean": share["imbalance"],
                                         "total_sites": share["total_sites"], "total_reads": share["total_reads"]}
        if el_all is not None:
            per = [float(x) / a.steps * 1e3 for x in el_all.cpu().tolist()]
            out["per_rank_ms_per_step"] = per
            out["rank_time_max_over_mean"] = max(per) / (sum(per) / len(per))
        if strong is not None:
            out["configs3_strong"] = strong
        if strong is not None and world > 1:
            # N > 1: the configuration BASELINE.json names for several GPUs is configs[3] -- ONE sample dealt to the ranks -- so THAT
            # is the line's value; the per-rank configs[2] replicas measured above move to a block of their own.  (N = 1 keeps
            # configs[2], the largest single-GPU configuration.  Efficiency against the N = 1 line compares two workloads -- 66 against
            # 50 algorithmic bytes per site: configs[3] whole on ONE GPU is in profiles/r06_bench_n1_c4_whole.json for that.)
            out["weak_replicas"] = {"value": out["value"], "unit": "sites/s", "ms_per_step": out["ms_per_step"], "scaling": "weak",
                                    "workload": out["config"]["workload"], "sites_per_gpu": out["config"]["sites_per_gpu"],
                                    "reads_per_gpu": out["config"]["reads_per_gpu"], "roofline": out["roofline"],
                                    "per_rank_ms_per_step": out.pop("per_rank_ms_per_step", None),
                                    "rank_time_max_over_mean": out.pop("rank_time_max_over_mean", None)}
            out["value"] = strong["value"]
            out["ms_per_step"] = strong["ms_per_step"]
            out["steps"] = strong["steps"]
            out["scaling"] = "strong"
            out["config"]["workload"] = strong["workload"]
            out["config"]["sites_per_gpu"] = strong["total_sites"] // world
            out["config"]["reads_per_gpu"] = strong["total_reads"] // world
            out["config"]["parallelism"] = "contig-sharded x%d (midas_amd.dist.shard_items), one RCCL all-gather of the summary rows per job" % world
            out["per_rank_ms_per_step"] = strong["per_rank_ms_per_step"]
            out["rank_time_max_over_mean"] = strong["rank_time_max_over_mean"]
            frac = strong["roofline_frac_slowest_rank"]
            out["roofline"] = {"bound": "hbm", "kernel": "the step's kernels on the SLOWEST rank: direct_ranges_kernel + pileup_direct_kernel over its share of configs[3]",
                               "achieved": frac * HBM_PEAK_GBPS, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac, "traffic": None,
                               "per_rank_kernels_ms": strong["per_rank_kernels_ms"],
                               "timing": "HIP events on every rank's stream around its index pass and pileup kernel, every timed step"}
            out["headline_note"] = ("N > 1: value = configs[3] (BASELINE's multi-GPU configuration), strong scaling; the configs[2] replicas "
                                    "per rank are in weak_replicas")
        if packed is not None:
            out["value_resident_packed"] = packed["value"]
            out["resident_packed"] = packed
        if world == 1:
            try:      # what this box streams right now: kernels built to saturate (4 workgroups per CU, 16 B per lane, 4 GiB buffers)
                rates = ctx.stream_rates(1 << 32, 3)
                ceil = max(rates.values())
                out["roofline"]["stream_rates_GBps_this_box"] = rates
                out["roofline"]["ceiling_GBps_this_box"] = ceil
                out["roofline"]["ceiling_method"] = ("midas_snps_stream_rates: read stream, write stream and copy over 4 GiB buffers, "
                                                     "3 passes each; the ceiling is the largest of the three")
                out["roofline"]["frac_of_ceiling_this_box"] = achieved / ceil
            except Exception as e:
                out["roofline"]["ceiling_error"] = str(e)
            if not a.no_pmc:
                live = measure_traffic(contigs, reads)
                if live is not None:
                    out["roofline"]["traffic_this_box"] = live
                    if live.get("hbm_bytes_per_step"):
                        out["roofline"]["traffic"] = live["hbm_bytes_per_step"]
                        out["roofline"]["traffic_source"] = "rocprofv3 --pmc on this box (roofline.traffic_this_box)"
                        out["roofline"]["traffic_over_algorithmic"] = live["hbm_bytes_per_step"] / info.algorithmic_bytes
                        if "ceiling_GBps_this_box" in out["roofline"]:
                            out["roofline"]["hbm_traffic_GBps"] = live["hbm_bytes_per_step"] / (step_kernels_ms * 1e-3) / 1e9
                            out["roofline"]["hbm_traffic_frac_of_ceiling_this_box"] = \
                                out["roofline"]["hbm_traffic_GBps"] / out["roofline"]["ceiling_GBps_this_box"]
                            # the step's traffic is a MIX of reads and writes: what this box gives each, one after the other
                            # (an optimistic bound: its own copy kernel, half and half, stays below the sum of its parts)
                            rd, wr = live.get("hbm_read_bytes"), live.get("hbm_write_bytes")
                            if rd and wr and rates.get("read_GBps") and rates.get("write_GBps"):
                                t_mix = rd / (rates["read_GBps"] * 1e9) + wr / (rates["write_GBps"] * 1e9)
                                out["roofline"]["mixed_traffic_floor_ms_this_box"] = t_mix * 1e3
                                out["roofline"]["frac_of_mixed_traffic_floor_this_box"] = t_mix * 1e3 / step_kernels_ms
        if world == 1 and not a.no_cpu:
            cb, cb_all, cb_species, ref = cpu_baseline(thr, contigs, reads, a.cpu_seconds)
            out["cpu_baseline"] = cb
            out["cpu_baseline_all_cores"] = cb_all
            out["cpu_baseline_reference_grain"] = cb_species
            batch.run(thr)
            counts, allele, stats = batch.fetch()
            st, _, oc, oa, os_ = ref
            out["parity_vs_oracle"] = bool(st == 0 and np.array_equal(counts, oc) and np.array_equal(allele, oa)
                                           and np.array_equal(stats, os_))
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
            out["speedup_vs_cpu_all_cores"] = out["value"] / cb_all["value"]
            if not a.no_merge50:
                try:
                    out["merge50"] = merge50(ctx)
                except Exception as e:      # (a block beside the graded line: never fail the bench on it)
                    out["merge50"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not a.no_stage:
                try:
                    out["stage_e2e"] = stage_e2e(contigs, reads, ref, args)
                except BaseException as e:      # (a block beside the graded line -- sys.exit of a stage included: never fail the bench on it)
                    out["stage_e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not a.no_bam_decode:
                try:
                    out["bam_decode"] = bam_decode(ctx, contigs, reads)
                except Exception as e:      # (a block beside the graded line: never fail the bench on it)
                    out["bam_decode"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                out["cpu_python_shaped_estimate"] = python_shaped_estimate(contigs, reads, args)
            except Exception as e:  # the estimate is a courtesy number; never fail the bench on it
                out["cpu_python_shaped_estimate"] = {"error": str(e)}
    native = None
    if collective and not _on_one_gpu():       # (every rank: the product's own RCCL binding beside the harness's process group)
        native = native_rccl(ctx, rank, world)
        if rank == 0:
            out["native_rccl"] = native
    if native is not None and native.get("hung"):       # (a thread sits in RCCL: say what was measured and leave at once)
        if rank == 0:
            sys.stdout.flush()
            print(json.dumps(out), flush=True)
        os._exit(0)
    batch.close()
    ctx.close()
    if collective:
        dist.destroy_process_group()
    if rank == 0:       # the JSON line is the last thing on stdout: push out what RCCL left in the C stdio buffer first
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
