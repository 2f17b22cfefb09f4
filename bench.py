#!/usr/bin/env python
"""bench.py -- genomic sites/sec of the MI355X SNP pileup (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one resident batch: the device builds its
per-tile read index (the index_bam analogue), filters every read (keep_read), walks the
CIGARs, tallies A/C/G/T per site, emits counts + ref allele and reduces the per-species
counters.  A second timed region (`value_incl_pack`) puts the device packer in front of every step: from the
BAM-native arrays resident in HBM (pos, mapq, NM, l_seq, CSR offsets, 4-bit SEQ, QUAL, CIGAR) -- CIGAR -> match
segments, mean quality by wave reduction, N-mask, tile order (radix sort), records + payload -- to the counts.  With N > 1 a rank's K steps are its share of the job and the job's one exchange -- the
all-gather of every rank's per-species summary rows over RCCL -- follows them, inside the timed region.  Inputs
(packed reads, reference letters) are resident in HBM before the timed region starts.

Workload (default): BASELINE.json configs[2], the largest single-GPU configuration -- 20 species, 80 Mb,
10 666 667 aligned synthetic 150 bp reads (20x) per GPU (`--config c2` = configs[1]: 1 species, 15 Mb, 1 M reads,
10x; `--config c4_rank` = one rank's share of configs[3]).  Multi-GPU is species-sharded weak scaling: every rank
owns its own species (own seed), no data-path collective, one all-gather of [K, n_species, 4] int64 summary rows
per job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


WORKLOADS = {
    "c2": "configs[1]: 1 species rep-genome (60 contigs x 250 kb = 15 Mb), 1M synthetic 150 bp reads at 10x per GPU",
    "c3": "configs[2]: 20 species (320 contigs x 250 kb = 80 Mb), 10 666 667 aligned synthetic 150 bp reads at 20x per GPU",
    "c4_rank": "one rank's share of configs[3]: 13 species (52 Mb), 10.4M aligned synthetic 150 bp reads at 30x",
}


def load_pmc_traffic(kernel, workload):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from) -- PMC counters cannot be read
    inside a plain run, so the figure is the one of the committed rocprofv3 --pmc passes of this same command."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        e = d.get(workload, {}).get(kernel)
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
    # all the CPUs this process may use: the hardware threads, or the cgroup's quota when that is less (a container that
    # shows 256 threads under a 16-CPU quota gets 16 CPUs' worth of work per second however many threads it starts)
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


def copy_ceiling_gbps(torch, nbytes=1 << 30, reps=10):
    """What a plain device-to-device copy reaches on THIS box right now (bytes read + bytes written per second): the
    practical HBM ceiling the guide quotes as ~6.3 TB/s, measured live because boxes of the pool differ by a few percent."""
    src = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda").random_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", help="workload from midas_amd.synth.CONFIGS (default c3 = BASELINE configs[2], "
                                                  "the largest single-GPU configuration; c2 = configs[1])")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pack-steps", type=int, default=20, help="steps of the second region (device packer + step)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the summary all-gather even with one rank (exercises the N>1 step on a 1-GPU box)")
    a = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    collective = world > 1 or a.force_collective
    if collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = dict(synth.CONFIGS[a.config])
    cfg["seed"] = cfg["seed"] + 1000 * rank       # every rank owns different species (weak scaling)
    contigs, reads = synth.make_dataset(**cfg)
    args = dict(abi.DEFAULT_ARGS)
    thr = abi.Thresholds.from_args(args)

    ctx = abi.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)           # launch on torch's stream: torch events / RCCL see the kernels
    batch = ctx.batch(contigs, reads)
    info = batch.info()
    n_sp = contigs.n_species
    # N > 1: a rank's K steps are its share of the job (K species batches); the job's one exchange -- the all-gather of
    # every rank's summary rows, what snps_summary needs to write summary.txt -- follows the last step, inside the timed
    # region, exactly as midas_amd/run/snps.py does it (all of a rank's species, then one all-gather).
    rows = torch.zeros((max(a.steps, a.warmup, 1), n_sp, abi.NUM_STATS), dtype=torch.int64, device="cuda")
    gathered = torch.zeros((world,) + tuple(rows.shape), dtype=torch.int64, device="cuda") if collective else None

    def job(n):
        for i in range(n):
            batch.run(thr)
            if collective:
                batch.stats_to_device(rows[i].data_ptr())
        if collective and n > 0:
            dist.all_gather_into_tensor(gathered, rows)

    # warm-up runs are timed with all three events (index kernel, pileup kernel); the timed region records only the two
    # around the pileup kernel, whose average feeds `roofline` -- every event record takes ~4 us of stream time
    if a.warmup > 0:
        batch.enable_timing(a.warmup)
    job(a.warmup)
    batch.sync()
    torch.cuda.synchronize()
    index_ms = float(np.median([batch.timing(i)["index_ms"] for i in range(a.warmup)])) if a.warmup > 0 else None
    batch.enable_timing(a.steps)
    batch.time_pileup_only(True)
    if collective:      # RCCL builds its communicator and channels on first use: never inside the timed region
        dist.all_gather_into_tensor(gathered, rows)
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
    if collective:      # the gathered table holds this rank's rows where they belong
        if not torch.equal(gathered[rank], rows):
            sys.exit("bench.py: all-gathered summary rows differ from this rank's rows")

    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    sites = torch.tensor([float(info.n_sites)], dtype=torch.float64, device="cuda")
    if collective:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(sites, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_sites = float(sites.item())

    tm = [batch.timing(i) for i in range(a.steps)]
    pile_ms = float(np.mean([t["pileup_ms"] for t in tm]))
    run_ms = pile_ms + index_ms if index_ms is not None else None

    # ---- second region: the device packer in front of every step (raw BAM-native arrays resident in HBM -> counts) ----
    kp = max(1, min(a.steps, a.pack_steps))
    batch.enable_timing(kp)
    batch.time_pileup_only(True)
    batch.pack()
    batch.run(thr)
    batch.sync()
    batch.enable_timing(kp)
    batch.time_pileup_only(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(kp):
        batch.pack()
        batch.run(thr)
    torch.cuda.synchronize()
    elapsed_pack = time.perf_counter() - t0
    batch.sync()
    ptm = [batch.pack_timing(i) for i in range(kp)]
    pack_ms = float(np.mean([t["pack_ms"] for t in ptm]))
    scatter_ms = float(np.mean([t["scatter_ms"] for t in ptm]))
    elp = torch.tensor([elapsed_pack], dtype=torch.float64, device="cuda")
    if collective:
        dist.all_reduce(elp, op=dist.ReduceOp.MAX)
    elapsed_pack = float(elp.item())

    out = None
    if rank == 0:
        kernel = "pileup_tiles_kernel"
        achieved = info.algorithmic_bytes / (pile_ms * 1e-3) / 1e9
        traffic, traffic_src = load_pmc_traffic(kernel, a.config)
        out = {
            "metric": "genomic sites/sec pileup+allele-count",
            "value": total_sites * a.steps / elapsed,
            "unit": "sites/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 integer tallies (fp64 only in the two keep_read ratio tests)",
            "data": "synthetic (seeded generator midas_amd/synth.py; SURVEY 8d distributions)",
            "config": {"workload": WORKLOADS.get(a.config, a.config),
                       "sites_per_gpu": int(info.n_sites), "reads_per_gpu": int(info.n_reads),
                       "thresholds": args, "parallelism": "species-sharded x%d, one RCCL all-gather of the summary rows per job" % world
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(info.algorithmic_bytes),
                         "kernel_ms_avg": pile_ms, "index_kernel_ms_warmup_median": index_ms, "kernels_ms_per_step": run_ms,
                         "frac_of_measured_copy_ceiling_6290": achieved / 6290.0},
        }
        if world == 1:
            try:      # the practical ceiling of this box, measured the same minute
                ceil = copy_ceiling_gbps(torch)
                out["roofline"]["copy_ceiling_GBps_this_box"] = ceil
                out["roofline"]["frac_of_copy_ceiling_this_box"] = achieved / ceil
                if traffic:
                    out["roofline"]["hbm_traffic_GBps"] = traffic / (pile_ms * 1e-3) / 1e9
                    out["roofline"]["hbm_traffic_frac_of_copy_ceiling_this_box"] = traffic / (pile_ms * 1e-3) / 1e9 / ceil
            except Exception as e:
                out["roofline"]["copy_ceiling_error"] = str(e)
        # pack + index + pileup from the resident raw arrays; the packer's dominant kernel reads the raw read
        # (SURVEY 8d's per-read figure) and writes its records + payload
        read_alg = int(info.algorithmic_bytes) - 17 * int(info.n_sites)
        pack_alg = read_alg + int(info.packed_bytes)
        pack_ach = pack_alg / (scatter_ms * 1e-3) / 1e9
        out["value_incl_pack"] = total_sites * kp / elapsed_pack
        out["ms_per_step_incl_pack"] = elapsed_pack / kp * 1e3
        out["steps_incl_pack"] = kp
        pack_traffic, pack_traffic_src = load_pmc_traffic("pack_scatter_kernel", a.config)
        out["roofline_pack"] = {"bound": "hbm", "kernel": "pack_scatter_kernel", "achieved": pack_ach, "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": pack_ach / HBM_PEAK_GBPS, "traffic": pack_traffic,
                                "traffic_source": pack_traffic_src,
                                "algorithmic_bytes_per_launch": pack_alg,
                                "algorithmic_bytes_note": "raw reads in (sum(ceil(l/2) + l + 4*n_cigar + 16)) + records and payload out",
                                "kernel_ms_avg": scatter_ms, "pack_ms_avg_all_kernels": pack_ms}
        if world == 1 and not a.no_cpu:
            cb, cb_all, cb_species, ref = cpu_baseline(thr, contigs, reads, a.cpu_seconds)
            out["cpu_baseline"] = cb
            out["cpu_baseline_all_cores"] = cb_all
            out["cpu_baseline_reference_grain"] = cb_species
            counts, allele, stats = batch.fetch()
            st, _, oc, oa, os_ = ref
            out["parity_vs_oracle"] = bool(st == 0 and np.array_equal(counts, oc) and np.array_equal(allele, oa)
                                           and np.array_equal(stats, os_))
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
            out["speedup_vs_cpu_all_cores"] = out["value"] / cb_all["value"]
            try:
                out["cpu_python_shaped_estimate"] = python_shaped_estimate(contigs, reads, args)
            except Exception as e:  # the estimate is a courtesy number; never fail the bench on it
                out["cpu_python_shaped_estimate"] = {"error": str(e)}
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
