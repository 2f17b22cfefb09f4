"""Multi-GPU plumbing for the pileup stage: one process per GPU, species sharded over ranks, and a single
all-gather of the per-species summary rows (RCCL over xGMI when the backend is "nccl", gloo in CPU tests).

The reference's only parallelism is `mp.Pool(threads)` with one task per species whose return value,
(species_id, aln_stats), is pickled back through a pipe (midas/run/snps.py:225-228, midas/utility.py:81-107).
Here a species lives on exactly one rank, per-site output never leaves its rank (the owner writes the
<species>.snps.gz), and only [n_species, 5] int64 counters are exchanged.
"""

import os

import numpy as np

_STAT_COLS = 5   # genome_length, covered_bases, total_depth, aligned_reads, mapped_reads


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(device_backend=None):
    """Join the process group torchrun described (RANK / WORLD_SIZE / MASTER_*); no-op for a single process."""
    import torch
    import torch.distributed as dist
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or dist.is_initialized():
        return world()
    backend = device_backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world()


def shard_species(weights, n_ranks):
    """Longest-processing-time bin packing of species onto ranks.

    weights: {species_id: cost} (aligned reads + genome length is a good proxy).  Deterministic: ties are
    broken by species id, so every rank computes the same assignment without talking.
    Returns {species_id: rank}."""
    load = [0.0] * n_ranks
    owner = {}
    for sp in sorted(weights, key=lambda s: (-weights[s], s)):
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        owner[sp] = r
        load[r] += weights[sp]
    return owner


def all_gather_summary(rows):
    """rows: int64 [n_species_total, 5], zero outside the species this rank owns.
    One all-gather of the rows (<= 100 species x 40 B per rank), then a local sum over ranks."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    rank, ws = world()
    if ws == 1:
        return rows.copy()
    on_gpu = dist.get_backend() == "nccl"
    t = torch.from_numpy(rows)
    if on_gpu:
        t = t.cuda()
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)     # rank-major concatenation along dim 0
    return out.reshape((ws,) + tuple(t.shape)).sum(dim=0).cpu().numpy()


def all_gather_rows_f64(rows):
    """Like all_gather_summary for float64 rows (the genes summary has means and medians): rows are zero outside the
    species this rank owns, so the sum over ranks is the owner's row (nan stays nan)."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    rank, ws = world()
    if ws == 1:
        return rows.copy()
    t = torch.from_numpy(rows)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.reshape((ws,) + tuple(t.shape)).sum(dim=0).cpu().numpy()


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
