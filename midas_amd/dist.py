"""Multi-GPU plumbing for the pileup stage: one process per GPU, contigs sharded over ranks, and a single
all-gather of the per-species summary rows (RCCL over xGMI when the backend is "nccl", gloo in CPU tests).

The reference's only parallelism is `mp.Pool(threads)` with one task per species whose return value,
(species_id, aln_stats), is pickled back through a pipe (midas/run/snps.py:225-228, midas/utility.py:81-107).
Here the unit of work is the contig -- the unit `count_coverage` is called on (midas/run/snps.py:187-199) -- so one
species keeps every GPU busy; per-site output never leaves its rank (the owner writes its contigs' part of
<species>.snps.gz, parts are concatenated in sorted-contig order), and only [n_species, 5] int64 counters (partial sums
per rank) are exchanged.
"""

import datetime
import os
import sys

import numpy as np

_STAT_COLS = 5   # genome_length, covered_bases, total_depth, aligned_reads, mapped_reads


def _alone():
    """One process and no process group can exist: torch is not even imported (a second and a half of the single-GPU command)."""
    return int(os.environ.get("WORLD_SIZE", "1") or 1) <= 1 and 'torch.distributed' not in sys.modules


def world():
    if _alone():
        return 0, 1
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# Rank 0 builds the bowtie2 database and aligns (tens of minutes to hours) while the other ranks wait at a barrier: the
# default collective timeout (10 min with nccl) would let the watchdog kill the job before the pileup starts.
COLLECTIVE_TIMEOUT = datetime.timedelta(hours=48)


def init_from_env(device_backend=None):
    """Join the process group torchrun described (RANK / WORLD_SIZE / MASTER_*); no-op for a single process."""
    if _alone():
        return 0, 1
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
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=COLLECTIVE_TIMEOUT)
    else:
        dist.init_process_group(backend, timeout=COLLECTIVE_TIMEOUT)
    return world()


def exit_message(e):
    """What a caught SystemExit means for agree_or_exit: sys.exit(), sys.exit(None) and sys.exit(0) are NOT failures (None);
    a string is the message the reference's sys.exit("\nError: ...") carries; any other status becomes a message naming it."""
    code = e.code if isinstance(e, SystemExit) else e
    if code is None or code == 0:
        return None
    if isinstance(code, str):
        return code if code else "\nError: a stage stopped without a message\n"
    return "\nError: a stage stopped with exit status %r\n" % (code,)


def agree_or_exit(error_message=None):
    """Every rank calls this in front of a collective with its own error (None = fine).  If any rank failed, ALL ranks
    leave together -- the failing ones with their message, as the reference's sys.exit("\nError: ...") would, the others
    naming the failed ranks -- instead of one rank exiting and its peers blocking in the collective until the
    watchdog kills them."""
    rank, ws = world()
    if ws == 1:
        if error_message is not None:
            sys.exit(error_message)
        return
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if error_message is not None else 0], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        flag = flag.cuda()
    flags = torch.zeros(ws, dtype=torch.int64, device=flag.device)
    dist.all_gather_into_tensor(flags, flag)
    failed = [r for r, f in enumerate(flags.cpu().tolist()) if f]
    if not failed:
        return
    if error_message is not None:
        sys.exit(error_message)
    sys.exit("\nError: rank(s) %s failed, see their message; rank %d stops with them\n" % (failed, rank))


def shard_species(weights, n_ranks):
    """Longest-processing-time bin packing of work items (contigs for the snps pileup, species for genes / merge) onto
    ranks.

    weights: {item: cost} (bytes of aligned reads + sites is a good proxy).  Deterministic: ties are
    broken by item id, so every rank computes the same assignment without talking.
    Returns {item: rank}."""
    load = [0.0] * n_ranks
    owner = {}
    for sp in sorted(weights, key=lambda s: (-weights[s], s)):
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        owner[sp] = r
        load[r] += weights[sp]
    return owner


shard_items = shard_species


def all_gather_summary(rows):
    """rows: int64 [n_species_total, 5], zero outside the species this rank owns.
    One all-gather of the rows (<= 100 species x 40 B per rank), then a local sum over ranks."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    rank, ws = world()
    if ws == 1:
        return rows.copy()
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend() == "nccl"
    t = torch.from_numpy(rows)
    if on_gpu:
        t = t.cuda()
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)     # rank-major concatenation along dim 0
    return out.reshape((ws,) + tuple(t.shape)).sum(dim=0).cpu().numpy()


def all_gather_i64(values):
    """values: int64 array of the same length on every rank -> [world, len] (rank-major)."""
    v = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
    rank, ws = world()
    if ws == 1:
        return v.reshape(1, -1).copy()
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(v)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = torch.empty(ws * t.shape[0], dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.reshape(ws, -1).cpu().numpy()


def all_gather_rows_f64(rows):
    """Like all_gather_summary for float64 rows (the genes summary has means and medians): rows are zero outside the
    species this rank owns, so the sum over ranks is the owner's row (nan stays nan)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    rank, ws = world()
    if ws == 1:
        return rows.copy()
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(rows)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.reshape((ws,) + tuple(t.shape)).sum(dim=0).cpu().numpy()


def all_to_all_v(parts):
    """parts: one 1-D array per rank (same dtype everywhere), parts[r] goes to rank r -> the arrays received, by source rank.
    One exchange of the counts (all-gather), then one all_to_all_single with uneven splits (RCCL on GPUs, gloo on CPUs)."""
    rank, ws = world()
    parts = [np.ascontiguousarray(p) for p in parts]
    assert len(parts) == ws
    if ws == 1:
        return [parts[0].copy()]
    import torch
    import torch.distributed as dist
    dtype = parts[0].dtype
    counts = all_gather_i64(np.array([p.size for p in parts], np.int64))        # counts[src, dst]
    send_n = [int(x) for x in counts[rank]]
    recv_n = [int(x) for x in counts[:, rank]]
    # (as bytes: every dtype travels the same way)
    send = torch.from_numpy(np.concatenate(parts).view(np.uint8) if sum(send_n) else np.zeros(0, np.uint8))
    isz = dtype.itemsize
    on_gpu = dist.get_backend() == "nccl"
    if on_gpu:
        send = send.cuda()
    recv = torch.empty(sum(recv_n) * isz, dtype=torch.uint8, device=send.device)
    dist.all_to_all_single(recv, send, output_split_sizes=[n * isz for n in recv_n], input_split_sizes=[n * isz for n in send_n])
    got = recv.cpu().numpy().view(dtype)
    out, at = [], 0
    for n in recv_n:
        out.append(got[at:at + n].copy())
        at += n
    return out


def barrier():
    if _alone():
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
