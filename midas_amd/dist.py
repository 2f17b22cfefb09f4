"""Multi-GPU plumbing for the pileup stage: one process per GPU, contigs sharded over ranks, and a single
all-gather of the per-species summary rows -- RCCL over xGMI.

The reference's only parallelism is `mp.Pool(threads)` with one task per species whose return value,
(species_id, aln_stats), is pickled back through a pipe (midas/run/snps.py:225-228, midas/utility.py:81-107).
Here the unit of work is the contig -- the unit `count_coverage` is called on (midas/run/snps.py:187-199) -- so one
species keeps every GPU busy; per-site output never leaves its rank (the owner writes its contigs' part of
<species>.snps.gz, parts are concatenated in sorted-contig order), and only [n_species, 5] int64 counters (partial sums
per rank) are exchanged.

Two transports behind the same functions:

* NATIVE (the product's, `init_from_env(rendezvous_dir=...)` under torchrun's RANK / WORLD_SIZE / LOCAL_RANK): no torch, no
  process group -- a rank starts as fast as a single process.  The ranks of a node meet in a directory of the sample
  (<outdir>/snps/temp): small control messages (does every rank still stand, the numbers of the rank-local decode plan) go
  through files there; once a rank has its device context (`attach_context`) the ranks form an RCCL communicator through the
  library's own binding (midas_comm_*, comm.cpp: ncclCommInitRank with the id rank 0 wrote) and the summary rows -- and the
  genes path's all-to-all -- travel over xGMI.  Ranks that share ONE device (tests on a one-GPU box; RCCL refuses that) stay on
  the files for those too.
* TORCH (`init_from_env("gloo")`, or a process group the caller already initialised -- bench.py, the CPU tests): the same
  collectives through torch.distributed.
"""

import datetime
import os
import sys
import time

import numpy as np

_STAT_COLS = 5   # genome_length, covered_bases, total_depth, aligned_reads, mapped_reads
_native = None   # the native transport once init_from_env started it


def _process_start(pid):
    """The kernel's start time of process `pid` (clock ticks since boot, /proc/<pid>/stat field 22), None when it is gone:
    with the pid it names ONE process of this node for good, however often pids are reused."""
    try:
        with open("/proc/%d/stat" % pid, "rb") as f:
            fields = f.read().rsplit(b")", 1)[1].split()
        if fields[0] in (b"Z", b"X"):          # (dead, only not reaped by its parent yet)
            return None
        return int(fields[19])
    except (OSError, ValueError, IndexError):
        return None


def _meeting_name():
    """The directory the ranks of ONE launch meet in: what every rank of the launch is told alike and no other launch on the
    node shares -- the rendezvous address torchrun hands out, and the run id when one was given (MIDAS_RUN_ID, or torchrun's
    --rdzv-id).  Nothing a rank has by itself (its parent, its pid): ranks started through per-rank wrapper shells meet too."""
    run = os.environ.get("MIDAS_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    key = "%s_%s" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"))
    if run and run != "none":
        key += "." + run
    return "ranks." + "".join(c if c.isalnum() or c in "._-" else "_" for c in key)


class _Native:
    """The ranks of one node, met in a directory.

    Meeting (once, at most MEET_TIMEOUT): the directory's name is the same for every launch on this address and port, and a
    run that crashed leaves its files in it -- so nothing found there is believed until it is tied to a LIVING process of this
    launch.  Every rank writes `hello.<rank>.<pid>` = its process start time and a random token.  Rank 0 waits for a hello of
    every rank whose process lives (pid + start time, /proc), clears what older generations left, makes a fresh subdirectory
    `gen.<random>` and publishes `current` = that name and the (pid, token) of every rank it met.  A rank trusts `current` only
    when it lists its own pid and token -- a stale one names dead processes -- and from then on every file of the exchange lives
    in the generation's subdirectory, which no other launch has ever written to.

    all_gather_bytes: every rank writes `<seq>.<rank>` (under a temporary name, then renamed) and reads the others'; a rank
    removes its file of two exchanges ago when it starts a new one -- by then every rank has read it (a rank that starts
    exchange k has finished k - 1, for which all had written k - 1, i.e. all had finished reading k - 2)."""

    def __init__(self, rank, ws, root):
        self.rank, self.ws = rank, ws
        self.root = os.path.join(root, _meeting_name())
        os.makedirs(self.root, exist_ok=True)
        self.seq = 0
        self.comm = None
        self.comm_state = "no device context yet"
        self.deadline = COLLECTIVE_TIMEOUT.total_seconds()
        self.peers = {}          # rank -> (pid, process start time) of the launch's ranks, as met
        self.dir = self._meet()

    def _meet(self):
        import glob
        import secrets
        pid, token, t0 = os.getpid(), secrets.token_hex(8), time.monotonic()
        limit = float(os.environ.get("MIDAS_MEET_TIMEOUT", MEET_TIMEOUT))
        for old in glob.glob(os.path.join(glob.escape(self.root), "hello.%d.*" % self.rank)):       # (this rank's place in older launches)
            try:
                os.remove(old)
            except OSError:
                pass
        mine = os.path.join(self.root, "hello.%d.%d" % (self.rank, pid))
        with open(mine + ".tmp", "w") as f:
            f.write("%d %s" % (_process_start(pid) or 0, token))
        os.replace(mine + ".tmp", mine)

        def late(what):
            sys.exit("\nError: rank %d of %d waited %.0f s %s in %s\n(the ranks of a launch meet there: they must run on ONE node, see the "
                     "same directory, and be given the same MASTER_ADDR / MASTER_PORT -- and MIDAS_RUN_ID, if set; MIDAS_MEET_TIMEOUT "
                     "changes the wait)\n" % (self.rank, self.ws, limit, what, self.root))
        current = os.path.join(self.root, "current")
        if self.rank == 0:
            met, nap = {}, 0.0005
            while len(met) < self.ws:
                for path in glob.glob(os.path.join(glob.escape(self.root), "hello.*.*")):
                    parts = os.path.basename(path).split(".")
                    try:
                        r, p = int(parts[1]), int(parts[2])
                        with open(path) as f:
                            started, tok = f.read().split()
                    except (OSError, ValueError, IndexError):
                        continue          # (half-written or just removed: looked at again)
                    if 0 <= r < self.ws and _process_start(p) == int(started) and int(started) != 0:
                        met[r] = (p, tok)
                if len(met) < self.ws:
                    if time.monotonic() - t0 > limit:
                        late("for rank(s) %s" % sorted(set(range(self.ws)) - set(met)))
                    time.sleep(nap)
                    nap = min(nap * 1.5, 0.02)
            import shutil
            for old in glob.glob(os.path.join(glob.escape(self.root), "gen.*")):
                shutil.rmtree(old, ignore_errors=True)
            gen = "gen." + secrets.token_hex(8)
            os.makedirs(os.path.join(self.root, gen))
            with open(current + ".tmp", "w") as f:
                f.write(gen + "\n" + "".join("%d %d %s\n" % (r, met[r][0], met[r][1]) for r in range(self.ws)))
            os.replace(current + ".tmp", current)
            self.peers = {r: (met[r][0], _process_start(met[r][0])) for r in range(self.ws)}
            return os.path.join(self.root, gen)
        nap = 0.0005
        while True:
            try:
                with open(current) as f:
                    lines = f.read().split("\n")
                if ("%d %d %s" % (self.rank, pid, token)) in lines[1:]:
                    for ln in lines[1:]:
                        if ln:
                            r, p = int(ln.split()[0]), int(ln.split()[1])
                            self.peers[r] = (p, _process_start(p))
                    return os.path.join(self.root, lines[0])
            except OSError:
                pass
            if time.monotonic() - t0 > limit:
                late("for rank 0's list of the ranks it met")
            time.sleep(nap)
            nap = min(nap * 1.5, 0.02)

    def _path(self, seq, r):
        return os.path.join(self.dir, "%d.%d" % (seq, r))

    def all_gather_bytes(self, data):
        seq = self.seq
        self.seq += 1
        if seq >= 2:
            try:
                os.remove(self._path(seq - 2, self.rank))
            except OSError:
                pass
        mine = self._path(seq, self.rank)
        with open(mine + ".tmp", "wb") as f:
            f.write(data)
        os.replace(mine + ".tmp", mine)
        out, t0, nap = [], time.monotonic(), 0.0002
        for r in range(self.ws):
            if r == self.rank:
                out.append(bytes(data))
                continue
            path = self._path(seq, r)
            looked = time.monotonic()
            while True:
                try:
                    with open(path, "rb") as f:
                        out.append(f.read())
                    break
                except FileNotFoundError:
                    now = time.monotonic()
                    if now - t0 > self.deadline:
                        sys.exit("\nError: rank %d waited %d s for rank %d at exchange %d (%s)\n" % (self.rank, self.deadline, r, seq, self.dir))
                    # a rank may wait for hours (rank 0 aligns) -- but not for a process that is gone: a rank that died without
                    # a word (killed, an uncaught exception) is noticed within a second, not at the collective's deadline
                    if now - looked > 0.5 and r in self.peers:
                        looked = now
                        p, started = self.peers[r]
                        if _process_start(p) != started and not os.path.exists(path):
                            sys.exit("\nError: rank %d (process %d) is gone without a message; rank %d, waiting for it at exchange %d, stops too\n"
                                     % (r, p, self.rank, seq))
                    time.sleep(nap)
                    nap = min(nap * 1.5, 0.005)
        return out

    def attach(self, ctx):
        """Form the RCCL communicator of the ranks' devices (once): rank 0 makes the id, a file carries it."""
        if self.comm is not None or not hasattr(ctx, '_h'):
            return
        from . import abi
        key = abi.Comm.device_key(ctx)
        keys = [k.decode() for k in self.all_gather_bytes(key.encode())]
        if len(set(keys)) < self.ws:
            self.comm_state = "ranks share a device (%s): RCCL refuses that, the exchange stays on files" % ", ".join(sorted(set(keys)))
            return
        # Every rank says whether it can use RCCL at all (the library loads and answers) BEFORE any of them enters
        # ncclCommInitRank, which waits for all ranks of the communicator: one that could not would hold the others there.
        ident, error = b"\0" * 128, b""
        try:
            abi.Comm.probe()
            if self.rank == 0:
                ident = abi.Comm.unique_id()
        except abi.MidasSnpsError as e:
            error = e.message.encode() or b"RCCL failed"
        except Exception as e:
            error = ("%s: %s" % (type(e).__name__, e)).encode()
        got = self.all_gather_bytes(ident + error)
        bad = ["rank %d: %s" % (r, g[128:].decode(errors="replace")) for r, g in enumerate(got) if g[128:]]
        if bad:
            self.comm_state = "RCCL is not available (%s): the exchange stays on files" % "; ".join(bad)
            return
        try:
            self.comm = abi.Comm(ctx, got[0][:128], self.rank, self.ws)
            state = b"ok"
        except abi.MidasSnpsError as e:
            state = e.message.encode() or b"midas_comm_create failed"
        except Exception as e:        # (whatever it was: the other ranks are told, nobody is left waiting for this one's "ok")
            state = ("%s: %s" % (type(e).__name__, e)).encode()
        states = self.all_gather_bytes(state)
        if any(x != b"ok" for x in states):
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            self.comm_state = "ncclCommInitRank failed (%s): the exchange stays on files" % b"; ".join(x for x in states if x != b"ok").decode()
            return
        self.comm_state = "RCCL communicator of %d ranks (librccl, ncclCommInitRank)" % self.ws

    def data_all_gather(self, data):
        """equal-sized payloads: over RCCL when the communicator stands"""
        if self.comm is not None:
            return self.comm.all_gather(data)
        return self.all_gather_bytes(data)

    def detach(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
            self.comm_state = "no device context any more"

    def close(self):
        """Leave: every other rank says so when it has read its last file; rank 0 waits for all of them -- it may remove the
        sample's temp directory right afterwards."""
        self.detach()
        bye = lambda r: os.path.join(self.dir, "bye.%d" % r)
        if self.rank != 0:
            open(bye(self.rank), "wb").close()
        else:
            t0 = time.monotonic()
            for r in range(1, self.ws):
                while not os.path.exists(bye(r)):
                    if time.monotonic() - t0 > 600:
                        break
                    time.sleep(0.001)
        # (a rank's last files stay until rank 0 clears the place: somebody may still be reading them)
        try:
            os.remove(os.path.join(self.root, "hello.%d.%d" % (self.rank, os.getpid())))
        except OSError:
            pass
        if self.rank == 0:          # (every other rank has left: the generation's files go, and the meeting place when it is empty)
            import shutil
            shutil.rmtree(self.dir, ignore_errors=True)
            try:
                os.remove(os.path.join(self.root, "current"))
                os.rmdir(self.root)
            except OSError:
                pass


def _alone():
    """One process and no process group can exist: torch is not even imported (a second and a half of the single-GPU command)."""
    return int(os.environ.get("WORLD_SIZE", "1") or 1) <= 1 and 'torch.distributed' not in sys.modules


def _torch_group():
    """torch.distributed when the caller initialised a group (bench.py, the gloo tests), else None -- without importing torch."""
    if 'torch.distributed' not in sys.modules:
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
    if _native is not None:
        return _native.rank, _native.ws
    if _alone():
        return 0, 1
    dist = _torch_group()
    if dist is not None:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# Rank 0 builds the bowtie2 database and aligns (tens of minutes to hours) while the other ranks wait at a barrier: the
# default collective timeout (10 min with nccl) would let the watchdog kill the job before the pileup starts.
COLLECTIVE_TIMEOUT = datetime.timedelta(hours=48)
# ... but the ranks MEET within minutes of their start or not at all (a launcher that gave them different directories or
# addresses, a rank that died at import): that wait is short and its message says where they were expected.
MEET_TIMEOUT = 120.0


def init_from_env(device_backend=None, rendezvous_dir=None):
    """Join the ranks torchrun described (RANK / WORLD_SIZE / MASTER_*); no-op for a single process.
    rendezvous_dir (and no backend named, no group initialised by the caller): the native transport, no torch.
    device_backend "gloo" / "nccl": a torch.distributed process group, as the tests and bench.py use."""
    global _native
    if _native is not None:
        return world()
    if _alone():
        return 0, 1
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if _torch_group() is not None:
        return world()
    if ws <= 1:
        return 0, 1
    # (the native transport's ranks meet in a directory and tell their devices apart by PCI bus id: ONE node.  A launch over
    # several nodes -- LOCAL_WORLD_SIZE below WORLD_SIZE -- takes the torch process group below, as before round 5.)
    one_node = int(os.environ.get("LOCAL_WORLD_SIZE", ws) or ws) == ws
    if device_backend is None and rendezvous_dir is not None and one_node and os.environ.get("MIDAS_DIST_BACKEND", "native") == "native":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _native = _Native(int(os.environ.get("RANK", "0")), ws, rendezvous_dir)
        return world()
    import torch
    import torch.distributed as dist
    backend = device_backend or os.environ.get("MIDAS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "native":
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=COLLECTIVE_TIMEOUT)
    else:
        dist.init_process_group(backend, timeout=COLLECTIVE_TIMEOUT)
    return world()


def attach_context(ctx):
    """The rank has its device context: the native transport forms its RCCL communicator on it (collective: every rank calls
    this at the same point).  Returns a line for the log, or None when there is nothing to say."""
    if _native is None:
        return None
    _native.attach(ctx)
    return "ranks: %s" % _native.comm_state


def detach_context():
    """The device context is about to go: the communicator goes first (the files carry what little follows)."""
    if _native is not None:
        _native.detach()


def finalize():
    global _native
    if _native is not None:
        _native.close()
        _native = None


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
    if _native is not None:
        flags = _native.all_gather_bytes(b"1" if error_message is not None else b"0")
        failed = [r for r, f in enumerate(flags) if f != b"0"]
        if not failed:
            return
        if error_message is not None:
            sys.exit(error_message)
        sys.exit("\nError: rank(s) %s failed, see their message; rank %d stops with them\n" % (failed, rank))
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
    if _native is not None:       # ncclAllGather over xGMI (midas_comm_all_gather) once the communicator stands
        got = _native.data_all_gather(rows.tobytes())
        return np.sum([np.frombuffer(g, np.int64).reshape(rows.shape) for g in got], axis=0)
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
    if _native is not None:
        return np.stack([np.frombuffer(g, np.int64) for g in _native.all_gather_bytes(v.tobytes())])
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(v)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = torch.empty(ws * t.shape[0], dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.reshape(ws, -1).cpu().numpy()


def all_gather_blob(data):
    """data: bytes of any length on every rank -> the ranks' blobs, by rank (the genomes' index: midas_amd/run/snps.py)."""
    data = bytes(data)
    rank, ws = world()
    if ws == 1:
        return [data]
    if _native is not None:
        return _native.all_gather_bytes(data)
    sizes = all_gather_i64([len(data)])[:, 0]
    width = (int(sizes.max()) + 7) // 8
    if width == 0:
        return [b""] * ws
    padded = np.zeros(width * 8, np.uint8)
    padded[:len(data)] = np.frombuffer(data, np.uint8)
    got = all_gather_i64(padded.view(np.int64))
    return [got[r].view(np.uint8)[:int(sizes[r])].tobytes() for r in range(ws)]


def all_gather_rows_f64(rows):
    """Like all_gather_summary for float64 rows (the genes summary has means and medians): rows are zero outside the
    species this rank owns, so the sum over ranks is the owner's row (nan stays nan)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    rank, ws = world()
    if ws == 1:
        return rows.copy()
    if _native is not None:
        got = _native.data_all_gather(rows.tobytes())
        return np.sum([np.frombuffer(g, np.float64).reshape(rows.shape) for g in got], axis=0)
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
    if _native is not None:
        dtype = parts[0].dtype
        counts = all_gather_i64(np.array([p.size for p in parts], np.int64))        # counts[src, dst]
        if _native.comm is not None:       # grouped ncclSend / ncclRecv (midas_comm_all_to_all_v)
            got = _native.comm.all_to_all_v([p.tobytes() for p in parts], [int(n) * dtype.itemsize for n in counts[:, rank]])
        else:                               # (ranks that share a device: every rank's parts through the files, each takes its own)
            sizes = np.array([p.nbytes for p in parts], np.int64)
            blobs = _native.all_gather_bytes(sizes.tobytes() + b"".join(p.tobytes() for p in parts))
            got = []
            for src in range(ws):
                sz = np.frombuffer(blobs[src][:8 * ws], np.int64)
                at = 8 * ws + int(sz[:rank].sum())
                got.append(blobs[src][at:at + int(sz[rank])])
        return [np.frombuffer(g, dtype).copy() for g in got]
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
    if _native is not None:
        _native.all_gather_bytes(b"")
        return
    if _alone():
        return
    dist = _torch_group()
    if dist is not None:
        dist.barrier()
