"""MI355X-native `run_midas.py snps` pipeline: the host side.

Mirrors /root/reference/midas/run/snps.py name for name (Species, Contig, initialize_species,
initialize_contigs, build_genome_db, genome_align, index_bam, species_pileup, pysam_pileup,
snps_summary, remove_tmp, run_pipeline) so that scripts/run_midas.py can call it exactly as the
reference's CLI calls the original, and so that the tests read like the reference's own.

What is different underneath:
  * pysam_pileup / species_pileup do not walk the BAM with pysam callbacks; they decode the BAM once
    (native reader), hand the records to the HIP library through the C-ABI (include/midas_snps.h) and
    let the native formatter write <species>.snps.gz.  There is NO CPU fallback: without the library
    or a gfx950 GPU the stage exits with an error.
  * index_bam does not run `samtools index`: the device builds its own per-tile read index.
  * the pileup is one task per *rank* (torch.distributed, one process per GPU), contigs sharded over
    ranks, not one mp.Pool task per species (which is also what breaks the reference on python3:
    args['log'] is not picklable).
"""

import os
import shutil
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor
from contextlib import ExitStack
from collections.abc import Mapping
from time import time

import numpy as np

from midas_amd import abi, bam, dist, fasta, pieces, utility


# zlib level of <species>.snps.gz.  The reference writes level 9 (gzip.open's default, midas/utility.py:194-206); the
# decompressed text is what downstream reads and it is identical at any level.  Measured on the rows of configs[1] per
# writer thread: level 9 0.2, 6 0.5, 4 1.7, 1 2.5 M rows/s for 28.5 / 29.0 / 30.4 / 34.5 MB -- 4 costs 5 % of file size and
# takes the formatter from the largest to the second smallest item of the stage.
GZ_LEVEL = 4
WRITERS = int(os.environ.get('MIDAS_SNPS_WRITERS', '6') or 6)      # tables written side by side from a batch (_write_jobs)
MAX_BATCH_READS = 1 << 30   # a rank's work items go to the device in batches of at most this many reads (args['max_batch_reads']) ...
MAX_BATCH_PAYLOAD = 24 << 30   # ... and about this many bytes of SEQ / QUAL / CIGAR (the library's own limits: 2 * 10^9 reads, 32 GiB)
SPLIT_LENGTH = 8 << 20      # contigs longer than this are dealt to the ranks in pieces (args['split_length']; 0: never)


class Species:
    """A species of the sample: id, where its representative genome lives, and (after the pileup) its counters."""
    __slots__ = ('id', 'paths', 'aligned_reads', 'mapped_reads', 'genome_length', 'covered_bases', 'total_depth',
                 'fraction_covered', 'mean_coverage')

    def __init__(self, id):
        self.id = id
        self.paths = {}
        self.aligned_reads = self.mapped_reads = self.genome_length = self.covered_bases = self.total_depth = 0
        self.fraction_covered = self.mean_coverage = 0

    def fetch_paths(self, ref_db):
        """genome.fna / genome.features of the species, plain or .gz (the .gz wins when both exist, as in the reference)."""
        base = os.path.join(ref_db, 'rep_genomes', self.id)
        for kind in ('fna', 'features'):
            for suffix in ('', '.gz'):
                candidate = os.path.join(base, 'genome.%s%s' % (kind, suffix))
                if os.path.isfile(candidate):
                    self.paths[kind] = candidate


class Contig:
    """One FASTA record of a representative genome: id, upper-cased sequence, length, owning species.  The sequence is kept
    as the bytes the device takes (`seq_bytes`); `seq` -- the str the reference's Contig carries -- is made when asked for."""
    __slots__ = ('id', 'seq_bytes', '_seq', 'length', 'species_id', 'pool', 'pool_at')

    def __init__(self, id, seq='', species_id=None, pool=None, pool_at=0):
        self.id = id
        self._seq = seq if isinstance(seq, str) else None
        self.seq_bytes = seq.encode('latin-1') if isinstance(seq, str) else seq      # (bytes-like: bytes or a uint8 array; None: known
        self.length = len(seq) if seq is not None else 0                             #  by id, species and length only -- DealtContigs)
        self.species_id = species_id
        self.pool, self.pool_at = pool, pool_at     # initialize_contigs: all sequences back to back in one array

    @property
    def seq(self):
        if self._seq is None:
            self._seq = bytes(self.seq_bytes).decode('latin-1')
        return self._seq


def _reference_bytes(parts):
    """The sequences parts = [(Contig, lo, hi)] back to back as one uint8 array: a view of the pool initialize_contigs read
    them into when they lie there in this order (every contig of every genome, in the order of the BAM header -- the usual
    case), else a copy."""
    pool = parts[0][0].pool if parts else None
    if pool is not None:
        at = parts[0][0].pool_at + parts[0][1]
        begin = at
        for c, lo, hi in parts:
            if c.pool is not pool or c.pool_at + lo != at:
                break
            at = c.pool_at + hi
        else:
            return pool[begin:at]
    return np.frombuffer(b''.join(c.seq_bytes[lo:hi] for c, lo, hi in parts), dtype=np.uint8)


def select_species(args, per_species='rep_genomes'):
    """Only --species_id can be honoured here: --species_cov / --species_topn read the abundance profile written by
    `run_midas.py species` (midas/run/species.py:191-227), a pipeline outside this build."""
    wanted = args.get('species_id')
    if not wanted:
        sys.exit("\nError: this build only selects species with --species_id "
                 "(--species_cov/--species_topn need `run_midas.py species`, which is out of scope)\n")
    for sp in wanted:
        if not os.path.isdir(os.path.join(args['db'], per_species, sp)):
            sys.exit("\nError: Species id not found in database: %s\n" % sp)
    return list(wanted)


def initialize_species(args):
    """{species_id: Species}: chosen now and recorded in snps/species.txt when the database is being built, else
    read back from that file (midas/run/snps.py:38-53)."""
    listing = os.path.join(args['outdir'], 'snps', 'species.txt')
    if args['build_db']:
        ids = select_species(args)
        with open(listing, 'w') as handle:
            handle.writelines(i + '\n' for i in ids)
    elif os.path.isfile(listing):
        with open(listing) as handle:
            ids = [line.rstrip() for line in handle]
    else:
        ids = []
    species = {i: Species(i) for i in ids}
    for sp in species.values():
        sp.fetch_paths(ref_db=args['db'])
    return species


def _records(sp):
    if 'fna' not in sp.paths:
        sys.exit("\nError: Could not locate the representative genome of species: %s\n" % sp.id)
    with utility.iopen(sp.paths['fna'], 'rb') as handle:
        for rec_id, rec_seq in fasta.parse_bytes(handle.read()):
            yield rec_id, rec_seq.upper()


FASTA_THREADS = 2   # the genomes are read BESIDE the BAM decode, whose copy threads need the cores (all cores: configs[3] 2.8 s instead of 2.2 s)


def initialize_contigs(species, threads=None):
    """{contig_id: Contig} over every species' representative genome, sequences upper-cased (midas/run/snps.py:55-67).  The
    files are read by the library's FASTA reader, one file a task on all cores (midas_fasta_load: the records of
    midas_amd/fasta.py, which the tests hold it to -- 400 Mb of genomes take the interpreter's own reader 0.8 s on its one core,
    beside a BAM decode that is done sooner)."""
    sps = list(species.values())
    if not sps:
        return {}
    for sp in sps:
        if 'fna' not in sp.paths:
            sys.exit("\nError: Could not locate the representative genome of species: %s\n" % sp.id)
    try:
        arr, recs = abi.read_fasta_files([sp.paths['fna'] for sp in sps], threads=FASTA_THREADS if threads is None else threads)
    except abi.MidasSnpsError as e:
        sys.exit("\nError: %s\n" % e.message)
    return {rid: Contig(rid, arr[at:at + n], sps[fi].id, arr, at) for rid, fi, at, n in recs}


class DealtContigs(Mapping):
    """N ranks: every rank would read every species' genome -- the same 400 Mb parsed eight times over, on two CPUs each, and at
    eight ranks the longest thing a rank does.  Instead the genome FILES are dealt to the ranks (contiguous runs of the species
    list, by file size: the order the alignment's contigs -- and with them the ranks' shares of the BAM -- follow); a rank reads
    its run beside its BAM decode; `exchange()` all-gathers what every rank learned -- (contig id, species, length), a few tens
    of bytes a contig -- so that every rank knows EVERY contig (the emit order, the weights and the owners need no more); the
    sequences of contigs this rank piles up but another rank read are parsed when `need()` asks (the runs and the shares rarely
    differ by more than a species at each end).  Looks like initialize_contigs' dict; a contig whose sequence is not here has
    `seq_bytes` None until needed."""

    def __init__(self, species, loaded):
        self._species, self._all = species, dict(loaded)
        self._have = {c.species_id for c in loaded.values()}         # species whose files this rank has parsed
        self._mine = list(loaded)

    def exchange(self):
        """Collective: every rank's (id, species, length) of the contigs it read -> the index of all contigs, on every rank."""
        order = {sp: k for k, sp in enumerate(self._species)}
        ids = list(self._species)
        mine = "\n".join("%s\t%d\t%d" % (cid, order[self._all[cid].species_id], self._all[cid].length) for cid in self._mine)
        for r, blob in enumerate(dist.all_gather_blob(mine.encode('latin-1'))):
            for line in blob.decode('latin-1').split("\n") if blob else ():
                cid, k, n = line.rsplit("\t", 2)
                if cid not in self._all:
                    c = Contig(cid, None, ids[int(k)])
                    c.length = int(n)
                    self._all[cid] = c
        return self

    def need(self, contig_ids):
        """The sequences of these contigs are about to be piled up: the files of those another rank read are parsed now."""
        missing = sorted({self._all[cid].species_id for cid in contig_ids if self._all[cid].seq_bytes is None} - self._have)
        if missing:
            self._all.update(initialize_contigs({sp: self._species[sp] for sp in missing}))
            self._have.update(missing)
        return len(missing)

    def __getitem__(self, key):
        return self._all[key]

    def __iter__(self):
        return iter(self._all)

    def __len__(self):
        return len(self._all)


def deal_species(species, rank, ws):
    """The species whose genome files rank `rank` of `ws` reads: a contiguous run of the species list, the runs even in bytes."""
    ids = list(species)
    size = []
    for sp in ids:
        try:
            size.append(max(1, os.path.getsize(species[sp].paths.get('fna', ''))))
        except OSError:
            size.append(1)          # (a missing genome: whoever is dealt it says so, midas/run/snps.py:57)
    total, acc, mine = float(sum(size)) or 1.0, 0.0, []
    for sp, s in zip(ids, size):
        if min(ws - 1, int(ws * (acc + s / 2.0) / total)) == rank:
            mine.append(sp)
        acc += s
    return mine


class ContigsInBackground(Mapping):
    """initialize_contigs on a thread of its own: the mapping is there at once and waits for the reader the first time it is
    looked into.  The pileup's first act is the BAM decode -- native code that does not hold the interpreter -- so the
    genomes are read while the alignments are inflated instead of in front of them.  What the reader raises (a missing
    genome ends the run, midas/run/snps.py:57) is raised where the mapping is first used.
    deal = (rank, ranks): this rank reads its run of the genome files only; wait() then hands out a DealtContigs."""

    def __init__(self, species, start=True, deal=None):
        self._out, self._err, self._thread, self._species = None, None, None, species
        self._deal = deal if deal and deal[1] > 1 else None
        if start:
            self.start()

    def start(self):
        """Begin reading (once).  run_pipeline starts it behind the build / align stages when those run: they read the same
        FASTA files on rank 0 and hold the machine for the length of an alignment."""
        if self._thread is not None or self._out is not None or self._err is not None:
            return

        def work():
            try:
                if self._deal is not None:
                    run = deal_species(self._species, *self._deal)
                    self._out = DealtContigs(self._species, initialize_contigs({sp: self._species[sp] for sp in run}))
                else:
                    self._out = initialize_contigs(self._species)
            except BaseException as e:       # (SystemExit included: it must end the main thread's run, not this thread)
                self._err = e
        self._thread = threading.Thread(target=work, name="read-genomes", daemon=True)
        self._thread.start()

    def wait(self):
        self.start()
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._err is not None:
            raise self._err
        return self._out

    def __getitem__(self, key):
        return self.wait()[key]

    def __iter__(self):
        return iter(self.wait())

    def __len__(self):
        return len(self.wait())


def _shell(args, stages):
    """Run `stage | stage | ...` through the shell, logging the command; a failing pipeline ends the run."""
    command = ' | '.join(' '.join(str(x) for x in stage) for stage in stages) + ' '
    args['log'].write('command: ' + command + '\n')
    process = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    utility.check_exit_code(process, command)


def build_genome_db(args, species):
    """snps/temp/genomes.fa = every selected genome, then `bowtie2-build` on it (midas/run/snps.py:69-95)."""
    temp = os.path.join(args['outdir'], 'snps', 'temp')
    n_seqs = n_bases = 0
    with open(os.path.join(temp, 'genomes.fa'), 'wb') as out:
        for sp in species.values():
            for rid, seq in _records(sp):
                out.write(b'>' + rid.encode('latin-1') + b'\n' + seq + b'\n')
                n_seqs += 1
                n_bases += len(seq)
    print("  total genomes: %s\n  total contigs: %s\n  total base-pairs: %s" % (len(species), n_seqs, n_bases))
    if not args.get('bowtie2-build'):
        sys.exit("\nError: bowtie2-build not found on PATH (needed for --build_db; the aligner is not part of this build)\n")
    _shell(args, [[args['bowtie2-build'], '--threads', args['threads'], os.path.join(temp, 'genomes.fa'),
                   os.path.join(temp, 'genomes')]])


def genome_align(args):
    """bowtie2 (no unaligned reads) | samtools view -b | samtools sort -> snps/temp/genomes.bam, with the same
    switches the reference passes (midas/run/snps.py:97-128)."""
    if not args.get('bowtie2') or not args.get('samtools'):
        sys.exit("\nError: bowtie2 / samtools not found on PATH (needed for --align; the aligner is not part of this build)\n")
    temp = os.path.join(args['outdir'], 'snps', 'temp')
    bt2 = [args['bowtie2'], '--no-unal', '-x', os.path.join(temp, 'genomes')]
    if args['max_reads']:
        bt2 += ['-u', args['max_reads']]
    if args['trim']:
        bt2 += ['--trim3', args['trim']]
    bt2 += ['--%s%s' % (args['speed'], '-local' if args['mode'] == 'local' else ''), '--threads', args['threads'],
            '-f' if args['file_type'] == 'fasta' else '-q']
    if args['m2']:
        bt2 += ['-1', args['m1'], '-2', args['m2']]
    elif args['interleaved']:
        bt2 += ['--interleaved', args['m1']]
    else:
        bt2 += ['-U', args['m1']]
    view = [args['samtools'], 'view', '-b', '-', '--threads', args['threads']]
    sort = [args['samtools'], 'sort', '-', '--threads', args['threads'], '-o', os.path.join(temp, 'genomes.bam')]
    _shell(args, [bt2, view, sort])
    print("  finished aligning")


def index_bam(args):
    """The reference runs `samtools index` here because pysam fetches through the .bai (midas/run/snps.py:130-139).
    The device builds its own per-tile read index at the start of every pileup pass (index_reads_kernel), so the
    host has nothing to do; the step is kept so that logs and callers look the same."""
    start = time()
    print("\nIndexing bamfile")
    args['log'].write("\nIndexing bamfile\ncommand: (none) per-tile read index is built on the GPU by index_reads_kernel\n")
    print("  %s minutes" % round((time() - start) / 60, 2))
    print("  %s Gb maximum memory" % utility.max_mem_usage())


_ERR_TEXT = {
    abi.ERR_READ_NO_SEQ: "an alignment has no SEQ (the reference raises TypeError in keep_read)",
    abi.ERR_READ_NO_NM: "an alignment has no NM tag (the reference raises KeyError: 'NM' in keep_read)",
    abi.ERR_READ_ZERO_ALIGN: "an alignment has aligned length 0 (the reference raises ZeroDivisionError in keep_read)",
    abi.ERR_READ_NO_QUAL: "an alignment has no QUAL (the reference raises TypeError in np.mean)",
    abi.ERR_READ_CIGAR_OVERRUN: "an alignment's CIGAR is longer than its SEQ (the reference raises IndexError)",
}


def _error_text(e):
    """The reference's convention for a failed stage: sys.exit("\\nError: ...") (midas/utility.py:227-232)."""
    msg = _ERR_TEXT.get(e.status, e.message)
    where = " [read %d of the rank's records]" % e.read_index if e.read_index >= 0 else ""
    return "\nError: %s%s\n%s\n" % (msg, where, e.message)


def _exit_on(e):
    if isinstance(e, abi.MidasSnpsError):
        sys.exit(_error_text(e))
    raise e


def _contig_table(species_ids, items, span, contigs, ref_names, ref_lens, refid, reads, halo=None, fetch=None):
    """(ContigTable, ReadsSoA, keys) for the given work items -- (contig id, piece number), span[item] = (lo, hi, last):
    contigs in BAM header order (pieces of one contig in position order), reads regrouped to match; keys[k] = the item of
    table entry k.  Whole contigs only: the table of the reference's loop (midas/run/snps.py:187-199); with pieces the
    entries carry their origin and the reads in front of a piece that reach into it (midas_amd/pieces.py)."""
    sp_index = {s: i for i, s in enumerate(species_ids)}
    order = {n: i for i, n in enumerate(ref_names)}
    mine = [contigs[cid] for cid in sorted({cid for cid, _ in items})]
    missing = [c.id for c in mine if c.id not in order]
    # a contig that is not in the BAM header: pysam would raise on count_coverage(contig.id, ...)
    if missing:
        sys.exit("\nError: contig '%s' is not in the BAM header (was the genome database rebuilt after alignment?)\n" % missing[0])
    mine = sorted(mine, key=lambda c: order[c.id])
    for c in mine:
        if ref_lens[order[c.id]] != c.length:
            sys.exit("\nError: contig '%s' has length %d in the BAM header but %d in the FASTA\n"
                     % (c.id, ref_lens[order[c.id]], c.length))
    ids = [c.id for c in mine]
    sub, read_begin = bam.group_by_contig(ref_names, refid, reads, ids, fetch=fetch)
    if all(span[it][0] == 0 and span[it][2] for it in items):
        ref = _reference_bytes([(c, 0, c.length) for c in mine])
        table = abi.ContigTable(length=[c.length for c in mine], species=[sp_index[c.species_id] for c in mine],
                                read_begin=read_begin, ref=ref, n_species=len(species_ids), ids=ids,
                                species_ids=list(species_ids))
        return table, sub, [(c.id, 0) for c in mine]
    at = {c.id: k for k, c in enumerate(mine)}
    keys = sorted(items, key=lambda it: (at[it[0]], it[1]))
    plan = [(at[cid], span[(cid, j)][0], span[(cid, j)][1], span[(cid, j)][2], int(halo.get(cid, 0)) if halo else 0) for cid, j in keys]
    if getattr(sub, 'device', None) is not None and fetch is not None:
        sub = fetch(sub)          # (cutting pieces slices SEQ / QUAL / CIGAR: in host memory)
    sub, read_begin = pieces.gather(sub, read_begin, plan)
    ref = _reference_bytes([(mine[k], lo, hi) for k, lo, hi, _, _ in plan])
    table = abi.ContigTable(length=[hi - lo for _, lo, hi, _, _ in plan], species=[sp_index[mine[k].species_id] for k, *_ in plan],
                            read_begin=read_begin, ref=ref, n_species=len(species_ids), ids=[cid for cid, _ in keys],
                            species_ids=list(species_ids), origin=[lo for _, lo, _, _, _ in plan])
    return table, sub, keys


def _species_contig_order(species_ids, contigs):
    """{species_id: its contig ids in the order the reference emits them}: sorted(contigs.keys()) filtered by species
    (midas/run/snps.py:187-192)."""
    out = {sp: [] for sp in species_ids}
    for cid in sorted(contigs):
        sp = contigs[cid].species_id
        if sp in out:
            out[sp].append(cid)
    return out


def _part_path(args, species_id, k):
    return '%s/snps/output/%s.snps.gz.part%06d' % (args['outdir'], species_id, k)


def _write_rows(args, path, table, pos, items, counts, allele, off, header, batch=None):
    """The rows of the work items `items` (table entries pos[item]), in that order."""
    ks = [pos[it] for it in items]
    cids = [it[0] for it in items]
    level, threads = int(args.get('gz_level', GZ_LEVEL)), int(args.get('threads', 1) or 1)
    if batch is not None:       # (the batch numbers a piece's rows from its origin by itself)
        batch.write_part(path, ks, cids, header=header is None or bool(header), gz_level=level, threads=threads)
        return
    first = [int(table.origin[k]) for k in ks] if table.origin is not None else None
    abi.write_table(path, cids, [allele[off[k]:off[k + 1]] for k in ks], [counts[off[k]:off[k + 1]] for k in ks],
                    gz_level=level, threads=threads, header=header, first_pos=first)


def _write_species(args, species_id, table, counts, allele):
    """<outdir>/snps/output/<species>.snps.gz of ONE species whose contigs are all in `table` -- header + rows in
    sorted(contig id) order (midas/run/snps.py:179-182, 187-192, 201-210), formatted and gzipped by the native writer."""
    sp = table.species_ids.index(species_id)
    pos = {(cid, 0): k for k, cid in enumerate(table.ids)}
    cids = sorted(it for it, k in pos.items() if table.species[k] == sp)   # (a species without contigs: header only)
    _write_rows(args, '%s/snps/output/%s.snps.gz' % (args['outdir'], species_id), table, pos, cids, counts, allele,
                table.site_offsets(), None)


def _whole(order, contigs):
    """Every contig one work item: ({species: [(contig id, 0)]}, {item: (0, length, True)})."""
    items = {sp: [(cid, 0) for cid in cids] for sp, cids in order.items()}
    return items, {(cid, 0): (0, contigs[cid].length, True) for cids in order.values() for cid in cids}


PHASES = None       # a list: every lap of the stage is appended as (name, seconds) -- bench.py's stage_e2e block sets it
_T_END = [0.0]      # when _count_alleles reached its last line (what follows is the release of its locals)


class BamInBackground:
    """One rank, a device that will decode: the BAM is opened -- mapped, its pages touched by several threads, its BGZF block
    table walked, its header parsed (midas_bam_open_share with one share: the whole file) -- on a thread of its own while the
    main thread brings the device context up (the library's code objects, the HIP runtime: 0.15-0.2 s that need no file).
    wait() hands the handle out, or None when anything went wrong (the ordinary decode then says what)."""

    def __init__(self, path):
        import threading
        self._share, self._path = None, path
        self._thread = threading.Thread(target=self._open, daemon=True)
        self._thread.start()

    def _open(self):
        try:
            self._share = abi.BamShare(self._path, 0, 1)
        except Exception:
            self._share = None

    def wait(self):
        self._thread.join()
        share, self._share = self._share, None
        return share


def _lap(name, t0):
    """(MIDAS_SNPS_TRACE=1: the stage's phases on stderr, beside the library's own laps)"""
    now = time()
    if PHASES is not None:
        PHASES.append((name, now - t0))
    if os.environ.get("MIDAS_SNPS_TRACE"):
        sys.stderr.write("[stage] %-44s %9.3f ms\n" % (name, (now - t0) * 1e3))
    return now


def _since_process_start():
    """Seconds this process has existed (its start time in /proc, to the clock tick): what lies in front of the stage's first
    line -- the interpreter, the imports -- belongs to the command's wall time too."""
    try:
        import time as _t
        with open("/proc/self/stat", "rb") as f:
            ticks = int(f.read().rsplit(b")", 1)[1].split()[19])
        return _t.clock_gettime(_t.CLOCK_BOOTTIME) - ticks / float(os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError, IndexError, AttributeError):
        return 0.0


def _batch_groups(args, mine, order, ref_names, refid, reads, ctx):
    """The rank's work items as the batches the device takes one after the other: consecutive items of the emit order (species
    by species, a species' contigs sorted) until a batch would hold more reads or payload than a batch can (the library's
    limits are 2 * 10^9 reads and 32 GiB of payload per batch) or than the device's memory has room for beside its results.
    The reference streams contig by contig (midas/run/snps.py:187-199) and has no such limit; nearly every job is ONE batch."""
    wanted = set(mine)
    emit = [it for sp in sorted(order) for it in order[sp] if it in wanted]
    if len(emit) <= 1 or not hasattr(ctx, 'batch'):
        return [emit]
    max_reads = int(args.get('max_batch_reads') or MAX_BATCH_READS)
    max_payload = MAX_BATCH_PAYLOAD
    try:        # raw arrays + the direct layout ~ 3.6 bytes per base, 17 bytes of results per site: keep to half the device
        max_payload = min(max_payload, int(ctx.device_info()['hbm_bytes']) // 8)
    except Exception:
        pass
    # (nearly every job: everything fits one batch -- two sums, no per-contig tables)
    resident = isinstance(reads, abi.ResidentReads)       # (every column but refID is on the device: the bases' total came with the decode)
    bases = float(reads.l_seq_total) if resident else float(reads.l_seq.sum(dtype=np.int64))
    if int(refid.size) <= max_reads and 1.6 * bases + 8.0 * float(refid.size) <= max_payload:
        return [emit]
    index_of = {n: i for i, n in enumerate(ref_names)}
    n_ref = len(ref_names)
    per_reads = np.bincount(refid, minlength=n_ref) if refid.size else np.zeros(n_ref, np.int64)
    if resident:
        per_bases = per_reads * (bases / max(1, int(refid.size)))
    else:
        per_bases = np.bincount(refid, weights=reads.l_seq, minlength=n_ref) if refid.size else np.zeros(n_ref)
    n_pieces = {}
    for cid, _ in emit:
        n_pieces[cid] = n_pieces.get(cid, 0) + 1
    groups, cur, cur_reads, cur_bytes = [], [], 0, 0.0
    for it in emit:
        r = index_of.get(it[0], -1)
        k = float(n_pieces[it[0]])
        nr = int(per_reads[r] / k) if r >= 0 else 0
        nb = (1.6 * float(per_bases[r]) + 8.0 * float(per_reads[r])) / k if r >= 0 else 0.0
        if cur and (cur_reads + nr > max_reads or cur_bytes + nb > max_payload):
            groups.append(cur)
            cur, cur_reads, cur_bytes = [], 0, 0.0
        cur.append(it)
        cur_reads += nr
        cur_bytes += nb
    groups.append(cur)
    return groups


def _pileup_contigs(args, species_ids, mine, order, owner, decoded, ctx, span, contigs, halo=None):
    """count_coverage + keep_read + emit for the work items `mine` (contigs, or pieces of long ones) on one GPU.  Returns
    {species_id: partial aln_stats} (sums over this rank's items).  Writes <species>.snps.gz directly when ONE batch of this
    rank holds every item of the species, else one part file per run of consecutive (emit order) items of a batch."""
    ref_names, ref_lens, refid, reads = decoded
    rank, _ = dist.world()
    groups = _batch_groups(args, mine, order, ref_names, refid, reads, ctx) if mine else [[]]
    if len(groups) > 1 and args.get('log') is not None:
        args['log'].write("the rank's %d work items go to the device in %d batches (reads per batch <= %d)\n"
                          % (len(mine), len(groups), int(args.get('max_batch_reads') or MAX_BATCH_READS)))
    if len(groups) > 1 and getattr(reads, 'device', None) is not None and hasattr(ctx, 'fetch_payload') and not isinstance(reads, abi.ResidentReads):
        # (each batch regroups its own reads: in host memory, once.  Resident records are taken run by run where they lie.)
        decoded = (ref_names, ref_lens, refid, ctx.fetch_payload(reads))
    total = {}
    for group in groups:
        here = set(group)
        part = _pileup_batch(args, species_ids, group, order, lambda it: owner.get(it, 0) == rank and it in here,
                             decoded, ctx, span, contigs, halo, first=group is groups[0])
        for sp, st in part.items():
            if sp in total:
                for k in st:
                    total[sp][k] += st[k]
            else:
                total[sp] = dict(st)
    return total


def _pileup_batch(args, species_ids, mine, order, owned, decoded, ctx, span, contigs, halo, first):
    """One batch of _pileup_contigs: the work items `mine`, of which owned(item) says "in this batch"."""
    ref_names, ref_lens, refid, reads = decoded
    t_lap = time()
    table, sub, keys = _contig_table(species_ids, mine, span, contigs, ref_names, ref_lens, refid, reads, halo,
                                     fetch=getattr(ctx, 'fetch_payload', None))
    t_lap = _lap("  contig table + regroup", t_lap)
    thr = abi.Thresholds.from_args(args)
    batch = None
    if mine and hasattr(ctx, 'batch'):
        # the results stay on the device: the row writer takes them slab by slab through the context's page-locked ring
        # (midas_snps_batch_write_part), formatting one slab while the next crosses the link
        batch = ctx.batch(table, sub)
        t_lap = _lap("  batch_create", t_lap)
        try:
            batch.run(thr)
            _, _, stats = batch.fetch(counts=False, allele=False)
        except BaseException:
            batch.close()
            raise
        t_lap = _lap("  device pass + counters down", t_lap)
        counts = allele = None
    elif mine:      # (a test double of the device: one-shot call, host arrays)
        counts, allele, stats = ctx.pileup(thr, table, sub)
    else:       # more ranks than contigs: nothing to do here
        counts, allele = np.zeros((0, 4), np.uint32), np.zeros(0, np.uint8)
        stats = np.zeros((len(species_ids), abi.NUM_STATS), np.int64)
    try:
        out = _emit_contigs(args, species_ids, table, keys, order, owned, counts, allele, stats, batch, first)
        t_lap = _lap("  rows", t_lap)
        return out
    finally:
        if batch is not None:
            batch.close()
            _lap("  batch closed", t_lap)


def _emit_contigs(args, species_ids, table, keys, order, owned_item, counts, allele, stats, batch, first=True):
    """The rows and the partial counters of one batch, from host arrays or (batch) from the device results.  owned_item(item):
    the item is in this batch; `first`: the rank's first batch (the one that writes the header-only files of empty species)."""
    rank, _ = dist.world()
    genome_length = np.bincount(table.species, weights=table.length, minlength=len(species_ids)).astype(np.int64)
    pos = {it: k for k, it in enumerate(keys)}
    off = table.site_offsets()
    out = {}
    jobs = []       # (path, items, header): one table or part each

    def _write_rows(args_, path, table_, pos_, items, counts_, allele_, off_, header, batch_):
        jobs.append((path, items, header))
    for i, sp in enumerate(species_ids):
        cids = order[sp]
        owned = [bool(owned_item(cid)) for cid in cids]
        if all(owned):       # (a species without contigs still gets its header-only file, from rank 0)
            if cids or (rank == 0 and first):
                _write_rows(args, '%s/snps/output/%s.snps.gz' % (args['outdir'], sp), table, pos, cids, counts, allele, off, None, batch)
        else:
            k = 0
            while k < len(cids):     # maximal runs of consecutive contigs this rank owns; part k starts at sorted index k
                if not owned[k]:
                    k += 1
                    continue
                e = k
                while e < len(cids) and owned[e]:
                    e += 1
                _write_rows(args, _part_path(args, sp, k), table, pos, cids[k:e], counts, allele, off, k == 0, batch)
                k = e
        if any(owned) or (not cids and rank == 0 and first):
            out[sp] = {'genome_length': int(genome_length[i]),
                       'total_depth': int(stats[i, abi.STAT_TOTAL_DEPTH]),
                       'covered_bases': int(stats[i, abi.STAT_COVERED_BASES]),
                       'aligned_reads': int(stats[i, abi.STAT_ALIGNED_READS]),
                       'mapped_reads': int(stats[i, abi.STAT_MAPPED_READS])}
    _write_jobs(args, jobs, table, pos, counts, allele, off, batch)
    return out


def _write_jobs(args, jobs, table, pos, counts, allele, off, batch):
    """The tables and parts of _emit_contigs.  From a batch the files are written side by side: one file takes buffered
    writes at 2-3 GB/s however many threads feed it, the device part of each call is short and taken in turn
    (midas_snps_batch_write_part), so a few writers at a time keep the device's row coder and the page cache both busy."""
    writers = min(len(jobs), WRITERS) if batch is not None else 1
    if writers <= 1:
        for path, items, header in jobs:
            _write_rows(args, path, table, pos, items, counts, allele, off, header, batch)
        return
    with ThreadPoolExecutor(max_workers=writers) as pool:
        for f in [pool.submit(_write_rows, args, path, table, pos, items, counts, allele, off, header, batch)
                  for path, items, header in jobs]:
            f.result()


def _join_parts(args, species_ids, order, owner, rank):
    """Species whose work items were spread over ranks -- or over several batches of one rank: concatenate the parts in emit
    order (sorted contigs, a contig's pieces by position; a part is named by the index of its first item) into
    <species>.snps.gz.  Done by the rank that owns the species' first item (its first part carries the header)."""
    import glob
    for sp in species_ids:
        cids = order[sp]
        if not cids or owner.get(cids[0], 0) != rank:
            continue
        final = '%s/snps/output/%s.snps.gz' % (args['outdir'], sp)
        # (a part is named by the index of its first item, %06d: seven digits from the millionth item on -- by number, not by name)
        parts = sorted((p for p in glob.glob(glob.escape(final) + '.part*') if p[len(final) + 5:].isdigit()),
                       key=lambda p: int(p[len(final) + 5:]))
        if not parts:
            continue
        with open(final + '.tmp', 'wb') as dst:
            for path in parts:
                with open(path, 'rb') as src:
                    shutil.copyfileobj(src, dst, 1 << 24)
        os.replace(final + '.tmp', final)
        for path in parts:
            os.remove(path)


def _remove_stale_parts(args):
    """Part files of an earlier, interrupted run must not end up in this run's tables."""
    import glob
    for path in glob.glob(glob.escape('%s/snps/output' % args['outdir']) + '/*.snps.gz.part*'):
        if path.rsplit('.part', 1)[1].isdigit():
            os.remove(path)


def species_pileup(args, species_id, contigs):
    """midas/run/snps.py:164-216 for ONE species on GPU 0: writes <species>.snps.gz, returns (species_id, aln_stats)."""
    bampath = '%s/snps/temp/genomes.bam' % args['outdir']
    order, span = _whole(_species_contig_order([species_id], contigs), contigs)
    try:
        with abi.Context(int(os.environ.get("LOCAL_RANK", "0"))) as ctx:
            decoded = abi.read_bam(bampath, ctx if _inflate_on_device(args, ctx, bampath, 1) else None)
            stats = _pileup_contigs(args, [species_id], order[species_id], order, {}, decoded, ctx, span, contigs)
    except abi.MidasSnpsError as e:
        _exit_on(e)
    return (species_id, stats[species_id])


def _rank_local_plan(bampath, rank, ws, inflater=None):
    """Phase 1 of the rank-local decode (include/midas_snps.h, midas_bam_open_slice): walk this rank's share of the BAM,
    all-gather {first record, end, sorted, first/last refID} and the per-reference {reads, bases, first record offset} of
    every slice, and accept the slices only if they chain: slice 0 starts at the header's end, every slice ends where the
    next one starts (so every GUESSED record boundary is confirmed by a walk that began at an exact one), the last ends at
    the end of the file, and the references never go backwards.  Returns None when they do not (the caller then decodes
    the whole file, as a single rank does), else what the assignment and the range loads need."""
    error, sl = None, None
    try:
        sl = abi.BamSlice(bampath, rank, ws, ctx=inflater)       # (a Context: the slice is inflated and walked on its device)
    except abi.MidasSnpsError as e:
        error = "\nError: could not read %s\n%s\n" % (bampath, e.message)
    dist.agree_or_exit(error)
    n_ref = len(sl.ref_names)
    pos_sorted, first_pos, last_pos, span, marks = sl.marks()
    H = 12
    mine = np.concatenate([np.array([sl.first, sl.end, sl.sorted, sl.first_ref, sl.last_ref, sl.rec_begin, sl.total, n_ref,
                                     pos_sorted, first_pos, last_pos, marks.shape[0]], np.int64),
                           sl.ref_reads, sl.ref_bases, sl.ref_first, span])
    allv = dist.all_gather_i64(mine)
    head = allv[:, :H]
    ok = bool((head[:, 7] == n_ref).all() and (head[:, 2] == 1).all() and head[0, 0] == head[0, 5] and head[-1, 1] == head[0, 6])
    last_ref, last_at = -1, -1
    in_order = bool((head[:, 8] == 1).all())      # positions inside every reference: needed only to cut long contigs
    for r in range(ws):
        if r + 1 < ws and head[r, 1] != head[r + 1, 0]:
            ok = False
        if head[r, 3] >= 0:
            if head[r, 3] < last_ref:
                ok = False
            if head[r, 3] == last_ref and head[r, 9] < last_at:
                in_order = False
            last_ref, last_at = head[r, 4], head[r, 10]
    if not ok:
        sl.close()
        return None
    reads_per = allv[:, H:H + n_ref].sum(axis=0)
    bases_per = allv[:, H + n_ref:H + 2 * n_ref].sum(axis=0)
    firsts = allv[:, H + 2 * n_ref:H + 3 * n_ref]
    ref_first = np.where(firsts >= 0, firsts, np.iinfo(np.int64).max).min(axis=0)
    ref_first[reads_per == 0] = -1
    ref_span = allv[:, H + 3 * n_ref:H + 4 * n_ref].max(axis=0)
    # the marks of every slice (a second, padded all-gather): per (reference, bin) the smallest offset
    n_marks = int(head[:, 11].max())
    pad = np.full((n_marks, 3), -1, np.int64)
    pad[:marks.shape[0]] = marks
    allm = dist.all_gather_i64(pad.reshape(-1)).reshape(-1, 3) if n_marks else pad
    allm = allm[allm[:, 0] >= 0]
    key = (allm[:, 0] << 32) | allm[:, 1]
    o = np.lexsort((allm[:, 2], key))
    key, off = key[o], allm[o, 2]
    firstk = np.ones(key.size, bool)
    firstk[1:] = key[1:] != key[:-1]
    # where every reference's records end: at the first record of the next reference that has any
    have = np.nonzero(ref_first >= 0)[0]
    ref_end = np.full(n_ref, -1, np.int64)
    for k, r in enumerate(have):
        ref_end[r] = int(ref_first[have[k + 1]]) if k + 1 < len(have) else int(head[0, 6])
    return dict(slice=sl, ref_names=sl.ref_names, ref_lens=sl.ref_lens, ref_reads=reads_per, ref_bases=bases_per.astype(np.float64),
                ref_first=ref_first, ref_end=ref_end, total=int(head[0, 6]), pos_sorted=in_order, ref_span=ref_span,
                mark_key=key[firstk], mark_off=off[firstk])


def _header_lengths(bampath):
    """The reference lengths of the BAM's header (None when it cannot be read: the decode that follows says why)."""
    try:        # (the header's blocks alone: mapping and walking the whole file for it cost a rank of eight a third of a second)
        return [int(x) for x in bam.read_header(bampath)[1]]
    except Exception:
        return None


def _one_pass_shares(bampath, rank, ws):
    """The one-pass form of the rank-local decode (include/midas_snps.h, midas_bam_open_share): the file is dealt to the ranks as
    contiguous runs of whole contigs -- every rank looks where its equal share of the bytes begins and walks a few blocks on to
    the next contig's first record; one all-gather of those offsets; rank r then decodes [first_r, first_r+1) ONCE, and that
    decode proves the next rank's guess (a range must end on a record border; rank 0 starts at the header's end).  Returns
    (handle, begin, end) or None when a rank found no contig border nearby (the caller plans with the slices, as before)."""
    error, sh = None, None
    # Each rank walks the BGZF chain over ITS 1 / N of the file's bytes only (a LOCAL block table: eight ranks that each walked --
    # and paged in the block headers of -- the whole 9 GB file spent a third of a second each on it); where a rank's walk begins is
    # a guess, so the walks are exchanged and believed only if they chain: rank 0 from offset 0, each ending where the next begins,
    # the last at the file's end.  Then every rank knows its table's place in the uncompressed stream.  If they do not chain (a
    # block header's look-alike inside compressed data): every rank walks the whole file, as before round 6.
    try:
        sh = abi.BamShare.open_local(bampath, rank, ws)
    except abi.MidasSnpsError as e:
        error = "\nError: could not read %s\n%s\n" % (bampath, e.message)
    dist.agree_or_exit(error)
    walks = dist.all_gather_i64(sh.walk)
    chained = bool(walks[0, 0] == 0 and (walks[:-1, 1] == walks[1:, 0]).all() and walks[-1, 1] == walks[0, 3]
                   and (walks[:, 3] == walks[0, 3]).all())
    try:
        if chained:
            sh.locate(int(walks[:rank, 2].sum()), int(walks[:, 2].sum()))
        else:
            sh.close()
            sh = abi.BamShare(bampath, rank, ws)
    except abi.MidasSnpsError as e:
        error = "\nError: could not read %s\n%s\n" % (bampath, e.message)
    dist.agree_or_exit(error)
    allv = dist.all_gather_i64([sh.first, sh.total, sh.rec_begin, len(sh.ref_names)])
    firsts = [int(x) for x in allv[:, 0]] + [int(allv[0, 1])]
    ok = bool((allv[:, 0] >= 0).all() and (allv[:, 1] == allv[0, 1]).all() and (allv[:, 3] == allv[0, 3]).all()
              and firsts[0] == int(allv[0, 2]) and all(a <= b for a, b in zip(firsts[:-1], firsts[1:])))
    if not ok:
        sh.close()
        return None
    return sh, firsts[rank], firsts[rank + 1]


def _offset_at(plan, ref, x):
    """An offset at or in front of reference `ref`'s first record at a position >= x, and behind every record at a position
    below x rounded down to the marks' grid (midas_bam_slice_marks; positions sorted inside the reference)."""
    b = int(x) // pieces.MARK_SPAN
    if b <= 0:
        return int(plan['ref_first'][ref])
    key = plan['mark_key']
    i = int(np.searchsorted(key, (int(ref) << 32) | b, side='left'))
    if i < key.size and int(key[i]) >> 32 == ref:
        return int(plan['mark_off'][i])
    return int(plan['ref_end'][ref])


def _piece_range(plan, ref, lo, hi, last):
    """Uncompressed [begin, end) holding the records piece [lo, hi) of reference `ref` needs: those that start in it and the
    ones in front that can reach into it (at most the reference's longest read span away)."""
    if plan['ref_first'][ref] < 0:
        return None
    b = _offset_at(plan, ref, max(0, lo - int(plan['ref_span'][ref]))) if lo > 0 else int(plan['ref_first'][ref])
    e = int(plan['ref_end'][ref]) if last else _offset_at(plan, ref, hi)
    return (b, max(b, e))


def _record_ranges(plan, wanted):
    """Uncompressed [begin, end) record ranges of the given (reference, lo, hi, last) pieces -- a whole reference: from its
    first record to the first record of the next reference that has any (the file is coordinate-sorted: _rank_local_plan
    checked) -- merged where they touch or overlap (a piece's halo lies in the piece before it)."""
    ranges = sorted(r for r in (_piece_range(plan, *w) for w in wanted) if r is not None)
    merged = []
    for b, e in ranges:
        if merged and merged[-1][1] >= b:
            merged[-1] = (merged[-1][0], max(e, merged[-1][1]))
        else:
            merged.append((b, e))
    return merged


def _device_context():
    return abi.Context(int(os.environ.get("LOCAL_RANK", "0")))


def pysam_pileup(args, species, contigs, make_context=_device_context):
    """midas/run/snps.py:219-244.  Name kept for drop-in; there is no pysam underneath.
    N ranks: contigs are the work items (the unit count_coverage is called on, :187-199), dealt to the ranks by
    longest-processing-time over bytes of aligned reads + sites; a rank piles up its contigs, writes their rows, and one
    all-gather of [n_species, 5] partial counters follows.  make_context: tests substitute a CPU double of the device.
    The context is opened first: the device may also inflate the BAM's blocks (_inflate_on_device)."""
    error, ctx = None, None
    stack = ExitStack()
    t_lap = time()
    try:
        ctx = stack.enter_context(make_context())
    except abi.MidasSnpsError as e:
        error = _error_text(e)
    except Exception as e:
        error = "\nError: %s: %s\n" % (type(e).__name__, e)
    dist.agree_or_exit(error)
    _lap("device context (library loaded, HIP runtime up)", t_lap)
    with stack:
        # the drop-in runs the dependency's rule for the CIGAR op P (pysam's get_aligned_pairs treats BAM_CPAD like an
        # insertion: the query position advances); --pad_rule spec selects the SAM specification's (P consumes nothing)
        if hasattr(ctx, 'set_pad_rule'):
            ctx.set_pad_rule(abi.PAD_SPEC if args.get('pad_rule', 'pysam') == 'spec' else abi.PAD_PYSAM)
        line = dist.attach_context(ctx)      # (N ranks: the RCCL communicator of their devices, for the summary rows)
        if line and dist.world()[0] == 0 and args.get('log') is not None:
            args['log'].write(line + "\n")
        try:
            return _count_alleles(args, species, contigs, ctx)
        finally:
            # (between _count_alleles' last line and here its locals went: the decoded BAM's handle -- the file's mapping, the
            # device arena with the inflated stream and the records -- and the batch's results)
            t_lap = _lap("the rank's reads and results released", _T_END[0] if _T_END[0] else time())
            dist.detach_context()
            stack.close()
            _lap("device context closed", t_lap)


def _inflate_on_device(args, ctx, bampath, ws):
    """args['device_inflate']: 'on' / True, 'off' / False, or 'auto' (the default) -- on where the measurements of DESIGN.md
    3.4 have the device ahead and its one-call arena (about eight times the compressed bytes) fits beside everything else:
      * one rank and a BAM of at least 0.5 GB whose decode + batch fit the device's memory (configs[2]: stage 0.17-0.18 s
        against 0.83-0.91 s with the host's threads inflating; smaller files: the host's threads are as fast).  A file of
        several device-fills of blocks is decoded group by group (snps_abi.hip device_decode_stream: ~2 x the compressed bytes
        of slots + the resident records, ~3 x): the rule below is the one-arena decode's, which a record longer than a
        group's tail, or a payload beyond the direct layout's 32 GiB, still falls back to;
      * a rank with few CPUs -- eight ranks share a node's cores under torchrun -- and a share of the BAM of 32 MB-8 GB (a
        1.26 GB BAM with the 2 CPUs of an 8-rank node's rank: decode 0.87 s against 3.0 s)."""
    want = args.get('device_inflate', 'auto')
    if not getattr(ctx, 'inflates', False) or want in (False, 'off'):
        return False
    if want in (True, 'on'):
        return True
    try:
        size = os.path.getsize(bampath)
    except OSError:
        return False
    if ws == 1 and size >= (512 << 20):
        try:
            hbm = int(ctx.device_info()['hbm_bytes'])
        except Exception:
            hbm = 0
        return size * 9 <= hbm           # arena ~8 x (the batch reads the decoder's own layout: no copy of its own); results < 1 x
    return utility.cpu_budget() <= 4 and (32 << 20) <= size // max(1, ws) <= (8 << 30)


def _count_alleles(args, species, contigs, ctx):
    start = time()
    rank, ws = dist.world()
    if rank == 0:
        print("\nCounting alleles")
        args['log'].write("\nCounting alleles\n")

    bampath = '%s/snps/temp/genomes.bam' % args['outdir']
    if rank == 0:
        _remove_stale_parts(args)       # (before the first point every rank waits at)
    inflater = ctx if _inflate_on_device(args, ctx, bampath, ws) else None
    # N ranks, ONE pass where the contigs are short beside a rank's share (the usual metagenome): contiguous shares of whole
    # contigs, every block inflated once, by the rank that piles its records up (_one_pass_shares).  A contig longer than the
    # split length wants to be cut into pieces, which needs the slices' walk: the two-pass plan below.
    error = None
    decoded = None
    plan, share = None, None
    split_length = int(args.get('split_length', SPLIT_LENGTH))
    if ws > 1:
        lengths = _header_lengths(bampath)
        if lengths is not None and (split_length <= 0 or max(lengths, default=0) <= pieces.piece_length(split_length)):
            share = _one_pass_shares(bampath, rank, ws)
        if share is not None:
            retry = 0
            try:
                refid, reads = share[0].load_ranges([(share[1], share[2])], inflater, resident=True)
                decoded = (share[0].ref_names, share[0].ref_lens, refid, reads)
                share[0].release_file()      # (the share's bytes are on the device / decoded: the mapping goes while the records are piled up)
            except abi.MidasSnpsError as e:
                if e.status == abi.ERR_BAD_LAYOUT:
                    # a guessed border that is no record border -- this rank's end, or its own start (then the decode meets
                    # something that is no record): plan with the slices instead, whose walk vouches for every border; a file
                    # that really is damaged is reported by that plan, with the block or record it stumbles over
                    retry = 1
                elif inflater is not None and args.get('device_inflate', 'auto') == 'auto' and e.status in (abi.ERR_OUT_OF_MEMORY, abi.ERR_HIP):
                    retry = 1
                else:
                    error = "\nError: could not read %s\n%s\n" % (bampath, e.message)
            dist.agree_or_exit(error)
            if int(dist.all_gather_i64([retry])[:, 0].max()):
                share[0].close()
                share, decoded = None, None
        # ... else two passes: every rank walks its share of the BAM's bytes, the ranks exchange a few numbers per reference, and
        # each decodes only the records of the contigs (or pieces) it ends up owning.  A BAM the slices cannot vouch for (not
        # coordinate-sorted, a guessed record boundary the neighbouring slice does not confirm): decoded whole, as one rank does.
        if share is None:
            plan = _rank_local_plan(bampath, rank, ws, inflater)
    if share is not None:
        # whole contigs per rank only if the file is grouped by reference: every share's refIDs ascending, and the next share's
        # first beyond this one's last
        ref_names, ref_lens, refid, reads = decoded
        grouped = int(refid.size < 2 or bool((refid[1:] >= refid[:-1]).all()))
        have = dist.all_gather_i64([int(refid[0]) if refid.size else -1, int(refid[-1]) if refid.size else -1, int(reads.n_reads), grouped])
        seen = -1
        for k in range(ws):
            if have[k, 0] >= 0:
                grouped = grouped and have[k, 0] > seen
                seen = int(have[k, 1])
        if not (grouped and bool((have[:, 3] == 1).all())):
            share[0].close()
            share, decoded = None, None
            plan = _rank_local_plan(bampath, rank, ws, inflater)
    if share is not None:
        read_bytes = np.zeros(len(ref_names))
        if rank == 0:
            line = "rank-local BAM decode: %d slices chained in ONE pass (contiguous shares of whole contigs), %d records; records decoded per rank: %s" % (
                ws, int(have[:, 2].sum()), ' '.join(str(int(x)) for x in have[:, 2]))
            print("  " + line)
            args['log'].write(line + "\n")
    elif plan is None:
        try:
            # (one rank, every contig its own: SEQ / QUAL / CIGAR can stay on the device the blocks were inflated on)
            try:
                # ... in the pileup kernel's own layout, every column included (abi.ResidentReads): ONE pass from the file to the tallies
                t_in = time()
                opened = args.pop('_bam_opener').wait() if args.get('_bam_opener') is not None else None
                t_in = _lap("  BAM opened: its block table waited for", t_in)
                if opened is not None and inflater is not None and ws == 1 and 0 <= opened.first < opened.total:
                    # (the file was mapped and its block table walked while the device context came up)
                    refid, rr = opened.load_ranges([(opened.first, opened.total)], inflater, resident=True)
                    t_in = _lap("  decode on the device (resident)", t_in)
                    opened.release_file()        # (its bytes are on the device: the mapping goes while the records are piled up)
                    _lap("  file released", t_in)
                    decoded = (opened.ref_names, opened.ref_lens, refid, rr)
                else:
                    if opened is not None:
                        opened.close()
                    decoded = abi.read_bam(bampath, inflater, resident=inflater is not None and ws == 1)
            except abi.MidasSnpsError as e:
                # 'auto' chose the device and the device could not (its memory, a HIP error): the host's threads can
                if inflater is None or args.get('device_inflate', 'auto') != 'auto' or e.status not in (abi.ERR_OUT_OF_MEMORY, abi.ERR_HIP):
                    raise
                args['log'].write("device decode failed (%s): decoding with the host's threads\n" % e.message)
                inflater = None
                decoded = abi.read_bam(bampath)
        except abi.MidasSnpsError as e:
            error = "\nError: could not read %s\n%s\n" % (bampath, e.message)
        dist.agree_or_exit(error)
        ref_names, ref_lens, refid, reads = decoded
        # (the weights only decide which rank takes which contig: with one rank there is nothing to decide, and a weighted
        # bincount over 80 M reads is half a second of one core)
        read_bytes = np.bincount(refid, weights=reads.l_seq, minlength=len(ref_names)) if refid.size and ws > 1 else np.zeros(len(ref_names))
    else:
        ref_names, ref_lens, read_bytes = plan['ref_names'], plan['ref_lens'], plan['ref_bases']
        if rank == 0:
            args['log'].write("rank-local BAM decode: %d slices chained, %d records\n" % (ws, int(plan['ref_reads'].sum())))

    # work item -> rank by bytes of aligned reads + sites (every rank computes the same assignment from the same numbers).
    # An item is a contig -- the unit count_coverage is called on -- or, for a contig longer than the split length in a BAM
    # whose positions are sorted, a piece of it (midas_amd/pieces.py): one 20 Mb chromosome must not pin the job to one GPU.
    all_ids = sorted(species)
    t_in = time()
    # first use of the genomes: what their reader raised (a missing genome's sys.exit, an OSError) ends every rank together
    try:
        if isinstance(contigs, ContigsInBackground):
            contigs = contigs.wait()
            t_in = _lap("  genomes waited for", t_in)
    except SystemExit as e:
        error = dist.exit_message(e)
    except Exception as e:
        error = "\nError: %s: %s\n" % (type(e).__name__, e)
    dist.agree_or_exit(error)
    if isinstance(contigs, DealtContigs):
        contigs.exchange()          # (every rank read its run of the genome files: now every rank knows every contig)
    by_species = _species_contig_order(all_ids, contigs)
    ref_index = {n: i for i, n in enumerate(ref_names)}
    piece_len = pieces.piece_length(int(args.get('split_length', SPLIT_LENGTH))) if plan is not None and plan['pos_sorted'] else 0
    order, span, weight, halo = {sp: [] for sp in all_ids}, {}, {}, {}
    n_cut = 0
    for sp in all_ids:
        for cid in by_species[sp]:
            r = ref_index.get(cid, -1)
            length = contigs[cid].length
            cuts = pieces.cut(length, piece_len) if r >= 0 and piece_len and plan['ref_first'][r] >= 0 else [(0, length)]
            n_cut += len(cuts) > 1
            for j, (lo, hi) in enumerate(cuts):
                it = (cid, j)
                order[sp].append(it)
                span[it] = (lo, hi, j + 1 == len(cuts))
                frac = 1.0
                if len(cuts) > 1:        # the piece's share of the contig's reads: its share of the record bytes
                    b, e = _piece_range(plan, r, lo, hi, span[it][2])
                    frac = (e - b) / float(max(1, plan['ref_end'][r] - plan['ref_first'][r]))
                    halo[cid] = int(plan['ref_span'][r])
                # ~1.63 B per aligned base, 17 B per site
                weight[it] = 1.6 * frac * float(read_bytes[r] if r >= 0 else 0.0) + 17.0 * (hi - lo)
    if share is not None:
        # a contig is piled up by the rank that decoded its records; one without any goes to the last rank whose records lie in
        # front of it (every rank computes the same assignment from the same few numbers)
        owner = {}
        for sp in all_ids:
            for it in order[sp]:
                r = ref_index.get(it[0], -1)
                at = 0
                for k in range(ws):
                    if have[k, 0] >= 0 and have[k, 0] <= r:
                        at = k
                owner[it] = at
    else:
        owner = dist.shard_items(weight, ws)
    mine = [it for sp in all_ids for it in order[sp] if owner[it] == rank]
    if plan is not None:
        try:
            refid, reads = plan['slice'].load_ranges(_record_ranges(plan, [(ref_index[cid],) + span[(cid, j)] for cid, j in mine
                                                                           if cid in ref_index]), inflater)
            decoded = (ref_names, ref_lens, refid, reads)
        except abi.MidasSnpsError as e:
            error = "\nError: could not read %s\n%s\n" % (bampath, e.message)
        dist.agree_or_exit(error)
        if n_cut:
            per_rank = dist.all_gather_i64([reads.n_reads])[:, 0]
            if rank == 0:
                args['log'].write("long contigs: %d cut into pieces of %d positions; records decoded per rank: %s of %d\n"
                                  % (n_cut, piece_len, ' '.join(str(int(x)) for x in per_rank), int(plan['ref_reads'].sum())))

    _lap("  work items", t_in)
    t_lap = _lap("decode, plan, genomes, work items", start)
    local = {}
    try:
        if isinstance(contigs, DealtContigs):
            late = contigs.need({cid for cid, _ in mine})
            if late and args.get('log') is not None:
                args['log'].write("genomes: %d species of this rank's contigs were in another rank's run of the files: read now\n" % late)
        local = _pileup_contigs(args, all_ids, mine, order, owner, decoded, ctx, span, contigs, halo)
        t_lap = _lap("pileup of the rank's contigs (all batches)", t_lap)
    except abi.MidasSnpsError as e:
        error = _error_text(e)
    except SystemExit as e:
        error = dist.exit_message(e)
    except Exception as e:      # (an OSError from the decoder, a MemoryError ...: the other ranks must not wait for this one)
        error = "\nError: %s: %s\n" % (type(e).__name__, e)
    dist.agree_or_exit(error)

    # one all-gather of the per-species partial counters; per-site output stays on its rank
    rows = np.zeros((len(all_ids), 5), dtype=np.int64)
    for i, sp in enumerate(all_ids):
        if sp in local:
            st = local[sp]
            rows[i] = [st['genome_length'], st['covered_bases'], st['total_depth'], st['aligned_reads'], st['mapped_reads']]
    rows = dist.all_gather_summary(rows)
    _join_parts(args, all_ids, order, owner, rank)       # (the all-gather is also the "every part is on disk" point)
    _lap("summary rows exchanged, parts joined", t_lap)

    # update alignment stats for species objects -- midas/run/snps.py:230-241
    for i, species_id in enumerate(all_ids):
        sp = species[species_id]
        sp.genome_length = int(rows[i, 0])
        sp.covered_bases = int(rows[i, 1])
        sp.total_depth = int(rows[i, 2])
        sp.aligned_reads = int(rows[i, 3])
        sp.mapped_reads = int(rows[i, 4])
        if sp.genome_length > 0:
            sp.fraction_covered = sp.covered_bases / float(sp.genome_length)
        if sp.covered_bases > 0:
            sp.mean_coverage = sp.total_depth / float(sp.covered_bases)

    if rank == 0:
        print("  %s minutes" % round((time() - start) / 60, 2))
        print("  %s Gb maximum memory" % utility.max_mem_usage())
    _T_END[0] = time()


def snps_summary(args, species):
    """Get summary of mapping statistics -- midas/run/snps.py:247-262"""
    fields = ['species_id', 'genome_length', 'covered_bases', 'fraction_covered', 'mean_coverage', 'aligned_reads', 'mapped_reads']
    outfile = open(args['outdir'] + '/snps/summary.txt', 'w')
    outfile.write('\t'.join(fields) + '\n')
    for sp in species.values():
        outfile.write(sp.id + '\t')
        outfile.write(str(sp.genome_length) + '\t')
        outfile.write(str(sp.covered_bases) + '\t')
        outfile.write(str(sp.fraction_covered) + '\t')
        outfile.write(str(sp.mean_coverage) + '\t')
        outfile.write(str(sp.aligned_reads) + '\t')
        outfile.write(str(sp.mapped_reads) + '\n')
    outfile.close()


def remove_tmp(args):
    """Remove specified temporary files -- midas/run/snps.py:264-266"""
    shutil.rmtree('/'.join([args['outdir'], 'snps/temp']))


def run_pipeline(args):
    """Run entire pipeline -- midas/run/snps.py:268-305"""
    # (N ranks meet in the sample's temp directory: no process group, no torch -- midas_amd/dist.py)
    t_lap = time()
    if PHASES is not None or os.environ.get("MIDAS_SNPS_TRACE"):
        _lap("process start -> run_pipeline (interpreter, imports, arguments)", t_lap - _since_process_start())
    rank, ws = dist.init_from_env(rendezvous_dir=os.path.join(args['outdir'], 'snps', 'temp'))
    t_lap = _lap("ranks met", t_lap)

    print("\nReading reference data")
    start = time()
    species = initialize_species(args)
    # (the genomes are read on a thread of their own and waited for where the pileup first needs them)
    # (with --build_db / --align in front of the pileup the reader starts behind them: rank 0 reads the same files there)
    # (N ranks: the genome files are dealt to the ranks -- DealtContigs)
    contigs = (ContigsInBackground(species, start=not (args['build_db'] or args['align']), deal=(rank, ws)) if args['call']
               else initialize_contigs(species))
    print("  %s minutes" % round((time() - start) / 60, 2))
    print("  %s Gb maximum memory" % utility.max_mem_usage())

    # rank 0 builds the database and maps the reads (bowtie2 / samtools on PATH, as in the reference); a failure there -- a
    # sys.exit of those stages, a missing binary, a full disk -- is agreed on by every rank instead of leaving the others at
    # the barrier until the collective's timeout
    error = None
    if rank == 0:
        try:
            if args['build_db']:
                print("\nBuilding database of representative genomes")
                args['log'].write("\nBuilding database of representative genomes\n")
                start = time()
                build_genome_db(args, species)
                print("  %s minutes" % round((time() - start) / 60, 2))
                print("  %s Gb maximum memory" % utility.max_mem_usage())
            if args['align']:
                args['file_type'] = utility.auto_detect_file_type(args['m1'])
                print("\nMapping reads to representative genomes")
                args['log'].write("\nMapping reads to representative genomes\n")
                start = time()
                genome_align(args)
                print("  %s minutes" % round((time() - start) / 60, 2))
                print("  %s Gb maximum memory" % utility.max_mem_usage())
        except SystemExit as e:
            error = dist.exit_message(e)
        except Exception as e:
            error = "\nError: %s: %s\n" % (type(e).__name__, e)
    dist.agree_or_exit(error)

    if args['call']:
        if isinstance(contigs, ContigsInBackground):
            contigs.start()
        if rank == 0:
            index_bam(args)
        if ws == 1 and args.get('device_inflate', 'auto') not in (False, 'off') and '_bam_opener' not in args:
            args['_bam_opener'] = BamInBackground('%s/snps/temp/genomes.bam' % args['outdir'])
        t_lap = _lap("species, genome reader started, index_bam", t_lap)
        pysam_pileup(args, species, contigs)
        t_lap = time()
        if rank == 0:
            snps_summary(args, species)
    dist.barrier()
    dist.finalize()      # (rank 0 returns when every rank has left the rendezvous directory: it may go with temp/)
    _lap("summary.txt, ranks leave", t_lap)

    if args['remove_temp'] and rank == 0:
        remove_tmp(args)
