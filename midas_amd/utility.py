"""Small host helpers shared by the snps commands.  Behaviour follows the helpers of the same names in the reference's
midas/utility.py (iopen :194-206, check_database :171-192, auto_detect_file_type :152-159, check_compression :161-169,
check_exit_code :227-232, max_mem_usage :218-225)."""

import bz2
import gzip
import io
import os
import platform
import resource
import shutil
import sys

_OPENERS = {'gz': gzip.open, 'bz2': bz2.BZ2File}


def iopen(inpath, mode='r'):
    """Text handle on a plain, gzip (.gz) or bzip2 (.bz2) file, chosen by extension."""
    opener = _OPENERS.get(inpath.rsplit('.', 1)[-1])
    if 'b' in mode:         # (bytes: the FASTA of a representative genome goes to the device as bytes)
        return opener(inpath, mode) if opener else open(inpath, mode)
    return io.TextIOWrapper(opener(inpath, mode)) if opener else open(inpath, mode)


def max_mem_usage():
    """Peak resident memory of this process and its children, in GB (ru_maxrss is kB on Linux, bytes on macOS)."""
    peak = sum(resource.getrusage(who).ru_maxrss for who in (resource.RUSAGE_SELF, resource.RUSAGE_CHILDREN))
    return round(peak / (1e6 if platform.system() == 'Linux' else 1e9), 2)


def check_exit_code(process, command):
    """Wait for a shell stage; a non-zero exit ends the run with the stage's stderr."""
    _, err = process.communicate()
    if process.returncode != 0:
        sys.exit("\nError encountered executing:\n%s\n\nError message:\n%s\n" % (command, err))


def auto_detect_file_type(inpath):
    """'fasta' or 'fastq' from the first character of the file."""
    with iopen(inpath) as handle:
        first = handle.read(1)
    kind = {'>': 'fasta', '@': 'fastq'}.get(first)
    if kind is None:
        sys.exit("Error: filetype [fasta, fastq] of %s could not be recognized" % inpath)
    return kind


def check_compression(inpath):
    """The extension has to tell the truth about the compression: try to read one line."""
    try:
        with iopen(inpath) as handle:
            next(handle)
    except Exception:
        sys.exit("\nError: File extension '%s' does not match expected compression" % inpath.rsplit('.', 1)[-1])


def check_database(args):
    """The pieces of a MIDAS database the snps command touches."""
    db = args['db']
    if db is None:
        sys.exit("\nError: No reference database specified\nUse the flag -d to specify a database,\n"
                 "Or set the MIDAS_DB environmental variable: export MIDAS_DB=/path/to/midas/db\n")
    if not os.path.isdir(db):
        sys.exit("\nError: Specified reference database does not exist: %s\n" % db)
    for name, kind in (('species_info.txt', 'file'), ('genome_info.txt', 'file'), ('marker_genes', 'directory'),
                       ('pan_genomes', 'directory'), ('rep_genomes', 'directory')):
        path = os.path.join(db, name)
        if not os.path.exists(path):
            sys.exit("\nError: Could not locate required database %s: %s\n" % (kind, path))


def find_executable(name):
    """bowtie2 / samtools for --build_db and --align come from PATH (the reference ships its own binaries; this
    build ships none).  None when absent: only those two stages need them."""
    return shutil.which(name)


def cpu_budget():
    """CPUs this process may use: os.cpu_count(), or fewer under a cgroup CPU quota (cpu.max of cgroup v2,
    cpu.cfs_quota_us / cpu.cfs_period_us of v1) -- the same rule as the native library's midas::cpu_budget()."""
    import math
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = period = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()[:2]
            if q != 'max':
                quota, period = float(q), float(p)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as fq, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fp:
                quota, period = float(fq.read()), float(fp.read())
        except (OSError, ValueError):
            pass
    if quota and period and quota > 0 and period > 0:
        n = max(1, min(n, int(math.ceil(quota / period))))
    try:        # one process per GPU under torchrun: the node's CPUs are shared by LOCAL_WORLD_SIZE ranks
        ranks = int(os.environ.get('LOCAL_WORLD_SIZE', '1'))
    except ValueError:
        ranks = 1
    if ranks > 1:
        n = max(1, n // ranks)
    return n
