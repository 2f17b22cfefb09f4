"""The few helpers of midas/utility.py that the snps pileup path touches (reference file:line cited per function)."""

import bz2
import gzip
import io
import os
import platform
import resource
import sys


def iopen(inpath, mode='r'):
    """Open a file regardless of compression [gzip, bzip] -- midas/utility.py:194-206 (python3 branch)."""
    ext = inpath.split('.')[-1]
    if ext == 'gz':
        return io.TextIOWrapper(gzip.open(inpath, mode))
    elif ext == 'bz2':
        return io.TextIOWrapper(bz2.BZ2File(inpath, mode))
    else:
        return open(inpath, mode)


def max_mem_usage():
    """Max mem usage (Gb) of self and child processes -- midas/utility.py:218-225."""
    max_mem_self = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    max_mem_child = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    if platform.system() == 'Linux':
        return round((max_mem_self + max_mem_child) / float(1e6), 2)
    else:
        return round((max_mem_self + max_mem_child) / float(1e9), 2)


def check_exit_code(process, command):
    """Capture stdout, stderr; exit on non-zero unix exit code -- midas/utility.py:227-232."""
    out, err = process.communicate()
    if process.returncode != 0:
        err_message = "\nError encountered executing:\n%s\n\nError message:\n%s\n" % (command, err)
        sys.exit(err_message)


def auto_detect_file_type(inpath):
    """FASTA or FASTQ from the first character -- midas/utility.py:152-159."""
    infile = iopen(inpath)
    for line in infile:
        if line[0] == '>':
            return 'fasta'
        elif line[0] == '@':
            return 'fastq'
        else:
            sys.exit("Error: filetype [fasta, fastq] of %s could not be recognized" % inpath)
    infile.close()


def check_compression(inpath):
    """The extension must match the compression -- midas/utility.py:161-169."""
    ext = inpath.split('.')[-1]
    file = iopen(inpath)
    try:
        next(file)
        file.close()
    except Exception:
        sys.exit("\nError: File extension '%s' does not match expected compression" % ext)


def check_database(args):
    """The MIDAS_DB layout the snps command needs -- midas/utility.py:171-192."""
    if args['db'] is None:
        error = "\nError: No reference database specified\n"
        error += "Use the flag -d to specify a database,\n"
        error += "Or set the MIDAS_DB environmental variable: export MIDAS_DB=/path/to/midas/db\n"
        sys.exit(error)
    if not os.path.isdir(args['db']):
        sys.exit("\nError: Specified reference database does not exist: %s\n" % args['db'])
    for file in ['species_info.txt', 'genome_info.txt']:
        path = '%s/%s' % (args['db'], file)
        if not os.path.exists(path):
            sys.exit("\nError: Could not locate required database file: %s\n" % path)
    for dir in ['marker_genes', 'pan_genomes', 'rep_genomes']:
        path = '%s/%s' % (args['db'], dir)
        if not os.path.exists(path):
            sys.exit("\nError: Could not locate required database directory: %s\n" % path)


def find_executable(name):
    """PATH lookup for the aligner stage (the reference ships prebuilt binaries, midas/utility.py:109-150;
    none are in this image, so --build_db / --align need bowtie2 and samtools on PATH)."""
    for d in os.environ.get('PATH', '').split(os.pathsep):
        p = os.path.join(d, name)
        if os.path.isfile(p) and os.access(p, os.X_OK):
            return p
    return None
