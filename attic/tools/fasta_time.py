"""Developer script: the genomes of a configs[3]-size database (100 species x 16 contigs x 250 kb) read by the interpreter's own
reader (midas_amd/fasta.py, what the host used until round 5) and by the library's (midas_fasta_load), first call of the process
and the next ones.  usage: python tools/fasta_time.py [workdir]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from midas_amd import abi, fasta  # noqa: E402

work = sys.argv[1] if len(sys.argv) > 1 else '/tmp/midas_fasta_time'
rng = np.random.default_rng(1)
paths = []
for sp in range(100):
    d = os.path.join(work, 'sp%03d' % sp)
    os.makedirs(d, exist_ok=True)
    paths.append(os.path.join(d, 'genome.fna'))
    if not os.path.exists(paths[-1]):
        with open(paths[-1], 'wb') as f:
            for c in range(16):
                seq = rng.choice(np.frombuffer(b'ACGT', np.uint8), 250000).tobytes()
                f.write(b'>c%d_%d some description\n' % (sp, c))
                f.write(b'\n'.join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b'\n')
order = sys.argv[2] if len(sys.argv) > 2 else 'native-first'
def py():
    t = time.perf_counter(); pool = bytearray()
    for p in paths:
        for rid, seq in fasta.parse_bytes(open(p, 'rb').read()):
            pool += seq.upper()
    print('interpreter: %.3f s (%d bytes)' % (time.perf_counter() - t, len(pool)), flush=True)
def nat():
    t = time.perf_counter(); pool, recs = abi.read_fasta_files(paths)
    print('library:     %.3f s (%d bytes, %d records)' % (time.perf_counter() - t, pool.size, len(recs)), flush=True)
if order == 'native-first':
    nat(); nat(); py(); nat()
else:
    py(); nat(); nat()
