"""Developer script: the decode kernels' dispatches out of a rocprofv3 --kernel-trace database (rocpd sqlite), as text:
python tools/decode_trace_dump.py <trace_results.db>   (every column of the `kernels` view for kernels named bgzf_* / bam_*)"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("COLUMNS", "\t".join(cols))
for row in cur.execute("select * from kernels where name like '%bgzf%' or name like '%bam_%' order by 1"):
    print("\t".join(str(x) for x in row))
