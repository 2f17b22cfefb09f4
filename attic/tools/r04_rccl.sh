#!/bin/bash
# GPU box: the N > 1 step on one MI355X -- RCCL initialised, the summary all-gather inside the timed region, traced by rocprofv3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_rccl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --config c2 --force-collective --configs3 --steps 5 --warmup 2 --no-cpu --no-pmc --sustain-seconds 0.2 > $OUT/bench_c2_force_collective.log 2>&1
tail -1 $OUT/bench_c2_force_collective.log > $REPO/gpurun_out/r04_rccl_line.json
cd $REPO
python - <<'PY' > gpurun_out/r04_rccl_n1.txt
import sqlite3, glob, json
db = glob.glob("gpurun_out/prof_rccl/trace/**/*results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
print("== rocprofv3 --kernel-trace --stats -- python bench.py --config c2 --force-collective --configs3 --steps 5 --warmup 2 (one MI355X, WORLD_SIZE 1)")
print("%-100s %6s %12s %10s" % ("kernel", "calls", "total_us", "avg_us"))
for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
    print("%-100s %6d %12.3f %10.3f" % (name[:100], calls, total, avg))
try:
    line = json.loads(open("gpurun_out/r04_rccl_line.json").read())
    print()
    print("bench line (abridged):", json.dumps({k: line[k] for k in ("metric", "value", "n_gpus", "ms_per_step", "config", "configs3_strong") if k in line}))
except Exception as e:
    print("no JSON line:", e)
PY
cat gpurun_out/r04_rccl_n1.txt | cut -c1-200
