#!/bin/bash
# GPU box: the round's evidence in one call -- the full -m gpu suite, smoke(), the bench lines (default = configs[2],
# configs[1], one rank of configs[3], the torchrun path on one GPU), and rocprofv3 stats + HBM-traffic PMC passes of the
# bench command for configs[2] and configs[1].  Text summaries only land in gpurun_out/evidence/ (the rocpd databases are
# tens of MiB each and are deleted here).   usage: tools/evidence.sh [tag]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
EV=$REPO/gpurun_out/evidence
mkdir -p $EV
cd $REPO
timeout 300 python tools/vram_prelude.py > $EV/vram_prelude.log 2>&1      # (a fresh box clears never-used VRAM at first allocation: not what the files below are about)
timeout 2400 python -m pytest tests -m gpu -x -q > $EV/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $EV/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $EV/smoke.log 2>&1
timeout 1200 python bench.py --steps 20 --warmup 5 > $EV/bench_n1_c3.json 2> $EV/bench_n1_c3.err
timeout 900 python bench.py --config c2 > $EV/bench_n1_c2.json 2> $EV/bench_n1_c2.err
timeout 900 python bench.py --config c4_rank --no-cpu > $EV/bench_n1_c4rank.json 2> $EV/bench_n1_c4rank.err
timeout 1500 python bench.py --config c4 --no-cpu --no-pmc --steps 20 > $EV/bench_n1_c4_whole.json 2> $EV/bench_n1_c4_whole.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu --no-pmc > $EV/bench_torchrun_n1.json 2> $EV/bench_torchrun_n1.err
prof() {   # prof <name> <bench args...>
  local name=$1; shift
  local out=$REPO/gpurun_out/prof_${TAG}_$name
  mkdir -p $out
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python $REPO/bench.py --no-cpu --no-pmc "$@" > $out/trace.log 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_1 -o pmc -- python $REPO/bench.py --no-cpu --no-pmc --steps 10 --warmup 2 --sustain-seconds 0 "$@" > $out/pmc_1.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_2 -o pmc -- python $REPO/bench.py --no-cpu --no-pmc --steps 10 --warmup 2 --sustain-seconds 0 "$@" > $out/pmc_2.log 2>&1 )
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-pmc $*   (then --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same command with --steps 10 --warmup 2 --sustain-seconds 0)"
    grep '^{' $out/trace.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('# bench line of the profiled run: value %.4g %s, ms_per_step %.4f, step kernels %.4f ms (ranges pass %.4f + pileup %.4f), frac %.3f' % (d['value'], d['unit'], d['ms_per_step'], r['kernels_ms_avg'], r['index_pass_ms_avg'], r['pileup_kernel_ms_avg'], r['frac']))" 2>/dev/null
    python tools/summarize_prof.py $out | grep -v "^JSON"; } > $EV/${TAG}_${name}_rocprofv3_stats_pmc.txt 2>&1
  rm -rf $out
}
prof c3
prof c2 --config c2
# the stage end to end at configs[1] (configs[2] takes three minutes of set-up: tools/e2e_stage.py c3, run by hand) and the
# row coder's kernel under rocprofv3
E2E_REPS=3 E2E_DEVICE_DECODE=1 timeout 900 python tools/e2e_stage.py c2 /tmp/e2e_c2 > $EV/e2e_stage_c2.txt 2>&1
MIDAS_SNPS_TRACE=1 E2E_REPS=2 E2E_DEVICE_DECODE=1 timeout 900 python tools/e2e_stage.py c3 /tmp/e2e_c3 2>&1 | grep -v "write rows\]\|rows on device\]\|write coded" > $EV/e2e_stage_c3.txt
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_rows -o rows -- python $REPO/tools/rows_probe.py c2 > $EV/rows_probe_c2.log 2>&1 )
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/rows_probe.py c2   (three device-coded and three host-coded writes of the 15 M rows of configs[1])"
  cat $EV/rows_probe_c2.log
  f=$(find $REPO/gpurun_out/prof_rows -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { echo "# kernel_stats.csv"; head -12 "$f"; }; } > $EV/${TAG}_rows_probe_c2.txt 2>&1
rm -rf $REPO/gpurun_out/prof_rows
# the merge kernel at configs[4]'s shape (50 samples), a soak of random shapes against the oracle on every path
bash tools/profile_merge.sh ${TAG}_merge 2000000 50 > /dev/null 2>&1; cp $REPO/gpurun_out/prof_${TAG}_merge/summary.txt $EV/${TAG}_merge_sites.txt
timeout 600 python tools/soak.py 300 5 > $EV/${TAG}_soak.txt 2>&1
ls -la $EV
