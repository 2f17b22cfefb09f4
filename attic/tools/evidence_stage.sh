#!/bin/bash
# GPU box: the stage-level evidence of a round -- the pileup stage end to end at configs[2] (host decode and whole decode on the
# device, alternating), configs[3] through the files with the command line, and the multi-rank product path as 1, 2 and 3
# processes on one GPU.  Text only, into gpurun_out/evidence/.   usage: tools/evidence_stage.sh [tag]
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
EV=$REPO/gpurun_out/evidence
mkdir -p $EV
cd $REPO
MIDAS_SNPS_TRACE=1 E2E_DEVICE_DECODE=1 E2E_REPS=2 timeout 900 python tools/e2e_stage.py c3 /tmp/e2e_c3 2>&1 | grep -v "rows on device\|write coded members\|write part" > $EV/${TAG}_e2e_stage_c3.txt
rm -rf /tmp/e2e_c3
timeout 1200 python tools/ranks_one_gpu.py c3 /tmp/midas_ranks > $EV/${TAG}_ranks_one_gpu.txt 2>&1
rm -rf /tmp/midas_ranks
timeout 1500 python tools/c4_files.py /tmp/midas_c4 > $EV/${TAG}_cli_stage_c4.txt 2>&1
rm -rf /tmp/midas_c4
ls -la $EV
