#!/bin/bash
# GPU box (round 5): dynamic instruction counts of the device inflater's kernels on a configs[2] BAM
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_inflate_pmc
mkdir -p $OUT
W=/tmp/midas_decode_c3
cd $REPO && DECODE_ON_DEVICE=2 timeout 900 python tools/decode_trace.py c3 $W > /dev/null 2>&1      # (makes the BAM)
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  D=$OUT/pmc_$(echo $C | md5sum | cut -c1-6)
  DECODE_ON_DEVICE=2 timeout 600 rocprofv3 --kernel-trace --pmc $C -d $D -o pmc -- python $REPO/tools/decode_trace.py c3 $W > $D.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT | grep -v "^JSON" | grep "bgzf_decode\|bgzf_place\|bgzf_inflate" > $REPO/gpurun_out/r05_inflate_pmc.txt
cat $REPO/gpurun_out/r05_inflate_pmc.txt
