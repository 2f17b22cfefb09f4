#!/bin/bash
# GPU box: dynamic instruction counts of the direct pileup kernel (product build and the no-tally ablation)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_insts
mkdir -p $OUT
python $REPO/tools/direct_check.py c3 2 > /dev/null 2>&1     # (makes the dataset cache)
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"
for V in product; do
  L=$REPO/midas_amd/lib/libmidas_snps_hip.so; [ $V != product ] && L=$REPO/midas_amd/lib/libmidas_snps_hip_$V.so
  MIDAS_SNPS_LIBRARY=$L rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$V -o pmc -- python $REPO/tools/direct_time.py c3 > $OUT/$V.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT | grep -v "^JSON" | grep "pileup_direct\|ranges" > $REPO/gpurun_out/r04_pmc_insts.txt
cat $REPO/gpurun_out/r04_pmc_insts.txt
