#!/bin/bash
# GPU box (round 5): the direct kernel's floors (tallies off / loads only / no second pass), per-wave cycle attribution, instruction counters
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=$PWD/gpurun_out/r05_floor.txt
L=$PWD/midas_amd/lib/libmidas_snps_hip
( DIRECT_CHECK_PATHS=1 timeout 600 python tools/direct_check.py c3 20 2>&1 | tail -1 ) > $O
for V in dbg1 dbg4 dbg16; do ( echo "== variant $V"; MIDAS_SNPS_LIBRARY=${L}_$V.so timeout 300 python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O; done
( echo "== product"; timeout 300 python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
( echo "== probe"; MIDAS_SNPS_LIBRARY=${L}_probe.so timeout 300 python tools/probe_direct.py c3 2>&1 | head -9 ) >> $O
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/gpurun_out/prof_r05_insts; mkdir -p $P
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $P/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/direct_time.py c3 > $P/$N.log 2>&1
done
cd $GRAFT_REPO_ROOT && python tools/summarize_prof.py $P | grep -v "^JSON" | grep "pileup_direct\|ranges" >> $O
cat $O
