# GPU box (developer): scripts/run_midas.py snps --pileup on a configs[2] sample, twice, with the stage's phases (MIDAS_SNPS_TRACE) and the wall time; then the import times
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0,'.')
from midas_amd import synth
work='/tmp/midas_cli'
contigs, reads = synth.make_dataset(**synth.CONFIGS['c3'])
synth.write_sample(work+'/sample', work+'/db', contigs, reads)
PY
TIMEFORMAT="wall %R s"; for k in 1 2; do
  rm -rf /tmp/midas_cli/sample/snps/output
  time env MIDAS_SNPS_TRACE=1 python scripts/run_midas.py snps /tmp/midas_cli/sample --pileup -d /tmp/midas_cli/db -t 16 2>&1 | grep "stage\]\|wall\|bam device decode\] device"
done
rm -rf /tmp/midas_cli/sample/snps/output
python -X importtime scripts/run_midas.py snps /tmp/midas_cli/sample --pileup -d /tmp/midas_cli/db -t 16 2> /tmp/imp.txt > /dev/null; sort -t'|' -k2 -n /tmp/imp.txt | tail -12
