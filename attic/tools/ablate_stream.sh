#!/bin/bash
# GPU box: the streaming pileup kernel against the barrier-phased one and its ablation variants (tools/build_variant.sh
# <name> -DMIDAS_STREAM_ABLATE=<bits> / -DMIDAS_SNPS_PHASED_KERNEL), alternating on the same box.
# usage: tools/ablate_stream.sh "<variants>" "<configs>"
run() { python bench.py --config $2 --no-cpu --steps 50 --warmup 3 --pack-steps 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 kernel %.1f us step %.1f us' % (d['roofline']['kernel_ms_avg']*1e3, d['ms_per_step']*1e3))"; }
for cfg in ${2:-c2 c3}; do
  run product $cfg
  for v in ${1:-phased}; do MIDAS_SNPS_LIBRARY=$GRAFT_REPO_ROOT/midas_amd/lib/libmidas_snps_hip_$v.so run $v $cfg; done
done
