#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python bench.py --steps 20 > gpurun_out/r04_bench_c3.log 2>&1
grep "^{" gpurun_out/r04_bench_c3.log | tail -1 > gpurun_out/r04_bench_n1_c3.json
NCCL_DEBUG=INFO python bench.py --config c2 --force-collective --steps 5 --warmup 2 --no-cpu --no-pmc --sustain-seconds 0.2 > gpurun_out/r04_rccl_debug.log 2>&1
grep -i "nccl\|rccl" gpurun_out/r04_rccl_debug.log | head -40 > gpurun_out/r04_rccl_init.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_n1_c3.json"))
r = d["roofline"]
print("value %.3e sites/s  ms/step %.4f  frac %.3f  kernels %.4f ms (ranges %.4f + pileup %.4f)" % (d["value"], d["ms_per_step"], r["frac"], r["kernels_ms_avg"], r["index_pass_ms_avg"], r["pileup_kernel_ms_avg"]))
print("stream rates", r.get("stream_rates_GBps_this_box"), "traffic", r.get("traffic"), "x alg", r.get("traffic_over_algorithmic"))
print("calibration", r.get("traffic_this_box", {}).get("calibration"))
print("per kernel", json.dumps(r.get("traffic_this_box", {}).get("per_kernel")))
print("cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity_vs_oracle"), "speedup", d.get("speedup_vs_cpu_baseline"))
PY
cat gpurun_out/r04_rccl_init.txt | cut -c1-220 | head -20
