"""bench.py's N > 1 code on ONE real GPU: two processes that share device 0 and talk over gloo (MIDAS_BENCH_ONE_GPU=1; RCCL refuses
two ranks on one device) -- the launch, the rank arithmetic, the timed region with its all-gather, the configs3_strong block and
the one JSON line are the ones the driver's `torch.distributed.run --nproc-per-node N bench.py --gpus N` goes through."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_of_bench_py_on_one_gpu():
    port = _free_port()
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = []
    for k in range(2):
        env = dict(base, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MIDAS_BENCH_ONE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "5",
                                       "--warmup", "2", "--sustain-seconds", "0", "--no-pmc"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=1500)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1].splitlines() if ln.startswith("{")]        # rank 0 prints the one line
    d = json.loads(lines[0])
    # N > 1: the line's own value is configs[3] -- BASELINE's multi-GPU configuration, ONE sample dealt to the ranks (strong) --
    # and the per-rank configs[2]-style replicas are a block of their own
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "strong" and d["unit"] == "sites/s"
    assert d["config"]["workload"].startswith("configs[3]") and d["config"]["sites_per_gpu"] == 200000000
    c3 = d["configs3_strong"]
    assert c3["ranks"] == 2 and c3["scaling"] == "strong" and c3["total_sites"] == 400000000 and c3["reads_counted_once"] is True
    assert len(c3["per_rank_ms_per_step"]) == 2 and c3["partition"]["weight_max_over_mean"] < 1.1
    assert d["value"] == c3["value"] and d["ms_per_step"] == c3["ms_per_step"] and c3["steps"] == 5
    # the whole job: every site of the sample, K times, over the slowest rank's time
    assert abs(d["value"] - 400000000 * 5 / (d["ms_per_step"] * 5 / 1e3)) / d["value"] < 1e-6
    assert 0.0 < d["roofline"]["frac"] < 1.0 and len(d["per_rank_ms_per_step"]) == 2
    w = d["weak_replicas"]
    assert w["scaling"] == "weak" and w["sites_per_gpu"] == 15000000 and "roofline" in w
    assert abs(w["value"] - 2 * 15000000 * 5 / (w["ms_per_step"] * 5 / 1e3)) / w["value"] < 1e-6
