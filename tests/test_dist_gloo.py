"""The N > 1 path on CPU: world_size-2 gloo process group (the same code runs over RCCL/xGMI with backend nccl).
Covers the species -> rank assignment and the single all-gather of per-species summary rows."""
import os
import socket
import subprocess
import sys

import numpy as np

from midas_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from midas_amd import dist
rank, ws = dist.init_from_env("gloo")
assert ws == 2
weights = {"sp_%%02d" %% i: float((i * 7919) %% 13 + 1) for i in range(9)}
owner = dist.shard_species(weights, ws)
ids = sorted(weights)
rows = np.zeros((len(ids), 5), dtype=np.int64)
for i, sp in enumerate(ids):
    if owner[sp] == rank:
        rows[i] = [1000 + i, 10 * i, 100 * i, 7 * i + rank, i]
tot = dist.all_gather_summary(rows)
dist.barrier()
print(json.dumps({"rank": rank, "owner": owner, "tot": tot.tolist()}))
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_species_is_deterministic_and_balanced():
    w = {"a": 10.0, "b": 9.0, "c": 1.0, "d": 1.0, "e": 8.0}
    o1 = dist.shard_species(w, 2)
    o2 = dist.shard_species(dict(reversed(list(w.items()))), 2)
    assert o1 == o2
    load = [sum(w[s] for s in w if o1[s] == r) for r in range(2)]
    assert max(load) <= (4.0 / 3.0) * max(sum(w.values()) / 2.0, max(w.values()))   # the LPT guarantee
    assert dist.shard_species(w, 1) == {s: 0 for s in w}
    assert set(dist.shard_species({"x": 1.0}, 8).values()) == {0}


def test_all_gather_summary_single_process_is_identity():
    rows = np.arange(15, dtype=np.int64).reshape(3, 5)
    np.testing.assert_array_equal(dist.all_gather_summary(rows), rows)


def test_two_ranks_gloo_all_gather_of_summary_rows(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e
        outs.append(json.loads(o.strip().splitlines()[-1]))
    assert outs[0]["owner"] == outs[1]["owner"]                 # same assignment without talking
    assert set(outs[0]["owner"].values()) == {0, 1}
    assert outs[0]["tot"] == outs[1]["tot"]                     # every rank ends with every species' row
    ids = sorted(outs[0]["owner"])
    for i, sp in enumerate(ids):
        r = outs[0]["owner"][sp]
        assert outs[0]["tot"][i] == [1000 + i, 10 * i, 100 * i, 7 * i + r, i]
