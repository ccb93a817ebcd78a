"""CPU, world_size 2 and 3 over gloo: the N>1 host path (exchange of packed candidate lists, shard-ordered replay,
streamed pipeline, dense fallback) of bbq_amd.distributed."""
import json
import os
import subprocess
import sys

import pytest

from bbqlib import ROOT

_PORT = [29610]


def _run(world, mode, tmp_path):
    out = str(tmp_path / ("out_%d_%s.json" % (world, mode)))
    _PORT[0] += 1
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_PORT[0]), os.path.join(ROOT, "tests", "dist_worker.py"), out, mode]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-4000:]
    return json.load(open(out))


@pytest.mark.parametrize("world,mode", [(2, "ties"), (3, "ties"), (2, "plain"), (2, "flag"), (4, "ties"), (4, "flag")])
def test_sharded_search_over_gloo(world, mode, tmp_path):
    res = _run(world, mode, tmp_path)
    assert res["ok"], res
    assert res["stream_ok"], res
    assert res["world"] == world
    if mode == "ties":
        assert res["ties"] == 1
        assert res["list_path_batches"] >= 1      # equal scores in the answer: the heap's history decides, the lists travel
    if mode == "plain":
        assert res["list_path_batches"] == 0      # tie-free data is answered from the shard-local answers alone
    if mode == "flag":
        assert res["list_path_batches"] >= 1
