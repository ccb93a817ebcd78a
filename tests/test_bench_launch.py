"""bench.py --gpus N starts N real ranks by itself (fresh child processes before any GPU call) or fails - it never prints
a line that claims N GPUs while measuring fewer.  CPU: the launch path with --dry-run (no device, no
measurement).  GPU: the same path with real work, two ranks sharing the box's one GPU over gloo."""
import json
import os
import subprocess
import sys

import pytest

from bbqlib import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, BENCH] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=e)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return r, lines


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_starts_n_ranks(n):
    r, lines = _run(["--gpus", str(n), "--dry-run"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, lines                       # ONE JSON line on stdout, whatever the ranks chatter about
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks"] == n and out["dry_run"] is True and out["value"] is None


def test_single_gpu_needs_no_launcher():
    r, lines = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0 and json.loads(lines[0])["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    r, lines = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and not lines
    assert "refusing" in r.stderr


def test_too_few_devices_is_an_error():
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("64 devices visible")
    r, lines = _run(["--gpus", "64", "--no-recall"])
    assert r.returncode != 0 and not lines
    assert "HIP device(s) visible" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_through_the_self_launch_path():
    r, lines = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--rows", "400000", "--steps", "2", "--warmup", "1",
                     "--batch", "64", "--no-recall", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo"
    assert out["parity_full_size"] is True
    assert out["value"] > 0
