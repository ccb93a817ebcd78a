"""GPU: the JavaScript drop-in API (N-API addon over libbbq) against the golden vectors, under node."""
import os
import shutil
import subprocess

import pytest

from bbqlib import ROOT


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_host_gpu_parity():
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "gpu_parity.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]
    assert "0 failures" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_host_gpu_parity_sharded_over_devices():
    """the same drop-in API with every index row-sharded over three shards (BBQ_DEVICES: all on GPU 0 here, one per GPU on a node):
    searchNearestNeighbors / quickSearch / computeBatchQuantizedScores / the rerank recipe must not change by a bit"""
    env = dict(os.environ, BBQ_DEVICES="0,0,0", BBQ_PILOT_ROWS="1024")
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "gpu_parity.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900, env=env)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]
    assert "0 failures" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_reference_suite():
    """the reference's own test expectations (recall thresholds on its closed-form datasets, batch == single, known answers)
    through the drop-in JavaScript API"""
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "reference_suite.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-4000:]
    assert "0 failures" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_quickstart_example():
    """examples/quickstart.js = the reference README's basic usage, unchanged except for the import"""
    r = subprocess.run(["node", os.path.join(ROOT, "examples", "quickstart.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_multi_gpu_and_multibit_example():
    """examples/multi_gpu_and_multibit.js with two shards on the one GPU"""
    r = subprocess.run(["node", os.path.join(ROOT, "examples", "multi_gpu_and_multibit.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300, env=dict(os.environ, BBQ_DEVICES="0,0"))
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]
    assert "shards: 2" in r.stdout and "batch agrees: true" in r.stdout and "bytes per row on the device: 68" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_bench_scale_matches_ctypes(tmp_path):
    """tests/js/bench_scale.js (what bench.py's `napi` leg runs at 10 M rows) at a small size: an index saved through ctypes is loaded by
    a node process through the N-API addon, searched with RAW fp32 queries through the reference-named API - batch and one call per
    query - and its answers equal the ctypes answers bit for bit"""
    import json
    import numpy as np
    import bench
    from bbqlib import bbq_amd as B
    n, dim, k, nq = 300_000, 768, 100, 24
    codes, corr = bench.synth_rows(1, 0, n, dim // 8)
    cen = bench.synth_centroid(dim)
    ix = B.Index(codes, corr, dim, float(B.centroid_dp(cen)))
    try:
        res = bench.napi_leg(B, ix, cen, dim, k, "COSINE", 1, 4, nq=nq)
    finally:
        ix.close()
    assert "error" not in res, res
    assert res["identical_to_ctypes"] is True and res["single_equals_batch"] is True
    assert res["value"] > 0 and res["p50_ms"] > 0
    json.dumps(res)
