"""throughput of ONE index row-sharded over several shards behind one handle (bbq_index_create_multi) next to the single-device
index, on the bench workload.  On a single-GPU box every shard sits on GPU 0, so the numbers price the sharded pipeline's overhead
(worker threads, packed lists over PCIe, host replay of every query), not a speed-up:  python scripts/time_multi_device.py [rows] [shards]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from bbqlib import bbq_amd as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dim, k, Q = 768, 100, 512
codes, corr = bench.synth_rows(1, 0, n, dim // 8)
qq, qc = bench.synth_queries(2, Q * 4, dim, 4)
ndev = B.device_count()
out = {"rows": n, "queries_per_call": Q}
for name, make in (("single", lambda: B.Index(codes, corr, dim, 0.01)),
                   ("multi_%d_shards" % shards, lambda: B.Index.create_multi(codes, corr, dim, 0.01, [i % ndev for i in range(shards)]))):
    ix = make()
    ix.set_option("replay_threads", 16)
    ix.set_option("pipeline_slots", 3)
    ix.search_batch(qq[:Q], qc[:Q], 4, 1, k)
    t0 = time.perf_counter()
    for i in range(1, 4):
        res = ix.search_batch(qq[i * Q:(i + 1) * Q], qc[i * Q:(i + 1) * Q], 4, 1, k)
    dt = (time.perf_counter() - t0) / 3
    by_round = {}
    if name != "single":    # rounds of a call are pipelined: round r + 1 is swept while round r is merged
        for rq in (128, 256, 512):
            ix.set_option("round_queries", rq)
            ix.search_batch(qq[:Q], qc[:Q], 4, 1, k)
            tr = time.perf_counter()
            for i in range(1, 4):
                ix.search_batch(qq[i * Q:(i + 1) * Q], qc[i * Q:(i + 1) * Q], 4, 1, k)
            by_round[rq] = round(Q / ((time.perf_counter() - tr) / 3))
    # the same three steps as ONE call: the multi-device handle pipelines its rounds (round r + 1 is swept while round r is merged)
    if name != "single":
        ix.set_option("round_queries", Q)
    tp = time.perf_counter()
    ix.search_batch(qq[Q:4 * Q], qc[Q:4 * Q], 4, 1, k)
    dtp = (time.perf_counter() - tp) / 3
    t1 = time.perf_counter()
    for i in range(20):
        ix.search(qq[i], qc[i], 4, 1, k)
    lat = (time.perf_counter() - t1) / 20
    out[name] = {"queries_per_s": round(Q / dt), "queries_per_s_three_steps_in_one_call": round(Q / dtp), "single_query_ms": round(lat * 1e3, 3),
                 "host_replays": ix.stats()["host_replays"], "queries_per_s_by_round_queries": by_round}
    ix.close()
print(out)
