#!/usr/bin/env python3
"""bench.py - queries/sec of the binary-quantized scan + exact top-k on MI355X.

  python bench.py --gpus 1 --steps K --warmup W          (single process)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): 10M x 768-dim 1-bit index, queryBits=4, k=100, COSINE.  One "step" = one batch
of --batch independent queries, each doing its OWN sweep of the index (no sweep sharing; B=1 per sweep).  With N
GPUs the same 10M-row index is row-sharded (strong scaling): every rank sweeps its shard, candidate lists are
gathered over RCCL and rank 0 replays the reference heap.  Inputs are synthetic and resident in HBM before the
timed region: the quantized index is synthesised directly (scan time is value-independent); a smaller
real fp32 -> quantize -> search run provides recall@100 against an fp32 brute force.

Prints ONE JSON line (rank 0) with the driver's contract plus "roofline" and "cpu_baseline".
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "better-binary-quantization_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
CHUNK = 65536          # rows per deterministic generation chunk


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_POP = np.array([bin(i).count("1") for i in range(256)], np.uint8)


def synth_rows(seed, row0, row1, pb):
    """quantized rows [row0,row1) of the synthetic index: uniform random bits, corrections around the values a
    real 768-d COSINE index shows, quantizedComponentSum = popcount as a real 1-bit index has"""
    codes = np.empty((row1 - row0, pb), np.uint8)
    corr = np.empty((row1 - row0, 4), np.float64)
    r = row0
    while r < row1:
        c0 = r // CHUNK
        lo, hi = c0 * CHUNK, (c0 + 1) * CHUNK
        rng = np.random.default_rng([seed, c0])
        cc = rng.integers(0, 256, size=(CHUNK, pb), dtype=np.uint8)
        u = rng.random((CHUNK, 3))
        a, b = max(r, lo), min(row1, hi)
        codes[a - row0:b - row0] = cc[a - lo:b - lo]
        corr[a - row0:b - row0, 0] = -0.04 * (0.9 + 0.2 * u[a - lo:b - lo, 0])
        corr[a - row0:b - row0, 1] = 0.04 * (0.9 + 0.2 * u[a - lo:b - lo, 1])
        corr[a - row0:b - row0, 2] = 1e-4 * (2 * u[a - lo:b - lo, 2] - 1)
        corr[a - row0:b - row0, 3] = _POP[cc[a - lo:b - lo]].sum(axis=1, dtype=np.int64)
        r = b
    return codes, corr


def synth_rows_multibit(seed, row0, row1, dim, ib):
    """rows [row0,row1) of a synthetic multi-bit index as quantizeVectors hands them over for indexBits > 1: one byte per dimension,
    uniform values < 2^ib (src/binaryQuantizationFormat.ts:241-245), quantizedComponentSum = the sum of the row's codes"""
    codes = np.empty((row1 - row0, dim), np.uint8)
    corr = np.empty((row1 - row0, 4), np.float64)
    r = row0
    while r < row1:
        c0 = r // CHUNK
        lo, hi = c0 * CHUNK, (c0 + 1) * CHUNK
        rng = np.random.default_rng([seed, c0, ib])
        cc = rng.integers(0, 1 << ib, size=(CHUNK, dim), dtype=np.uint8)
        u = rng.random((CHUNK, 3))
        a, b = max(r, lo), min(row1, hi)
        codes[a - row0:b - row0] = cc[a - lo:b - lo]
        corr[a - row0:b - row0, 0] = -0.04 * (0.9 + 0.2 * u[a - lo:b - lo, 0])
        corr[a - row0:b - row0, 1] = 0.04 * (0.9 + 0.2 * u[a - lo:b - lo, 1])
        corr[a - row0:b - row0, 2] = 1e-4 * (2 * u[a - lo:b - lo, 2] - 1)
        corr[a - row0:b - row0, 3] = cc[a - lo:b - lo].sum(axis=1, dtype=np.int64)
        r = b
    return codes, corr


def synth_queries(seed, nq, dim, qb=4):
    rng = np.random.default_rng([seed, 777])
    qq = rng.integers(0, 1 << qb, size=(nq, dim), dtype=np.uint8)
    qc = np.empty((nq, 4), np.float64)
    qc[:, 0] = -0.15 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 1] = 0.148 * (0.9 + 0.2 * rng.random(nq))
    qc[:, 2] = -0.0028 * rng.random(nq)
    qc[:, 3] = qq.sum(axis=1)
    return qq, qc


def recall_probe(B, device, n=1_000_000, dim=768, nq=32, k=100):
    """real fp32 vectors -> quantizeVectors on the device -> GPU search, recall@k against an fp32 brute force (torch, same GPU).
    SURVEY 8(d) asks for a 1 M x 768 fp32 run; the default is the headline size, 10 M x 768 (30.7 GB of fp32).
    The vectors are generated ON THE DEVICE block by block (seeded per block, so the brute force regenerates them instead of
    reading them back); the host copy exists only because bbq_index_build / bbq_vectors_create take host arrays, as the
    reference's API does."""
    import torch
    dev = "cuda:%d" % device
    BLOCK = 500_000
    # embedding-like data: 64 latent factors + isotropic noise (uniform random vectors have no neighbour structure to recall,
    # tight clusters larger than k make every member an equally good neighbour).  Neighbour distances form a continuum.
    g0 = torch.Generator(device=dev)
    g0.manual_seed(99)
    W = torch.randn(64, dim, generator=g0, device=dev)
    queries_d = torch.randn(nq, 64, generator=g0, device=dev) @ W + 0.8 * torch.randn(nq, dim, generator=g0, device=dev)

    def block(i0):
        m = min(BLOCK, n - i0)
        g = torch.Generator(device=dev)
        g.manual_seed(1000 + i0)
        return torch.randn(m, 64, generator=g, device=dev) @ W + 0.8 * torch.randn(m, dim, generator=g, device=dev)

    t_gen = time.perf_counter()
    base = np.empty((n, dim), np.float32)
    for i in range(0, n, BLOCK):
        b = block(i)
        base[i:i + b.shape[0]] = b.cpu().numpy()
        del b
    queries = queries_d.cpu().numpy()
    t_gen = time.perf_counter() - t_gen
    sim = 1
    t_build = time.perf_counter()
    ix, _, _, cen = B.Index.build(base, sim, device=device, want_host_copy=False)
    t_build = time.perf_counter() - t_build
    qq, qc = B.quantize_queries(queries, cen, sim, 4)
    idx, sc, cnt = ix.search_batch(qq, qc, 4, sim, k)
    # the reference's recall recipe (src/topKSelector.ts:29-79): oversample x3, exact cosine rerank on the device
    dv = B.Vectors(base, device)
    del base
    B.search_rerank_batch(ix, dv, queries, qq, qc, 4, sim, k, 3, 0, 1)
    t0 = time.perf_counter()
    ridx, _, _, _ = B.search_rerank_batch(ix, dv, queries, qq, qc, 4, sim, k, 3, 0, 1)
    rerank_ms = (time.perf_counter() - t0) * 1e3 / nq
    dv.close()
    ix.close()
    # fp32 brute force over the regenerated blocks
    tq = queries_d / queries_d.norm(dim=1, keepdim=True)
    best_s = torch.full((nq, k), -2.0, device=dev)
    best_i = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    for i in range(0, n, BLOCK):
        tb = block(i)
        tb = tb / tb.norm(dim=1, keepdim=True)
        s_blk, i_blk = (tq @ tb.T).topk(min(k, tb.shape[0]), dim=1)
        cat_s, cat_i = torch.cat([best_s, s_blk], 1), torch.cat([best_i, i_blk + i], 1)
        best_s, pick = cat_s.topk(k, dim=1)
        best_i = torch.gather(cat_i, 1, pick)
        del tb
    truth = best_i.cpu().numpy()
    rec = np.mean([len(set(truth[i].tolist()) & set(idx[i].tolist())) / float(k) for i in range(nq)])
    rec3 = np.mean([len(set(truth[i].tolist()) & set(ridx[i].tolist())) / float(k) for i in range(nq)])
    return float(rec), {"n": n, "dim": dim, "queries": nq, "data": "64 latent gaussian factors x random 64x768 map + 0.8 sigma isotropic noise (generated on the device)",
                        "recall_at_100_oversample3_rerank": float(rec3), "oversample3_rerank_ms_per_query": round(rerank_ms, 3),
                        "generate_s": round(t_gen, 1), "build_s": round(t_build, 1)}


def oracle_scores(O, codes, corr, dim, qq, qc, qb, sim, cdp, ib):
    """the checker's f32 scores of every row: the pinned restatement of the reference where the reference answers, the documented
    extension (per-row 4-bit form over computeQuantizedDotProduct) where it throws (multi-bit index, queryBits other than 1 / 4)"""
    if ib == 1:
        return O.score_all(codes, corr, dim, qq, qc, qb, sim, cdp)[2]
    if qb in (1, 4):
        return O.score_all(codes, corr, dim, qq, qc, qb, sim, cdp, ib)[2]
    return O.score_all_multibit_ext(codes, corr, dim, qq, qc, qb, sim, cdp)[2]


def host_cpu():
    """model name and logical cores of the box the CPU baseline runs on"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(dim, k, codes, corr, qq, qc, cdp, qb, sim, ib=1, budget_s=16.0):
    """the oracle (CPU restatement of the reference loops, 1 thread) over the WHOLE index of the timed workload - no
    extrapolation over rows - for as many of the timed queries as fit the budget (at least one)"""
    import orclib as O
    n = codes.shape[0]
    c, r = np.ascontiguousarray(codes), np.ascontiguousarray(corr)
    t0 = time.perf_counter()
    done = 0
    while True:
        s32 = oracle_scores(O, c, r, dim, qq[done % len(qq)], qc[done % len(qq)], qb, sim, cdp, ib)
        O.heap_topk(s32, k)
        done += 1
        if time.perf_counter() - t0 > budget_s or done >= 8:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, n, dt


def synth_centroid(dim, cdp_target=0.0009110655808639536):
    """a centroid for the synthetic index (the scan only needs centroid . centroid; raw-query legs quantize against the vector itself):
    seeded gaussian scaled to the squared magnitude a real 768-d COSINE index shows"""
    rng = np.random.default_rng([7, dim])
    c = rng.standard_normal(dim)
    return (c * np.sqrt(cdp_target / float(c @ c))).astype(np.float32)


def synth_raw_queries(seed, nq, dim):
    """raw fp32 queries (what the reference's searchNearestNeighbors takes): unit-scale gaussians"""
    return np.random.default_rng([seed, 4242]).standard_normal((nq, dim)).astype(np.float32)


def time_steps(ix, batches, QB, SIM, k):
    """search_batch over the given batches: (seconds, results)"""
    res = []
    t0 = time.perf_counter()
    for qq, qc in batches:
        res.append(ix.search_batch(qq, qc, QB, SIM, k))
    return time.perf_counter() - t0, res


# Part of every sweep is served by the 256 MiB Infinity Cache (library option resident_mb, bbq_stats.resident_bytes): the default run's
# rate of ALGORITHMIC bytes is therefore not an HBM rate.  What the roofline object calls `frac` is the strict figure: the same step with
# resident_mb = 0 (nothing kept in the cache, every byte of every sweep streamed from HBM).
RESIDENT_NOTE = ("frac / achieved = the dominant launch with resident_mb 0: nothing is kept in the Infinity Cache, every byte of every sweep comes "
                 "from HBM (the HBM roofline in its strict sense).  frac_algorithmic / achieved_algorithmic = the default run: "
                 "cache_resident_frac_of_sweep of the bytes a query sweeps stay in the 256 MiB Infinity Cache between the sweeps of successive "
                 "queries, so the algorithmic bytes per second exceed what HBM delivered; it is NOT an HBM fraction.  `traffic` (FETCH_SIZE x2 "
                 "+ WRITE_SIZE) counts the bytes that left the L2s; the Infinity Cache sits behind that counter")


def flush_infinity_cache(torch):
    """the strict runs stream with non-temporal loads, which neither allocate in the 256 MiB Infinity Cache nor displace what is there:
    lines the default mode left resident would go on serving hits.  Two passes of ordinary writes and reads over 1 GiB evict them."""
    buf = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
    for _ in range(2):
        buf.fill_(1.0)
        float(buf[::4096].sum().item())
        torch.cuda.synchronize()
    del buf


def roofline_object(strict_gbps, algorithmic_gbps, resident_frac, extra):
    """the bench line's roofline object.  `frac` is ALWAYS an HBM fraction in the strict sense (resident_mb 0) or null when that leg was
    skipped; the default run's cache-assisted rate stands beside it under names that do not claim HBM.  An index that lives in the
    Infinity Cache (resident fraction >= 0.9) is labelled as bound by the cache in its default mode."""
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "achieved": strict_gbps, "frac": (strict_gbps / HBM_PEAK_GBS) if strict_gbps else None,
         "frac_is": "strict HBM: dominant launch with nothing kept in the Infinity Cache (resident_mb 0)" if strict_gbps else "not measured (--no-hbm-only)",
         "achieved_algorithmic": algorithmic_gbps, "frac_algorithmic": algorithmic_gbps / HBM_PEAK_GBS,
         "default_mode_bound": "infinity_cache" if resident_frac >= 0.9 else "hbm + infinity_cache (%.0f %% of a sweep's bytes cache-resident)" % (100 * resident_frac),
         "cache_resident_frac_of_sweep": resident_frac, "note": RESIDENT_NOTE}
    r.update(extra)
    return r


def config_leg(B, torch, name, N, dim, k, QB, IB, sim_name, device, steps, warmup, Q, slots, replay_threads, parity=True):
    """one of the other BASELINE configs as a short leg of the default run: the same step as the headline (Q independent queries per
    step, each sweeping the index on its own), the dominant launch priced by HIP events, one full-size query held to the oracle"""
    import orclib as O
    SIM = {"EUCLIDEAN": 0, "COSINE": 1, "MAXIMUM_INNER_PRODUCT": 2}[sim_name]
    pb = (dim + 7) // 8
    t0 = time.perf_counter()
    codes, corr = synth_rows(1, 0, N, pb) if IB == 1 else synth_rows_multibit(1, 0, N, dim, IB)
    cdp = float(B.centroid_dp(synth_centroid(dim)))
    ix = B.Index(codes, corr, dim, cdp, device=device, index_bits=IB)
    ix.set_option("pipeline_slots", slots)
    ix.set_option("replay_threads", replay_threads)
    build_s = time.perf_counter() - t0
    bpr = ix.bytes_per_row
    qq_all, qc_all = synth_queries(2, (warmup + steps) * Q, dim, QB)
    batches = [(qq_all[i * Q:(i + 1) * Q], qc_all[i * Q:(i + 1) * Q]) for i in range(warmup + steps)]
    time_steps(ix, batches[:warmup], QB, SIM, k)
    ix.reset_stats()
    torch.cuda.synchronize()
    dt, res = time_steps(ix, batches[warmup:], QB, SIM, k)
    torch.cuda.synchronize()
    st = ix.stats()
    launch_bytes = st["total_scan_bytes"] / max(st["total_scan_launches"], 1)
    launch_ms = st["total_scan_ms"] / max(st["total_scan_launches"], 1)
    achieved = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    qps = steps * Q / dt
    # the same step with nothing kept in the Infinity Cache (resident_mb 0): every byte of every sweep from HBM - the HBM roofline's figure
    ix.set_option("resident_mb", 0)
    strict_steps = min(steps, max(4, steps // 2))
    flush_infinity_cache(torch)
    time_steps(ix, batches[:2], QB, SIM, k)
    ix.reset_stats()
    torch.cuda.synchronize()
    dts, res_s = time_steps(ix, batches[warmup:warmup + strict_steps], QB, SIM, k)
    torch.cuda.synchronize()
    sts = ix.stats()
    ix.set_option("resident_mb", -1)
    lbs = sts["total_scan_bytes"] / max(sts["total_scan_launches"], 1)
    lms = sts["total_scan_ms"] / max(sts["total_scan_launches"], 1)
    strict = lbs / (lms * 1e-3) / 1e9 if lms > 0 else 0.0
    out = {"workload": "%dx%d-dim %d-bit index, queryBits=%d, k=%d, %s" % (N, dim, IB, QB, k, sim_name), "value": qps, "unit": "queries/s",
           "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "queries_per_step": Q, "bytes_per_row": bpr,
           "roofline": roofline_object(strict, achieved, st["resident_bytes"] / float(N * bpr),
                                       {"bytes_per_launch": launch_bytes, "avg_launch_ms": launch_ms, "launches_timed": st["total_scan_launches"],
                                        "strict_avg_launch_ms": lms, "strict_launches_timed": sts["total_scan_launches"],
                                        "how": "HIP events around the largest-segment launch on its own stream",
                                        "cache_resident_bytes_per_sweep": st["resident_bytes"]}),
           "end_to_end_hbm_frac_strict": (strict_steps * Q / dts) * N * bpr / 1e9 / HBM_PEAK_GBS,
           "end_to_end_frac_algorithmic": qps * N * bpr / 1e9 / HBM_PEAK_GBS,
           "strict_identical_to_default": bool((res_s[0][0] == res[0][0]).all() and (res_s[0][1].view(np.uint32) == res[0][1].view(np.uint32)).all()),
           "host_replays": st["host_replays"], "dense_fallbacks": st["dense_fallbacks"], "build_s": round(build_s, 1)}
    if parity:
        s32 = oracle_scores(O, codes, corr, dim, qq_all[warmup * Q], qc_all[warmup * Q], QB, SIM, cdp, IB)
        oi, osc = O.heap_topk(s32, k)
        gi, gs, _ = res[0]
        out["parity_full_size"] = bool((gi[0] == oi).all() and (gs[0].view(np.uint32) == osc.view(np.uint32)).all())
    if IB != 1 and QB not in (1, 4):
        out["parity_note"] = "the reference throws for queryBits=%d on an indexBits=%d index: integer dot pinned by fixtures, float score parity unpinned" % (QB, IB)
    ix.close()
    return out


def napi_leg(B, ix, centroid, dim, k, sim_name, SIM, QB, nq=512):
    """the drop-in boundary at scale: the timed index is saved, a node process loads it through bbq_napi.node and runs the reference's
    call shapes with RAW fp32 queries (tests/js/bench_scale.js); its answers must equal this process's ctypes answers bit for bit"""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("node") is None:
        return {"error": "node not installed"}
    addon = os.path.join(ROOT, "better-binary-quantization_amd", "lib", "bbq_napi.node")
    if not os.path.exists(addon):
        return {"error": "bbq_napi.node is not built"}
    tmp = tempfile.mkdtemp(prefix="bbq_napi_")
    try:
        prefix = os.path.join(tmp, "index")
        ix.save(prefix, centroid, SIM)
        raw = synth_raw_queries(5, nq, dim)
        raw.tofile(os.path.join(tmp, "queries.f32"))
        qq, qc = B.quantize_queries(raw, centroid, SIM, QB)
        gi, gs, gc = ix.search_batch(qq, qc, QB, SIM, k)
        node_cmd = ["node"] + (["--trace-gc"] if os.environ.get("BBQ_BENCH_TRACE_GC") else [])
        r = subprocess.run(node_cmd + [os.path.join(ROOT, "tests", "js", "bench_scale.js"), prefix, os.path.join(tmp, "queries.f32"), str(dim), str(k),
                            sim_name, os.path.join(tmp, "answers.bin"), "10", "200"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": ("node exited with %d: " % r.returncode) + r.stderr[-300:]}
        js = json.loads(r.stdout.strip().splitlines()[-1])
        if os.environ.get("BBQ_BENCH_TRACE_GC"):
            log("node --trace-gc:\n" + "\n".join(r.stdout.strip().splitlines()[-40:-1]))
        blob = np.fromfile(os.path.join(tmp, "answers.bin"), np.uint8)
        ji = blob[:nq * k * 4].view(np.int32).reshape(nq, k)
        jsx = blob[nq * k * 4:nq * k * 8].view(np.uint32).reshape(nq, k)
        jc = blob[nq * k * 8:].view(np.int32)
        same = bool((jc == gc).all() and (ji == gi).all() and (jsx == gs.view(np.uint32)).all())
        return {"value": js["batch_queries_per_s"], "unit": "queries/s", "call": "format.searchNearestNeighborsBatch(raw fp32 queries, values, k) through bbq_napi.node, %d queries per call" % nq,
                "batch_ms_per_call": js["batch_ms_per_call"], "batch_ms_per_call_inside_addon": js.get("batch_ms_per_call_inside_addon"), "p50_ms": js["single_p50_ms"], "p99_ms": js["single_p99_ms"], "min_ms": js["single_min_ms"],
                "single_call": "format.searchNearestNeighbors(raw fp32 query, values, k): normalise + quantize + sweep + top-k per call (src/binaryQuantizationFormat.ts:308-412)",
                "single_queries_per_s": js["single_queries_per_s"], "single_equals_batch": js["single_equals_batch"], "identical_to_ctypes": same,
                "single_slowest_calls": js.get("single_slowest_calls"),
                "index": "loaded by the node process from the .veb/.vemb pair this process saved (%d rows)" % js["rows"], "load_ms": js["load_ms"], "node": js["node"]}
    except Exception as e:  # informational leg: never fail the bench for it
        return {"error": str(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def c1_leg():
    """BASELINE config 1 (the reference's README bench: quickSearch over 1000 x 128-d, k = 10, COSINE; /root/reference/README.md:115-116 quotes
    "~2 ms" per quickSearch on its CPU path): the same calls through the JavaScript drop-in API on this box (tests/js/bench_c1.js) - ms per
    quickSearch (re-quantizes and re-uploads the 1000 targets every call, as src/index.ts:95-111 does) and per search on a pre-built
    index; the top 10 must be the reference's."""
    import shutil
    import subprocess
    if shutil.which("node") is None:
        return {"error": "node not installed"}
    if not os.path.exists(os.path.join(ROOT, "better-binary-quantization_amd", "lib", "bbq_napi.node")):
        return {"error": "bbq_napi.node is not built"}
    try:
        r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "bench_c1.js")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": ("node exited with %d: " % r.returncode) + r.stderr[-300:]}
        js = json.loads(r.stdout.strip().splitlines()[-1])
        js["reference_readme"] = "~2 ms per quickSearch, pure-TS CPU path (README.md:115-116); 25.7 ms / 0.53 ms measured for quickSearch / pre-built search in SURVEY 3.5"
        js["what"] = "not a roofline case: 1000 rows are one launch chain; the call is bound by the host (quantizing 1000 targets per quickSearch) and launch latency"
        return js
    except Exception as e:  # informational leg
        return {"error": str(e)[:300]}


def shard_shape_leg(args, headline_qps):
    """The 8-GPU shard shape on ONE GPU, driver-timed, in two child processes (the process group and the sharded code path must not leak
    into this one), each a rank of 8 of the headline index alone in an RCCL group of ONE rank, 2048 queries per batch:
      rank 0 (rows [0, 1.25 M), no pilot replica) through ShardedSearcher: every per-rank cost of the --gpus 8 step - sweep, packing,
             exchange (with itself), merge, answers, list path - and the answers checked against a plain index over the same rows;
      rank 1 (rows [1.25 M, 2.5 M) + the 32 K-row pilot replica 7 of the 8 ranks carry): its sweep alone (bbq_shard_scan_begin / _wait).
             Its merge cannot be rehearsed alone: the replica's rows belong to rank 0."""
    import subprocess

    def child(shard_of, port):
        cmd = [sys.executable, os.path.abspath(__file__), "--force-dist", "--shard-of", shard_of, "--batch", "2048", "--steps", str(max(args.steps // 4, 4)),
               "--warmup", "2", "--rows", str(args.rows), "--dim", str(args.dim), "--k", str(args.k), "--no-recall", "--no-cpu-baseline", "--no-parity",
               "--latency-calls", "0", "--shared-sweep", "0", "--no-configs", "--no-napi", "--no-raw", "--no-hbm-only", "--inprocess-shards", "0",
               "--slots", str(args.slots), "--replay-threads", str(args.replay_threads)]
        env = dict(os.environ, MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
        if r.returncode != 0:
            raise RuntimeError(("child %s exited with %d: " % (shard_of, r.returncode)) + r.stderr[-400:])
        return json.loads(r.stdout.strip().splitlines()[-1])

    try:
        c0 = child("0/8", 29571)
        c1 = child("1/8", 29572)
        ph = (c0.get("sharded_phases") or {}).get("rank0")
        scan1 = c1.get("piloted_scan") or {}
        per_rank = min(c0["value"], scan1.get("queries_per_s") or c0["value"])
        return {"value_per_rank": per_rank, "unit": "queries/s", "queries_per_batch": 2048,
                "rank0_of_8": {"value": c0["value"], "ms_per_batch": c0["ms_per_step"], "rows": c0["config"].get("shard_rows"), "pilot_rows": 0,
                               "phases_ms_per_batch": ph, "replayed_queries_per_batch": c0.get("replayed_queries_per_batch"),
                               "identical_to_direct": c0.get("identical_to_direct"),
                               "frac_algorithmic_dominant_launch": c0["roofline"].get("frac_algorithmic")},
                "rank1_of_8_sweep_only": {"value": scan1.get("queries_per_s"), "ms_per_batch": scan1.get("ms_per_batch"),
                                          "rows": c1["config"].get("shard_rows"), "pilot_rows": c1["config"].get("pilot_rows")},
                "identical_to_direct": c0.get("identical_to_direct"),
                "projected_8gpu": per_rank, "projected_8gpu_vs_n1_headline": per_rank / headline_qps if headline_qps else None,
                "what": "at --gpus 8 every rank does this per batch of 2048 queries while the others do the same, so the job's rate is the slowest "
                        "rank's rate as long as the exchange (k + 3 words per query and pair of ranks over xGMI) stays hidden behind the next "
                        "batch's sweep; value_per_rank = min(rank 0 through the whole searcher, rank 1's sweep with its pilot replica)"}
    except Exception as e:  # informational leg
        return {"error": str(e)[:500]}


def inprocess_only(args):
    """child process of an N-GPU run (rank 0 starts it after every rank has closed its index): the SAME synthetic index behind ONE
    handle row-sharded over the N devices of this process (bbq_index_create_multi - what the TypeScript host uses), the same step.
    A process of its own so that nothing it does can take the ranks' line down.  Prints one JSON object."""
    import bbq_amd as B
    N, dim, k, Q = args.rows, args.dim, args.k, args.batch if args.batch > 0 else min(2048, max(512, 256 * args.gpus))
    SIM = {"EUCLIDEAN": 0, "COSINE": 1, "MAXIMUM_INNER_PRODUCT": 2}[args.sim]
    QB, IB = args.query_bits, args.index_bits
    pb = (dim + 7) // 8
    codes, corr = synth_rows(1, 0, N, pb) if IB == 1 else synth_rows_multibit(1, 0, N, dim, IB)
    cdp = float(B.centroid_dp(synth_centroid(dim)))
    devices = [0] * args.gpus if args.same_device else list(range(args.gpus))
    mx = B.Index.create_multi(codes, corr, dim, cdp, devices, index_bits=IB, pilot_rows=args.pilot)
    mx.set_option("pipeline_slots", args.slots)
    mx.set_option("replay_threads", args.replay_threads)
    mx.set_option("round_queries", Q)
    n_steps = args.warmup + args.steps
    qq_all, qc_all = synth_queries(2, n_steps * Q, dim, QB)
    batches = [(qq_all[i * Q:(i + 1) * Q], qc_all[i * Q:(i + 1) * Q]) for i in range(n_steps)]
    # all steps in ONE call: the handle pipelines its rounds (round r + 1 is on the devices while round r is merged)
    wq = np.concatenate([b[0] for b in batches[:args.warmup]]) if args.warmup else None
    if wq is not None:
        mx.search_batch(wq, np.concatenate([b[1] for b in batches[:args.warmup]]), QB, SIM, k)
    mx.reset_stats()
    tq, tc = np.concatenate([b[0] for b in batches[args.warmup:]]), np.concatenate([b[1] for b in batches[args.warmup:]])
    t0 = time.perf_counter()
    res = mx.search_batch(tq, tc, QB, SIM, k)
    dt = time.perf_counter() - t0
    st = mx.stats()
    out = {"value": args.steps * Q / dt, "unit": "queries/s", "shards": mx.shards, "devices": devices, "queries_per_round": Q, "steps": args.steps,
           "ms_per_step": dt / args.steps * 1e3, "host_replays": st["host_replays"], "dense_fallbacks": st["dense_fallbacks"],
           "aggregate_dominant_GBps": st["total_scan_bytes"] / (st["total_scan_ms"] * 1e-3) / 1e9 if st["total_scan_ms"] > 0 else None,
           "path": "bbq_index_create_multi: one handle, one process, a host thread per shard, answers merged on the calling thread"}
    if not args.no_parity:
        import orclib as O
        s32 = oracle_scores(O, codes, corr, dim, tq[0], tc[0], QB, SIM, cdp, IB)
        oi, osc = O.heap_topk(s32, k)
        out["parity_full_size"] = bool((res[0][0] == oi).all() and (res[1][0].view(np.uint32) == osc.view(np.uint32)).all())
    mx.close()
    print(json.dumps(out), flush=True)
    return 0


def run_inprocess_child(args):
    """rank 0 of an N-GPU run: the in-process multi-device leg in a fresh process (see inprocess_only)"""
    import subprocess
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                   "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "HIP_VISIBLE_DEVICES",
                                                                   "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    for k_ in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):   # a launcher that narrows devices per rank would hide the others
        if k_ in os.environ and len(os.environ[k_].split(",")) >= args.gpus:
            env[k_] = os.environ[k_]
    cmd = [sys.executable, os.path.abspath(__file__), "--inprocess-only"] + [a for a in sys.argv[1:] if a != "--force-dist"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
        if r.returncode != 0:
            return {"error": "child exited with %d: %s" % (r.returncode, r.stderr[-300:])}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": str(e)[:300]}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks (one per GPU) with torch.distributed.run
    as a FRESH child process - before this process has made any GPU/HIP call - relay rank 0's JSON line (the children inherit
    stdout) and return the children's exit code.  Returns None when this process is itself a rank (or N == 1).
    Never prints a line that claims N GPUs while measuring fewer: a launcher/flag mismatch or too few devices is an error."""
    import subprocess
    in_group = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if in_group:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            log("bench.py: launched with WORLD_SIZE=%d but --gpus %d: refusing to print a mislabelled line" % (world, args.gpus))
            return 2
        return None
    if args.gpus < 1:
        log("bench.py: --gpus must be >= 1")
        return 2
    if args.gpus == 1:
        return None
    if not (args.dry_run or args.same_device):
        import torch  # importing torch and counting devices does not initialise the GPU runtime
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            log("bench.py: --gpus %d but only %d HIP device(s) visible" % (args.gpus, ndev))
            return 3
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")   # RCCL's kernels must get in between sweeps that keep every CU busy
    env.setdefault("OMP_NUM_THREADS", "1")
    # the ranks' stdout is filtered: the JSON line goes to stdout, library chatter (gloo prints its connection report there) to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def dry_run(args):
    """the launch path and the group plumbing of a run, with no device and no measurement"""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
        t = torch.tensor([0.001 * (1 + dist.get_rank())], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        ranks, backend, rank = dist.get_world_size(), dist.get_backend(), dist.get_rank()
        dist.destroy_process_group()
    else:
        ranks, backend, rank = 1, None, 0
    if rank == 0:
        print(json.dumps({"metric": "dry run (launch rehearsal, nothing measured)", "value": None, "unit": "queries/s", "n_gpus": ranks,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "ranks": ranks, "backend": backend}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--batch", type=int, default=0,
                    help="queries per step (each sweeps the index on its own); default 256 on one GPU, 256 x world (512..2048) when sharded")
    ap.add_argument("--sim", default="COSINE", choices=["EUCLIDEAN", "COSINE", "MAXIMUM_INNER_PRODUCT"])
    ap.add_argument("--query-bits", type=int, default=4)
    ap.add_argument("--index-bits", type=int, default=1, help="2..8: multi-bit index (rows handed over as one byte per dimension)")
    ap.add_argument("--config", default=None, choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs: c2 = 1Mx768 qb4/ib1 k100; c3 = the default (10Mx768); c4 = 10Mx1536 MAXIMUM_INNER_PRODUCT; "
                         "c5 = 1Mx1024 queryBits 8 / indexBits 2")
    ap.add_argument("--sub-batch", type=int, default=0, help="queries per device launch sequence (pipelined inside a step); 0 = the library's choice by index size")
    ap.add_argument("--slots", type=int, default=3, help="pipeline slots (streams) inside the library")
    ap.add_argument("--replay-threads", type=int, default=16, help="host threads replaying the reference heap")
    ap.add_argument("--pilot", type=int, default=32768, help="replicated pilot rows per non-root shard (multi-GPU)")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--recall-rows", type=int, default=0, help="rows of the fp32 recall probe (0 = the rows of the timed index, at most 10 M: the headline size)")
    ap.add_argument("--latency-calls", type=int, default=300, help="single-query calls timed for the latency object (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size oracle check of one timed query")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value")
    ap.add_argument("--batched-sub-batch", type=int, default=0, help="queries per launch chain of the batched mode (0: the library's choice, 64 on the matrix cores)")
    ap.add_argument("--shared-sweep", type=int, default=32, help="also time the batched mode (queries per shared sweep; 0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1: nccl (= RCCL) or gloo (rehearsal)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: take the sharded code path (process group, collectives) even with one rank")
    ap.add_argument("--shard-of", default=None, help=argparse.SUPPRESS)   # "r/w" with --force-dist: this process is rank r of w of the index (shard_shape_leg)
    ap.add_argument("--no-shard-shape", action="store_true", help="skip the leg that times the 8-GPU shard shape on this GPU (a child process with an RCCL group of one rank)")
    ap.add_argument("--no-c1", action="store_true", help="skip BASELINE config 1 through the JavaScript host (tests/js/bench_c1.js)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of BASELINE configs 2, 4 and 5 (default run on one GPU only)")
    ap.add_argument("--no-napi", action="store_true", help="skip the Node/N-API leg (the drop-in boundary driven from node at the timed size)")
    ap.add_argument("--no-hbm-only", action="store_true", help="skip the leg that repeats the timed step with nothing kept in the Infinity Cache")
    ap.add_argument("--no-raw", action="store_true", help="skip the raw-query leg (query quantization inside the timed region)")
    ap.add_argument("--quantize-threads", type=int, default=16, help="host threads of bbq_quantize_queries in the raw-query leg")
    ap.add_argument("--inprocess-shards", type=int, default=4,
                    help="one GPU: also time the in-process multi-device index (bbq_index_create_multi) with this many shards, all on GPU 0 "
                         "(prices the sharded pipeline's own overhead); 0 = skip.  With --gpus N the leg runs over the N devices instead")
    ap.add_argument("--inprocess-only", action="store_true", help=argparse.SUPPRESS)   # child process of an N-GPU run, see inprocess_only()
    ap.add_argument("--dry-run", action="store_true",
                    help="launch rehearsal without a device: start the ranks, form the process group, run the barriers and the max-over-ranks "
                         "reduction, print the line with value null - NO measurement is made (CPU test of the --gpus N launch path)")
    args = ap.parse_args()
    if args.config == "c2":
        args.rows, args.dim = 1_000_000, 768
    elif args.config == "c4":
        args.rows, args.dim, args.sim = 10_000_000, 1536, "MAXIMUM_INNER_PRODUCT"
    elif args.config == "c5":
        args.rows, args.dim, args.query_bits, args.index_bits = 1_000_000, 1024, 8, 2

    if args.inprocess_only:
        raise SystemExit(inprocess_only(args))
    rc = self_launch(args)
    if rc is not None:
        raise SystemExit(rc)
    if args.dry_run:
        raise SystemExit(dry_run(args))

    # stdout carries ONE JSON line and nothing else: libraries that print to file descriptor 1 (RCCL's version banner, gloo's connection
    # report) are sent to stderr for the rest of the run; the line itself is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import bbq_amd as B

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.same_device:
            local = 0
        ndev = torch.cuda.device_count()
        if ndev > 0:
            local %= ndev   # a launcher that narrows the visible devices per rank leaves each rank with device 0
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")   # the collectives' kernels share the GPU with sweeps that keep every CU busy
            # a rank that fails must not leave the others waiting in a collective for the default ten minutes
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=300))
    if world != args.gpus:   # self_launch() has already refused this; kept as a guard for callers that bypass main()
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    device = local if world > 1 else 0
    if B.device_count() < 1:
        raise SystemExit("bench.py: no HIP device (libbbq has no CPU fallback)")
    torch.cuda.set_device(device)

    if args.batch <= 0:
        # sharded: every rank sweeps only 1/world of the rows per query, so the batch grows with the world to keep the sweep of one
        # batch (and with it the share of the per-batch exchange and pipeline drain) what it is on one GPU
        args.batch = 256 if world == 1 else min(2048, max(512, 256 * world))
    N, dim, k, Q = args.rows, args.dim, args.k, args.batch
    SIM = {"EUCLIDEAN": 0, "COSINE": 1, "MAXIMUM_INNER_PRODUCT": 2}[args.sim]
    QB, IB = args.query_bits, args.index_bits
    pb = (dim + 7) // 8
    centroid = synth_centroid(dim)
    cdp = float(B.centroid_dp(centroid))   # getCentroidDP(undefined) = centroid . centroid

    def rows_of(a, b):
        return synth_rows(1, a, b, pb) if IB == 1 else synth_rows_multibit(1, a, b, dim, IB)

    shard = (N + world - 1) // world
    r0, r1 = min(rank * shard, N), min((rank + 1) * shard, N)
    emu = None
    if args.shard_of:   # one rank of a larger world, alone in its process group (shard_shape_leg)
        er, ew = [int(x) for x in args.shard_of.split("/")]
        shard = (N + ew - 1) // ew
        r0, r1 = min(er * shard, N), min((er + 1) * shard, N)
        emu = (er, ew)
    t0 = time.perf_counter()
    codes, corr = rows_of(r0, r1)
    pilot = None
    if rank > 0 or (emu and emu[0] > 0):
        P = min(args.pilot, r0) // 1024 * 1024
        if P > 0:
            pilot = rows_of(0, P)
    ix = B.Index(codes, corr, dim, cdp, device=device, index_bits=IB, row_base=r0,
                 pilot_codes=None if pilot is None else pilot[0], pilot_corr=None if pilot is None else pilot[1])
    # 0: the library's choice by the rows of the index / shard and the queries of the call (bbq_core.cpp effective_batch): 32 from 6 M rows,
    # 64 from 2.5 M, 128 below, halved while the call would have fewer than four sub-batches
    if args.sub_batch > 0:
        sub_batch = min(args.sub_batch, Q)
    else:
        sub_batch = 32 if r1 - r0 >= 6_000_000 else 64 if r1 - r0 >= 2_500_000 else 128
        while sub_batch > 32 and Q < 4 * sub_batch:   # at least four sub-batches per call where possible
            sub_batch //= 2
    ix.set_option("batch_queries", min(args.sub_batch, Q))
    ix.set_option("pipeline_slots", args.slots)
    ix.set_option("replay_threads", args.replay_threads)
    for o in args.opt:
        name, val = o.split("=")
        ix.set_option(name, int(val))
    bytes_per_row = ix.bytes_per_row
    log("rank %d: shard rows [%d,%d) built in %.1fs, %d B/row" % (rank, r0, r1, time.perf_counter() - t0, bytes_per_row))
    n_steps = args.warmup + args.steps
    qq_all, qc_all = synth_queries(2, n_steps * Q, dim, QB)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    results = []
    batches = [(qq_all[i * Q:(i + 1) * Q], qc_all[i * Q:(i + 1) * Q]) for i in range(n_steps)]
    if dist is None:
        def run(bs):
            for qq, qc in bs:
                results.append(ix.search_batch(qq, qc, QB, SIM, k))
    else:
        from bbq_amd.distributed import ShardedSearcher
        searcher = ShardedSearcher(ix, N, k, Q, query_bits=QB, sim=SIM, replay_threads=max(args.replay_threads, 16), device="cuda",
                                   collective_device="cuda" if args.backend == "nccl" else "cpu")

        def run(bs):
            results.extend(searcher.search_stream(bs))

    run(batches[:args.warmup])
    ix.reset_stats()
    if dist is not None:
        searcher.phases_ms()   # forget the warm-up
    barrier()
    t0 = time.perf_counter()
    run(batches[args.warmup:])
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = ix.stats()
    sharded = None
    if dist is not None:
        # where a batch's time goes on every rank (scan on the scanner thread, the rest next to it on the main thread): max over ranks
        mine = searcher.phases_ms()
        keys = ["scan", "exchange", "to_host", "merge", "answers", "lists"]
        tt = torch.tensor([mine[k_] for k_ in keys], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sharded = {"max_over_ranks_ms_per_batch": {k_: round(float(v), 3) for k_, v in zip(keys, tt.tolist())}, "rank0": mine,
                   "what": "scan = sweep of a batch, enqueue to device-done (overlaps the previous batch's exchange + merge); exchange = all_to_all of the "
                           "shard-local answers ((k + 3) x 8 B per shard and query); merge = bbq_merge_answers of the owner's block; answers = "
                           "all_gather of the owners' per-query status + gather to rank 0; lists = the candidate lists of the queries with equal "
                           "scores in their answer (those queries only) to rank 0 + their heap replay"}
        # per-rank roofline fraction of the dominant launch (HIP events on its own stream)
        lb = st["total_scan_bytes"] / max(st["total_scan_launches"], 1)
        lm = st["total_scan_ms"] / max(st["total_scan_launches"], 1)
        fr = torch.tensor([lb / (lm * 1e-3) / 1e9 / HBM_PEAK_GBS if lm > 0 else 0.0], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        allf = [torch.zeros_like(fr) for _ in range(world)]
        dist.all_gather(allf, fr)
        sharded["frac_hipevent_per_rank"] = [round(float(x.item()), 4) for x in allf]
        sharded["list_path_batches"] = searcher.list_batches
        sharded["last_exchange"] = searcher.last_exchange
    piloted_scan = None
    if dist is not None and args.shard_of and pilot is not None:
        # A shard WITH a pilot replica cannot be merged alone (the replica's rows belong to rank 0, which is not in this group: the merge
        # then sees boundaries it cannot prove and replays nearly every query).  What such a rank costs is its sweep: batches through
        # bbq_shard_scan_begin / _wait alone, two in flight as the searcher keeps them.
        tks = []
        t_s = time.perf_counter()
        for i, (qq_s, qc_s) in enumerate(batches[args.warmup:]):
            tks.append(searcher._begin(searcher.bufs[i % len(searcher.bufs)], qq_s, qc_s))
            if len(tks) == 2:
                searcher._finish(tks.pop(0))
        while tks:
            searcher._finish(tks.pop(0))
        torch.cuda.synchronize()
        d_s = time.perf_counter() - t_s
        piloted_scan = {"ms_per_batch": d_s / args.steps * 1e3, "queries_per_s": args.steps * Q / d_s}
    shard_direct = None
    if dist is not None and args.shard_of and pilot is None:
        # shard_shape_leg: the shard's own rows behind a plain index (no pilot replica, rows numbered from 0: a shard handle only answers
        # through the shard scan) must give the same answer - the top k of rows [r0, r1)
        dix = B.Index(codes, corr, dim, cdp, device=device, index_bits=IB)
        dres = dix.search_batch(batches[args.warmup][0], batches[args.warmup][1], QB, SIM, k)
        dix.close()
        sres = results[args.warmup]
        shard_direct = bool((dres[0] + r0 == sres[0]).all() and (dres[1].view(np.uint32) == sres[1].view(np.uint32)).all())
    batched = None
    if dist is None and args.shared_sweep in (4, 8, 32) and IB == 1:
        # API extension, reported separately: `shared_sweep` queries share one sweep of the index
        ix.set_option("sweep_share", args.shared_sweep)
        # on the matrix cores a workgroup serves two groups of 32 queries per tile load: 64 queries per launch chain
        # (the library's choice there, bbq_core.cpp effective_batch)
        b_sub = min(Q, args.batched_sub_batch or 64) if args.shared_sweep == 32 else sub_batch
        if args.batched_sub_batch > 0:
            ix.set_option("batch_queries", min(Q, args.batched_sub_batch))
        res_b = ix.search_batch(batches[args.warmup][0], batches[args.warmup][1], QB, SIM, k)
        same = bool((res_b[0] == results[args.warmup][0]).all() and (res_b[1].view(np.uint32) == results[args.warmup][1].view(np.uint32)).all())
        ix.reset_stats()
        barrier()
        tb = time.perf_counter()
        for qq_b, qc_b in batches[args.warmup:]:
            ix.search_batch(qq_b, qc_b, QB, SIM, k)
        barrier()
        dtb = time.perf_counter() - tb
        stb = ix.stats()
        lb = stb["total_scan_bytes"] / max(stb["total_scan_launches"], 1)
        lms = stb["total_scan_ms"] / max(stb["total_scan_launches"], 1)
        bq = args.steps * Q / dtb
        stream_gbps = lb / (lms * 1e-3) / 1e9 if lms > 0 else None
        # matrix-core work of the shared sweep: rows x dim x 2 ops per query.  Query values <= 15 run as FP6 x FP4 on
        # v_mfma_f32_32x32x64_f8f6f4 (dense peak ~10 PFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md), larger ones as int8 (~5 POP/s)
        fp_form = args.shared_sweep == 32 and QB <= 4
        mx_peak = 10.0e15 if fp_form else 5.0e15
        batched = {"queries_per_sweep": args.shared_sweep, "value": bq, "unit": "queries/s",
                   "roofline": {"frac_hbm": (stream_gbps / HBM_PEAK_GBS) if stream_gbps else None,
                                "frac_matrix_peak": bq * N * dim * 2 / mx_peak if args.shared_sweep == 32 else None,
                                "matrix_form": ("FP6 x FP4 (v_mfma_f32_32x32x64_f8f6f4)" if fp_form else "int8 (v_mfma_i32_32x32x32_i8)") if args.shared_sweep == 32 else None,
                                "bound": "vector issue next to the matrix cores: vector and matrix instructions of different waves barely overlap on a SIMD "
                                         "(profiles/r04_mfma_pmc.json: vector ALUs active 58 %, matrix cores busy 56 % of the SIMD time at 1.8 GHz, 2 084 cycles per tile and group of 32 queries; "
                                         "scripts/ubench/valu_mfma_overlap.hip; DESIGN.md 'Shared sweeps')",
                                "index_stream_GBps": stream_gbps, "matrix_peak_ops": mx_peak,
                                # bytes the dominant launch reads per index row for ALL its queries (120 = one load of the 96-byte codes +
                                # 24 B of exact corrections serves the launch's 64 queries)
                                "bytes_per_row_of_the_launch": (lb / max(stb["last_scan_rows"], 1) * b_sub) if stb.get("last_scan_rows") else None},
                   "identical_to_unshared": same, "bound": "matrix cores + vector issue (groups of 32 queries; two groups share each tile load)" if args.shared_sweep == 32 else "valu (popcount + f64 bound per query; HBM bytes amortised over the sweep)",
                   "index_stream_GBps": lb / (lms * 1e-3) / 1e9 if lms > 0 else None}
        batched["queries_per_launch"] = b_sub
        # the same mode with four times as many queries per call (the step's 256 queries are four launch chains: the first chain's small
        # launches and the last chain's tail run alone; a longer call spends more of its time with several chains in flight)
        big_calls = [(np.concatenate([b[0] for b in batches[i:i + 4]]), np.concatenate([b[1] for b in batches[i:i + 4]]))
                     for i in range(args.warmup, len(batches) - 3, 4)]
        if big_calls:
            ix.search_batch(big_calls[0][0], big_calls[0][1], QB, SIM, k)
            barrier()
            tb4 = time.perf_counter()
            for qq_b, qc_b in big_calls:
                res_b4 = ix.search_batch(qq_b, qc_b, QB, SIM, k)
            barrier()
            dtb4 = time.perf_counter() - tb4
            batched["value_at_4x_queries_per_call"] = len(big_calls) * 4 * Q / dtb4
            batched["queries_per_call_4x"] = 4 * Q
            i_last = args.warmup + 4 * (len(big_calls) - 1)   # the batches the last long call was made of
            batched["identical_to_unshared_4x"] = bool(
                (res_b4[0] == np.concatenate([r[0] for r in results[i_last:i_last + 4]])).all() and
                (res_b4[1].view(np.uint32) == np.concatenate([r[1] for r in results[i_last:i_last + 4]]).view(np.uint32)).all())
        ix.set_option("sweep_share", 1)
        ix.set_option("batch_queries", min(args.sub_batch, Q))

    raw = None
    if dist is None and not args.no_raw:
        # the reference's timed unit starts from the RAW query: searchNearestNeighbors normalises and quantizes it first
        # (src/binaryQuantizationFormat.ts:337-347).  Same step as above with bbq_quantize_queries inside the timed region.
        raw_all = synth_raw_queries(6, n_steps * Q, dim)
        raw_batches = [raw_all[i * Q:(i + 1) * Q] for i in range(n_steps)]
        for rb in raw_batches[:args.warmup]:
            ix.search_raw_batch(rb, centroid, SIM, QB, k, n_threads=args.quantize_threads)
        barrier()
        tr = time.perf_counter()
        for rb in raw_batches[args.warmup:]:
            raw_res = ix.search_raw_batch(rb, centroid, SIM, QB, k, n_threads=args.quantize_threads)
        barrier()
        dtr = time.perf_counter() - tr
        # what the two-call form gives for the last batch (and what it costs): quantize everything, then search
        t1 = time.perf_counter()
        qq_r, qc_r = B.quantize_queries(raw_batches[-1], centroid, SIM, QB, n_threads=args.quantize_threads)
        tq = time.perf_counter() - t1
        two = ix.search_batch(qq_r, qc_r, QB, SIM, k)
        raw = {"value": args.steps * Q / dtr, "unit": "queries/s", "ms_per_step": dtr / args.steps * 1e3,
               "quantize_threads": args.quantize_threads, "quantize_us_per_query_wall_unpipelined": tq / Q * 1e6,
               "identical_to_quantize_then_search": bool((raw_res[0] == two[0]).all() and (raw_res[1].view(np.uint32) == two[1].view(np.uint32)).all()),
               "what": "the same step from raw fp32 queries through bbq_search_raw_batch: normalise + quantizeQueryVector on host threads, chunk by chunk, "
                       "while the sub-batches in front are already on the device, + sweep + top-k"}

    hbm_only = None
    if dist is not None and not args.no_hbm_only and not args.shard_of and not any(o.startswith("resident_mb=") for o in args.opt):
        # sharded run: the same batches with nothing kept in the Infinity Cache on any rank; the slowest rank's dominant launch is the
        # job's strict HBM figure (roofline.frac)
        ix.set_option("resident_mb", 0)
        if args.backend == "nccl" or torch.cuda.is_available():
            flush_infinity_cache(torch)
        run(batches[:1])
        ix.reset_stats()
        barrier()
        th0 = time.perf_counter()
        strict_batches = batches[args.warmup:args.warmup + max(2, args.steps // 2)]
        run(strict_batches)
        barrier()
        dth = time.perf_counter() - th0
        sth = ix.stats()
        ix.set_option("resident_mb", -1)
        lb = sth["total_scan_bytes"] / max(sth["total_scan_launches"], 1)
        lm = sth["total_scan_ms"] / max(sth["total_scan_launches"], 1)
        tt = torch.tensor([lb / (lm * 1e-3) / 1e9 if lm > 0 else 0.0, -dth], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        gb, dth = float(tt[0].item()), -float(tt[1].item())
        hbm_only = {"value": len(strict_batches) * Q / dth, "unit": "queries/s", "ms_per_step": dth / len(strict_batches) * 1e3,
                    "end_to_end_hbm_frac": (len(strict_batches) * Q / dth) * (N / world) * bytes_per_row / 1e9 / HBM_PEAK_GBS,
                    "roofline_frac_dominant_launch": gb / HBM_PEAK_GBS if gb > 0 else None, "dominant_launch_GBps": gb if gb > 0 else None,
                    "dominant_launch_ms": lm, "what": "library option resident_mb=0 on every rank; dominant launch = the slowest rank's, by HIP events"}
        run(batches[:1])   # warm the caches again
    if dist is None and not args.no_hbm_only and not any(o.startswith("resident_mb=") for o in args.opt):
        # the same timed step with NOTHING kept in the Infinity Cache (resident_mb 0: every byte of every sweep streamed from HBM with
        # non-temporal loads): the figure the HBM roofline in its strict sense applies to, next to the default above
        ix.set_option("resident_mb", 0)
        flush_infinity_cache(torch)   # what the default mode left resident must not serve the strict run
        time_steps(ix, batches[:1], QB, SIM, k)
        ix.reset_stats()
        dth, res_h = time_steps(ix, batches[args.warmup:], QB, SIM, k)
        sth = ix.stats()
        lb = sth["total_scan_bytes"] / max(sth["total_scan_launches"], 1)
        lm = sth["total_scan_ms"] / max(sth["total_scan_launches"], 1)
        hbm_only = {"value": args.steps * Q / dth, "unit": "queries/s", "ms_per_step": dth / args.steps * 1e3,
                    "end_to_end_hbm_frac": (args.steps * Q / dth) * N * bytes_per_row / 1e9 / HBM_PEAK_GBS,
                    "roofline_frac_dominant_launch": (lb / (lm * 1e-3) / 1e9 / HBM_PEAK_GBS) if lm > 0 else None,
                    "dominant_launch_GBps": (lb / (lm * 1e-3) / 1e9) if lm > 0 else None, "dominant_launch_ms": lm,
                    "cache_resident_bytes": sth["resident_bytes"],
                    "identical_to_default": bool((res_h[0][0] == results[args.warmup][0]).all() and
                                                 (res_h[0][1].view(np.uint32) == results[args.warmup][1].view(np.uint32)).all()),
                    "what": "library option resident_mb=0 after the Infinity Cache has been flushed (1 GiB written and read twice): no part of the index "
                            "is kept in, or served by, the cache between sweeps"}
        ix.set_option("resident_mb", -1)
        time_steps(ix, batches[:1], QB, SIM, k)    # warm the cache again for the legs below

    inproc = None
    if dist is None and args.inprocess_shards > 1:
        # the in-process multi-device index (what the TS host uses, bbq_index_create_multi) with every shard on THIS GPU: the same
        # rows, the same queries; prices the sharded pipeline's own overhead (pilot replicas, per-shard launch chains, answer merge)
        try:
            mx = B.Index.create_multi(codes, corr, dim, cdp, [device] * args.inprocess_shards, index_bits=IB, pilot_rows=args.pilot)
            mx.set_option("pipeline_slots", args.slots)
            mx.set_option("replay_threads", args.replay_threads)
            mres = [mx.search_batch(qq, qc, QB, SIM, k) for qq, qc in batches[:max(1, args.warmup)]][0]
            mx.reset_stats()
            barrier()
            tm = time.perf_counter()
            for qq_m, qc_m in batches[args.warmup:]:
                mx.search_batch(qq_m, qc_m, QB, SIM, k)
            barrier()
            dtm = time.perf_counter() - tm
            mst = mx.stats()
            inproc = {"value": args.steps * Q / dtm, "unit": "queries/s", "shards": mx.shards, "devices": "all shards on GPU %d (one-GPU run)" % device,
                      "vs_single_index": (args.steps * Q / dtm) / (args.steps * Q / dt), "host_replays": mst["host_replays"],
                      "dense_fallbacks": mst["dense_fallbacks"], "cache_resident_bytes_all_shards": mst["resident_bytes"],
                      "identical_to_single_index": bool((mres[0] == results[0][0]).all() and (mres[1].view(np.uint32) == results[0][1].view(np.uint32)).all())}
            mx.close()
        except Exception as e:  # informational leg
            inproc = {"error": str(e)[:300]}

    latency = None
    if dist is None and args.latency_calls > 0:
        # the reference's own call shape: ONE synchronous query per call (searchNearestNeighbors), through bbq_search
        lq, lc = synth_queries(3, 64, dim, QB)
        for i in range(20):
            ix.search(lq[i % 64], lc[i % 64], QB, SIM, k)
        ts = []
        for i in range(args.latency_calls):
            t1 = time.perf_counter()
            ix.search(lq[i % 64], lc[i % 64], QB, SIM, k)
            ts.append((time.perf_counter() - t1) * 1e3)
        ts = np.sort(np.array(ts))
        latency = {"queries_per_call": 1, "calls": len(ts), "p50_ms": float(ts[len(ts) // 2]), "p99_ms": float(ts[min(len(ts) - 1, int(len(ts) * 0.99))]),
                   "min_ms": float(ts[0]), "one_sweep_at_peak_ms": N * bytes_per_row / (HBM_PEAK_GBS * 1e9) * 1e3,
                   "path": "bbq_search (ctypes): quantized query in, top-k out; host replays per call: %d" % ix.stats()["host_replays"]}

    if rank == 0:
        qps = args.steps * Q / dt
        launch_bytes = st["total_scan_bytes"] / max(st["total_scan_launches"], 1)
        launch_ms = st["total_scan_ms"] / max(st["total_scan_launches"], 1)
        achieved = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        # committed rocprofv3 evidence for THIS kernel and layout (profiles/dominant_kernel.json, scripts/summarize_profiles.py): the
        # PMC traffic per row and the kernel-trace average; anything measured on another layout / width is not quoted
        traffic, traffic_src, frac_rocprof, rocprof_src, frac_rocprof_strict, rocprof_src_strict = None, None, None, None, None, None
        try:
            reg = json.load(open(os.path.join(ROOT, "profiles", "dominant_kernel.json")))
            for e in reg["entries"]:
                if (e["rows"], e["dim"], e["index_bits"], e["query_bits"], e["bytes_per_row"]) == (N, dim, IB, QB, bytes_per_row) and world == 1:
                    if e.get("hbm_bytes_per_row") is not None:
                        traffic = e["hbm_bytes_per_row"] * (launch_bytes / bytes_per_row)
                        traffic_src = {"file": "profiles/dominant_kernel.json", "hbm_bytes_per_row": e["hbm_bytes_per_row"], "kernel": e["kernel"],
                                       "bytes_per_row": e["bytes_per_row"], "collected": e["collected"], "how": e["pmc_how"]}
                    if e.get("trace_avg_GBps") is not None:
                        frac_rocprof = e["trace_avg_GBps"] / HBM_PEAK_GBS
                        rocprof_src = {"file": "profiles/dominant_kernel.json", "trace_avg_us": e["trace_avg_us"], "launches": e["trace_launches"],
                                       "bytes_per_launch": e["trace_bytes_per_launch"], "collected": e["collected"]}
                    if e.get("strict_trace_avg_GBps") is not None:   # the same kernel traced with resident_mb 0
                        frac_rocprof_strict = e["strict_trace_avg_GBps"] / HBM_PEAK_GBS
                        rocprof_src_strict = {"file": "profiles/dominant_kernel.json", "trace_avg_us": e["strict_trace_avg_us"],
                                              "launches": e["strict_trace_launches"], "collected": e["collected"]}
                    break
        except Exception:
            pass
        out = {
            # BASELINE.json's metric string for the headline configuration; a descriptive one for the other configs
            "metric": ("queries/sec + recall@100 vs fp32 brute-force, 10M\u00d7768 1-bit index, k=100"
                       if (N, dim, k, QB, IB, args.sim) == (10_000_000, 768, 100, 4, 1, "COSINE")
                       else "queries/sec, %dx%d %d-bit index, queryBits=%d, k=%d, %s" % (N, dim, IB, QB, k, args.sim)),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 popcount + f64 score epilogue" if IB == 1 else "u4/u8 packed integer dot (u32 accumulate) + f64 score epilogue",
            "data": "synthetic",
            "config": {"workload": "%dx%d-dim %d-bit index, queryBits=%d, k=%d, %s, row-sharded over %d GPU(s)" % (N, dim, IB, QB, k, args.sim, world),
                       "queries_per_step": Q, "queries_per_launch": sub_batch, "sweeps_per_query": 1,
                       "queries_per_step_rule": "256 at --gpus 1; min(2048, max(512, 256 x N)) at --gpus N: every rank sweeps 1/N of the rows per "
                                                "query, so the step grows with N to keep a rank's sweep per step what it is on one GPU "
                                                "(512 / 1024 / 2048 at 2 / 4 / 8 GPUs); each query still sweeps the whole index once",
                       "pipeline_slots": args.slots, "replay_threads": args.replay_threads, "bytes_per_row": bytes_per_row,
                       "parallelism": "row-shard x%d" % world},
            # frac: this run's strict HBM figure (hbm_only leg), HIP events on the kernel's own stream; frac_rocprof_avg(_strict): the
            # committed kernel-trace averages of the same kernel on the same layout (a different box and run: the pool has faster and
            # slower boxes; `collected` names the round)
            "roofline": roofline_object(
                (hbm_only or {}).get("dominant_launch_GBps"), achieved, st["resident_bytes"] / float(max(1, (N // world) * bytes_per_row)),
                {"frac_hipevent_algorithmic": achieved / HBM_PEAK_GBS, "frac_rocprof_avg": frac_rocprof, "rocprof_source": rocprof_src,
                 "frac_rocprof_avg_strict": frac_rocprof_strict, "rocprof_source_strict": rocprof_src_strict,
                 "traffic": traffic, "traffic_source": traffic_src, "kernel": "bbq_scan_kernel (largest segment launch)",
                 "bytes_per_launch": launch_bytes, "avg_launch_ms": launch_ms, "launches_timed": st["total_scan_launches"],
                 "strict_avg_launch_ms": (hbm_only or {}).get("dominant_launch_ms"),
                 "cache_resident_bytes_per_sweep": st["resident_bytes"]}),
            "ranks": dist.get_world_size() if dist is not None else 1, "backend": (dist.get_backend() if dist is not None else None),
            "end_to_end_frac_algorithmic": (qps * (N / world) * bytes_per_row / 1e9) / HBM_PEAK_GBS,
            "end_to_end_hbm_frac_strict": (hbm_only or {}).get("end_to_end_hbm_frac"),
            "candidates_per_query": st["candidates"] / float(Q) if dist is None else None,
            "dense_fallbacks": st["dense_fallbacks"],
        }
        out["argv"] = " ".join(sys.argv[1:])
        out["config"]["shard_rows"] = r1 - r0
        out["config"]["pilot_rows"] = 0 if pilot is None else int(pilot[0].shape[0])
        if shard_direct is not None:
            out["identical_to_direct"] = shard_direct
            out["replayed_queries_per_batch"] = (searcher.last_exchange or {}).get("replayed_queries")
        if piloted_scan is not None:
            out["piloted_scan"] = piloted_scan
        if sharded is not None:
            out["sharded_phases"] = sharded
        if latency is not None:
            out["latency"] = latency
        if batched is not None:
            out["batched"] = batched
        if raw is not None:
            out["raw_queries"] = raw
        if hbm_only is not None:
            out["hbm_only"] = hbm_only
        if inproc is not None:
            out["inprocess_multi"] = inproc
        if world == 1 and dist is None and not args.no_napi and IB == 1:
            out["napi"] = napi_leg(B, ix, centroid, dim, k, args.sim, SIM, QB)
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed at N=1 only
            cqps, done, rows, secs = cpu_baseline(dim, k, codes, corr, qq_all, qc_all, cdp, QB, SIM, IB)
            model, ncpu = host_cpu()
            out["cpu_baseline"] = {"value": cqps, "unit": "queries/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the timed queries x all %d rows of the same index in %.1fs (%.3f us/row), one thread; nothing extrapolated"
                                             % (done, rows, secs, secs / (done * rows) * 1e6),
                                   "host_cpu": model, "host_logical_cores": ncpu}
            # the same loops restated in JS, under node on this box's host, 1 core
            import shutil
            import subprocess
            if shutil.which("node") and IB == 1:
                try:
                    js_rows = min(N, 1_000_000)   # the reference's per-row object layout cannot hold 10 M rows in Node
                    r = subprocess.run(["node", os.path.join(ROOT, "oracle", "bbq_oracle_js_baseline.js"), str(js_rows), str(dim), str(k), "3"],
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=180)
                    js = json.loads(r.stdout.strip().splitlines()[-1])
                    out["cpu_baseline_js"] = {"value": 1e6 / (js["us_per_row"] * N), "unit": "queries/s", "cores": 1,
                                              "kind": "port (JavaScript restatement of the reference loops, node %s)" % js["node"],
                                              "sample": "%d queries x %d rows in %.1fs (%.3f us/row)%s"
                                                        % (js["queries"], js["rows"], js["seconds"], js["us_per_row"],
                                                           "" if js["rows"] == N else ", per-row cost scaled to %d rows (Node cannot hold the reference's per-row objects for that many)" % N),
                                              "host_cpu": model, "host_logical_cores": ncpu}
                except Exception as e:  # the baseline is informational: never fail the bench for it
                    out["cpu_baseline_js"] = {"error": str(e)[:200]}
        if not args.no_parity:
            # parity check of the timed configuration (any N): one full-size query of the timed region against the oracle
            import orclib as O
            if world == 1:
                fc, fr = codes, corr
            else:
                fc, fr = rows_of(0, N)  # rank 0 rebuilds the whole index for the checker only
            s32 = oracle_scores(O, fc, fr, dim, qq_all[args.warmup * Q], qc_all[args.warmup * Q], QB, SIM, cdp, IB)
            oi, osc = O.heap_topk(s32, k)
            gi, gs, _ = results[args.warmup]
            out["parity_full_size"] = bool((gi[0] == oi).all() and (gs[0].view(np.uint32) == osc.view(np.uint32)).all())
        if IB != 1 and QB not in (1, 4):
            out["parity_note"] = ("the reference throws for queryBits=%d on an indexBits=%d index (src/binaryQuantizedScorer.ts:95-97): the integer dot "
                                  "product is pinned by fixtures of computeQuantizedDotProduct, the float score is its per-row 4-bit form - parity unpinned" % (QB, IB))
        headline = (N, dim, k, QB, IB, args.sim) == (10_000_000, 768, 100, 4, 1, "COSINE")
        if world == 1 and dist is None and headline and not args.no_configs:
            # the other BASELINE configs as short legs of the same run (c3 = the headline line itself; c1 is the CPU-sized README case:
            # tests/js/bench_c1.js).  Each leg builds its own synthetic index, so the headline index goes first.
            ix.close()
            del codes, corr
            legs = {}
            for name, (cN, cdim, cqb, cib, csim) in {"c2": (1_000_000, 768, 4, 1, "COSINE"), "c4": (10_000_000, 1536, 4, 1, "MAXIMUM_INNER_PRODUCT"),
                                                     "c5": (1_000_000, 1024, 8, 2, "COSINE")}.items():
                try:
                    # a step of a 1 M-row config lasts 4-10 ms: more steps than the headline's, so that a leg times at least ~0.2 s
                    leg_steps = args.steps if cN >= 10_000_000 else max(args.steps * 8, 40)
                    legs[name] = config_leg(B, torch, name, cN, cdim, k, cqb, cib, csim, device, leg_steps, max(args.warmup, 4), 256, args.slots,
                                            args.replay_threads, parity=not args.no_parity)
                except Exception as e:
                    legs[name] = {"error": str(e)[:300]}
                log("config leg %s: %s" % (name, json.dumps(legs[name])[:400]))
            if not args.no_c1:
                legs["c1"] = c1_leg()
                log("config leg c1: %s" % json.dumps(legs["c1"])[:400])
            out["configs"] = legs
            notes = {n_: l_["parity_note"] for n_, l_ in legs.items() if isinstance(l_, dict) and "parity_note" in l_}
            if notes:
                out["parity_notes"] = notes   # what no fixture of the reference can pin (it throws there), said at the top level
        if world == 1 and dist is None and headline and not args.no_shard_shape:
            try:
                ix.close()
            except Exception:
                pass
            out["shard_shape_8gpu"] = shard_shape_leg(args, qps)
            log("shard shape leg: %s" % json.dumps(out["shard_shape_8gpu"])[:500])
        if not args.no_recall and IB == 1:
            ix.close()   # the probe needs the memory (30.7 GB of fp32 at the headline size, twice on the device while the index is built)
            rec, desc = recall_probe(B, device, n=args.recall_rows if args.recall_rows > 0 else min(N, 10_000_000), dim=dim)
            out["recall_at_100"] = rec
            out["recall_config"] = desc
    ix.close()
    if dist is not None:
        dist.barrier()   # every rank has released its shard
        if rank == 0 and args.inprocess_shards != 0:
            # the other deployment of the same sharding - one process, one handle over the N devices - timed by a child process while
            # the ranks wait (their GPUs are idle now)
            out["inprocess_multi"] = run_inprocess_child(args)
        dist.barrier()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
