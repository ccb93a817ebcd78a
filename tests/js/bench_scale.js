'use strict';
// The drop-in TypeScript/JavaScript boundary at BASELINE scale: the reference's call shapes - one synchronous
// format.searchNearestNeighbors(query, values, k) per query (src/binaryQuantizationFormat.ts:308-412, src/index.ts:95-111) and
// the batch extension - through the N-API addon (bbq_napi.node) on an index of 1 M - 10 M rows, with RAW fp32 queries
// (normalisation + query quantization inside the timed call, as in the reference).
//
//   node bench_scale.js <index prefix> <queries.f32> <dim> <k> <similarity> <answers.bin> [batchReps] [singleCalls]
//
// <index prefix>.veb/.vemb: an index some other host of libbbq has saved (bench.py saves the very index it times through ctypes);
// <queries.f32>: nq x dim little-endian floats.  Writes the batch answers as [nq][k] int32 indices + [nq][k] f32 scores + [nq] int32
// counts to <answers.bin> (the caller compares them bit for bit with its own) and prints ONE JSON line with the timings.
const fs = require('fs');
const path = require('path');
const bbq = require(path.join(__dirname, '..', '..', 'better-binary-quantization_amd', 'js', 'index.js'));

const argv = process.argv.slice(2);
if (argv.length < 6) { console.error('usage: bench_scale.js prefix queries.f32 dim k similarity answers.bin [batchReps] [singleCalls]'); process.exit(2); }
const prefix = argv[0], dim = Number(argv[2]), k = Number(argv[3]), sim = argv[4], outPath = argv[5];
const batchReps = Number(argv[6] || 3), singleCalls = Number(argv[7] || 200);

const qbuf = fs.readFileSync(argv[1]);
const flat = new Float32Array(qbuf.buffer, qbuf.byteOffset, qbuf.length / 4);
const nq = flat.length / dim;
const queries = [];
for (let i = 0; i < nq; i++) queries.push(flat.subarray(i * dim, (i + 1) * dim));

const format = bbq.createBinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: sim, lambda: 0.1, iters: 5 } });
let t0 = process.hrtime.bigint();
const values = format.loadIndex(prefix);
const loadMs = Number(process.hrtime.bigint() - t0) / 1e6;

// ---- batch call shape (extension): every query still sweeps the index on its own
let res = format.searchNearestNeighborsBatch(queries, values, k);   // warm-up: workspace allocation,
res = format.searchNearestNeighborsBatch(queries, values, k);       // and the host's own loops compiled (the first calls run them interpreted)
// how much of a call is spent inside the addon (bbq_search_raw_batch: quantize + sweeps + top-k), the rest being this host's own
// work (flattening the queries, 51 K result objects per call, their garbage collection)
const insideBefore = bbq._hostClock.insideAddonNs;
t0 = process.hrtime.bigint();
for (let r = 0; r < batchReps; r++) res = format.searchNearestNeighborsBatch(queries, values, k);
const batchS = Number(process.hrtime.bigint() - t0) / 1e9;
const nativeNs = bbq._hostClock.insideAddonNs - insideBefore;

const idx = new Int32Array(nq * k), sc = new Float32Array(nq * k), cnt = new Int32Array(nq);
for (let i = 0; i < nq; i++) {
  cnt[i] = res[i].length;
  for (let j = 0; j < res[i].length; j++) { idx[i * k + j] = res[i][j].index; sc[i * k + j] = res[i][j].score; }
}
fs.writeFileSync(outPath, Buffer.concat([Buffer.from(idx.buffer), Buffer.from(sc.buffer), Buffer.from(cnt.buffer)]));

// ---- the reference's own call shape: one synchronous searchNearestNeighbors per query
for (let i = 0; i < 20; i++) format.searchNearestNeighbors(queries[i % nq], values, k);
const ms = [];
let sameAsBatch = true;
for (let i = 0; i < singleCalls; i++) {
  const q = i % nq;
  const t1 = process.hrtime.bigint();
  const r = format.searchNearestNeighbors(queries[q], values, k);
  ms.push(Number(process.hrtime.bigint() - t1) / 1e6);
  if (r.length !== res[q].length) sameAsBatch = false;
  for (let j = 0; j < r.length && sameAsBatch; j++) if (r[j].index !== res[q][j].index || r[j].score !== res[q][j].score) sameAsBatch = false;
}
const slowest = ms.map(function (v, i) { return { call: i, ms: v }; }).sort(function (a, b) { return b.ms - a.ms; }).slice(0, 5);
ms.sort(function (a, b) { return a - b; });
const stats = values.deviceStats();
console.log(JSON.stringify({
  rows: values.size(), dim: values.dimension(), k: k, queries: nq, node: process.version, load_ms: loadMs,
  batch_queries_per_s: nq * batchReps / batchS, batch_ms_per_call: batchS / batchReps * 1e3, batch_ms_per_call_inside_addon: Number(nativeNs) / 1e6 / batchReps,
  single_p50_ms: ms[ms.length >> 1], single_p99_ms: ms[Math.min(ms.length - 1, Math.floor(ms.length * 0.99))], single_min_ms: ms[0],
  single_queries_per_s: 1e3 / (ms.reduce(function (a, b) { return a + b; }, 0) / ms.length),
  single_equals_batch: sameAsBatch, host_replays_last_call: stats.hostReplays, single_slowest_calls: slowest,
}));
process.exit(0);   // the device copy goes with the process (dispose() would first fetch the rows back for vectorValue())
