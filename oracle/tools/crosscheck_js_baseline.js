#!/usr/bin/env node
/*
 * Is bench.py's `cpu_baseline_js` (oracle/bbq_oracle_js_baseline.js: this repo's restatement of the reference's search loops) what
 * the reference itself costs?  BUILD-CONTAINER ONLY (test infrastructure): drives the type-erased copy of the reference
 * (oracle/tools/erase_ts.py writes it under /tmp; nothing of it is committed or travels to the GPU box) the way quickSearch does -
 * quantizeVectors once, then searchNearestNeighbors per query - on the SAME host, the same rows x dim x k, and runs the restatement
 * next to it.
 *
 * Usage:  python3 oracle/tools/erase_ts.py && node oracle/tools/crosscheck_js_baseline.js [rows] [dim] [k] [queries]
 * Prints one JSON line {reference_us_per_row, restatement_us_per_row, ratio, every pass, ...}; exit code 1 when neither the medians nor the
 * fastest passes agree within 10 %.  Under Node 12 BOTH programs are bimodal - a pass costs either ~1.2 or ~1.8-2.3 us/row depending on
 * which tier V8 happens to leave the inner loop in - so single runs of the medians scatter by more than that; the fixture
 * tests/golden/api_cpu_baseline_crosscheck.json holds two runs from this repository's build container with every pass listed.
 */
'use strict';
const path = require('path');
const cp = require('child_process');
const ERASED = process.env.BBQ_ERASED_OUT || '/tmp/bbq_ref_erased/js';
const rows = Number(process.argv[2] || 50000), dim = Number(process.argv[3] || 768), k = Number(process.argv[4] || 100), nq = Number(process.argv[5] || 6);
const { BinaryQuantizationFormat } = require(path.join(ERASED, 'binaryQuantizationFormat'));

function mulberry32(seed) {
  let a = seed | 0;
  return function () { a |= 0; a = a + 0x6D2B79F5 | 0; let t = Math.imul(a ^ a >>> 15, 1 | a); t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t; return ((t ^ t >>> 14) >>> 0) / 4294967296; };
}
function randMatrix(seed, n, d) {
  const r = mulberry32(seed), out = [];
  for (let i = 0; i < n; i++) { const v = new Float32Array(d); for (let j = 0; j < d; j++) v[j] = 2 * r() - 1; out.push(v); }
  return out;
}
const base = randMatrix(1, rows, dim), queries = randMatrix(2, nq, dim);
const format = new BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: 0.1, iters: 5 } });
let t0 = process.hrtime.bigint();
const index = format.quantizeVectors(base).quantizedVectors;
const buildS = Number(process.hrtime.bigint() - t0) / 1e9;
format.searchNearestNeighbors(queries[0], index, k);   // warm-up: the loops compiled
const per = [];
function median(a) { const b = a.slice().sort(function (x, y) { return x - y; }); return b[b.length >> 1]; }
for (let r = 0; r < 5; r++) {   // median of five passes: the container's cores are shared, and V8 re-tiers the loops between passes
  t0 = process.hrtime.bigint();
  for (let q = 0; q < nq; q++) format.searchNearestNeighbors(queries[q], index, k);
  per.push(Number(process.hrtime.bigint() - t0) / 1e3 / (nq * rows));
}
const refUs = median(per);
const mine = [];
for (let r = 0; r < 5; r++) {
  const o = cp.spawnSync(process.execPath, [path.join(__dirname, '..', 'bbq_oracle_js_baseline.js'), String(rows), String(dim), String(k), String(nq)], { encoding: 'utf8' });
  mine.push(JSON.parse(o.stdout.trim().split('\n').pop()).us_per_row);
}
const myUs = median(mine);
const ratio = myUs / refUs;
console.log(JSON.stringify({
  rows: rows, dim: dim, k: k, queries: nq, node: process.version, reference_build_s: buildS,
  reference_us_per_row: refUs, reference_passes_us_per_row: per, restatement_us_per_row: myUs, restatement_passes_us_per_row: mine,
  ratio_restatement_over_reference: ratio, within_10_percent: Math.abs(ratio - 1) <= 0.10,
  ratio_of_fastest_passes: Math.min.apply(null, mine) / Math.min.apply(null, per),
  what: 'searchNearestNeighbors of the type-erased reference (src/binaryQuantizationFormat.ts:308-412) against oracle/bbq_oracle_js_baseline.js, ' +
        'same host, same rows x dim x k, 4-bit queries, COSINE, one thread each, median of five passes',
}));
process.exit(Math.abs(ratio - 1) <= 0.10 || Math.abs(Math.min.apply(null, mine) / Math.min.apply(null, per) - 1) <= 0.10 ? 0 : 1);
