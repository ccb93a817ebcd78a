#!/usr/bin/env node
/*
 * JS restatement of the reference's search loops, for the CPU-baseline leg of bench.py ONLY (test infrastructure;
 * the product never loads this).  The reference's TypeScript cannot travel to the GPU box and cannot be compiled
 * there (no tsc), so BASELINE.md section 4 asks for "this repo's own JS restatement of the same loops (per-bit multiply
 * inner loop, 1000-row batches, per-row result objects, JS min-heap), run with node, 1 core".  Follows:
 *   searchNearestNeighbors            src/binaryQuantizationFormat.ts:349-411
 *   computeBatchQuantizedScores       src/binaryQuantizedScorer.ts:315-400
 *   createDirectPackedBuffer          src/batchDotProduct.ts:420-436
 *   computeBatchFourBitDotProduct...  src/utils/computeBatchFourBitDotProductDirectPacked.ts:10-53
 *   computeBatchFourBitSimilarityScores (COSINE branch)  src/batchDotProduct.ts:554-617
 *   MinHeap                           src/minHeap.ts:9-130
 * Usage: node bbq_oracle_js_baseline.js <rows> <dim> <k> <queries>   -> one JSON line {us_per_row, ...}
 * Synthetic quantized index (scan time is value-independent); parity of these loops is not what this file is for -
 * the C oracle is the pinned checker.
 */
'use strict';
const rows = Number(process.argv[2] || 100000), dim = Number(process.argv[3] || 768), k = Number(process.argv[4] || 100);
const nq = Number(process.argv[5] || 2);
const pb = Math.ceil(dim / 8);
function mulberry32(seed) {
  let a = seed | 0;
  return function () { a |= 0; a = a + 0x6D2B79F5 | 0; let t = Math.imul(a ^ a >>> 15, 1 | a); t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t; return ((t ^ t >>> 14) >>> 0) / 4294967296; };
}
const r = mulberry32(7);
const vectors = [], corrections = [];
for (let i = 0; i < rows; i++) {
  const v = new Uint8Array(pb); let ones = 0;
  for (let j = 0; j < pb; j++) { v[j] = (r() * 256) | 0; }
  for (let j = 0; j < pb; j++) { let b = v[j]; while (b) { ones += b & 1; b >>= 1; } }
  vectors.push(v);
  corrections.push({ lowerInterval: -0.04 * (0.9 + 0.2 * r()), upperInterval: 0.04 * (0.9 + 0.2 * r()), additionalCorrection: 1e-4 * (2 * r() - 1), quantizedComponentSum: ones });
}
const index = { vectorValue: function (o) { return vectors[o]; }, getCorrectiveTerms: function (o) { return corrections[o]; }, dimension: function () { return dim; }, size: function () { return rows; } };
const FOUR_BIT_SCALE = 1.0 / 15;
function createDirectPackedBuffer(tv, ords, size) {
  const buf = new Uint8Array(size * ords.length);
  for (let i = 0; i < ords.length; i++) buf.set(tv.vectorValue(ords[i]), i * size);
  return buf;
}
function fourBitDots(q, buf, n, dimension) {
  const results = new Array(n).fill(0), pd = Math.ceil(dimension / 8), main = Math.floor(dimension / 8);
  for (let i = 0; i < n; i++) {
    let dot = 0; const off = i * pd;
    for (let j = 0; j < main; j++) {
      const p = buf[off + j], qo = j * 8;
      dot += q[qo] * ((p >> 7) & 1); dot += q[qo + 1] * ((p >> 6) & 1); dot += q[qo + 2] * ((p >> 5) & 1); dot += q[qo + 3] * ((p >> 4) & 1);
      dot += q[qo + 4] * ((p >> 3) & 1); dot += q[qo + 5] * ((p >> 2) & 1); dot += q[qo + 6] * ((p >> 1) & 1); dot += q[qo + 7] * (p & 1);
    }
    const rem = main * 8;
    if (rem < dimension) { const last = buf[off + main]; for (let d = rem; d < dimension; d++) dot += q[d] * ((last >> (7 - (d % 8))) & 1); }
    results[i] = dot;
  }
  return results;
}
function scores4(qcDists, qc, tv, ords, dimension, cdp) {
  const out = [];
  for (let i = 0; i < ords.length; i++) {
    const ic = tv.getCorrectiveTerms(ords[i]);
    const x1 = ic.quantizedComponentSum, ax = ic.lowerInterval, lx = ic.upperInterval - ax, ay = qc.lowerInterval;
    const ly = (qc.upperInterval - ay) * FOUR_BIT_SCALE, y1 = qc.quantizedComponentSum;
    const score = ax * ay * dimension + ay * lx * x1 + ax * ly * y1 + lx * ly * qcDists[i];
    const adj = score + qc.additionalCorrection + ic.additionalCorrection - cdp;
    out.push(Math.max((1 + adj) / 2, 0));
  }
  return out;
}
function batchScores(q, qc, tv, ords) {
  // the reference runs its batch path inside try { } catch { fall back to the per-row scorer } (src/binaryQuantizedScorer.ts:331-418);
  // the handler is part of what the loops cost under V8 (Node 12: 1.3 us/row without it, 1.9 with it - the reference's own figure,
  // oracle/tools/crosscheck_js_baseline.js)
  try {
    const buf = createDirectPackedBuffer(tv, ords, Math.ceil(tv.dimension() / 8));
    const d = fourBitDots(q, buf, ords.length, tv.dimension());
    const s = scores4(d, qc, tv, ords, tv.dimension(), 0.00091);
    const res = [];
    for (let i = 0; i < ords.length; i++) res.push({ score: s[i], bitDotProduct: d[i], corrections: { query: qc, index: tv.getCorrectiveTerms(ords[i]) } });
    return res;
  } catch (error) {
    console.warn('批量计算失败，回退到原始方法:', error);
    return [];
  }
}
class MinHeap {
  constructor(c) { this.heap = []; this.c = c; }
  size() { return this.heap.length; } peek() { return this.heap[0]; } isEmpty() { return this.heap.length === 0; }
  push(x) { const h = this.heap; h.push(x); let i = h.length - 1; while (i > 0) { const p = Math.floor((i - 1) / 2); if (this.c(h[i], h[p]) >= 0) break; const t = h[i]; h[i] = h[p]; h[p] = t; i = p; } }
  pop() { const h = this.heap; if (!h.length) return null; const m = h[0], l = h.pop(); if (h.length) { h[0] = l; let i = 0; for (;;) { let s = i; const a = 2 * i + 1, b = 2 * i + 2; if (a < h.length && this.c(h[a], h[s]) < 0) s = a; if (b < h.length && this.c(h[b], h[s]) < 0) s = b; if (s === i) break; const t = h[i]; h[i] = h[s]; h[s] = t; i = s; } } return m; }
}
function search(q, qc, tv, k) {
  const n = tv.size(), scores = new Float32Array(n), indices = new Int32Array(n);
  for (let i = 0; i < n; i++) indices[i] = i;
  for (let i = 0; i < n; i += 1000) {
    const end = Math.min(i + 1000, n);
    const ords = Array.from({ length: end - i }, function (_, j) { return i + j; });
    const res = batchScores(q, qc, tv, ords);
    for (let j = 0; j < res.length; j++) scores[i + j] = res[j].score;
  }
  const heap = new MinHeap(function (a, b) { return a.score - b.score; }), k2 = Math.min(k, n);
  for (let i = 0; i < n; i++) {
    const s = scores[i];
    if (heap.size() < k2) heap.push({ score: s, index: indices[i] });
    else { const p = heap.peek(); if (p && s > p.score) { heap.pop(); heap.push({ score: s, index: indices[i] }); } }
  }
  const out = []; while (!heap.isEmpty()) out.push(heap.pop()); out.reverse(); return out;
}
const q = new Uint8Array(dim); for (let d = 0; d < dim; d++) q[d] = (r() * 16) | 0;
let y1 = 0; for (let d = 0; d < dim; d++) y1 += q[d];
const qc = { lowerInterval: -0.15, upperInterval: 0.148, additionalCorrection: -0.0028, quantizedComponentSum: y1 };
search(q, qc, index, k);  // warm up the JIT
const t0 = process.hrtime.bigint();
let top = null;
for (let i = 0; i < nq; i++) top = search(q, qc, index, k);
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
console.log(JSON.stringify({ us_per_row: dt / (nq * rows) * 1e6, rows: rows, dim: dim, k: k, queries: nq, seconds: dt, node: process.version, top0: top[0] }));
