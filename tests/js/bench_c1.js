'use strict';
// BASELINE config 1 (README bench shape): quickSearch over 1000 x 128-dim Float32Array, k=10, COSINE, through the
// JavaScript drop-in API.  Prints ms per quickSearch call (re-quantizes + re-uploads the 1000 targets every call, as the
// reference does, src/index.ts:109) and ms per search on a pre-built index.
const T = require('./common');
const bbq = T.bbq;
const base = T.randMatrix(3, 1000, 128), query = T.randMatrix(4, 1, 128)[0];
let res = bbq.quickSearch(query, base, 10);
const want = [438, 839, 190, 656, 637, 545, 630, 174, 862, 42];
const ok = JSON.stringify(res.map(function (r) { return r.index; })) === JSON.stringify(want);
let t0 = process.hrtime.bigint();
const reps = 30;
for (let i = 0; i < reps; i++) res = bbq.quickSearch(query, base, 10);
const msQuick = Number(process.hrtime.bigint() - t0) / 1e6 / reps;
const fmt = bbq.createBinaryQuantizationFormat();
const index = fmt.quantizeVectors(base).quantizedVectors;
fmt.searchNearestNeighbors(query, index, 10);
t0 = process.hrtime.bigint();
for (let i = 0; i < 200; i++) fmt.searchNearestNeighbors(query, index, 10);
const ms10 = Number(process.hrtime.bigint() - t0) / 1e6 / 200;
t0 = process.hrtime.bigint();
for (let i = 0; i < 200; i++) fmt.searchNearestNeighbors(query, index, 100);
const ms100 = Number(process.hrtime.bigint() - t0) / 1e6 / 200;
console.log(JSON.stringify({ config: 'quickSearch 1000x128 k=10 COSINE', top10_matches_reference: ok, ms_per_quickSearch: msQuick,
  ms_per_search_prebuilt_k10: ms10, ms_per_search_prebuilt_k100: ms100, node: process.version }));
