#!/usr/bin/env node
/*
 * Golden-vector generator (TEST INFRASTRUCTURE, build-container only).
 *
 * Drives the type-erased copy of the reference (oracle/tools/erase_ts.py writes it
 * under /tmp; it is never committed) exactly the way the reference's public API
 * does (src/index.ts:95-111 quickSearch: new BinaryQuantizationFormat ->
 * quantizeVectors -> searchNearestNeighbors) and records inputs + every
 * intermediate the parity tests pin:
 *   centroid, packed codes, corrections (f64), quantized query, query corrections,
 *   centroidDP, integer qcDist per row, f64 score per row, final top-k (index, f32 score).
 *
 * Usage:  python3 oracle/tools/erase_ts.py && node oracle/tools/gen_fixtures.js [outdir]
 * Output: tests/golden/*.json  (typed arrays as base64 of little-endian bytes)
 *
 * Runs on Node >= 12 (no `??`, no `?.`).  Needs /root/reference only through the
 * erased copy; nothing here runs on the GPU box.
 */
'use strict';
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');

const ERASED = process.env.BBQ_ERASED_OUT || '/tmp/bbq_ref_erased/js';
const OUT = process.argv[2] || path.join(__dirname, '..', '..', 'tests', 'golden');
const { BinaryQuantizationFormat } = require(path.join(ERASED, 'binaryQuantizationFormat'));
const { normalizeVector } = require(path.join(ERASED, 'vectorOperations'));
const { computeQuantizedDotProduct } = require(path.join(ERASED, 'bitwiseDotProduct'));
const { getOversampledTopKWithHeap, getOversampledTopKWithSort } = require(path.join(ERASED, 'topKSelector'));
const { computeSimilarity } = require(path.join(ERASED, 'vectorSimilarity'));

// ---------------------------------------------------------------- PRNG (SURVEY 8d)
function mulberry32(seed) {
  let a = seed | 0;
  return function () {
    a |= 0; a = a + 0x6D2B79F5 | 0;
    let t = Math.imul(a ^ a >>> 15, 1 | a);
    t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t;
    return ((t ^ t >>> 14) >>> 0) / 4294967296;
  };
}
function randMatrix(seed, n, dim) {
  const r = mulberry32(seed);
  const out = [];
  for (let i = 0; i < n; i++) {
    const v = new Float32Array(dim);
    for (let j = 0; j < dim; j++) v[j] = 2 * r() - 1;
    out.push(v);
  }
  return out;
}
// rows drawn (with repetition) from a small pool -> many exactly equal scores (heap tie stress)
function dupPoolMatrix(seed, seed2, n, dim, pool) {
  const p = randMatrix(seed, pool, dim);
  const r = mulberry32(seed2);
  const out = [];
  for (let i = 0; i < n; i++) out.push(new Float32Array(p[Math.floor(r() * pool)]));
  return out;
}
// the reference's own closed-form recall dataset (tests/recall.test.ts:20-57); inputs are stored inline
function closedForm(n, dim, offset) {
  const out = [];
  for (let i = 0; i < n; i++) {
    const v = new Float32Array(dim);
    for (let j = 0; j < dim; j++) {
      const s = (i + offset) * 1000 + j;
      v[j] = Math.sin(s) * 0.5 + Math.cos(s * 0.7) * 0.3;
    }
    out.push(v);
  }
  return out;
}

// ---------------------------------------------------------------- encoding helpers
function b64(typed) { return Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength).toString('base64'); }
function sha(typed) {
  return crypto.createHash('sha256').update(Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength)).digest('hex');
}
function flat32(vs) {
  const dim = vs[0].length; const f = new Float32Array(vs.length * dim);
  for (let i = 0; i < vs.length; i++) f.set(vs[i], i * dim);
  return f;
}

// ---------------------------------------------------------------- one case
function runCase(c) {
  const t0 = Date.now();
  let base, queries;
  const gen = c.gen;
  if (gen.kind === 'mulberry32') {
    base = randMatrix(gen.base_seed, c.n, c.dim);
    queries = randMatrix(gen.query_seed, c.nq, c.dim);
  } else if (gen.kind === 'dup_pool') {
    base = dupPoolMatrix(gen.base_seed, gen.pick_seed, c.n, c.dim, gen.pool);
    queries = randMatrix(gen.query_seed, c.nq, c.dim);
  } else if (gen.kind === 'closed_form') {
    base = closedForm(c.n, c.dim, 0);
    queries = closedForm(c.nq, c.dim, 1000);
  } else if (gen.kind === 'inline') {
    base = gen.base.map(function (r) { return new Float32Array(r); });
    queries = gen.queries.map(function (r) { return new Float32Array(r); });
  } else throw new Error('gen?');

  const cfg = { queryBits: c.qb, indexBits: c.ib, quantizer: { similarityFunction: c.sim, lambda: c.lambda, iters: c.iters } };
  const format = new BinaryQuantizationFormat(cfg);
  const warnings = [];
  const origWarn = console.warn;
  console.warn = function () { warnings.push(String(arguments[0])); };

  const out = {
    name: c.name, sim: c.sim, qb: c.qb, ib: c.ib, lambda: c.lambda, iters: c.iters,
    dim: c.dim, n: c.n, k: c.k, nq: c.nq, full: !!c.full,
    gen: gen.kind === 'inline' ? { kind: 'inline' } : gen,
  };
  if (gen.kind === 'inline' || gen.kind === 'closed_form') {
    out.base_f32 = b64(flat32(base)); out.queries_f32 = b64(flat32(queries));
  } else {
    out.base_sha256 = sha(flat32(base)); out.queries_sha256 = sha(flat32(queries));
  }

  const built = format.quantizeVectors(base);
  const index = built.quantizedVectors;
  const n = index.size(), dim = index.dimension();
  const rowBytes = index.vectorValue(0).length;
  const codes = new Uint8Array(n * rowBytes);
  const corr = new Float64Array(n * 4);
  for (let i = 0; i < n; i++) {
    codes.set(index.vectorValue(i), i * rowBytes);
    const t = index.getCorrectiveTerms(i);
    corr[4 * i] = t.lowerInterval; corr[4 * i + 1] = t.upperInterval;
    corr[4 * i + 2] = t.additionalCorrection; corr[4 * i + 3] = t.quantizedComponentSum;
  }
  out.row_bytes = rowBytes;
  out.centroid_f32 = b64(index.getCentroid());
  out.centroid_dp_f64 = b64(new Float64Array([index.getCentroidDP()]));
  if (c.full) { out.codes_u8 = b64(codes); out.corr_f64 = b64(corr); }
  out.codes_sha256 = sha(codes); out.corr_sha256 = sha(corr);
  // always keep a few rows inline so a mismatch can be localised
  const keep = Math.min(n, 4);
  out.head_codes_u8 = b64(codes.subarray(0, keep * rowBytes));
  out.head_corr_f64 = b64(corr.subarray(0, keep * 4));

  out.queries = [];
  for (let qi = 0; qi < queries.length; qi++) {
    const query = queries[qi];
    const rec = {};
    // search path input to the scorer (double normalisation for COSINE, A.5-1)
    const processed = c.sim === 'COSINE' ? normalizeVector(query) : query;
    const qq = format.quantizeQueryVector(processed, index.getCentroid());
    rec.qquant_u8 = b64(qq.quantizedQuery);
    rec.qcorr_f64 = b64(new Float64Array([qq.queryCorrections.lowerInterval, qq.queryCorrections.upperInterval,
      qq.queryCorrections.additionalCorrection, qq.queryCorrections.quantizedComponentSum]));
    const qcd = new Int32Array(n), sc = new Float64Array(n);
    let perRowOk = true;
    try {
      for (let i = 0; i < n; i += 1000) {
        const ords = [];
        for (let j = i; j < Math.min(i + 1000, n); j++) ords.push(j);
        const res = format.getScorer().computeBatchQuantizedScores(qq.quantizedQuery, qq.queryCorrections, index, ords, c.qb);
        for (let j = 0; j < res.length; j++) { qcd[i + j] = res[j].bitDotProduct; sc[i + j] = res[j].score; }
      }
    } catch (e) { perRowOk = false; rec.per_row_error = String(e.message); }
    if (perRowOk) {
      if (c.full) { rec.qcdist_i32 = b64(qcd); rec.score_f64 = b64(sc); }
      rec.qcdist_sha256 = sha(qcd); rec.score_sha256 = sha(sc);
      const sf = new Float32Array(sc); rec.score_f32_sha256 = sha(sf);
      rec.head_qcdist_i32 = b64(qcd.subarray(0, keep)); rec.head_score_f64 = b64(sc.subarray(0, keep));
    }
    const ks = c.ks || [c.k];
    rec.topk = [];
    for (let kk = 0; kk < ks.length; kk++) {
      try {
        const r = format.searchNearestNeighbors(query, index, ks[kk]);
        const idx = new Int32Array(r.length), s32 = new Float32Array(r.length);
        for (let i = 0; i < r.length; i++) { idx[i] = r[i].index; s32[i] = r[i].score; }
        rec.topk.push({ k: ks[kk], idx_i32: b64(idx), score_f32: b64(s32) });
      } catch (e) { rec.topk.push({ k: ks[kk], error: String(e.message) }); }
    }
    if (c.qb === 1 || c.qb === 4) {
      // the single-row scorer (src/binaryQuantizedScorer.ts:69-301; fallback path, defined differently from the batch path for
      // 4-bit queries: centroidDP = query . centroid when the original query is passed, 0 otherwise)
      const rows = Math.min(n, 6), a = new Float64Array(rows), b = new Float64Array(rows), dots = new Int32Array(rows);
      for (let i = 0; i < rows; i++) {
        const r0 = format.getScorer().computeQuantizedScore(qq.quantizedQuery, qq.queryCorrections, index, i, c.qb);
        const r1 = format.getScorer().computeQuantizedScore(qq.quantizedQuery, qq.queryCorrections, index, i, c.qb, query);
        a[i] = r0.score; b[i] = r1.score; dots[i] = r0.bitDotProduct;
      }
      rec.single_row = { score_f64: b64(a), score_with_query_f64: b64(b), dot_i32: b64(dots) };
    }
    if (c.ib === 1 && c.qb !== 1) {
      // the optional 6th argument of computeBatchQuantizedScores (src/binaryQuantizedScorer.ts:315-321, :372-381): centroidDP becomes
      // query . centroid for multi-bit queries; searchNearestNeighbors never passes it, so only this entry pins it
      const rows = Math.min(n, 6), ords = [], b = new Float64Array(rows);
      for (let i = 0; i < rows; i++) ords.push(i);
      const res = format.getScorer().computeBatchQuantizedScores(qq.quantizedQuery, qq.queryCorrections, index, ords, c.qb, query);
      for (let i = 0; i < rows; i++) b[i] = res[i].score;
      rec.batch_with_query = { score_f64: b64(b) };
    }
    if (c.oversample) {
      const r = getOversampledTopKWithHeap(query, index, base, c.k, c.oversample, format);
      rec.oversample = { factor: c.oversample, idx: r.map(function (x) { return x.index; }) };
    }
    out.queries.push(rec);
  }
  console.warn = origWarn;
  out.warnings = warnings.length;
  out.first_warning = warnings.length ? warnings[0] : null;
  fs.writeFileSync(path.join(OUT, c.name + '.json'), JSON.stringify(out));
  console.log(c.name, 'n=' + n, 'dim=' + dim, 'warnings=' + warnings.length, (Date.now() - t0) + ' ms');
}

// ---------------------------------------------------------------- exact rerank pin (SURVEY 8f-3)
// computeSimilarity (src/vectorSimilarity.ts:14-126) for every (query, row) under all three functions, and both
// oversample-then-rerank selectors (src/topKSelector.ts:29-115) on a COSINE index of the same rows.
function runRerank(c) {
  const base = randMatrix(c.base_seed, c.n, c.dim), queries = randMatrix(c.query_seed, c.nq, c.dim);
  // hostile rows: zero vector (norm 0 -> cosine 0), huge and tiny magnitudes, a copy of query 0 (cosine ~1, distance 0)
  if (c.n > 12) {
    base[5].fill(0);
    for (let j = 0; j < c.dim; j++) { base[7][j] *= 1e18; base[9][j] *= 1e-30; base[11][j] = queries[0][j]; }
  }
  if (c.zero_query) queries[c.nq - 1].fill(0);
  const flat = new Float32Array(c.n * c.dim), qflat = new Float32Array(c.nq * c.dim);
  base.forEach(function (v, i) { flat.set(v, i * c.dim); });
  queries.forEach(function (v, i) { qflat.set(v, i * c.dim); });
  const out = { name: c.name, dim: c.dim, n: c.n, nq: c.nq, k: c.k, lambda: 0.1, iters: 5,
    base_f32: b64(flat), queries_f32: b64(qflat), true_f64: {}, oversample: [] };
  SIMS.forEach(function (sim) {
    const t = new Float64Array(c.nq * c.n);
    for (let qi = 0; qi < c.nq; qi++) for (let i = 0; i < c.n; i++) t[qi * c.n + i] = computeSimilarity(queries[qi], base[i], sim);
    out.true_f64[sim] = b64(t);
  });
  const format = new BinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: 'COSINE', lambda: 0.1, iters: 5 } });
  const index = format.quantizeVectors(base).quantizedVectors;
  const pack = function (r) {
    const idx = new Int32Array(r.length), qs = new Float32Array(r.length), ts = new Float64Array(r.length);
    for (let i = 0; i < r.length; i++) { idx[i] = r[i].index; qs[i] = r[i].quantizedScore; ts[i] = r[i].trueScore; }
    return { idx_i32: b64(idx), quantized_f32: b64(qs), true_f64: b64(ts) };
  };
  c.factors.forEach(function (f) {
    for (let qi = 0; qi < c.nq; qi++) {
      out.oversample.push({ query: qi, factor: f,
        heap: pack(getOversampledTopKWithHeap(queries[qi], index, base, c.k, f, format)),
        sort: pack(getOversampledTopKWithSort(queries[qi], index, base, c.k, f, format)) });
    }
  });
  fs.writeFileSync(path.join(OUT, c.name + '.json'), JSON.stringify(out));
  console.log(c.name, 'n=' + c.n, 'dim=' + c.dim);
}

// ---------------------------------------------------------------- integer-dot pin for multi-bit index (H4: float score unpinned)
function runIntDot(c) {
  const base = randMatrix(c.base_seed, c.n, c.dim), queries = randMatrix(c.query_seed, c.nq, c.dim);
  const format = new BinaryQuantizationFormat({ queryBits: c.qb, indexBits: c.ib, quantizer: { similarityFunction: c.sim, lambda: 0.1, iters: 5 } });
  const index = format.quantizeVectors(base).quantizedVectors;
  const n = index.size();
  const codes = new Uint8Array(n * c.dim), corr = new Float64Array(n * 4);
  for (let i = 0; i < n; i++) {
    codes.set(index.vectorValue(i), i * c.dim);
    const t = index.getCorrectiveTerms(i);
    corr[4 * i] = t.lowerInterval; corr[4 * i + 1] = t.upperInterval; corr[4 * i + 2] = t.additionalCorrection; corr[4 * i + 3] = t.quantizedComponentSum;
  }
  const out = { name: c.name, sim: c.sim, qb: c.qb, ib: c.ib, dim: c.dim, n: n, nq: c.nq, lambda: 0.1, iters: 5,
    gen: { kind: 'mulberry32', base_seed: c.base_seed, query_seed: c.query_seed },
    codes_unpacked_u8: b64(codes), corr_f64: b64(corr), centroid_f32: b64(index.getCentroid()), queries: [] };
  for (let qi = 0; qi < c.nq; qi++) {
    const processed = c.sim === 'COSINE' ? normalizeVector(queries[qi]) : queries[qi];
    const qq = format.quantizeQueryVector(processed, index.getCentroid());
    const d = new Int32Array(n);
    for (let i = 0; i < n; i++) d[i] = computeQuantizedDotProduct(qq.quantizedQuery, index.vectorValue(i));
    let searchErr = null;
    const ow = console.warn; console.warn = function () {};
    try { format.searchNearestNeighbors(queries[qi], index, 5); } catch (e) { searchErr = String(e.message); }
    console.warn = ow;
    out.queries.push({ qquant_u8: b64(qq.quantizedQuery), qcdist_i32: b64(d), search_error: searchErr });
  }
  fs.writeFileSync(path.join(OUT, c.name + '.json'), JSON.stringify(out));
  console.log(c.name, 'ok');
}

// ---------------------------------------------------------------- error-message / API-behaviour pins
function runApiBehaviour() {
  const out = { name: 'api_behaviour', cases: [] };
  function attempt(label, fn) {
    try { const r = fn(); out.cases.push({ label: label, ok: true, result: r === undefined ? null : r }); }
    catch (e) { out.cases.push({ label: label, ok: false, message: String(e.message) }); }
  }
  const base = randMatrix(21, 20, 8), q = randMatrix(22, 1, 8)[0];
  const F = function (extra) {
    const cfg = { quantizer: { similarityFunction: 'COSINE', lambda: 0.1, iters: 5 } };
    Object.keys(extra || {}).forEach(function (k) { cfg[k] = extra[k]; });
    return new BinaryQuantizationFormat(cfg);
  };
  attempt('ctor queryBits=0', function () { F({ queryBits: 0 }); });
  attempt('ctor queryBits=9', function () { F({ queryBits: 9 }); });
  attempt('ctor indexBits=0', function () { F({ indexBits: 0 }); });
  attempt('ctor indexBits=9', function () { F({ indexBits: 9 }); });
  attempt('quantize empty', function () { F().quantizeVectors([]); });
  attempt('quantize dim mismatch', function () { F().quantizeVectors([new Float32Array(4), new Float32Array(5)]); });
  attempt('quantize NaN', function () { F({ quantizer: { similarityFunction: 'EUCLIDEAN' } }).quantizeVectors([new Float32Array([1, NaN])]); });
  attempt('quantize Infinity', function () { F({ quantizer: { similarityFunction: 'EUCLIDEAN' } }).quantizeVectors([new Float32Array([1, Infinity])]); });
  const f = F(); const idx = f.quantizeVectors(base).quantizedVectors;
  attempt('search null query', function () { f.searchNearestNeighbors(null, idx, 3); });
  attempt('search null index', function () { f.searchNearestNeighbors(q, null, 3); });
  attempt('search k<0', function () { f.searchNearestNeighbors(q, idx, -1); });
  attempt('search dim mismatch', function () { f.searchNearestNeighbors(new Float32Array(7), idx, 3); });
  attempt('search k=0', function () { return f.searchNearestNeighbors(q, idx, 0); });
  attempt('search k>N length', function () { return f.searchNearestNeighbors(q, idx, 50).length; });
  attempt('vectorValue out of range', function () { idx.vectorValue(99); });
  attempt('getCorrectiveTerms out of range', function () { idx.getCorrectiveTerms(99); });
  attempt('index dimension', function () { return idx.dimension(); });
  attempt('index size', function () { return idx.size(); });
  attempt('getConfig', function () { return f.getConfig(); });
  // the small helpers / constants the package root re-exports (src/index.ts:20-37): values on fixed inputs
  const U = require(path.join(ERASED, 'utils')), VO = require(path.join(ERASED, 'vectorOperations')), VU = require(path.join(ERASED, 'vectorUtils'));
  const CN = require(path.join(ERASED, 'constants'));
  const va = new Float32Array([0.5, -1.25, 3, 0.1, -7.5]), vb = new Float32Array([2, 0.75, -0.3, 1e-3, 4]);
  const arr = function (t) { return Array.from(t); };
  out.helpers = { inputs: { a: arr(va), b: arr(vb) }, values: {}, errors: {} };
  const H = out.helpers.values;
  H.computeL2Norm = U.computeL2Norm(va); H.computeMean = U.computeMean(va); H.computeStd = U.computeStd(va, U.computeMean(va));
  H.clamp = [U.clamp(5, 0, 1), U.clamp(-5, 0, 1), U.clamp(0.25, 0, 1), U.clamp(NaN, 0, 1)].map(String);
  H.bitCount = [0, 1, 255, 0xF0F0F0F0, -1, 0x80000000].map(function (n) { return U.bitCount(n); });
  H.bitCountBytes = U.bitCountBytes(new Uint8Array([0, 255, 170, 1])); H.bitCountBytesOptimized = U.bitCountBytesOptimized(new Uint8Array([0, 255, 170, 1]));
  H.getBitCount = [0, 7, 255, 256 + 3].map(function (n) { return U.getBitCount(n); });
  H.BIT_COUNT_LOOKUP_TABLE_sha256 = sha(U.BIT_COUNT_LOOKUP_TABLE);
  H.isNearZero = [U.isNearZero(1e-9), U.isNearZero(1e-7), U.isNearZero(0.5, 1)];
  H.isNearEqual = [U.isNearEqual(1, 1 + 1e-9), U.isNearEqual(1, 1.1), U.isNearEqual(1, 1.1, 0.5)];
  H.scaleMaxInnerProductScore = [U.scaleMaxInnerProductScore(-3), U.scaleMaxInnerProductScore(0), U.scaleMaxInnerProductScore(2.5)];
  H.addVectors = arr(VO.addVectors(va, vb)); H.subtractVectors = arr(VO.subtractVectors(va, vb)); H.scaleVector = arr(VO.scaleVector(va, 0.3));
  H.centerVector = arr(VO.centerVector(va, vb)); H.copyVector = arr(VO.copyVector(va));
  H.computeVectorMagnitude = VU.computeVectorMagnitude(vb); H.createZeroVector = arr(VU.createZeroVector(3));
  H.createRandomVector_length = VU.createRandomVector(7, 2, 3).length;
  H.MINIMUM_MSE_GRID = CN.MINIMUM_MSE_GRID; H.FILE_EXTENSIONS = CN.FILE_EXTENSIONS; H.COMPONENT_NAMES = CN.COMPONENT_NAMES;
  H.NUMERICAL_CONSTANTS = CN.NUMERICAL_CONSTANTS;
  H.constants = { QUERY_BITS: CN.QUERY_BITS, INDEX_BITS: CN.INDEX_BITS, FOUR_BIT_SCALE: CN.FOUR_BIT_SCALE, DEFAULT_LAMBDA: CN.DEFAULT_LAMBDA, DEFAULT_ITERS: CN.DEFAULT_ITERS };
  [['addVectors', function () { VO.addVectors(va, new Float32Array(2)); }], ['subtractVectors', function () { VO.subtractVectors(va, new Float32Array(2)); }],
    ['centerVector', function () { VO.centerVector(va, new Float32Array(2)); }]].forEach(function (p) {
    try { p[1](); out.helpers.errors[p[0]] = null; } catch (e) { out.helpers.errors[p[0]] = String(e.message); }
  });
  // computeAccuracy (src/index.ts:120-134) = computeQuantizationAccuracy of a format with lambda 0.1 / iters 5
  // (src/binaryQuantizationFormat.ts:420-476, src/binaryQuantizedScorer.ts:429-617): the statistics, the scorer's small helpers, the errors
  out.accuracy = { n: 40, dim: 32, base_seed: 910, query_seed: 911, results: {}, scorer: {}, errors: {} };
  const accBase = randMatrix(910, 40, 32), accQueries = randMatrix(911, 40, 32);
  SIMS.forEach(function (sim) {
    [[4, 1], [1, 1], [4, 2]].forEach(function (qi) {
      const fmt = new BinaryQuantizationFormat({ queryBits: qi[0], indexBits: qi[1], quantizer: { similarityFunction: sim, lambda: 0.1, iters: 5 } });
      out.accuracy.results[sim + '_qb' + qi[0] + '_ib' + qi[1]] = fmt.computeQuantizationAccuracy(accBase, accQueries);
    });
  });
  const sc = F().getScorer();
  out.accuracy.scorer.compareScores = [[0.5, 0.4], [0, 0], [0, 1], [2, 2], [-1, 3]].map(function (p) {
    const r = sc.compareScores(p[0], p[1]);
    return { a: p[0], b: p[1], difference: r.difference, relativeError: String(r.relativeError), correlation: r.correlation };
  });
  out.accuracy.scorer.computeOriginalScore = SIMS.map(function (sim) { return sc.computeOriginalScore(accQueries[0], accBase[0], sim); });
  out.accuracy.scorer.computeQuantizationAccuracy = sc.computeQuantizationAccuracy([0.1, 0.5, 0.9, 0.3], [0.12, 0.45, 0.97, 0.3]);
  out.accuracy.scorer.constantScores = sc.computeQuantizationAccuracy([0.5, 0.5], [0.4, 0.6]);
  out.accuracy.scorer.getSimilarityFunction = sc.getSimilarityFunction();
  [['empty originals', function () { F().computeQuantizationAccuracy([], accQueries); }],
    ['empty queries', function () { F().computeQuantizationAccuracy(accBase, []); }],
    ['length mismatch', function () { F().computeQuantizationAccuracy(accBase, accQueries.slice(0, 3)); }],
    ['scores length mismatch', function () { sc.computeQuantizationAccuracy([1, 2], [1]); }],
    ['bad similarity', function () { sc.computeOriginalScore(accQueries[0], accBase[0], 'NOPE'); }]].forEach(function (p) {
    try { p[1](); out.accuracy.errors[p[0]] = null; } catch (e) { out.accuracy.errors[p[0]] = String(e.message); }
  });
  // the quantizer's remaining public utilities (src/optimizedScalarQuantizer.ts:67-93, 460-627)
  const OSQ = require(path.join(ERASED, 'optimizedScalarQuantizer')).OptimizedScalarQuantizer;
  const qz = out.quantizer_utils = { values: {}, errors: {} };
  qz.values.discretize = [[0, 8], [1, 8], [8, 8], [9, 8], [100, 64], [7.5, 4], [-3, 4]].map(function (p) { return OSQ.discretize(p[0], p[1]); });
  const q4 = new Uint8Array(19); { const r = mulberry32(930); for (let i = 0; i < q4.length; i++) q4[i] = Math.floor(r() * 16); }
  qz.q4 = arr(q4);
  { const o = new Uint8Array(q4.length * 4); OSQ.transposeHalfByte(q4, o); qz.values.transposeHalfByte = arr(o); }
  { const o = new Uint8Array(Math.ceil(q4.length / 8) * 4); OSQ.transposeHalfByteFast(q4, o); qz.values.transposeHalfByteFast = arr(o); }
  { const o = new Uint8Array(16 * 4); OSQ.transposeHalfByteFast(q4.subarray(0, 16), o); qz.values.transposeHalfByteFast_wide_output = arr(o); }
  OSQ.clearTransposeCache();
  qz.values.cache = [];
  { const o = new Uint8Array(q4.length * 4), o2 = new Uint8Array(q4.length * 4), copy = new Uint8Array(q4);
    qz.values.cache.push(OSQ.getTransposeCacheStats());
    OSQ.transposeHalfByteOptimized(q4, o); qz.values.cache.push(OSQ.getTransposeCacheStats());
    OSQ.transposeHalfByteOptimized(q4, o2); qz.values.cache.push(OSQ.getTransposeCacheStats());
    OSQ.transposeHalfByteOptimized(copy, o2); qz.values.cache.push(OSQ.getTransposeCacheStats());     // same values, another array: a miss
    OSQ.transposeHalfByteOptimized(copy, o2, false); qz.values.cache.push(OSQ.getTransposeCacheStats());  // uncached: not counted
    q4[0] ^= 1; OSQ.transposeHalfByteOptimized(q4, o2); q4[0] ^= 1;                                     // a hit returns the stale planes
    qz.values.cache_stale_hit_equals_first = arr(o2).join() === arr(o).join();
    OSQ.clearTransposeCache(); qz.values.cache.push(OSQ.getTransposeCacheStats()); }
  { const qn = new OSQ({ similarityFunction: 'EUCLIDEAN', lambda: 0.1, iters: 5 });
    const v = randMatrix(931, 1, 24)[0], cen = randMatrix(932, 1, 24)[0].map(function (x) { return x * 0.1; });
    const d = [new Uint8Array(24), new Uint8Array(24), new Uint8Array(24)];
    const rs = qn.multiScalarQuantize(v, d, [1, 4, 7], Float32Array.from(cen));
    qz.values.multiScalarQuantize = { seeds: [931, 932], dim: 24, bits: [1, 4, 7], destinations: d.map(arr), results: rs };
    [['multi length mismatch', function () { qn.multiScalarQuantize(v, d, [1, 4], Float32Array.from(cen)); }],
      ['transpose null', function () { OSQ.transposeHalfByte(null, new Uint8Array(4)); }],
      ['transpose length', function () { OSQ.transposeHalfByte(new Uint8Array(3), new Uint8Array(4)); }],
      ['transpose value', function () { OSQ.transposeHalfByte(new Uint8Array([1, 16]), new Uint8Array(8)); }]].forEach(function (p) {
      try { p[1](); qz.errors[p[0]] = null; } catch (e) { qz.errors[p[0]] = String(e.message); }
    }); }
  fs.writeFileSync(path.join(OUT, 'api_behaviour.json'), JSON.stringify(out, null, 1));
  console.log('api_behaviour ok');
}

// ---------------------------------------------------------------- case list
const SIMS = ['EUCLIDEAN', 'COSINE', 'MAXIMUM_INNER_PRODUCT'];
const cases = [];
// C1: BASELINE config 1 (SURVEY App. C known answers)
cases.push({ name: 'c1_1000x128_cos_qb4', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 128, n: 1000, k: 10, ks: [10, 100], nq: 3, full: true,
  gen: { kind: 'mulberry32', base_seed: 3, query_seed: 4 } });
// similarity x queryBits x dims, incl. dim not a multiple of 8 (100) and of 64 (72)
let seed = 100;
[[64, 300], [100, 257], [72, 130], [768, 200]].forEach(function (dn) {
  SIMS.forEach(function (sim) {
    [1, 4].forEach(function (qb) {
      cases.push({ name: 'm_' + dn[0] + 'd_' + sim.slice(0, 3).toLowerCase() + '_qb' + qb, sim: sim, qb: qb, ib: 1, lambda: 0.1, iters: 5,
        dim: dn[0], n: dn[1], k: 10, ks: [1, 10, 100], nq: 2, full: true, gen: { kind: 'mulberry32', base_seed: seed, query_seed: seed + 1 } });
      seed += 2;
    });
  });
});
// other query widths take the "4-bit" branch (A.5-3)
[2, 3, 8].forEach(function (qb) {
  cases.push({ name: 'qb' + qb + '_128d_cos', sim: 'COSINE', qb: qb, ib: 1, lambda: 0.1, iters: 5, dim: 128, n: 300, k: 10, nq: 2, full: true,
    gen: { kind: 'mulberry32', base_seed: 60 + qb, query_seed: 70 + qb } });
  cases.push({ name: 'qb' + qb + '_96d_mip', sim: 'MAXIMUM_INNER_PRODUCT', qb: qb, ib: 1, lambda: 0.1, iters: 5, dim: 96, n: 200, k: 10, nq: 1, full: true,
    gen: { kind: 'mulberry32', base_seed: 80 + qb, query_seed: 90 + qb } });
});
// heap tie stress: rows drawn from a pool of 40 vectors -> massive exact score ties
SIMS.forEach(function (sim, i) {
  cases.push({ name: 'ties_' + sim.slice(0, 3).toLowerCase() + '_qb4', sim: sim, qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 64, n: 3000, k: 100, ks: [1, 7, 100, 500], nq: 3, full: false,
    gen: { kind: 'dup_pool', base_seed: 200 + i, pick_seed: 210 + i, query_seed: 220 + i, pool: 40 } });
});
cases.push({ name: 'ties_cos_qb1', sim: 'COSINE', qb: 1, ib: 1, lambda: 0.1, iters: 5, dim: 64, n: 3000, k: 100, ks: [10, 100], nq: 2, full: false,
  gen: { kind: 'dup_pool', base_seed: 230, pick_seed: 231, query_seed: 232, pool: 40 } });
// short vectors: 1-bit x 1-bit on dim 16 gives few distinct scores -> ties without duplicates
cases.push({ name: 'ties_16d_qb1', sim: 'COSINE', qb: 1, ib: 1, lambda: 0.1, iters: 5, dim: 16, n: 4000, k: 50, ks: [50, 300], nq: 3, full: false,
  gen: { kind: 'mulberry32', base_seed: 240, query_seed: 241 } });
// the reference's closed-form recall dataset (tests/recall.test.ts), its lambda/iters
cases.push({ name: 'closed_100x128_qb4', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.001, iters: 20, dim: 128, n: 100, k: 10, nq: 10, full: true, oversample: 3, gen: { kind: 'closed_form' } });
cases.push({ name: 'closed_100x128_qb1', sim: 'COSINE', qb: 1, ib: 1, lambda: 0.001, iters: 20, dim: 128, n: 100, k: 10, nq: 10, full: true, gen: { kind: 'closed_form' } });
// debug_scorer.ts scenario (SURVEY App. C)
cases.push({ name: 'debug_scorer_3x4', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.01, iters: 10, dim: 4, n: 3, k: 3, nq: 1, full: true,
  gen: { kind: 'inline', base: [[1, 0, 0, 0], [0.9, 0.1, 0, 0], [0, 1, 0, 0]].map(function (r) { return Array.from(normalizeVector(new Float32Array(r))); }),
    queries: [Array.from(normalizeVector(new Float32Array([1, 0, 0, 0])))] } });
// edge shapes: N=1, k>N, zero vector, constant vector (degenerate interval), dim 1
cases.push({ name: 'edge_n1', sim: 'EUCLIDEAN', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 8, n: 1, k: 5, nq: 1, full: true,
  gen: { kind: 'inline', base: [[0.5, -0.25, 0.125, 1, -1, 0.75, 0.3, -0.6]], queries: [[0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]] } });
cases.push({ name: 'edge_zero_const', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 8, n: 5, k: 5, nq: 2, full: true,
  gen: { kind: 'inline', base: [[0, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 1], [1, 2, 3, 4, 5, 6, 7, 8], [-1, 2, -3, 4, -5, 6, -7, 8], [2, 2, 2, 2, 2, 2, 2, 2]],
    queries: [[1, 2, 3, 4, 5, 6, 7, 8], [0, 0, 0, 0, 0, 0, 0, 0]] } });
cases.push({ name: 'edge_zero_const_euc', sim: 'EUCLIDEAN', qb: 1, ib: 1, lambda: 0.1, iters: 5, dim: 8, n: 5, k: 3, nq: 2, full: true,
  gen: { kind: 'inline', base: [[0, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 1], [1, 2, 3, 4, 5, 6, 7, 8], [-1, 2, -3, 4, -5, 6, -7, 8], [2, 2, 2, 2, 2, 2, 2, 2]],
    queries: [[1, 2, 3, 4, 5, 6, 7, 8], [0, 0, 0, 0, 0, 0, 0, 0]] } });
cases.push({ name: 'edge_dim1', sim: 'MAXIMUM_INNER_PRODUCT', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 1, n: 6, k: 3, nq: 1, full: true,
  gen: { kind: 'inline', base: [[0.5], [-0.5], [2], [3], [-7], [0.25]], queries: [[1.5]] } });
// larger heap-replay stress, hashes only
cases.push({ name: 'big_20000x128_cos', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 128, n: 20000, k: 100, ks: [100], nq: 2, full: false,
  gen: { kind: 'mulberry32', base_seed: 31, query_seed: 32 } });
cases.push({ name: 'big_50000x768_cos', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 768, n: 50000, k: 100, ks: [100], nq: 2, full: false,
  gen: { kind: 'mulberry32', base_seed: 11, query_seed: 12 } });
cases.push({ name: 'big_30000x1536_mip', sim: 'MAXIMUM_INNER_PRODUCT', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 1536, n: 30000, k: 100, ks: [100], nq: 1, full: false,
  gen: { kind: 'mulberry32', base_seed: 41, query_seed: 42 } });

// dim 1024 (BASELINE config 5's width): the compile-time 8-chunk kernel instantiations
cases.push({ name: 'm_1024d_cos_qb4', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 1024, n: 200, k: 10, ks: [1, 10, 100], nq: 2, full: true,
  gen: { kind: 'mulberry32', base_seed: 401, query_seed: 402 } });
cases.push({ name: 'm_1024d_euc_qb1', sim: 'EUCLIDEAN', qb: 1, ib: 1, lambda: 0.1, iters: 5, dim: 1024, n: 200, k: 10, ks: [1, 10, 100], nq: 2, full: true,
  gen: { kind: 'mulberry32', base_seed: 403, query_seed: 404 } });
cases.push({ name: 'm_1024d_max_qb8', sim: 'MAXIMUM_INNER_PRODUCT', qb: 8, ib: 1, lambda: 0.1, iters: 5, dim: 1024, n: 200, k: 10, ks: [1, 10, 100], nq: 2, full: true,
  gen: { kind: 'mulberry32', base_seed: 405, query_seed: 406 } });
cases.push({ name: 'big_20000x1024_cos', sim: 'COSINE', qb: 4, ib: 1, lambda: 0.1, iters: 5, dim: 1024, n: 20000, k: 100, ks: [100], nq: 2, full: false,
  gen: { kind: 'mulberry32', base_seed: 407, query_seed: 408 } });
cases.push({ name: 'big_20000x1024_euc_qb8', sim: 'EUCLIDEAN', qb: 8, ib: 1, lambda: 0.1, iters: 5, dim: 1024, n: 20000, k: 100, ks: [100], nq: 1, full: false,
  gen: { kind: 'mulberry32', base_seed: 409, query_seed: 410 } });
// multi-bit index (indexBits > 1): rows are unpacked bytes (src/binaryQuantizationFormat.ts:241-245), the batch scorer throws on
// them and the reference answers through its per-row fallback (src/binaryQuantizedScorer.ts:403-419, :69-301) for queryBits 1
// and 4 (centroidDP = 0 for 4-bit queries, SURVEY A.7); every other queryBits makes searchNearestNeighbors throw
seed = 500;
[[2, 64, 300], [2, 100, 257], [4, 96, 200], [3, 72, 130], [8, 64, 150], [2, 1024, 120]].forEach(function (bdn) {
  SIMS.forEach(function (sim) {
    [1, 4].forEach(function (qb) {
      if (bdn[0] !== 2 && !(sim === 'COSINE' && qb === 4) && !(sim === 'MAXIMUM_INNER_PRODUCT' && qb === 1) && !(sim === 'EUCLIDEAN' && qb === 4 && bdn[0] === 4)) { seed += 2; return; }
      if (bdn[1] === 1024 && !((sim === 'COSINE' && qb === 4) || (sim === 'EUCLIDEAN' && qb === 1))) { seed += 2; return; }
      cases.push({ name: 'ib' + bdn[0] + '_' + bdn[1] + 'd_' + sim.slice(0, 3).toLowerCase() + '_qb' + qb, sim: sim, qb: qb, ib: bdn[0], lambda: 0.1, iters: 5,
        dim: bdn[1], n: bdn[2], k: 10, ks: [1, 10, 100], nq: 2, full: true, gen: { kind: 'mulberry32', base_seed: seed, query_seed: seed + 1 } });
      seed += 2;
    });
  });
});
cases.push({ name: 'ib2_ties_cos_qb4', sim: 'COSINE', qb: 4, ib: 2, lambda: 0.1, iters: 5, dim: 64, n: 3000, k: 100, ks: [1, 7, 100, 500], nq: 2, full: false,
  gen: { kind: 'dup_pool', base_seed: 600, pick_seed: 601, query_seed: 602, pool: 40 } });
cases.push({ name: 'ib2_big_20000x128_euc', sim: 'EUCLIDEAN', qb: 4, ib: 2, lambda: 0.1, iters: 5, dim: 128, n: 20000, k: 100, ks: [100], nq: 1, full: false,
  gen: { kind: 'mulberry32', base_seed: 603, query_seed: 604 } });
// queryBits the fallback does not know: the reference throws (recorded as such)
cases.push({ name: 'ib2_64d_cos_qb8_throws', sim: 'COSINE', qb: 8, ib: 2, lambda: 0.1, iters: 5, dim: 64, n: 50, k: 5, nq: 1, full: true,
  gen: { kind: 'mulberry32', base_seed: 605, query_seed: 606 } });
// dim 1: the one width where the batch scorer does NOT throw on unpacked rows (dim == ceil(dim/8))
cases.push({ name: 'ib2_edge_dim1', sim: 'MAXIMUM_INNER_PRODUCT', qb: 4, ib: 2, lambda: 0.1, iters: 5, dim: 1, n: 6, k: 3, nq: 1, full: true,
  gen: { kind: 'inline', base: [[0.5], [-0.5], [2], [3], [-7], [0.25]], queries: [[1.5]] } });

const only = process.env.BBQ_ONLY ? new RegExp(process.env.BBQ_ONLY) : null;
fs.mkdirSync(OUT, { recursive: true });
cases.forEach(function (c) { if (!only || only.test(c.name)) runCase(c); });
if (!only || only.test('intdot')) {
  runIntDot({ name: 'intdot_ib2_qb8_64d', sim: 'EUCLIDEAN', qb: 8, ib: 2, dim: 64, n: 200, nq: 2, base_seed: 51, query_seed: 52 });
  runIntDot({ name: 'intdot_ib2_qb8_1024d', sim: 'COSINE', qb: 8, ib: 2, dim: 1024, n: 120, nq: 2, base_seed: 55, query_seed: 56 });
  runIntDot({ name: 'intdot_ib4_qb8_96d', sim: 'MAXIMUM_INNER_PRODUCT', qb: 8, ib: 4, dim: 96, n: 120, nq: 2, base_seed: 57, query_seed: 58 });
  runIntDot({ name: 'intdot_ib8_qb8_72d', sim: 'EUCLIDEAN', qb: 8, ib: 8, dim: 72, n: 100, nq: 2, base_seed: 59, query_seed: 60 });
  runIntDot({ name: 'intdot_ib2_qb4_100d', sim: 'COSINE', qb: 4, ib: 2, dim: 100, n: 150, nq: 2, base_seed: 53, query_seed: 54 });
}
if (!only || only.test('rerank')) {
  runRerank({ name: 'rerank_768d', dim: 768, n: 300, nq: 3, k: 10, factors: [1, 3, 5], base_seed: 301, query_seed: 302, zero_query: true });
  runRerank({ name: 'rerank_100d', dim: 100, n: 500, nq: 4, k: 25, factors: [2, 4], base_seed: 303, query_seed: 304, zero_query: false });
  runRerank({ name: 'rerank_3d', dim: 3, n: 40, nq: 2, k: 5, factors: [3, 10], base_seed: 305, query_seed: 306, zero_query: false });
}
if (!only || only.test('api_behaviour')) runApiBehaviour();
