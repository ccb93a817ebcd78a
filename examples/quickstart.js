'use strict';
// The reference README's "basic usage" (README.md:26-51 of leolee9086/Better-Binary-Quantization), unchanged except for the
// import: the same calls run on the MI355X through libbbq.  node examples/quickstart.js
const path = require('path');
const {
  createBinaryQuantizationFormat, quickQuantize, quickSearch, VectorSimilarityFunction,
  getOversampledTopKWithHeap, createDeviceVectors,
} = require(path.join(__dirname, '..', 'better-binary-quantization_amd', 'js'));

const format = createBinaryQuantizationFormat();

const vectors = [
  new Float32Array([1, 2, 3, 4]),
  new Float32Array([5, 6, 7, 8]),
  new Float32Array([9, 10, 11, 12]),
];

const { quantizedVectors, queryQuantizer } = quickQuantize(vectors);
console.log('quantized', quantizedVectors.size(), 'vectors of dimension', quantizedVectors.dimension(), '- quantizer lambda', queryQuantizer.lambda);

const queryVector = new Float32Array([1, 2, 3, 4]);
const results = quickSearch(queryVector, vectors, 2);
console.log(results);

// a larger collection: build once, search many times, rerank exactly
const n = 20000, dim = 256;
let seed = 7;
const rnd = function () { seed = (seed * 1664525 + 1013904223) >>> 0; return seed / 4294967296 - 0.5; };
const base = [];
for (let i = 0; i < n; i++) { const v = new Float32Array(dim); for (let j = 0; j < dim; j++) v[j] = rnd(); base.push(v); }
const fmt = createBinaryQuantizationFormat({ queryBits: 4, indexBits: 1, quantizer: { similarityFunction: VectorSimilarityFunction.COSINE, lambda: 0.1, iters: 5 } });
const index = fmt.quantizeVectors(base).quantizedVectors;          // HIP kernels; the index stays in HBM
const top = fmt.searchNearestNeighbors(base[123], index, 5);
console.log('top-5 of row 123:', top.map(function (r) { return r.index; }).join(' '));
const resident = createDeviceVectors(base);                        // fp32 rows resident for the exact rerank
const reranked = getOversampledTopKWithHeap(base[123], index, resident, 5, 3, fmt);
console.log('after 3x oversample + exact rerank:', reranked.map(function (r) { return r.index + ':' + r.trueScore.toFixed(4); }).join(' '));
quantizedVectors.dispose(); index.dispose(); resident.dispose(); format.getConfig();
if (results.length !== 2 || results[0].index !== 0 || top[0].index !== 123 || reranked[0].index !== 123) process.exit(1);
