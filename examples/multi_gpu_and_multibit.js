'use strict';
// Two things the drop-in adds without changing the reference's API:
//   1. several GPUs behind the same objects:  BBQ_DEVICES=8 node examples/multi_gpu_and_multibit.js   (a list works too:
//      BBQ_DEVICES=0,0 puts two shards on GPU 0 - how a single-GPU box exercises the path)
//   2. multi-bit indexes (indexBits 2..8) scanned on the device; answers = what the reference returns for them
const bbq = require('../better-binary-quantization_amd/js');

function randomVectors(n, dim, seed) {
  let a = seed | 0;
  const next = function () { a = a + 0x6D2B79F5 | 0; let t = Math.imul(a ^ a >>> 15, 1 | a); t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t; return ((t ^ t >>> 14) >>> 0) / 4294967296; };
  const out = [];
  for (let i = 0; i < n; i++) { const v = new Float32Array(dim); for (let j = 0; j < dim; j++) v[j] = 2 * next() - 1; out.push(v); }
  return out;
}

const base = randomVectors(20000, 256, 1), queries = randomVectors(8, 256, 2);

// the reference's default configuration (4-bit queries, 1-bit index)
const fmt = bbq.createBinaryQuantizationFormat();
const index = fmt.quantizeVectors(base).quantizedVectors;           // quantized on the device
const one = fmt.searchNearestNeighbors(queries[0], index, 10);      // one synchronous call, as in the reference
const many = fmt.searchNearestNeighborsBatch(queries, index, 10);   // extension: pipelined on the device
console.log('1-bit index, shards:', index.deviceStats().shards, 'top hit', one[0], 'batch agrees:', JSON.stringify(many[0]) === JSON.stringify(one));

// a 2-bit index with 4-bit queries: the reference answers it through its per-row fallback (with a warning per batch); same results here
const fmt2 = new bbq.BinaryQuantizationFormat({ queryBits: 4, indexBits: 2, quantizer: { similarityFunction: bbq.VectorSimilarityFunction.COSINE } });
const index2 = fmt2.quantizeVectors(base).quantizedVectors;
const warn = console.warn; console.warn = function () {};
const hits2 = fmt2.searchNearestNeighbors(queries[0], index2, 10);
console.warn = warn;
console.log('2-bit index, bytes per row on the device:', index2.deviceStats().bytesPerRow, 'top hit', hits2[0]);
index.dispose(); index2.dispose();
