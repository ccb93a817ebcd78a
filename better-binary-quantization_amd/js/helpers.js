'use strict';
/*
 * The small host-side helpers and constants the reference re-exports from its package root (src/index.ts:20-37:
 * `export *` of constants, utils, vectorOperations, vectorUtils).  None of them is on the search path - they are plain
 * arithmetic on a handful of numbers - but a drop-in must not break `import { clamp } from ...`.  Values and error messages are
 * pinned against the reference by the `helpers` entries of tests/golden/api_behaviour.json.
 */
const NUMERICAL_CONSTANTS = Object.freeze({ CONVERGENCE_THRESHOLD: 1e-8, MIN_DETERMINANT: 1e-12, EPSILON: 1e-8 });  // src/constants.ts:70-77
const FILE_EXTENSIONS = Object.freeze({ VECTOR_DATA: 'veb', META: 'vemb' });                                        // src/constants.ts:52-57
const COMPONENT_NAMES = Object.freeze({ BINARIZED_VECTOR: 'BVEC' });                                                // src/constants.ts:62-65
// src/constants.ts:38-47: initial interval per bit width, in standard deviations
const MINIMUM_MSE_GRID = [0.798, 1.493, 2.051, 2.514, 2.916, 3.278, 3.611, 3.922].map(function (g) { return [-g, g]; });

// src/utils.ts:9-18
const BIT_COUNT_LOOKUP_TABLE = Uint8Array.from({ length: 256 }, function (_, v) { let c = 0; for (; v; v >>>= 1) c += v & 1; return c; });

function sumOfSquares(v) { let s = 0; for (let i = 0; i < v.length; i++) s += v[i] * v[i]; return s; }
const computeL2Norm = function (vector) { return Math.sqrt(sumOfSquares(vector)); };            // src/utils.ts:25-34
const computeVectorMagnitude = function (vector) { return Math.sqrt(sumOfSquares(vector)); };   // src/vectorUtils.ts:11-20
function computeMean(vector) {                                                                  // src/utils.ts:41-50
  let s = 0;
  for (let i = 0; i < vector.length; i++) s += vector[i];
  return s / vector.length;
}
function computeStd(vector, mean) {                                                             // src/utils.ts:58-68
  let s = 0;
  for (let i = 0; i < vector.length; i++) { const d = vector[i] - mean; s += d * d; }
  return Math.sqrt(s / vector.length);
}
const clamp = function (x, min, max) { return Math.min(Math.max(x, min), max); };               // src/utils.ts:79-81
function bitCount(n) {                                                                          // src/utils.ts:89-97 (SWAR)
  n >>>= 0;
  n -= (n >>> 1) & 0x55555555;
  n = (n & 0x33333333) + ((n >>> 2) & 0x33333333);
  n = (n + (n >>> 4)) & 0x0F0F0F0F;
  n += n >>> 8;
  n += n >>> 16;
  return n & 0x3F;
}
function bitCountBytes(bytes) {                                                                 // src/utils.ts:106-115
  let c = 0;
  for (let i = 0; i < bytes.length; i++) c += BIT_COUNT_LOOKUP_TABLE[bytes[i]];
  return c;
}
const getBitCount = function (byte) { return BIT_COUNT_LOOKUP_TABLE[byte & 0xFF]; };            // src/utils.ts:140-142
function isNearZero(value, threshold) {                                                         // src/utils.ts:150-152
  return Math.abs(value) < (threshold === undefined ? NUMERICAL_CONSTANTS.CONVERGENCE_THRESHOLD : threshold);
}
function isNearEqual(a, b, epsilon) {                                                           // src/utils.ts:161-163
  return Math.abs(a - b) < (epsilon === undefined ? NUMERICAL_CONSTANTS.EPSILON : epsilon);
}
const scaleMaxInnerProductScore = function (score) { return score < 0 ? 1 / (1 - score) : score + 1; };  // src/utils.ts:171-176

// element-wise helpers store through a Float32Array like the reference does (src/vectorOperations.ts:42-120, :193-195)
function zip(a, b, message, f) {
  if (a.length !== b.length) throw new Error(message);
  const out = new Float32Array(a.length);
  for (let i = 0; i < a.length; i++) out[i] = f(a[i], b[i]);
  return out;
}
const addVectors = function (a, b) { return zip(a, b, '向量维度不匹配', function (x, y) { return x + y; }); };
const subtractVectors = function (a, b) { return zip(a, b, '向量维度不匹配', function (x, y) { return x - y; }); };
const centerVector = function (vector, centroid) { return zip(vector, centroid, '向量和质心维度不匹配', function (x, y) { return x - y; }); };
const scaleVector = function (vector, scalar) { return Float32Array.from(vector, function (x) { return x * scalar; }); };
const copyVector = function (vector) { return new Float32Array(vector); };
function createRandomVector(dimension, min, max) {                                              // src/vectorUtils.ts:29-35
  const lo = min === undefined ? -1 : min, hi = max === undefined ? 1 : max;
  return Float32Array.from({ length: dimension }, function () { return Math.random() * (hi - lo) + lo; });
}
const createZeroVector = function (dimension) { return new Float32Array(dimension); };          // src/vectorUtils.ts:42-44

module.exports = {
  NUMERICAL_CONSTANTS, FILE_EXTENSIONS, COMPONENT_NAMES, MINIMUM_MSE_GRID, BIT_COUNT_LOOKUP_TABLE,
  computeL2Norm, computeMean, computeStd, clamp, bitCount, bitCountBytes, bitCountBytesOptimized: bitCountBytes, getBitCount,
  isNearZero, isNearEqual, scaleMaxInnerProductScore, addVectors, subtractVectors, scaleVector, centerVector, copyVector,
  computeVectorMagnitude, createRandomVector, createZeroVector,
};
