// TypeScript surface of the MI355X drop-in (mirrors the reference's src/index.ts exports used on the search path).
export declare enum VectorSimilarityFunction { EUCLIDEAN = 'EUCLIDEAN', COSINE = 'COSINE', MAXIMUM_INNER_PRODUCT = 'MAXIMUM_INNER_PRODUCT' }
export interface QuantizationResult { lowerInterval: number; upperInterval: number; additionalCorrection: number; quantizedComponentSum: number; }
export interface QuantizerConfig { similarityFunction: VectorSimilarityFunction; lambda?: number; iters?: number; }
export interface BinaryQuantizationConfig { queryBits?: number; indexBits?: number; quantizer: QuantizerConfig; }
export interface BinarizedByteVectorValues {
  dimension(): number;
  vectorValue(ord: number): Uint8Array;
  getUnpackedVector(ord: number): Uint8Array;
  getCorrectiveTerms(ord: number): QuantizationResult;
  getCentroidDP(queryVector?: Float32Array): number;
  getCentroid(): Float32Array;
  size(): number;
  clearUnpackedVectorCache?(): void;
  /** releases the device-resident copy (extension) */
  dispose(): void;
  /** libbbq tuning knob on the device index, e.g. ('sweep_share', 32), ('resident_mb', 0) (extension; include/bbq.h lists them) */
  setDeviceOption(name: string, value: number): void;
  /** libbbq counters of the last call on the device index (extension; bbq_stats in include/bbq.h) */
  deviceStats(): { lastScanMs: number; lastScanBytes: number; candidates: number; denseFallbacks: number; hostReplays: number;
                   residentBytes: number; shards: number; bytesPerRow: number };
}
export interface QuantizedScoreResult { score: number; bitDotProduct: number; corrections: { query: QuantizationResult; index: QuantizationResult }; }
export declare class OptimizedScalarQuantizer {
  constructor(config: QuantizerConfig);
  scalarQuantize(vector: Float32Array, destination: Uint8Array, bits: number, centroid: Float32Array): QuantizationResult;
  multiScalarQuantize(vector: Float32Array, destinations: Uint8Array[], bits: number[], centroid: Float32Array): QuantizationResult[];
  static packAsBinary(vector: Uint8Array, packed: Uint8Array): void;
  static discretize(value: number, bucket: number): number;
  /** one byte per bit: quantQueryByte.length === q.length * 4 */
  static transposeHalfByte(q: Uint8Array, quantQueryByte: Uint8Array): void;
  static transposeHalfByteOptimized(q: Uint8Array, quantQueryByte: Uint8Array, useCache?: boolean): void;
  /** packed planes: quantQueryByte.length === ceil(q.length / 8) * 4 */
  static transposeHalfByteFast(q: Uint8Array, quantQueryByte: Uint8Array): void;
  static clearTransposeCache(): void;
  static getTransposeCacheStats(): { size: number; hitRate: number };
}
export declare class BinaryQuantizedScorer {
  constructor(similarityFunction: VectorSimilarityFunction);
  /** the single-row path of the reference (src/binaryQuantizedScorer.ts:69-301): host arithmetic, queryBits 1 or 4 */
  computeQuantizedScore(quantizedQuery: Uint8Array, queryCorrections: QuantizationResult, targetVectors: BinarizedByteVectorValues, targetOrd: number, queryBits: number, originalQueryVector?: Float32Array): QuantizedScoreResult;
  computeBatchQuantizedScores(quantizedQuery: Uint8Array, queryCorrections: QuantizationResult, targetVectors: BinarizedByteVectorValues,
    targetOrds: number[], queryBits: number, originalQueryVector?: Float32Array): QuantizedScoreResult[];
  /** src/binaryQuantizedScorer.ts:429-617 */
  computeOriginalScore(originalQuery: Float32Array, targetVector: Float32Array, similarityFunction: VectorSimilarityFunction): number;
  compareScores(originalScore: number, quantizedScore: number): { difference: number; relativeError: number; correlation: number };
  computeQuantizationAccuracy(originalScores: number[], quantizedScores: number[]): { meanError: number; maxError: number; minError: number; stdError: number; correlation: number };
  getSimilarityFunction(): VectorSimilarityFunction;
}
export declare class BinaryQuantizationFormat {
  constructor(config: BinaryQuantizationConfig);
  quantizeVectors(vectors: Float32Array[]): { quantizedVectors: BinarizedByteVectorValues; queryQuantizer: OptimizedScalarQuantizer };
  quantizeQueryVector(queryVector: Float32Array, centroid: Float32Array): { quantizedQuery: Uint8Array; queryCorrections: QuantizationResult };
  searchNearestNeighbors(queryVector: Float32Array, targetVectors: BinarizedByteVectorValues, k: number): Array<{ index: number; score: number }>;
  /** extension: many independent queries per call, pipelined on the device */
  searchNearestNeighborsBatch(queryVectors: Float32Array[], targetVectors: BinarizedByteVectorValues, k: number): Array<Array<{ index: number; score: number }>>;
  /** src/binaryQuantizationFormat.ts:483-566 with the double-pack bug fixed: binaryValues is the packed row */
  serializeVectorData(vectors: Float32Array[]): { vectorData: VectorDataFormat[]; metadata: MetadataFormat };
  deserializeVectorData(vectorData: VectorDataFormat[], metadata: MetadataFormat): BinarizedByteVectorValues;
  /** extension: <prefix>.veb (device tile records) + <prefix>.vemb (MetadataFormat + centroid) */
  saveIndex(quantizedVectors: BinarizedByteVectorValues, pathPrefix: string): void;
  loadIndex(pathPrefix: string): BinarizedByteVectorValues;
  /** src/binaryQuantizationFormat.ts:420-476: every query against vector 0, quantized vs fp32 score */
  computeQuantizationAccuracy(originalVectors: Float32Array[], queryVectors: Float32Array[]): { meanError: number; maxError: number; minError: number; stdError: number; correlation: number };
  getConfig(): BinaryQuantizationConfig;
  getQuantizer(): OptimizedScalarQuantizer;
  getScorer(): BinaryQuantizedScorer;
}
export interface VectorDataFormat { binaryValues: Uint8Array; lowerInterval: number; upperInterval: number; additionalCorrection: number; quantizedComponentSum: number; }
export interface MetadataFormat { fieldNumber: number; vectorEncodingOrdinal: number; vectorSimilarityOrdinal: number; dimensions: number; vectorDataOffset: number; vectorDataLength: number; vectorCount: number; centroid: Float32Array; centroidSquareMagnitude: number; }
export interface SiftVector { dimension: number; values: Float32Array; }
export interface SiftDataset { vectors: SiftVector[]; count: number; dimension: number; }
export declare function loadSiftVectors(filePath: string, maxVectors?: number): SiftDataset;
export declare function loadSiftDataset(datasetDir: string, fileType?: 'base' | 'learn' | 'query', maxVectors?: number): SiftDataset;
export declare function loadSiftQueries(datasetDir: string, maxQueries?: number): { queries: SiftVector[]; groundtruth: number[][] };
export declare function normalizeVector(vector: Float32Array): Float32Array;
export declare function computeCentroid(vectors: Float32Array[]): Float32Array;
export declare function computeDotProduct(a: Float32Array, b: Float32Array): number;
export declare function computeEuclideanDistance(a: Float32Array, b: Float32Array): number;
export declare function computeEuclideanSimilarity(a: Float32Array, b: Float32Array): number;
export declare function computeCosineSimilarity(a: Float32Array, b: Float32Array): number;
export declare function computeMaximumInnerProduct(a: Float32Array, b: Float32Array): number;
export declare function computeSimilarity(a: Float32Array, b: Float32Array, similarityFunction: 'EUCLIDEAN' | 'COSINE' | 'MAXIMUM_INNER_PRODUCT'): number;
export declare function computeQuantizedDotProduct(q: Uint8Array, d: Uint8Array): number;
export declare function computeInt4BitDotProduct(q: Uint8Array, d: Uint8Array): number;
export declare function computeInt1BitDotProduct(q: Uint8Array, d: Uint8Array): number;
export interface TopKCandidate { index: number; quantizedScore: number; trueScore: number; }
/** the original fp32 vectors resident on the GPU; accepted wherever the selectors take `vectors` */
export declare class DeviceVectors {
  constructor(vectors: Float32Array[], device?: number);
  readonly length: number;
  readonly dim: number;
  /** computeSimilarity(query, vectors[rows[j]]) (f64, bit-identical to src/vectorSimilarity.ts); default COSINE */
  trueScores(query: Float32Array, rows: ArrayLike<number>, similarityFunction?: VectorSimilarityFunction): Float64Array;
  dispose(): void;
}
export declare function createDeviceVectors(vectors: Float32Array[], device?: number): DeviceVectors;
export declare function getOversampledTopKWithHeap(query: Float32Array, quantizedVectors: any, vectors: Float32Array[] | DeviceVectors, k: number, oversampleFactor: number, format: BinaryQuantizationFormat): TopKCandidate[];
export declare function getOversampledTopKWithSort(query: Float32Array, quantizedVectors: any, vectors: Float32Array[] | DeviceVectors, k: number, oversampleFactor: number, format: BinaryQuantizationFormat): TopKCandidate[];
/** extension: the whole recipe for many queries in one native call */
export declare function getOversampledTopKBatch(queries: Float32Array[], quantizedVectors: any, vectors: DeviceVectors, k: number, oversampleFactor: number, format: BinaryQuantizationFormat, selector?: 'heap' | 'sort'): TopKCandidate[][];
export declare const DEFAULT_CONFIG: { readonly queryBits: 4; readonly indexBits: 1; readonly quantizer: { readonly similarityFunction: VectorSimilarityFunction.COSINE; readonly lambda: 0.1; readonly iters: 5 } };
export declare function createBinaryQuantizationFormat(config?: BinaryQuantizationConfig): BinaryQuantizationFormat;
export declare function quickQuantize(vectors: Float32Array[], similarityFunction?: VectorSimilarityFunction): { quantizedVectors: BinarizedByteVectorValues; queryQuantizer: OptimizedScalarQuantizer };
export declare function quickSearch(queryVector: Float32Array, targetVectors: Float32Array[], k: number, similarityFunction?: VectorSimilarityFunction): Array<{ index: number; score: number }>;
export declare const VERSION: string;
export declare function deviceCount(): number;

// helpers and constants the reference re-exports from its root (src/index.ts:20-37)
export declare const NUMERICAL_CONSTANTS: { readonly CONVERGENCE_THRESHOLD: 1e-8; readonly MIN_DETERMINANT: 1e-12; readonly EPSILON: 1e-8 };
export declare const FILE_EXTENSIONS: { readonly VECTOR_DATA: 'veb'; readonly META: 'vemb' };
export declare const COMPONENT_NAMES: { readonly BINARIZED_VECTOR: 'BVEC' };
export declare const MINIMUM_MSE_GRID: number[][];
export declare const BIT_COUNT_LOOKUP_TABLE: Uint8Array;
export declare function computeL2Norm(vector: Float32Array): number;
export declare function computeMean(vector: Float32Array): number;
export declare function computeStd(vector: Float32Array, mean: number): number;
export declare function clamp(x: number, min: number, max: number): number;
export declare function bitCount(n: number): number;
export declare function bitCountBytes(bytes: Uint8Array): number;
export declare function bitCountBytesOptimized(bytes: Uint8Array): number;
export declare function getBitCount(byte: number): number;
export declare function isNearZero(value: number, threshold?: number): boolean;
export declare function isNearEqual(a: number, b: number, epsilon?: number): boolean;
export declare function scaleMaxInnerProductScore(score: number): number;
export declare function addVectors(a: Float32Array, b: Float32Array): Float32Array;
export declare function subtractVectors(a: Float32Array, b: Float32Array): Float32Array;
export declare function scaleVector(vector: Float32Array, scalar: number): Float32Array;
export declare function centerVector(vector: Float32Array, centroid: Float32Array): Float32Array;
export declare function copyVector(vector: Float32Array): Float32Array;
export declare function computeVectorMagnitude(vector: Float32Array): number;
export declare function createRandomVector(dimension: number, min?: number, max?: number): Float32Array;
export declare function createZeroVector(dimension: number): Float32Array;
export interface QuantizationAccuracy { meanError: number; maxError: number; minError: number; stdError: number; correlation: number; }
/** src/index.ts:120-134: computeQuantizationAccuracy of a format with lambda 0.1 / 5 iterations */
export declare function computeAccuracy(originalVectors: Float32Array[], queryVectors: Float32Array[], similarityFunction?: VectorSimilarityFunction): QuantizationAccuracy;
