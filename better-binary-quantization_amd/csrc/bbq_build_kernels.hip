// bbq_build_kernels.hip - index build on the device (gfx950): BinaryQuantizationFormat.quantizeVectors
// (reference src/binaryQuantizationFormat.ts:165-263) for every indexBits, bit-exact.
//
//   transpose_in   [n][dim] f32 (as uploaded)  ->  vT4[dim/4][npad] float4   (lane = vector => every later access is coalesced)
//   normalize      COSINE: normalizeVector (src/vectorOperations.ts:11-34), one thread per vector, f64 sum in index order
//   validate       first NaN / Infinity in row-major order (src/binaryQuantizationFormat.ts:196-211)
//   centroid       computeCentroid (src/vectorOperations.ts:126-163): the Float32Array accumulator is rounded after EVERY
//                  += and the sum runs over the vectors in order, so it is a serial chain per dimension; one wave per
//                  4 dimensions streams its vT4 row with coalesced 1 KiB loads and walks the 64 values through LDS broadcasts
//   quantize       OptimizedScalarQuantizer.scalarQuantize (src/optimizedScalarQuantizer.ts:108-227, getInitialInterval
//                  :245-265, optimizeIntervals :280-353, computeLoss :373-407), one thread per vector, every reduction in
//                  the reference's index order, f64 without FMA contraction; the last pass packs the bits
//                  (packAsBinary :420-446) straight into the scan kernel's tile records and writes the corrections
//   untile         tile records -> row-major packed rows (only when the host asks for the codes)
//
// All arithmetic follows the JavaScript number model; -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <float.h>
#include "bbq_device.h"
#include "bbq_launch.h"

#pragma clang fp contract(off)

namespace bbq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2b __attribute__((ext_vector_type(2)));
typedef double f64x2b __attribute__((ext_vector_type(2)));

// Math.min / Math.max / Math.round as V8 evaluates them
__device__ __forceinline__ double jmin(double a, double b) {
  if (a != a || b != b) return __longlong_as_double(0x7ff8000000000000ll);
  if (a == 0.0 && b == 0.0) return (__double_as_longlong(a) < 0 || __double_as_longlong(b) < 0) ? -0.0 : 0.0;
  return a < b ? a : b;
}
__device__ __forceinline__ double jmax(double a, double b) {
  if (a != a || b != b) return __longlong_as_double(0x7ff8000000000000ll);
  if (a == 0.0 && b == 0.0) return (__double_as_longlong(a) < 0 && __double_as_longlong(b) < 0) ? -0.0 : 0.0;
  return a > b ? a : b;
}
__device__ __forceinline__ double jclamp(double x, double lo, double hi) { return jmin(jmax(x, lo), hi); }
__device__ __forceinline__ double jround(double x) {
  if (x != x || fabs(x) == __longlong_as_double(0x7ff0000000000000ll)) return x;
  double r = floor(x);
  if (x - r >= 0.5) r += 1.0;
  return r;
}

// ------------------------------------------------------------------------------------------------ transpose_in
// block 256 threads: tile of 64 vectors x 16 float4 (64 dims) through LDS, both sides coalesced
__global__ __launch_bounds__(256) void bbq_transpose_in_kernel(const float *__restrict__ in, int64_t n, int32_t dim, int32_t dim4,
                                                              int64_t npad, f32x4 *__restrict__ vT4) {
  __shared__ f32x4 tile[64][17];
  const int64_t v0 = (int64_t)blockIdx.x * 64;
  const int i40 = blockIdx.y * 16;
  const int t = threadIdx.x;
  {
    const int c = t & 15;  // float4 column inside the tile
    const int i4 = i40 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int vr = (t >> 4) + 16 * r;
      const int64_t vec = v0 + vr;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (vec < n && i4 < dim4) {
        const float *src = in + vec * (int64_t)dim + 4 * i4;
        if ((dim & 3) == 0) {
          x = *reinterpret_cast<const f32x4 *>(src);
        } else {
          if (4 * i4 + 0 < dim) x.x = src[0];
          if (4 * i4 + 1 < dim) x.y = src[1];
          if (4 * i4 + 2 < dim) x.z = src[2];
          if (4 * i4 + 3 < dim) x.w = src[3];
        }
      }
      tile[vr][c] = x;
    }
  }
  __syncthreads();
  {
    const int vr = t & 63;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = (t >> 6) + 4 * r;
      const int i4 = i40 + c;
      if (i4 < dim4 && v0 + vr < npad) vT4[(int64_t)i4 * npad + v0 + vr] = tile[vr][c];
    }
  }
}

// ------------------------------------------------------------------------------------------------ normalize (COSINE)
__global__ __launch_bounds__(256) void bbq_normalize_kernel(f32x4 *__restrict__ vT4, int64_t n, int32_t dim, int32_t dim4, int64_t npad) {
  const int64_t vec = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vec >= n) return;
  double n2 = 0;
  for (int i4 = 0; i4 < dim4; ++i4) {
    const f32x4 v = vT4[(int64_t)i4 * npad + vec];
    n2 += (double)v.x * (double)v.x;
    if (4 * i4 + 1 < dim) n2 += (double)v.y * (double)v.y;
    if (4 * i4 + 2 < dim) n2 += (double)v.z * (double)v.z;
    if (4 * i4 + 3 < dim) n2 += (double)v.w * (double)v.w;
  }
  const double norm = sqrt(n2);
  for (int i4 = 0; i4 < dim4; ++i4) {
    f32x4 v = vT4[(int64_t)i4 * npad + vec];
    if (norm == 0) {
      v = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      v.x = (float)((double)v.x / norm);
      v.y = (4 * i4 + 1 < dim) ? (float)((double)v.y / norm) : 0.f;
      v.z = (4 * i4 + 2 < dim) ? (float)((double)v.z / norm) : 0.f;
      v.w = (4 * i4 + 3 < dim) ? (float)((double)v.w / norm) : 0.f;
    }
    vT4[(int64_t)i4 * npad + vec] = v;
  }
}

// ------------------------------------------------------------------------------------------------ validate
__global__ __launch_bounds__(256) void bbq_validate_kernel(const f32x4 *__restrict__ vT4, int64_t n, int32_t dim, int32_t dim4,
                                                          int64_t npad, unsigned long long *__restrict__ first_bad) {
  const int64_t vec = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int i4 = blockIdx.y;
  if (vec >= n) return;
  const f32x4 v = vT4[(int64_t)i4 * npad + vec];
  const float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = 4 * i4 + k;
    if (d < dim && !(fabsf(c[k]) <= FLT_MAX)) {  // NaN or +-Infinity
      atomicMin(first_bad, (unsigned long long)(vec * (int64_t)dim + d));
      return;
    }
  }
}

// ------------------------------------------------------------------------------------------------ centroid
// one wave per float4 row (4 dimensions): serial, exactly rounded f32 accumulation over the vectors in order.
// The wave loads 64 consecutive vectors' values with one coalesced 1 KiB load, parks them in LDS and every lane walks
// them in order (uniform LDS reads broadcast); the next 1 KiB is already in flight while the chain runs.
__global__ __launch_bounds__(64) void bbq_centroid_kernel(const f32x4 *__restrict__ vT4, int64_t n, int32_t dim, int64_t npad,
                                                         float *__restrict__ centroid) {
  __shared__ f32x4 s_buf[2][64];
  const int i4 = blockIdx.x;
  const int lane = threadIdx.x;
  const f32x4 *__restrict__ row = vT4 + (int64_t)i4 * npad;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
  f32x4 x = row[lane];  // npad is a multiple of 64: always in bounds
  int pb = 0;
  for (int64_t j0 = 0; j0 < n; j0 += 64, pb ^= 1) {
    s_buf[pb][lane] = x;
    if (j0 + 64 < n) x = row[j0 + 64 + lane];  // prefetch
    __syncthreads();
    const int cnt = (int)((n - j0) < 64 ? (n - j0) : 64);
    for (int t = 0; t < cnt; ++t) {
      const f32x4 a = s_buf[pb][t];
      if (j0 == 0 && t == 0) {  // centroid[i] = vectors[0][i]
        c0 = a.x; c1 = a.y; c2 = a.z; c3 = a.w;
      } else {                  // centroid[i] += val  (Float32Array element: f64 add, then round to f32)
        c0 = (float)((double)c0 + (double)a.x);
        c1 = (float)((double)c1 + (double)a.y);
        c2 = (float)((double)c2 + (double)a.z);
        c3 = (float)((double)c3 + (double)a.w);
      }
    }
    // the other buffer is overwritten next iteration; every lane has finished reading it one iteration ago
  }
  if (lane == 0) {
    const double dn = (double)n;
    if (4 * i4 + 0 < dim) centroid[4 * i4 + 0] = (float)((double)c0 / dn);
    if (4 * i4 + 1 < dim) centroid[4 * i4 + 1] = (float)((double)c1 / dn);
    if (4 * i4 + 2 < dim) centroid[4 * i4 + 2] = (float)((double)c2 / dn);
    if (4 * i4 + 3 < dim) centroid[4 * i4 + 3] = (float)((double)c3 / dn);
  }
}

// ------------------------------------------------------------------------------------------------ quantize (1 bit)

struct BuildOut {
  uint8_t *tiles;      // scan layout (bbq_device.h)
  double *exact;       // kLayoutCompact side array, or null
  double *corr_rm;     // [n][4] row-major corrections for the host, or null (bits > 1: required)
  uint8_t *codes_rm;   // bits > 1: [n][dim] one byte per dimension
  int32_t w16, tile_stride, layout;
};

// one pass of computeLoss over the vector (src/optimizedScalarQuantizer.ts:373-407); pm1 = points - 1 = 2^bits - 1
__device__ __forceinline__ double loss_pass(const f32x4 *__restrict__ col, int64_t npad, const float *__restrict__ s_cen, int dim,
                                            int dim4, double a, double b, double norm2, double lambda, double pm1) {
  const double step = (b - a) / pm1;
  const double step_inv = 1.0 / step;
  double xe = 0.0, e = 0.0;
  for (int i4 = 0; i4 < dim4; ++i4) {
    const f32x4 v = col[(int64_t)i4 * npad];
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = 4 * i4 + k;
      if (d < dim) {
        const double xi = (double)(float)((double)vv[k] - (double)s_cen[d]);
        const double kq = jround((jclamp(xi, a, b) - a) * step_inv);
        const double xiq = a + step * kq;
        xe += xi * (xi - xiq);
        e += (xi - xiq) * (xi - xiq);
      }
    }
  }
  return (1.0 - lambda) * xe * xe / norm2 + lambda * e;
}

// bits == 1 packs the row straight into the scan layout; bits > 1 writes what the reference keeps for such an index - one byte per
// dimension (src/binaryQuantizationFormat.ts:241-245) - to out.codes_rm, and the caller builds the tile records from that
__global__ __launch_bounds__(256) void bbq_quantize1_kernel(const f32x4 *__restrict__ vT4, int64_t n, int32_t dim, int32_t dim4,
                                                           int64_t npad, const float *__restrict__ centroid, int32_t sim,
                                                           double lambda, int32_t iters, int32_t bits, BuildOut out) {
  extern __shared__ float s_cen[];
  for (int i = threadIdx.x; i < dim4 * 4; i += 256) s_cen[i] = i < dim ? centroid[i] : 0.f;
  __syncthreads();
  const int64_t vec = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vec >= npad) return;
  const bool valid = vec < n;
  const f32x4 *__restrict__ col = vT4 + vec;
  const double ddim = (double)dim;

  // pass 1 (:155-178): centroid dot on the uncentred input, min/max of the f64 differences, sum of the f32 centred values
  double cdot = 0, mn = DBL_MAX, mx = -DBL_MAX, sum = 0;
  if (valid) {
    for (int i4 = 0; i4 < dim4; ++i4) {
      const f32x4 v = col[(int64_t)i4 * npad];
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = 4 * i4 + k;
        if (d < dim) {
          const double vd = (double)vv[k], cd = (double)s_cen[d];
          if (sim != 0) cdot += vd * cd;
          const double diff = vd - cd;
          mn = jmin(mn, diff);
          mx = jmax(mx, diff);
          sum += (double)(float)diff;
        }
      }
    }
  }
  const double mean = sum / ddim;
  // pass 2 (:181-183): std and L2 norm of the centred vector
  double var = 0, n2 = 0;
  if (valid) {
    for (int i4 = 0; i4 < dim4; ++i4) {
      const f32x4 v = col[(int64_t)i4 * npad];
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = 4 * i4 + k;
        if (d < dim) {
          const double w = (double)(float)((double)vv[k] - (double)s_cen[d]);
          const double dd = w - mean;
          var += dd * dd;
          n2 += w * w;
        }
      }
    }
  }
  const double sd = sqrt(var / ddim);
  const double norm2 = sqrt(n2);
  const double kGrid[8] = {0.798, 1.493, 2.051, 2.514, 2.916, 3.278, 3.611, 3.922};  // MINIMUM_MSE_GRID, src/constants.ts:38-47
  const double g = kGrid[bits - 1];
  const double pm1 = (double)((1 << bits) - 1);  // points - 1
  double iv0 = jclamp(-g * sd + mean, mn, mx), iv1 = jclamp(g * sd + mean, mn, mx);

  // optimizeIntervals (:280-353)
  {
    double best = valid ? loss_pass(col, npad, s_cen, dim, dim4, iv0, iv1, norm2, lambda, pm1) : 0.0;
    const double scale = (1.0 - lambda) / norm2;
    bool active = valid && (fabs(scale) <= DBL_MAX);  // isFinite(scale)
    for (int it = 0; it < iters; ++it) {
      if (!__any(active)) break;
      if (active) {
        const double a = iv0, b = iv1;
        const double step_inv = pm1 / (b - a);  // (points - 1) / (b - a)
        double daa = 0, dab = 0, dbb = 0, dax = 0, dbx = 0;
        for (int i4 = 0; i4 < dim4; ++i4) {
          const f32x4 v = col[(int64_t)i4 * npad];
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int d = 4 * i4 + k;
            if (d < dim) {
              const double xi = (double)(float)((double)vv[k] - (double)s_cen[d]);
              const double kq = jround((jclamp(xi, a, b) - a) * step_inv);
              const double s = kq / pm1;
              daa += (1.0 - s) * (1.0 - s);
              dab += (1.0 - s) * s;
              dbb += s * s;
              dax += xi * (1.0 - s);
              dbx += xi * s;
            }
          }
        }
        const double m0 = scale * dax * dax + lambda * daa;
        const double m1 = scale * dax * dbx + lambda * dab;
        const double m2 = scale * dbx * dbx + lambda * dbb;
        const double det = m0 * m2 - m1 * m1;
        if (fabs(det) < 1e-12) {
          active = false;
        } else {
          const double a_opt = (m2 * dax - m1 * dbx) / det;
          const double b_opt = (m0 * dbx - m1 * dax) / det;
          if (fabs(iv0 - a_opt) < 1e-8 && fabs(iv1 - b_opt) < 1e-8) {
            active = false;
          } else {
            const double nl = loss_pass(col, npad, s_cen, dim, dim4, a_opt, b_opt, norm2, lambda, pm1);
            if (nl > best) {
              active = false;
            } else {
              iv0 = a_opt;
              iv1 = b_opt;
              best = nl;
            }
          }
        }
      }
    }
  }

  if (bits > 1) {
    // final pass (:192-216) for more than one bit: assignment = round((clamp(x) - a) * stepInv), dest = min(assignment, nSteps), the
    // component sum adds the assignment as it is
    const double a = iv0, b = iv1;
    const double step = (b - a) / pm1;
    const double step_inv = step > 0 ? 1.0 / step : 0.0;
    double qsum = 0;
    if (valid) {
      uint8_t *__restrict__ dst = out.codes_rm + vec * (int64_t)dim;
      for (int i4 = 0; i4 < dim4; ++i4) {
        const f32x4 v = col[(int64_t)i4 * npad];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = 4 * i4 + k;
          if (d < dim) {
            const double xi = (double)(float)((double)vv[k] - (double)s_cen[d]);
            const double assignment = jround((jclamp(xi, a, b) - a) * step_inv);
            qsum += assignment;
            const double capped = jmin(assignment, pm1);
            // Uint8Array store of a double: ToUint8 (NaN -> 0, otherwise modulo 256 of the truncated value)
            dst[d] = (capped != capped) ? (uint8_t)0 : (uint8_t)(unsigned int)(long long)capped;
          }
        }
      }
      double *cm = out.corr_rm + vec * 4;
      cm[0] = iv0; cm[1] = iv1; cm[2] = (sim == 0) ? norm2 : cdot; cm[3] = qsum;
    }
    return;
  }

  // final pass (:192-216): 1-bit threshold at the interval midpoint, packed MSB-first (packAsBinary :420-446) straight into
  // the tile record: 16-byte chunk j of row r at (j*64 + r)*16
  const int64_t tile = vec / kTileRows;
  const int r = (int)(vec % kTileRows);
  uint8_t *tp = out.tiles + tile * (int64_t)out.tile_stride;
  const double a = iv0, b = iv1;
  const double thr = (a + b) / 2;
  double qsum = 0;
  uint32_t wcur = 0;
  u32x4b chunk = {0, 0, 0, 0};
  for (int i4 = 0; i4 < out.w16 * 32; ++i4) {
    if (valid && i4 < dim4) {
      const f32x4 v = col[(int64_t)i4 * npad];
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = 4 * i4 + k;
        if (d < dim) {
          const double xi = (double)(float)((double)vv[k] - (double)s_cen[d]);
          if (jclamp(xi, a, b) >= thr) {
            wcur |= 1u << (8 * ((d >> 3) & 3) + 7 - (d & 7));
            qsum += 1.0;
          }
        }
      }
    }
    if ((i4 & 7) == 7) {  // 32 dims = one little-endian 32-bit word of the packed row
      const int wi = (i4 >> 3) & 3;
      if (wi == 0) chunk.x = wcur; else if (wi == 1) chunk.y = wcur; else if (wi == 2) chunk.z = wcur; else chunk.w = wcur;
      wcur = 0;
      if (wi == 3) {
        reinterpret_cast<u32x4b *>(tp)[(i4 >> 5) * kTileRows + r] = chunk;
        chunk = u32x4b{0, 0, 0, 0};
      }
    }
  }
  double lower = 0, upper = 0, add = 0;
  if (valid) {
    lower = iv0;
    upper = iv1;
    add = (sim == 0) ? norm2 : cdot;  // :219
  }
  uint8_t *cr = tp + (size_t)out.w16 * (kTileRows * 16);
  if (out.layout == kLayoutCompact) {
    // the tiles' additive-correction ranges are computed afterwards from exact[] (launch_tile_add_range)
    reinterpret_cast<uint32_t *>(cr)[r] = (__float_as_uint((float)lower) >> 16) | ((__float_as_uint((float)upper) >> 16) << 16);
    double *e = out.exact + vec * 4;
    e[0] = lower; e[1] = upper; e[2] = add; e[3] = 0.0;
  } else {
    f64x2b lu = {lower, upper};
    reinterpret_cast<f64x2b *>(cr)[r] = lu;
    reinterpret_cast<double *>(cr + 1024)[r] = add;
  }
  if (out.corr_rm && valid) {
    double *cm = out.corr_rm + vec * 4;
    cm[0] = lower; cm[1] = upper; cm[2] = add; cm[3] = qsum;
  }
}

// ------------------------------------------------------------------------------------------------ untile (codes for the host)
__global__ __launch_bounds__(256) void bbq_untile_kernel(const uint8_t *__restrict__ tiles, int64_t n, int32_t pb, int32_t w16,
                                                        int32_t tile_stride, uint8_t *__restrict__ codes_rm) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = gid / w16;
  const int j = (int)(gid % w16);
  if (row >= n) return;
  const uint8_t *tp = tiles + (row / kTileRows) * (int64_t)tile_stride;
  const u32x4b c = reinterpret_cast<const u32x4b *>(tp)[j * kTileRows + (int)(row % kTileRows)];
  const uint32_t w[4] = {c.x, c.y, c.z, c.w};
  uint8_t *dst = codes_rm + row * (int64_t)pb;
  for (int b = 0; b < 16; ++b) {
    const int byte = j * 16 + b;
    if (byte < pb) dst[byte] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
  }
}

// ------------------------------------------------------------------------------------------------ launch wrappers

hipError_t launch_build_transpose(const float *in, int64_t n, int32_t dim, int64_t npad, float *vT4, hipStream_t s) {
  const int dim4 = (dim + 3) / 4;
  dim3 grid((unsigned)(npad / 64), (unsigned)((dim4 + 15) / 16));
  hipLaunchKernelGGL(bbq_transpose_in_kernel, grid, dim3(256), 0, s, in, n, dim, dim4, npad, reinterpret_cast<f32x4 *>(vT4));
  return hipGetLastError();
}
hipError_t launch_build_normalize(float *vT4, int64_t n, int32_t dim, int64_t npad, hipStream_t s) {
  const int dim4 = (dim + 3) / 4;
  hipLaunchKernelGGL(bbq_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<f32x4 *>(vT4), n, dim, dim4, npad);
  return hipGetLastError();
}
hipError_t launch_build_validate(const float *vT4, int64_t n, int32_t dim, int64_t npad, unsigned long long *first_bad, hipStream_t s) {
  const int dim4 = (dim + 3) / 4;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)dim4);
  hipLaunchKernelGGL(bbq_validate_kernel, grid, dim3(256), 0, s, reinterpret_cast<const f32x4 *>(vT4), n, dim, dim4, npad, first_bad);
  return hipGetLastError();
}
hipError_t launch_build_centroid(const float *vT4, int64_t n, int32_t dim, int64_t npad, float *centroid, hipStream_t s) {
  const int dim4 = (dim + 3) / 4;
  hipLaunchKernelGGL(bbq_centroid_kernel, dim3((unsigned)dim4), dim3(64), 0, s, reinterpret_cast<const f32x4 *>(vT4), n, dim, npad, centroid);
  return hipGetLastError();
}
hipError_t launch_build_quantize1(const float *vT4, int64_t n, int32_t dim, int64_t npad, const float *centroid, int32_t sim,
                                  double lambda, int32_t iters, uint8_t *tiles, double *exact, double *corr_rm, int32_t w16,
                                  int32_t tile_stride, int32_t layout, hipStream_t s) {
  const int dim4 = (dim + 3) / 4;
  BuildOut o{tiles, exact, corr_rm, nullptr, w16, tile_stride, layout};
  hipLaunchKernelGGL(bbq_quantize1_kernel, dim3((unsigned)(npad / 256 + (npad % 256 ? 1 : 0))), dim3(256), (size_t)dim4 * 16, s,
                     reinterpret_cast<const f32x4 *>(vT4), n, dim, dim4, npad, centroid, sim, lambda, iters, 1, o);
  return hipGetLastError();
}
hipError_t launch_build_quantize_bits(const float *vT4, int64_t n, int32_t dim, int64_t npad, const float *centroid, int32_t sim,
                                      double lambda, int32_t iters, int32_t bits, uint8_t *codes_rm, double *corr_rm, hipStream_t s) {
  const int dim4 = (dim + 3) / 4;
  BuildOut o{nullptr, nullptr, corr_rm, codes_rm, 0, 0, 0};
  hipLaunchKernelGGL(bbq_quantize1_kernel, dim3((unsigned)(npad / 256 + (npad % 256 ? 1 : 0))), dim3(256), (size_t)dim4 * 16, s,
                     reinterpret_cast<const f32x4 *>(vT4), n, dim, dim4, npad, centroid, sim, lambda, iters, bits, o);
  return hipGetLastError();
}
hipError_t launch_build_untile(const uint8_t *tiles, int64_t n, int32_t pb, int32_t w16, int32_t tile_stride, uint8_t *codes_rm,
                               hipStream_t s) {
  const int64_t threads = n * w16;
  if (threads <= 0) return hipSuccess;
  hipLaunchKernelGGL(bbq_untile_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, tiles, n, pb, w16, tile_stride, codes_rm);
  return hipGetLastError();
}

}  // namespace bbq
