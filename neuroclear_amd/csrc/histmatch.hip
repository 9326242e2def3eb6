// --histogram_match of test_dice.py (reference util/assemble_dice.py:149-151): every network output cube is matched to
// the histogram of its input cube with skimage.exposure.match_histograms before it is queued.  scikit-image 0.18.3
// (pinned by the reference's conda environment) implements it as
//     src_values, src_unique_indices, src_counts = np.unique(source.ravel(), return_inverse=True, return_counts=True)
//     tmpl_values, tmpl_counts = np.unique(template.ravel(), return_counts=True)
//     src_quantiles = np.cumsum(src_counts) / source.size;  tmpl_quantiles = np.cumsum(tmpl_counts) / template.size
//     return np.interp(src_quantiles, tmpl_quantiles, tmpl_values)[src_unique_indices].reshape(source.shape)
// i.e. out(v) = interp(rank(v) / n) through the template's empirical CDF, rank(v) = #{source <= v}, float64 arithmetic.
//
// On the device, with both cubes of n voxels resident:
//   1. order-preserving 32-bit keys of both arrays, LSD radix sort (8 passes of 4 bits, stable: a thread owns 8
//      consecutive keys of its block's chunk; per-(digit, thread) counters in LDS, no atomics in the scatter);
//   2. per source voxel: r = upper_bound(sorted source, v) -- the cumulative count np.cumsum(src_counts) of its value;
//      in the sorted template T the CDF knots are the run ends, so np.interp's bracket [xp[j], xp[j+1]) around r / n is
//      found with two more binary searches around T[r], and np.interp's own expression
//      slope * (x - xp[j]) + fp[j], slope = (fp[j+1] - fp[j]) / (xp[j+1] - xp[j]) is evaluated in float64 with one
//      rounding per operation (no FMA contraction), including its special cases (x below the first knot, x exactly on a
//      knot, x on the last knot).
// Integer counts only, so the result does not depend on scheduling.  Output is the float64 result rounded to float32
// (the reference adds the float64 cube into a float32 stack).
#include "common.hpp"

namespace nc {
namespace {

constexpr int RS_T = 256, RS_E = 8, RS_CH = RS_T * RS_E;  // keys per block and pass

__device__ __forceinline__ unsigned fkey(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void __launch_bounds__(256) k_hm_keys(const float* __restrict__ x, unsigned* __restrict__ k, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) k[i] = fkey(x[i] + 0.f);  // + 0: -0.0 sorts (and compares) as +0.0, like np.unique
}

// hist[d * B + block] = number of keys of the block's chunk with digit d
__global__ void __launch_bounds__(RS_T) k_rs_hist(const unsigned* __restrict__ k, long n, int shift,
                                                  unsigned* __restrict__ hist, int B) {
  __shared__ unsigned h[16];
  if (threadIdx.x < 16) h[threadIdx.x] = 0;
  __syncthreads();
  const long base = (long)blockIdx.x * RS_CH + threadIdx.x * RS_E;
#pragma unroll
  for (int e = 0; e < RS_E; ++e)
    if (base + e < n) atomicAdd(&h[(k[base + e] >> shift) & 15u], 1u);
  __syncthreads();
  if (threadIdx.x < 16) hist[threadIdx.x * B + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `total` counters in place, one workgroup
__global__ void __launch_bounds__(1024) k_rs_scan(unsigned* __restrict__ a, int total) {
  __shared__ unsigned part[1024];
  const int t = threadIdx.x;
  const int per = (total + 1023) / 1024;
  const int lo = t * per, hi = min(total, lo + per);
  unsigned s = 0;
  for (int i = lo; i < hi; ++i) s += a[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
    const unsigned v = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned run = t ? part[t - 1] : 0u;
  for (int i = lo; i < hi; ++i) {
    const unsigned v = a[i];
    a[i] = run;
    run += v;
  }
}

__global__ void __launch_bounds__(RS_T) k_rs_scatter(const unsigned* __restrict__ kin, unsigned* __restrict__ kout, long n,
                                                     int shift, const unsigned* __restrict__ base, int B) {
  __shared__ unsigned short cnt[16 * RS_T];  // [digit][thread]: keys of that digit owned by that thread
  const int t = threadIdx.x;
  for (int i = t; i < 16 * RS_T; i += RS_T) cnt[i] = 0;
  __syncthreads();
  const long b0 = (long)blockIdx.x * RS_CH + t * RS_E;
  unsigned key[RS_E];
  unsigned short rk[RS_E];
#pragma unroll
  for (int e = 0; e < RS_E; ++e) {
    key[e] = b0 + e < n ? kin[b0 + e] : 0u;
    rk[e] = 0;
    if (b0 + e < n) rk[e] = cnt[((key[e] >> shift) & 15u) * RS_T + t]++;  // column t is private to this thread
  }
  __syncthreads();
  if (t < 16) {  // exclusive scan over the threads, per digit
    unsigned short run = 0;
    for (int i = 0; i < RS_T; ++i) {
      const unsigned short v = cnt[t * RS_T + i];
      cnt[t * RS_T + i] = run;
      run += v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < RS_E; ++e)
    if (b0 + e < n) {
      const unsigned d = (key[e] >> shift) & 15u;
      kout[base[d * B + blockIdx.x] + cnt[d * RS_T + t] + rk[e]] = key[e];
    }
}

__device__ __forceinline__ long ubound(const unsigned* __restrict__ a, long n, unsigned v) {  // #{a <= v}
  long lo = 0, hi = n;
  while (lo < hi) {
    const long m = (lo + hi) >> 1;
    if (a[m] <= v) lo = m + 1; else hi = m;
  }
  return lo;
}
__device__ __forceinline__ long lbound(const unsigned* __restrict__ a, long n, unsigned v) {  // #{a < v}
  long lo = 0, hi = n;
  while (lo < hi) {
    const long m = (lo + hi) >> 1;
    if (a[m] < v) lo = m + 1; else hi = m;
  }
  return lo;
}

__global__ void __launch_bounds__(256) k_hm_apply(const float* __restrict__ src, const unsigned* __restrict__ S,
                                                  const unsigned* __restrict__ T, float* __restrict__ out, long n) {
#pragma clang fp contract(off)  // np.interp rounds the product and the sum separately
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long r = ubound(S, n, fkey(src[i] + 0.f));  // cumulative count of this source value: x = r / n
  double res;
  if (r >= n) {
    res = (double)fkey_inv(T[n - 1]);  // x on the last knot
  } else {
    const unsigned tv = T[r];
    const long lo = lbound(T, n, tv);  // knots: run ends; index r lies in the run [lo, hi) of value tv
    if (lo == 0) {
      res = (double)fkey_inv(T[0]);  // x below the first knot -> fp[0]
    } else if (r == lo) {
      res = (double)fkey_inv(T[lo - 1]);  // x exactly on knot j = the run that ends at lo
    } else {
      const long hi = ubound(T, n, tv);
      const double dn = (double)n;
      const double xj = (double)lo / dn, xj1 = (double)hi / dn, x = (double)r / dn;
      const double fj = (double)fkey_inv(T[lo - 1]), fj1 = (double)fkey_inv(tv);
      const double slope = (fj1 - fj) / (xj1 - xj);
      const double prod = slope * (x - xj);  // plain operators under `fp contract(off)`: one rounding each (the
      res = prod + fj;                       // __dmul_rn / __dadd_rn header inlines carry the command-line contract flag)
    }
  }
  out[i] = (float)res;
}

size_t hm_align(size_t b) { return (b + 255) & ~(size_t)255; }

int radix_sort(unsigned* a, unsigned* tmp, long n, unsigned* hist, int B, hipStream_t s) {
  unsigned* in = a;
  unsigned* outp = tmp;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 4 * pass;
    hipLaunchKernelGGL(k_rs_hist, dim3(B), dim3(RS_T), 0, s, in, n, shift, hist, B);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, s, hist, 16 * B);
    hipLaunchKernelGGL(k_rs_scatter, dim3(B), dim3(RS_T), 0, s, in, outp, n, shift, hist, B);
    unsigned* t = in; in = outp; outp = t;
  }
  return check_launch("radix_sort");  // 8 passes: the sorted keys are back in `a`
}

}  // namespace
}  // namespace nc

using namespace nc;

extern "C" {

size_t nc_match_histograms_ws_bytes(long n) {
  if (n < 1) return 0;
  const long B = (n + RS_CH - 1) / RS_CH;
  return 3 * hm_align((size_t)n * 4) + hm_align((size_t)16 * B * 4) + 256;
}

int nc_match_histograms(const float* source, const float* tmpl, float* out, long n, void* ws, size_t ws_bytes,
                        void* stream) {
  if (!source || !tmpl || !out) { set_error("match_histograms: null pointer"); return NC_ERR_ARG; }
  if (n < 1 || n >= (1L << 31)) { set_error("match_histograms: bad size"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < nc_match_histograms_ws_bytes(n)) { set_error("match_histograms: workspace too small"); return NC_ERR_WS; }
  hipStream_t s = (hipStream_t)stream;
  const int B = (int)((n + RS_CH - 1) / RS_CH);
  const size_t kb = hm_align((size_t)n * 4);
  unsigned* S = (unsigned*)ws;
  unsigned* T = (unsigned*)((char*)ws + kb);
  unsigned* tmp = (unsigned*)((char*)ws + 2 * kb);
  unsigned* hist = (unsigned*)((char*)ws + 3 * kb);
  const unsigned g = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_hm_keys, dim3(g), dim3(256), 0, s, source, S, n);
  hipLaunchKernelGGL(k_hm_keys, dim3(g), dim3(256), 0, s, tmpl, T, n);
  if (int e = radix_sort(S, tmp, n, hist, B, s)) return e;
  if (int e = radix_sort(T, tmp, n, hist, B, s)) return e;
  hipLaunchKernelGGL(k_hm_apply, dim3(g), dim3(256), 0, s, source, S, T, out, n);
  return check_launch("match_histograms");
}

}  // extern "C"
