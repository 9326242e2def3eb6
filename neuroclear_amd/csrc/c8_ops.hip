// Everything BETWEEN the 16-bit convolutions of the 16-bit end-to-end path (BASELINE.json configs[3]), on the kernels'
// own operand layout "C8" = [N][C/8][S voxels][8 channels] 16-bit (conv_h.hip), so that no fp32 activation tensor is
// written or re-read between two convolutions:
//   InstanceNorm3d(affine=False) + ReLU  (reference models/networks.py:33-34, 422-423): statistics in fp32 / fp64
//       ("fp32 InstanceNorm accumulate"), normalise + activate C8 -> C8, and the backward of the pair;
//   MaxPool3d(2)                         (:491,494) forward, and backward fused with the add of the skip gradient;
//   the 64 -> 1 pointwise head           (:507-508 one_by_one; the collapsed 1x1 tail of deep_linear_gen, :903-911).
// All of these are HBM-bound: one 16-byte unit (a voxel's 8 channels) per lane per access, coalesced along voxels.
// A tensor argument (ptr, ctot, c0) addresses channels [c0, c0 + C) of a [N][ctot/8][S][8] buffer -- a half of a skip
// concat buffer is read / written in place, there is no concat copy.
#include "common.hpp"

namespace nc {
namespace {

template <int DT>
__device__ __forceinline__ float cvtf(unsigned short u) {
  if constexpr (DT == NC_DT_F16) return (float)__builtin_bit_cast(_Float16, u);
  else return __builtin_bit_cast(float, (unsigned)u << 16);
}
template <int DT>
__device__ __forceinline__ unsigned short cvth(float f) {
  if constexpr (DT == NC_DT_F16) {
    const _Float16 v = (_Float16)f;
    return __builtin_bit_cast(unsigned short, v);
  } else {
    const __bf16 v = (__bf16)f;
    return __builtin_bit_cast(unsigned short, v);
  }
}
template <int DT>
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = cvtf<DT>((unsigned short)(u.x & 0xffff)); f[1] = cvtf<DT>((unsigned short)(u.x >> 16));
  f[2] = cvtf<DT>((unsigned short)(u.y & 0xffff)); f[3] = cvtf<DT>((unsigned short)(u.y >> 16));
  f[4] = cvtf<DT>((unsigned short)(u.z & 0xffff)); f[5] = cvtf<DT>((unsigned short)(u.z >> 16));
  f[6] = cvtf<DT>((unsigned short)(u.w & 0xffff)); f[7] = cvtf<DT>((unsigned short)(u.w >> 16));
}
template <int DT>
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 o;
  o.x = cvth<DT>(f[0]) | ((unsigned)cvth<DT>(f[1]) << 16); o.y = cvth<DT>(f[2]) | ((unsigned)cvth<DT>(f[3]) << 16);
  o.z = cvth<DT>(f[4]) | ((unsigned)cvth<DT>(f[5]) << 16); o.w = cvth<DT>(f[6]) | ((unsigned)cvth<DT>(f[7]) << 16);
  return o;
}

constexpr int kSplitMax = 256;
int splits_for(long S) {
  long s = cdiv(S, 256 * 64);
  return (int)(s < 1 ? 1 : s > kSplitMax ? kSplitMax : s);
}

// block reduction of NV per-thread doubles -> out[NV] (thread 0..NV-1 hold the sums); 256 threads
template <int NV>
__device__ __forceinline__ void block_reduce(double* v, double* lds /* [4][NV] */, double* out) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double a = v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    v[j] = a;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[wave * NV + j] = v[j];
  __syncthreads();
  if (threadIdx.x < NV) out[threadIdx.x] = lds[threadIdx.x] + lds[NV + threadIdx.x] + lds[2 * NV + threadIdx.x] + lds[3 * NV + threadIdx.x];
}

// ---- InstanceNorm statistics: part[(ncb * splits + split)][16] = 8 sums, 8 sums of squares ------------------------
template <int DT>
__global__ void __launch_bounds__(256) k_stats_c8(const uint4* __restrict__ x, long S, int splits, double* __restrict__ part) {
  __shared__ double lds[4 * 16];
  const long ncb = blockIdx.y;
  const int split = blockIdx.x;
  const long lo = S * split / splits, hi = S * (split + 1) / splits;
  const uint4* xs = x + ncb * S;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long v = lo + threadIdx.x; v < hi; v += 256) {
    float f[8];
    unpack8<DT>(xs[v], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
  }
  double d[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) { d[j] = s[j]; d[8 + j] = q[j]; }
  block_reduce<16>(d, lds, part + (ncb * splits + split) * 16);
}

// one wave per 8-channel block: lane = (split lane kl = lane / 8, channel j = lane % 8) walks the partials kl, kl + 8, ...;
// the eight split lanes of a channel are then added in a fixed (butterfly) order
__device__ __forceinline__ void c8_final_sums(const double* __restrict__ part, int ncb, int splits, int lane, double& s, double& q) {
  const int j = lane & 7, kl = lane >> 3;
  s = 0;
  q = 0;
  for (int k = kl; k < splits; k += 8) {
    s += part[((long)ncb * splits + k) * 16 + j];
    q += part[((long)ncb * splits + k) * 16 + 8 + j];
  }
#pragma unroll
  for (int o = 32; o >= 8; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
}

__global__ void __launch_bounds__(64) k_stats_final_c8(const double* __restrict__ part, int NC, int splits, long S, float eps,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
  const int ncb = blockIdx.x, lane = threadIdx.x;
  double s, q;
  c8_final_sums(part, ncb, splits, lane, s, q);
  const int i = ncb * 8 + lane;  // n * C + c
  if (lane < 8 && i < NC) {
    const double m = s / (double)S;
    double var = q / (double)S - m * m;
    if (var < 0) var = 0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- normalise + ReLU / LeakyReLU, C8 -> C8 (out: channels [c0, c0 + C) of a ctot-channel buffer) --------------------
template <int DT>
__global__ void __launch_bounds__(256) k_apply_c8(const uint4* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, float slope, uint4* __restrict__ y, int CB,
                                                  long S, int ctot8, int c08) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const int ncb = blockIdx.y, n = ncb / CB, cb = ncb - n * CB;
  float f[8];
  unpack8<DT>(x[(long)ncb * S + v], f);
  const float* m = mean + (long)ncb * 8;
  const float* r = rstd + (long)ncb * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = (f[j] - m[j]) * r[j];
    f[j] = t > 0.f ? t : t * slope;
  }
  y[((long)n * ctot8 + c08 + cb) * S + v] = pack8<DT>(f);
}

// ---- backward of the pair.  g: gradient at the activation's output (channels [gc0, ..) of a gctot buffer, 16-bit DG),
//      x: the raw convolution output (dense, DT).  sums: part[...][16] = 8 x sum(g'), 8 x sum(g' * xhat) -----------------
template <int DT, int DG>
__global__ void __launch_bounds__(256) k_bwd_sums_c8(const uint4* __restrict__ g, int gctot8, int gc08, const uint4* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, float slope,
                                                     int CB, long S, int splits, double* __restrict__ part) {
  __shared__ double lds[4 * 16];
  const int ncb = blockIdx.y, n = ncb / CB, cb = ncb - n * CB;
  const int split = blockIdx.x;
  const long lo = S * split / splits, hi = S * (split + 1) / splits;
  const uint4* xs = x + (long)ncb * S;
  const uint4* gs = g + ((long)n * gctot8 + gc08 + cb) * S;
  float m[8], r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m[j] = mean[(long)ncb * 8 + j]; r[j] = rstd[(long)ncb * 8 + j]; }
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long v = lo + threadIdx.x; v < hi; v += 256) {
    float f[8], gg[8];
    unpack8<DT>(xs[v], f);
    unpack8<DG>(gs[v], gg);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (f[j] - m[j]) * r[j];
      const float gp = xh > 0.f ? gg[j] : gg[j] * slope;
      s1[j] += gp;
      s2[j] = fmaf(gp, xh, s2[j]);
    }
  }
  double d[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) { d[j] = s1[j]; d[8 + j] = s2[j]; }
  block_reduce<16>(d, lds, part + ((long)ncb * splits + split) * 16);
}

__global__ void __launch_bounds__(64) k_bwd_final_c8(const double* __restrict__ part, int NC, int splits, long S, float* __restrict__ m1,
                                                     float* __restrict__ m2) {
  const int ncb = blockIdx.x, lane = threadIdx.x;
  double a, b;
  c8_final_sums(part, ncb, splits, lane, a, b);
  const int i = ncb * 8 + lane;
  if (lane < 8 && i < NC) {
    m1[i] = (float)(a / (double)S);
    m2[i] = (float)(b / (double)S);
  }
}

// dx = rstd * (g' - mean(g') - xhat * mean(g' xhat)) -> C8 (DG); dbp[(ncb * nbx + blockIdx.x)][8] = this block's sums of dx
template <int DT, int DG>
__global__ void __launch_bounds__(256) k_bwd_apply_c8(const uint4* __restrict__ g, int gctot8, int gc08, const uint4* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ m1, const float* __restrict__ m2, float slope,
                                                      int CB, long S, int per_thread, uint4* __restrict__ dx, double* __restrict__ dbp) {
  __shared__ double lds[4 * 8];
  const int ncb = blockIdx.y, n = ncb / CB, cb = ncb - n * CB;
  const uint4* xs = x + (long)ncb * S;
  const uint4* gs = g + ((long)n * gctot8 + gc08 + cb) * S;
  uint4* ds = dx + (long)ncb * S;
  float m[8], r[8], a1[8], a2[8], sb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m[j] = mean[(long)ncb * 8 + j]; r[j] = rstd[(long)ncb * 8 + j];
    a1[j] = m1[(long)ncb * 8 + j]; a2[j] = m2[(long)ncb * 8 + j];
  }
  const long base = (long)blockIdx.x * 256 * per_thread;
  for (int k = 0; k < per_thread; ++k) {
    const long v = base + (long)k * 256 + threadIdx.x;
    if (v < S) {
      float f[8], gg[8];
      unpack8<DT>(xs[v], f);
      unpack8<DG>(gs[v], gg);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (f[j] - m[j]) * r[j];
        const float gp = xh > 0.f ? gg[j] : gg[j] * slope;
        const float d = r[j] * (gp - a1[j] - xh * a2[j]);
        sb[j] += d;
        f[j] = d;
      }
      ds[v] = pack8<DG>(f);
    }
  }
  if (dbp) {
    double d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = sb[j];
    block_reduce<8>(d, lds, dbp + ((long)ncb * gridDim.x + blockIdx.x) * 8);
  }
}

// dbias[c] = sum over samples and blocks of the partial sums: one block per channel, fixed-shape tree (deterministic)
__global__ void __launch_bounds__(256) k_dbias_final_c8(const double* __restrict__ dbp, int N, int C, int nbx, float* __restrict__ dbias) {
  __shared__ double lds[4];
  const int c = blockIdx.x;
  const int CB = C >> 3, cb = c >> 3, j = c & 7;
  double a = 0;
  const int total = N * nbx;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int n = i / nbx, b = i - n * nbx;
    a += dbp[(((long)n * CB + cb) * nbx + b) * 8 + j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) dbias[c] = (float)(lds[0] + lds[1] + lds[2] + lds[3]);
}

// ---- MaxPool3d(2): x = channels [c0, ..) of a ctot buffer at (D, H, W); y dense at (D/2, H/2, W/2) -----------------------
template <int DT>
__global__ void __launch_bounds__(256) k_pool_c8(const uint4* __restrict__ x, int ctot8, int c08, uint4* __restrict__ y, int CB,
                                                 int D, int H, int W) {
  const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
  const long So = (long)Do * Ho * Wo, S = (long)D * H * W;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= So) return;
  const int ncb = blockIdx.y, n = ncb / CB, cb = ncb - n * CB;
  const int ow = (int)(o % Wo), oh = (int)((o / Wo) % Ho), od = (int)(o / ((long)Wo * Ho));
  const uint4* xs = x + ((long)n * ctot8 + c08 + cb) * S;
  float best[8];
  unsigned short bh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bh[j] = 0; }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint4 u = xs[((long)(2 * od + a) * H + (2 * oh + b)) * W + 2 * ow + c];
        const unsigned short hv[8] = {(unsigned short)(u.x & 0xffff), (unsigned short)(u.x >> 16), (unsigned short)(u.y & 0xffff),
                                      (unsigned short)(u.y >> 16),    (unsigned short)(u.z & 0xffff), (unsigned short)(u.z >> 16),
                                      (unsigned short)(u.w & 0xffff), (unsigned short)(u.w >> 16)};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = cvtf<DT>(hv[j]);
          if (f > best[j] || f != f) { best[j] = f; bh[j] = hv[j]; }
        }
      }
  uint4 out;
  out.x = bh[0] | ((unsigned)bh[1] << 16); out.y = bh[2] | ((unsigned)bh[3] << 16);
  out.z = bh[4] | ((unsigned)bh[5] << 16); out.w = bh[6] | ((unsigned)bh[7] << 16);
  y[(long)ncb * So + o] = out;
}

// dx (dense, DG) = skip (channels [sc0, ..) of an sctot buffer, DG) + the pool's gradient: dp goes to the FIRST maximum
// of each 2x2x2 window in scan order (d, h, w), as MaxPool3d's backward does
template <int DT, int DG>
__global__ void __launch_bounds__(256) k_pool_bwd_add_c8(const uint4* __restrict__ dp, const uint4* __restrict__ x, int ctot8, int c08,
                                                         const uint4* __restrict__ skip, int sctot8, int sc08,
                                                         uint4* __restrict__ dx, int CB, int D, int H, int W) {
  const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
  const long So = (long)Do * Ho * Wo, S = (long)D * H * W;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= So) return;
  const int ncb = blockIdx.y, n = ncb / CB, cb = ncb - n * CB;
  const int ow = (int)(o % Wo), oh = (int)((o / Wo) % Ho), od = (int)(o / ((long)Wo * Ho));
  const uint4* xs = x + ((long)n * ctot8 + c08 + cb) * S;
  const uint4* ss = skip + ((long)n * sctot8 + sc08 + cb) * S;
  uint4* ds = dx + (long)ncb * S;
  float best[8];
  int arg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int a = t >> 2, b = (t >> 1) & 1, c = t & 1;
    float f[8];
    unpack8<DT>(xs[((long)(2 * od + a) * H + (2 * oh + b)) * W + 2 * ow + c], f);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (f[j] > best[j] || f[j] != f[j]) { best[j] = f[j]; arg[j] = t; }
  }
  float gp[8];
  unpack8<DG>(dp[(long)ncb * So + o], gp);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int a = t >> 2, b = (t >> 1) & 1, c = t & 1;
    const long v = ((long)(2 * od + a) * H + (2 * oh + b)) * W + 2 * ow + c;
    float f[8];
    unpack8<DG>(ss[v], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] += arg[j] == t ? gp[j] : 0.f;
    ds[v] = pack8<DG>(f);
  }
}

// ---- C8 -> fp32 NCDHW (channels [c0, ..) of a ctot buffer -> dense [N][C][S]) -------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256) k_from_c8(const uint4* __restrict__ x, int ctot8, int c08, float* __restrict__ y, int CB,
                                                 long S) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const int ncb = blockIdx.y, n = ncb / CB, cb = ncb - n * CB;
  float f[8];
  unpack8<DT>(x[((long)n * ctot8 + c08 + cb) * S + v], f);
  float* ys = y + (long)ncb * 8 * S + v;
#pragma unroll
  for (int j = 0; j < 8; ++j) ys[j * S] = f[j];
}

// ---- 64 -> 1 pointwise head on a dense 64-channel C8 tensor ---------------------------------------------------------------
// out[n][v] = bias + sum_c w[c] x[n][c][v]
template <int DT>
__global__ void __launch_bounds__(256) k_dot64_c8(const uint4* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ out, long S) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const int n = blockIdx.y;
  float acc = bias ? bias[0] : 0.f;
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    float f[8];
    unpack8<DT>(x[((long)n * 8 + cb) * S + v], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(w[cb * 8 + j], f[j], acc);
  }
  out[(long)n * S + v] = acc;
}

// backward of it: dx[n][c][v] = w[c] g[n][v] (C8, DG); part[block][65] = sum_v g x[c] (64) and sum_v g
template <int DT, int DG>
__global__ void __launch_bounds__(256) k_outer64_c8(const float* __restrict__ g, const uint4* __restrict__ x, const float* __restrict__ w,
                                                    uint4* __restrict__ dx, long S, int per_thread, double* __restrict__ part) {
  __shared__ double lds[4 * 65];
  const int n = blockIdx.y;
  float q[65];
#pragma unroll
  for (int j = 0; j < 65; ++j) q[j] = 0.f;
  const long base = (long)blockIdx.x * 256 * per_thread;
  for (int k = 0; k < per_thread; ++k) {
    const long v = base + (long)k * 256 + threadIdx.x;
    if (v < S) {
      const float gv = g[(long)n * S + v];
      q[64] += gv;
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        float f[8], o[8];
        unpack8<DT>(x[((long)n * 8 + cb) * S + v], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          q[cb * 8 + j] = fmaf(gv, f[j], q[cb * 8 + j]);
          o[j] = w[cb * 8 + j] * gv;
        }
        if (dx) dx[((long)n * 8 + cb) * S + v] = pack8<DG>(o);
      }
    }
  }
  double d[65];
#pragma unroll
  for (int j = 0; j < 65; ++j) d[j] = q[j];
  block_reduce<65>(d, lds, part + ((long)n * gridDim.x + blockIdx.x) * 65);
}

// one block per output (64 weight sums + the plain sum), fixed-shape tree
__global__ void __launch_bounds__(256) k_outer64_final(const double* __restrict__ part, int nblocks, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double lds[4];
  const int j = blockIdx.x;
  double a = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) a += part[(long)b * 65 + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = lds[0] + lds[1] + lds[2] + lds[3];
    if (j < 64) dw[j] = (float)t;
    else if (db) db[0] = (float)t;
  }
}

#define DISPATCH_DT(dt, CALL)            \
  do {                                   \
    if ((dt) == NC_DT_F16) { CALL(NC_DT_F16); } \
    else { CALL(NC_DT_BF16); }           \
  } while (0)

}  // namespace

// ---- host side (internal API, used by gen_nets_lp.hip and the extern "C" wrappers below) ------------------------------------
size_t c8_stats_ws_bytes(int N, int C, long S) {
  const size_t a = (size_t)N * (C / 8) * splits_for(S) * 16 * sizeof(double);
  const long nbx = cdiv(S, 256 * 8);
  const size_t b = (size_t)N * (C / 8) * nbx * 8 * sizeof(double);
  return a + b + 2 * (size_t)N * C * sizeof(float) + 256;
}

int c8_instnorm_stats(const void* x, int N, int C, long S, float eps, float* mean, float* rstd, int dt, void* ws, size_t wsb,
                      hipStream_t s) {
  if (C % 8 || N < 1 || S < 1) { set_error("c8_instnorm_stats: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || wsb < c8_stats_ws_bytes(N, C, S)) { set_error("c8_instnorm_stats: workspace too small"); return NC_ERR_WS; }
  const int sp = splits_for(S);
  double* part = (double*)ws;
#define CALL(D) hipLaunchKernelGGL((k_stats_c8<D>), dim3(sp, N * C / 8), dim3(256), 0, s, (const uint4*)x, S, sp, part)
  DISPATCH_DT(dt, CALL);
#undef CALL
  hipLaunchKernelGGL(k_stats_final_c8, dim3((unsigned)(N * C / 8)), dim3(64), 0, s, part, N * C, sp, S, eps, mean, rstd);
  return check_launch("c8_instnorm_stats");
}

int c8_instnorm_apply(const void* x, const float* mean, const float* rstd, float slope, void* y, int ctot, int c0, int N, int C,
                      long S, int dt, hipStream_t s) {
  if (C % 8 || ctot % 8 || c0 % 8 || c0 + C > ctot) { set_error("c8_instnorm_apply: bad channel range"); return NC_ERR_SHAPE; }
#define CALL(D) hipLaunchKernelGGL((k_apply_c8<D>), dim3((unsigned)cdiv(S, 256), N * C / 8), dim3(256), 0, s, (const uint4*)x, mean, rstd, slope, (uint4*)y, C / 8, S, ctot / 8, c0 / 8)
  DISPATCH_DT(dt, CALL);
#undef CALL
  return check_launch("c8_instnorm_apply");
}

// g: gradient (bf16) in channels [gc0, ..) of a gctot buffer; x: raw conv output (dt); dx: dense bf16; dbias nullable
int c8_instnorm_bwd(const void* g, int gctot, int gc0, const void* x, const float* mean, const float* rstd, float slope, void* dx,
                    float* dbias, int N, int C, long S, int dt, void* ws, size_t wsb, hipStream_t s) {
  if (C % 8 || gctot % 8 || gc0 % 8 || gc0 + C > gctot) { set_error("c8_instnorm_bwd: bad channel range"); return NC_ERR_SHAPE; }
  if (!ws || wsb < c8_stats_ws_bytes(N, C, S)) { set_error("c8_instnorm_bwd: workspace too small"); return NC_ERR_WS; }
  const int sp = splits_for(S);
  double* part = (double*)ws;
  const long nbx = cdiv(S, 256 * 8);
  double* dbp = part + (size_t)N * (C / 8) * sp * 16;
  float* m1 = (float*)(dbp + (size_t)N * (C / 8) * nbx * 8);
  float* m2 = m1 + (size_t)N * C;
#define CALL(D) hipLaunchKernelGGL((k_bwd_sums_c8<D, NC_DT_BF16>), dim3(sp, N * C / 8), dim3(256), 0, s, (const uint4*)g, gctot / 8, gc0 / 8, (const uint4*)x, mean, rstd, slope, C / 8, S, sp, part)
  DISPATCH_DT(dt, CALL);
#undef CALL
  hipLaunchKernelGGL(k_bwd_final_c8, dim3((unsigned)(N * C / 8)), dim3(64), 0, s, part, N * C, sp, S, m1, m2);
#define CALL(D) hipLaunchKernelGGL((k_bwd_apply_c8<D, NC_DT_BF16>), dim3((unsigned)nbx, N * C / 8), dim3(256), 0, s, (const uint4*)g, gctot / 8, gc0 / 8, (const uint4*)x, mean, rstd, m1, m2, slope, C / 8, S, 8, (uint4*)dx, dbias ? dbp : nullptr)
  DISPATCH_DT(dt, CALL);
#undef CALL
  if (dbias)
    hipLaunchKernelGGL(k_dbias_final_c8, dim3(C), dim3(256), 0, s, dbp, N, C, (int)nbx, dbias);
  return check_launch("c8_instnorm_bwd");
}

int c8_maxpool_fwd(const void* x, int ctot, int c0, void* y, int N, int C, int D, int H, int W, int dt, hipStream_t s) {
  if (C % 8 || ctot % 8 || c0 % 8 || c0 + C > ctot || (D & 1) || (H & 1) || (W & 1)) { set_error("c8_maxpool_fwd: bad shape"); return NC_ERR_SHAPE; }
  const long So = (long)(D / 2) * (H / 2) * (W / 2);
#define CALL(T) hipLaunchKernelGGL((k_pool_c8<T>), dim3((unsigned)cdiv(So, 256), N * C / 8), dim3(256), 0, s, (const uint4*)x, ctot / 8, c0 / 8, (uint4*)y, C / 8, D, H, W)
  DISPATCH_DT(dt, CALL);
#undef CALL
  return check_launch("c8_maxpool_fwd");
}

int c8_maxpool_bwd_add(const void* dp, const void* x, int ctot, int c0, const void* skip, int sctot, int sc0, void* dx, int N, int C,
                       int D, int H, int W, int dt, hipStream_t s) {
  if (C % 8 || ctot % 8 || c0 % 8 || c0 + C > ctot || sctot % 8 || sc0 % 8 || sc0 + C > sctot || (D & 1) || (H & 1) || (W & 1)) {
    set_error("c8_maxpool_bwd_add: bad shape");
    return NC_ERR_SHAPE;
  }
  const long So = (long)(D / 2) * (H / 2) * (W / 2);
#define CALL(T) hipLaunchKernelGGL((k_pool_bwd_add_c8<T, NC_DT_BF16>), dim3((unsigned)cdiv(So, 256), N * C / 8), dim3(256), 0, s, (const uint4*)dp, (const uint4*)x, ctot / 8, c0 / 8, (const uint4*)skip, sctot / 8, sc0 / 8, (uint4*)dx, C / 8, D, H, W)
  DISPATCH_DT(dt, CALL);
#undef CALL
  return check_launch("c8_maxpool_bwd_add");
}

int c8_to_f32(const void* x, int ctot, int c0, float* y, int N, int C, long S, int dt, hipStream_t s) {
  if (C % 8 || ctot % 8 || c0 % 8 || c0 + C > ctot) { set_error("c8_to_f32: bad channel range"); return NC_ERR_SHAPE; }
#define CALL(T) hipLaunchKernelGGL((k_from_c8<T>), dim3((unsigned)cdiv(S, 256), N * C / 8), dim3(256), 0, s, (const uint4*)x, ctot / 8, c0 / 8, y, C / 8, S)
  DISPATCH_DT(dt, CALL);
#undef CALL
  return check_launch("c8_to_f32");
}

int c8_dot64(const void* x, const float* w, const float* bias, float* out, int N, long S, int dt, hipStream_t s) {
#define CALL(T) hipLaunchKernelGGL((k_dot64_c8<T>), dim3((unsigned)cdiv(S, 256), N), dim3(256), 0, s, (const uint4*)x, w, bias, out, S)
  DISPATCH_DT(dt, CALL);
#undef CALL
  return check_launch("c8_dot64");
}

size_t c8_outer64_ws_bytes(int N, long S) { return (size_t)N * cdiv(S, 256 * 16) * 65 * sizeof(double) + 256; }

// dx (nullable) = w (x) g in bf16 C8; dw[64] = sum g x; db (nullable) = sum g
int c8_outer64(const float* g, const void* x, const float* w, void* dx, float* dw, float* db, int N, long S, int dt, void* ws,
               size_t wsb, hipStream_t s) {
  if (!ws || wsb < c8_outer64_ws_bytes(N, S)) { set_error("c8_outer64: workspace too small"); return NC_ERR_WS; }
  const long nbx = cdiv(S, 256 * 16);
  double* part = (double*)ws;
#define CALL(T) hipLaunchKernelGGL((k_outer64_c8<T, NC_DT_BF16>), dim3((unsigned)nbx, N), dim3(256), 0, s, g, (const uint4*)x, w, (uint4*)dx, S, 16, part)
  DISPATCH_DT(dt, CALL);
#undef CALL
  hipLaunchKernelGGL(k_outer64_final, dim3(65), dim3(256), 0, s, part, (int)(N * nbx), dw, db);
  return check_launch("c8_outer64");
}

}  // namespace nc
