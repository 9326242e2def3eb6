// Generic implicit-GEMM convolution on the fp32 matrix cores (gfx950): any kernel size, stride, padding, 2-D or 3-D.
// Covers what the brick kernels (conv_mfma_*.hip) do not: the strided 4x4 PatchGAN convs (reference
// models/networks.py:1030-1057), the 1x1 tails (:507-508, :905-911) and the Cin = 1 first layers (:420, :899).
//
//   FWD   : Y[k][p]      = sum_r  W[k][r]        * Xcol[r][p]      r = (c, tap), p = (n, od, oh, ow)
//   DGRAD : dX[c][q]     = sum_r' W'[c][r']      * dYcol[r'][q]    r' = (k, tap), q = (n, iz, iy, ix)
//   WGRAD : dW[k][r]     = sum_p  dY[k][p]       * Xcol[r][p]      (reduction over positions, split over grid.z)
//
// One workgroup = 4 waves = a 64 x 64 output tile, v_mfma_f32_32x32x2_f32 (one 32x32 accumulator per wave); the
// reduction runs in chunks of 16 staged through LDS as [r][64] images (32 consecutive floats per half-wave read:
// conflict-free).  The B image is gathered (im2col on the fly): the reduction index of a gathered element is
// wave-uniform, so its (channel, tap) decode lives on the scalar unit; the per-lane output position is decoded once.
// Register-staged double buffering (global loads of chunk i+1 are issued before the MFMAs of chunk i).
#include <cstdlib>

#include "common.hpp"

namespace nc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// G_DGRAD_P: data gradient of a STRIDED convolution, one output-parity class per grid.z slice.  An input position
// (iz, iy, ix) only receives taps t = (i + pad) mod s, + s, + 2s ... in every strided dimension; the plain gather
// multiplies the other (s^dims - 1) / s^dims of the reduction by zeros.  Positions are enumerated per class
// (i = s * i' + parity), the reduction per class is r' = (k, sub-tap): 4x less work for the 4 x 4 stride-2 PatchGAN
// layers (networks.py:1030-1046), 8x in 3-D.
enum { G_FWD = 0, G_DGRAD = 1, G_WGRAD = 2, G_DGRAD_P = 3 };

// n / d for 0 <= n < 2^31 with m = ceil(2^32 / d): the high product is floor(n / d) or one more
struct Div {
  unsigned m, d;
};
static Div mkdiv(long d) {
  Div r;
  r.d = (unsigned)(d < 1 ? 1 : d);
  r.m = (unsigned)(((1ull << 32) + r.d - 1) / r.d);
  return r;
}
__device__ __forceinline__ int qdiv(int n, const Div& v) {
  if (v.d == 1) return n;
  unsigned q = __umulhi((unsigned)n, v.m);
  if ((unsigned long long)q * v.d > (unsigned)n) --q;
  return (int)q;
}


struct GemmParams {
  const float* a;   // FWD/DGRAD: weights; WGRAD: dy
  const float* b;   // FWD: x; DGRAD: dy; WGRAD: x
  const float* bias;
  float* out;       // FWD: y; DGRAD: dx; WGRAD: slab[split][K][R]
  ConvDims d;
  int M, N, R;      // GEMM sizes (rows, columns, reduction)
  int taps, khw;    // kd*kh*kw, kh*kw
  long S, So;       // input / output positions per image
  int splits, rper; // WGRAD: reduction range per split (multiple of 16)
  // G_DGRAD_P: class grid (ceil(D/sd), ceil(H/sh), ceil(W/sw)), sub-taps per dimension (k / s), their product
  int cd, ch, cw, td, th, tw, ptaps;
  Div dtaps, dkhw, dkw, dSo, dHoWo, dWo, dptaps, dthtw, dtw;  // divisors of the per-chunk index decodes
  Div dsd, dsh, dsw;                                          // strides (data gradient: output index = u / stride)
  int shd, shh, shw;                                          // G_DGRAD_P: log2 of the (power-of-two) strides
};

// TM = rows of the output tile: 64 (4 waves as 2 x 2, one 32 x 32 accumulator each) or 128 (4 waves stacked along M, two
// accumulators each).  The gathered B image -- the expensive part: ~12 vector instructions of index arithmetic and
// bounds checks per element -- is shared by twice as many rows in the 128-row tile, which is what the long batched
// GEMMs of the Athena discriminators (M = 128..512, N = 10^4..10^5 columns) are bound by.
// KG = wave groups per workgroup: with KG = 2 a second set of four waves takes the other half of every reduction chunk
// (its own accumulators, added through LDS at the end).  Twice the waves on the same tile and LDS: a single 4-wave
// workgroup per ~2 tiles per CU leaves the matrix pipe 35-47 % busy (profiles/README.md), the staging work per thread
// halves as well.
template <int MODE, int TM, int KG>
__global__ __launch_bounds__(256 * KG) void k_conv_gemm(GemmParams p) {
  constexpr int kAP = TM + 1;               // pitch of the A image (floats)
  constexpr int WN = TM == 64 ? 2 : 1;      // waves along N
  constexpr int NBLK = 2 / WN;              // 32-column accumulator blocks per wave
  constexpr int MBLK = TM == 256 ? 2 : 1;   // 32-row accumulator blocks per wave
  constexpr int AJ = TM / (16 * KG);        // A elements per thread and chunk
  constexpr int BJ = 4 / KG;                // gathered B elements per thread and chunk
  constexpr int NW = 4 * KG;                // waves
  __shared__ float As[2][16 * kAP];
  __shared__ float Bs[2][16 * 64];
  const ConvDims& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int kg = wave >> 2, w4 = wave & 3;  // reduction half, position of the wave in the tile
  const int wm = w4 / WN, wn = w4 % WN;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * 64;
  // the reduction may be split over grid.z (small outputs with a long reduction: PatchGAN tail layers, wgrad)
  const int zsplit = MODE == G_DGRAD_P ? blockIdx.z % p.splits : blockIdx.z;
  const int cls = MODE == G_DGRAD_P ? blockIdx.z / p.splits : 0;
  const int r_begin = p.splits > 1 ? zsplit * p.rper : 0;
  const int r_end = p.splits > 1 ? min(p.R, r_begin + p.rper) : p.R;
  // parity of this class and first valid tap per dimension
  const int px = cls % d.sw, py = (cls / d.sw) % d.sh, pz = cls / (d.sw * d.sh);
  const int t0x = (px + d.pw) % d.sw, t0y = (py + d.ph) % d.sh, t0z = (pz + d.pd) % d.sd;
  const int HW = d.H * d.W, HoWo = d.Ho * d.Wo;

  // ---- per-lane column (n) decode for the gathered B image: n_l = tid & 63
  const int ncol = n0 + (tid & 63);
  bool ncol_ok = ncol < p.N;
  int nb = 0, c0 = 0, c1 = 0, c2 = 0;  // FWD: (n, od, oh, ow); DGRAD: (n, iz, iy, ix); WGRAD: (c, tz, ty, tx)
  if (MODE == G_DGRAD_P && ncol_ok) {
    const int cs = p.cd * p.ch * p.cw;
    nb = ncol / cs;
    const int pos = ncol - nb * cs;
    const int z1 = pos / (p.ch * p.cw), y1 = (pos - z1 * p.ch * p.cw) / p.cw, x1 = pos - z1 * p.ch * p.cw - y1 * p.cw;
    c0 = z1 * d.sd + pz; c1 = y1 * d.sh + py; c2 = x1 * d.sw + px;
    ncol_ok = c0 < d.D && c1 < d.H && c2 < d.W;
  } else if (ncol_ok) {
    if (MODE == G_FWD) {
      nb = ncol / (int)p.So;
      const int pos = ncol - nb * (int)p.So;
      c0 = pos / HoWo; c1 = (pos - c0 * HoWo) / d.Wo; c2 = pos - c0 * HoWo - c1 * d.Wo;
    } else if (MODE == G_DGRAD) {
      nb = ncol / (int)p.S;
      const int pos = ncol - nb * (int)p.S;
      c0 = pos / HW; c1 = (pos - c0 * HW) / d.W; c2 = pos - c0 * HW - c1 * d.W;
    } else {
      nb = ncol / p.taps;  // channel c
      const int tap = ncol - nb * p.taps;
      c0 = tap / p.khw; c1 = (tap - c0 * p.khw) / d.kw; c2 = tap - c0 * p.khw - c1 * d.kw;
    }
  }
  // G_DGRAD_P: u = c + pad - t0 - s * j is a multiple of the stride, so the output index is (c + pad - t0) / s - j: the
  // quotient is a per-lane constant, a gathered element costs three subtractions and three unsigned compares
  int boz = 0, boy = 0, box = 0;
  long bo_off = 0;
  if (MODE == G_DGRAD_P) {
    boz = (c0 + d.pd - t0z) >> p.shd; boy = (c1 + d.ph - t0y) >> p.shh; box = (c2 + d.pw - t0x) >> p.shw;
    bo_off = (long)nb * d.K * p.So + (long)boz * HoWo + boy * d.Wo + box;
  }
  // ---- A image assignment: r_l = tid & 15 (fast, contiguous in memory), m_l = (tid >> 4) + 16 j
  const int ar = tid & 15, am = tid >> 4;

  float ra[AJ], rb[BJ];
  auto load_chunk = [&](int r0) {
    // A
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = m0 + am + 16 * KG * j, r = r0 + ar;
      float v = 0.f;
      if (m < p.M && r < r_end) {
        if (MODE == G_FWD) {
          v = p.a[(long)m * p.R + r];
        } else if (MODE == G_DGRAD) {
          const int k = qdiv(r, p.dtaps), tap = r - k * p.taps;
          v = p.a[((long)k * d.C + m) * p.taps + tap];
        } else if (MODE == G_DGRAD_P) {
          const int k = qdiv(r, p.dptaps), sub = r - k * p.ptaps;
          const int jz = qdiv(sub, p.dthtw), s2 = sub - jz * p.th * p.tw;
          const int jy = qdiv(s2, p.dtw), jx = s2 - jy * p.tw;
          const int tap = ((t0z + d.sd * jz) * d.kh + t0y + d.sh * jy) * d.kw + t0x + d.sw * jx;
          v = p.a[((long)k * d.C + m) * p.taps + tap];
        } else {
          const int b = d.N > 1 ? qdiv(r, p.dSo) : 0;
          const int pos = r - b * (int)p.So;
          v = p.a[((long)b * d.K + m) * p.So + pos];
        }
      }
      ra[j] = v;
    }
    // B (gathered): reduction index is wave-uniform
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int r = r0 + wave + NW * j;
      float v = 0.f;
      if (r < r_end && ncol_ok) {
        if (MODE == G_FWD) {
          const int c = qdiv(r, p.dtaps), tap = r - c * p.taps;
          const int tz = qdiv(tap, p.dkhw), t2 = tap - tz * p.khw;
          const int ty = qdiv(t2, p.dkw), tx = t2 - ty * d.kw;
          const int iz = c0 * d.sd - d.pd + tz, iy = c1 * d.sh - d.ph + ty, ix = c2 * d.sw - d.pw + tx;
          if ((unsigned)iz < (unsigned)d.D && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W)
            v = p.b[((long)nb * d.C + c) * p.S + (long)iz * HW + iy * d.W + ix];
        } else if (MODE == G_DGRAD) {
          const int k = qdiv(r, p.dtaps), tap = r - k * p.taps;
          const int tz = qdiv(tap, p.dkhw), t2 = tap - tz * p.khw;
          const int ty = qdiv(t2, p.dkw), tx = t2 - ty * d.kw;
          const int uz = c0 + d.pd - tz, uy = c1 + d.ph - ty, ux = c2 + d.pw - tx;
          if (uz >= 0 && uy >= 0 && ux >= 0) {
            const int od = qdiv(uz, p.dsd), oh = qdiv(uy, p.dsh), ow = qdiv(ux, p.dsw);
            if (od * d.sd == uz && oh * d.sh == uy && ow * d.sw == ux && od < d.Do && oh < d.Ho && ow < d.Wo)
              v = p.b[((long)nb * d.K + k) * p.So + (long)od * HoWo + oh * d.Wo + ow];
          }
        } else if (MODE == G_DGRAD_P) {
          const int k = qdiv(r, p.dptaps), sub = r - k * p.ptaps;
          const int jz = qdiv(sub, p.dthtw), s2 = sub - jz * p.th * p.tw;
          const int jy = qdiv(s2, p.dtw), jx = s2 - jy * p.tw;
          const int od = boz - jz, oh = boy - jy, ow = box - jx;
          if ((unsigned)od < (unsigned)d.Do && (unsigned)oh < (unsigned)d.Ho && (unsigned)ow < (unsigned)d.Wo)
            v = p.b[bo_off + ((long)k * p.So - ((long)jz * HoWo + jy * d.Wo + jx))];
        } else {
          const int b = d.N > 1 ? qdiv(r, p.dSo) : 0;
          const int pos = r - b * (int)p.So;
          const int od = qdiv(pos, p.dHoWo), p2 = pos - od * HoWo;
          const int oh = qdiv(p2, p.dWo), ow = p2 - oh * d.Wo;
          const int iz = od * d.sd - d.pd + c0, iy = oh * d.sh - d.ph + c1, ix = ow * d.sw - d.pw + c2;
          if ((unsigned)iz < (unsigned)d.D && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W)
            v = p.b[((long)b * d.C + nb) * p.S + (long)iz * HW + iy * d.W + ix];
        }
      }
      rb[j] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) As[buf][ar * kAP + am + 16 * KG * j] = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j) Bs[buf][(wave + NW * j) * 64 + (tid & 63)] = rb[j];
  };

  f32x16 acc[MBLK][NBLK];
#pragma unroll
  for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
    for (int nb2 = 0; nb2 < NBLK; ++nb2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb2][r] = 0.f;

  int buf = 0;
  if (r_begin < r_end) {
    load_chunk(r_begin);
    store_chunk(0);
  }
  __syncthreads();
  for (int r0 = r_begin; r0 < r_end; r0 += 16) {
    const bool more = r0 + 16 < r_end;
    if (more) load_chunk(r0 + 16);
    const float* A = As[buf] + wm * 32 * MBLK + li + h * kAP;
    const float* B = Bs[buf] + wn * 32 * NBLK + li + h * 64;
#pragma unroll
    for (int kq = 0; kq < 8 / KG; ++kq) {
      const int kk = kg * (8 / KG) + kq;
      float av[MBLK], bv[NBLK];
#pragma unroll
      for (int mb = 0; mb < MBLK; ++mb) av[mb] = A[2 * kk * kAP + mb * 32];
#pragma unroll
      for (int nb2 = 0; nb2 < NBLK; ++nb2) bv[nb2] = B[2 * kk * 64 + nb2 * 32];
#pragma unroll
      for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
        for (int nb2 = 0; nb2 < NBLK; ++nb2)
          acc[mb][nb2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb], bv[nb2], acc[mb][nb2], 0, 0, 0);
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- KG = 2: the second wave group hands its accumulators over through LDS (the A image is free now), one 32-row
  //      block at a time: [w4][nb2][r][lane]
  if constexpr (KG == 2) {
    float* X = &As[0][0];
    static_assert(KG == 1 || 2 * 16 * (TM + 1) >= 4 * 16 * 64, "LDS scratch of the accumulator hand-over");
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
      for (int nb2 = 0; nb2 < NBLK; ++nb2) {  // one 32 x 32 block per wave and round: [w4][r][lane]
        __syncthreads();
        if (kg == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) X[(w4 * 16 + r) * 64 + lane] = acc[mb][nb2][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][nb2][r] += X[(w4 * 16 + r) * 64 + lane];
        }
      }
    if (kg == 1) return;
  }
  // ---- epilogue: row (m) = (r&3) + 8*(r>>2) + 4*h, column (n) = li
#pragma unroll
  for (int nb2 = 0; nb2 < NBLK; ++nb2) {
  const int n = n0 + (wn * NBLK + nb2) * 32 + li;
  if (n < p.N) {
    long base;
    long mstride;
    if (MODE == G_FWD) {
      const int b = n / (int)p.So;
      base = (long)blockIdx.z * d.N * d.K * p.So + (long)b * d.K * p.So + (n - (long)b * p.So);
      mstride = p.So;
    } else if (MODE == G_DGRAD) {
      const int b = n / (int)p.S;
      base = (long)blockIdx.z * d.N * d.C * p.S + (long)b * d.C * p.S + (n - (long)b * p.S);
      mstride = p.S;
    } else if (MODE == G_DGRAD_P) {
      const int cs = p.cd * p.ch * p.cw;
      const int b = n / cs;
      const int pos = n - b * cs;
      const int z1 = pos / (p.ch * p.cw), y1 = (pos - z1 * p.ch * p.cw) / p.cw, x1 = pos - z1 * p.ch * p.cw - y1 * p.cw;
      const int iz = z1 * d.sd + pz, iy = y1 * d.sh + py, ix = x1 * d.sw + px;
      if (iz >= d.D || iy >= d.H || ix >= d.W) continue;
      base = (long)zsplit * d.N * d.C * p.S + (long)b * d.C * p.S + (long)iz * HW + iy * d.W + ix;
      mstride = p.S;
    } else {
      base = (long)blockIdx.z * p.M * p.N + n;
      mstride = p.N;
    }
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * MBLK + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < p.M) {
        float v = acc[mb][nb2][r];
        if (MODE == G_FWD && p.bias && p.splits == 1) v += p.bias[m];
        p.out[base + (long)m * mstride] = v;
      }
    }
  }
  }
}

// out[i] = bias[(i / inner) % K] + sum_k slab[k][i]   (fixed order: deterministic)
__global__ void k_gemm_split_reduce(const float* __restrict__ slab, float* __restrict__ out, long n, int splits,
                                    const float* __restrict__ bias, long inner, int K) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = bias ? bias[(i / inner) % K] : 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(long)k * n + i];
    out[i] = s;
  }
}

// Stride-1 data gradient = forward convolution of dy with the flipped, channel-transposed kernel and padding k - 1 - p:
// wt[c][k][t] = w[k][c][T - 1 - t].  The forward gather has no stride divisions and its lanes read consecutive
// positions (27 -> 67 TFLOP/s on the 256 -> 512 PatchGAN layer at Athena's batch).
__global__ void k_flip_transpose_w(const float* __restrict__ w, float* __restrict__ wt, int K, int C, int T) {
  const long total = (long)K * C * T;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int k = (int)((i / T) % K);
    const int c = (int)(i / ((long)T * K));
    wt[i] = w[((long)k * C + c) * T + (T - 1 - t)];
  }
}

static bool dgrad_as_fwd(const ConvDims& d) {
  return d.sd == 1 && d.sh == 1 && d.sw == 1 && d.kd - 1 - d.pd >= 0 && d.kh - 1 - d.ph >= 0 && d.kw - 1 - d.pw >= 0 &&
         d.ph == d.pw && (d.kd == 1 || d.pd == d.ph) && d.C >= 16;
}
static ConvDims flipped_dims(const ConvDims& d) {  // the forward problem whose output is dx
  ConvDims f = d;
  f.C = d.K; f.K = d.C;
  f.D = d.Do; f.H = d.Ho; f.W = d.Wo;
  f.pd = d.kd - 1 - d.pd; f.ph = d.kh - 1 - d.ph; f.pw = d.kw - 1 - d.pw;
  f.Do = d.D; f.Ho = d.H; f.Wo = d.W;
  return f;
}
static size_t wt_bytes(const ConvDims& d) { return (((size_t)d.K * d.C * d.kd * d.kh * d.kw * sizeof(float)) + 255) & ~(size_t)255; }

static void gemm_common(GemmParams& p, const ConvDims& d) {
  p.d = d;
  p.taps = d.kd * d.kh * d.kw;
  p.khw = d.kh * d.kw;
  p.S = (long)d.D * d.H * d.W;
  p.So = (long)d.Do * d.Ho * d.Wo;
  p.splits = 1;
  p.rper = 0;
  p.dtaps = mkdiv(p.taps); p.dkhw = mkdiv(p.khw); p.dkw = mkdiv(d.kw);
  p.dSo = mkdiv(p.So); p.dHoWo = mkdiv((long)d.Ho * d.Wo); p.dWo = mkdiv(d.Wo);
  p.dptaps = mkdiv(1); p.dthtw = mkdiv(1); p.dtw = mkdiv(1);
  p.dsd = mkdiv(d.sd); p.dsh = mkdiv(d.sh); p.dsw = mkdiv(d.sw);
}

static bool gemm_range_ok(const ConvDims& d) {
  const long S = (long)d.D * d.H * d.W, So = (long)d.Do * d.Ho * d.Wo;
  return (long)d.N * S < (1L << 31) && (long)d.N * So < (1L << 31) && (long)d.C * d.kd * d.kh * d.kw < (1L << 31) &&
         (long)d.K * d.kd * d.kh * d.kw < (1L << 31);
}

// MFMA tiles are 64 x 64: a GEMM with fewer than 16 rows wastes most of the matrix core, which only pays when the
// whole (padded) problem is small and the reduction is long enough to want the split-R parallelism anyway.
static bool padded_ok(long M, long N, long R) {
  if (M >= 16) return true;
  const double padded = 2.0 * cdiv(M, 64) * 64 * cdiv(N, 64) * 64 * (double)R;
  return R >= 256 && padded <= 30e9;
}
static long Rf(const ConvDims& d) { return (long)d.C * d.kd * d.kh * d.kw; }
static long Rd(const ConvDims& d) { return (long)d.K * d.kd * d.kh * d.kw; }
static long Po(const ConvDims& d) { return (long)d.N * d.Do * d.Ho * d.Wo; }
static long Pi(const ConvDims& d) { return (long)d.N * d.D * d.H * d.W; }

bool gemm_fwd_supported(const ConvDims& d) { return gemm_range_ok(d) && padded_ok(d.K, Po(d), Rf(d)); }
bool gemm_dgrad_supported(const ConvDims& d) { return gemm_range_ok(d) && padded_ok(d.C, Pi(d), Rd(d)); }
bool gemm_wgrad_supported(const ConvDims& d) { return gemm_range_ok(d) && padded_ok(d.K, Rf(d), Po(d)); }

static int pick_splits(long M, long N, long R) {
  const long tiles = cdiv(M, 64) * cdiv(N, 64);
  if (tiles >= 256) return 1;
  long s = cdiv(512, tiles);      // aim at >= 512 workgroups
  const long cap = cdiv(R, 128);  // at least 8 chunks of 16 per split
  if (s > cap) s = cap;
  if (s > 256) s = 256;
  return (int)(s < 1 ? 1 : s);
}
// with >= 128 rows the kernel uses 128-/256-row tiles (tile_rows): split the reduction further until those fill the chip
static int fill_splits(long M, long cols64, long R, long s) {
  if (M >= 128) {
    const long t128 = cdiv(M, 128) * cols64;
    while (t128 * s < 768 && s * 2 <= cdiv(R, 128) && s * 2 <= 256) s *= 2;
  }
  return (int)s;
}
static int fwd_splits(const ConvDims& d) {
  return fill_splits(d.K, cdiv(Po(d), 64), Rf(d), pick_splits(d.K, Po(d), Rf(d)));
}
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static bool dgrad_parity_ok(const ConvDims& d) {
  return (d.sd > 1 || d.sh > 1 || d.sw > 1) && d.kd % d.sd == 0 && d.kh % d.sh == 0 && d.kw % d.sw == 0 &&
         pow2(d.sd) && pow2(d.sh) && pow2(d.sw);
}
static int dgrad_splits(const ConvDims& d) {
  if (dgrad_as_fwd(d)) return fill_splits(d.C, cdiv(Pi(d), 64), Rd(d), pick_splits(d.C, Pi(d), Rd(d)));
  if (dgrad_parity_ok(d)) {
    const long ncls = (long)d.sd * d.sh * d.sw;
    const long npar = (long)d.N * cdiv(d.D, d.sd) * cdiv(d.H, d.sh) * cdiv(d.W, d.sw);
    return fill_splits(d.C, cdiv(npar, 64) * ncls, Rd(d) / ncls, pick_splits(d.C, npar * ncls, Rd(d) / ncls));
  }
  return pick_splits(d.C, Pi(d), Rd(d));
}
static int wgrad_splits(const ConvDims& d) {
  return fill_splits(d.K, cdiv(Rf(d), 64), Po(d), pick_splits(d.K, Rf(d), Po(d)));
}

size_t gemm_ws_bytes(const ConvDims& d) {
  size_t need = 0;
  auto upd = [&](bool ok, int splits, size_t elems) {
    if (ok && splits > 1 && (size_t)splits * elems * sizeof(float) > need) need = (size_t)splits * elems * sizeof(float);
  };
  upd(gemm_fwd_supported(d), fwd_splits(d), (size_t)d.K * Po(d));
  upd(gemm_dgrad_supported(d), dgrad_splits(d), (size_t)d.C * Pi(d));
  if (gemm_dgrad_supported(d) && dgrad_as_fwd(d)) need += wt_bytes(d);
  upd(gemm_wgrad_supported(d), wgrad_splits(d), (size_t)d.K * Rf(d));
  return need;
}

// rows of the output tile: 256 / 128 when the GEMM has that many rows and is large enough to fill the chip with such
// tiles (>= 1.5 / 2 workgroups per CU), else 64
static int tile_rows(const GemmParams& p) {
  constexpr int cap = 256;
  const int ncls = p.ptaps ? p.d.sd * p.d.sh * p.d.sw : 1;
  const long cols = cdiv(p.N, 64) * p.splits * ncls;
  if (cap >= 256 && p.M >= 256 && cdiv(p.M, 256) * cols >= 384) return 256;
  if (cap >= 128 && p.M >= 128 && cdiv(p.M, 128) * cols >= 512) return 128;
  return 64;
}
template <int MODE>
static void launch_gemm(const GemmParams& p, unsigned gz, hipStream_t s) {
  const int tm = tile_rows(p);
  dim3 grid((unsigned)cdiv(p.N, 64), (unsigned)cdiv(p.M, tm), gz);
  if (tm == 256) hipLaunchKernelGGL((k_conv_gemm<MODE, 256, 2>), grid, dim3(512), 0, s, p);
  else if (tm == 128) hipLaunchKernelGGL((k_conv_gemm<MODE, 128, 2>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((k_conv_gemm<MODE, 64, 1>), grid, dim3(256), 0, s, p);
}

static int split_setup(GemmParams& p, int splits, float* out, void* ws, size_t wsb, size_t out_elems,
                       const char* what) {
  p.splits = splits;
  p.rper = (int)(cdiv(cdiv(p.R, splits), 16) * 16);
  if (splits > 1) {
    if (!ws || wsb < (size_t)splits * out_elems * sizeof(float)) {
      set_error("%s: workspace too small", what);
      return NC_ERR_WS;
    }
    p.out = (float*)ws;
  } else {
    p.out = out;
  }
  return NC_OK;
}
static int split_reduce(const GemmParams& p, float* out, long n, const float* bias, long inner, int K,
                        hipStream_t s) {
  if (p.splits == 1) return NC_OK;
  const long nb = cdiv(n, 256);
  hipLaunchKernelGGL(k_gemm_split_reduce, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, s,
                     (const float*)p.out, out, n, p.splits, bias, inner, K);
  return check_launch("gemm_split_reduce");
}

int conv_fwd_gemm(const float* x, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                  hipStream_t s) {
  GemmParams p{};
  gemm_common(p, d);
  p.a = w; p.b = x; p.bias = b;
  p.M = d.K; p.N = (int)Po(d); p.R = (int)Rf(d);
  const long n = (long)d.K * Po(d);
  if (int e = split_setup(p, fwd_splits(d), y, ws, wsb, n, "conv_fwd_gemm")) return e;
  launch_gemm<G_FWD>(p, (unsigned)p.splits, s);
  if (int e = check_launch("conv_fwd_gemm")) return e;
  return split_reduce(p, y, n, b, p.So, d.K, s);
}

int conv_dgrad_gemm(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s) {
  GemmParams p{};
  gemm_common(p, d);
  if (dgrad_as_fwd(d)) {
    const size_t wb = wt_bytes(d);
    if (!ws || wsb < wb) { set_error("conv_dgrad_gemm: workspace too small"); return NC_ERR_WS; }
    float* wt = (float*)ws;
    const int T = d.kd * d.kh * d.kw;
    const long tot = (long)d.K * d.C * T;
    hipLaunchKernelGGL(k_flip_transpose_w, dim3((unsigned)(cdiv(tot, 256) > 1024 ? 1024 : cdiv(tot, 256))), dim3(256), 0, s, w,
                       wt, d.K, d.C, T);
    const ConvDims f = flipped_dims(d);
    gemm_common(p, f);
    p.a = wt; p.b = dy; p.bias = nullptr;
    p.M = f.K; p.N = (int)Po(f); p.R = (int)Rf(f);
    const long n = (long)f.K * Po(f);
    if (int e = split_setup(p, dgrad_splits(d), dx, (char*)ws + wb, wsb - wb, n, "conv_dgrad_gemm")) return e;
    launch_gemm<G_FWD>(p, (unsigned)p.splits, s);
    if (int e = check_launch("conv_dgrad_gemm_flipped")) return e;
    return split_reduce(p, dx, n, nullptr, p.So, f.K, s);
  }
  p.a = w; p.b = dy; p.bias = nullptr;
  if (dgrad_parity_ok(d)) {
    p.cd = (int)cdiv(d.D, d.sd); p.ch = (int)cdiv(d.H, d.sh); p.cw = (int)cdiv(d.W, d.sw);
    p.td = d.kd / d.sd; p.th = d.kh / d.sh; p.tw = d.kw / d.sw;
    p.ptaps = p.td * p.th * p.tw;
    p.dptaps = mkdiv(p.ptaps); p.dthtw = mkdiv((long)p.th * p.tw); p.dtw = mkdiv(p.tw);
    p.shd = ilog2(d.sd); p.shh = ilog2(d.sh); p.shw = ilog2(d.sw);
    const int ncls = d.sd * d.sh * d.sw;
    p.M = d.C; p.N = d.N * p.cd * p.ch * p.cw; p.R = d.K * p.ptaps;
    const long n = (long)d.C * Pi(d);
    if (int e = split_setup(p, dgrad_splits(d), dx, ws, wsb, n, "conv_dgrad_gemm")) return e;
    launch_gemm<G_DGRAD_P>(p, (unsigned)(p.splits * ncls), s);
    if (int e = check_launch("conv_dgrad_gemm_parity")) return e;
    return split_reduce(p, dx, n, nullptr, 1, 1, s);
  }
  p.M = d.C; p.N = (int)Pi(d); p.R = (int)Rd(d);
  const long n = (long)d.C * Pi(d);
  if (int e = split_setup(p, dgrad_splits(d), dx, ws, wsb, n, "conv_dgrad_gemm")) return e;
  launch_gemm<G_DGRAD>(p, (unsigned)p.splits, s);
  if (int e = check_launch("conv_dgrad_gemm")) return e;
  return split_reduce(p, dx, n, nullptr, 1, 1, s);
}

int conv_wgrad_gemm(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s) {
  GemmParams p{};
  gemm_common(p, d);
  p.a = dy; p.b = x; p.bias = nullptr;
  p.M = d.K; p.N = (int)Rf(d); p.R = (int)Po(d);
  const long n = (long)p.M * p.N;
  if (int e = split_setup(p, wgrad_splits(d), dw, ws, wsb, n, "conv_wgrad_gemm")) return e;
  launch_gemm<G_WGRAD>(p, (unsigned)p.splits, s);
  if (int e = check_launch("conv_wgrad_gemm")) return e;
  return split_reduce(p, dw, n, nullptr, 1, 1, s);
}

}  // namespace nc
