// Weight gradient of Conv3d (3^3 / 5^3, stride 1, same padding) on the fp32 matrix cores of gfx950.
// Replaces autograd's backward-weights of nn.Conv3d (loss_G.backward(), apollo_model.py:283) for the layers of
// models/networks.py:420-425,442,460-469,900-902.
//
// GEMM view:  dW[co][(ci,tap)] = sum_{n,v} dY[co][v] * X[ci][v + tap]        (M = co, N = ci x taps, K = voxels)
//   * "weight stationary": a workgroup owns the accumulators of ONE dz plane of the kernel for a 64 x CIW block of
//     (co, ci) -- KS^2 taps x (AB x 16 co) x 16 ci per wave in registers -- and walks over its share of voxel tiles.
//     The kernel's dz planes / channel blocks are separate workgroup groups (grid.y); voxel partitions are grid.x.
//   * v_mfma_f32_16x16x4_f32 (exact fp32): A = dY[16 co][4 voxels], B = X[4 voxels][16 ci], 4 accumulator registers per
//     tile, so 9 (25) taps x AB co-blocks fit a wave.
//   * tiles: Ty rows x Tx columns of one z plane, flattened with pitch Pp = Tx+2p (as in the forward kernel) so a tap is
//     a constant LDS offset.  dY pad positions are staged as ZERO, so whatever X holds there contributes nothing.
//   * LDS images are channel-major with pitch == 2 (mod 32) floats: the 16 channels x 2 voxels of a half-wave hit 32
//     distinct banks for any tap shift (ds_read_b32).
//   * register-staged double buffering across tiles (global loads for tile t+1 are issued before the MFMA loop of t).
//   * deterministic: partial slabs per workgroup + a fixed-order reduce kernel; no atomics.
#include "common.hpp"

namespace nc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kNLDW = 32;
static constexpr int kLdsMaxW = 160 * 1024;

struct WgParams {
  const float* x;
  const float* dy;
  float* slab;
  int C, K, N, D, H, W;
  int Ty, Tx, nty, ntx, Pp;
  int QT, QT4, RX, PA, PB, NEdy, NEx, nDy;  // floats: dy real region, rounded, x real region, pitches, element counts
  unsigned mPp, mQT, mRX;
  int CB, KBK, parts;  // channel blocks (C/CIW, K/64), voxel partitions
  unsigned mNyx, mNx;  // magics for nty*ntx and ntx
};

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m) { return __umulhi(n, m); }
// divisor may be 1 (magic would overflow): d is wave-uniform, so the select is a scalar branch
__device__ __forceinline__ unsigned fdivd(unsigned n, unsigned m, unsigned d) { return d == 1 ? n : __umulhi(n, m); }

template <int KS, int AB>
__global__ __launch_bounds__(512) void k_wgrad_mfma(WgParams p) {
  constexpr int NT = 512;
  constexpr int PAD = KS / 2;
  constexpr int T = KS * KS;
  constexpr int CIW = 32 * AB;     // input channels per workgroup
  constexpr int NCIB = CIW / 16;   // ci blocks (waves along ci)
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cib = wave % NCIB, cog = wave / NCIB;
  const int l15 = lane & 15, kq = lane >> 4;

  // group decode: g = (dz * CB + cb) * KBK + kbk
  const int g = blockIdx.y;
  const int kbk = g % p.KBK, cb = (g / p.KBK) % p.CB, dz = g / (p.KBK * p.CB);
  const int part = blockIdx.x;

  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  // valid output planes z for this dz: z + dz - PAD in [0, D)
  const int zlo = max(0, PAD - dz), zhi = min(p.D, p.D + PAD - dz);
  const int Dv = max(0, zhi - zlo);
  const int nyx = p.nty * p.ntx;
  const long ntiles = (long)p.N * Dv * nyx;
  const long t0 = ntiles * part / p.parts, t1 = ntiles * (part + 1) / p.parts;

  const int bufsz = 64 * p.PA + CIW * p.PB;
  float* buf0 = lds;
  float* buf1 = lds + bufsz;
  for (int i = tid; i < 2 * bufsz; i += NT) lds[i] = 0.f;
  __syncthreads();

  float st[kNLDW];
  auto stage_load = [&](long t) {
    const unsigned tt = (unsigned)t;
    const unsigned nz = fdivd(tt, p.mNyx, nyx);
    const unsigned yx = tt - nz * nyx;
    const unsigned tyi = fdivd(yx, p.mNx, p.ntx), txi = yx - tyi * p.ntx;
    const int n = (int)(nz / (unsigned)max(Dv, 1)), z = zlo + (int)(nz % (unsigned)max(Dv, 1));
    const int y0 = tyi * p.Ty, x0 = txi * p.Tx;
    const float* dyb = p.dy + ((long)n * p.K + kbk * 64) * S + (long)z * HW;
    const float* xb = p.x + ((long)n * p.C + cb * CIW) * S + (long)(z + dz - PAD) * HW;
    const int Si = (int)S;
    // buffer descriptors: an invalid element uses offset -1 and reads back 0 from the range check
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dyb, 0, (int)(64 * S * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)(CIW * S * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < kNLDW; ++i) {
      // iteration i is entirely dY (i < nDy) or entirely X: the choice is wave-uniform
      if (i < p.nDy) {
        const int e = tid + i * NT;
        const unsigned co = fdiv(e, p.mQT);
        const unsigned q = e - co * p.QT;
        const unsigned ty = fdiv(q, p.mPp), xx = q - ty * p.Pp;
        const int y = y0 + (int)ty, x = x0 + (int)xx;
        const bool ok = e < p.NEdy && (int)xx < p.Tx && x < p.W && y < p.H;
        const int off = ok ? ((int)co * Si + y * p.W + x) * 4 : -1;
        st[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, off, 0, 0));
      } else {
        const int e2 = tid + (i - p.nDy) * NT;
        const unsigned ci = fdiv(e2, p.mRX);
        const unsigned f = e2 - ci * p.RX;
        const unsigned yy = fdiv(f, p.mPp), xx = f - yy * p.Pp;
        const int y = y0 + (int)yy - PAD, x = x0 + (int)xx - PAD;
        const bool ok = e2 < p.NEx && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const int off = ok ? ((int)ci * Si + y * p.W + x) * 4 : -1;
        st[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep decode_i -> load_i together: 32 hoisted decodes would spill
    }
  };
  auto stage_store = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < kNLDW; ++i) {
      if (i < p.nDy) {
        const int e = tid + i * NT;
        const unsigned co = fdiv(e, p.mQT);
        if (e < p.NEdy) buf[e + co * (p.PA - p.QT)] = st[i];
      } else {
        const int e2 = tid + (i - p.nDy) * NT;
        const unsigned ci = fdiv(e2, p.mRX);
        if (e2 < p.NEx) buf[64 * p.PA + e2 + ci * (p.PB - p.RX)] = st[i];
      }
    }
  };

  f32x4 acc[AB][T];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (t0 < t1) {
    stage_load(t0);
    stage_store(buf0);
  }
  __syncthreads();

  const int a_off = (cog * 16 * AB + l15) * p.PA + kq;
  const int b_off = 64 * p.PA + (cib * 16 + l15) * p.PB + kq;
  for (long t = t0; t < t1; ++t) {
    const float* cur = ((t - t0) & 1) ? buf1 : buf0;
    float* nxt = ((t - t0) & 1) ? buf0 : buf1;
    const bool more = t + 1 < t1;
    if (more) stage_load(t + 1);
    const float* pa = cur + a_off;
    const float* pb = cur + b_off;
#pragma unroll 1
    for (int q4 = 0; q4 < p.QT4; q4 += 4) {
      float a[AB];
#pragma unroll
      for (int k = 0; k < AB; ++k) a[k] = pa[q4 + k * 16 * p.PA];
#pragma unroll
      for (int ty = 0; ty < KS; ++ty)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
          const float b = pb[q4 + ty * p.Pp + tx];
#pragma unroll
          for (int k = 0; k < AB; ++k)
            acc[k][ty * KS + tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b, acc[k][ty * KS + tx], 0, 0, 0);
        }
    }
    if (more) stage_store(nxt);
    __syncthreads();
  }

  // ---- partial slab: slab[part][g][co 64][ci CIW][T];  C/D layout of 16x16 MFMA: col (ci) = lane & 15,
  //      row (co) = 4 * (lane >> 4) + r
  float* sl = p.slab + ((long)part * gridDim.y + g) * (64L * CIW * T);
#pragma unroll
  for (int k = 0; k < AB; ++k)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cog * 16 * AB + k * 16 + 4 * kq + r;
        const int ci = cib * 16 + l15;
        sl[((long)co * CIW + ci) * T + t] = acc[k][t][r];
      }
}

// dw[k][c][dz*T + t] = sum_part slab[part][(dz*CB + c/CIW)*KBK + k/64][k%64][c%CIW][t]
__global__ void k_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ dw, int C, int K, int KS, int CIW,
                               int parts, int G) {
  const int T = KS * KS, taps = KS * T;
  const int CB = C / CIW, KBK = K / 64;
  const long total = (long)K * C * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps), c = (int)((i / taps) % C), k = (int)(i / ((long)taps * C));
    const int dz = tap / T, t = tap % T;
    const int g = (dz * CB + c / CIW) * KBK + k / 64;
    const long off = (long)g * (64L * CIW * T) + ((long)(k % 64) * CIW + (c % CIW)) * T + t;
    const long pstride = (long)G * (64L * CIW * T);
    float s = 0.f;
    for (int pidx = 0; pidx < parts; ++pidx) s += slab[off + pidx * pstride];
    dw[i] = s;
  }
}

struct WgPlan {
  int AB, Ty, Tx, nty, ntx, Pp, QT, QT4, RX, PA, PB, NEdy, NEx, nDy, lds_bytes, G, parts;
  double score;
};

static unsigned magicw(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }
static int pitch2(int n) {  // smallest p >= n with p % 32 == 2
  int p = (n / 32) * 32 + 2;
  return p >= n ? p : p + 32;
}

static bool plan_wgrad(const ConvDims& d, WgPlan& best) {
  const int KS = d.kd, pad = KS / 2;
  const int AB = KS == 3 ? 2 : 1;
  const int CIW = 32 * AB;
  if (d.C % CIW || d.K % 64) return false;
  bool found = false;
  for (int Ty = 1; Ty <= 8 && Ty <= d.H; ++Ty) {
    for (int ntx = 1; ntx <= 32; ++ntx) {
      const int Tx = (d.W + ntx - 1) / ntx;
      const int Pp = Tx + 2 * pad;
      const int QT = Ty * Pp, QT4 = (QT + 3) & ~3;
      const int RX = (Ty + 2 * pad) * Pp;
      const int PA = pitch2(QT4 + 4);
      const int PB = pitch2(QT4 + (KS - 1) * Pp + KS + 4 > RX ? QT4 + (KS - 1) * Pp + KS + 4 : RX);
      const int NEdy = 64 * QT, NEx = CIW * RX;
      const int nDy = (NEdy + 511) / 512, nX = (NEx + 511) / 512;
      const long bytes = 2L * (64 * PA + CIW * PB) * 4;
      if (nDy + nX > kNLDW || bytes > kLdsMaxW) continue;
      const int nty = (d.H + Ty - 1) / Ty;
      const int ntx2 = (d.W + Tx - 1) / Tx;
      // MFMA efficiency of the k loop x a mild preference for fewer, larger tiles (less staging per MFMA)
      const double eff = (double)d.H * d.W / ((double)nty * ntx2 * QT4);
      const double score = eff * (1.0 - 1.5 / (QT4 / 4 + 2));
      if (!found || score > best.score) {
        found = true;
        best = WgPlan{AB, Ty, Tx, nty, ntx2, Pp, QT, QT4, RX, PA, PB, NEdy, NEx, nDy, (int)bytes, 0, 0, score};
      }
    }
  }
  if (!found) return false;
  best.G = KS * (d.C / CIW) * (d.K / 64);
  int parts = 256 / best.G;
  if (parts < 1) parts = 1;
  const long tiles = (long)d.N * d.D * best.nty * best.ntx;
  if (parts > tiles) parts = (int)tiles;
  best.parts = parts;
  return true;
}

static bool wg_shape_ok(const ConvDims& d) {
  if (d.kd != d.kh || d.kh != d.kw) return false;
  if (d.kd != 3 && d.kd != 5) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.pd != d.kd / 2 || d.ph != d.kd / 2 || d.pw != d.kd / 2) return false;
  if ((long)d.N * d.D * d.H * d.W >= (1L << 31) || 64L * d.D * d.H * d.W * 4 >= (1L << 31)) return false;
  return true;
}

bool mfma_wgrad_supported(const ConvDims& d) {
  WgPlan pl;
  return wg_shape_ok(d) && plan_wgrad(d, pl);
}

size_t mfma_ws_bytes(const ConvDims& d) {
  size_t need = 0;
  if (wg_shape_ok(d)) {
    const size_t pack = (size_t)d.C * d.K * d.kd * d.kh * d.kw * sizeof(float);
    need = pack;
    WgPlan pl;
    if (plan_wgrad(d, pl)) {
      const int CIW = 32 * pl.AB;
      const size_t slab = (size_t)pl.parts * pl.G * 64 * CIW * d.kd * d.kd * sizeof(float);
      if (slab > need) need = slab;
    }
  }
  return need;
}

template <int KS, int AB>
static int launch_wg(const WgParams& p, dim3 grid, int lds_bytes, hipStream_t s) {
  auto kern = k_wgrad_mfma<KS, AB>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLdsMaxW) != hipSuccess) {
      set_error("wgrad_mfma: cannot raise dynamic LDS limit");
      return NC_ERR_HIP;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, s, p);
  return check_launch("wgrad_mfma");
}

int conv_wgrad_mfma(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s) {
  WgPlan pl;
  if (!wg_shape_ok(d) || !plan_wgrad(d, pl)) {
    set_error("wgrad_mfma: unsupported shape");
    return NC_ERR_SHAPE;
  }
  const int CIW = 32 * pl.AB;
  const size_t need = (size_t)pl.parts * pl.G * 64 * CIW * d.kd * d.kd * sizeof(float);
  if (!ws || wsb < need) {
    set_error("wgrad_mfma: workspace too small (%zu < %zu)", wsb, need);
    return NC_ERR_WS;
  }
  WgParams p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws;
  p.C = d.C; p.K = d.K; p.N = d.N; p.D = d.D; p.H = d.H; p.W = d.W;
  p.Ty = pl.Ty; p.Tx = pl.Tx; p.nty = pl.nty; p.ntx = pl.ntx; p.Pp = pl.Pp;
  p.QT = pl.QT; p.QT4 = pl.QT4; p.RX = pl.RX; p.PA = pl.PA; p.PB = pl.PB; p.NEdy = pl.NEdy; p.NEx = pl.NEx; p.nDy = pl.nDy;
  p.mPp = magicw(pl.Pp); p.mQT = magicw(pl.QT); p.mRX = magicw(pl.RX);
  p.CB = d.C / CIW; p.KBK = d.K / 64; p.parts = pl.parts;
  p.mNyx = magicw(pl.nty * pl.ntx); p.mNx = magicw(pl.ntx);
  dim3 grid(pl.parts, pl.G);
  int e;
  if (d.kd == 3) e = launch_wg<3, 2>(p, grid, pl.lds_bytes, s);
  else e = launch_wg<5, 1>(p, grid, pl.lds_bytes, s);
  if (e) return e;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(1024), dim3(256), 0, s, (const float*)ws, dw, d.C, d.K, d.kd, CIW, pl.parts,
                     pl.G);
  return check_launch("wgrad_reduce");
}

}  // namespace nc
