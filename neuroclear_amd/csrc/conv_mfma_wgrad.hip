// Weight gradient of Conv3d (3^3 / 5^3, stride 1, same padding) on the fp32 matrix cores of gfx950.
// Replaces autograd's backward-weights of nn.Conv3d (loss_G.backward(), apollo_model.py:283) for the layers of
// models/networks.py:420-425,442,460-469,900-902.
//
// GEMM view:  dW[co][(ci,tap)] = sum_{n,v} dY[co][v] * X[ci][v + tap]        (M = co, N = ci x taps, K = voxels)
//   * weight stationary: a workgroup owns the accumulators of ONE dz plane of the kernel for a 64 x CIW block of
//     (co, ci) -- KS^2 taps x (AB x 16 co) x 16 ci per wave in registers (v_mfma_f32_16x16x4_f32, exact fp32) -- and
//     streams its share of the volume through LDS.  dz planes / channel blocks are workgroup groups (grid.y), voxel
//     partitions are grid.x; partial slabs + a fixed-order reduce kernel (deterministic, no atomics).
//   * ROW STREAMING: the unit of work is one full image row (fixed n, z, y; all W columns).  Each step stages ONE new
//     dY row [64 co][W] and ONE new X row [CIW ci][W] (row y+PAD of plane z+dz-PAD) into a ring of KS rows, so X is
//     read once per dz group with no halo re-reads, and tile y's taps (dy, dx) are ring-slot + column offsets.
//     Staging needs no per-lane index arithmetic at all: a wave copies whole rows, lane = column, the row's base
//     address is wave-uniform (scalar unit).  Global loads of step s+1 are issued before the MFMA loop of step s.
//   * LDS images are channel-major with pitch == 2 (mod 32): the 16 channels x 2 voxels of a half-wave hit 32 distinct
//     banks for every tap shift (ds_read_b32); dY pad columns stay zero, so the 4-voxel k-steps may run past W.
#include "common.hpp"

namespace nc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kNSEG = 3;  // 64-column segments per row: W <= 192
static constexpr int kLdsMaxW = 160 * 1024;

struct WgParams {
  const float* x;
  const float* dy;
  float* slab;
  int C, K, N, D, H, W;
  int PR, PAr, QT4;    // X row pitch per channel, dY pitch per channel (floats), W rounded up to 4
  int CB, KBK, parts;  // channel blocks (C/CIW, K/64), voxel partitions
};

template <int KS, int AB>
__global__ __launch_bounds__(512) void k_wgrad_mfma(WgParams p) {
  constexpr int PAD = KS / 2;
  constexpr int T = KS * KS;
  constexpr int CIW = 32 * AB;    // input channels per workgroup
  constexpr int NCIB = CIW / 16;  // ci blocks (waves along ci)
  constexpr int XR = CIW / 8;     // X rows staged per wave per step
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
  // output planes z whose input plane z + dz - PAD exists
  const int zlo = max(0, PAD - dz), zhi = min(p.D, p.D + PAD - dz);
  const int Dv = max(0, zhi - zlo);
  const long nrows = (long)p.N * Dv * p.H;
  const long r0 = nrows * part / p.parts, r1 = nrows * (part + 1) / p.parts;

  const int SLOT = CIW * p.PR;  // one ring slot = one X row of all CIW channels
  float* xT = lds;
  float* dyT = lds + KS * SLOT;
  for (int i = tid; i < KS * SLOT + 64 * p.PAr; i += 512) lds[i] = 0.f;

  f32x4 acc[AB][T];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging registers: XR x-rows and 8 dY-rows per wave, kNSEG segments of 64 columns each
  float sx[XR][kNSEG], sd[8][kNSEG];
  const bool c0 = lane < p.W, c1 = lane + 64 < p.W, c2 = lane + 128 < p.W;

  // a "load step" brings X row yy (plane z + dz - PAD) into the ring and, when it completes a tile, dY row yy - PAD.
  auto issue_loads = [&](int n, int z, int yy, bool with_dy) {
    const bool xrow_ok = yy >= 0 && yy < p.H;
    const float* xb = p.x + ((long)n * p.C + cb * CIW + wave) * S + (long)(z + dz - PAD) * HW + (long)yy * p.W + lane;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const float* r = xb + (long)(8 * j) * S;
      sx[j][0] = (xrow_ok && c0) ? r[0] : 0.f;
      sx[j][1] = (xrow_ok && c1) ? r[64] : 0.f;
      sx[j][2] = (xrow_ok && c2) ? r[128] : 0.f;
    }
    if (with_dy) {
      const float* db = p.dy + ((long)n * p.K + kbk * 64 + wave) * S + (long)z * HW + (long)(yy - PAD) * p.W + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* r = db + (long)(8 * j) * S;
        sd[j][0] = c0 ? r[0] : 0.f;
        sd[j][1] = c1 ? r[64] : 0.f;
        sd[j][2] = c2 ? r[128] : 0.f;
      }
    }
  };
  auto write_lds = [&](int yy, bool with_dy) {
    const int slot = (yy + 8 * KS) % KS;
    float* xs = xT + slot * SLOT + wave * p.PR + PAD + lane;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      float* r = xs + (8 * j) * p.PR;
      if (c0) r[0] = sx[j][0];
      if (c1) r[64] = sx[j][1];
      if (c2) r[128] = sx[j][2];
    }
    if (with_dy) {
      float* ds = dyT + wave * p.PAr + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float* r = ds + (8 * j) * p.PAr;
        if (c0) r[0] = sd[j][0];
        if (c1) r[64] = sd[j][1];
        if (c2) r[128] = sd[j][2];
      }
    }
  };

  // ---- walk the rows [r0, r1): per (n, z) plane segment [ya, yb) the load steps run yy = ya-PAD .. yb-1+PAD
  long r = r0;
  int n = 0, z = 0, ya = 0, yb = 0, yy = 0;
  bool have = false;
  auto next_segment = [&]() {
    if (r >= r1) {
      have = false;
      return;
    }
    const long plane = r / p.H;
    ya = (int)(r - plane * p.H);
    const long rem = r1 - r;
    yb = (int)min((long)p.H, ya + rem);
    n = (int)(plane / Dv);
    z = zlo + (int)(plane - (long)n * Dv);
    yy = ya - PAD;
    r += yb - ya;
    have = true;
  };
  next_segment();
  if (have) issue_loads(n, z, yy, yy - PAD >= ya);

  const int a_off = (cog * 16 * AB + l15) * p.PAr + kq;
  const int b_off = (cib * 16 + l15) * p.PR + kq;
  while (have) {
    const int cyy = yy, cya = ya;
    const bool cdy = cyy - PAD >= cya;  // this step completes tile y = cyy - PAD
    __syncthreads();                    // the previous tile's MFMAs are done: ring slot and dY buffer are free
    write_lds(cyy, cdy);
    __syncthreads();
    // advance and prefetch the next step's rows while this tile is being multiplied
    ++yy;
    if (yy > yb - 1 + PAD) next_segment();
    if (have) issue_loads(n, z, yy, yy - PAD >= ya);
    if (cdy) {
      const int y = cyy - PAD;
      const float* pa = dyT + a_off;
      int soff[KS];
#pragma unroll
      for (int ty = 0; ty < KS; ++ty) soff[ty] = ((y - PAD + ty + 8 * KS) % KS) * SLOT + b_off;
#pragma unroll 1
      for (int q4 = 0; q4 < p.QT4; q4 += 4) {
        float a[AB];
#pragma unroll
        for (int k = 0; k < AB; ++k) a[k] = pa[q4 + k * 16 * p.PAr];
#pragma unroll
        for (int ty = 0; ty < KS; ++ty)
#pragma unroll
          for (int tx = 0; tx < KS; ++tx) {
            const float b = xT[soff[ty] + q4 + tx];
#pragma unroll
            for (int k = 0; k < AB; ++k)
              acc[k][ty * KS + tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b, acc[k][ty * KS + tx], 0, 0, 0);
          }
      }
    }
  }

  // ---- partial slab: slab[part][g][co 64][ci CIW][T];  C/D layout of 16x16 MFMA: col (ci) = lane & 15,
  //      row (co) = 4 * (lane >> 4) + r
  float* sl = p.slab + ((long)part * gridDim.y + g) * (64L * CIW * T);
#pragma unroll
  for (int k = 0; k < AB; ++k)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = cog * 16 * AB + k * 16 + 4 * kq + rr;
        const int ci = cib * 16 + l15;
        sl[((long)co * CIW + ci) * T + t] = acc[k][t][rr];
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
  int AB, PR, PAr, QT4, lds_bytes, G, parts;
};

static int pitch2(int n) {  // smallest p >= n with p % 32 == 2
  int p = (n / 32) * 32 + 2;
  return p >= n ? p : p + 32;
}

static bool wg_shape_ok(const ConvDims& d) {
  if (d.kd != d.kh || d.kh != d.kw) return false;
  if (d.kd != 3 && d.kd != 5) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.pd != d.kd / 2 || d.ph != d.kd / 2 || d.pw != d.kd / 2) return false;
  if (d.W > 64 * kNSEG || d.K % 64) return false;
  return true;
}

static bool plan_wgrad(const ConvDims& d, WgPlan& pl) {
  if (!wg_shape_ok(d)) return false;
  const int KS = d.kd;
  const int QT4 = (d.W + 3) & ~3;
  const int PR = pitch2(QT4 + KS + 1), PAr = pitch2(QT4 + 1);
  const int abList[2] = {KS == 3 ? 2 : 1, 1};
  for (int i = 0; i < 2; ++i) {
    const int AB = abList[i], CIW = 32 * AB;
    if (d.C % CIW) continue;
    const long bytes = ((long)KS * CIW * PR + 64L * PAr) * 4;
    if (bytes > kLdsMaxW) continue;
    pl.AB = AB; pl.PR = PR; pl.PAr = PAr; pl.QT4 = QT4; pl.lds_bytes = (int)bytes;
    pl.G = KS * (d.C / CIW) * (d.K / 64);
    int parts = 256 / pl.G;
    if (parts < 1) parts = 1;
    const long rows = (long)d.N * d.D * d.H;
    if (parts > rows) parts = (int)rows;
    pl.parts = parts;
    return true;
  }
  return false;
}

bool mfma_wgrad_supported(const ConvDims& d) {
  WgPlan pl;
  return plan_wgrad(d, pl);
}

size_t mfma_ws_bytes(const ConvDims& d) {
  size_t need = mfma_fwd_ws_bytes(d);  // packed weights + stream-K partial slots of the fwd / dgrad kernels
  WgPlan pl;
  if (plan_wgrad(d, pl)) {
    const size_t slab = (size_t)pl.parts * pl.G * 64 * (32 * pl.AB) * d.kd * d.kd * sizeof(float);
    if (slab > need) need = slab;
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
  if (!plan_wgrad(d, pl)) {
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
  p.PR = pl.PR; p.PAr = pl.PAr; p.QT4 = pl.QT4;
  p.CB = d.C / CIW; p.KBK = d.K / 64; p.parts = pl.parts;
  dim3 grid(pl.parts, pl.G);
  int e;
  if (d.kd == 3 && pl.AB == 2) e = launch_wg<3, 2>(p, grid, pl.lds_bytes, s);
  else if (d.kd == 3) e = launch_wg<3, 1>(p, grid, pl.lds_bytes, s);
  else e = launch_wg<5, 1>(p, grid, pl.lds_bytes, s);
  if (e) return e;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(1024), dim3(256), 0, s, (const float*)ws, dw, d.C, d.K, d.kd, CIW, pl.parts,
                     pl.G);
  return check_launch("wgrad_reduce");
}

}  // namespace nc
