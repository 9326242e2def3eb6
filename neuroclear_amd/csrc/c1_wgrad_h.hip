// Weight gradient of the ONE-input-channel layers (reference models/networks.py:420 first U-Net layer 3^3, :899 first
// deep_linear_gen layer 7^3; 64 output channels) on the 16-bit matrix cores, for the 16-bit end-to-end path:
//   dW[k][dz][dy][dx] = sum over (n, z, y, x) of g[n][k][z][y][x] * xin[n][z + dz - p][y + dy - p][x + dx - p].
// The fp32 tap-axis kernel (conv_c1.hip, 95 TFLOP/s) cost 6.0 ms + 0.8 ms per configs[3] step, plus two 1 ms conversions of
// the C8 gradient to fp32 to feed it.  Here the K-dim of v_mfma_f32_32x32x16_bf16 is 16 consecutive voxels of an x row:
//   * B comes from PLANAR 16-bit rows (pitch PX, a multiple of 16, zero beyond W): the pseudo-channel planes
//     XP[n][j][row][x] = xin[row][x + j - p] (k_build_xp: the dx taps become 8 "channels", as in conv_h.hip's forward of the same
//     layer) -- a lane's 8 k values are 16 contiguous, aligned bytes of the row shifted by (dz, dy); the 32 columns are
//     n = (dy quad member, j): one MFMA covers 4 dy x 8 dx taps of one dz for 32 output channels;
//   * A comes straight from the C8 gradient row (8 blocks x W units of 16 bytes, staged as it is) through gfx950's transposing
//     LDS read ds_read_b64_tr_b16 -- per 16-lane group a 4-voxel x 16-channel block comes back channel-major, two reads build
//     the 8-voxel fragment (the lane roles of conv_h.hip's k_wgrad_h);
//   * a workgroup owns one (sample, z) plane and one group of 4 dz: its 8 waves are (channel half, dz); it walks the rows y
//     with the 2p + 1 rows of each of its dz planes in an 8-slot LDS ring and the gradient row double-buffered (LDS-DMA);
//   * one partial [64][4 dz][8 dy][8 dx] per workgroup, summed over the planes in a fixed order (k_c1_wgrad_reduce).
#include "common.hpp"

namespace nc {
NC_ZERO_PAGE()
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ unsigned short to_bf16(float f) {
  const __bf16 v = (__bf16)f;
  return __builtin_bit_cast(unsigned short, v);
}

// fp32 x [N][R][W] -> XP [N][8][R][PX]: plane j holds x shifted by j - p along the row (zero outside the row, for j >= KS and
// for x >= W)
__global__ void __launch_bounds__(256) k_build_xp(const float* __restrict__ x, unsigned short* __restrict__ xp, int W, int PX, int KS,
                                                  long R, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % PX);
  const long q = i / PX;
  const long r = q % R;
  const long n = q / R;
  const float* row = x + (n * R + r) * W;
  const int p = KS / 2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int xs = xx + j - p;
    const float v = (xx < W && j < KS && (unsigned)xs < (unsigned)W) ? row[xs] : 0.f;
    xp[((n * 8 + j) * R + r) * PX + xx] = to_bf16(v);
  }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* ltr_t;

struct CwParams {
  const uint4* gc8;          // C8 gradient [N][8][S][8]
  const unsigned short* xp;
  const float* zeros;
  float* part;
  int N, D, H, W, PX, PL, nxg;
};

template <int KS>
__global__ void __launch_bounds__(512) k_c1_wgrad_p(const CwParams p) {
  constexpr int P = KS / 2, NTT = KS > 4 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* const zrow = lds;                                   // [PL] zeros
  unsigned short* const Gs = lds + p.PL;                              // [2][8 blocks][PX units of 8 channels]
  unsigned short* const XPs = Gs + 2 * 64 * p.PX;                     // [4 dzl][8 slots][8 j][PL]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int mt = wave & 1, dzl = wave >> 1;
  const int n = blockIdx.x / p.D, z = blockIdx.x - n * p.D;
  const int grp = blockIdx.y;
  const long R = (long)p.D * p.H;
  const long S = R * p.W;
  const int nl = p.PX / 8;  // 16-byte lanes per planar row

  for (int i = tid; i < p.PL; i += 512) zrow[i] = 0;
  // units x >= W of the gradient rows are never copied: zero them once (their B operand is zero too, but 0 x garbage
  // could be NaN)
  for (int i = tid; i < 2 * 8 * (p.PX - p.W) * 8; i += 512) {
    const int e = i & 7, u = i >> 3;
    const int xr = u % (p.PX - p.W), bb = u / (p.PX - p.W);  // bb = buffer * 8 + block
    Gs[((long)bb * p.PX + p.W + xr) * 8 + e] = 0;
  }

  // staging: wave w copies block w of the gradient row (W units of 16 bytes, contiguous in the C8 tensor) and the
  // pseudo-channel rows (dzl', j) = 4 w .. 4 w + 3
  auto stage_g = [&](int y, int buf) {
    const uint4* src = p.gc8 + ((long)n * 8 + wave) * S + ((long)z * p.H + y) * p.W;
    unsigned short* dst = Gs + ((long)buf * 8 + wave) * p.PX * 8;
    for (int u0 = 0; u0 < p.W; u0 += 64)
      if (u0 + lane < p.W) nc_dma_lds16((src + u0 + lane), nc_lds_addr((dst + (long)u0 * 8)));
  };
  auto stage_x = [&](int yy) {  // input row yy (may be outside the plane) of this group's 4 dz planes -> ring slot (yy + P) & 7
    const int slot = (yy + P) & 7;
    if (lane < nl) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = wave * 4 + q;  // 0..31 = (dzl', j)
        const int dl = idx >> 3, j = idx & 7;
        const int dz = grp * 4 + dl, zz = z + dz - P;
        const bool ok = dz < KS && (unsigned)zz < (unsigned)p.D && (unsigned)yy < (unsigned)p.H;
        const unsigned short* src =
            ok ? p.xp + (((long)n * 8 + j) * R + (long)zz * p.H + yy) * p.PX + lane * 8 : reinterpret_cast<const unsigned short*>(p.zeros);
        nc_dma_lds16(src, nc_lds_addr((XPs + (((long)dl * 8 + slot) * 8 + j) * p.PL)));
      }
    }
  };

  f32x16 acc[NTT];
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  for (int yy = -P; yy <= P; ++yy) stage_x(yy);
  stage_g(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const bool dzok = grp * 4 + dzl < KS;
  const int tsub = li >> 3, j = li & 7;
  for (int y = 0; y < p.H; ++y) {
    if (y + 1 < p.H) {
      stage_x(y + P + 1);
      stage_g(y + 1, (y + 1) & 1);
    }
    if (dzok) {
      // transposed-read roles (k_wgrad_h): group g = lane / 16 -> channels 16 (g & 1) .. of this wave's 32, voxel half g / 2;
      // inside the group lane 4 q + pp supplies the address of voxel row q, channels 4 pp .. 4 pp + 3
      const int g16 = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
      const int cbsel = 2 * (g16 & 1) + (pp >> 1);
      const char* abase = reinterpret_cast<const char*>(Gs) + (((long)(y & 1) * 8 + mt * 4 + cbsel) * p.PX) * 16 + (pp & 1) * 8 +
                          (8 * (g16 >> 1) + q4) * 16;
      const unsigned short* brow[NTT];
      int bstep[NTT];
#pragma unroll
      for (int t = 0; t < NTT; ++t) {
        const int dy = 4 * t + tsub;
        brow[t] = dy < KS ? XPs + (((long)dzl * 8 + ((y + dy) & 7)) * 8 + j) * p.PL + 8 * h : zrow;
        bstep[t] = dy < KS ? 16 : 0;  // (the dummy 8th dy of the second quad reads the zero row)
      }
#pragma unroll 2
      for (int xg = 0; xg < p.nxg; ++xg) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(abase + xg * 256));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(abase + xg * 256 + 64));
        const i32x4 a = __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
          const i32x4 b = *reinterpret_cast<const i32x4*>(brow[t] + xg * bstep[t]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // partial [workgroup][k][dzl][dy][dx]: rows = channels (e & 3) + 8 (e >> 2) + 4 h, column li = (dy member, dx)
  float* o = p.part + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 64) * 256;
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      o[((long)k * 4 + dzl) * 64 + t * 32 + li] = acc[t][e];
    }
  if constexpr (NTT == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      o[((long)k * 4 + dzl) * 64 + 32 + li] = 0.f;
    }
  }
}

// dw[k][dz][dy][dx] = sum over the planes of the dz's group, in plane order: one workgroup per (k, dzl, group); 256 threads =
// 4 planes x 64 (dy, dx) entries per trip, the four running sums added in a fixed order
__global__ void __launch_bounds__(256) k_c1_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int nplanes, int KS) {
  __shared__ double red[4][64];
  const int k = blockIdx.x >> 2, dzl = blockIdx.x & 3, grp = blockIdx.y;
  const int dz = grp * 4 + dzl;
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  double s = 0.0;
  for (int pl = q; pl < nplanes; pl += 4) s += (double)part[(((long)grp * nplanes + pl) * 64 + k) * 256 + dzl * 64 + e];
  red[q][e] = s;
  __syncthreads();
  if (q == 0 && dz < KS) {
    const int dy = e >> 3, dx = e & 7;
    if (dy < KS && dx < KS) dw[(((long)k * KS + dz) * KS + dy) * KS + dx] = (float)(((red[0][e] + red[1][e]) + red[2][e]) + red[3][e]);
  }
}

struct CwPlan {
  bool ok;
  int PX, PL, nxg, ngrp, lds;
  size_t gp_bytes, xp_bytes, part_bytes;
};
CwPlan cw_plan(int N, int D, int H, int W, int KS) {
  CwPlan c{};
  if ((KS != 3 && KS != 7) || N < 1 || D < 1 || H < 1 || W < 1 || W > 192) return c;
  c.PX = (W + 15) / 16 * 16;
  c.PL = c.PX + 8;
  c.nxg = c.PX / 16;
  c.ngrp = KS > 4 ? 2 : 1;
  c.lds = (c.PL + 2 * 64 * c.PX + 4 * 8 * 8 * c.PL) * 2;
  if (c.lds > 160 * 1024) return c;
  const size_t R = (size_t)D * H;
  c.gp_bytes = 0;  // (the gradient is read as C8: no planar copy)
  c.xp_bytes = ((size_t)N * 8 * R * c.PX * 2 + 255) & ~(size_t)255;
  c.part_bytes = ((size_t)c.ngrp * N * D * 64 * 256 * 4 + 255) & ~(size_t)255;
  c.ok = true;
  return c;
}

}  // namespace

bool c1_wgrad_h_supported(int N, int D, int H, int W, int KS) { return cw_plan(N, D, H, W, KS).ok; }
size_t c1_wgrad_h_ws_bytes(int N, int D, int H, int W, int KS) {
  const CwPlan c = cw_plan(N, D, H, W, KS);
  return c.ok ? c.gp_bytes + c.xp_bytes + c.part_bytes : 0;
}

// x fp32 [N][1][D][H][W], dyh C8 bf16 [N][8][D*H*W][8] (64 channels), dw fp32 [64][1][KS][KS][KS]
int conv_c1_wgrad_h(const float* x, const void* dyh, float* dw, int N, int D, int H, int W, int KS, void* ws, size_t wsb,
                    hipStream_t s) {
  const CwPlan c = cw_plan(N, D, H, W, KS);
  if (!c.ok) { set_error("conv_c1_wgrad_h: shape not covered"); return NC_ERR_SHAPE; }
  if (!ws || wsb < c.gp_bytes + c.xp_bytes + c.part_bytes) { set_error("conv_c1_wgrad_h: workspace too small"); return NC_ERR_WS; }
  const float* zeros = nc_zero_page();
  if (!zeros) { set_error("conv_c1_wgrad_h: no zero page"); return NC_ERR_HIP; }
  unsigned short* xp = (unsigned short*)((char*)ws + c.gp_bytes);
  float* part = (float*)((char*)ws + c.gp_bytes + c.xp_bytes);
  const long R = (long)D * H;
  {
    const long total = (long)N * R * c.PX;
    hipLaunchKernelGGL(k_build_xp, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, x, xp, W, c.PX, KS, R, total);
  }
  if (int e = check_launch("c1_wgrad_h prep")) return e;
  CwParams p{};
  p.gc8 = (const uint4*)dyh; p.xp = xp; p.zeros = zeros; p.part = part;
  p.N = N; p.D = D; p.H = H; p.W = W; p.PX = c.PX; p.PL = c.PL; p.nxg = c.nxg;
  const dim3 grid((unsigned)(N * D), (unsigned)c.ngrp);
  auto launch = [&](auto kern) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      set_error("conv_c1_wgrad_h: cannot raise dynamic LDS limit");
      return NC_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), c.lds, s, p);
    return check_launch("c1_wgrad_h");
  };
  if (int e = KS == 7 ? launch(k_c1_wgrad_p<7>) : launch(k_c1_wgrad_p<3>)) return e;
  hipLaunchKernelGGL(k_c1_wgrad_reduce, dim3(64 * 4, c.ngrp), dim3(256), 0, s, part, dw, N * D, KS);
  return check_launch("c1_wgrad_h reduce");
}

}  // namespace nc
