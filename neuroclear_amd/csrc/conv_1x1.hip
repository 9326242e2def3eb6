// Weight gradient of the 1x1(x1) convolutions (stride 1, no padding, <= 64 channels either side) on gfx950.
// Replaces autograd's backward-weights of the pointwise layers: G_B's tail 64->32->16->1 (models/networks.py:903-911)
// and Unet_deconv's one_by_one / one_by_one_2 (:507-508).
//
// dW[k][c] = sum_{n,v} dY[n][k][v] * X[n][c][v] is a [K x C] outer-product sum over every voxel: 0.5 GB of reads for
// 5 GFLOP at 108^3, i.e. HBM-bound.  The voxel axis is flat (no rows, no halo), so
//   * a workgroup streams chunks of 128 voxels: every channel's 128 floats go global -> LDS as 16-byte LDS-DMA lanes
//     (1 KiB per instruction, per-lane source address; channel rows >= K / >= C, the 8 pad floats of every LDS row
//     and voxels past the end of the volume come from a zero page), double buffered, one barrier per chunk;
//   * v_mfma_f32_16x16x4_f32 (exact fp32) with M = co, N = ci, K = voxels: MFMA m of a 16-voxel iteration reduces
//     over voxels q + 4*kq + m, so both operands of four MFMAs are one aligned ds_read_b128 each (LDS pitch 136 = 8
//     mod 16: conflict-free); each of the 8 waves owns up to two 16 x 16 blocks of dW;
//   * per-workgroup partial dW + a fixed-order reduce kernel (deterministic, no atomics).
#include "common.hpp"

namespace nc {
NC_ZERO_PAGE()


typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kVC = 128;      // voxels per chunk
static constexpr int kPitch = 136;   // LDS floats per channel row (128 + 8, = 8 mod 16)
static constexpr int kMaxP1 = 9;     // DMA pieces per wave per chunk

struct W1Params {
  const float* x;
  const float* dy;
  float* slab;         // [nwg][Kp][Cp]
  const float* zeros;  // >= 16 B of zeros in global memory
  int C, K, N;
  long S;              // voxels per (n, channel)
  int Kp, Cp;          // channels rounded up to 16
  int npd, npx;        // 256-float pieces of the dY / X region of one buffer
  long nchunks;        // N * ceil(S / 128)
  long cps;            // chunks per sample
};

__global__ __launch_bounds__(512) void k_wgrad_1x1(W1Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int MB = p.Kp / 16, NB = p.Cp / 16, nblk = MB * NB;
  const int bufsz = (p.npd + p.npx) * 256;
  const long c0 = p.nchunks * blockIdx.x / gridDim.x, c1 = p.nchunks * (blockIdx.x + 1) / gridDim.x;

  // per-lane source of this wave's pieces, independent of the chunk: channel * S + column, or -1 for the zero page;
  // pieces [0, npd) hold dY rows, [npd, npd + npx) hold X rows
  int goff[kMaxP1], gcol[kMaxP1];
#pragma unroll
  for (int i = 0; i < kMaxP1; ++i) {
    const int j = wave + 8 * i;
    const bool isx = j >= p.npd;
    const int f = ((isx ? j - p.npd : j) * 64 + lane) * 4;
    const int row = f / kPitch, col = f - row * kPitch;
    const bool ok = col < kVC && row < (isx ? p.C : p.K);
    goff[i] = ok ? (int)(row * p.S + col) : -1;
    gcol[i] = col;
  }
  auto issue = [&](long chunk, float* buf) {
    const long n = chunk / p.cps;
    const long v0 = (chunk - n * p.cps) * kVC;
    const float* db = p.dy + n * p.K * p.S + v0;
    const float* xb = p.x + n * p.C * p.S + v0;
    const int vleft = (int)min((long)kVC, p.S - v0);
#pragma unroll
    for (int i = 0; i < kMaxP1; ++i) {
      const int j = wave + 8 * i;
      if (j < p.npd + p.npx) {
        const float* base = j >= p.npd ? xb : db;
        const float* src = (goff[i] >= 0 && gcol[i] < vleft) ? base + goff[i] : p.zeros;
        nc_dma_lds16(src, nc_lds_addr((buf + j * 256)));
      }
    }
  };

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  int aoff[2], boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int b = wave + 8 * t;
    const int mb = b < nblk ? b / NB : 0, nb = b < nblk ? b % NB : 0;
    aoff[t] = (mb * 16 + l15) * kPitch + 4 * kq;
    boff[t] = p.npd * 256 + (nb * 16 + l15) * kPitch + 4 * kq;
  }

  if (c0 < c1) issue(c0, lds);
  for (long c = c0; c < c1; ++c) {
    const int par = (int)((c - c0) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the chunk has landed for every wave, and the previous chunk's MFMAs are done
    if (c + 1 < c1) issue(c + 1, lds + (par ^ 1) * bufsz);
    const float* buf = lds + par * bufsz;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (wave + 8 * t < nblk) {
#pragma unroll
        for (int q = 0; q < kVC; q += 16) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(buf + aoff[t] + q, 16));
          const f32x4 b = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(buf + boff[t] + q, 16));
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[m], acc[t], 0, 0, 0);
        }
      }
    }
  }
  // C/D layout of the 16x16 MFMA: col (ci) = lane & 15, row (co) = 4 * (lane >> 4) + r
  float* sl = p.slab + (long)blockIdx.x * p.Kp * p.Cp;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int b = wave + 8 * t;
    if (b < nblk) {
      const int mb = b / NB, nb = b % NB;
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[(mb * 16 + 4 * kq + r) * p.Cp + nb * 16 + l15] = acc[t][r];
    }
  }
}

// one wave per output element: lane l adds the slabs l, l + 64, ... in order, then a fixed-order butterfly over the
// 64 lanes (deterministic; a single thread walking 256 strided slabs was latency bound: 60 us)
__global__ void k_wgrad_1x1_reduce(const float* __restrict__ slab, float* __restrict__ dw, int nwg, int K, int C,
                                   int Kp, int Cp) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= K * C) return;
  const int lane = threadIdx.x & 63;
  const int k = i / C, c = i - k * C;
  float s = 0.f;
  for (int w = lane; w < nwg; w += 64) s += slab[((long)w * Kp + k) * Cp + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) dw[i] = s;
}

bool wgrad_1x1_supported(const ConvDims& d) {
  if (d.kd != 1 || d.kh != 1 || d.kw != 1 || d.sh != 1 || d.sw != 1 || d.ph != 0 || d.pw != 0) return false;
  if (d.K > 64 || d.C > 64) return false;
  const long S = (long)d.D * d.H * d.W;
  if (S % 4 || S < 16384 || S * 64 >= (1L << 31)) return false;  // small planes stay on the gather-GEMM path
  return true;
}

static int wg1_nwg(const ConvDims& d) {
  const long S = (long)d.D * d.H * d.W;
  const long chunks = (long)d.N * ((S + kVC - 1) / kVC);
  return (int)(chunks < 256 ? chunks : 256);
}

size_t wgrad_1x1_ws_bytes(const ConvDims& d) {
  if (!wgrad_1x1_supported(d)) return 0;
  const int Kp = (d.K + 15) & ~15, Cp = (d.C + 15) & ~15;
  return (size_t)wg1_nwg(d) * Kp * Cp * sizeof(float) + 256;
}

int conv_wgrad_1x1(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  const size_t need = wgrad_1x1_ws_bytes(d);
  if (!need) {
    set_error("wgrad_1x1: unsupported shape");
    return NC_ERR_SHAPE;
  }
  if (!ws || wsb < need) {
    set_error("wgrad_1x1: workspace too small (%zu < %zu)", wsb, need);
    return NC_ERR_WS;
  }
  W1Params p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws;
  p.zeros = nc_zero_page();
  if (!p.zeros) { set_error("wgrad_1x1: no zero page"); return NC_ERR_HIP; }
  p.C = d.C; p.K = d.K; p.N = d.N;
  p.S = (long)d.D * d.H * d.W;
  p.Kp = (d.K + 15) & ~15; p.Cp = (d.C + 15) & ~15;
  p.npd = (p.Kp * kPitch + 255) / 256; p.npx = (p.Cp * kPitch + 255) / 256;
  p.cps = (p.S + kVC - 1) / kVC;
  p.nchunks = (long)d.N * p.cps;
  const int nwg = wg1_nwg(d);
  const int lds_bytes = 2 * (p.npd + p.npx) * 256 * (int)sizeof(float);
  if (int e = raise_dyn_lds(k_wgrad_1x1, 160 * 1024, "wgrad_1x1")) return e;
  hipLaunchKernelGGL(k_wgrad_1x1, dim3(nwg), dim3(512), lds_bytes, s, p);
  if (int e = check_launch("wgrad_1x1")) return e;
  hipLaunchKernelGGL(k_wgrad_1x1_reduce, dim3((d.K * d.C + 3) / 4), dim3(256), 0, s, (const float*)ws, dw, nwg, d.K,
                     d.C, p.Kp, p.Cp);
  return check_launch("wgrad_1x1_reduce");
}

// ---------------------------------------------------------------------------------------------------------------
// Forward and data gradient of the same pointwise layers: out[m][v] = sum_r A[m][r] * in[r][v] (+ bias[m]) with
// A = W (forward: m = k, r = c) or W^T (data gradient: m = c, r = k), M, R <= 64, on the flat voxel axis.
// HBM-bound (one read of `in`, one write of `out`): chunks of 128 voxels of every input channel arrive by LDS-DMA
// (double buffered), v_mfma_f32_16x16x4_f32 with the weights held in registers as the A operand, and the output tile
// goes through LDS so that every store instruction writes 1 KiB of one channel row.
static constexpr int kPitchF = 144;  // input rows: 128 + 16, = 16 (mod 32): the two k rows of a half-wave read disjoint banks
static constexpr int kPitchO = 132;  // output staging rows
struct F1Params {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  const float* zeros;
  int M, R, N;
  int sm, sr;   // A[m][r] = w[m * sm + r * sr]
  long S;
  int Rp;       // R rounded up to 4
  int npi;      // 256-float pieces of one input buffer
  long nchunks, cps;
};

__global__ __launch_bounds__(512) void k_flat_1x1(F1Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int MB = (p.M + 15) / 16;
  const int bufsz = p.npi * 256;
  float* ost = lds + 2 * bufsz;  // [MB * 16][kPitchO]
  const long c0 = p.nchunks * blockIdx.x / gridDim.x, c1 = p.nchunks * (blockIdx.x + 1) / gridDim.x;

  int goff[5], gcol[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int f = ((wave + 8 * i) * 64 + lane) * 4;
    const int row = f / kPitchF, col = f - row * kPitchF;
    goff[i] = (col < kVC && row < p.R) ? (int)(row * p.S + col) : -1;
    gcol[i] = col;
  }
  auto issue = [&](long chunk, float* buf) {
    const long n = chunk / p.cps;
    const long v0 = (chunk - n * p.cps) * kVC;
    const float* xb = p.in + n * p.R * p.S + v0;
    const int vleft = (int)min((long)kVC, p.S - v0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int j = wave + 8 * i;
      if (j < p.npi) {
        const float* src = (goff[i] >= 0 && gcol[i] < vleft) ? xb + goff[i] : p.zeros;
        nc_dma_lds16(src, nc_lds_addr((buf + j * 256)));
      }
    }
  };
  // this wave: output row block mb = wave % MB' and every (8 / waves-per-mb)-th voxel block; weights of the row block
  // as the A operand of all k-steps
  const int wpm = MB >= 4 ? 2 : MB == 2 ? 4 : 8;  // waves per row block (MB = 3 is served as 4)
  const int mb = wave / wpm;
  const bool mb_on = mb < MB;
  float aw[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int m = mb * 16 + l15, r = 4 * ks + kq;
    aw[ks] = (mb_on && m < p.M && r < p.R) ? p.w[(long)m * p.sm + (long)r * p.sr] : 0.f;
  }
  const int nks = p.Rp / 4;
  const int b_off = kq * kPitchF + l15;

  if (c0 < c1) issue(c0, lds);
  for (long c = c0; c < c1; ++c) {
    const int par = (int)((c - c0) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunk landed; the previous chunk's output tile has been written out
    if (c + 1 < c1) issue(c + 1, lds + (par ^ 1) * bufsz);
    const float* buf = lds + par * bufsz + b_off;
    if (mb_on) {
      for (int nb = wave % wpm; nb < 8; nb += wpm) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          if (ks < nks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks], buf[4 * ks * kPitchF + nb * 16], acc, 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ost[(mb * 16 + 4 * kq + rr) * kPitchO + nb * 16 + l15] = acc[rr];
      }
    }
    __syncthreads();
    // write the tile: one wave-instruction = two channel rows x 128 voxels (float4 per lane)
    const long n = c / p.cps;
    const long v0 = (c - n * p.cps) * kVC;
    const int vleft = (int)min((long)kVC, p.S - v0);
    for (int m = 2 * wave + (lane >> 5); m < p.M; m += 16) {
      const int col = (lane & 31) * 4;
      if (col < vleft) {
        f32x4 v = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(ost + m * kPitchO + col, 16));
        if (p.bias) {
          const float b = p.bias[m];
          v += f32x4{b, b, b, b};
        }
        *reinterpret_cast<f32x4*>(p.out + (n * p.M + m) * p.S + v0 + col) = v;
      }
    }
  }
}

static bool flat_1x1_shape(const ConvDims& d) {
  if (d.kd != 1 || d.kh != 1 || d.kw != 1 || d.sh != 1 || d.sw != 1 || d.ph != 0 || d.pw != 0) return false;
  if (d.K > 64 || d.C > 64) return false;
  const long S = (long)d.D * d.H * d.W;
  return S % 4 == 0 && S >= 16384 && S * 64 < (1L << 31);
}
bool flat_1x1_supported(const ConvDims& d) { return flat_1x1_shape(d); }

static int launch_flat(const float* in, const float* w, const float* bias, float* out, int M, int R, int sm, int sr,
                       const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!ws || wsb < 256) {
    set_error("flat_1x1: workspace too small");
    return NC_ERR_WS;
  }
  F1Params p{};
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.zeros = nc_zero_page();
  if (!p.zeros) { set_error("flat_1x1: no zero page"); return NC_ERR_HIP; }
  p.M = M; p.R = R; p.N = d.N; p.sm = sm; p.sr = sr;
  p.S = (long)d.D * d.H * d.W;
  p.Rp = (R + 3) & ~3;
  p.npi = (p.Rp * kPitchF + 255) / 256;  // the k rows R .. Rp-1 of the last k-step are zero-page rows, not stale LDS
  p.cps = (p.S + kVC - 1) / kVC;
  p.nchunks = (long)d.N * p.cps;
  const int nwg = (int)(p.nchunks < 512 ? p.nchunks : 512);
  const int MB = (M + 15) / 16;
  const int lds_bytes = (2 * p.npi * 256 + MB * 16 * kPitchO) * (int)sizeof(float);
  if (int e = raise_dyn_lds(k_flat_1x1, 160 * 1024, "flat_1x1")) return e;
  hipLaunchKernelGGL(k_flat_1x1, dim3(nwg), dim3(512), lds_bytes, s, p);
  return check_launch("flat_1x1");
}

int conv_fwd_1x1(const float* x, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                 hipStream_t s) {
  return launch_flat(x, w, b, y, d.K, d.C, d.C, 1, d, ws, wsb, s);
}
int conv_dgrad_1x1(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                   hipStream_t s) {
  return launch_flat(dy, w, nullptr, dx, d.C, d.K, 1, d.C, d, ws, wsb, s);
}

}  // namespace nc
