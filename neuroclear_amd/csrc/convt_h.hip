// ConvTranspose3d(kernel 2, stride 2) on the 16-bit matrix cores, C8 in -> C8 out (reference models/networks.py:500,503:
// t_conv2 256 -> 128 at the quarter level, t_conv1 128 -> 64 at the half level; backward = autograd of the same call).
// A k2 / s2 transposed convolution has no overlapping taps: every fine voxel (2z+a, 2y+b, 2x+c) receives exactly one
// product per input channel, so each of the 8 sub-positions t = (a, b, c) is a plain [K x C] x [C x voxels] GEMM.
//   forward : out[k][fine(v, t)] = bias[k] + sum_ci w[ci][k][t] x[ci][v]        M = k,  N = coarse voxels, K-dim = ci
//   dgrad   : dx[ci][v]          = sum_(k, t) w[ci][k][t] dy[k][fine(v, t)]     M = ci, N = coarse voxels, K-dim = (t, k)
//   wgrad   : dw[ci][k][t]       = sum_(n, v) x[ci][v] dy[k][fine(v, t)]        M = ci, N = k,             K-dim = voxels
// In the C8 layout a voxel's 8 channels are one 16-byte unit, which IS the B fragment of v_mfma_f32_32x32x16 when the
// K-dim is channels (lane (r, h) = voxel r, channels 8h..8h+7): forward and dgrad load their B operand straight from
// global memory, one unit per lane, no LDS; the packed weights (<= 1 MiB) come from L1 / L2.  The weight gradient
// reduces over voxels, i.e. needs both operands voxel-major: its fragments are gathered with 2-byte loads (FLOP are
// ~0.3 % of the step; the gather keeps the matrix pipe ~half busy, which is plenty).  All three are HBM-bound or small.
#include <cstdlib>

#include "common.hpp"

namespace nc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ f32x16 mfma16(const i32x4& a, const i32x4& b, const f32x16& c) {
  if constexpr (DT == NC_DT_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ unsigned short cvt16(float f) {
  if constexpr (DT == NC_DT_F16) {
    const _Float16 v = (_Float16)f;
    return __builtin_bit_cast(unsigned short, v);
  } else {
    const __bf16 v = (__bf16)f;
    return __builtin_bit_cast(unsigned short, v);
  }
}

// packed forward weights:  wp[t][kt = k/32][chunk = ci/16][h][r = k%32][8]  element j = input channel chunk*16 + 8h + j
// packed dgrad weights:    wp[t][ct = ci/32][chunk = k/16][h][r = ci%32][8] element j = output channel chunk*16 + 8h + j
template <int DT>
__global__ void __launch_bounds__(256) k_pack_wT(const float* __restrict__ w, unsigned short* __restrict__ wp, int C, int K, int dgrad,
                                                 long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int r = (int)(q & 31); q >>= 5;
  const int h = (int)(q & 1); q >>= 1;
  const int RT = dgrad ? C / 32 : K / 32, NCH = dgrad ? K / 16 : C / 16;
  const int chunk = (int)(q % NCH); q /= NCH;
  const int rt = (int)(q % RT);
  const int t = (int)(q / RT);
  const int row = rt * 32 + r, red = chunk * 16 + 8 * h + j;
  const int ci = dgrad ? row : red, k = dgrad ? red : row;
  wp[i] = cvt16<DT>(w[((long)ci * K + k) * 8 + t]);
}

struct TParams {
  const uint4* x;      // C8 operand with channels on the K-dim (fwd: x dense; dgrad: dy view)
  const uint4* wp;
  const float* bias;
  uint2* out;          // C8 output (fwd: view; dgrad: dense)
  int N, C, K;
  int Dc, Hc, Wc;      // coarse spatial size
  int xctot8, xc08;    // view of the K-dim operand
  int octot8, oc08;    // view of the output
  long Sc;
};

// forward: block = 4 waves, a wave = tiles of 32 coarse voxels; blockIdx.y = 32-channel output tile.  The packed weights of
// the output tile (8 sub-positions x C / 16 chunks x 1 KiB) are copied into LDS once per workgroup and every wave walks several
// voxel tiles with them: read from global per (sub-position, chunk) they were 64 KB per 32 voxels -- 6.5 GB of L2 traffic per
// launch at 128 -> 64 / 4 x 74^3, which was what the kernel spent its time on.
template <int DT>
__global__ void __launch_bounds__(256) k_convT_fwd_h(const TParams p) {
  extern __shared__ __attribute__((aligned(16))) uint4 wl[];  // [8 t][NCH][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int kt = blockIdx.y, NCH = p.C / 16, KT = p.K / 32;
  for (int i = threadIdx.x; i < 8 * NCH * 64; i += 256) {
    const int t = i / (NCH * 64), rem = i - t * NCH * 64;
    wl[i] = p.wp[((long)t * KT + kt) * NCH * 64 + rem];
  }
  __syncthreads();
  const long tiles_per_n = (p.Sc + 31) / 32, ntiles = tiles_per_n * p.N;
  const int Hf = 2 * p.Hc, Wf = 2 * p.Wc;
  const long Sf = 8 * p.Sc;
  float bv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bv[e] = p.bias ? p.bias[kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] : 0.f;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
  const int n = (int)(tile / tiles_per_n);
  const long v0 = (tile - (long)n * tiles_per_n) * 32;
  const long v = v0 + r;
  const bool ok = v < p.Sc;
  const long vc = ok ? v : p.Sc - 1;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const uint4* xb = p.x + ((long)n * p.xctot8 + p.xc08) * p.Sc + vc;
  // all the tile's input units first (8 chunks in flight: one HBM round trip per 8, not per chunk), then the products
  for (int c0 = 0; c0 < NCH; c0 += 8) {
    i32x4 b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c0 + c < NCH) b[c] = __builtin_bit_cast(i32x4, xb[(long)((c0 + c) * 2 + h) * p.Sc]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c0 + c < NCH) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const i32x4 a = __builtin_bit_cast(i32x4, wl[(t * NCH + c0 + c) * 64 + lane]);
          acc[t] = mfma16<DT>(a, b[c], acc[t]);
        }
      }
  }
  if (!ok) continue;
  const int xw = (int)(v % p.Wc), yh = (int)((v / p.Wc) % p.Hc), zd = (int)(v / ((long)p.Wc * p.Hc));
  // A lane holds channels 4h .. 4h + 3 of the four 8-channel blocks, i.e. one 8-byte half of a unit, for both x sub-positions
  // c = 0 / 1 of a (z, y) sub-position.  The two lanes (r, 0) / (r, 1) swap halves so that lane h writes the WHOLE unit of
  // c = h: 64 lanes x 16 bytes = 1 KiB of contiguous output per store (consecutive r are consecutive fine x pairs).
  uint4* ob = reinterpret_cast<uint4*>(p.out) + ((long)n * p.octot8 + p.oc08 + kt * 4) * Sf;
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) {
    const long vf = ((long)(2 * zd + (ab >> 1)) * Hf + (2 * yh + (ab & 1))) * Wf + 2 * xw + h;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      uint2 o[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x16& a = acc[2 * ab + c];
        o[c].x = cvt16<DT>(a[4 * g4] + bv[4 * g4]) | ((unsigned)cvt16<DT>(a[4 * g4 + 1] + bv[4 * g4 + 1]) << 16);
        o[c].y = cvt16<DT>(a[4 * g4 + 2] + bv[4 * g4 + 2]) | ((unsigned)cvt16<DT>(a[4 * g4 + 3] + bv[4 * g4 + 3]) << 16);
      }
      const uint2 send = h ? o[0] : o[1];  // what the partner lane needs: its c's other half
      uint2 recv;
      recv.x = (unsigned)__shfl_xor((int)send.x, 32);
      recv.y = (unsigned)__shfl_xor((int)send.y, 32);
      const uint4 u = h ? make_uint4(recv.x, recv.y, o[1].x, o[1].y) : make_uint4(o[0].x, o[0].y, recv.x, recv.y);
      ob[(long)g4 * Sf + vf] = u;
    }
  }
  }  // tiles
}

// dgrad: wave = 32 coarse voxels x CTW 32-channel tiles of ci; K-dim = (t, k)
template <int DT, int CTW>
__global__ void __launch_bounds__(256) k_convT_dgrad_h(const TParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int ct0 = blockIdx.y * CTW, NCH = p.K / 16, CT = p.C / 32;
  const long tile = (long)blockIdx.x * 4 + wave;
  const long tiles_per_n = (p.Sc + 31) / 32;
  if (tile >= tiles_per_n * p.N) return;
  const int n = (int)(tile / tiles_per_n);
  const long v = (tile - (long)n * tiles_per_n) * 32 + r;
  const bool ok = v < p.Sc;
  const long vc = ok ? v : p.Sc - 1;
  const int xw = (int)(vc % p.Wc), yh = (int)((vc / p.Wc) % p.Hc), zd = (int)(vc / ((long)p.Wc * p.Hc));
  const int Hf = 2 * p.Hc, Wf = 2 * p.Wc;
  const long Sf = 8 * p.Sc;
  f32x16 acc[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  const uint4* yb = p.x + ((long)n * p.xctot8 + p.xc08) * Sf;
#pragma unroll 1
  for (int t = 0; t < 8; ++t) {
    const long vf = ((long)(2 * zd + (t >> 2)) * Hf + (2 * yh + ((t >> 1) & 1))) * Wf + 2 * xw + (t & 1);
    const uint4* wb = p.wp + (((long)t * CT + ct0) * NCH * 2 + h) * 32 + r;
    for (int chunk = 0; chunk < NCH; ++chunk) {
      const i32x4 b = __builtin_bit_cast(i32x4, yb[(long)(chunk * 2 + h) * Sf + vf]);
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const i32x4 a = __builtin_bit_cast(i32x4, wb[((long)c * NCH + chunk) * 64]);
        acc[c] = mfma16<DT>(a, b, acc[c]);
      }
    }
  }
  if (!ok) return;
  uint2* ob = p.out + (((long)n * p.octot8 + p.oc08 + ct0 * 4) * p.Sc + v) * 2 + h;
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      uint2 o;
      o.x = cvt16<DT>(acc[c][4 * g4]) | ((unsigned)cvt16<DT>(acc[c][4 * g4 + 1]) << 16);
      o.y = cvt16<DT>(acc[c][4 * g4 + 2]) | ((unsigned)cvt16<DT>(acc[c][4 * g4 + 3]) << 16);
      ob[(long)(c * 4 + g4) * p.Sc * 2] = o;
    }
}

struct TWParams {
  const unsigned short* x;   // dense C8 [N][C/8][Sc][8]
  const unsigned short* dy;  // view of a C8 buffer at the fine size: channels [c0, c0 + K) of ctot
  float* part;               // [splits][C][K][8]
  int N, C, K, Dc, Hc, Wc, dctot8, dc08, splits;
  long Sc;
};

// wgrad: one wave = a (ct, kt) tile of 32 x 32 for all 8 sub-positions, over a contiguous share of the 16-voxel steps.
// Per step the wave loads the 16 voxels x 32 channels of x and, for each sub-position, of dy as whole 16-byte units
// (lane = (C8 block, voxel): coalesced), parks them in a private LDS image [block][voxel][8 channels] and takes the MFMA
// fragments -- 8 consecutive VOXELS of one channel per lane -- with the transposing LDS read ds_read_b64_tr_b16 (lane
// 4q + p of a 16-lane group supplies the address of voxel row q, channels 4p..4p+3; lane i receives channel i of the 4 rows).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* ltr_t;

template <int DT>
__global__ void __launch_bounds__(64) k_convT_wgrad_h(const TWParams p) {
  constexpr int BS = 20;  // units between the 4 blocks of an image (16 voxels + 4: spreads the banks)
  __shared__ __attribute__((aligned(16))) uint4 img[9 * 4 * BS];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int KT = p.K / 32;
  const int kt = blockIdx.x % KT, ct = blockIdx.x / KT, split = blockIdx.y;
  const long steps_per_n = (p.Sc + 15) / 16, steps = steps_per_n * p.N;
  const long s_lo = steps * split / p.splits, s_hi = steps * (split + 1) / p.splits;
  const int Hf = 2 * p.Hc, Wf = 2 * p.Wc;
  const long Sf = 8 * p.Sc;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  // loader role: C8 block cb of the tile, voxel j of the step
  const int cb = lane >> 4, j = lane & 15;
  const uint4* xg = reinterpret_cast<const uint4*>(p.x);
  const uint4* yg = reinterpret_cast<const uint4*>(p.dy);
  // reader role (transposed read): group g -> channels 16 (g & 1).., voxel half g >> 1
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int cbsel = 2 * (g & 1) + (pp >> 1);
  const unsigned rd = (unsigned)((cbsel * BS + 8 * (g >> 1) + q) * 16 + (pp & 1) * 8);  // + 64 for voxels 4..7 of the half
  unsigned char* lds = reinterpret_cast<unsigned char*>(img);
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  uint4 nx = zero4, ny[8];
  auto fetch = [&](long st) {
    const int n = (int)(st / steps_per_n);
    const long v = (st - (long)n * steps_per_n) * 16 + j;
    const bool ok = v < p.Sc;
    const long vc = ok ? v : p.Sc - 1;
    const int xw = (int)(vc % p.Wc), yh = (int)((vc / p.Wc) % p.Hc), zd = (int)(vc / ((long)p.Wc * p.Hc));
    const long vf = ((long)(2 * zd) * Hf + 2 * yh) * Wf + 2 * xw;
    nx = ok ? xg[((long)n * (p.C >> 3) + ct * 4 + cb) * p.Sc + vc] : zero4;
    const uint4* yb = yg + ((long)n * p.dctot8 + p.dc08 + kt * 4 + cb) * Sf + vf;
#pragma unroll
    for (int t = 0; t < 8; ++t) ny[t] = ok ? yb[((long)(t >> 2) * Hf + ((t >> 1) & 1)) * Wf + (t & 1)] : zero4;
  };
  if (s_lo < s_hi) fetch(s_lo);
  for (long st = s_lo; st < s_hi; ++st) {
    img[cb * BS + j] = nx;
#pragma unroll
    for (int t = 0; t < 8; ++t) img[(1 + t) * 4 * BS + cb * BS + j] = ny[t];
    if (st + 1 < s_hi) fetch(st + 1);  // next step's units fly under this step's MFMAs
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the image is written (one wave: no barrier needed)
    i32x4 a;
    {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + rd));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + rd + 64));
      a = __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const unsigned base = (unsigned)((1 + t) * 4 * BS * 16) + rd;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + base));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + base + 64));
      const i32x4 b = __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      acc[t] = mfma16<DT>(a, b, acc[t]);
    }
  }
  // rows = ci, lanes = k: part[split][ci][k][t]
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      p.part[(((long)split * p.C + row) * p.K + kt * 32 + r) * 8 + t] = acc[t][e];
    }
}

// The same with the dY fragments SHARED: a workgroup of C / 32 waves owns one 32-channel k tile, wave w the input-channel
// tile w.  The one-wave kernel above re-reads dY once per input-channel tile (4 x at 128 -> 64: 6.6 GB per launch at 4 x 148^3
// for 1.66 GB of dY); here every 16-voxel step's dY units are fetched once per workgroup (wave w brings the sub-positions
// w, w + C/32, ...) into one of two shared LDS images, the x units stay wave-private.  One barrier per step: a wave can only
// be one step ahead of the slowest one, so the image of step s + 2 never overwrites fragments still being read.
template <int DT, int VS>  // VS 16-voxel groups per step (one barrier per step)
__global__ void __launch_bounds__(512) k_convT_wgrad_h2(const TWParams p) {
  constexpr int BS = VS * 16 + 4;
  extern __shared__ __attribute__((aligned(16))) uint4 dimg[];  // [NW][4 * BS] x, then [2][8][4 * BS] dY
  const int NW = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kt = blockIdx.x, ct = wave, split = blockIdx.y;
  const int TPW = 8 / NW;  // dY sub-positions fetched per wave
  const long G = VS * 16;
  const long steps_per_n = (p.Sc + G - 1) / G, steps = steps_per_n * p.N;
  const long s_lo = steps * split / p.splits, s_hi = steps * (split + 1) / p.splits;
  const int Hf = 2 * p.Hc, Wf = 2 * p.Wc;
  const long Sf = 8 * p.Sc;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const int cb = lane >> 4, j = lane & 15;
  const uint4* xg = reinterpret_cast<const uint4*>(p.x);
  const uint4* yg = reinterpret_cast<const uint4*>(p.dy);
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int cbsel = 2 * (g & 1) + (pp >> 1);
  const unsigned rd = (unsigned)((cbsel * BS + 8 * (g >> 1) + q) * 16 + (pp & 1) * 8);
  uint4* const ximg = dimg + wave * 4 * BS;
  uint4* const yimg = dimg + NW * 4 * BS;
  const unsigned char* xl = reinterpret_cast<const unsigned char*>(ximg);
  const unsigned char* yl = reinterpret_cast<const unsigned char*>(yimg);
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  uint4 nx[VS], ny[VS][8];
  auto fetch = [&](long st) {
    const int n = (int)(st / steps_per_n);
#pragma unroll
    for (int ss = 0; ss < VS; ++ss) {
      const long v = (st - (long)n * steps_per_n) * G + ss * 16 + j;
      const bool ok = v < p.Sc;
      const long vc = ok ? v : p.Sc - 1;
      const int xw = (int)(vc % p.Wc), yh = (int)((vc / p.Wc) % p.Hc), zd = (int)(vc / ((long)p.Wc * p.Hc));
      const long vf = ((long)(2 * zd) * Hf + 2 * yh) * Wf + 2 * xw;
      nx[ss] = ok ? xg[((long)n * (p.C >> 3) + ct * 4 + cb) * p.Sc + vc] : zero4;
      const uint4* yb = yg + ((long)n * p.dctot8 + p.dc08 + kt * 4 + cb) * Sf + vf;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < TPW) {
          const int t = wave + i * NW;
          ny[ss][i] = ok ? yb[((long)(t >> 2) * Hf + ((t >> 1) & 1)) * Wf + (t & 1)] : zero4;
        }
    }
  };
  if (s_lo < s_hi) fetch(s_lo);
  for (long st = s_lo; st < s_hi; ++st) {
    const int buf = (int)((st - s_lo) & 1);
#pragma unroll
    for (int ss = 0; ss < VS; ++ss) {
      ximg[cb * BS + ss * 16 + j] = nx[ss];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < TPW) yimg[(buf * 8 + wave + i * NW) * 4 * BS + cb * BS + ss * 16 + j] = ny[ss][i];
    }
    if (st + 1 < s_hi) fetch(st + 1);
    __syncthreads();
#pragma unroll
    for (int ss = 0; ss < VS; ++ss) {
      i32x4 a;
      {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(xl + rd + ss * 256));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(xl + rd + ss * 256 + 64));
        a = __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const unsigned base = (unsigned)((buf * 8 + t) * 4 * BS * 16) + rd + ss * 256;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(yl + base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(yl + base + 64));
        const i32x4 b = __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        acc[t] = mfma16<DT>(a, b, acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      p.part[(((long)split * p.C + row) * p.K + kt * 32 + r) * 8 + t] = acc[t][e];
    }
}

__global__ void __launch_bounds__(256) k_convT_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, long n, int splits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = 0; s < splits; ++s) a += part[(long)s * n + i];
  dw[i] = a;
}

// per-channel sum of a C8 view over samples and voxels (the transposed convolution's bias gradient)
template <int DT>
__global__ void __launch_bounds__(256) k_chan_sum_c8(const uint4* __restrict__ x, int ctot8, int c08, int CB, long S, int splits,
                                                     double* __restrict__ part) {
  __shared__ double lds[4 * 8];
  const int ncb = blockIdx.y, n = ncb / CB, cb = ncb - n * CB, split = blockIdx.x;
  const long lo = S * split / splits, hi = S * (split + 1) / splits;
  const uint4* xs = x + ((long)n * ctot8 + c08 + cb) * S;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long v = lo + threadIdx.x; v < hi; v += 256) {
    const uint4 u = xs[v];
    const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned short hv = (unsigned short)((uu[j >> 1] >> ((j & 1) * 16)) & 0xffff);
      s[j] += DT == NC_DT_F16 ? (float)__builtin_bit_cast(_Float16, hv) : __builtin_bit_cast(float, (unsigned)hv << 16);
    }
  }
  double d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double a = s[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    d[j] = a;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[wave * 8 + j] = d[j];
  __syncthreads();
  if (threadIdx.x < 8)
    part[((long)ncb * splits + split) * 8 + threadIdx.x] = lds[threadIdx.x] + lds[8 + threadIdx.x] + lds[16 + threadIdx.x] + lds[24 + threadIdx.x];
}

__global__ void k_chan_sum_final(const double* __restrict__ part, int N, int C, int splits, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int CB = C >> 3, cb = c >> 3, j = c & 7;
  double a = 0;
  for (int n = 0; n < N; ++n)
    for (int s = 0; s < splits; ++s) a += part[(((long)n * CB + cb) * splits + s) * 8 + j];
  out[c] = (float)a;
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
int wgrad_splits(int C, int K) {
  int s = 2048 / ((C / 32) * (K / 32));
  return s < 1 ? 1 : s;
}

}  // namespace

bool convT_h_supported(int C, int K) { return C % 32 == 0 && K % 32 == 0 && C >= 32 && K >= 32; }

size_t convT_h_ws_bytes(int N, int C, int D, int H, int W, int K) {
  (void)D; (void)H; (void)W;
  const size_t wpack = align256((size_t)C * K * 8 * 2);
  const size_t part = align256((size_t)wgrad_splits(C, K) * C * K * 8 * sizeof(float));
  const size_t bsum = align256((size_t)N * (K / 8) * 64 * 8 * sizeof(double));
  return wpack + part + bsum + 256;
}

template <int DT>
static int pack_T(const float* w, void* wp, int C, int K, int dgrad, hipStream_t s) {
  const long total = (long)C * K * 8;
  hipLaunchKernelGGL((k_pack_wT<DT>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (unsigned short*)wp, C, K, dgrad, total);
  return check_launch("pack_wT");
}

// x: dense C8 [N][C/8][D*H*W][8]; out: channels [oc0, oc0 + K) of an octot-channel C8 buffer at (2D, 2H, 2W)
int convT_fwd_h(const void* x, const float* w, const float* bias, void* out, int octot, int oc0, int N, int C, int D, int H, int W,
                int K, int dt, void* ws, size_t wsb, hipStream_t s) {
  if (!convT_h_supported(C, K) || octot % 8 || oc0 % 32 || oc0 + K > octot) { set_error("convT_fwd_h: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || wsb < convT_h_ws_bytes(N, C, D, H, W, K)) { set_error("convT_fwd_h: workspace too small"); return NC_ERR_WS; }
  TParams p{};
  p.x = (const uint4*)x; p.wp = (const uint4*)ws; p.bias = bias; p.out = (uint2*)out;
  p.N = N; p.C = C; p.K = K; p.Dc = D; p.Hc = H; p.Wc = W; p.Sc = (long)D * H * W;
  p.xctot8 = C / 8; p.xc08 = 0; p.octot8 = octot / 8; p.oc08 = oc0 / 8;
  // ~2048 workgroups in all (each keeps its output tile's weights in LDS and walks several voxel tiles)
  long gxl = cdiv(cdiv(p.Sc, 32) * N, 4);
  const long cap = cdiv(2048, K / 32);
  if (gxl > cap) gxl = cap;
  const unsigned gx = (unsigned)gxl;
  const size_t lds = (size_t)8 * (C / 16) * 64 * 16;
  if (lds > 160 * 1024) { set_error("convT_fwd_h: more than 320 input channels"); return NC_ERR_SHAPE; }
  auto launch = [&](auto kern) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      set_error("convT_fwd_h: cannot raise dynamic LDS limit");
      return NC_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(gx, K / 32), dim3(256), lds, s, p);
    return check_launch("convT_fwd_h");
  };
  if (dt == NC_DT_F16) {
    if (int e = pack_T<NC_DT_F16>(w, ws, C, K, 0, s)) return e;
    return launch(k_convT_fwd_h<NC_DT_F16>);
  }
  if (int e = pack_T<NC_DT_BF16>(w, ws, C, K, 0, s)) return e;
  return launch(k_convT_fwd_h<NC_DT_BF16>);
}

// dy: channels [dc0, dc0 + K) of a dctot-channel C8 buffer at (2D, 2H, 2W) (bf16); dx: dense C8 [N][C/8][D*H*W][8] (bf16)
int convT_dgrad_h(const void* dy, int dctot, int dc0, const float* w, void* dx, int N, int C, int D, int H, int W, int K, void* ws,
                  size_t wsb, hipStream_t s) {
  if (!convT_h_supported(C, K) || dctot % 8 || dc0 % 8 || dc0 + K > dctot) { set_error("convT_dgrad_h: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || wsb < convT_h_ws_bytes(N, C, D, H, W, K)) { set_error("convT_dgrad_h: workspace too small"); return NC_ERR_WS; }
  if (int e = pack_T<NC_DT_BF16>(w, ws, C, K, 1, s)) return e;
  TParams p{};
  p.x = (const uint4*)dy; p.wp = (const uint4*)ws; p.out = (uint2*)dx;
  p.N = N; p.C = C; p.K = K; p.Dc = D; p.Hc = H; p.Wc = W; p.Sc = (long)D * H * W;
  p.xctot8 = dctot / 8; p.xc08 = dc0 / 8; p.octot8 = C / 8; p.oc08 = 0;
  const unsigned gx = (unsigned)cdiv(cdiv(p.Sc, 32) * N, 4);
  if ((C / 32) % 4 == 0) hipLaunchKernelGGL((k_convT_dgrad_h<NC_DT_BF16, 4>), dim3(gx, C / 128), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((k_convT_dgrad_h<NC_DT_BF16, 1>), dim3(gx, C / 32), dim3(256), 0, s, p);
  return check_launch("convT_dgrad_h");
}

// x: dense C8 (bf16); dy: view (bf16); dw fp32 [C][K][2][2][2]; dbias fp32 [K] (nullable)
int convT_wgrad_h(const void* x, const void* dy, int dctot, int dc0, float* dw, float* dbias, int N, int C, int D, int H, int W,
                  int K, void* ws, size_t wsb, hipStream_t s) {
  if (!convT_h_supported(C, K) || dctot % 8 || dc0 % 8 || dc0 + K > dctot) { set_error("convT_wgrad_h: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || wsb < convT_h_ws_bytes(N, C, D, H, W, K)) { set_error("convT_wgrad_h: workspace too small"); return NC_ERR_WS; }
  const int splits = wgrad_splits(C, K);
  float* part = (float*)((char*)ws + align256((size_t)C * K * 8 * 2));
  TWParams p{};
  p.x = (const unsigned short*)x; p.dy = (const unsigned short*)dy; p.part = part;
  p.N = N; p.C = C; p.K = K; p.Dc = D; p.Hc = H; p.Wc = W; p.dctot8 = dctot / 8; p.dc08 = dc0 / 8; p.splits = splits;
  p.Sc = (long)D * H * W;
  static const bool shared_dy = true;  // A/B switch
  const int NW = C / 32;
  if (shared_dy && (NW == 1 || NW == 2 || NW == 4 || NW == 8)) {
    static const int vs = 1;  // (2 = 32 voxels per barrier: measured 40 % slower)
    if (vs == 2) {
      const size_t lds = (size_t)(NW * 4 * 36 + 2 * 8 * 4 * 36) * 16;
      hipLaunchKernelGGL((k_convT_wgrad_h2<NC_DT_BF16, 2>), dim3(K / 32, splits), dim3(64 * NW), lds, s, p);
    } else {
      const size_t lds = (size_t)(NW * 4 * 20 + 2 * 8 * 4 * 20) * 16;
      hipLaunchKernelGGL((k_convT_wgrad_h2<NC_DT_BF16, 1>), dim3(K / 32, splits), dim3(64 * NW), lds, s, p);
    }
  } else {
    hipLaunchKernelGGL((k_convT_wgrad_h<NC_DT_BF16>), dim3((C / 32) * (K / 32), splits), dim3(64), 0, s, p);
  }
  const long n = (long)C * K * 8;
  hipLaunchKernelGGL(k_convT_wgrad_reduce, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, part, dw, n, splits);
  if (dbias) {
    double* bp = (double*)((char*)part + align256((size_t)splits * C * K * 8 * sizeof(float)));
    const long Sf = 8 * p.Sc;
    hipLaunchKernelGGL((k_chan_sum_c8<NC_DT_BF16>), dim3(64, N * K / 8), dim3(256), 0, s, (const uint4*)dy, dctot / 8, dc0 / 8, K / 8, Sf,
                       64, bp);
    hipLaunchKernelGGL(k_chan_sum_final, dim3((unsigned)cdiv(K, 64)), dim3(64), 0, s, bp, N, K, 64, dbias);
  }
  return check_launch("convT_wgrad_h");
}

}  // namespace nc
