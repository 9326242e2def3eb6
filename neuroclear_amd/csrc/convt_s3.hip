// ConvTranspose3d(kernel 2, stride 2) forward on the bf16 matrix cores with the exact three-term operand split of conv_split.hip
// (models/networks.py:471-478: the two nn.ConvTranspose3d of unet_deconv).  With kernel == stride each of the 8 sub-positions
// q = (a, b, c) of an output voxel (2z + a, 2y + b, 2x + c) is a plain GEMM
//     Y_q[k][v] = sum_ci W[ci][k][q] X[ci][v]
// M = output channels, N = input voxels, K-dim = input channels (128 or 256: 4 or 8 k-steps of 32).  The input arrives in S3 form
// (three bf16 terms, [N][C/8][3][S] 16-byte units): a B fragment of v_mfma_f32_16x16x32_bf16 -- lane (voxel m = lane % 16, lane
// group g = lane / 16) holding 8 consecutive K-values -- IS one unit of chunk 4s + g, so B comes straight from global memory,
// coalesced, and is shared by all sub-positions.  A workgroup owns 16 output channels x QN sub-positions; its packed weights (all
// k-steps: (C / 32) QN 3 KiB <= 96 KiB) sit in LDS as A fragments; its 8 waves walk 64-voxel tiles (4 column blocks), each holding
// QN x 4 accumulator tiles.  Six products per fp32 product, smallest first, fp32 accumulation (K-dim <= 256: no accumulator restarts).
// The results leave as the S3 form of the output (8-byte halves of a unit: a lane holds channels 4g .. 4g + 3) and / or as fp32.
// The fp32 matrix-core kernel this replaces (convt.hip k_convT_fwd_mfma) re-read the input once per (32 channels, a) and ran at
// ~75 TFLOP/s; the 256 -> 128 layer was still on the VALU kernel.
#include <cstdlib>

#include "common.hpp"
#include "s3_common.hpp"

namespace nc {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kTWaves = 8;
constexpr int kTThreads = kTWaves * 64;
constexpr int kTCB = 4;  // column blocks (16 voxels) per wave and tile

// wp[kg][qg][s][ql][term][lane][8] bf16: lane (m = lane % 16, g = lane / 16) = W[ci = (4 s + g) 8 + j][k = kg 16 + m][q = qg QN + ql]
// NT = 2: the two fp16 terms of w * 2^k, k from the weights' cell (h2.hip)
template <int NT>
__global__ void __launch_bounds__(256) k_pack_wT_s3(const float* __restrict__ w, unsigned short* __restrict__ wp, int C, int K, int QN, long total,
                                                    const unsigned* __restrict__ wcell) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7);
  long r = i >> 3;
  const int lane = (int)(r & 63); r >>= 6;
  const int term = (int)(r % NT); r /= NT;
  const int ql = (int)(r % QN); r /= QN;
  const int NS = C / 32;
  const int s = (int)(r % NS); r /= NS;
  const int nqg = 8 / QN;
  const int qg = (int)(r % nqg);
  const int kg = (int)(r / nqg);
  const int m = lane & 15, g = lane >> 4;
  const int ci = (4 * s + g) * 8 + j, k = kg * 16 + m, q = qg * QN + ql;
  unsigned short t[3];
  if constexpr (NT == 3) s3_split(w[((long)ci * K + k) * 8 + q], t);
  else h2_split(w[((long)ci * K + k) * 8 + q] * h2_scale(*wcell), t);
  wp[i] = t[term];
}

struct TParams3 {
  const uint4* xs;    // S3 input [N][C/8][3][S]
  const uint4* wp;    // packed weights
  const float* bias;  // nullable
  float* y;           // nullable: fp32 output [N][K][8 S]
  uint2* ys;          // nullable: S3 output, halves of units: channels [c0, c0 + K) of [N][oblocks][3][8 S][2]
  int N, C, K, D, H, W;
  int oblocks, ob0;
  const unsigned *xcell, *wcell;  // NT = 2: the input is an H2 tensor with this cell, the packed weights carry that one
  const unsigned* h2cell;  // nullable: ys is an H2 tensor (two fp16 terms of y * 2^k, h2.hip) and this its cell -- a BOUND of |y| set before the launch
  int ngroups;        // (K / 16) * (8 / QN)
  long ntiles;        // N * ceil(S / 512)
};

template <int QN, int NT = 3>
__global__ void __launch_bounds__(kTThreads, 1) k_convT_s3(const TParams3 p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  const long S = (long)p.D * p.H * p.W;
  const int NS = p.C / 32;
  constexpr int NQG = 8 / QN;
  // neighbours on an XCD (blockIdx % 8) are the groups of ONE tile slot: they read the same input units out of that XCD's L2
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int grp = jx % p.ngroups;
  const long slot = (long)(jx / p.ngroups) * 8 + xcd;
  const long nslots = (long)(gridDim.x / (8 * p.ngroups)) * 8;
  const int kg = grp / NQG, qg = grp % NQG;

  // this workgroup's weights -> LDS (A fragments, 1 KiB each: [s][ql][term][lane])
  {
    const uint4* src = p.wp + (long)grp * NS * QN * NT * 64;
    uint4* dst = reinterpret_cast<uint4*>(lds_raw);
    for (int i = tid; i < NS * QN * NT * 64; i += kTThreads) dst[i] = src[i];
  }
  __syncthreads();
  const i32x4* const wl = reinterpret_cast<const i32x4*>(lds_raw) + lane;

  const long tiles_per_n = (S + 511) / 512;
  const int H2 = 2 * p.H, W2 = 2 * p.W;
  const long S2 = 8 * S;
  float bb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bb[e] = p.bias ? p.bias[kg * 16 + 4 * g + e] : 0.f;
  float oscx = 1.f, oscw = 1.f;  // NT = 2: the sums are scaled back by 2^-(kx + kw)
  if constexpr (NT == 2) { const float2 f = h2_unscale2(*p.xcell, *p.wcell); oscx = f.x; oscw = f.y; }

  for (long tile = slot; tile < p.ntiles; tile += nslots) {
    const int n = (int)(tile / tiles_per_n);
    const long v0 = (tile - (long)n * tiles_per_n) * 512 + wave * 64;
    if (v0 >= S) continue;
    long vv[kTCB];  // this lane's voxel in column block cb (clamped; stores are masked)
#pragma unroll
    for (int cb = 0; cb < kTCB; ++cb) {
      const long v = v0 + cb * 16 + m16;
      vv[cb] = v < S ? v : S - 1;
    }
    f32x4 acc[QN][kTCB];
#pragma unroll
    for (int q = 0; q < QN; ++q)
#pragma unroll
      for (int cb = 0; cb < kTCB; ++cb) acc[q][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4* const xn = p.xs + (long)n * (p.C / 8) * NT * S;
    i32x4 B[kTCB][NT];
    auto load_b = [&](int s) __attribute__((always_inline)) {
      const uint4* xc = xn + (long)(4 * s + g) * NT * S;
#pragma unroll
      for (int cb = 0; cb < kTCB; ++cb)
#pragma unroll
        for (int t = 0; t < NT; ++t) B[cb][t] = __builtin_bit_cast(i32x4, xc[(long)t * S + vv[cb]]);
    };
    load_b(0);
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      i32x4 Bc[kTCB][NT];
#pragma unroll
      for (int cb = 0; cb < kTCB; ++cb)
#pragma unroll
        for (int t = 0; t < NT; ++t) Bc[cb][t] = B[cb][t];
      if (s + 1 < NS) load_b(s + 1);  // the next k-step's units are requested while this one is multiplied
      const i32x4* ws = wl + (long)s * QN * NT * 64;
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        // six (NT = 2: three) products per fp32 product, smallest first: (term of A, term of B); the four column blocks alternate so that
        // consecutive MFMAs go to different accumulators
        i32x4 A[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) A[t] = ws[(q * NT + t) * 64];
        constexpr int NP = NT == 3 ? 6 : 3;
        constexpr int TA[6] = {NT - 1, NT == 3 ? 1 : 0, 0, 1, 0, 0};
        constexpr int TB[6] = {0, 1, NT == 3 ? 2 : 0, 0, 1, 0};
#pragma unroll
        for (int m = 0; m < NP; ++m)
#pragma unroll
          for (int cb = 0; cb < kTCB; ++cb) {
            if constexpr (NT == 3)
              acc[q][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[TA[m]]), __builtin_bit_cast(bf16x8, Bc[cb][TB[m]]),
                                                                   acc[q][cb], 0, 0, 0);
            else
              acc[q][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[TA[m]]), __builtin_bit_cast(f16x8, Bc[cb][TB[m]]),
                                                                  acc[q][cb], 0, 0, 0);
          }
      }
    }
    // ---- results: accumulator element e of (q, cb) = channel kg 16 + 4 g + e at voxel cb 16 + m16, sub-position qg QN + q
#pragma unroll
    for (int cb = 0; cb < kTCB; ++cb) {
      const long v = v0 + cb * 16 + m16;
      if (v >= S) continue;
      const int ix = (int)(v % p.W), iy = (int)((v / p.W) % p.H), iz = (int)(v / ((long)p.W * p.H));
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int qq = qg * QN + q;
        const long o = ((long)(2 * iz + (qq >> 2)) * H2 + (2 * iy + ((qq >> 1) & 1))) * W2 + 2 * ix + (qq & 1);
        float val[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = NT == 2 ? acc[q][cb][e] * oscx * oscw + bb[e] : acc[q][cb][e] + bb[e];
        if (p.y) {
          float* yk = p.y + ((long)n * p.K + kg * 16 + 4 * g) * S2 + o;
#pragma unroll
          for (int e = 0; e < 4; ++e) yk[(long)e * S2] = val[e];
        }
        if (p.ys && p.h2cell) {
          unsigned short t3[4][3];
          const float sc = h2_scale(*p.h2cell);
#pragma unroll
          for (int e = 0; e < 4; ++e) h2_split(val[e] * sc, t3[e]);
          const long blk = ((long)n * p.oblocks + p.ob0 + kg * 2 + (g >> 1)) * 2;
#pragma unroll
          for (int t = 0; t < 2; ++t)
            p.ys[((blk + t) * S2 + o) * 2 + (g & 1)] = make_uint2(t3[0][t] | ((unsigned)t3[1][t] << 16), t3[2][t] | ((unsigned)t3[3][t] << 16));
        } else if (p.ys) {
          unsigned short t3[4][3];
#pragma unroll
          for (int e = 0; e < 4; ++e) s3_split(val[e], t3[e]);
          const long blk = ((long)n * p.oblocks + p.ob0 + kg * 2 + (g >> 1)) * 3;
#pragma unroll
          for (int t = 0; t < 3; ++t)
            p.ys[((blk + t) * S2 + o) * 2 + (g & 1)] = make_uint2(t3[0][t] | ((unsigned)t3[1][t] << 16), t3[2][t] | ((unsigned)t3[3][t] << 16));
        }
      }
    }
  }
}

// |y| of ConvTranspose3d(k 2, s 2) is bounded without looking at y: every output voxel receives exactly ONE tap per input channel, so
// |y[co]| <= max over (co, tap) of sum_ci |w[ci][co][tap]| * max|x| + max|b|.  With x an InstanceNorm output (max|x| <= sqrt(S)) the bound is
// loose by 2^7-2^9 against typical values -- inside what fp16's exponent range forgives (s3_common.hpp) -- and it lets the kernel write the
// H2 form of its output itself.  cell (zeroed by the caller) <- float bits of the bound, atomicMax over the blocks.
// (64 columns x 4 slices of the input channels per workgroup, loads of a slice in flight together: the first version walked all C channels in
// one dependent chain per thread -- 75 us for a 262k-element weight tensor, twice per inference cube and per training step)
__global__ void __launch_bounds__(256) k_convT_bound(const float* __restrict__ w, const float* __restrict__ bias, int C, int K, float in_bound,
                                                     unsigned* __restrict__ cell) {
  __shared__ float part[4][64];
  __shared__ float red[64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int kq = blockIdx.x * 64 + col;  // one (output channel, tap) column: coalesced over the columns
  float sabs = 0.f;
  if (kq < K * 8) {
#pragma unroll 8
    for (int ci = sl; ci < C; ci += 4) sabs += fabsf(w[(long)ci * K * 8 + kq]);
  }
  part[sl][col] = sabs;
  __syncthreads();
  if (sl == 0) {
    const float t = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
    red[col] = kq < K * 8 ? t * in_bound + (bias ? fabsf(bias[kq >> 3]) : 0.f) : 0.f;
  }
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x + o] > red[threadIdx.x] ? red[threadIdx.x + o] : red[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(cell, __float_as_uint(red[0] * 1.001f) & 0x7fffffffu);  // (the cell was zeroed by the caller)
}

int qn_for(int C) { return (C / 32) * 8 * 3 * 1024 <= 128 * 1024 ? 8 : 4; }

}  // namespace

bool convT_s3x_supported(int N, int C, int D, int H, int W, int K) {
  static const bool on = !(getenv("NC_CONVT_S3X") && atoi(getenv("NC_CONVT_S3X")) == 0);  // A/B switch: the fp32 kernels of convt.hip
  const long S = (long)D * H * W;
  if (!on || C % 32 || K % 16 || C > 256 || N < 1) return false;
  if ((C / 32) * qn_for(C) * 3 * 1024 > 128 * 1024) return false;
  return (long)N * K * 8 * S < (1L << 40) && S * 16 * 3 * (C / 8) < (1L << 40);
}
size_t convT_s3x_ws_bytes(int C, int K) { return (size_t)C * K * 8 * 3 * 2 + 256; }

// xs: the input in S3 form; y (nullable) fp32 output; ys (nullable) channels [c0, c0 + K) of a ctot-channel S3 tensor; ws: packed weights
int convT_h2_bound(const float* w, const float* bias, int C, int K, float in_bound, unsigned* cell, hipStream_t s) {
  hipLaunchKernelGGL(k_convT_bound, dim3((unsigned)cdiv((long)K * 8, 64)), dim3(256), 0, s, w, bias, C, K, in_bound, cell);
  return check_launch("convT_h2_bound");
}

// h2cell (nullable): ys is an H2 tensor (4 bytes per element) and *h2cell the bound its power of two comes from (convT_h2_bound).
// xcell (nullable): the INPUT xs is an H2 tensor with this cell -- the two-term form of the kernel (three products; the weights' cell is
// measured into the workspace behind the packed weights)
int convT_fwd_s3x(const void* xs, const float* w, const float* bias, float* y, void* ys, int ctot, int c0, int N, int C, int D, int H, int W,
                  int K, void* ws, size_t wsb, hipStream_t s, const unsigned* h2cell, const unsigned* xcell) {
  if (!xs || !w || (!y && !ys) || !ws) { set_error("convT_fwd_s3x: null pointer"); return NC_ERR_ARG; }
  if (!convT_s3x_supported(N, C, D, H, W, K) || ctot % 8 || c0 % 8) { set_error("convT_fwd_s3x: shape not covered"); return NC_ERR_SHAPE; }
  if (wsb < convT_s3x_ws_bytes(C, K)) { set_error("convT_fwd_s3x: workspace too small"); return NC_ERR_WS; }
  const int QN = qn_for(C);
  const int NT = xcell ? 2 : 3;
  const long total = (long)C * K * 8 * NT;
  unsigned* wcell = (unsigned*)((char*)ws + (size_t)C * K * 8 * 3 * 2);  // (the 256 bytes of slack behind the packed weights)
  if (NT == 2) {
    if (int e = h2_zero_cells(wcell, 1, s)) return e;
    if (int e = h2_absmax(w, (long)C * K * 8, wcell, s)) return e;
    hipLaunchKernelGGL(k_pack_wT_s3<2>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (unsigned short*)ws, C, K, QN, total, (const unsigned*)wcell);
  } else {
    hipLaunchKernelGGL(k_pack_wT_s3<3>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (unsigned short*)ws, C, K, QN, total, (const unsigned*)nullptr);
  }
  if (int e = check_launch("pack_wT_s3")) return e;
  const long S = (long)D * H * W;
  TParams3 p{};
  p.xs = (const uint4*)xs; p.wp = (const uint4*)ws; p.bias = bias; p.y = y; p.ys = (uint2*)ys;
  p.N = N; p.C = C; p.K = K; p.D = D; p.H = H; p.W = W;
  p.oblocks = ctot / 8; p.ob0 = c0 / 8; p.h2cell = ys ? h2cell : nullptr; p.xcell = xcell; p.wcell = wcell;
  p.ngroups = (K / 16) * (8 / QN);
  p.ntiles = (long)N * cdiv(S, 512);
  // 256 workgroups (one per CU: the weights take most of its LDS), a whole number of tile slots per XCD
  int per_xcd = 32 / p.ngroups;
  if (per_xcd < 1) per_xcd = 1;
  const unsigned grid = (unsigned)(8 * p.ngroups * per_xcd);
  const size_t lds = (size_t)(C / 32) * QN * NT * 1024;
  auto launch = [&](auto kern) -> int {
    if (int e = raise_dyn_lds(kern, 160 * 1024, "convT_fwd_s3x")) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTThreads), lds, s, p);
    return check_launch("convT_s3");
  };
  if (NT == 2) return QN == 8 ? launch(k_convT_s3<8, 2>) : launch(k_convT_s3<4, 2>);
  return QN == 8 ? launch(k_convT_s3<8, 3>) : launch(k_convT_s3<4, 3>);
}

}  // namespace nc
