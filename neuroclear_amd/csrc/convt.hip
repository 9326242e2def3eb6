// ConvTranspose3d(kernel 2, stride 2) forward / dgrad / wgrad (reference models/networks.py:500,503: t_conv2 256->128,
// t_conv1 128->64).  With k == s the op is a per-voxel matrix product: every output voxel depends on exactly one
// input voxel, out[k, 2z+a, 2y+b, 2x+c] = bias[k] + sum_c x[c, z, y, x] * w[c, k, a, b, c'].
// One lane owns one INPUT voxel and KT output channels x 8 taps of accumulators; weights are wave-uniform (scalar
// cache).  1.9 % of the U-Net's FLOPs -- kept on the VALU for round 1.
#include <cstdlib>

#include "common.hpp"
#include "s3_common.hpp"

namespace nc {

template <int KT>
__global__ __launch_bounds__(256) void k_convT_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y, int C, int D,
                                                   int H, int W, int K) {
  const long S = (long)D * H * W;
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * KT, n = blockIdx.z;
  const bool valid = pos < S;
  const long p = valid ? pos : 0;
  const int ix = (int)(p % W), iy = (int)((p / W) % H), iz = (int)(p / ((long)W * H));
  float acc[KT][8];
#pragma unroll
  for (int j = 0; j < KT; ++j)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[j][t] = bias ? bias[k0 + j] : 0.f;
  const float* xn = x + (long)n * C * S + p;
  for (int c = 0; c < C; ++c) {
    const float xv = xn[(long)c * S];
    const float* wc = w + ((long)c * K + k0) * 8;
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[j][t] = fmaf(xv, wc[j * 8 + t], acc[j][t]);
  }
  if (!valid) return;
  const int H2 = 2 * H, W2 = 2 * W;
  const long S2 = 8 * S;
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    float* yk = y + ((long)n * K + k0 + j) * S2;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float2* dst = reinterpret_cast<float2*>(yk + ((long)(2 * iz + a) * H2 + (2 * iy + b)) * W2 + 2 * ix);
        *dst = make_float2(acc[j][(a * 2 + b) * 2], acc[j][(a * 2 + b) * 2 + 1]);
      }
  }
}

// The same forward on the fp32 matrix cores (K % 32 == 0, C even): with k == s each sub-position (a, b, c) is a plain GEMM
// Y_abc[k][voxel] = sum_ci W[ci][k][abc] X[ci][voxel].  A workgroup owns 32 output channels and one a; its weights
// (C x 32 x 4 floats, laid out as MFMA A operands) are gathered into LDS once, and each of its 4 waves walks tiles of 32 voxels:
// ALL input values of a tile are requested first (C / 2 dwords per lane, one HBM round trip per tile), then C / 2 k-steps of
// 4 MFMAs (the (b, c) sub-positions) with the A operand from LDS.  The c = 0 / 1 results of a voxel leave as one float2.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>  // C / 2, compile time: the input values of a tile live in registers
__global__ __launch_bounds__(256, 2) void k_convT_fwd_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C, int D, int H,
                                                           int W, int K, int N, uint2* __restrict__ ys, int oblocks, int ob0) {
  extern __shared__ float wl[];  // [CH][4][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int kt = blockIdx.y >> 1, a = blockIdx.y & 1;
  for (int i = threadIdx.x; i < CH * 4 * 64; i += 256) {
    const int s2 = i >> 8, q = (i >> 6) & 3, ln = i & 63;
    wl[i] = w[((long)(2 * s2 + (ln >> 5)) * K + kt * 32 + (ln & 31)) * 8 + a * 4 + q];
  }
  __syncthreads();
  const long S = (long)D * H * W;
  const long tiles_per_n = (S + 31) / 32, ntiles = tiles_per_n * N;
  const int H2 = 2 * H, W2 = 2 * W;
  const long S2 = 8 * S;
  float bb[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bb[e] = bias ? bias[kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] : 0.f;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const int n = (int)(tile / tiles_per_n);
    const long pos = (tile - (long)n * tiles_per_n) * 32 + li;
    const bool valid = pos < S;
    const long p = valid ? pos : S - 1;
    const float* xp = x + ((long)n * C + h) * S + p;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    // input values in groups of 16 k-steps, the next group requested before the current one is multiplied
    float bc[16], bn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bc[i] = xp[(long)(2 * i) * S];
#pragma unroll 1
    for (int s0 = 0; s0 < CH; s0 += 16) {
      if (s0 + 16 < CH) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bn[i] = xp[(long)(2 * (s0 + 16 + i)) * S];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[((s0 + i) * 4 + q) * 64 + lane], bc[i], acc[q], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) bc[i] = bn[i];
    }
    if (!valid) continue;
    const int ix = (int)(p % W), iy = (int)((p / W) % H), iz = (int)(p / ((long)W * H));
    if (y) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float* yk = y + ((long)n * K + k) * S2;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float2* dst = reinterpret_cast<float2*>(yk + ((long)(2 * iz + a) * H2 + (2 * iy + b)) * W2 + 2 * ix);
          *dst = make_float2(acc[b * 2][e] + bb[e], acc[b * 2 + 1][e] + bb[e]);
        }
      }
    }
    // the three-term (S3) form of the same values for a split-operand convolution that consumes them (conv_split.hip): accumulator
    // rows 4u .. 4u + 3 of this lane are channels 4h .. 4h + 3 of 8-channel block kt * 4 + u -- one 8-byte half of a unit per term
    if (ys) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long blk = ((long)n * oblocks + ob0 + kt * 4 + u) * 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long o = ((long)(2 * iz + a) * H2 + (2 * iy + (q >> 1))) * W2 + 2 * ix + (q & 1);
          unsigned short t3[4][3];
#pragma unroll
          for (int i = 0; i < 4; ++i) s3_split(acc[q][4 * u + i] + bb[4 * u + i], t3[i]);
#pragma unroll
          for (int t = 0; t < 3; ++t)
            ys[((blk + t) * S2 + o) * 2 + h] = make_uint2(t3[0][t] | ((unsigned)t3[1][t] << 16), t3[2][t] | ((unsigned)t3[3][t] << 16));
        }
      }
    }
  }
}

template <int CT>
__global__ __launch_bounds__(256) void k_convT_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                     float* __restrict__ dx, int C, int D, int H, int W, int K) {
  const long S = (long)D * H * W;
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * CT, n = blockIdx.z;
  const bool valid = pos < S;
  const long p = valid ? pos : 0;
  const int ix = (int)(p % W), iy = (int)((p / W) % H), iz = (int)(p / ((long)W * H));
  const int H2 = 2 * H, W2 = 2 * W;
  const long S2 = 8 * S;
  float acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = 0.f;
  for (int k = 0; k < K; ++k) {
    const float* dyk = dy + ((long)n * K + k) * S2;
    float g[8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float2 v = *reinterpret_cast<const float2*>(dyk + ((long)(2 * iz + a) * H2 + (2 * iy + b)) * W2 + 2 * ix);
        g[(a * 2 + b) * 2] = v.x;
        g[(a * 2 + b) * 2 + 1] = v.y;
      }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const float* wc = w + ((long)(c0 + j) * K + k) * 8;
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[j] = fmaf(g[t], wc[t], acc[j]);
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < CT; ++j) dx[((long)n * C + c0 + j) * S + pos] = acc[j];
  }
}

// dw[c][k][t] = sum_{n,pos} x[n][c][pos] * dy[n][k][2pos+t].  Workgroup = (CB input channels) x (KB output channels);
// lanes stride over positions.  CB*KB*8 accumulators per lane.
template <int CB, int KB>
__global__ __launch_bounds__(256) void k_convT_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                     float* __restrict__ dw, int N, int C, int D, int H, int W,
                                                     int K) {
  const int c0 = blockIdx.x * CB, k0 = blockIdx.y * KB;
  const long S = (long)D * H * W, S2 = 8 * S;
  const int H2 = 2 * H, W2 = 2 * W;
  float acc[CB][KB][8];
#pragma unroll
  for (int i = 0; i < CB; ++i)
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[i][j][t] = 0.f;
  for (int n = 0; n < N; ++n) {
    for (long pos = threadIdx.x; pos < S; pos += 256) {
      const int ix = (int)(pos % W), iy = (int)((pos / W) % H), iz = (int)(pos / ((long)W * H));
      float xv[CB];
#pragma unroll
      for (int i = 0; i < CB; ++i) xv[i] = x[((long)n * C + c0 + i) * S + pos];
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const float* dyk = dy + ((long)n * K + k0 + j) * S2;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const float2 v =
                *reinterpret_cast<const float2*>(dyk + ((long)(2 * iz + a) * H2 + (2 * iy + b)) * W2 + 2 * ix);
#pragma unroll
            for (int i = 0; i < CB; ++i) {
              acc[i][j][(a * 2 + b) * 2] = fmaf(xv[i], v.x, acc[i][j][(a * 2 + b) * 2]);
              acc[i][j][(a * 2 + b) * 2 + 1] = fmaf(xv[i], v.y, acc[i][j][(a * 2 + b) * 2 + 1]);
            }
          }
      }
    }
  }
  __shared__ float red[4][CB * KB * 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CB; ++i)
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float v = acc[i][j][t];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0) red[wv][(i * KB + j) * 8 + t] = v;
      }
  __syncthreads();
  if (threadIdx.x < CB * KB * 8) {
    const int e = threadIdx.x, t = e & 7, j = (e >> 3) % KB, i = (e >> 3) / KB;
    dw[((long)(c0 + i) * K + k0 + j) * 8 + t] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
}

static int pick(int n, int a, int b, int c) { return n % a == 0 ? a : n % b == 0 ? b : c; }

}  // namespace nc

using namespace nc;

extern "C" {

// ConvTranspose(k=2,s=2) is the adjoint of a stride-2 2x2x2 convolution with the SAME weight memory layout
// (w[Cin_T][Cout_T][2][2][2] == conv weight [K][C][taps] with K = Cin_T, C = Cout_T): its weight gradient is that
// convolution's weight gradient with the roles of the two tensors swapped, which puts it on the MFMA gather-GEMM.
static bool convT_as_conv(ConvDims& d, int N, int C, int D, int H, int W, int K) {
  return make_dims(d, N, K, 2 * D, 2 * H, 2 * W, C, 2, 2, 2, 2, 0) && gemm_wgrad_supported(d);
}

size_t nc_convT_ws_bytes(int N, int C, int D, int H, int W, int K) {
  size_t b = kBiasGradWsBytes;
  ConvDims d;
  if (convT_as_conv(d, N, C, D, H, W, K) && gemm_ws_bytes(d) > b) b = gemm_ws_bytes(d);
  return (b + 255) & ~(size_t)255;
}

static int convT_check(const char* what, int N, int C, int D, int H, int W, int K) {
  if (N < 1 || C < 1 || D < 1 || H < 1 || W < 1 || K < 1 || K > 65535 * 4 || N > 65535) {
    set_error("%s: bad shape N=%d C=%d D=%d H=%d W=%d K=%d", what, N, C, D, H, W, K);
    return NC_ERR_SHAPE;
  }
  return NC_OK;
}

}  // extern "C"
namespace nc {
// Can the transposed convolution write the S3 form of its output itself (convT_fwd_s3)?  Only the matrix-core kernel does.
bool convT_fwd_s3_supported(int N, int C, int D, int H, int W, int K) {
  static const bool mfma_on = true;
  const long S = (long)D * H * W;
  return mfma_on && !g_force_direct && K % 32 == 0 && C == 128 && (long)N * K * 8 * S < (1L << 40);
}
static int convT_fwd_impl(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W, int K,
                          void* ys, int ctot, int c0, void* stream);
// y (nullable) = the fp32 output; ys = channels [c0, c0 + K) of a ctot-channel S3 tensor [N][ctot/8][3][8S][8] bf16
int convT_fwd_s3(const float* x, const float* w, const float* bias, float* y, void* ys, int ctot, int c0, int N, int C, int D, int H,
                 int W, int K, void* stream) {
  if (!x || !w || !ys) { set_error("convT_fwd_s3: null pointer"); return NC_ERR_ARG; }
  if (!convT_fwd_s3_supported(N, C, D, H, W, K) || ctot % 8 || c0 % 8) { set_error("convT_fwd_s3: shape not covered"); return NC_ERR_SHAPE; }
  return convT_fwd_impl(x, w, bias, y, N, C, D, H, W, K, ys, ctot, c0, stream);
}
}  // namespace nc
extern "C" {

int nc_convT_k2s2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W,
                      int K, void* stream) {
  if (!x || !w || !y) { set_error("convT_fwd: null pointer"); return NC_ERR_ARG; }
  return convT_fwd_impl(x, w, bias, y, N, C, D, H, W, K, nullptr, 0, 0, stream);
}
}  // extern "C"
static int nc::convT_fwd_impl(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W, int K,
                              void* ys, int ctot, int c0, void* stream) {
  if (int e = convT_check("convT_fwd", N, C, D, H, W, K)) return e;
  const long S = (long)D * H * W;
  hipStream_t s = (hipStream_t)stream;
  static const bool mfma_on = true;  // A/B switch
  // (measured: 128 -> 64 at 70^3 0.77 -> 0.55 ms, at 54^3 0.34 -> 0.28; 256 -> 128 at 35^3 0.30 -> 0.36: the VALU kernel keeps it)
  if (mfma_on && !g_force_direct && K % 32 == 0 && C == 128 && (long)N * K * 8 * S < (1L << 40)) {
    long gx = cdiv(cdiv(S, 32) * N, 4);
    const long cap = cdiv(2048, 2 * (K / 32));
    if (gx > cap) gx = cap;
    const dim3 g((unsigned)gx, 2 * (K / 32));
    const size_t lds = (size_t)(C / 2) * 4 * 64 * sizeof(float);
    auto launch = [&](auto kern) -> int {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("convT_fwd: cannot raise dynamic LDS limit");
        return NC_ERR_HIP;
      }
      hipLaunchKernelGGL(kern, g, dim3(256), lds, s, x, w, bias, y, C, D, H, W, K, N, (uint2*)ys, ctot / 8, c0 / 8);
      return check_launch("convT_fwd_mfma");
    };
    return launch(k_convT_fwd_mfma<64>);
  }
  if (ys || !y) { set_error("convT_fwd: the S3 output needs the matrix-core kernel"); return NC_ERR_SHAPE; }
  const int kt = pick(K, 4, 2, 1);
  dim3 grid((unsigned)cdiv(S, 256), K / kt, N);
  if (kt == 4) hipLaunchKernelGGL(k_convT_fwd<4>, grid, dim3(256), 0, s, x, w, bias, y, C, D, H, W, K);
  else if (kt == 2) hipLaunchKernelGGL(k_convT_fwd<2>, grid, dim3(256), 0, s, x, w, bias, y, C, D, H, W, K);
  else hipLaunchKernelGGL(k_convT_fwd<1>, grid, dim3(256), 0, s, x, w, bias, y, C, D, H, W, K);
  return check_launch("convT_fwd");
}
extern "C" {

int nc_convT_k2s2_dgrad(const float* dy, const float* w, float* dx, int N, int C, int D, int H, int W, int K,
                        void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !w || !dx) { set_error("convT_dgrad: null pointer"); return NC_ERR_ARG; }
  if (int e = convT_check("convT_dgrad", N, C, D, H, W, K)) return e;
  ConvDims cd;
  if (!g_force_direct && C >= 64 && convT_as_conv(cd, N, C, D, H, W, K) && gemm_fwd_supported(cd) &&
      (gemm_ws_bytes(cd) == 0 || (ws && ws_bytes >= gemm_ws_bytes(cd)))) {
    // dx[ci][pos] = sum_(co, t) w[ci][co][t] * dy[co][2 pos + t]  ==  the FORWARD pass of Conv3d(K -> C, k 2, s 2, p 0) on dy
    // with the transposed-conv weight read as [C][K][2][2][2]: MFMA gather GEMM (60-70 TFLOP/s) instead of the VALU
    // kernel (33 TFLOP/s at 128 -> 64 channels)
    return conv_fwd_gemm(dy, w, nullptr, dx, cd, ws, ws_bytes, (hipStream_t)stream);
  }
  const long S = (long)D * H * W;
  const int ct = pick(C, 8, 4, 1);
  dim3 grid((unsigned)cdiv(S, 256), C / ct, N);
  hipStream_t s = (hipStream_t)stream;
  if (ct == 8) hipLaunchKernelGGL(k_convT_dgrad<8>, grid, dim3(256), 0, s, dy, w, dx, C, D, H, W, K);
  else if (ct == 4) hipLaunchKernelGGL(k_convT_dgrad<4>, grid, dim3(256), 0, s, dy, w, dx, C, D, H, W, K);
  else hipLaunchKernelGGL(k_convT_dgrad<1>, grid, dim3(256), 0, s, dy, w, dx, C, D, H, W, K);
  return check_launch("convT_dgrad");
}

int nc_convT_k2s2_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int D, int H, int W,
                        int K, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !dy || !dw) { set_error("convT_wgrad: null pointer"); return NC_ERR_ARG; }
  if (int e = convT_check("convT_wgrad", N, C, D, H, W, K)) return e;
  hipStream_t s = (hipStream_t)stream;
  ConvDims cd;
  if (!g_force_direct && convT_as_conv(cd, N, C, D, H, W, K) && ws && ws_bytes >= gemm_ws_bytes(cd)) {
    // dW_T[ci][co][t] = sum_pos x[ci][pos] * dy[co][2 pos + t]  ==  conv_wgrad(input = dy, grad_output = x)
    if (int e = conv_wgrad_gemm(dy, x, dw, cd, ws, ws_bytes, s)) return e;
  } else if (C % 4 == 0 && K % 4 == 0) {
    hipLaunchKernelGGL((k_convT_wgrad<4, 4>), dim3(C / 4, K / 4), dim3(256), 0, s, x, dy, dw, N, C, D, H, W, K);
  } else {
    hipLaunchKernelGGL((k_convT_wgrad<1, 1>), dim3(C, K), dim3(256), 0, s, x, dy, dw, N, C, D, H, W, K);
  }
  if (int e = check_launch("convT_wgrad")) return e;
  if (dbias) return bias_grad(dy, dbias, N, K, 8L * D * H * W, ws, ws_bytes, s);
  return NC_OK;
}

}  // extern "C"
