// Conv3d 1 -> K channels, 3^3, stride 1, padding 1, forward: the first layer of unet_deconv (models/networks.py:420-425 with
// in_channels = 1).  27 taps are the whole reduction, so the general brick kernel (conv_mfma_fwd.hip: 8-channel chunks) spent 7 of 8
// MFMA k-steps on padding channels and ran at ~20 TFLOP/s; the layer is bound by writing its output (K x volume fp32).
// Here: v_mfma_f32_32x32x2_f32 with K-dim = taps (14 k-steps of 2: taps 0 .. 26 and one zero-weight tap), rows = 32 output
// channels, columns = 32 consecutive positions of one row group.  A workgroup stages the 3 x (R + 2) x (W + 2) input rows it needs
// (zero padded) in LDS; a tap is then a constant offset of a lane's position; the weights of a wave's 64 output channels live in 28
// registers.  fp32 products, fp32 accumulation: the arithmetic of the kernel it replaces, in a different summation order.
#include <cstdlib>

#include "common.hpp"

namespace nc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct C1Params {
  const float* x;     // [N][1][D][H][W]
  const float* w;     // [K][1][3][3][3]
  const float* bias;  // nullable
  float* y;           // [N][K][D][H][W]
  float2* stats;      // ST: [gridDim.x][K / 64][wave][64 channels] (sum, sum of squares) of this wave's bias-free outputs (N = 1 launches only)
  int N, D, H, W, K;
  int R;              // output rows per workgroup
  int P;              // W + 2
  int YB;             // ceil(H / R)
  long ngroups;       // N * D * YB
};

constexpr int kC1Waves = 4;

// ST: the wave also leaves the sum and the sum of squares of its 64 channels' outputs WITHOUT the bias (as conv_s3x.hip's ST epilogue does for the
// 64-channel layers): the InstanceNorm of the U-Net's first block then needs no pass over the 64-channel output (k_in_stats: 0.2 ms of a
// 140^3 inference cube).  fp32 per lane (a few dozen values), fp64 across waves and workgroups in c1k3_stats_finalize, fixed order.
template <bool ST>
__global__ void __launch_bounds__(kC1Waves * 64) k_conv_c1k3(const C1Params p) {
  extern __shared__ float xt[];  // [3][R + 2][P]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, kh = lane >> 5;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int plane = (p.R + 2) * p.P;
  const int kt = blockIdx.y;  // 64 output channels

  // A operands: lane (row r = lane % 32, kh) holds w[kt 64 + a 32 + r][tap 2 s + kh] for k-step s, a = 0, 1
  float aw[2][14];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int t = 2 * s + kh;
      aw[a][s] = t < 27 ? p.w[(long)(kt * 64 + a * 32 + col) * 27 + t] : 0.f;
    }
  // tap offsets inside the staged tile for this lane's half of every k-step (tap 27: any in-range offset, its weight is zero)
  int toff[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int t = 2 * s + kh < 27 ? 2 * s + kh : 26;
    toff[s] = (t / 9) * plane + ((t / 3) % 3) * p.P + t % 3;
  }
  float bv[2][16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[a][e] = p.bias ? p.bias[kt * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh] : 0.f;

  float ss[2][16], sq[2][16];
  if constexpr (ST) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) { ss[a][e] = 0.f; sq[a][e] = 0.f; }
  }
  for (long grp = blockIdx.x; grp < p.ngroups; grp += gridDim.x) {
    const int yb = (int)(grp % p.YB);
    const int z = (int)((grp / p.YB) % p.D);
    const int n = (int)(grp / ((long)p.YB * p.D));
    const int y0 = yb * p.R;
    const int rows = p.H - y0 < p.R ? p.H - y0 : p.R;
    __syncthreads();  // the previous group's reads are done
    for (int i = tid; i < 3 * plane; i += kC1Waves * 64) {
      const int dz = i / plane, rem = i - dz * plane;
      const int ry = rem / p.P, cx = rem - ry * p.P;
      const int zz = z + dz - 1, yy = y0 + ry - 1, xx = cx - 1;
      const bool ok = (unsigned)zz < (unsigned)p.D && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      xt[i] = ok ? p.x[(long)n * S + (long)zz * HW + (long)yy * p.W + xx] : 0.f;
    }
    __syncthreads();
    const int npos = rows * p.W;  // output positions of this group, row-major
    for (int c0 = wave * 32; c0 < npos; c0 += kC1Waves * 32) {
      const int q = c0 + col;
      const int qc = q < npos ? q : npos - 1;
      const int ry = qc / p.W, cx = qc - ry * p.W;
      const float* xb = xt + ry * p.P + cx;
      f32x16 acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
      float b[14];
#pragma unroll
      for (int s = 0; s < 14; ++s) b[s] = xb[toff[s]];
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[0][s], b[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[1][s], b[s], acc1, 0, 0, 0);
      }
      if (q < npos) {
        float* yo = p.y + ((long)n * p.K + kt * 64 + 4 * kh) * S + (long)z * HW + (long)(y0 + ry) * p.W + cx;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long ro = (long)((e & 3) + 8 * (e >> 2)) * S;
          yo[ro] = acc0[e] + bv[0][e];
          yo[ro + 32 * S] = acc1[e] + bv[1][e];
          if constexpr (ST) {
            ss[0][e] += acc0[e]; sq[0][e] = fmaf(acc0[e], acc0[e], sq[0][e]);
            ss[1][e] += acc1[e]; sq[1][e] = fmaf(acc1[e], acc1[e], sq[1][e]);
          }
        }
      }
    }
  }
  if constexpr (ST) {
    // the 32 lanes of a half (same kh) hold the same 32 channels at different positions: butterfly over them, lane col == 0 writes
    float2* rec = p.stats + (((long)blockIdx.x * gridDim.y + kt) * kC1Waves + wave) * 64;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float u = ss[a][e], v = sq[a][e];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { u += __shfl_xor(u, o); v += __shfl_xor(v, o); }
        if (col == 0) rec[a * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh] = make_float2(u, v);
      }
  }
}

// one workgroup per channel: the records of every (workgroup, wave) in index order, fp64
__global__ void __launch_bounds__(256) k_c1k3_stats_final(const float2* __restrict__ part, int nrec, int K, const float* __restrict__ bias, long S, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ double rs[256], rq[256];
  const int c = blockIdx.x, kt = c >> 6, ci = c & 63, t = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int r = t; r < nrec; r += 256) {  // r = workgroup * kC1Waves + wave
    const float2 v = part[(((long)(r / kC1Waves) * (K / 64) + kt) * kC1Waves + r % kC1Waves) * 64 + ci];
    s += (double)v.x; q += (double)v.y;
  }
  rs[t] = s; rq[t] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { rs[t] += rs[t + o]; rq[t] += rq[t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    const double m0 = rs[0] / (double)S;
    double var = rq[0] / (double)S - m0 * m0;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)(m0 + (bias ? (double)bias[c] : 0.0));
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

}  // namespace

bool c1k3_fwd_supported(const ConvDims& d) {
  static const bool on = !(getenv("NC_C1K3") && atoi(getenv("NC_C1K3")) == 0);  // A/B switch: the general brick kernel
  if (!on || d.C != 1 || d.K % 64 || d.kd != 3 || d.kh != 3 || d.kw != 3) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != 1 || d.ph != 1 || d.pw != 1) return false;
  // the smallest tile (one output row: 3 planes x 3 rows of W + 2 floats) has to fit the 48 KB the launch stays within
  if ((size_t)36 * (d.W + 2) > 48 * 1024) return false;
  return d.W >= 8 && (long)d.N * d.K * d.D * d.H * d.W < (1L << 40);
}

static long c1k3_grid(const ConvDims& d, int& R) {
  R = (512 + d.W - 1) / d.W;
  if (R < 1) R = 1;
  if (R > d.H) R = d.H;
  while (R > 1 && (size_t)3 * (R + 2) * (d.W + 2) * sizeof(float) > 48 * 1024) --R;
  const long ngroups = (long)d.N * d.D * ((d.H + R - 1) / R);
  return ngroups < 2048 ? ngroups : 2048;
}
size_t c1k3_stats_bytes(const ConvDims& d) {
  int R;
  return d.N == 1 && c1k3_fwd_supported(d) ? (size_t)c1k3_grid(d, R) * (d.K / 64) * kC1Waves * 64 * sizeof(float2) + 256 : 0;
}
int c1k3_stats_finalize(const float* stats_part, const float* bias, const ConvDims& d, float eps, float* mean, float* rstd, hipStream_t s) {
  int R;
  if (!c1k3_stats_bytes(d) || !stats_part) { set_error("c1k3_stats_finalize: shape not covered"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_c1k3_stats_final, dim3((unsigned)d.K), dim3(256), 0, s, (const float2*)stats_part, (int)c1k3_grid(d, R) * kC1Waves, d.K, bias,
                     (long)d.D * d.H * d.W, eps, mean, rstd);
  return check_launch("c1k3_stats_finalize");
}

int conv_fwd_c1k3(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, hipStream_t s, float* stats_part) {
  C1Params p{};
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.stats = (float2*)stats_part;
  if (stats_part && d.N != 1) { set_error("conv_c1k3: epilogue statistics for one sample per call only"); return NC_ERR_ARG; }
  p.N = d.N; p.D = d.D; p.H = d.H; p.W = d.W; p.K = d.K;
  p.P = d.W + 2;
  // rows per workgroup: ~512 positions, and the tile within 48 KB of LDS
  int R = (512 + d.W - 1) / d.W;
  if (R < 1) R = 1;
  if (R > d.H) R = d.H;
  while (R > 1 && (size_t)3 * (R + 2) * p.P * sizeof(float) > 48 * 1024) --R;
  p.R = R;
  p.YB = (d.H + R - 1) / R;
  p.ngroups = (long)d.N * d.D * p.YB;
  const size_t lds = (size_t)3 * (R + 2) * p.P * sizeof(float);
  long gx = p.ngroups < 2048 ? p.ngroups : 2048;
  if (stats_part) hipLaunchKernelGGL(k_conv_c1k3<true>, dim3((unsigned)gx, (unsigned)(d.K / 64)), dim3(kC1Waves * 64), lds, s, p);
  else hipLaunchKernelGGL(k_conv_c1k3<false>, dim3((unsigned)gx, (unsigned)(d.K / 64)), dim3(kC1Waves * 64), lds, s, p);
  return check_launch("conv_c1k3");
}

}  // namespace nc
