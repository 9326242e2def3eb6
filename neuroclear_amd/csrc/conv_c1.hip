// Convolution onto ONE output channel from many input channels with a large cubic kernel (7^3, stride 1, same padding):
// the data gradient of deep_linear_gen's first layer (reference models/networks.py:899: Conv3d(1, 64, 7, padding=3),
// dgrad = correlation of the 64-channel dY with the flipped kernels).  M = 1 makes the matrix cores useless (a 32-row
// MFMA would be 97 % padding), so this one runs on the VALU at its own roofline: each lane owns 2 (z) x 8 (x) outputs,
// weights are wave-uniform scalars (s_load -> SGPR operand of v_fmac), inputs come from an LDS brick read as aligned
// 16-byte vectors; every 14-float row read feeds 2 x 56 FMAs.
#include "common.hpp"

namespace nc {

template <int KS>
__global__ __launch_bounds__(256) void k_conv_to1(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ y, int C, int D,
                                                  int H, int W, int flip) {
  constexpr int PAD = KS / 2, TZ = 8, TY = 8, TX = 64, XB = 8;
  constexpr int PZ = TZ + KS - 1, PY = TY + KS - 1, PXU = TX + KS - 1;
  constexpr int PITCH = 76;  // >= PXU, multiple of 4, == 12 (mod 64): conflict-free ds_read_b128 over two rows
  constexpr int TAPS = KS * KS * KS;
  static_assert(PXU <= PITCH, "pitch");
  __shared__ __attribute__((aligned(16))) float lds[PZ * PY * PITCH];
  // this channel's kernel in the order the loop uses it, one 8-float (32-byte) row per (dz, dy): broadcast
  // ds_read_b128 instead of KS dependent scalar loads per row (the scalar-load latency was exposed)
  __shared__ __attribute__((aligned(16))) float wl[KS * KS * 8];
  const int tid = threadIdx.x;
  const int xb = tid & 7, ty = (tid >> 3) & 7, tzp = tid >> 6;
  const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
  const int bx = blockIdx.x % ntx, by = (blockIdx.x / ntx) % nty, bz = blockIdx.x / (ntx * nty);
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ, n = blockIdx.y;
  const long HW = (long)H * W, S = (long)D * HW;
  const float* xn = x + (long)n * C * S;

  float acc[2][XB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < XB; ++i) acc[a][i] = 0.f;

  for (int c = 0; c < C; ++c) {
    const float* xc = xn + (long)c * S;
    __syncthreads();
    for (int e = tid; e < PZ * PY * PXU; e += 256) {
      const int xx = e % PXU, yy = (e / PXU) % PY, pz = e / (PXU * PY);
      const int gz = z0 + pz - PAD, gy = y0 + yy - PAD, gx = x0 + xx - PAD;
      float v = 0.f;
      if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = xc[(long)gz * HW + (long)gy * W + gx];
      lds[(pz * PY + yy) * PITCH + xx] = v;
    }
    const float* wc = w + (long)c * TAPS;
    for (int e = tid; e < TAPS; e += 256) wl[(e / KS) * 8 + e % KS] = wc[flip ? TAPS - 1 - e : e];
    __syncthreads();
#pragma unroll 1
    for (int pz = 0; pz <= KS; ++pz) {
      const float* plane = lds + ((2 * tzp + pz) * PY + ty) * PITCH + xb * XB;
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        float row[XB + KS + 1];  // 16 floats = 4 aligned float4 (XB + KS - 1 = 14 used)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          const float4 t = *reinterpret_cast<const float4*>(plane + dy * PITCH + 4 * v4);
          row[4 * v4] = t.x; row[4 * v4 + 1] = t.y; row[4 * v4 + 2] = t.z; row[4 * v4 + 3] = t.w;
        }
        if (pz < KS) {  // output plane z0 + 2 tzp uses tap dz = pz
          const float4 wa = *reinterpret_cast<const float4*>(wl + (pz * KS + dy) * 8);
          const float4 wb = *reinterpret_cast<const float4*>(wl + (pz * KS + dy) * 8 + 4);
          const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int dx = 0; dx < KS; ++dx)
#pragma unroll
            for (int i = 0; i < XB; ++i) acc[0][i] = fmaf(wv[dx], row[i + dx], acc[0][i]);
        }
        if (pz >= 1) {  // output plane z0 + 2 tzp + 1 uses tap dz = pz - 1
          const float4 wa = *reinterpret_cast<const float4*>(wl + ((pz - 1) * KS + dy) * 8);
          const float4 wb = *reinterpret_cast<const float4*>(wl + ((pz - 1) * KS + dy) * 8 + 4);
          const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int dx = 0; dx < KS; ++dx)
#pragma unroll
            for (int i = 0; i < XB; ++i) acc[1][i] = fmaf(wv[dx], row[i + dx], acc[1][i]);
        }
      }
    }
  }
  const float b0 = bias ? bias[0] : 0.f;
  const int gy = y0 + ty;
  if (gy < H) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int gz = z0 + 2 * tzp + a;
      if (gz < D) {
        float* yr = y + (long)n * S + (long)gz * HW + (long)gy * W;
#pragma unroll
        for (int i = 0; i < XB; ++i) {
          const int gx = x0 + xb * XB + i;
          if (gx < W) yr[gx] = acc[a][i] + b0;
        }
      }
    }
  }
}

// dgrad of a Conv3d with Cin == 1: dx[n][0][q] = sum_k sum_t w[k][0][t] * dy[n][k][q + p - t]
bool to1_dgrad_supported(const ConvDims& d) {
  return d.C == 1 && d.K >= 8 && d.kd == 7 && d.kh == 7 && d.kw == 7 && d.sd == 1 && d.sh == 1 && d.sw == 1 &&
         d.pd == 3 && d.ph == 3 && d.pw == 3 && d.N <= 65535;
}

int conv_dgrad_to1(const float* dy, const float* w, float* dx, const ConvDims& d, hipStream_t s) {
  const int ntx = (d.W + 63) / 64, nty = (d.H + 7) / 8, ntz = (d.D + 7) / 8;
  dim3 grid((unsigned)(ntx * nty * ntz), d.N);
  hipLaunchKernelGGL(k_conv_to1<7>, grid, dim3(256), 0, s, dy, w, (const float*)nullptr, dx, d.K, d.D, d.H, d.W, 1);
  return check_launch("conv_dgrad_to1");
}

}  // namespace nc
