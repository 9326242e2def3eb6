// Direct (VALU) convolution kernels: the general path for every shape the MFMA implicit-GEMM kernels do not cover
// (Cin = 1 / Cout = 1 layers, the 1x1 tails, the strided 4x4 PatchGAN convs) and the on-device cross-check for them.
// Replaces nn.Conv3d / nn.Conv2d forward + autograd backward (reference models/networks.py:361-369 call sites).
//
// Design: one lane = one output position, register-blocked over KT output channels; weights are wave-uniform so
// they travel through the scalar cache (s_load), x reads are coalesced along W.  HBM-side the kernels are simple
// streams; there is no LDS staging here on purpose -- the hot 3^3/5^3 layers live in conv_mfma.hip.
#include "common.hpp"

namespace nc {

template <int KT>
__global__ __launch_bounds__(256) void k_conv_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ b, float* __restrict__ y, ConvDims d) {
  const long So = (long)d.Do * d.Ho * d.Wo;
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * KT, n = blockIdx.z;
  const bool valid = pos < So;
  const long p = valid ? pos : 0;
  const int ow = (int)(p % d.Wo), oh = (int)((p / d.Wo) % d.Ho), od = (int)(p / ((long)d.Wo * d.Ho));
  float acc[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) acc[j] = b ? b[k0 + j] : 0.f;
  const long HW = (long)d.H * d.W, S = (long)d.D * HW;
  const int taps = d.kd * d.kh * d.kw;
  const float* xn = x + (long)n * d.C * S;
  const long wks = (long)d.C * taps;  // stride between output channels in w
  const int iz0 = od * d.sd - d.pd, iy0 = oh * d.sh - d.ph, ix0 = ow * d.sw - d.pw;
  for (int c = 0; c < d.C; ++c) {
    const float* xc = xn + c * S;
    const float* wc = w + (long)k0 * wks + (long)c * taps;
    for (int kz = 0; kz < d.kd; ++kz) {
      const int iz = iz0 + kz;
      const bool vz = iz >= 0 && iz < d.D;
      for (int ky = 0; ky < d.kh; ++ky) {
        const int iy = iy0 + ky;
        const bool vy = vz && iy >= 0 && iy < d.H;
        const float* xr = xc + (long)iz * HW + (long)iy * d.W;
        const float* wr = wc + (kz * d.kh + ky) * d.kw;
        for (int kx = 0; kx < d.kw; ++kx) {
          const int ix = ix0 + kx;
          const float xv = (vy && ix >= 0 && ix < d.W) ? xr[ix] : 0.f;
#pragma unroll
          for (int j = 0; j < KT; ++j) acc[j] = fmaf(xv, wr[j * wks + kx], acc[j]);
        }
      }
    }
  }
  if (valid) {
    float* yn = y + ((long)n * d.K + k0) * So + pos;
#pragma unroll
    for (int j = 0; j < KT; ++j) yn[j * So] = acc[j];
  }
}

template <int CT>
__global__ __launch_bounds__(256) void k_conv_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                    float* __restrict__ dx, ConvDims d) {
  const long S = (long)d.D * d.H * d.W, So = (long)d.Do * d.Ho * d.Wo;
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * CT, n = blockIdx.z;
  const bool valid = pos < S;
  const long p = valid ? pos : 0;
  const int ix = (int)(p % d.W), iy = (int)((p / d.W) % d.H), iz = (int)(p / ((long)d.W * d.H));
  float acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = 0.f;
  const int taps = d.kd * d.kh * d.kw;
  const float* dyn = dy + (long)n * d.K * So;
  const float* wc = w + (long)c0 * taps;
  // taps outside, output channels inside: the stride divisions and bounds of a tap are evaluated once, not once per k
  // (the 64 -> 1 first PatchGAN layer, networks.py:1030, took 1.17 ms at Athena's batch with the loops the other way)
  for (int kz = 0; kz < d.kd; ++kz) {
    const int tz = iz + d.pd - kz;
    const int od = tz / d.sd;
    const bool vz = tz >= 0 && od * d.sd == tz && od < d.Do;
    for (int ky = 0; ky < d.kh; ++ky) {
      const int ty = iy + d.ph - ky;
      const int oh = ty / d.sh;
      const bool vy = vz && ty >= 0 && oh * d.sh == ty && oh < d.Ho;
      for (int kx = 0; kx < d.kw; ++kx) {
        const int tx = ix + d.pw - kx;
        const int ow = tx / d.sw;
        const bool v = vy && tx >= 0 && ow * d.sw == tx && ow < d.Wo;
        if (!v) continue;
        const float* dyp = dyn + ((long)od * d.Ho + oh) * d.Wo + ow;
        const float* wt = wc + (kz * d.kh + ky) * d.kw + kx;
        int k = 0;
        for (; k + 8 <= d.K; k += 8) {  // 8 loads in flight per lane: the loop is latency-bound otherwise
          float g[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) g[u] = dyp[(long)(k + u) * So];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[j] = fmaf(g[u], wt[((long)(k + u) * d.C + j) * taps], acc[j]);
        }
        for (; k < d.K; ++k) {
          const float g = dyp[(long)k * So];
#pragma unroll
          for (int j = 0; j < CT; ++j) acc[j] = fmaf(g, wt[((long)k * d.C + j) * taps], acc[j]);
        }
      }
    }
  }
  if (valid) {
    float* dxn = dx + ((long)n * d.C + c0) * S + pos;
#pragma unroll
    for (int j = 0; j < CT; ++j) dxn[j * S] = acc[j];
  }
}

// dw[k][c][t0..t0+TC) : one workgroup per (tap chunk, c, k); lanes stride over (n, output position).
template <int TC>
__global__ __launch_bounds__(256) void k_conv_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                    float* __restrict__ dw, ConvDims d) {
  const int t0 = blockIdx.x * TC, c = blockIdx.y, k = blockIdx.z;
  const int taps = d.kd * d.kh * d.kw;
  const long HW = (long)d.H * d.W, S = (long)d.D * HW, So = (long)d.Do * d.Ho * d.Wo;
  int tz[TC], ty[TC], tx[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int t = min(t0 + j, taps - 1);
    tx[j] = t % d.kw - d.pw;
    ty[j] = (t / d.kw) % d.kh - d.ph;
    tz[j] = t / (d.kw * d.kh) - d.pd;
  }
  float acc[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j) acc[j] = 0.f;
  for (int n = 0; n < d.N; ++n) {
    const float* xc = x + ((long)n * d.C + c) * S;
    const float* dyk = dy + ((long)n * d.K + k) * So;
    for (long pos = threadIdx.x; pos < So; pos += 256) {
      const float g = dyk[pos];
      const int ow = (int)(pos % d.Wo), oh = (int)((pos / d.Wo) % d.Ho), od = (int)(pos / ((long)d.Wo * d.Ho));
      const int bz = od * d.sd, by = oh * d.sh, bx = ow * d.sw;
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const int iz = bz + tz[j], iy = by + ty[j], ix = bx + tx[j];
        const bool v = iz >= 0 && iz < d.D && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
        const float xv = v ? xc[(long)iz * HW + (long)iy * d.W + ix] : 0.f;
        acc[j] = fmaf(g, xv, acc[j]);
      }
    }
  }
  __shared__ float red[4][TC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    float v = acc[j];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wv][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < TC && t0 + (int)threadIdx.x < taps) {
    const int j = threadIdx.x;
    dw[((long)k * d.C + c) * taps + t0 + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  }
}

// db[k] = sum over n and positions of dy[n][k][:]  -- two-stage, fixed order (deterministic): grid (splits * nsplit, K)
// partial sums in fp64 (a block = one range of positions x one range of samples), then one lane per k adds the partials.
__global__ __launch_bounds__(256) void k_bias_grad_part(const float* __restrict__ dy, double* __restrict__ part, int N,
                                                        int K, long S, int splits, int nsplit) {
  const int k = blockIdx.y, sp = blockIdx.x % splits, ns = blockIdx.x / splits;
  long chunk = (S + splits - 1) / splits;
  chunk = (chunk + 3) & ~3L;
  const long b = (long)sp * chunk, e = b + chunk < S ? b + chunk : S;
  const int nper = (N + nsplit - 1) / nsplit;
  const int n0 = ns * nper, n1 = n0 + nper < N ? n0 + nper : N;
  double acc = 0.0;
  for (int n = n0; n < n1; ++n) {
    const float* p = dy + ((long)n * K + k) * S;
    if ((S & 3) == 0 && ((uintptr_t)dy & 15) == 0) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      for (long i = b / 4 + threadIdx.x; i < e / 4; i += 256) {
        const float4 v = p4[i];
        acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      }
    } else {
      for (long i = b + threadIdx.x; i < e; i += 256) acc += (double)p[i];
    }
  }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)k * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void k_bias_grad_final(const double* __restrict__ part, float* __restrict__ db, int K, int splits) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double s = 0.0;
  for (int i = 0; i < splits; ++i) s += part[(long)k * splits + i];
  db[k] = (float)s;
}

static int pick_tile(int n) { return n % 8 == 0 ? 8 : n % 4 == 0 ? 4 : n % 2 == 0 ? 2 : 1; }

int conv_fwd_direct(const float* x, const float* w, const float* b, float* y, const ConvDims& d, hipStream_t s) {
  const long So = (long)d.Do * d.Ho * d.Wo;
  const int kt = pick_tile(d.K);
  dim3 grid((unsigned)cdiv(So, 256), d.K / kt, d.N);
  switch (kt) {
    case 8: hipLaunchKernelGGL(k_conv_fwd<8>, grid, dim3(256), 0, s, x, w, b, y, d); break;
    case 4: hipLaunchKernelGGL(k_conv_fwd<4>, grid, dim3(256), 0, s, x, w, b, y, d); break;
    case 2: hipLaunchKernelGGL(k_conv_fwd<2>, grid, dim3(256), 0, s, x, w, b, y, d); break;
    default: hipLaunchKernelGGL(k_conv_fwd<1>, grid, dim3(256), 0, s, x, w, b, y, d); break;
  }
  return check_launch("conv_fwd_direct");
}

int conv_dgrad_direct(const float* dy, const float* w, float* dx, const ConvDims& d, hipStream_t s) {
  const long S = (long)d.D * d.H * d.W;
  const int ct = pick_tile(d.C);
  dim3 grid((unsigned)cdiv(S, 256), d.C / ct, d.N);
  switch (ct) {
    case 8: hipLaunchKernelGGL(k_conv_dgrad<8>, grid, dim3(256), 0, s, dy, w, dx, d); break;
    case 4: hipLaunchKernelGGL(k_conv_dgrad<4>, grid, dim3(256), 0, s, dy, w, dx, d); break;
    case 2: hipLaunchKernelGGL(k_conv_dgrad<2>, grid, dim3(256), 0, s, dy, w, dx, d); break;
    default: hipLaunchKernelGGL(k_conv_dgrad<1>, grid, dim3(256), 0, s, dy, w, dx, d); break;
  }
  return check_launch("conv_dgrad_direct");
}

int conv_wgrad_direct(const float* x, const float* dy, float* dw, const ConvDims& d, hipStream_t s) {
  const int taps = d.kd * d.kh * d.kw;
  if (d.C > 65535 || d.K > 65535) {
    set_error("conv_wgrad_direct: channel count exceeds grid limits");
    return NC_ERR_SHAPE;
  }
  if (taps <= 1) {
    dim3 grid(1, d.C, d.K);
    hipLaunchKernelGGL(k_conv_wgrad<1>, grid, dim3(256), 0, s, x, dy, dw, d);
  } else {
    dim3 grid((unsigned)cdiv(taps, 16), d.C, d.K);
    hipLaunchKernelGGL(k_conv_wgrad<16>, grid, dim3(256), 0, s, x, dy, dw, d);
  }
  return check_launch("conv_wgrad_direct");
}

int bias_grad(const float* dy, float* db, int N, int K, long S, void* ws, size_t wsb, hipStream_t s) {
  long splits = cdiv(1024, K);
  const long cap = cdiv(S, 4096);
  if (splits > cap) splits = cap;
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  // batches of short planes (the 2-D PatchGAN layers: hundreds of samples): split the sample axis too
  long nsplit = cdiv(2048, (long)K * splits);
  if (nsplit > N) nsplit = N;
  while (nsplit > 1 && (long)K * splits * nsplit > 65536) --nsplit;
  if (nsplit < 1) nsplit = 1;
  const long parts = splits * nsplit;
  if (!ws || wsb < (size_t)K * parts * sizeof(double)) {
    set_error("bias_grad: workspace too small");
    return NC_ERR_WS;
  }
  hipLaunchKernelGGL(k_bias_grad_part, dim3((unsigned)parts, K), dim3(256), 0, s, dy, (double*)ws, N, K, S,
                     (int)splits, (int)nsplit);
  hipLaunchKernelGGL(k_bias_grad_final, dim3((unsigned)cdiv(K, 128)), dim3(128), 0, s, (const double*)ws, db, K,
                     (int)parts);
  return check_launch("bias_grad");
}

}  // namespace nc
