// Spectral normalisation of a convolution weight (reference models/networks.py:1069-1111, NLayerDiscriminatorSN: every
// conv wrapped in torch.nn.utils.spectral_norm).  W = weight_orig viewed as [K][M] (M = C * taps).  Per forward pass in
// training mode ONE power iteration (torch's n_power_iterations = 1, eps = 1e-12):
//     v <- W^T u / max(|W^T u|, eps);   u <- W v / max(|W v|, eps);   sigma = u . (W v);   weight = W / sigma
// (u, v updated in place, treated as constants by the backward).  The matrices are small (<= 512 x 8192): one workgroup
// of 1024 threads does the whole thing in one launch -- two passes over W (L2-resident) plus the scaling pass.
// Backward of weight = W / (u^T W v):   dW = (G - <G, weight> u v^T) / sigma.
#include "common.hpp"

namespace nc {
namespace {

constexpr int kT = 1024;

__device__ __forceinline__ float block_sum(float a, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < kT / 64; ++i) t += red[i];
  return t;
}

__global__ void __launch_bounds__(kT) k_spectral_fwd(const float* __restrict__ W, float* __restrict__ u, float* __restrict__ v,
                                                     float* __restrict__ wout, float* __restrict__ sigma_out,
                                                     float* __restrict__ s /* [K] scratch */, int K, int M, int power_iter, float eps) {
  __shared__ float red[kT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (power_iter) {
    // t = W^T u: thread per column m (coalesced along m), then normalise into v
    float nrm = 0.f;
    for (int m = tid; m < M; m += kT) {
      float a = 0.f;
      for (int k = 0; k < K; ++k) a = fmaf(W[(long)k * M + m], u[k], a);
      v[m] = a;
      nrm = fmaf(a, a, nrm);
    }
    const float n2 = block_sum(nrm, red);
    const float inv = 1.f / fmaxf(sqrtf(n2), eps);
    for (int m = tid; m < M; m += kT) v[m] *= inv;
    __syncthreads();
  }
  // s = W v: one wave per row k (lanes along m), wave-reduce
  for (int k = wave; k < K; k += kT / 64) {
    float a = 0.f;
    for (int m = lane; m < M; m += 64) a = fmaf(W[(long)k * M + m], v[m], a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if (lane == 0) s[k] = a;
  }
  __syncthreads();
  if (power_iter) {
    float nrm = 0.f;
    for (int k = tid; k < K; k += kT) nrm = fmaf(s[k], s[k], nrm);
    const float n2 = block_sum(nrm, red);
    const float inv = 1.f / fmaxf(sqrtf(n2), eps);
    for (int k = tid; k < K; k += kT) u[k] = s[k] * inv;
    __syncthreads();
  }
  float d = 0.f;
  for (int k = tid; k < K; k += kT) d = fmaf(u[k], s[k], d);
  const float sigma = block_sum(d, red);
  if (tid == 0) sigma_out[0] = sigma;
  const float is = 1.f / sigma;
  const long total = (long)K * M;
  for (long i = tid; i < total; i += kT) wout[i] = W[i] / sigma;
  (void)is;
}

__global__ void __launch_bounds__(kT) k_spectral_bwd(const float* __restrict__ G, const float* __restrict__ wsn, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ sigma,
                                                     float* __restrict__ dW, int K, int M) {
  __shared__ float red[kT / 64];
  const int tid = threadIdx.x;
  const long total = (long)K * M;
  float a = 0.f;
  for (long i = tid; i < total; i += kT) a = fmaf(G[i], wsn[i], a);
  const float c = block_sum(a, red);
  const float sg = sigma[0];
  for (long i = tid; i < total; i += kT) {
    const int k = (int)(i / M), m = (int)(i - (long)k * M);
    dW[i] = (G[i] - c * u[k] * v[m]) / sg;
  }
}

}  // namespace
}  // namespace nc

using namespace nc;

extern "C" {

int nc_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* w_out, float* sigma, float* scratch_k, int K, int M,
                         int power_iteration, float eps, void* stream) {
  if (!w_orig || !u || !v || !w_out || !sigma || !scratch_k) { set_error("spectral_norm_fwd: null pointer"); return NC_ERR_ARG; }
  if (K < 1 || M < 1) { set_error("spectral_norm_fwd: bad shape"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_spectral_fwd, dim3(1), dim3(kT), 0, (hipStream_t)stream, w_orig, u, v, w_out, sigma, scratch_k, K, M,
                     power_iteration, eps);
  return check_launch("spectral_norm_fwd");
}

int nc_spectral_norm_bwd(const float* g, const float* w_sn, const float* u, const float* v, const float* sigma, float* dw_orig, int K,
                         int M, void* stream) {
  if (!g || !w_sn || !u || !v || !sigma || !dw_orig) { set_error("spectral_norm_bwd: null pointer"); return NC_ERR_ARG; }
  if (K < 1 || M < 1) { set_error("spectral_norm_bwd: bad shape"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_spectral_bwd, dim3(1), dim3(kT), 0, (hipStream_t)stream, g, w_sn, u, v, sigma, dw_orig, K, M);
  return check_launch("spectral_norm_bwd");
}

}  // extern "C"
