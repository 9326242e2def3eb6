// Small kernels around the generator step: slice / MIP (apollo_model.py:328-351), LSGAN-MSE and L1 losses
// (networks.py:276,299-313; apollo:128,279), Adam (apollo:131-136).  All HBM-bound and tiny next to the convolutions;
// written so that no step needs a host synchronisation (loss values and incoming gradients stay on the device).
#include "common.hpp"

namespace nc {

// ---------------------------------------------------------------- slice / MIP
// axis 0/1/2 = D/H/W.  Output plane dims: (H,W), (D,W), (D,H).
__device__ __forceinline__ void plane_dims(int D, int H, int W, int axis, int& A, int& B) {
  A = axis == 0 ? H : D;
  B = axis == 2 ? H : W;
}
__device__ __forceinline__ long vol_index(int D, int H, int W, int axis, int a, int b, int s) {
  // (a,b) plane coordinates, s coordinate along the axis
  if (axis == 0) return ((long)s * H + a) * W + b;
  if (axis == 1) return ((long)a * H + s) * W + b;
  return ((long)a * H + b) * W + s;
}

__global__ void k_slice_bwd(const float* __restrict__ dout, float* __restrict__ dvol, int NC, int D, int H, int W,
                            int axis, int index) {
  int A, B;
  plane_dims(D, H, W, axis, A, B);
  const long S = (long)D * H * W, total = (long)NC * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long nc = i / S, r = i % S;
    const int x = (int)(r % W), y = (int)((r / W) % H), z = (int)(r / ((long)W * H));
    const int s = axis == 0 ? z : axis == 1 ? y : x;
    const int a = axis == 0 ? y : z;
    const int b = axis == 2 ? y : x;
    dvol[i] = s == index ? dout[(nc * A + a) * B + b] : 0.f;
  }
}

__global__ void k_mip_fwd(const float* __restrict__ vol, float* __restrict__ out, int32_t* __restrict__ arg, int NC,
                          int D, int H, int W, int axis, int start, int depth) {
  int A, B;
  plane_dims(D, H, W, axis, A, B);
  const long total = (long)NC * A * B;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i % B), a = (int)((i / B) % A);
    const long nc = i / ((long)A * B);
    const float* p = vol + nc * D * H * W;
    float best = p[vol_index(D, H, W, axis, a, b, start)];
    int bi = start;
    for (int s = start + 1; s < start + depth; ++s) {
      const float v = p[vol_index(D, H, W, axis, a, b, s)];
      if (v > best || (v != v && best == best)) {
        best = v;
        bi = s;
      }
    }
    out[i] = best;
    if (arg) arg[i] = bi;
  }
}

__global__ void k_mip_bwd(const float* __restrict__ dout, const int32_t* __restrict__ arg, float* __restrict__ dvol,
                          int NC, int D, int H, int W, int axis) {
  int A, B;
  plane_dims(D, H, W, axis, A, B);
  const long S = (long)D * H * W, total = (long)NC * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long nc = i / S, r = i % S;
    const int x = (int)(r % W), y = (int)((r / W) % H), z = (int)(r / ((long)W * H));
    const int s = axis == 0 ? z : axis == 1 ? y : x;
    const int a = axis == 0 ? y : z;
    const int b = axis == 2 ? y : x;
    const long o = (nc * A + a) * B + b;
    dvol[i] = arg[o] == s ? dout[o] : 0.f;
  }
}

// ---------------------------------------------------------------- all slices along an axis, as a batch
// Athena's iter_f (axial_to_lateral_gan_athena_model.py:286-296) applies the 2-D discriminator to EVERY slice along
// an axis in a Python loop and stacks the outputs; InstanceNorm is per instance, so the same numbers come out of one
// batched call on slices[(n*L + s)][c][a][b] = vol[n][c][...axis index s...].  dir 0: vol -> slices, 1: slices -> vol
// (the backward: a pure permutation).
__global__ void k_slices(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int D, int H, int W,
                         int axis, int dir) {
  int A, B;
  plane_dims(D, H, W, axis, A, B);
  const int L = axis == 0 ? D : axis == 1 ? H : W;
  const long S = (long)D * H * W, total = (long)N * C * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i enumerates the slices tensor: [(n, s)][c][a][b]
    const int b = (int)(i % B), a = (int)((i / B) % A);
    const int c = (int)((i / ((long)A * B)) % C);
    const long ns = i / ((long)A * B * C);
    const int s = (int)(ns % L), n = (int)(ns / L);
    const long v = ((long)n * C + c) * S + vol_index(D, H, W, axis, a, b, s);
    if (dir == 0) dst[i] = src[v];
    else dst[v] = src[i];
  }
}

// ---------------------------------------------------------------- losses
// mode 0: (p - target)^2 ; mode 1: |a - b| ; mode 2: binary cross entropy of the logit p against the constant `target`
// (nn.BCEWithLogitsLoss, networks.py:278: max(p, 0) - p t + log(1 + exp(-|p|)), torch's stable form) ; mode 3: p itself (the
// 'wgangp' objective -+mean(p), networks.py:314-318; the sign is applied by the caller).  Two-stage deterministic reduction (fp64
// partials, fixed order).
__global__ __launch_bounds__(256) void k_loss_partial(const float* __restrict__ a, const float* __restrict__ b,
                                                      float target, int mode, long n, double* __restrict__ part) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (mode >= 2) {
      const float x = a[i];
      acc += mode == 3 ? (double)x : (double)(fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x))));
      continue;
    }
    const float d = mode == 0 ? a[i] - target : a[i] - b[i];
    acc += mode == 0 ? (double)d * (double)d : (double)fabsf(d);
  }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void k_loss_final(const double* __restrict__ part, int nb, long n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += part[i];
    out[0] = (float)(s / (double)n);
  }
}
__global__ void k_mse_bwd(const float* __restrict__ p, long n, float target, const float* __restrict__ gscale,
                          float* __restrict__ dp) {
  const float g = gscale[0] * (2.0f / (float)n);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dp[i] = (p[i] - target) * g;
}
// d/dp mean(BCEWithLogits(p, t)) = (sigmoid(p) - t) / n ; d/dp mean(p) = 1 / n  (mode 3; gscale carries the sign)
__global__ void k_logit_bwd(const float* __restrict__ p, long n, float target, int mode, const float* __restrict__ gscale,
                            float* __restrict__ dp) {
  const float g = gscale[0] / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dp[i] = mode == 3 ? g : (1.f / (1.f + expf(-p[i])) - target) * g;
}
__global__ void k_l1_bwd(const float* __restrict__ a, const float* __restrict__ b, long n,
                         const float* __restrict__ gscale, float* __restrict__ da) {
  const float g = gscale[0] / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    da[i] = d > 0.f ? g : d < 0.f ? -g : 0.f;
  }
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam, amsgrad=False, wd=0)
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long n, float b1, float b2, float eps, float step_size,
                       float bc2_sqrt) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float wl = 1.f - b1;                              // exp_avg.lerp_(grad, 1 - beta1), ATen's two-sided form
    const float mi = wl < 0.5f ? m[i] + wl * (gi - m[i]) : gi - (gi - m[i]) * (1.f - wl);
    const float vi = v[i] * b2 + gi * gi * (1.f - b2);     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

static unsigned flat_grid(long n) {
  long b = cdiv(n, 256);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}
static constexpr int kLossBlocks = 256;

}  // namespace nc

using namespace nc;

extern "C" {

static int vol_check(const char* what, int NC, int D, int H, int W, int axis) {
  if (NC < 1 || D < 1 || H < 1 || W < 1 || axis < 0 || axis > 2) {
    set_error("%s: bad shape/axis", what);
    return NC_ERR_SHAPE;
  }
  return NC_OK;
}
static int axis_len(int D, int H, int W, int axis) { return axis == 0 ? D : axis == 1 ? H : W; }

int nc_slice_fwd(const float* vol, float* out, int NC, int D, int H, int W, int axis, int index, void* stream) {
  if (!vol || !out) { set_error("slice_fwd: null pointer"); return NC_ERR_ARG; }
  if (int e = vol_check("slice_fwd", NC, D, H, W, axis)) return e;
  if (index < 0 || index >= axis_len(D, H, W, axis)) { set_error("slice_fwd: index out of range"); return NC_ERR_SHAPE; }
  // a slice is a depth-1 projection: one gather kernel serves both
  hipLaunchKernelGGL(k_mip_fwd, dim3(flat_grid((long)NC * D * H * W / axis_len(D, H, W, axis))), dim3(256), 0,
                     (hipStream_t)stream, vol, out, (int32_t*)nullptr, NC, D, H, W, axis, index, 1);
  return check_launch("slice_fwd");
}
int nc_slice_bwd(const float* dout, float* dvol, int NC, int D, int H, int W, int axis, int index, void* stream) {
  if (!dout || !dvol) { set_error("slice_bwd: null pointer"); return NC_ERR_ARG; }
  if (int e = vol_check("slice_bwd", NC, D, H, W, axis)) return e;
  if (index < 0 || index >= axis_len(D, H, W, axis)) { set_error("slice_bwd: index out of range"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_slice_bwd, dim3(flat_grid((long)NC * D * H * W)), dim3(256), 0, (hipStream_t)stream, dout, dvol,
                     NC, D, H, W, axis, index);
  return check_launch("slice_bwd");
}
int nc_mip_fwd(const float* vol, float* out, int32_t* arg, int NC, int D, int H, int W, int axis, int start, int depth,
               void* stream) {
  if (!vol || !out || !arg) { set_error("mip_fwd: null pointer"); return NC_ERR_ARG; }
  if (int e = vol_check("mip_fwd", NC, D, H, W, axis)) return e;
  if (start < 0 || depth < 1 || start + depth > axis_len(D, H, W, axis)) { set_error("mip_fwd: slab out of range"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_mip_fwd, dim3(flat_grid((long)NC * D * H * W / axis_len(D, H, W, axis))), dim3(256), 0,
                     (hipStream_t)stream, vol, out, arg, NC, D, H, W, axis, start, depth);
  return check_launch("mip_fwd");
}
int nc_mip_bwd(const float* dout, const int32_t* arg, float* dvol, int NC, int D, int H, int W, int axis,
               void* stream) {
  if (!dout || !arg || !dvol) { set_error("mip_bwd: null pointer"); return NC_ERR_ARG; }
  if (int e = vol_check("mip_bwd", NC, D, H, W, axis)) return e;
  hipLaunchKernelGGL(k_mip_bwd, dim3(flat_grid((long)NC * D * H * W)), dim3(256), 0, (hipStream_t)stream, dout, arg,
                     dvol, NC, D, H, W, axis);
  return check_launch("mip_bwd");
}

int nc_volume_slices(const float* src, float* dst, int N, int C, int D, int H, int W, int axis, int to_volume,
                     void* stream) {
  if (!src || !dst) { set_error("volume_slices: null pointer"); return NC_ERR_ARG; }
  if (N < 1 || C < 1) { set_error("volume_slices: bad shape"); return NC_ERR_SHAPE; }
  if (int e = vol_check("volume_slices", N * C, D, H, W, axis)) return e;
  hipLaunchKernelGGL(k_slices, dim3(flat_grid((long)N * C * D * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, N,
                     C, D, H, W, axis, to_volume ? 1 : 0);
  return check_launch("volume_slices");
}

size_t nc_loss_ws_bytes(long n) { (void)n; return kLossBlocks * sizeof(double); }

static int loss_fwd(const char* what, const float* a, const float* b, float target, int mode, long n, float* out,
                    void* ws, size_t ws_bytes, void* stream) {
  if (!a || !out || (mode == 1 && !b)) { set_error("%s: null pointer", what); return NC_ERR_ARG; }
  if (n < 1) { set_error("%s: empty input", what); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < nc_loss_ws_bytes(n)) { set_error("%s: workspace too small", what); return NC_ERR_WS; }
  long nb = cdiv(n, 1024);
  if (nb > kLossBlocks) nb = kLossBlocks;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_loss_partial, dim3((unsigned)nb), dim3(256), 0, s, a, b, target, mode, n, (double*)ws);
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, s, (const double*)ws, (int)nb, n, out);
  return check_launch(what);
}
int nc_mse_const_fwd(const float* pred, long n, float target, float* out, void* ws, size_t ws_bytes, void* stream) {
  return loss_fwd("mse_const_fwd", pred, nullptr, target, 0, n, out, ws, ws_bytes, stream);
}
int nc_bce_logits_const_fwd(const float* pred, long n, float target, float* out, void* ws, size_t ws_bytes, void* stream) {
  return loss_fwd("bce_logits_const_fwd", pred, nullptr, target, 2, n, out, ws, ws_bytes, stream);
}
int nc_mean_fwd(const float* pred, long n, float* out, void* ws, size_t ws_bytes, void* stream) {
  return loss_fwd("mean_fwd", pred, nullptr, 0.f, 3, n, out, ws, ws_bytes, stream);
}
int nc_bce_logits_const_bwd(const float* pred, long n, float target, const float* gscale, float* dpred, void* stream) {
  if (!pred || !gscale || !dpred) { set_error("bce_logits_const_bwd: null pointer"); return NC_ERR_ARG; }
  hipLaunchKernelGGL(k_logit_bwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, pred, n, target, 2, gscale, dpred);
  return check_launch("bce_logits_const_bwd");
}
int nc_mean_bwd(long n, const float* gscale, float* dpred, void* stream) {
  if (!gscale || !dpred) { set_error("mean_bwd: null pointer"); return NC_ERR_ARG; }
  hipLaunchKernelGGL(k_logit_bwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, n, 0.f, 3, gscale, dpred);
  return check_launch("mean_bwd");
}
int nc_l1_fwd(const float* a, const float* b, long n, float* out, void* ws, size_t ws_bytes, void* stream) {
  return loss_fwd("l1_fwd", a, b, 0.f, 1, n, out, ws, ws_bytes, stream);
}
int nc_mse_const_bwd(const float* pred, long n, float target, const float* gscale, float* dpred, void* stream) {
  if (!pred || !gscale || !dpred) { set_error("mse_const_bwd: null pointer"); return NC_ERR_ARG; }
  hipLaunchKernelGGL(k_mse_bwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, pred, n, target, gscale, dpred);
  return check_launch("mse_const_bwd");
}
int nc_l1_bwd(const float* a, const float* b, long n, const float* gscale, float* da, void* stream) {
  if (!a || !b || !gscale || !da) { set_error("l1_bwd: null pointer"); return NC_ERR_ARG; }
  hipLaunchKernelGGL(k_l1_bwd, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, gscale, da);
  return check_launch("l1_bwd");
}

int nc_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 int step, void* stream) {
  if (!p || !g || !m || !v) { set_error("adam_step: null pointer"); return NC_ERR_ARG; }
  if (n < 1 || step < 1) { set_error("adam_step: bad n/step"); return NC_ERR_SHAPE; }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(k_adam, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps,
                     step_size, bc2_sqrt);
  return check_launch("adam_step");
}

}  // extern "C"
