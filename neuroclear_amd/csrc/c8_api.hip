// extern "C" surface of the 16-bit end-to-end ("C8") operators: thin argument checks over c8_ops.hip / convt_h.hip /
// conv_h.hip.  The whole-network entry points (gen_nets_lp.hip) call the same internals directly.
#include "common.hpp"

namespace nc {
size_t c8_stats_ws_bytes(int N, int C, long S);
int c8_instnorm_stats(const void* x, int N, int C, long S, float eps, float* mean, float* rstd, int dt, void* ws, size_t wsb, hipStream_t s);
int c8_instnorm_apply(const void* x, const float* mean, const float* rstd, float slope, void* y, int ctot, int c0, int N, int C, long S, int dt, hipStream_t s);
int c8_instnorm_bwd(const void* g, int gctot, int gc0, const void* x, const float* mean, const float* rstd, float slope, void* dx, float* dbias, int N, int C, long S, int dt, void* ws, size_t wsb, hipStream_t s);
int c8_maxpool_fwd(const void* x, int ctot, int c0, void* y, int N, int C, int D, int H, int W, int dt, hipStream_t s);
int c8_maxpool_bwd_add(const void* dp, const void* x, int ctot, int c0, const void* skip, int sctot, int sc0, void* dx, int N, int C, int D, int H, int W, int dt, hipStream_t s);
int c8_to_f32(const void* x, int ctot, int c0, float* y, int N, int C, long S, int dt, hipStream_t s);
bool convT_h_supported(int C, int K);
size_t convT_h_ws_bytes(int N, int C, int D, int H, int W, int K);
int convT_fwd_h(const void* x, const float* w, const float* bias, void* out, int octot, int oc0, int N, int C, int D, int H, int W, int K, int dt, void* ws, size_t wsb, hipStream_t s);
int convT_dgrad_h(const void* dy, int dctot, int dc0, const float* w, void* dx, int N, int C, int D, int H, int W, int K, void* ws, size_t wsb, hipStream_t s);
int convT_wgrad_h(const void* x, const void* dy, int dctot, int dc0, float* dw, float* dbias, int N, int C, int D, int H, int W, int K, void* ws, size_t wsb, hipStream_t s);
}  // namespace nc

using namespace nc;

static bool dt_ok(int dt) { return dt == NC_DT_F16 || dt == NC_DT_BF16; }

extern "C" {

int nc_conv_fwd_c8(const void* xh, const float* w, const float* bias, void* yh, int out_ctot, int out_c0, int N, int C, int D, int H,
                   int W, int K, int kd, int kh, int kw, int stride, int pad, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!xh || !w || !yh) { set_error("conv_fwd_c8: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype)) { set_error("conv_fwd_c8: dtype must be NC_DT_F16 or NC_DT_BF16"); return NC_ERR_ARG; }
  ConvDims d;
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad) || !h_fwd_supported(d)) { set_error("conv_fwd_c8: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  if (out_ctot % 8 || out_c0 % 8 || out_c0 + K > out_ctot) { set_error("conv_fwd_c8: bad output channel range"); return NC_ERR_SHAPE; }
  return conv_fwd_h_c8(xh, w, bias, yh, out_ctot, out_c0, d, dtype, ws, ws_bytes, (hipStream_t)stream);
}

int nc_conv_dgrad_c8(const void* dyh, const float* w, void* dxh, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw,
                     int stride, int pad, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!dyh || !w || !dxh) { set_error("conv_dgrad_c8: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype)) { set_error("conv_dgrad_c8: dtype must be NC_DT_F16 or NC_DT_BF16"); return NC_ERR_ARG; }
  ConvDims d;
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad) || !h_dgrad_supported(d)) { set_error("conv_dgrad_c8: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  return conv_dgrad_h_c8(dyh, w, dxh, C, 0, d, dtype, ws, ws_bytes, (hipStream_t)stream);
}

size_t nc_conv_c1_c8_ws_bytes(int N, int D, int H, int W, int ks) { return c1_h_supported(D, H, W, ks) ? c1_h_ws_bytes(N, D, H, W, ks) : 0; }

int nc_conv_c1_fwd_c8(const float* x, const float* w, const float* bias, void* yh, int out_ctot, int out_c0, int N, int D, int H, int W,
                      int ks, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w || !yh) { set_error("conv_c1_fwd_c8: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype)) { set_error("conv_c1_fwd_c8: dtype must be NC_DT_F16 or NC_DT_BF16"); return NC_ERR_ARG; }
  if (N < 1 || !c1_h_supported(D, H, W, ks) || out_ctot % 8 || out_c0 % 8 || out_c0 + 64 > out_ctot) {
    set_error("conv_c1_fwd_c8: shape not covered (kernel 3 or 7, 64 output channels)");
    return NC_ERR_SHAPE;
  }
  return conv_c1_fwd_h(x, w, bias, yh, out_ctot, out_c0, N, D, H, W, ks, dtype, ws, ws_bytes, (hipStream_t)stream);
}

int nc_conv_c1_dgrad_c8(const void* dyh, const float* w, float* dx, int N, int D, int H, int W, int ks, void* ws, size_t ws_bytes,
                        void* stream) {
  if (!dyh || !w || !dx) { set_error("conv_c1_dgrad_c8: null pointer"); return NC_ERR_ARG; }
  if (N < 1 || !c1_h_supported(D, H, W, ks)) { set_error("conv_c1_dgrad_c8: shape not covered"); return NC_ERR_SHAPE; }
  return conv_c1_dgrad_h(dyh, w, dx, N, D, H, W, ks, ws, ws_bytes, (hipStream_t)stream);
}

size_t nc_conv_c1_wgrad_c8_ws_bytes(int N, int D, int H, int W, int ks) { return c1_wgrad_h_ws_bytes(N, D, H, W, ks); }

int nc_conv_c1_wgrad_c8(const float* x, const void* dyh, float* dw, int N, int D, int H, int W, int ks, void* ws, size_t ws_bytes,
                        void* stream) {
  if (!x || !dyh || !dw) { set_error("conv_c1_wgrad_c8: null pointer"); return NC_ERR_ARG; }
  if (!c1_wgrad_h_supported(N, D, H, W, ks)) { set_error("conv_c1_wgrad_c8: shape not covered (kernel 3 or 7, W <= 192)"); return NC_ERR_SHAPE; }
  return conv_c1_wgrad_h(x, dyh, dw, N, D, H, W, ks, ws, ws_bytes, (hipStream_t)stream);
}

size_t nc_c8_instnorm_ws_bytes(int N, int C, long S) { return (C % 8 || N < 1 || S < 1) ? 0 : c8_stats_ws_bytes(N, C, S); }

int nc_c8_instnorm_stats(const void* xh, int N, int C, long S, float eps, float* mean, float* rstd, int dtype, void* ws,
                         size_t ws_bytes, void* stream) {
  if (!xh || !mean || !rstd) { set_error("c8_instnorm_stats: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype)) { set_error("c8_instnorm_stats: bad dtype"); return NC_ERR_ARG; }
  return c8_instnorm_stats(xh, N, C, S, eps, mean, rstd, dtype, ws, ws_bytes, (hipStream_t)stream);
}

int nc_c8_instnorm_act_fwd(const void* xh, const float* mean, const float* rstd, float slope, void* yh, int out_ctot, int out_c0,
                           int N, int C, long S, int dtype, void* stream) {
  if (!xh || !mean || !rstd || !yh) { set_error("c8_instnorm_act_fwd: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype) || N < 1 || S < 1) { set_error("c8_instnorm_act_fwd: bad dtype / shape"); return NC_ERR_ARG; }
  return c8_instnorm_apply(xh, mean, rstd, slope, yh, out_ctot, out_c0, N, C, S, dtype, (hipStream_t)stream);
}

int nc_c8_instnorm_act_bwd(const void* gh, int g_ctot, int g_c0, const void* xh, const float* mean, const float* rstd, float slope,
                           void* dxh, float* dbias, int N, int C, long S, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!gh || !xh || !mean || !rstd || !dxh) { set_error("c8_instnorm_act_bwd: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype) || N < 1 || S < 1) { set_error("c8_instnorm_act_bwd: bad dtype / shape"); return NC_ERR_ARG; }
  return c8_instnorm_bwd(gh, g_ctot, g_c0, xh, mean, rstd, slope, dxh, dbias, N, C, S, dtype, ws, ws_bytes, (hipStream_t)stream);
}

int nc_c8_maxpool2_fwd(const void* xh, int x_ctot, int x_c0, void* yh, int N, int C, int D, int H, int W, int dtype, void* stream) {
  if (!xh || !yh) { set_error("c8_maxpool2_fwd: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype)) { set_error("c8_maxpool2_fwd: bad dtype"); return NC_ERR_ARG; }
  return c8_maxpool_fwd(xh, x_ctot, x_c0, yh, N, C, D, H, W, dtype, (hipStream_t)stream);
}

int nc_c8_maxpool2_bwd_add(const void* dph, const void* xh, int x_ctot, int x_c0, const void* skiph, int s_ctot, int s_c0, void* dxh,
                           int N, int C, int D, int H, int W, int dtype, void* stream) {
  if (!dph || !xh || !skiph || !dxh) { set_error("c8_maxpool2_bwd_add: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype)) { set_error("c8_maxpool2_bwd_add: bad dtype"); return NC_ERR_ARG; }
  return c8_maxpool_bwd_add(dph, xh, x_ctot, x_c0, skiph, s_ctot, s_c0, dxh, N, C, D, H, W, dtype, (hipStream_t)stream);
}

int nc_from_c8(const void* xh, int x_ctot, int x_c0, float* y, int N, int C, long S, int dtype, void* stream) {
  if (!xh || !y) { set_error("from_c8: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype) || N < 1 || S < 1) { set_error("from_c8: bad dtype / shape"); return NC_ERR_ARG; }
  return c8_to_f32(xh, x_ctot, x_c0, y, N, C, S, dtype, (hipStream_t)stream);
}

size_t nc_convT_c8_ws_bytes(int N, int C, int D, int H, int W, int K) { return convT_h_supported(C, K) ? convT_h_ws_bytes(N, C, D, H, W, K) : 0; }

int nc_convT_k2s2_fwd_c8(const void* xh, const float* w, const float* bias, void* yh, int out_ctot, int out_c0, int N, int C, int D,
                         int H, int W, int K, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!xh || !w || !yh) { set_error("convT_k2s2_fwd_c8: null pointer"); return NC_ERR_ARG; }
  if (!dt_ok(dtype)) { set_error("convT_k2s2_fwd_c8: bad dtype"); return NC_ERR_ARG; }
  return convT_fwd_h(xh, w, bias, yh, out_ctot, out_c0, N, C, D, H, W, K, dtype, ws, ws_bytes, (hipStream_t)stream);
}

int nc_convT_k2s2_dgrad_c8(const void* dyh, int dy_ctot, int dy_c0, const float* w, void* dxh, int N, int C, int D, int H, int W, int K,
                           void* ws, size_t ws_bytes, void* stream) {
  if (!dyh || !w || !dxh) { set_error("convT_k2s2_dgrad_c8: null pointer"); return NC_ERR_ARG; }
  return convT_dgrad_h(dyh, dy_ctot, dy_c0, w, dxh, N, C, D, H, W, K, ws, ws_bytes, (hipStream_t)stream);
}

int nc_convT_k2s2_wgrad_c8(const void* xh, const void* dyh, int dy_ctot, int dy_c0, float* dw, float* dbias, int N, int C, int D, int H,
                           int W, int K, void* ws, size_t ws_bytes, void* stream) {
  if (!xh || !dyh || !dw) { set_error("convT_k2s2_wgrad_c8: null pointer"); return NC_ERR_ARG; }
  return convT_wgrad_h(xh, dyh, dy_ctot, dy_c0, dw, dbias, N, C, D, H, W, K, ws, ws_bytes, (hipStream_t)stream);
}

// ---- fp32 3^3 convolutions as six bf16 MFMA products of a three-term operand split (conv_split.hip) -----------------------
int nc_conv_split_supported(int what, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad) {
  ConvDims d;
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return 0;
  return what == 0 ? s3_fwd_supported(d) : what == 1 ? s3_dgrad_supported(d) : what == 2 ? s3_wgrad_supported(d) : 0;
}

size_t nc_conv_split_ws_bytes(int N, int C, int D, int H, int W, int K, int ks) {
  ConvDims d;
  if (!make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2)) return 0;
  const size_t a = s3_ws_bytes(d), b = s3_wgrad_ws_bytes(d);
  return a > b ? a : b;
}

size_t nc_s3_bytes(int N, int C, long S) { return (C % 8 || N < 1 || S < 1) ? 0 : s3_tensor_bytes(N, C, S); }

int nc_to_s3(const float* x, void* xs, int N, int C, long S, void* stream) {
  if (!x || !xs) { set_error("to_s3: null pointer"); return NC_ERR_ARG; }
  if (N < 1 || C < 8 || C % 8 || S < 1) { set_error("to_s3: channels must be a multiple of 8"); return NC_ERR_SHAPE; }
  return split3_to(x, xs, N, C, S, (hipStream_t)stream);
}

int nc_conv_fwd_split(const float* x, const void* xs, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W,
                      int K, int ks, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if ((!x && !xs) || !w || !y) { set_error("conv_fwd_split: null pointer"); return NC_ERR_ARG; }
  if (!make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) || !s3_fwd_supported(d)) { set_error("conv_fwd_split: shape not covered"); return NC_ERR_SHAPE; }
  ProfScope ps(0, 1, d, 1, (hipStream_t)stream);
  ForceThreeTerm f3;
  return conv_fwd_s3(x, xs, w, bias, y, d, ws, ws_bytes, (hipStream_t)stream);
}

int nc_conv_dgrad_split(const float* dy, const void* dys, const float* w, float* dx, int N, int C, int D, int H, int W, int K, int ks,
                        void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if ((!dy && !dys) || !w || !dx) { set_error("conv_dgrad_split: null pointer"); return NC_ERR_ARG; }
  if (!make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) || !s3_dgrad_supported(d)) { set_error("conv_dgrad_split: shape not covered"); return NC_ERR_SHAPE; }
  ProfScope ps(1, 1, d, 1, (hipStream_t)stream);
  ForceThreeTerm f3;
  return conv_dgrad_s3(dy, dys, w, dx, d, ws, ws_bytes, (hipStream_t)stream);
}

int nc_conv_wgrad_split(const float* x, const void* xs, const float* dy, const void* dys, float* dw, int N, int C, int D, int H, int W,
                        int K, int ks, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if ((!x && !xs) || (!dy && !dys) || !dw) { set_error("conv_wgrad_split: null pointer"); return NC_ERR_ARG; }
  if (!make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) || !s3_wgrad_supported(d)) { set_error("conv_wgrad_split: shape not covered"); return NC_ERR_SHAPE; }
  ProfScope ps(2, 9, d, 0, (hipStream_t)stream);
  ForceThreeTerm f3;
  return conv_wgrad_s3(x, xs, dy, dys, dw, d, ws, ws_bytes, (hipStream_t)stream);
}

// ConvTranspose3d(k = 2, s = 2) forward on the same arithmetic (convt_s3.hip): xs = the input in S3 form, or NULL (then x is converted
// into the workspace); y (nullable) fp32 output, ys (nullable) channels [ys_c0, ys_c0 + K) of a ys_ctot-channel S3 tensor
static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
int nc_convT_k2s2_split_supported(int N, int C, int D, int H, int W, int K) { return convT_s3x_supported(N, C, D, H, W, K) ? 1 : 0; }
// ... and does the library take it by itself (nc_unet_deconv_fwd / _train_fwd; the layer-by-layer host path mirrors the choice)?
int nc_conv_lp_uses_c8x(int what, int fp32_out, int N, int C, int D, int H, int W, int K, int ks) {
  if (what == 0) return c8x_supported(N, C, D, H, W, K, ks, fp32_out != 0) ? 1 : 0;
  if (what == 1) return c8x_supported(N, K, D, H, W, C, ks, fp32_out != 0) ? 1 : 0;
  return 0;
}
void nc_set_c8x_mode(int mode) { c8x_set_mode(mode); }
int nc_get_c8x_mode(void) { return c8x_get_mode(); }
int nc_convT_k2s2_split_active(int N, int C, int D, int H, int W, int K) { return nc_get_conv_split() && convT_s3x_supported(N, C, D, H, W, K) ? 1 : 0; }
size_t nc_convT_k2s2_split_ws_bytes(int N, int C, int D, int H, int W, int K) {
  if (!convT_s3x_supported(N, C, D, H, W, K)) return 0;
  return al256(convT_s3x_ws_bytes(C, K)) + al256(s3_tensor_bytes(N, C, (long)D * H * W)) + 256;
}
int nc_convT_k2s2_fwd_split(const float* x, const void* xs, const float* w, const float* bias, float* y, void* ys, int ys_ctot, int ys_c0,
                            int N, int C, int D, int H, int W, int K, void* ws, size_t ws_bytes, void* stream) {
  if ((!x && !xs) || !w || (!y && !ys) || !ws) { set_error("convT_k2s2_fwd_split: null pointer"); return NC_ERR_ARG; }
  if (!convT_s3x_supported(N, C, D, H, W, K)) { set_error("convT_k2s2_fwd_split: shape not covered"); return NC_ERR_SHAPE; }
  const size_t wb = al256(convT_s3x_ws_bytes(C, K));
  const long S = (long)D * H * W;
  if (ws_bytes < wb + (xs ? 0 : al256(s3_tensor_bytes(N, C, S)))) { set_error("convT_k2s2_fwd_split: workspace too small"); return NC_ERR_WS; }
  if (!xs) {
    void* t = (char*)ws + wb;
    if (int e = split3_to(x, t, N, C, S, (hipStream_t)stream)) return e;
    xs = t;
  }
  return convT_fwd_s3x(xs, w, bias, y, ys, ys ? ys_ctot : 8, ys ? ys_c0 : 0, N, C, D, H, W, K, ws, wb, (hipStream_t)stream);
}

}  // extern "C"

namespace nc {
// ... in the two-term form, for the training forward (gen_nets.hip): x fp32 (an InstanceNorm + ReLU output: bound sqrt(S)) is converted into
// the workspace, y (fp32) and ys (channels [c0, c0 + K) of an H2 tensor whose cell `out_cell` the caller set with convT_h2_bound) are written
int convT_fwd_split_h2(const float* x, const float* w, const float* bias, float* y, void* ys, int ys_ctot, int ys_c0, int N, int C, int D, int H,
                       int W, int K, const unsigned* out_cell, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!x || !w || !ys || !out_cell || !ws) { set_error("convT_fwd_split_h2: null pointer"); return NC_ERR_ARG; }
  if (!convT_s3x_supported(N, C, D, H, W, K)) { set_error("convT_fwd_split_h2: shape not covered"); return NC_ERR_SHAPE; }
  const size_t wb = al256(convT_s3x_ws_bytes(C, K));
  const long S = (long)D * H * W;
  const size_t ex = (size_t)N * C * S;
  if (ws_bytes < wb + h2_cells_offset(ex) + 256) { set_error("convT_fwd_split_h2: workspace too small"); return NC_ERR_WS; }
  void* t = (char*)ws + wb;
  unsigned* xc = h2_cells_of(t, ex);
  if (int e = h2_set_cell(xc, sqrtf((float)S), s)) return e;
  if (int e = split2h_into(x, (long)C * S, t, N, C, S, C, 0, xc, s)) return e;
  return convT_fwd_s3x(t, w, bias, y, ys, ys_ctot, ys_c0, N, C, D, H, W, K, ws, wb, s, out_cell, xc);
}
}  // namespace nc
