// 16-bit END-TO-END training passes of the two generators (BASELINE.json configs[3]: "fp16 MFMA path with fp32
// InstanceNorm accumulate"), one C call per direction:
//   Unet_deconv          reference models/networks.py:478-538
//   DeepLinearGenerator  reference models/networks.py:893-917
// Activations and their gradients live in HBM ONLY as 16-bit C8 tensors (conv_h.hip's operand layout): the convolution
// epilogue writes C8, InstanceNorm statistics (fp32 / fp64 accumulation) and normalise + ReLU read and write C8
// (c8_ops.hip), max-pool, the skip concats (written in place, no copy), the transposed convolutions (convt_h.hip) and
// the 64 -> 1 head work on C8 -- the fp32 NCDHW store + conversion passes of the layer-by-layer 16-bit path are gone
// (~8 GB instead of ~18 GB of HBM traffic per 64-channel full-resolution layer at 4 x 148^3).  Master weights, weight
// gradients, statistics, the one-channel first layers (1 -> 64: 3^3 of the U-Net, 7^3 of deep_linear_gen), losses and
// Adam stay fp32.  dtype: NC_DT_BF16 (operands and stored tensors bf16; gradients bf16).
//
// deep_linear_gen's pointwise tail 64 -> 32 -> 16 -> 1 has no bias and no activation: it IS a single 64-vector
// w_eff = W6 W5 W4, and every weight gradient of the three layers is an outer product with q = sum_v dy[v] f3[:, v]
// (one output channel => rank 1).  The tail is evaluated in that form (k_lin_tail_*): no 32- / 16-channel tensors exist.
#include <cstdlib>

#include "common.hpp"

namespace nc {
// c8_ops.hip
size_t c8_stats_ws_bytes(int N, int C, long S);
int c8_instnorm_stats(const void* x, int N, int C, long S, float eps, float* mean, float* rstd, int dt, void* ws, size_t wsb, hipStream_t s);
int c8_instnorm_apply(const void* x, const float* mean, const float* rstd, float slope, void* y, int ctot, int c0, int N, int C, long S, int dt, hipStream_t s);
int c8_instnorm_bwd(const void* g, int gctot, int gc0, const void* x, const float* mean, const float* rstd, float slope, void* dx, float* dbias, int N, int C, long S, int dt, void* ws, size_t wsb, hipStream_t s);
int c8_maxpool_fwd(const void* x, int ctot, int c0, void* y, int N, int C, int D, int H, int W, int dt, hipStream_t s);
int c8_maxpool_bwd_add(const void* dp, const void* x, int ctot, int c0, const void* skip, int sctot, int sc0, void* dx, int N, int C, int D, int H, int W, int dt, hipStream_t s);
int c8_to_f32(const void* x, int ctot, int c0, float* y, int N, int C, long S, int dt, hipStream_t s);
int c8_dot64(const void* x, const float* w, const float* bias, float* out, int N, long S, int dt, hipStream_t s);
size_t c8_outer64_ws_bytes(int N, long S);
int c8_outer64(const float* g, const void* x, const float* w, void* dx, float* dw, float* db, int N, long S, int dt, void* ws, size_t wsb, hipStream_t s);
// convt_h.hip
bool convT_h_supported(int C, int K);
size_t convT_h_ws_bytes(int N, int C, int D, int H, int W, int K);
int convT_fwd_h(const void* x, const float* w, const float* bias, void* out, int octot, int oc0, int N, int C, int D, int H, int W, int K, int dt, void* ws, size_t wsb, hipStream_t s);
int convT_dgrad_h(const void* dy, int dctot, int dc0, const float* w, void* dx, int N, int C, int D, int H, int W, int K, void* ws, size_t wsb, hipStream_t s);
int convT_wgrad_h(const void* x, const void* dy, int dctot, int dc0, float* dw, float* dbias, int N, int C, int D, int H, int W, int K, void* ws, size_t wsb, hipStream_t s);
}  // namespace nc

using namespace nc;

#define NC_TRY(expr) do { int e_ = (expr); if (e_) return e_; } while (0)

namespace {

size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

struct UBlock { int C, K, lvl; };
const UBlock kUB[10] = {{1, 64, 0},    {64, 64, 0},    {64, 128, 1},  {128, 128, 1}, {128, 256, 2},
                        {256, 256, 2}, {256, 256, 2}, {256, 128, 1}, {128, 128, 1}, {128, 64, 0}};

struct UOff { size_t w[14], b[14], total; };
UOff u_offsets() {
  struct L { int id; size_t wn, bn; };
  const L order[14] = {{0, 64 * 1 * 27, 64},     {1, 64 * 64 * 27, 64},    {2, 128 * 64 * 27, 128},  {3, 128 * 128 * 27, 128},
                       {4, 256 * 128 * 27, 256}, {5, 256 * 256 * 27, 256}, {6, 256 * 256 * 27, 256}, {10, 256 * 128 * 8, 128},
                       {7, 128 * 256 * 27, 128}, {8, 128 * 128 * 27, 128}, {11, 128 * 64 * 8, 64},   {9, 64 * 128 * 27, 64},
                       {12, 64, 1},              {13, 1, 1}};
  UOff o{};
  size_t off = 0;
  for (const L& l : order) {
    o.w[l.id] = off; off += l.wn;
    o.b[l.id] = off; off += l.bn;
  }
  o.total = off;
  return o;
}

// byte offsets into `saved` / the backward scratch
struct ULp {
  int N, d[3][3];
  long S[3];
  size_t raw0, a1, cat1, p1, a2, cat2, p2, b1, b2, b3, e2a, e2b, e1, t1, raw[10], mean[10], rstd[10], saved;
  size_t G1, G2, G3, H1, H2, H3, Q1, Q2, s1, s2, F1, F2, grads;
  size_t conv_ws, in_ws, c8_ws, convT_ws, o64_ws, f32conv_ws, c1_ws;
  bool ok, c1;  // c1: block 0 (1 -> 64 channels) on the 16-bit cores in pseudo-channel form (conv_h.hip)
};

bool ulp_plan(ULp& p, int N, int S0, int S1, int S2) {
  p = ULp{};
  if (N < 1 || S0 < 4 || S1 < 4 || S2 < 4 || (S0 & 3) || (S1 & 3) || (S2 & 3)) return false;
  p.N = N;
  for (int l = 0; l < 3; ++l) {
    p.d[l][0] = S0 >> l; p.d[l][1] = S1 >> l; p.d[l][2] = S2 >> l;
    p.S[l] = (long)p.d[l][0] * p.d[l][1] * p.d[l][2];
  }
  const size_t n = (size_t)N, S = (size_t)p.S[0], Sh = (size_t)p.S[1], Sq = (size_t)p.S[2];
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes); return r; };
  p.c1 = c1_h_supported(S0, S1, S2, 3);
  p.raw0 = take(n * 64 * S * (p.c1 ? 2 : 4));
  p.a1 = take(n * 64 * S * 2); p.cat1 = take(n * 128 * S * 2); p.p1 = take(n * 64 * Sh * 2); p.a2 = take(n * 128 * Sh * 2);
  p.cat2 = take(n * 256 * Sh * 2); p.p2 = take(n * 128 * Sq * 2); p.b1 = take(n * 256 * Sq * 2); p.b2 = take(n * 256 * Sq * 2);
  p.b3 = take(n * 256 * Sq * 2); p.e2a = take(n * 128 * Sh * 2); p.e2b = take(n * 128 * Sh * 2); p.e1 = take(n * 64 * S * 2);
  p.t1 = take(n * S * 4);
  for (int i = 0; i < 10; ++i) {
    p.raw[i] = i == 0 ? p.raw0 : take(n * kUB[i].K * (size_t)p.S[kUB[i].lvl] * 2);
    p.mean[i] = take(n * kUB[i].K * 4);
    p.rstd[i] = take(n * kUB[i].K * 4);
  }
  p.saved = off;
  off = 0;
  p.G1 = take(n * 64 * S * 2); p.G2 = take(n * 64 * S * 2); p.G3 = take(n * 128 * S * 2);
  p.H1 = take(n * 128 * Sh * 2); p.H2 = take(n * 128 * Sh * 2); p.H3 = take(n * 256 * Sh * 2);
  p.Q1 = take(n * 256 * Sq * 2); p.Q2 = take(n * 256 * Sq * 2); p.s1 = take(n * S * 4); p.s2 = take(n * S * 4);
  p.F1 = take(n * 64 * S * 4); p.F2 = take(n * 64 * S * 4);
  p.grads = off;
  p.ok = true;
  for (int i = 1; i < 10; ++i) {
    const int* d = p.d[kUB[i].lvl];
    ConvDims cd;
    if (!make_dims(cd, N, kUB[i].C, d[0], d[1], d[2], kUB[i].K, 3, 3, 3, 1, 1) || !h_fwd_supported(cd) || !h_dgrad_supported(cd) ||
        !h_wgrad_supported(cd)) { p.ok = false; return true; }
    const size_t b = h_ws_bytes(cd);
    if (b > p.conv_ws) p.conv_ws = b;
  }
  p.f32conv_ws = nc_conv_ws_bytes(N, 1, S0, S1, S2, 64, 3, 3, 3, 1, 1);
  const size_t t = nc_conv_ws_bytes(N, 1, S0, S1, S2, 1, 1, 1, 1, 1, 0);
  if (t > p.f32conv_ws) p.f32conv_ws = t;
  p.in_ws = nc_instnorm_bwd_dbias_ws_bytes(N * 64, (long)S);
  p.c8_ws = c8_stats_ws_bytes(N, 256, (long)S);
  p.convT_ws = convT_h_ws_bytes(N, 256, p.d[2][0], p.d[2][1], p.d[2][2], 128);
  const size_t c2 = convT_h_ws_bytes(N, 128, p.d[1][0], p.d[1][1], p.d[1][2], 64);
  if (c2 > p.convT_ws) p.convT_ws = c2;
  p.o64_ws = c8_outer64_ws_bytes(N, (long)S);
  p.c1_ws = p.c1 ? c1_h_ws_bytes(N, S0, S1, S2, 3) : 0;
  return true;
}

size_t ulp_ws_bytes(const ULp& p, bool bwd) {
  return al(p.conv_ws) + al(p.f32conv_ws) + al(p.in_ws) + al(p.c8_ws) + al(p.convT_ws) + al(p.o64_ws) + al(p.c1_ws) +
         (bwd ? p.grads : 0) + 256;
}

struct UWs { void *cws, *fws, *iws, *c8ws, *tws, *ows, *c1ws; char* G; };
UWs ulp_ws(const ULp& p, void* ws) {
  UWs u;
  char* b = (char*)ws;
  u.cws = b; b += al(p.conv_ws);
  u.fws = b; b += al(p.f32conv_ws);
  u.iws = b; b += al(p.in_ws);
  u.c8ws = b; b += al(p.c8_ws);
  u.tws = b; b += al(p.convT_ws);
  u.ows = b; b += al(p.o64_ws);
  u.c1ws = b; b += al(p.c1_ws);
  u.G = b;
  return u;
}

}  // namespace

namespace {
// NC_C1_WGRAD=0: the one-channel layers' weight gradient back on the fp32 tap-axis kernel (A/B runs)
bool c1_wgrad_on() {
  static const bool on = !(getenv("NC_C1_WGRAD") && atoi(getenv("NC_C1_WGRAD")) == 0);
  return on;
}
struct WeightDiffusion {
  WeightDiffusion() { nc::h_set_weight_diffusion(1); }
  ~WeightDiffusion() { nc::h_set_weight_diffusion(0); }
};
}  // namespace

extern "C" {

int nc_unet_deconv_lp_supported(int N, int S0, int S1, int S2, int dtype) {
  ULp p;
  return dtype == NC_DT_BF16 && ulp_plan(p, N, S0, S1, S2) && p.ok;
}

size_t nc_unet_deconv_lp_saved_bytes(int N, int S0, int S1, int S2) {
  ULp p;
  return ulp_plan(p, N, S0, S1, S2) && p.ok ? p.saved : 0;
}

size_t nc_unet_deconv_lp_ws_bytes(int N, int S0, int S1, int S2) {
  ULp p;
  return ulp_plan(p, N, S0, S1, S2) && p.ok ? ulp_ws_bytes(p, true) : 0;
}

int nc_unet_deconv_lp_fwd(const float* params, const float* x, float* y, void* saved, int N, int S0, int S1, int S2, int dtype,
                          void* ws, size_t ws_bytes, void* stream) {
  if (!params || !x || !y || !saved) { set_error("unet_deconv_lp_fwd: null pointer"); return NC_ERR_ARG; }
  if (dtype != NC_DT_BF16) { set_error("unet_deconv_lp_fwd: the 16-bit end-to-end path stores bf16 (dtype NC_DT_BF16)"); return NC_ERR_ARG; }
  ULp p;
  if (!ulp_plan(p, N, S0, S1, S2) || !p.ok) { set_error("unet_deconv_lp_fwd: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < ulp_ws_bytes(p, false)) { set_error("unet_deconv_lp_fwd: workspace too small"); return NC_ERR_WS; }
  const UOff o = u_offsets();
  const UWs u = ulp_ws(p, ws);
  hipStream_t hs = (hipStream_t)stream;
  char* V = (char*)saved;
  const float* P = params;
  const int dt = dtype;
  auto F = [&](size_t off) { return (float*)(V + off); };
  // block i >= 1: conv (C8 -> C8 raw), statistics, normalise + ReLU into channels [c0, c0 + K) of an out buffer of ctot
  auto block = [&](int i, const void* in, void* out, int ctot, int c0) -> int {
    const UBlock& b = kUB[i];
    const int* d = p.d[b.lvl];
    const long S = p.S[b.lvl];
    ConvDims cd;
    make_dims(cd, N, b.C, d[0], d[1], d[2], b.K, 3, 3, 3, 1, 1);
    NC_TRY(conv_fwd_h_c8(in, P + o.w[i], P + o.b[i], V + p.raw[i], b.K, 0, cd, dt, u.cws, p.conv_ws, hs));
    NC_TRY(c8_instnorm_stats(V + p.raw[i], N, b.K, S, 1e-5f, F(p.mean[i]), F(p.rstd[i]), dt, u.c8ws, p.c8_ws, hs));
    return c8_instnorm_apply(V + p.raw[i], F(p.mean[i]), F(p.rstd[i]), 0.f, out, ctot, c0, N, b.K, S, dt, hs);
  };
  const long S = p.S[0];
  const int *d0 = p.d[0], *d1 = p.d[1], *d2 = p.d[2];
  if (p.c1) {
    // block 0 (1 -> 64 channels) in pseudo-channel form on the 16-bit cores: raw output, statistics and activation as C8
    NC_TRY(conv_c1_fwd_h(x, P + o.w[0], P + o.b[0], V + p.raw0, 64, 0, N, d0[0], d0[1], d0[2], 3, dt, u.c1ws, p.c1_ws, hs));
    NC_TRY(c8_instnorm_stats(V + p.raw0, N, 64, S, 1e-5f, F(p.mean[0]), F(p.rstd[0]), dt, u.c8ws, p.c8_ws, hs));
    NC_TRY(c8_instnorm_apply(V + p.raw0, F(p.mean[0]), F(p.rstd[0]), 0.f, V + p.a1, 64, 0, N, 64, S, dt, hs));
  } else {
    // (shapes the 16-bit one-channel form does not cover) the fp32 one-channel kernel; the activation leaves as C8
    NC_TRY(nc_conv_fwd(x, P + o.w[0], P + o.b[0], F(p.raw0), N, 1, d0[0], d0[1], d0[2], 64, 3, 3, 3, 1, 1, u.fws, p.f32conv_ws, stream));
    NC_TRY(nc_instnorm_stats(F(p.raw0), N * 64, S, 1e-5f, F(p.mean[0]), F(p.rstd[0]), u.iws, p.in_ws, stream));
    NC_TRY(nc_instnorm_act_fwd_c8(F(p.raw0), F(p.mean[0]), F(p.rstd[0]), 0.f, nullptr, V + p.a1, N, 64, S, dt, stream));
  }
  NC_TRY(block(1, V + p.a1, V + p.cat1, 128, 0));
  NC_TRY(c8_maxpool_fwd(V + p.cat1, 128, 0, V + p.p1, N, 64, d0[0], d0[1], d0[2], dt, hs));
  NC_TRY(block(2, V + p.p1, V + p.a2, 128, 0));
  NC_TRY(block(3, V + p.a2, V + p.cat2, 256, 0));
  NC_TRY(c8_maxpool_fwd(V + p.cat2, 256, 0, V + p.p2, N, 128, d1[0], d1[1], d1[2], dt, hs));
  NC_TRY(block(4, V + p.p2, V + p.b1, 256, 0));
  NC_TRY(block(5, V + p.b1, V + p.b2, 256, 0));
  NC_TRY(block(6, V + p.b2, V + p.b3, 256, 0));
  NC_TRY(convT_fwd_h(V + p.b3, P + o.w[10], P + o.b[10], V + p.cat2, 256, 128, N, 256, d2[0], d2[1], d2[2], 128, dt, u.tws, p.convT_ws, hs));
  NC_TRY(block(7, V + p.cat2, V + p.e2a, 128, 0));
  NC_TRY(block(8, V + p.e2a, V + p.e2b, 128, 0));
  NC_TRY(convT_fwd_h(V + p.e2b, P + o.w[11], P + o.b[11], V + p.cat1, 128, 64, N, 128, d1[0], d1[1], d1[2], 64, dt, u.tws, p.convT_ws, hs));
  NC_TRY(block(9, V + p.cat1, V + p.e1, 64, 0));
  // head: one_by_one (64 -> 1) on C8, one_by_one_2 (1 -> 1) + sigmoid in fp32
  NC_TRY(c8_dot64(V + p.e1, P + o.w[12], P + o.b[12], F(p.t1), N, S, dt, hs));
  NC_TRY(nc_conv_fwd(F(p.t1), P + o.w[13], P + o.b[13], y, N, 1, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, u.fws, p.f32conv_ws, stream));
  return nc_sigmoid_fwd(y, y, (long)N * S, stream);
}

int nc_unet_deconv_lp_bwd(const float* params, const float* x, const float* y, const void* saved, const float* dy, float* dx,
                          float* dparams, int N, int S0, int S1, int S2, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (!params || !x || !y || !saved || !dy || !dparams) { set_error("unet_deconv_lp_bwd: null pointer"); return NC_ERR_ARG; }
  if (dtype != NC_DT_BF16) { set_error("unet_deconv_lp_bwd: dtype must be NC_DT_BF16"); return NC_ERR_ARG; }
  ULp p;
  if (!ulp_plan(p, N, S0, S1, S2) || !p.ok) { set_error("unet_deconv_lp_bwd: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < ulp_ws_bytes(p, true)) { set_error("unet_deconv_lp_bwd: workspace too small"); return NC_ERR_WS; }
  const UOff o = u_offsets();
  const UWs u = ulp_ws(p, ws);
  hipStream_t hs = (hipStream_t)stream;
  const char* V = (const char*)saved;
  char* G = u.G;
  const float* P = params;
  float* DP = dparams;
  const int dt = dtype;
  auto F = [&](size_t off) { return (const float*)(V + off); };
  const long S = p.S[0];
  const int *d0 = p.d[0], *d1 = p.d[1], *d2 = p.d[2];
  // block i >= 1 backward: g = gradient at the block's output = channels [gc0, ..) of a gctot buffer; `in` = the block's
  // (dense C8) input; draw <- InstanceNorm / ReLU backward (+ bias gradient); gin (nullable, dense C channels) <- dgrad
  auto block_bwd = [&](int i, const void* g, int gctot, int gc0, const void* in, void* draw, void* gin) -> int {
    const UBlock& b = kUB[i];
    const int* d = p.d[b.lvl];
    const long Sl = p.S[b.lvl];
    ConvDims cd;
    make_dims(cd, N, b.C, d[0], d[1], d[2], b.K, 3, 3, 3, 1, 1);
    NC_TRY(c8_instnorm_bwd(g, gctot, gc0, V + p.raw[i], F(p.mean[i]), F(p.rstd[i]), 0.f, draw, DP + o.b[i], N, b.K, Sl, dt, u.c8ws,
                           p.c8_ws, hs));
    if (gin) NC_TRY(conv_dgrad_h_c8(draw, P + o.w[i], gin, b.C, 0, cd, NC_DT_BF16, u.cws, p.conv_ws, hs));
    ProfScope ps(2, 1, cd, 1, hs);
    return conv_wgrad_h(nullptr, in, nullptr, draw, DP + o.w[i], cd, NC_DT_BF16, u.cws, p.conv_ws, hs);
  };
  // head
  float* s1 = (float*)(G + p.s1);
  float* s2 = (float*)(G + p.s2);
  NC_TRY(nc_sigmoid_bwd(dy, y, s1, (long)N * S, stream));
  NC_TRY(nc_conv_wgrad(F(p.t1), s1, DP + o.w[13], DP + o.b[13], N, 1, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, u.fws, p.f32conv_ws, stream));
  NC_TRY(nc_conv_dgrad(s1, P + o.w[13], s2, N, 1, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, u.fws, p.f32conv_ws, stream));
  NC_TRY(c8_outer64(s2, V + p.e1, P + o.w[12], G + p.G1, DP + o.w[12], DP + o.b[12], N, S, dt, u.ows, p.o64_ws, hs));
  // ex_conv1_1
  NC_TRY(block_bwd(9, G + p.G1, 64, 0, V + p.cat1, G + p.G2, G + p.G3));
  // t_conv1: its output is the second half of cat1, so its dy is channels [64, 128) of dcat1 (read in place)
  NC_TRY(convT_dgrad_h(G + p.G3, 128, 64, P + o.w[11], G + p.H1, N, 128, d1[0], d1[1], d1[2], 64, u.tws, p.convT_ws, hs));
  NC_TRY(convT_wgrad_h(V + p.e2b, G + p.G3, 128, 64, DP + o.w[11], DP + o.b[11], N, 128, d1[0], d1[1], d1[2], 64, u.tws, p.convT_ws, hs));
  NC_TRY(block_bwd(8, G + p.H1, 128, 0, V + p.e2a, G + p.H2, G + p.H1));
  NC_TRY(block_bwd(7, G + p.H1, 128, 0, V + p.cat2, G + p.H2, G + p.H3));
  NC_TRY(convT_dgrad_h(G + p.H3, 256, 128, P + o.w[10], G + p.Q1, N, 256, d2[0], d2[1], d2[2], 128, u.tws, p.convT_ws, hs));
  NC_TRY(convT_wgrad_h(V + p.b3, G + p.H3, 256, 128, DP + o.w[10], DP + o.b[10], N, 256, d2[0], d2[1], d2[2], 128, u.tws, p.convT_ws, hs));
  NC_TRY(block_bwd(6, G + p.Q1, 256, 0, V + p.b2, G + p.Q2, G + p.Q1));
  NC_TRY(block_bwd(5, G + p.Q1, 256, 0, V + p.b1, G + p.Q2, G + p.Q1));
  NC_TRY(block_bwd(4, G + p.Q1, 256, 0, V + p.p2, G + p.Q2, G + p.Q1));
  // conv2 = cat2[:, :128] feeds the pool AND the skip: pool backward + first half of dcat2, in one pass
  NC_TRY(c8_maxpool_bwd_add(G + p.Q1, V + p.cat2, 256, 0, G + p.H3, 256, 0, G + p.H1, N, 128, d1[0], d1[1], d1[2], dt, hs));
  NC_TRY(block_bwd(3, G + p.H1, 128, 0, V + p.a2, G + p.H2, G + p.H1));
  NC_TRY(block_bwd(2, G + p.H1, 128, 0, V + p.p1, G + p.H2, G + p.H1));
  NC_TRY(c8_maxpool_bwd_add(G + p.H1, V + p.cat1, 128, 0, G + p.G3, 128, 0, G + p.G1, N, 64, d0[0], d0[1], d0[2], dt, hs));
  NC_TRY(block_bwd(1, G + p.G1, 64, 0, V + p.a1, G + p.G2, G + p.G1));
  float* f1 = (float*)(G + p.F1);
  float* f2 = (float*)(G + p.F2);
  if (p.c1) {
    // block 0: InstanceNorm backward on C8; the data gradient (if wanted) in pseudo-channel form; the weight gradient
    // of the one-channel layer on the fp32 tap-axis kernel, fed with the fp32 copy of the C8 gradient
    NC_TRY(c8_instnorm_bwd(G + p.G1, 64, 0, V + p.raw0, F(p.mean[0]), F(p.rstd[0]), 0.f, G + p.G2, DP + o.b[0], N, 64, S, dt, u.c8ws,
                           p.c8_ws, hs));
    if (dx) NC_TRY(conv_c1_dgrad_h(G + p.G2, P + o.w[0], dx, N, d0[0], d0[1], d0[2], 3, u.c1ws, p.c1_ws, hs));
    {
      // weight gradient on the 16-bit cores straight from the C8 gradient (c1_wgrad_h.hip); its planar copies live in the
      // fp32 gradient buffers F1 + F2, which this path does not use
      const size_t fbytes = (size_t)2 * N * 64 * S * 4, need = c1_wgrad_h_ws_bytes(N, d0[0], d0[1], d0[2], 3);
      if (c1_wgrad_on() && need > 0 && need <= fbytes)
        return conv_c1_wgrad_h(x, G + p.G2, DP + o.w[0], N, d0[0], d0[1], d0[2], 3, f1, fbytes, hs);
    }
    NC_TRY(c8_to_f32(G + p.G2, 64, 0, f2, N, 64, S, NC_DT_BF16, hs));
    return nc_conv_wgrad(x, f2, DP + o.w[0], nullptr, N, 1, d0[0], d0[1], d0[2], 64, 3, 3, 3, 1, 1, u.fws, p.f32conv_ws, stream);
  }
  // block 0 in fp32 (one input channel): gradient C8 -> fp32, then the fp32 kernels
  NC_TRY(c8_to_f32(G + p.G1, 64, 0, f1, N, 64, S, NC_DT_BF16, hs));
  NC_TRY(nc_instnorm_act_bwd_dbias(f1, F(p.raw0), F(p.mean[0]), F(p.rstd[0]), 0.f, f2, DP + o.b[0], N, 64, S, u.iws, p.in_ws, stream));
  if (dx) NC_TRY(nc_conv_dgrad(f2, P + o.w[0], dx, N, 1, d0[0], d0[1], d0[2], 64, 3, 3, 3, 1, 1, u.fws, p.f32conv_ws, stream));
  return nc_conv_wgrad(x, f2, DP + o.w[0], nullptr, N, 1, d0[0], d0[1], d0[2], 64, 3, 3, 3, 1, 1, u.fws, p.f32conv_ws, stream);
}

}  // extern "C"

// ---- deep_linear_gen ---------------------------------------------------------------------------------------------------
namespace {

// w_eff = W6 W5 W4 (64), u1 = W6 W5 (32): one block of 64 threads
__global__ void k_lin_tail_prep(const float* __restrict__ W4, const float* __restrict__ W5, const float* __restrict__ W6,
                                float* __restrict__ weff, float* __restrict__ u1) {
  __shared__ float su[32];
  const int t = threadIdx.x;
  if (t < 32) {
    float a = 0.f;
    for (int i = 0; i < 16; ++i) a = fmaf(W6[i], W5[i * 32 + t], a);
    su[t] = a;
    u1[t] = a;
  }
  __syncthreads();
  float a = 0.f;
  for (int j = 0; j < 32; ++j) a = fmaf(su[j], W4[j * 64 + t], a);
  weff[t] = a;
}

// weight gradients of the three pointwise layers from q[c] = sum_v dy[v] f3[c][v]:
//   dW4[j][c] = u1[j] q[c];  p1 = W4 q;  dW5[i][j] = W6[i] p1[j];  dW6[i] = (W5 p1)[i]
__global__ void k_lin_tail_wgrad(const float* __restrict__ W4, const float* __restrict__ W5, const float* __restrict__ W6,
                                 const float* __restrict__ u1, const float* __restrict__ q, float* __restrict__ dW4,
                                 float* __restrict__ dW5, float* __restrict__ dW6) {
  __shared__ float p1[32];
  const int t = threadIdx.x;  // 256 threads
  if (t < 32) {
    float a = 0.f;
    for (int c = 0; c < 64; ++c) a = fmaf(W4[t * 64 + c], q[c], a);
    p1[t] = a;
  }
  __syncthreads();
  for (int i = t; i < 32 * 64; i += 256) dW4[i] = u1[i >> 6] * q[i & 63];
  for (int i = t; i < 16 * 32; i += 256) dW5[i] = W6[i >> 5] * p1[i & 31];
  if (t < 16) {
    float a = 0.f;
    for (int j = 0; j < 32; ++j) a = fmaf(W5[t * 32 + j], p1[j], a);
    dW6[t] = a;
  }
}

// Dsh[a][v] = dy[v - (a - 1)] inside the volume (27 shifts of the one-channel dy, padded to 32 channels) as a bf16 C8 tensor: the x operand of
// layer 1's weight gradient in its rank-structured form (gen_nets.hip, "layer 1's weight gradient from the rank structure of its dY")
__global__ void __launch_bounds__(256) k_dl_shift27_c8(const float* __restrict__ dy, uint4* __restrict__ dsh, int D, int H, int W) {
  const long S = (long)D * H * W;
  const int q = blockIdx.y, n = blockIdx.z;  // q: chunk of 8 shifts
  const float* in = dy + (long)n * S;
  uint4* out = dsh + ((long)n * 4 + q) * S;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < S; v += (long)gridDim.x * 256) {
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long)W * H));
    unsigned short h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int a = q * 8 + j;
      float r = 0.f;
      if (a < 27) {
        const int sx = x - (a % 3 - 1), sy = y - ((a / 3) % 3 - 1), sz = z - (a / 9 - 1);
        if ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H && (unsigned)sz < (unsigned)D) r = in[((long)sz * H + sy) * W + sx];
      }
      const __bf16 b = (__bf16)r;
      h[j] = __builtin_bit_cast(unsigned short, b);
    }
    uint4 o;
    o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16); o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
    out[v] = o;
  }
}

struct LLp {
  int N, d[3];
  long S;
  size_t w[6];
  size_t f1h, f2h, f3h, weff, u1, saved;        // bytes into saved
  size_t F, A, B, q, tail, grads;                // bytes into the scratch
  size_t conv_ws, f32conv_ws, o64_ws, c1_ws;
  bool ok, c1;
  bool c3;  // the 3^3 one-channel kernels cover the shape too: layers 2 .. 5 can run in collapsed form (gen_nets.hip, "the collapsed tail")
};

bool llp_plan(LLp& p, int N, int S0, int S1, int S2) {
  p = LLp{};
  if (N < 1 || S0 < 1 || S1 < 1 || S2 < 1) return false;
  p.N = N; p.d[0] = S0; p.d[1] = S1; p.d[2] = S2;
  p.S = (long)S0 * S1 * S2;
  const int ks[6] = {7, 5, 3, 1, 1, 1}, C[6] = {1, 64, 64, 64, 32, 16}, K[6] = {64, 64, 64, 32, 16, 1};
  size_t off = 0;
  for (int i = 0; i < 6; ++i) { p.w[i] = off; off += (size_t)K[i] * C[i] * ks[i] * ks[i] * ks[i]; }
  const size_t n = (size_t)N, S = (size_t)p.S;
  off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes); return r; };
  p.f1h = take(n * 64 * S * 2); p.f2h = take(n * 64 * S * 2); p.f3h = take(n * 64 * S * 2); p.weff = take(64 * 4); p.u1 = take(32 * 4);
  p.saved = off;
  off = 0;
  p.F = take(n * 64 * S * 4); p.A = take(n * 64 * S * 2); p.B = take(n * 64 * S * 2); p.q = take(65 * 4); p.tail = take(dl_tail_bytes());
  p.grads = off;
  p.ok = true;
  for (int i = 1; i <= 2; ++i) {
    ConvDims cd;
    if (!make_dims(cd, N, 64, S0, S1, S2, 64, ks[i], ks[i], ks[i], 1, ks[i] / 2) || !h_fwd_supported(cd) || !h_dgrad_supported(cd) ||
        !h_wgrad_supported(cd)) { p.ok = false; return true; }
    const size_t b = h_ws_bytes(cd);
    if (b > p.conv_ws) p.conv_ws = b;
  }
  p.f32conv_ws = nc_conv_ws_bytes(N, 1, S0, S1, S2, 64, 7, 7, 7, 1, 3);
  p.o64_ws = c8_outer64_ws_bytes(N, p.S);
  p.c1 = c1_h_supported(S0, S1, S2, 7);
  p.c1_ws = p.c1 ? c1_h_ws_bytes(N, S0, S1, S2, 7) : 0;
  p.c3 = p.c1 && c1_h_supported(S0, S1, S2, 3) && c1_wgrad_h_supported(N, S0, S1, S2, 3) &&
         c1_wgrad_h_ws_bytes(N, S0, S1, S2, 3) <= (size_t)N * 64 * p.S * 4;
  if (p.c3 && c1_h_ws_bytes(N, S0, S1, S2, 3) > p.c1_ws) p.c1_ws = c1_h_ws_bytes(N, S0, S1, S2, 3);
  return true;
}

// the rank-structured backward of the 5^3 layer (gen_nets.hip) is available at this shape: 32-channel shapes of the weight-gradient and
// forward kernels fit the plan's workspace
bool llp_rank_ok(const LLp& p, ConvDims& dsh) {
  static const bool rank_on = !(getenv("NC_DL_RANK_WGRAD") && atoi(getenv("NC_DL_RANK_WGRAD")) == 0);
  return rank_on && p.c1 && p.c3 && make_dims(dsh, p.N, 32, p.d[0], p.d[1], p.d[2], 64, 5, 5, 5, 1, 2) && c8x_wgrad_supported(dsh) &&
         c8x_wgrad_part_bytes(dsh) <= p.conv_ws && h_fwd_supported(dsh) && h_ws_bytes(dsh) <= p.conv_ws;
}

size_t llp_ws_bytes(const LLp& p) { return al(p.conv_ws) + al(p.f32conv_ws) + al(p.o64_ws) + al(p.c1_ws) + p.grads + 256; }

}  // namespace

extern "C" {

int nc_deep_linear_lp_supported(int N, int S0, int S1, int S2, int dtype) {
  LLp p;
  return dtype == NC_DT_BF16 && llp_plan(p, N, S0, S1, S2) && p.ok;
}
size_t nc_deep_linear_lp_saved_bytes(int N, int S0, int S1, int S2) {
  LLp p;
  return llp_plan(p, N, S0, S1, S2) && p.ok ? p.saved : 0;
}
size_t nc_deep_linear_lp_ws_bytes(int N, int S0, int S1, int S2) {
  LLp p;
  return llp_plan(p, N, S0, S1, S2) && p.ok ? llp_ws_bytes(p) : 0;
}

// `kept` (out, as on the fp32 path): which FORM this forward took -- it depends on the process-wide switches at forward time (nc_set_dl_collapse,
// the one-channel weight-gradient switch) and decides what `saved` holds: kLpLayered (f1, f2, f3, w_eff, u1), kLpCollapsed (f1, f2) or
// kLpNoF2 (f1 only).  The backward follows `kept`, not the switches of its own moment.
enum : unsigned { kLpLayered = 0, kLpCollapsed = 1, kLpNoF2 = 2 };

int nc_deep_linear_lp_fwd(const float* params, const float* x, float* y, void* saved, int N, int S0, int S1, int S2, int dtype,
                          void* ws, size_t ws_bytes, void* stream, unsigned* kept) {
  if (!params || !x || !y || !saved || !kept) { set_error("deep_linear_lp_fwd: null pointer"); return NC_ERR_ARG; }
  *kept = kLpLayered;
  if (dtype != NC_DT_BF16) { set_error("deep_linear_lp_fwd: dtype must be NC_DT_BF16"); return NC_ERR_ARG; }
  LLp p;
  if (!llp_plan(p, N, S0, S1, S2) || !p.ok) { set_error("deep_linear_lp_fwd: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < llp_ws_bytes(p)) { set_error("deep_linear_lp_fwd: workspace too small"); return NC_ERR_WS; }
  const WeightDiffusion wd_;  // the bias-free, norm-free stack rounds its weights tap-diffused (conv_h.hip)
  hipStream_t hs = (hipStream_t)stream;
  char* cws = (char*)ws;
  char* fws = cws + al(p.conv_ws);
  char* c1ws = fws + al(p.f32conv_ws) + al(p.o64_ws);
  char* G = c1ws + al(p.c1_ws);
  char* V = (char*)saved;
  float* Ff = (float*)(G + p.F);
  if (p.c1) {
    // 7^3, one input channel, in pseudo-channel form on the 16-bit cores: straight to C8
    NC_TRY(conv_c1_fwd_h(x, params + p.w[0], nullptr, V + p.f1h, 64, 0, N, S0, S1, S2, 7, dtype, c1ws, p.c1_ws, hs));
  } else {
    // fp32 one-channel kernel, then the 64-channel result becomes C8
    NC_TRY(nc_conv_fwd(x, params + p.w[0], nullptr, Ff, N, 1, S0, S1, S2, 64, 7, 7, 7, 1, 3, fws, p.f32conv_ws, stream));
    NC_TRY(to_c8(Ff, V + p.f1h, N, 64, p.S, dtype, hs));
  }
  ConvDims c5, c3;
  make_dims(c5, N, 64, S0, S1, S2, 64, 5, 5, 5, 1, 2);
  make_dims(c3, N, 64, S0, S1, S2, 64, 3, 3, 3, 1, 1);
  ConvDims dshf;
  static const bool k32_on = !(getenv("NC_DL_K32") && atoi(getenv("NC_DL_K32")) == 0);
  if (k32_on && dl_collapse_on() && c1_wgrad_on() && llp_rank_ok(p, dshf)) {
    // layers 1 .. 5 without f2 (gen_nets.hip, "the forward without act1"): Z = F (*) f1 on the first 32 channels of a 64-channel tile (half the
    // matrix work of the 5^3 layer), y = its 27-term shifted sum.  The backward in its rank form does not miss f2 (q comes from P).
    char* tail = G + p.tail;
    NC_TRY(dl_tail_compose(params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail, hs));
    const float* F = dl_fold_fwd64(tail, params + p.w[1], hs);
    if (!F) return NC_ERR_HIP;
    NC_TRY(conv_fwd_h_na1(V + p.f1h, F, Ff, c5, dtype, cws, p.conv_ws, hs));
    *kept = kLpNoF2;
    return dl_combine27(Ff, y, N, S0, S1, S2, 64, hs);
  }
  NC_TRY(conv_fwd_h_c8(V + p.f1h, params + p.w[1], nullptr, V + p.f2h, 64, 0, c5, dtype, cws, p.conv_ws, hs));
  if (p.c3 && dl_collapse_on() && c1_wgrad_on()) {
    // layers 2 .. 5 as ONE 64 -> 1 convolution of f2 (gen_nets.hip, "the collapsed tail"): the data-gradient form of the one-channel 3^3 kernel
    // with the tap-flipped composed weights IS that convolution.
    char* tail = G + p.tail;
    NC_TRY(dl_tail_compose(params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail, hs));
    *kept = kLpCollapsed;
    return conv_c1_dgrad_h(V + p.f2h, dl_tail_Ef(tail), y, N, S0, S1, S2, 3, c1ws, p.c1_ws, hs);
  }
  NC_TRY(conv_fwd_h_c8(V + p.f2h, params + p.w[2], nullptr, V + p.f3h, 64, 0, c3, dtype, cws, p.conv_ws, hs));
  hipLaunchKernelGGL(k_lin_tail_prep, dim3(1), dim3(64), 0, hs, params + p.w[3], params + p.w[4], params + p.w[5], (float*)(V + p.weff),
                     (float*)(V + p.u1));
  NC_TRY(check_launch("lin_tail_prep"));
  return c8_dot64(V + p.f3h, (const float*)(V + p.weff), nullptr, y, N, p.S, dtype, hs);
}

int nc_deep_linear_lp_bwd(const float* params, const float* x, const void* saved, const float* dy, float* dx, float* dparams,
                          int N, int S0, int S1, int S2, int dtype, void* ws, size_t ws_bytes, void* stream, unsigned kept) {
  if (!params || !x || !saved || !dy || !dparams) { set_error("deep_linear_lp_bwd: null pointer"); return NC_ERR_ARG; }
  if (kept > kLpNoF2) { set_error("deep_linear_lp_bwd: `kept` is not a value nc_deep_linear_lp_fwd returns"); return NC_ERR_ARG; }
  if (dtype != NC_DT_BF16) { set_error("deep_linear_lp_bwd: dtype must be NC_DT_BF16"); return NC_ERR_ARG; }
  LLp p;
  if (!llp_plan(p, N, S0, S1, S2) || !p.ok) { set_error("deep_linear_lp_bwd: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < llp_ws_bytes(p)) { set_error("deep_linear_lp_bwd: workspace too small"); return NC_ERR_WS; }
  const WeightDiffusion wd_;  // the bias-free, norm-free stack rounds its weights tap-diffused (conv_h.hip)
  hipStream_t hs = (hipStream_t)stream;
  char* cws = (char*)ws;
  char* fws = cws + al(p.conv_ws);
  char* ows = fws + al(p.f32conv_ws);
  char* c1ws = ows + al(p.o64_ws);
  char* G = c1ws + al(p.c1_ws);
  const char* V = (const char*)saved;
  float* q = (float*)(G + p.q);
  ConvDims c5, c3;
  make_dims(c5, N, 64, S0, S1, S2, 64, 5, 5, 5, 1, 2);
  make_dims(c3, N, 64, S0, S1, S2, 64, 3, 3, 3, 1, 1);
  bool rank_w1 = false;
  if (kept != kLpLayered) {  // what the FORWARD took (the switches may have moved since)
    if (!p.c3) { set_error("deep_linear_lp_bwd: `kept` names a form this shape has no kernels for"); return NC_ERR_ARG; }
    // collapsed tail (gen_nets.hip): dW2 .. dW5 in weight space from q[c][t] = sum_v dy[v] f2[c][v + t - 1]; the 5^3 layer's backward from the rank
    // structure of df2 = E . Dsh (Dsh: 27 shifted copies of dy as a bf16 C8 tensor): its weight gradient a 32 x 64 problem P, its data gradient
    // a forward convolution 32 -> 64 of Dsh with composed weights -- half the matrix work each; q is a contraction of the same P, and df2
    // itself is never written
    char* tail = G + p.tail;
    NC_TRY(dl_tail_compose(params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail, hs));
    ConvDims dsh;
    const bool rank = llp_rank_ok(p, dsh);
    if (!rank && kept == kLpNoF2) { set_error("deep_linear_lp_bwd: the forward kept no f2, and the rank forms that do without it are off"); return NC_ERR_ARG; }
    if (rank) {
      hipLaunchKernelGGL(k_dl_shift27_c8, dim3(512, 4, (unsigned)N), dim3(256), 0, hs, dy, (uint4*)(G + p.B), S0, S1, S2);
      NC_TRY(check_launch("deep_linear_lp_bwd: shifted copies of dy"));
      {
        ProfScope ps(2, 1, dsh, 1, hs);
        NC_TRY(conv_wgrad_c8x(G + p.B, V + p.f1h, dl_tail_P(tail), dsh, NC_DT_BF16, cws, p.conv_ws, hs));
      }
      NC_TRY(dl_q_from_p(tail, params + p.w[1], hs));
    } else {
      // q = the one-channel weight gradient with x := dy, dY := f2
      NC_TRY(conv_c1_wgrad_h(dy, V + p.f2h, dl_tail_q(tail), N, S0, S1, S2, 3, G + p.F, (size_t)N * 64 * p.S * 4, hs));
    }
    NC_TRY(dl_tail_grads(params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail, dparams + p.w[2], dparams + p.w[3], dparams + p.w[4],
                         dparams + p.w[5], hs));
    if (rank) {
      NC_TRY(dl_w1_contract(tail, dparams + p.w[1], hs));
      const float* wf = dl_w1_fold(tail, params + p.w[1], hs);
      if (!wf) return NC_ERR_HIP;
      NC_TRY(conv_fwd_h_c8(G + p.B, wf, nullptr, G + p.A, 64, 0, dsh, NC_DT_BF16, cws, p.conv_ws, hs));
      rank_w1 = true;
    } else {
      NC_TRY(conv_c1_fwd_h(dy, dl_tail_Ef(tail), nullptr, G + p.B, 64, 0, N, S0, S1, S2, 3, NC_DT_BF16, c1ws, p.c1_ws, hs));  // df2 = flip(E) (*) dy as C8
    }
  } else {
  // tail: df3 = w_eff (x) dy (C8), q = sum dy f3
  NC_TRY(c8_outer64(dy, V + p.f3h, (const float*)(V + p.weff), G + p.A, q, nullptr, N, p.S, dtype, ows, p.o64_ws, hs));
  hipLaunchKernelGGL(k_lin_tail_wgrad, dim3(1), dim3(256), 0, hs, params + p.w[3], params + p.w[4], params + p.w[5],
                     (const float*)(V + p.u1), q, dparams + p.w[3], dparams + p.w[4], dparams + p.w[5]);
  NC_TRY(check_launch("lin_tail_wgrad"));
  // 3^3 layer
  {
    ProfScope ps(2, 1, c3, 1, hs);
    NC_TRY(conv_wgrad_h(nullptr, V + p.f2h, nullptr, G + p.A, dparams + p.w[2], c3, NC_DT_BF16, cws, p.conv_ws, hs));
  }
  NC_TRY(conv_dgrad_h_c8(G + p.A, params + p.w[2], G + p.B, 64, 0, c3, NC_DT_BF16, cws, p.conv_ws, hs));
  }
  // 5^3 layer: its data gradient feeds the fp32 one-channel 7^3 kernels, so it leaves as fp32 NCDHW
  if (!rank_w1) {
    ProfScope ps(2, 1, c5, 1, hs);
    NC_TRY(conv_wgrad_h(nullptr, V + p.f1h, nullptr, G + p.B, dparams + p.w[1], c5, NC_DT_BF16, cws, p.conv_ws, hs));
  }
  float* Ff = (float*)(G + p.F);
  if (p.c1) {
    // df1 stays C8 (in A: df3 is no longer needed); data gradient of the 7^3 layer in pseudo-channel form; its weight
    // gradient on the fp32 tap-axis kernel, fed with the fp32 copy of df1
    if (!rank_w1) NC_TRY(conv_dgrad_h_c8(G + p.B, params + p.w[1], G + p.A, 64, 0, c5, NC_DT_BF16, cws, p.conv_ws, hs));
    if (dx) NC_TRY(conv_c1_dgrad_h(G + p.A, params + p.w[0], dx, N, S0, S1, S2, 7, c1ws, p.c1_ws, hs));
    {
      const size_t fbytes = (size_t)N * 64 * p.S * 4, need = c1_wgrad_h_ws_bytes(N, S0, S1, S2, 7);
      if (c1_wgrad_on() && need > 0 && need <= fbytes)  // 16-bit weight gradient from the C8 gradient; scratch = the unused fp32 buffer
        return conv_c1_wgrad_h(x, G + p.A, dparams + p.w[0], N, S0, S1, S2, 7, Ff, fbytes, hs);
    }
    NC_TRY(c8_to_f32(G + p.A, 64, 0, Ff, N, 64, p.S, NC_DT_BF16, hs));
    return nc_conv_wgrad(x, Ff, dparams + p.w[0], nullptr, N, 1, S0, S1, S2, 64, 7, 7, 7, 1, 3, fws, p.f32conv_ws, stream);
  }
  {
    ProfScope ps(1, 1, c5, 1, hs);
    NC_TRY(conv_dgrad_h(nullptr, G + p.B, params + p.w[1], Ff, c5, NC_DT_BF16, cws, p.conv_ws, hs));
  }
  // 7^3 layer (fp32)
  if (dx) NC_TRY(nc_conv_dgrad(Ff, params + p.w[0], dx, N, 1, S0, S1, S2, 64, 7, 7, 7, 1, 3, fws, p.f32conv_ws, stream));
  return nc_conv_wgrad(x, Ff, dparams + p.w[0], nullptr, N, 1, S0, S1, S2, 64, 7, 7, 7, 1, 3, fws, p.f32conv_ws, stream);
}

}  // extern "C"
