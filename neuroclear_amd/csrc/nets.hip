// Whole-network entry points for the PatchGAN discriminator (reference models/networks.py:1009-1067,
// NLayerDiscriminator with InstanceNorm: conv k4 s2 + LeakyReLU, (conv k4 s2 + IN + LeakyReLU) x (n_layers - 1),
// conv k4 s1 + IN + LeakyReLU, conv k4 s1 -> 1 channel; every conv biased), 2-D or 3-D.
//
// Why: a discriminator pass on a handful of 108^2 planes is ~25 kernels of 5-40 us.  Driven op by op from Python the
// host needs 30-80 us per launch, so the four discriminator chains of the generator loss were enqueue-bound (the GPU
// sat idle between a chain's forward and its backward).  One C call per forward and one per backward launches the same
// kernels back to back.  Same kernels, same order, same numbers as the op-by-op path.
//
// Parameter blob = the 2 * (n_layers + 2) tensors in state-dict order (model.0.weight, model.0.bias, model.2.weight,
// ...), packed back to back.  `saved` (nc_patchgan_saved_floats) receives what the backward needs: every conv's raw
// output, every activation, the InstanceNorm statistics.
#include <cstdlib>

#include "common.hpp"

using namespace nc;

namespace {

struct PgLayer {
  int C, K, stride;   // channels in / out, stride
  int iD, iH, iW;     // input spatial size (iD = 1 for 2-D)
  int oD, oH, oW;
  bool norm;          // InstanceNorm + LeakyReLU after the conv (else: LeakyReLU only for layer 0, nothing for the head)
  size_t w_off, b_off;        // floats into the parameter blob
  size_t raw_off, act_off;    // floats into `saved`: raw conv output; activation that feeds the NEXT conv
  size_t stat_off;            // mean [B*K] then rstd [B*K]
};

struct PgPlan {
  int nl;            // number of convs = n_layers + 2
  PgLayer L[8];
  size_t params, saved;       // floats
  bool fused0;                // first layer = conv + LeakyReLU in one kernel (patchgan_edge.hip): only its activation is stored
  size_t max_act;             // largest per-layer tensor (floats), for the gradient ping-pong buffers
  size_t conv_ws, in_ws;      // bytes
  int oD, oH, oW;
};

bool pg_plan(PgPlan& P, int B, int D, int H, int W, int n_layers, int ndf, int nd) {
  if (B < 1 || H < 4 || W < 4 || n_layers < 1 || n_layers > 6 || ndf < 1 || (nd != 2 && nd != 3)) return false;
  if (nd == 2 && D != 1) return false;
  if (nd == 3 && D < 4) return false;
  P = PgPlan{};
  P.nl = n_layers + 2;
  int cin = 1, mult = 1;
  int d = D, h = H, w = W;
  size_t po = 0, so = 0;
  const int k3 = nd == 3 ? 64 : 16;
  for (int i = 0; i < P.nl; ++i) {
    PgLayer& l = P.L[i];
    const bool head = i == P.nl - 1;
    if (i == 0) {
      mult = 1;
      static const bool pg1 = !(getenv("NC_PG1") && atoi(getenv("NC_PG1")) == 0);
      ConvDims c0;
      P.fused0 = pg1 && nd == 2 && make_dims(c0, B, 1, 1, h, w, ndf, 1, 4, 4, 2, 1) && pg1_supported(c0);
    }
    else if (!head) mult = (1 << i) < 8 ? (1 << i) : 8;
    l.C = cin;
    l.K = head ? 1 : ndf * mult;
    l.stride = (i < n_layers) ? 2 : 1;
    l.norm = i > 0 && !head;
    l.iD = d; l.iH = h; l.iW = w;
    ConvDims cd;
    if (!make_dims(cd, B, l.C, d, h, w, l.K, nd == 3 ? 4 : 1, 4, 4, l.stride, 1)) return false;
    l.oD = cd.Do; l.oH = cd.Ho; l.oW = cd.Wo;
    l.w_off = po; po += (size_t)l.K * l.C * k3;
    l.b_off = po; po += (size_t)l.K;
    const size_t on = (size_t)B * l.K * l.oD * l.oH * l.oW;
    l.raw_off = so; so += on;
    l.act_off = so;
    if (!head) so += on;
    l.stat_off = so;
    if (l.norm) so += 2 * (size_t)B * l.K;
    if (on > P.max_act) P.max_act = on;
    const size_t in_n = (size_t)B * l.C * d * h * w;
    if (in_n > P.max_act) P.max_act = in_n;
    const size_t cw = nc_conv_ws_bytes(B, l.C, d, h, w, l.K, nd == 3 ? 4 : 1, 4, 4, l.stride, 1);
    if (cw > P.conv_ws) P.conv_ws = cw;
    if (l.norm) {
      const size_t iw = nc_instnorm_bwd_dbias_ws_bytes(B * l.K, (long)l.oD * l.oH * l.oW);
      if (iw > P.in_ws) P.in_ws = iw;
    }
    cin = l.K; d = l.oD; h = l.oH; w = l.oW;
  }
  P.params = po; P.saved = so;
  P.oD = d; P.oH = h; P.oW = w;
  return true;
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

#define NC_TRY(expr) do { int e_ = (expr); if (e_) return e_; } while (0)

extern "C" {

size_t nc_patchgan_param_floats(int n_layers, int ndf, int nd) {
  PgPlan P;
  return pg_plan(P, 1, nd == 3 ? 64 : 1, 64, 64, n_layers, ndf, nd) ? P.params : 0;
}

size_t nc_patchgan_saved_floats(int B, int D, int H, int W, int n_layers, int ndf, int nd) {
  PgPlan P;
  return pg_plan(P, B, D, H, W, n_layers, ndf, nd) ? P.saved : 0;
}

// scratch for fwd and bwd: conv workspace + InstanceNorm workspace + two gradient ping-pong buffers
size_t nc_patchgan_ws_bytes(int B, int D, int H, int W, int n_layers, int ndf, int nd) {
  PgPlan P;
  if (!pg_plan(P, B, D, H, W, n_layers, ndf, nd)) return 0;
  return align256(P.conv_ws) + align256(P.in_ws) + 2 * align256(P.max_act * sizeof(float)) + 256;
}

int nc_patchgan_out_shape(int B, int D, int H, int W, int n_layers, int ndf, int nd, int* oD, int* oH, int* oW) {
  PgPlan P;
  if (!pg_plan(P, B, D, H, W, n_layers, ndf, nd)) { set_error("patchgan: bad shape"); return NC_ERR_SHAPE; }
  *oD = P.oD; *oH = P.oH; *oW = P.oW;
  return NC_OK;
}

// forward of planes b0 .. b0 + B - 1 of a batch of Btot: `saved` has the layout of the whole batch (P is its plan), x / y
// are the B planes themselves.  Convolutions, InstanceNorm and LeakyReLU are per sample, so a part of the batch is the
// same kernels on offset pointers.
static int pg_fwd(const PgPlan& P, const float* params, const float* x, float* y, float* saved, int Btot, int b0, int B,
                  int nd, void* ws, void* stream) {
  void* cws = ws;
  void* iws = (char*)ws + align256(P.conv_ws);
  const int kd = nd == 3 ? 4 : 1;
  const float* in = x;
  for (int i = 0; i < P.nl; ++i) {
    const PgLayer& l = P.L[i];
    const bool head = i == P.nl - 1;
    const long S = (long)l.oD * l.oH * l.oW;
    const size_t boff = (size_t)b0 * l.K * S;
    float* raw = head ? y : saved + l.raw_off + boff;
    float* act = saved + l.act_off + boff;
    if (i == 0 && P.fused0) {  // conv + bias + LeakyReLU as one HBM stream; the raw slot stays unwritten
      ConvDims c0;
      make_dims(c0, B, 1, 1, l.iH, l.iW, l.K, 1, 4, 4, 2, 1);
      NC_TRY(conv_fwd_pg1(in, params + l.w_off, params + l.b_off, act, c0, 0.2f, (hipStream_t)stream));
      in = act;
      continue;
    }
    NC_TRY(nc_conv_fwd(in, params + l.w_off, params + l.b_off, raw, B, l.C, l.iD, l.iH, l.iW, l.K, kd, 4, 4, l.stride, 1,
                       cws, P.conv_ws, stream));
    if (head) break;  // y is the output; the head's slot in `saved` stays unused
    if (l.norm) {
      float* mean = saved + l.stat_off + (size_t)b0 * l.K;
      float* rstd = saved + l.stat_off + (size_t)Btot * l.K + (size_t)b0 * l.K;
      NC_TRY(nc_instnorm_fwd(raw, 1e-5f, 0.2f, mean, rstd, act, B * l.K, S, iws, P.in_ws, stream));
    } else {
      NC_TRY(nc_leaky_relu_fwd(raw, 0.2f, act, (long)B * l.K * S, stream));
    }
    in = act;
  }
  return NC_OK;
}

int nc_patchgan_fwd(const float* params, const float* x, float* y, float* saved, int B, int D, int H, int W,
                    int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream) {
  return nc_patchgan_fwd_part(params, x, y, saved, B, 0, B, D, H, W, n_layers, ndf, nd, ws, ws_bytes, stream);
}

int nc_patchgan_fwd_part(const float* params, const float* x, float* y, float* saved, int Btot, int b0, int B, int D, int H,
                         int W, int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream) {
  if (!params || !x || !y || !saved) { set_error("patchgan_fwd: null pointer"); return NC_ERR_ARG; }
  PgPlan P;
  if (B < 1 || b0 < 0 || b0 + B > Btot || !pg_plan(P, Btot, D, H, W, n_layers, ndf, nd)) { set_error("patchgan_fwd: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < nc_patchgan_ws_bytes(Btot, D, H, W, n_layers, ndf, nd)) { set_error("patchgan_fwd: workspace too small"); return NC_ERR_WS; }
  return pg_fwd(P, params, x, y, saved, Btot, b0, B, nd, ws, stream);
}

// dparams (nullable): packed like params, OVERWRITTEN with this call's parameter gradients.  dx (nullable): gradient
// with respect to the input planes.
static int pg_bwd(const PgPlan& P, const float* params, const float* x, const float* saved, const float* dy, float* dx,
                  float* dparams, int Btot, int b0, int B, int nd, void* ws, void* stream) {
  void* cws = ws;
  void* iws = (char*)ws + align256(P.conv_ws);
  float* ga = (float*)((char*)iws + align256(P.in_ws));
  float* gb = (float*)((char*)ga + align256(P.max_act * sizeof(float)));
  const int kd = nd == 3 ? 4 : 1;
  const float* g = dy;  // gradient with respect to the current layer's raw conv output
  bool have_db = false;  // this layer's bias gradient was already taken by the backward of the norm behind it
  for (int i = P.nl - 1; i >= 0; --i) {
    const PgLayer& l = P.L[i];
    const float* in = x;
    if (i > 0) {
      const PgLayer& q = P.L[i - 1];
      in = saved + q.act_off + (size_t)b0 * q.K * ((long)q.oD * q.oH * q.oW);
    }
    if (i == 0 && P.fused0) {
      // g is still the gradient BEHIND the first layer's LeakyReLU (no pass was spent on it below): both gradients of the
      // layer pull it through the mask of the stored activation on the fly
      ConvDims c0;
      make_dims(c0, B, 1, 1, l.iH, l.iW, l.K, 1, 4, 4, 2, 1);
      const float* act0 = saved + l.act_off + (size_t)b0 * l.K * ((long)l.oD * l.oH * l.oW);
      if (dparams)
        NC_TRY(conv_wgrad_pg1(in, g, act0, 0.2f, dparams + l.w_off, dparams + l.b_off, c0, cws, P.conv_ws, (hipStream_t)stream));
      if (dx) NC_TRY(conv_dgrad_pg1(g, act0, 0.2f, params + l.w_off, dx, c0, (hipStream_t)stream));
      break;
    }
    if (dparams)
      NC_TRY(nc_conv_wgrad(in, g, dparams + l.w_off, have_db ? nullptr : dparams + l.b_off, B, l.C, l.iD, l.iH, l.iW, l.K, kd,
                           4, 4, l.stride, 1, cws, P.conv_ws, stream));
    if (i == 0) {
      if (dx) NC_TRY(nc_conv_dgrad(g, params + l.w_off, dx, B, l.C, l.iD, l.iH, l.iW, l.K, kd, 4, 4, l.stride, 1, cws,
                                   P.conv_ws, stream));
      break;
    }
    // gradient with respect to this conv's input = the previous layer's activation ...
    float* gin = (g == ga) ? gb : ga;
    NC_TRY(nc_conv_dgrad(g, params + l.w_off, gin, B, l.C, l.iD, l.iH, l.iW, l.K, kd, 4, 4, l.stride, 1, cws, P.conv_ws,
                         stream));
    // ... pulled back through that layer's activation (and InstanceNorm) to its raw conv output
    const PgLayer& pl = P.L[i - 1];
    const long S = (long)pl.oD * pl.oH * pl.oW;
    float* graw = (gin == ga) ? gb : ga;
    const float* praw = saved + pl.raw_off + (size_t)b0 * pl.K * S;
    have_db = false;
    if (pl.norm) {
      const float* mean = saved + pl.stat_off + (size_t)b0 * pl.K;
      const float* rstd = saved + pl.stat_off + (size_t)Btot * pl.K + (size_t)b0 * pl.K;
      // the norm's backward has dx -- the gradient at the previous conv's output -- in registers: its per-channel sum is
      // that conv's bias gradient (same hand-over as ops.BiasLink on the op-by-op path)
      if (dparams && ((long)B * pl.K <= 65535 || S <= 2048)) {
        NC_TRY(nc_instnorm_act_bwd_dbias(gin, praw, mean, rstd, 0.2f, graw, dparams + pl.b_off, B, pl.K, S, iws, P.in_ws,
                                         stream));
        have_db = true;
      } else {
        NC_TRY(nc_instnorm_act_bwd(gin, praw, mean, rstd, 0.2f, graw, B * pl.K, S, iws, P.in_ws, stream));
      }
    } else if (i == 1 && P.fused0) {
      graw = gin;  // the first layer's gradients apply the mask themselves
    } else {
      NC_TRY(nc_leaky_relu_bwd(gin, praw, 0.2f, graw, (long)B * pl.K * S, stream));
    }
    g = graw;
  }
  return NC_OK;
}

int nc_patchgan_bwd(const float* params, const float* x, const float* saved, const float* dy, float* dx, float* dparams,
                    int B, int D, int H, int W, int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream) {
  if (!params || !x || !saved || !dy) { set_error("patchgan_bwd: null pointer"); return NC_ERR_ARG; }
  PgPlan P;
  if (!pg_plan(P, B, D, H, W, n_layers, ndf, nd)) { set_error("patchgan_bwd: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < nc_patchgan_ws_bytes(B, D, H, W, n_layers, ndf, nd)) { set_error("patchgan_bwd: workspace too small"); return NC_ERR_WS; }
  return pg_bwd(P, params, x, saved, dy, dx, dparams, B, 0, B, nd, ws, stream);
}

int nc_patchgan_bwd_part(const float* params, const float* x, const float* saved, const float* dy, float* dx, int Btot, int b0,
                         int B, int D, int H, int W, int n_layers, int ndf, int nd, void* ws, size_t ws_bytes, void* stream) {
  if (!params || !x || !saved || !dy || !dx) { set_error("patchgan_bwd_part: null pointer"); return NC_ERR_ARG; }
  PgPlan P;
  if (B < 1 || b0 < 0 || b0 + B > Btot || !pg_plan(P, Btot, D, H, W, n_layers, ndf, nd)) { set_error("patchgan_bwd_part: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < nc_patchgan_ws_bytes(Btot, D, H, W, n_layers, ndf, nd)) { set_error("patchgan_bwd_part: workspace too small"); return NC_ERR_WS; }
  return pg_bwd(P, params, x, saved, dy, dx, nullptr, Btot, b0, B, nd, ws, stream);
}
}  // extern "C"
