// Whole-network TRAINING entry points of the two generators: one C call per direction, as nets.hip does for the PatchGAN.
//   Unet_deconv          reference models/networks.py:478-538   (forward at :512-538; backward = autograd of it, driven
//                        by loss_G.backward() at models/axial_to_lateral_gan_apollo_model.py:283)
//   DeepLinearGenerator  reference models/networks.py:893-917   (bias-free linear chain 7^3 / 5^3 / 3^3 / 1 / 1 / 1)
// Same kernels in the same order as the layer-by-layer Python path (neuroclear_amd/models/networks.py), so outputs and
// gradients are bit-identical to it (DeepLinearGenerator: with nc_set_dl_collapse(0); its default evaluation exploits the chain's linearity,
// see "the collapsed tail" below); what the single call removes is ~60 autograd nodes per direction on the host, the
// torch.cat copies of the two skip connections (the producers write straight into halves of the concat buffers) and the
// at::native adds that merge the two gradients of a skip tensor (the max-pool backward adds the skip gradient itself).
//
// Parameter blob = the tensors in state-dict order, packed back to back.  `saved` receives what the
// backward needs (raw conv outputs, activations, InstanceNorm statistics); sizes from the *_saved_floats queries.

#include <atomic>
#include <cstdlib>

#include "common.hpp"

using namespace nc;

#define NC_TRY(expr) do { int e_ = (expr); if (e_) return e_; } while (0)

namespace {

size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

// Which S3 (three-term) copies of convolution inputs a training forward left in its `saved` buffer is a bit mask (bit i = layer i) that
// the forward RETURNS to the caller (`kept`, a host word) and the caller hands to the backward of the same `saved` buffer: the
// forward of a layer that runs on the split-operand kernels converts its input INTO the saved buffer (conv_fwd_keep) and the backward
// gives that tensor to the weight gradient instead of converting x again.  The decision (library switch, shape coverage) is host
// state, so the mask is host data; it travels with the autograd context that owns `saved`, not with a table keyed by its address.
size_t s3_floats(size_t elems) { return up64((elems * 6 + 3) / 4); }
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- Unet_deconv ---------------------------------------------------------------------------------------------------
struct UBlock { int C, K, lvl; };  // 3^3 conv + InstanceNorm + ReLU; lvl 0 = full resolution, 1 = half, 2 = quarter
const UBlock kUB[10] = {{1, 64, 0},    {64, 64, 0},    {64, 128, 1},  {128, 128, 1}, {128, 256, 2},
                        {256, 256, 2}, {256, 256, 2}, {256, 128, 1}, {128, 128, 1}, {128, 64, 0}};

struct UPlan {
  int N, d[3][3];      // spatial size per level
  long S[3];           // voxels per level
  // saved (floats): activations, raw conv outputs, statistics
  size_t a1, cat1, p1, a2, cat2, p2, b1, b2, b3, e2a, e2b, e1, t1, raw[10], mean[10], rstd[10], saved;
  size_t xs3[10];      // S3 copy of block i's input (i >= 1), see `kept`
  // backward scratch (floats)
  size_t G1, G2, G3, H1, H2, H3, Q1, Q2, s1, s2, T, grads;
  size_t conv_ws, in_ws, convT_ws;  // bytes
};

bool u_plan(UPlan& p, int N, int S0, int S1, int S2) {
  if (N < 1 || S0 < 4 || S1 < 4 || S2 < 4 || (S0 & 3) || (S1 & 3) || (S2 & 3)) return false;
  p = UPlan{};
  p.N = N;
  for (int l = 0; l < 3; ++l) {
    p.d[l][0] = S0 >> l; p.d[l][1] = S1 >> l; p.d[l][2] = S2 >> l;
    p.S[l] = (long)p.d[l][0] * p.d[l][1] * p.d[l][2];
  }
  const size_t n = (size_t)N, S = (size_t)p.S[0], Sh = (size_t)p.S[1], Sq = (size_t)p.S[2];
  size_t off = 0;
  auto take = [&](size_t k) { size_t r = off; off += up64(k); return r; };
  p.a1 = take(n * 64 * S); p.cat1 = take(n * 128 * S); p.p1 = take(n * 64 * Sh); p.a2 = take(n * 128 * Sh);
  p.cat2 = take(n * 256 * Sh); p.p2 = take(n * 128 * Sq); p.b1 = take(n * 256 * Sq); p.b2 = take(n * 256 * Sq);
  p.b3 = take(n * 256 * Sq); p.e2a = take(n * 128 * Sh); p.e2b = take(n * 128 * Sh); p.e1 = take(n * 64 * S);
  p.t1 = take(n * S);
  for (int i = 0; i < 10; ++i) {
    p.raw[i] = take(n * kUB[i].K * (size_t)p.S[kUB[i].lvl]);
    p.mean[i] = take(n * kUB[i].K);
    p.rstd[i] = take(n * kUB[i].K);
  }
  for (int i = 1; i < 10; ++i) { p.xs3[i] = off; off += s3_floats(n * kUB[i].C * (size_t)p.S[kUB[i].lvl]); }
  p.saved = off;
  off = 0;
  p.G1 = take(n * 64 * S); p.G2 = take(n * 64 * S); p.G3 = take(n * 128 * S);
  p.H1 = take(n * 128 * Sh); p.H2 = take(n * 128 * Sh); p.H3 = take(n * 256 * Sh);
  p.Q1 = take(n * 256 * Sq); p.Q2 = take(n * 256 * Sq); p.s1 = take(n * S); p.s2 = take(n * S);
  p.T = take(n * 64 * S);  // dense copy of a concat half's gradient when N > 1
  p.grads = off;
  auto upd = [&](int C, int K, int l, int k) {
    const size_t b = nc_conv_ws_bytes(N, C, p.d[l][0], p.d[l][1], p.d[l][2], K, k, k, k, 1, k / 2);
    if (b > p.conv_ws) p.conv_ws = b;
  };
  for (int i = 0; i < 10; ++i) upd(kUB[i].C, kUB[i].K, kUB[i].lvl, 3);
  upd(64, 1, 0, 1); upd(1, 1, 0, 1);
  p.in_ws = nc_instnorm_bwd_dbias_ws_bytes(N * 256, (long)S);
  for (int i = 1; i < 10; ++i) {  // the same region holds the partial sums of a block whose convolution takes its InstanceNorm statistics itself
    const size_t b = s3x_stats_bytes(N, p.d[kUB[i].lvl][0], p.d[kUB[i].lvl][1], p.d[kUB[i].lvl][2], kUB[i].K, 3);
    if (b > p.in_ws) p.in_ws = b;
  }
  p.convT_ws = nc_convT_ws_bytes(N, 256, p.d[2][0], p.d[2][1], p.d[2][2], 128);
  const size_t c2 = nc_convT_ws_bytes(N, 128, p.d[1][0], p.d[1][1], p.d[1][2], 64);
  if (c2 > p.convT_ws) p.convT_ws = c2;
  return true;
}

// offsets (floats) of the 28 tensors inside the packed blob, state-dict order:
// dc1.0 dc1.3 dc2.0 dc2.3 bot.0 bot.3 bot.6 t_conv2 ex2.0 ex2.3 t_conv1 ex1.0 1x1 1x1_2
struct UOff { size_t w[14], b[14], total; };  // ids 0..9 = blocks, 10 = t_conv2, 11 = t_conv1, 12 = one_by_one, 13 = one_by_one_2
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

// dst[n][0..C) <- src[n][c0..c0+C) of a [N][Ctot][S] tensor (dense result); N == 1 needs no copy
int gather_half(const float* src, float* dst, int N, int Ctot, int c0, int C, long S, hipStream_t s) {
  if (hipMemcpy2DAsync(dst, (size_t)C * S * 4, src + (size_t)c0 * S, (size_t)Ctot * S * 4, (size_t)C * S * 4, N,
                       hipMemcpyDeviceToDevice, s) != hipSuccess) {
    set_error("gather_half: hipMemcpy2DAsync failed");
    return NC_ERR_HIP;
  }
  return NC_OK;
}

size_t u_ws_bytes(const UPlan& p, bool bwd) {
  return align256(p.conv_ws) + align256(p.in_ws) + align256(p.convT_ws) + (bwd ? p.grads * sizeof(float) : 0) + 256;
}

}  // namespace

extern "C" {

size_t nc_unet_deconv_param_floats(void) { return u_offsets().total; }

size_t nc_unet_deconv_saved_floats(int N, int S0, int S1, int S2) {
  UPlan p;
  return u_plan(p, N, S0, S1, S2) ? p.saved : 0;
}

size_t nc_unet_deconv_train_ws_bytes(int N, int S0, int S1, int S2) {
  UPlan p;
  return u_plan(p, N, S0, S1, S2) ? u_ws_bytes(p, true) : 0;
}

int nc_unet_deconv_train_fwd(const float* params, const float* x, float* y, float* saved, int N, int S0, int S1, int S2,
                             void* ws, size_t ws_bytes, void* stream, unsigned* kept) {
  NetworkScope net_scope;
  if (!params || !x || !y || !saved) { set_error("unet_deconv_train_fwd: null pointer"); return NC_ERR_ARG; }
  UPlan p;
  if (!u_plan(p, N, S0, S1, S2)) {
    set_error("unet_deconv_train_fwd: every edge must be a positive multiple of 4 (got %d,%d,%d)", S0, S1, S2);
    return NC_ERR_SHAPE;
  }
  if (!ws || ws_bytes < u_ws_bytes(p, false)) { set_error("unet_deconv_train_fwd: workspace too small"); return NC_ERR_WS; }
  const UOff o = u_offsets();
  void* cws = ws;
  void* iws = (char*)ws + align256(p.conv_ws);
  float* V = saved;
  const float* P = params;
  if (kept) *kept = 0;
  hipStream_t hs = (hipStream_t)stream;
  // Which blocks run on the split-operand kernels (forward and weight gradient)?  Their input is wanted in S3 form, and the producer of
  // that input -- the normalisation + ReLU pass of the block in front, or the conversion of a transposed convolution's output -- writes
  // the S3 form straight into the consumer's slot of `saved` (xs3[i]) instead of a separate conversion pass over the fp32 tensor
  // (k_act_split3; a block fed by a max-pool still converts inside conv_fwd_keep).  NC_S3_TRAIN_FUSE=0: every block converts for itself.
  static const bool fuse_on = !(getenv("NC_S3_TRAIN_FUSE") && atoi(getenv("NC_S3_TRAIN_FUSE")) == 0);
  bool use[10], pre[10], h2l[10];  // h2l: the block's operands in the two-term form (nc_set_split_terms(2); conv_split.hip s3_layer_h2)
  ConvDims cd[10];
  for (int i = 0; i < 10; ++i) {
    const UBlock& b = kUB[i];
    const int* d = p.d[b.lvl];
    use[i] = fuse_on && i >= 1 && conv_keep_supported(N, b.C, d[0], d[1], d[2], b.K, 3);
    pre[i] = false;
    h2l[i] = make_dims(cd[i], N, b.C, d[0], d[1], d[2], b.K, 3, 3, 3, 1, 1) && use[i] && conv_layer_h2(cd[i]);
  }
  // conv (3^3, pad 1) -> raw; statistics; normalise + ReLU into `out`, where sample n's K planes start at
  // out + n * out_stride (out_stride = K * S for a dense tensor, Ctot * S for a half of a concat buffer).
  // to >= 0: block `to` consumes this output as channels [0, K) of its `to_ctot`-channel input
  unsigned kept_mask = 0;
  auto block = [&](int i, const float* in, float* out, size_t out_stride, int to, int to_ctot) -> int {
    const UBlock& b = kUB[i];
    const int* d = p.d[b.lvl];
    const long S = p.S[b.lvl];
    // Two-term form with the input already converted by its producer (a power of two known by construction: nothing can fall back): the
    // convolution's epilogue can leave the partial InstanceNorm sums (conv_s3x.hip ST) instead of a statistics pass over its output --
    // nc_set_epi_stats(2): an option, not the default (no measurable gain in the training step)
    const bool epi = pre[i] && h2l[i] && epi_stats_mode() == 2 && s3x_stats_bytes(N, d[0], d[1], d[2], b.K, 3) && s3x_stats_bytes(N, d[0], d[1], d[2], b.K, 3) <= p.in_ws;
    if (pre[i]) {  // the producers left the S3 input in saved
      NC_TRY(conv_fwd_pre(V + p.xs3[i], P + o.w[i], P + o.b[i], V + p.raw[i], N, b.C, d[0], d[1], d[2], b.K, 3, cws, p.conv_ws, stream,
                          epi ? (float*)iws : nullptr));
      kept_mask |= 1u << i;
      if (h2l[i]) kept_mask |= 1u << (16 + i);  // (bits 16..: the kept copy is an H2 tensor -- a backward under another setting of the switch ignores it)
    } else {
      bool kept1 = false;
      NC_TRY(conv_fwd_keep(in, P + o.w[i], P + o.b[i], V + p.raw[i], N, b.C, d[0], d[1], d[2], b.K, 3, cws, p.conv_ws, stream,
                           i >= 1 ? (void*)(V + p.xs3[i]) : nullptr, &kept1));
      if (kept1) kept_mask |= 1u << i;
      if (kept1 && i >= 1 && conv_layer_h2(cd[i])) kept_mask |= 1u << (16 + i);
    }
    if (epi) NC_TRY(s3x_stats_finalize((const float*)iws, P + o.b[i], N, d[0], d[1], d[2], b.K, 3, 1e-5f, V + p.mean[i], V + p.rstd[i], hs));
    else NC_TRY(nc_instnorm_stats(V + p.raw[i], N * b.K, S, 1e-5f, V + p.mean[i], V + p.rstd[i], iws, p.in_ws, stream));
    if (to >= 0 && use[to]) {  // fp32 (the backward of the pool / the fallback paths read it) AND the consumer's S3 operand in one pass
      NC_TRY(act_operand(cd[to], V + p.raw[i], V + p.mean[i], V + p.rstd[i], 0.f, out, (long)out_stride, V + p.xs3[to], N, b.K, S, to_ctot, 0, hs));
      if (to_ctot == b.K) pre[to] = true;  // (a concat input is complete once its second half has been converted, below)
      return NC_OK;
    }
    if (out_stride == (size_t)b.K * S || N == 1)
      return nc_instnorm_act_fwd(V + p.raw[i], V + p.mean[i], V + p.rstd[i], 0.f, out, N * b.K, S, stream);
    for (int n = 0; n < N; ++n)
      NC_TRY(nc_instnorm_act_fwd(V + p.raw[i] + (size_t)n * b.K * S, V + p.mean[i] + n * b.K, V + p.rstd[i] + n * b.K, 0.f,
                                 out + n * out_stride, b.K, S, stream));
    return NC_OK;
  };
  const long S = p.S[0], Sh = p.S[1], Sq = p.S[2];
  const int *d0 = p.d[0], *d1 = p.d[1], *d2 = p.d[2];
  NC_TRY(block(0, x, V + p.a1, (size_t)64 * S, 1, 64));
  NC_TRY(block(1, V + p.a1, V + p.cat1, (size_t)128 * S, 9, 128));
  for (int n = 0; n < N; ++n)
    NC_TRY(nc_maxpool2_fwd(V + p.cat1 + (size_t)n * 128 * S, V + p.p1 + (size_t)n * 64 * Sh, 64, d0[0], d0[1], d0[2], stream));
  NC_TRY(block(2, V + p.p1, V + p.a2, (size_t)128 * Sh, 3, 128));
  NC_TRY(block(3, V + p.a2, V + p.cat2, (size_t)256 * Sh, 7, 256));
  for (int n = 0; n < N; ++n)
    NC_TRY(nc_maxpool2_fwd(V + p.cat2 + (size_t)n * 256 * Sh, V + p.p2 + (size_t)n * 128 * Sq, 128, d1[0], d1[1], d1[2], stream));
  NC_TRY(block(4, V + p.p2, V + p.b1, (size_t)256 * Sq, 5, 256));
  NC_TRY(block(5, V + p.b1, V + p.b2, (size_t)256 * Sq, 6, 256));
  NC_TRY(block(6, V + p.b2, V + p.b3, (size_t)256 * Sq, -1, 0));
  // The transposed convolutions run on the bf16 matrix cores from an S3 copy of their input (convt_s3.hip; the fp32 input stays in
  // `saved` for their backward) whenever the split-operand kernels are on (the layer-by-layer host path makes the same choice:
  // nc_convT_k2s2_split_active), and write the S3 form of their half of the concatenation themselves when the consuming block takes it
  const bool ct2 = nc_convT_k2s2_split_active(1, 256, d2[0], d2[1], d2[2], 128) &&
                   p.conv_ws >= nc_convT_k2s2_split_ws_bytes(1, 256, d2[0], d2[1], d2[2], 128);
  // Two-term form: the transposed convolution writes the H2 form of its half itself -- its power of two comes from a BOUND (one tap per input
  // channel and output voxel: max column sum of |w| times the InstanceNorm bound of its input, convt_s3.hip), so nothing is measured and the
  // whole forward is independent of what else is in the batch.  NC_CONVT_H2=0: fp32 output, measured and converted.
  static const bool ct_h2 = !(getenv("NC_CONVT_H2") && atoi(getenv("NC_CONVT_H2")) == 0);
  const bool ct2h = ct2 && h2l[7] && ct_h2;
  unsigned* cell7 = h2_cells_of(V + p.xs3[7], (size_t)N * 256 * Sh) + 1;
  if (ct2h) {
    NC_TRY(h2_zero_cells(cell7, 1, hs));
    NC_TRY(convT_h2_bound(P + o.w[10], P + o.b[10], 256, 128, sqrtf((float)Sq), cell7, hs));
  }
  for (int n = 0; n < N; ++n) {  // t_conv2 writes the second half of cat2
    if (ct2h)
      NC_TRY(convT_fwd_split_h2(V + p.b3 + (size_t)n * 256 * Sq, P + o.w[10], P + o.b[10], V + p.cat2 + ((size_t)n * 256 + 128) * Sh,
                                (char*)(V + p.xs3[7]) + (size_t)n * 256 * Sh * 4, 256, 128, 1, 256, d2[0], d2[1], d2[2], 128, cell7, cws, p.conv_ws, hs));
    else if (ct2)
      NC_TRY(nc_convT_k2s2_fwd_split(V + p.b3 + (size_t)n * 256 * Sq, nullptr, P + o.w[10], P + o.b[10], V + p.cat2 + ((size_t)n * 256 + 128) * Sh,
                                     use[7] && !h2l[7] ? (char*)(V + p.xs3[7]) + (size_t)n * 256 * Sh * 6 : nullptr, 256, 128, 1, 256, d2[0], d2[1], d2[2], 128, cws,
                                     p.conv_ws, stream));
    else
      NC_TRY(nc_convT_k2s2_fwd(V + p.b3 + (size_t)n * 256 * Sq, P + o.w[10], P + o.b[10], V + p.cat2 + ((size_t)n * 256 + 128) * Sh,
                               1, 256, d2[0], d2[1], d2[2], 128, stream));
  }
  if (use[7]) {  // ... and its S3 form completes block 7's input (the first half came from block 3's normalisation pass)
    // (two-term form: the transposed convolution's half is measured after the fact -- its power of two cannot be known while it is written)
    if (!ct2 || (h2l[7] && !ct2h)) NC_TRY(operand_into(cd[7], V + p.cat2 + (size_t)128 * Sh, (long)256 * Sh, V + p.xs3[7], N, 128, Sh, 256, 128, hs));
    pre[7] = true;
  }
  NC_TRY(block(7, V + p.cat2, V + p.e2a, (size_t)128 * Sh, 8, 128));
  NC_TRY(block(8, V + p.e2a, V + p.e2b, (size_t)128 * Sh, -1, 0));
  const bool ct1 = nc_convT_k2s2_split_active(1, 128, d1[0], d1[1], d1[2], 64) &&
                   p.conv_ws >= nc_convT_k2s2_split_ws_bytes(1, 128, d1[0], d1[1], d1[2], 64);
  const bool ct_s3 = use[9] && !h2l[9] && (ct1 || convT_fwd_s3_supported(1, 128, d1[0], d1[1], d1[2], 64));  // t_conv1 writes the S3 form of its output itself
  const bool ct1h = ct1 && h2l[9] && ct_h2;
  unsigned* cell9 = h2_cells_of(V + p.xs3[9], (size_t)N * 128 * S) + 1;
  if (ct1h) {
    NC_TRY(h2_zero_cells(cell9, 1, hs));
    NC_TRY(convT_h2_bound(P + o.w[11], P + o.b[11], 128, 64, sqrtf((float)Sh), cell9, hs));
  }
  for (int n = 0; n < N; ++n) {  // t_conv1 writes the second half of cat1
    if (ct1h)
      NC_TRY(convT_fwd_split_h2(V + p.e2b + (size_t)n * 128 * Sh, P + o.w[11], P + o.b[11], V + p.cat1 + ((size_t)n * 128 + 64) * S,
                                (char*)(V + p.xs3[9]) + (size_t)n * 128 * S * 4, 128, 64, 1, 128, d1[0], d1[1], d1[2], 64, cell9, cws, p.conv_ws, hs));
    else if (ct1)
      NC_TRY(nc_convT_k2s2_fwd_split(V + p.e2b + (size_t)n * 128 * Sh, nullptr, P + o.w[11], P + o.b[11], V + p.cat1 + ((size_t)n * 128 + 64) * S,
                                     use[9] && !h2l[9] ? (char*)(V + p.xs3[9]) + (size_t)n * 128 * S * 6 : nullptr, 128, 64, 1, 128, d1[0], d1[1], d1[2], 64, cws,
                                     p.conv_ws, stream));
    else if (ct_s3)
      NC_TRY(convT_fwd_s3(V + p.e2b + (size_t)n * 128 * Sh, P + o.w[11], P + o.b[11], V + p.cat1 + ((size_t)n * 128 + 64) * S,
                          (char*)(V + p.xs3[9]) + (size_t)n * 128 * S * 6, 128, 64, 1, 128, d1[0], d1[1], d1[2], 64, stream));
    else
      NC_TRY(nc_convT_k2s2_fwd(V + p.e2b + (size_t)n * 128 * Sh, P + o.w[11], P + o.b[11], V + p.cat1 + ((size_t)n * 128 + 64) * S,
                               1, 128, d1[0], d1[1], d1[2], 64, stream));
  }
  if (use[9]) {
    if (!ct_s3 && !ct1h) NC_TRY(operand_into(cd[9], V + p.cat1 + (size_t)64 * S, (long)128 * S, V + p.xs3[9], N, 64, S, 128, 64, hs));
    pre[9] = true;
  }
  NC_TRY(block(9, V + p.cat1, V + p.e1, (size_t)64 * S, -1, 0));
  // the 1x1 tail
  NC_TRY(nc_conv_fwd(V + p.e1, P + o.w[12], P + o.b[12], V + p.t1, N, 64, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, cws, p.conv_ws, stream));
  // one_by_one_2 + sigmoid: y doubles as the buffer of the pre-sigmoid value (the sigmoid kernel is elementwise in place)
  NC_TRY(nc_conv_fwd(V + p.t1, P + o.w[13], P + o.b[13], y, N, 1, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, cws, p.conv_ws, stream));
  if (kept) *kept = kept_mask;
  return nc_sigmoid_fwd(y, y, (long)N * S, stream);
}

// dparams: packed like params, OVERWRITTEN with this call's parameter gradients.  dx nullable (the U-Net's input is the
// real volume in the Apollo / Athena steps: its gradient -- the data gradient of the first convolution -- is skipped).
int nc_unet_deconv_bwd(const float* params, const float* x, const float* y, const float* saved, const float* dy, float* dx,
                       float* dparams, int N, int S0, int S1, int S2, void* ws, size_t ws_bytes, void* stream, unsigned kept) {
  NetworkScope net_scope;
  if (!params || !x || !y || !saved || !dy || !dparams) { set_error("unet_deconv_bwd: null pointer"); return NC_ERR_ARG; }
  UPlan p;
  if (!u_plan(p, N, S0, S1, S2)) { set_error("unet_deconv_bwd: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < u_ws_bytes(p, true)) { set_error("unet_deconv_bwd: workspace too small"); return NC_ERR_WS; }
  const UOff o = u_offsets();
  hipStream_t hs = (hipStream_t)stream;
  void* cws = ws;
  void* iws = (char*)ws + align256(p.conv_ws);
  void* tws = (char*)iws + align256(p.in_ws);
  float* G = (float*)((char*)tws + align256(p.convT_ws));
  const float* V = saved;
  const float* P = params;
  float* DP = dparams;
  const long S = p.S[0], Sh = p.S[1], Sq = p.S[2];
  const int *d0 = p.d[0], *d1 = p.d[1], *d2 = p.d[2];
  const unsigned kept_mask = kept;
  static const bool fuse_bwd = !(getenv("NC_S3_TRAIN_FUSE") && atoi(getenv("NC_S3_TRAIN_FUSE")) == 0) &&
                               true;
  // backward of block i: g = gradient at the block's (post-ReLU) output, dense [N][K][S]; `in` = the block's input.
  // draw <- InstanceNorm/ReLU backward (+ the conv's bias gradient); dW <- wgrad; gin (nullable) <- dgrad
  auto block_bwd = [&](int i, const float* g, const float* in, float* draw, float* gin) -> int {
    const UBlock& b = kUB[i];
    const int* d = p.d[b.lvl];
    const long Sl = p.S[b.lvl];
    ConvDims cdk;
    const bool now_h2 = make_dims(cdk, N, b.C, d[0], d[1], d[2], b.K, 3, 3, 3, 1, 1) && conv_layer_h2(cdk);
    const void* xs = ((kept_mask >> i) & 1) && (((kept_mask >> (16 + i)) & 1) != 0) == now_h2 ? (const void*)(V + p.xs3[i]) : nullptr;
    // the norm's backward writes the convolution's dY straight in S3 form at the head of the convolution workspace (where the
    // conversion phase of the split-operand backward would put it): no fp32 tensor, no conversion pass
    if (fuse_bwd && i >= 1 && conv_bwd_pre_supported(N, b.C, d[0], d[1], d[2], b.K, 3, gin != nullptr, p.conv_ws) &&
        instnorm_bwd_s3_supported(N, b.K, Sl)) {
      if (now_h2)  // (with the range guard's words of the dY operand: the norm backward counts, decides and -- flagged -- rewrites it as S3)
        NC_TRY(instnorm_act_bwd_dbias_h2(g, V + p.raw[i], V + p.mean[i], V + p.rstd[i], 0.f, cws, DP + o.b[i], N, b.K, Sl, iws, p.in_ws, stream,
                                         conv_bwd_guard_words(cws, N, b.K, Sl)));
      else
        NC_TRY(instnorm_act_bwd_dbias_s3(g, V + p.raw[i], V + p.mean[i], V + p.rstd[i], 0.f, cws, DP + o.b[i], N, b.K, Sl, iws, p.in_ws, stream));
      return conv_bwd_pre(in, xs, P + o.w[i], gin, DP + o.w[i], N, b.C, d[0], d[1], d[2], b.K, 3, cws, p.conv_ws, stream, now_h2 && h2_guard_can_flip());
    }
    NC_TRY(nc_instnorm_act_bwd_dbias(g, V + p.raw[i], V + p.mean[i], V + p.rstd[i], 0.f, draw, DP + o.b[i], N, b.K, Sl, iws,
                                     p.in_ws, stream));
    return conv_bwd_keep(in, xs, draw, P + o.w[i], gin, DP + o.w[i], N, b.C, d[0], d[1], d[2], b.K, 3, cws, p.conv_ws, stream);
  };
  // gradient of the second half of a concat buffer as a dense tensor
  auto upper_half = [&](const float* dcat, int Ctot, long Sl, const float** out) -> int {
    const int C = Ctot / 2;
    if (N == 1) { *out = dcat + (size_t)C * Sl; return NC_OK; }
    NC_TRY(gather_half(dcat, G + p.T, N, Ctot, C, C, Sl, hs));
    *out = G + p.T;
    return NC_OK;
  };
  // 1x1 tail
  NC_TRY(nc_sigmoid_bwd(dy, y, G + p.s1, (long)N * S, stream));
  NC_TRY(nc_conv_wgrad(V + p.t1, G + p.s1, DP + o.w[13], DP + o.b[13], N, 1, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, cws, p.conv_ws, stream));
  NC_TRY(nc_conv_dgrad(G + p.s1, P + o.w[13], G + p.s2, N, 1, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, cws, p.conv_ws, stream));
  NC_TRY(nc_conv_wgrad(V + p.e1, G + p.s2, DP + o.w[12], DP + o.b[12], N, 64, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, cws, p.conv_ws, stream));
  NC_TRY(nc_conv_dgrad(G + p.s2, P + o.w[12], G + p.G1, N, 64, d0[0], d0[1], d0[2], 1, 1, 1, 1, 1, 0, cws, p.conv_ws, stream));
  // ex_conv1_1 (cat1 -> e1)
  NC_TRY(block_bwd(9, G + p.G1, V + p.cat1, G + p.G2, G + p.G3));
  // t_conv1 (e2b -> cat1[:, 64:])
  const float* g;
  NC_TRY(upper_half(G + p.G3, 128, S, &g));
  NC_TRY(nc_convT_k2s2_dgrad(g, P + o.w[11], G + p.H1, N, 128, d1[0], d1[1], d1[2], 64, tws, p.convT_ws, stream));
  NC_TRY(nc_convT_k2s2_wgrad(V + p.e2b, g, DP + o.w[11], DP + o.b[11], N, 128, d1[0], d1[1], d1[2], 64, tws, p.convT_ws, stream));
  // ex_double_conv2 (cat2 -> e2a -> e2b)
  NC_TRY(block_bwd(8, G + p.H1, V + p.e2a, G + p.H2, G + p.H1));
  NC_TRY(block_bwd(7, G + p.H1, V + p.cat2, G + p.H2, G + p.H3));
  // t_conv2 (b3 -> cat2[:, 128:])
  NC_TRY(upper_half(G + p.H3, 256, Sh, &g));
  NC_TRY(nc_convT_k2s2_dgrad(g, P + o.w[10], G + p.Q1, N, 256, d2[0], d2[1], d2[2], 128, tws, p.convT_ws, stream));
  NC_TRY(nc_convT_k2s2_wgrad(V + p.b3, g, DP + o.w[10], DP + o.b[10], N, 256, d2[0], d2[1], d2[2], 128, tws, p.convT_ws, stream));
  // bottom_layer (p2 -> b1 -> b2 -> b3)
  NC_TRY(block_bwd(6, G + p.Q1, V + p.b2, G + p.Q2, G + p.Q1));
  NC_TRY(block_bwd(5, G + p.Q1, V + p.b1, G + p.Q2, G + p.Q1));
  NC_TRY(block_bwd(4, G + p.Q1, V + p.p2, G + p.Q2, G + p.Q1));
  // conv2 = cat2[:, :128] feeds the pool AND the skip: gradient = pool backward + the first half of dcat2
  for (int n = 0; n < N; ++n)
    NC_TRY(nc_maxpool2_bwd_add(G + p.Q1 + (size_t)n * 128 * Sq, V + p.cat2 + (size_t)n * 256 * Sh, G + p.H3 + (size_t)n * 256 * Sh,
                               G + p.H1 + (size_t)n * 128 * Sh, 128, d1[0], d1[1], d1[2], stream));
  // double_conv2 (p1 -> a2 -> conv2)
  NC_TRY(block_bwd(3, G + p.H1, V + p.a2, G + p.H2, G + p.H1));
  NC_TRY(block_bwd(2, G + p.H1, V + p.p1, G + p.H2, G + p.H1));
  for (int n = 0; n < N; ++n)
    NC_TRY(nc_maxpool2_bwd_add(G + p.H1 + (size_t)n * 64 * Sh, V + p.cat1 + (size_t)n * 128 * S, G + p.G3 + (size_t)n * 128 * S,
                               G + p.G1 + (size_t)n * 64 * S, 64, d0[0], d0[1], d0[2], stream));
  // double_conv1 (x -> a1 -> conv1)
  NC_TRY(block_bwd(1, G + p.G1, V + p.a1, G + p.G2, G + p.G1));
  return block_bwd(0, G + p.G1, x, G + p.G2, dx);
}

}  // extern "C"

// ---- DeepLinearGenerator ------------------------------------------------------------------------------------------
namespace {

struct LLayer { int C, K, k; };
const LLayer kLL[6] = {{1, 64, 7}, {64, 64, 5}, {64, 64, 3}, {64, 32, 1}, {32, 16, 1}, {16, 1, 1}};

struct LPlan {
  int N, d[3];
  long S;
  size_t w[6], params;        // floats into the packed blob
  size_t act[5], saved;       // outputs of layers 0..4 (inputs of layers 1..5)
  size_t xs3[6];              // S3 copy of the input of layer i (the 5^3 and 3^3 layers), see `kept`
  size_t g[2], grads;         // gradient ping-pong
  size_t conv_ws;
};

// ---- the collapsed tail ----------------------------------------------------------------------------------------------------------------
// Layers 2 .. 5 (3^3 64 -> 64, then 1 x 1: 64 -> 32 -> 16 -> 1; reference networks.py:902-911) have no bias and nothing between them, and the
// 1 x 1 layers need no padding, so with  a = W5 W4 (1 x 32),  e = a W3 (1 x 64)  and  E[c][t] = sum_k e[k] W2[k][c][t]:
//     y = E (*) act1                                   ONE 64 -> 1 convolution, 3^3, padding 1 -- exact at the borders too: the only padded
//                                                      tensor of the four layers is act1 itself
//     dL/dact1 = flip(E) (*) dy                        a 1 -> 64 convolution of the one-channel dy
//     q[c][t]  = sum_v dy[v] act1[c][v + t - 1]        64 x 27 numbers: a weight gradient with ONE "output" channel
//     dW2[k][c][t] = e[k] q[c][t];   r = sum_{c,t} W2[.][c][t] q[c][t]  (= sum_v dy[v] act2[.][v]);   dW3 = a^T r^T;   s = W3 r;
//     dW4 = W5^T s^T;   dW5 = (W4 s)^T
// -- the same output and the same six parameter gradients as the layer-by-layer evaluation (every product of the chain rule is there, the
// rank-one factors are simply never expanded over the voxels), at 2 x 27 x 64 instead of 2 x (27 x 64 x 64 + 64 x 32 + 32 x 16 + 16) MACs per
// voxel and direction, and act2 .. act4 are neither written nor read.  The weight-space products run in fp64.  Numerically this is at least
// as close to the exact result as the fp32 chain (tests/test_gpu_nets.py: against the fp64 oracle, and against the layered path).
// nc_set_dl_collapse(0) / NC_DL_COLLAPSE=0: the layered evaluation.  The forward records its choice in `kept` (bit 31): the backward of a
// collapsed forward is collapsed whatever the switch says by then (act2 .. act4 do not exist).
struct LTail {  // byte offsets into the tail scratch
  static constexpr size_t a = 0, e = 32 * 8, r = e + 64 * 8, E = r + 64 * 8, Ef = E + 1728 * 4, q = Ef + 1728 * 4, P = q + 1728 * 4 + 256,
                          Wf = P + (size_t)64 * 32 * 125 * 4 + 256, bytes = Wf + (size_t)64 * 32 * 125 * 4 + 256;
};

// ---- layer 1's weight gradient from the rank structure of its dY ---------------------------------------------------------------------------
// With the tail collapsed, dL/dact1 = g1 is 64 channels made from ONE: g1[k][v] = sum_a E[k][a] dy[v - (a - 1)] on the volume (27 shifts a).
// Write Dsh[a][v] = dy[v - (a - 1)] for v inside the volume (27 channels, padded to 32): g1 = E . Dsh, and the 5^3 layer's weight gradient
//     dW1[k][c][t] = sum_v g1[k][v] act0[c][v + t - 2] = sum_a E[k][a] P[a][c][t],   P[a][c][t] = sum_v Dsh[a][v] act0[c][v + t - 2]
// -- P is a 5^3 weight gradient between a 32- and a 64-channel tensor: HALF the matrix work of the 64 x 64 one, same kernel (k_wgrad_s3x with
// act0's kept two-term copy in the dY role, which turns P around: Pq[c][a][t'] = P[a][c][124 - t']), then 64 x 64 x 125 x 27 MACs in weight
// space.  Exact: the truncation of g1 to the volume is IN Dsh.  The data gradient of the layer still takes g1 itself.
__global__ void __launch_bounds__(256) k_dl_shift27(const float* __restrict__ dy, float* __restrict__ dsh, int D, int H, int W) {
  const long S = (long)D * H * W;
  const int a = blockIdx.y, n = blockIdx.z;
  float* out = dsh + ((long)n * 32 + a) * S;
  const float* in = dy + (long)n * S;
  const int az = a / 9 - 1, ay = (a / 3) % 3 - 1, ax = a % 3 - 1;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < S; v += (long)gridDim.x * 256) {
    float r = 0.f;
    if (a < 27) {
      const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long)W * H));
      const int sx = x - ax, sy = y - ay, sz = z - az;
      if ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H && (unsigned)sz < (unsigned)D) r = in[((long)sz * H + sy) * W + sx];
    }
    out[v] = r;
  }
}
// The data gradient of the same layer in the same form (used where the convolution kernel takes 32 input channels: the 16-bit path):
//     dL/dact0[c][u] = sum_{k,t} W1[k][c][t] g1[k][u - t + 2] = sum_{a,s} Wf[c][a][s] Dsh[a][u + s - 2],   Wf[c][a][s] = sum_k W1[k][c][124 - s] E[k][a]
// -- a FORWARD 5^3 convolution 32 -> 64 of Dsh with weights composed here (zero for the five padding shifts).
__global__ void __launch_bounds__(128) k_dl_w1_fold(const float* __restrict__ E, const float* __restrict__ w1, float* __restrict__ wf) {
  const int c = blockIdx.x, a = blockIdx.y, sidx = threadIdx.x;
  if (sidx >= 125) return;
  double v = 0.0;
  if (a < 27)
    for (int k = 0; k < 64; ++k) v += (double)w1[((long)k * 64 + c) * 125 + 124 - sidx] * (double)E[k * 27 + a];
  wf[((long)c * 32 + a) * 125 + sidx] = (float)v;
}
// ---- the forward without act1 ---------------------------------------------------------------------------------------------------------------
// Once the backward takes q from P, act1 is wanted by ONE consumer: y = E (*) act1.  With F_t[c][s] = sum_c' E[c'][t] W1[c'][c][s] (27 kernels
// 64 -> 1, 5^3):   y[v] = sum_t [v + t - 1 inside the volume] Z_t[v + t - 1],   Z_t = F_t (*) act0  on the volume
// -- a 5^3 convolution 64 -> 27 (padded to 32: k_conv_s3x K32, half the matrix work of 64 -> 64) and a 27-term shifted sum; the bracket IS
// the zero padding of act1.  act1 is never written; the backward (rank forms above) does not miss it.
__global__ void __launch_bounds__(128) k_dl_fold_fwd(const float* __restrict__ E, const float* __restrict__ w1, float* __restrict__ F) {
  const int t = blockIdx.x, c = blockIdx.y, sidx = threadIdx.x;
  if (sidx >= 125) return;
  double v = 0.0;
  if (t < 27)
    for (int cp = 0; cp < 64; ++cp) v += (double)E[cp * 27 + t] * (double)w1[((long)cp * 64 + c) * 125 + sidx];
  F[((long)t * 64 + c) * 125 + sidx] = (float)v;
}
__global__ void __launch_bounds__(256) k_dl_combine27(const float* __restrict__ Z, float* __restrict__ y, int D, int H, int W, int zch) {
  const long S = (long)D * H * W;
  const int n = blockIdx.y;
  const float* z = Z + (long)n * zch * S;  // (zch: channels per sample of the Z tensor, 32 or 64; the first 27 are read)
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < S; v += (long)gridDim.x * 256) {
    const int x = (int)(v % W), yy = (int)((v / W) % H), zz = (int)(v / ((long)W * H));
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int dz = t / 9 - 1, dy = (t / 3) % 3 - 1, dx = t % 3 - 1;
      if ((unsigned)(x + dx) < (unsigned)W && (unsigned)(yy + dy) < (unsigned)H && (unsigned)(zz + dz) < (unsigned)D)
        acc += z[(long)t * S + v + ((long)dz * H + dy) * W + dx];
    }
    y[(long)n * S + v] = acc;
  }
}

// q itself is a contraction of the same P: q[c'][t] = sum_{c,s} W1[c'][c][s] P[t][c][s] (act1 = W1 (*) act0, and the shift a = t of dy IS the
// mask "v + t - 1 inside the volume") -- no pass over act1.  Written tap-flipped, as k_dl_tail_w2 reads it.  Block (c', t), fixed-order tree.
__global__ void __launch_bounds__(256) k_dl_q_from_p(const float* __restrict__ w1, const float* __restrict__ Pq, float* __restrict__ qf) {
  __shared__ double red[256];
  const int cp = blockIdx.x, t = blockIdx.y, th = threadIdx.x;
  double v = 0.0;
  for (int i = th; i < 64 * 125; i += 256) {
    const int c = i / 125, sidx = i - c * 125;
    v += (double)w1[((long)cp * 64 + c) * 125 + sidx] * (double)Pq[((long)c * 32 + t) * 125 + 124 - sidx];
  }
  red[th] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (th < o) red[th] += red[th + o];
    __syncthreads();
  }
  if (th == 0) qf[cp * 27 + 26 - t] = (float)red[0];
}
// dW1[k][c][t] = sum_a E[k][a] Pq[c][a][124 - t]
__global__ void __launch_bounds__(128) k_dl_w1_contract(const float* __restrict__ E, const float* __restrict__ Pq, float* __restrict__ dw1) {
  const int k = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
  if (t >= 125) return;
  double v = 0.0;
#pragma unroll
  for (int a = 0; a < 27; ++a) v += (double)E[k * 27 + a] * (double)Pq[((long)c * 32 + a) * 125 + 124 - t];
  dw1[((long)k * 64 + c) * 125 + t] = (float)v;
}
constexpr unsigned kKeptCollapsed = 1u << 31;
constexpr unsigned kKeptNoAct1 = 1u << 30;
constexpr unsigned kKeptTyped = 1u << 29;  // ... layers 1 .. 5 ran as ONE position-typed 7^3 kernel (dl_typed.hip): neither act1 nor a two-term copy of act0 exists  // ... and layer 1 ran in its 64 -> 27 form: act1 was never written ("the forward without act1")

// (8 workgroups, each derives a and e for itself and 216 of E's 1728 entries; loops unrolled so that the loads of a sum are in flight together)
__global__ void __launch_bounds__(256) k_dl_compose(const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ w4,
                                                    const float* __restrict__ w5, char* __restrict__ tail) {
  __shared__ double sa[32], se[64];
  const int t = threadIdx.x;
  if (t < 32) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += (double)w5[i] * (double)w4[i * 32 + t];
    sa[t] = v;
    if (blockIdx.x == 0) ((double*)(tail + LTail::a))[t] = v;
  }
  __syncthreads();
  if (t < 64) {
    double v = 0.0;
#pragma unroll
    for (int b = 0; b < 32; ++b) v += sa[b] * (double)w3[b * 64 + t];
    se[t] = v;
    if (blockIdx.x == 0) ((double*)(tail + LTail::e))[t] = v;
  }
  __syncthreads();
  if (t < 216) {
    const int i = blockIdx.x * 216 + t;  // i = c * 27 + tap
    double v = 0.0;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) v += se[k] * (double)w2[(long)k * 1728 + i];
    const int c = i / 27, tap = i - c * 27;
    ((float*)(tail + LTail::E))[i] = (float)v;
    ((float*)(tail + LTail::Ef))[c * 27 + 26 - tap] = (float)v;
  }
}

// block k: dW2[k][.][.] and r[k].  qf = the weight gradient of the SWAPPED problem (x := dy, dY := act1): qf[c][t] = q[c][26 - t]
__global__ void __launch_bounds__(256) k_dl_tail_w2(const float* __restrict__ w2, char* __restrict__ tail, float* __restrict__ dw2) {
  __shared__ double red[256];
  const int k = blockIdx.x, t = threadIdx.x;
  const float* qf = (const float*)(tail + LTail::q);
  const double ek = ((const double*)(tail + LTail::e))[k];
  double part = 0.0;
  for (int i = t; i < 1728; i += 256) {
    const int c = i / 27, tap = i - c * 27;
    const double q = (double)qf[c * 27 + 26 - tap];
    dw2[(long)k * 1728 + i] = (float)(ek * q);
    part += (double)w2[(long)k * 1728 + i] * q;
  }
  red[t] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) ((double*)(tail + LTail::r))[k] = red[0];
}

__global__ void __launch_bounds__(256) k_dl_tail_w345(const float* __restrict__ w3, const float* __restrict__ w4, const float* __restrict__ w5,
                                                      const char* __restrict__ tail, float* __restrict__ dw3, float* __restrict__ dw4,
                                                      float* __restrict__ dw5) {
  __shared__ double ss[32];
  const int t = threadIdx.x;
  const double* a = (const double*)(tail + LTail::a);
  const double* r = (const double*)(tail + LTail::r);
  for (int i = t; i < 32 * 64; i += 256) dw3[i] = (float)(a[i / 64] * r[i % 64]);
  if (t < 32) {
    double v = 0.0;
    for (int c = 0; c < 64; ++c) v += (double)w3[t * 64 + c] * r[c];
    ss[t] = v;
  }
  __syncthreads();
  for (int i = t; i < 16 * 32; i += 256) dw4[i] = (float)((double)w5[i / 32] * ss[i % 32]);
  if (t < 16) {
    double v = 0.0;
    for (int b = 0; b < 32; ++b) v += (double)w4[t * 32 + b] * ss[b];
    dw5[t] = (float)v;
  }
}

// 0: layer by layer; 1: the collapsed tail + the rank forms of the 5^3 layer (round 5); 2 (default): layers 1 .. 5 as one position-typed 7^3 kernel
// where the shape admits it (dl_typed.hip), else as 1
std::atomic<int> g_dl_collapse{getenv("NC_DL_COLLAPSE") ? (atoi(getenv("NC_DL_COLLAPSE")) < 0 ? 0 : atoi(getenv("NC_DL_COLLAPSE")) > 2 ? 2 : atoi(getenv("NC_DL_COLLAPSE"))) : 2};

}  // namespace

// the same weight-space steps for the 16-bit path (gen_nets_lp.hip)
namespace nc {
bool dl_collapse_on() { return g_dl_collapse.load(std::memory_order_relaxed) != 0; }
size_t dl_tail_bytes() { return LTail::bytes; }
const float* dl_tail_E(const char* tail) { return (const float*)(tail + LTail::E); }
const float* dl_tail_Ef(const char* tail) { return (const float*)(tail + LTail::Ef); }
float* dl_tail_q(char* tail) { return (float*)(tail + LTail::q); }
int dl_tail_compose(const float* w2, const float* w3, const float* w4, const float* w5, char* tail, hipStream_t s) {
  hipLaunchKernelGGL(k_dl_compose, dim3(8), dim3(256), 0, s, w2, w3, w4, w5, tail);
  return check_launch("deep_linear: compose");
}
float* dl_tail_P(char* tail) { return (float*)(tail + LTail::P); }
const float* dl_w1_fold(char* tail, const float* w1, hipStream_t s) {  // composes Wf into the tail scratch and returns it (NULL: launch failed)
  hipLaunchKernelGGL(k_dl_w1_fold, dim3(64, 32), dim3(128), 0, s, (const float*)(tail + LTail::E), w1, (float*)(tail + LTail::Wf));
  return check_launch("deep_linear_bwd: composed data-gradient weights") ? nullptr : (const float*)(tail + LTail::Wf);
}
// the forward without act1 for the 16-bit path: F as a [64][64][125] tensor (rows 27 .. 63 zero) in the tail scratch (the P and Wf regions,
// adjacent, 2 MB + 512 B; free in a forward), and the shifted sum over a Z tensor of `zch` channels per sample
const float* dl_fold_fwd64(char* tail, const float* w1, hipStream_t s) {
  static_assert(LTail::Wf == LTail::P + (size_t)64 * 32 * 125 * 4 + 256, "P and Wf adjacent");
  float* F = (float*)(tail + LTail::P);
  if (hipMemsetAsync(F + (size_t)32 * 64 * 125, 0, (size_t)32 * 64 * 125 * 4, s) != hipSuccess) return nullptr;
  hipLaunchKernelGGL(k_dl_fold_fwd, dim3(32, 64), dim3(128), 0, s, (const float*)(tail + LTail::E), w1, F);
  return check_launch("deep_linear_fwd: composed forward weights") ? nullptr : F;
}
int dl_combine27(const float* Z, float* y, int N, int D, int H, int W, int zch, hipStream_t s) {
  hipLaunchKernelGGL(k_dl_combine27, dim3(1024, (unsigned)N), dim3(256), 0, s, Z, y, D, H, W, zch);
  return check_launch("deep_linear_fwd: shifted sum");
}
int dl_q_from_p(char* tail, const float* w1, hipStream_t s) {
  hipLaunchKernelGGL(k_dl_q_from_p, dim3(64, 27), dim3(256), 0, s, w1, (const float*)(tail + LTail::P), (float*)(tail + LTail::q));
  return check_launch("deep_linear_bwd: q");
}
int dl_w1_contract(const char* tail, float* dw1, hipStream_t s) {
  hipLaunchKernelGGL(k_dl_w1_contract, dim3(64, 64), dim3(128), 0, s, (const float*)(tail + LTail::E), (const float*)(tail + LTail::P), dw1);
  return check_launch("deep_linear_bwd: dW1");
}
int dl_tail_grads(const float* w2, const float* w3, const float* w4, const float* w5, char* tail, float* dw2, float* dw3, float* dw4, float* dw5,
                  hipStream_t s) {
  hipLaunchKernelGGL(k_dl_tail_w2, dim3(64), dim3(256), 0, s, w2, tail, dw2);
  hipLaunchKernelGGL(k_dl_tail_w345, dim3(1), dim3(256), 0, s, w3, w4, w5, (const char*)tail, dw3, dw4, dw5);
  return check_launch("deep_linear: tail gradients");
}
}  // namespace nc

namespace {

bool l_plan(LPlan& p, int N, int S0, int S1, int S2) {
  if (N < 1 || S0 < 1 || S1 < 1 || S2 < 1) return false;
  p = LPlan{};
  p.N = N; p.d[0] = S0; p.d[1] = S1; p.d[2] = S2;
  p.S = (long)S0 * S1 * S2;
  size_t off = 0;
  for (int i = 0; i < 6; ++i) { p.w[i] = off; off += (size_t)kLL[i].K * kLL[i].C * kLL[i].k * kLL[i].k * kLL[i].k; }
  p.params = off;
  off = 0;
  for (int i = 0; i < 5; ++i) { p.act[i] = off; off += up64((size_t)N * kLL[i].K * p.S); }
  for (int i = 0; i < 6; ++i) {
    p.xs3[i] = off;
    if (kLL[i].k > 1 && kLL[i].C % 8 == 0) off += s3_floats((size_t)N * kLL[i].C * p.S);
  }
  p.saved = off;
  p.g[0] = 0; p.g[1] = up64((size_t)N * 64 * p.S);
  p.grads = 2 * up64((size_t)N * 64 * p.S);
  for (int i = 0; i < 6; ++i) {
    const size_t b = nc_conv_ws_bytes(N, kLL[i].C, S0, S1, S2, kLL[i].K, kLL[i].k, kLL[i].k, kLL[i].k, 1, kLL[i].k / 2);
    if (b > p.conv_ws) p.conv_ws = b;
  }
  const size_t b1 = nc_conv_ws_bytes(N, 1, S0, S1, S2, 64, 3, 3, 3, 1, 1);  // the collapsed tail's two one-channel problems
  if (b1 > p.conv_ws) p.conv_ws = b1;
  if (conv_fwd_to1_k3_ws_bytes(N, S0, S1, S2) > p.conv_ws) p.conv_ws = conv_fwd_to1_k3_ws_bytes(N, S0, S1, S2);
  return true;
}

}  // namespace

extern "C" {

size_t nc_deep_linear_param_floats(void) {
  LPlan p;
  l_plan(p, 1, 8, 8, 8);
  return p.params;
}

size_t nc_deep_linear_saved_floats(int N, int S0, int S1, int S2) {
  LPlan p;
  return l_plan(p, N, S0, S1, S2) ? p.saved : 0;
}

size_t nc_deep_linear_ws_bytes(int N, int S0, int S1, int S2) {
  LPlan p;
  return l_plan(p, N, S0, S1, S2) ? align256(p.conv_ws) + p.grads * sizeof(float) + 256 + align256(LTail::bytes) + dl_typed_bytes(N, S0, S1, S2) : 0;
}

void nc_set_dl_collapse(int on) { g_dl_collapse.store(on < 0 ? 0 : on > 2 ? 2 : on, std::memory_order_relaxed); }
int nc_get_dl_collapse(void) { return g_dl_collapse.load(std::memory_order_relaxed); }

// saved == NULL: inference -- the intermediate activations ping-pong through the workspace instead
int nc_deep_linear_fwd(const float* params, const float* x, float* y, float* saved, int N, int S0, int S1, int S2, void* ws,
                       size_t ws_bytes, void* stream, unsigned* kept) {
  NetworkScope net_scope;
  if (!params || !x || !y) { set_error("deep_linear_fwd: null pointer"); return NC_ERR_ARG; }
  LPlan p;
  if (!l_plan(p, N, S0, S1, S2)) { set_error("deep_linear_fwd: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < nc_deep_linear_ws_bytes(N, S0, S1, S2)) { set_error("deep_linear_fwd: workspace too small"); return NC_ERR_WS; }
  void* cws = ws;
  float* G = (float*)((char*)ws + align256(p.conv_ws));
  const float* in = x;
  unsigned kept_mask = 0;
  if (kept) *kept = 0;
  const bool collapse = g_dl_collapse.load(std::memory_order_relaxed) != 0;
  char* tail = (char*)ws + align256(p.conv_ws) + p.grads * sizeof(float) + 256;
  hipStream_t hs = (hipStream_t)stream;
  static const bool k32_on = !(getenv("NC_DL_K32") && atoi(getenv("NC_DL_K32")) == 0);
  char* typed = tail + align256(LTail::bytes);
  ConvDims d0;
  // layers 1 .. 5 as ONE position-typed 7^3 convolution 64 -> 1 of act0 (dl_typed.hip): the interior kernel on the two-term
  // pseudo-channel kernel of Conv3d(1, 64, 7)'s data gradient, the voxels on a face recomputed with their own kernels
  const bool typed_ok = g_dl_collapse.load(std::memory_order_relaxed) == 2 && saved && dl_typed_supported(N, S0, S1, S2) && make_dims(d0, N, 1, S0, S1, S2, 64, 7, 7, 7, 1, 3) && c1k7_h2_supported(d0) &&
                        c1_wgrad_supported(d0) && c1k7_h2_dgrad_ws_bytes(d0) <= p.conv_ws && c1k7_h2_ws_bytes(d0) <= p.conv_ws &&
                        c1_wgrad_ws_bytes(d0) <= p.conv_ws;
  for (int i = 0; i < 6; ++i) {
    ConvDims dk;
    if (typed_ok && i == 1) {
      hipLaunchKernelGGL(k_dl_compose, dim3(8), dim3(256), 0, hs, params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail);
      hipLaunchKernelGGL(k_dl_fold_fwd, dim3(32, 64), dim3(128), 0, hs, (const float*)(tail + LTail::E), params + p.w[1], (float*)(tail + LTail::Wf));
      NC_TRY(check_launch("deep_linear_fwd: composed kernels"));
      NC_TRY(dl_typed_compose((const float*)(tail + LTail::Wf), typed, N, S0, S1, S2, hs));
      NC_TRY(conv_c1k7_h2_dgrad(in, dl_typed_wp(typed, N, S0, S1, S2), y, d0, cws, p.conv_ws, hs));
      NC_TRY(dl_typed_fwd_boundary(in, y, typed, N, S0, S1, S2, hs));
      kept_mask |= kKeptCollapsed | kKeptNoAct1 | kKeptTyped;
      break;
    }
    if (collapse && i == 1 && saved && k32_on && make_dims(dk, N, 64, S0, S1, S2, 32, 5, 5, 5, 1, 2) && conv_fwd_h2_k32_supported(dk) &&
        (size_t)256 + s3x_packed_bytes(64, 32, 5, 2) + 512 <= p.conv_ws) {
      // layers 1 .. 5 without act1 ("the forward without act1" above): Z = the 64 -> 27 convolution of act0, y = its shifted sum
      hipLaunchKernelGGL(k_dl_compose, dim3(8), dim3(256), 0, hs, params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail);
      hipLaunchKernelGGL(k_dl_fold_fwd, dim3(32, 64), dim3(128), 0, hs, (const float*)(tail + LTail::E), params + p.w[1], (float*)(tail + LTail::Wf));
      NC_TRY(check_launch("deep_linear_fwd: composed forward weights"));
      float* Z = saved + p.act[1];  // (the slot act1 would take: 32 of its 64 channels)
      NC_TRY(conv_fwd_h2_k32_keep(in, (const float*)(tail + LTail::Wf), Z, dk, cws, p.conv_ws, hs, saved + p.xs3[1]));
      hipLaunchKernelGGL(k_dl_combine27, dim3(1024, (unsigned)N), dim3(256), 0, hs, (const float*)Z, y, S0, S1, S2, 32);
      NC_TRY(check_launch("deep_linear_fwd: shifted sum"));
      kept_mask |= kKeptCollapsed | kKeptNoAct1 | (1u << 1) | (1u << 17);
      break;
    }
    if (collapse && i == 2) {  // layers 2 .. 5 as one 64 -> 1 convolution of act1 (see "the collapsed tail" above)
      hipLaunchKernelGGL(k_dl_compose, dim3(8), dim3(256), 0, hs, params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail);
      NC_TRY(check_launch("deep_linear_fwd: compose"));
      NC_TRY(conv_fwd_to1_k3(in, (const float*)(tail + LTail::E), y, N, 64, S0, S1, S2, cws, p.conv_ws, hs));
      kept_mask |= kKeptCollapsed;
      break;
    }
    const LLayer& l = kLL[i];
    float* out = i == 5 ? y : (saved ? saved + p.act[i] : G + p.g[i & 1]);
    bool kept = false;
    void* keep = saved && l.k > 1 && l.C % 8 == 0 ? (void*)(saved + p.xs3[i]) : nullptr;
    if (l.k > 1)
      NC_TRY(conv_fwd_keep(in, params + p.w[i], nullptr, out, N, l.C, S0, S1, S2, l.K, l.k, cws, p.conv_ws, stream, keep, &kept));
    else
      NC_TRY(nc_conv_fwd(in, params + p.w[i], nullptr, out, N, l.C, S0, S1, S2, l.K, l.k, l.k, l.k, 1, l.k / 2, cws, p.conv_ws, stream));
    if (kept) kept_mask |= 1u << i;
    ConvDims cdl;
    if (kept && make_dims(cdl, N, l.C, S0, S1, S2, l.K, l.k, l.k, l.k, 1, l.k / 2) && conv_layer_h2(cdl)) kept_mask |= 1u << (16 + i);
    in = out;
  }
  if (kept) *kept = kept_mask;
  return NC_OK;
}

// dx nullable; dparams overwritten
int nc_deep_linear_bwd(const float* params, const float* x, const float* saved, const float* dy, float* dx, float* dparams,
                       int N, int S0, int S1, int S2, void* ws, size_t ws_bytes, void* stream, unsigned kept) {
  NetworkScope net_scope;
  if (!params || !x || !saved || !dy || !dparams) { set_error("deep_linear_bwd: null pointer"); return NC_ERR_ARG; }
  LPlan p;
  if (!l_plan(p, N, S0, S1, S2)) { set_error("deep_linear_bwd: bad shape"); return NC_ERR_SHAPE; }
  if (!ws || ws_bytes < nc_deep_linear_ws_bytes(N, S0, S1, S2)) { set_error("deep_linear_bwd: workspace too small"); return NC_ERR_WS; }
  void* cws = ws;
  float* G = (float*)((char*)ws + align256(p.conv_ws));
  const float* g = dy;
  const unsigned kept_mask = kept;
  int top = 5;
  // layer 1's backward from the rank structure of its dY (see k_dl_shift27): decided here, because with both halves in that form dL/dact1 is
  // never formed
  ConvDims dsh, cd1;
  static const bool rank_on = !(getenv("NC_DL_RANK_WGRAD") && atoi(getenv("NC_DL_RANK_WGRAD")) == 0);
  static const bool rank_dg_on = !(getenv("NC_DL_RANK_DGRAD") && atoi(getenv("NC_DL_RANK_DGRAD")) == 0);
  const bool h2_1 = make_dims(cd1, N, 64, S0, S1, S2, 64, 5, 5, 5, 1, 2) && conv_layer_h2(cd1);
  const bool have1 = ((kept_mask >> 1) & 1) && ((kept_mask >> 17) & 1) != 0;
  const bool rank_w = (kept_mask & kKeptCollapsed) && rank_on && h2_1 && have1 && make_dims(dsh, N, 32, S0, S1, S2, 64, 5, 5, 5, 1, 2) &&
                      wgrad_h2_supported(dsh) && s3_wgrad_ws_bytes(dsh) <= p.conv_ws;
  const bool rank_dgrad = rank_w && rank_dg_on && conv_fwd_h2_c32_supported(dsh) && conv_fwd_h2_c32_ws_bytes(dsh) <= p.conv_ws;
  const bool rank_all = rank_dgrad;
  if (kept_mask & kKeptTyped) {  // layers 1 .. 5 ran as one position-typed 7^3 kernel (dl_typed.hip): their whole backward from dy, act0 and the weights
    hipStream_t hs = (hipStream_t)stream;
    char* tail = (char*)ws + align256(p.conv_ws) + p.grads * sizeof(float) + 256;
    char* typed = tail + align256(LTail::bytes);
    ConvDims d0;
    // (its kernels exist in the two-term form only and convert their fp32 operands themselves: a caller who moved nc_set_split_terms after the
    // forward still gets this form's backward)
    ForceTwoTerm two_;
    if (!dl_typed_supported(N, S0, S1, S2) || !make_dims(d0, N, 1, S0, S1, S2, 64, 7, 7, 7, 1, 3) || !c1k7_h2_supported(d0) || !c1_wgrad_supported(d0)) {
      set_error("deep_linear_bwd: the forward ran the position-typed form, which this shape does not admit");
      return NC_ERR_ARG;
    }
    const float* act0 = saved + p.act[0];
    hipLaunchKernelGGL(k_dl_compose, dim3(8), dim3(256), 0, hs, params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail);
    hipLaunchKernelGGL(k_dl_fold_fwd, dim3(32, 64), dim3(128), 0, hs, (const float*)(tail + LTail::E), params + p.w[1], (float*)(tail + LTail::Wf));
    NC_TRY(check_launch("deep_linear_bwd: composed kernels"));
    NC_TRY(dl_typed_compose((const float*)(tail + LTail::Wf), typed, N, S0, S1, S2, hs));
    // dL/dact0: the 1 -> 64 convolution of dy with the interior kernel (the pseudo-channel forward kernel), then the boundary voxels' differences
    float* g0 = G + p.g[1];
    NC_TRY(conv_c1k7_h2(dy, dl_typed_wp(typed, N, S0, S1, S2), nullptr, g0, d0, cws, p.conv_ws, hs));
    NC_TRY(dl_typed_dgrad_boundary(act0, dy, g0, typed, N, S0, S1, S2, hs));
    // every parameter gradient through dH: the full correlation of dy and act0 (the one-channel 7^3 weight gradient with the operands' roles swapped)
    // and the boundary types' sums -> P, then weight space as in the rank forms
    NC_TRY(conv_wgrad_c1(dy, act0, dl_typed_dwsw(typed, N, S0, S1, S2), d0, cws, p.conv_ws, hs));
    NC_TRY(dl_typed_p(act0, dy, (float*)(tail + LTail::P), typed, N, S0, S1, S2, hs));
    hipLaunchKernelGGL(k_dl_q_from_p, dim3(64, 27), dim3(256), 0, hs, params + p.w[1], (const float*)(tail + LTail::P), (float*)(tail + LTail::q));
    hipLaunchKernelGGL(k_dl_w1_contract, dim3(64, 64), dim3(128), 0, hs, (const float*)(tail + LTail::E), (const float*)(tail + LTail::P), dparams + p.w[1]);
    hipLaunchKernelGGL(k_dl_tail_w2, dim3(64), dim3(256), 0, hs, params + p.w[2], tail, dparams + p.w[2]);
    hipLaunchKernelGGL(k_dl_tail_w345, dim3(1), dim3(256), 0, hs, params + p.w[3], params + p.w[4], params + p.w[5], (const char*)tail, dparams + p.w[3],
                       dparams + p.w[4], dparams + p.w[5]);
    NC_TRY(check_launch("deep_linear_bwd: typed parameter gradients"));
    g = g0;
    top = 0;
  } else if (kept_mask & kKeptCollapsed) {  // the forward left no act2 .. act4: layers 2 .. 5 from dy, act1 and the weights alone
    hipStream_t hs = (hipStream_t)stream;
    char* tail = (char*)ws + align256(p.conv_ws) + p.grads * sizeof(float) + 256;
    const float* act1 = saved + p.act[1];  // (kKeptNoAct1: not there)
    hipLaunchKernelGGL(k_dl_compose, dim3(8), dim3(256), 0, hs, params + p.w[2], params + p.w[3], params + p.w[4], params + p.w[5], tail);
    NC_TRY(check_launch("deep_linear_bwd: compose"));
    if (rank_w) {
      // the 32 x 64 weight gradient P between the 27 shifted copies of dy and act0 (see k_dl_shift27): dW1 and q are both contractions of it
      float* Dsh = G + p.g[1];  // (free until layer 1's data gradient is written there, below)
      hipLaunchKernelGGL(k_dl_shift27, dim3(256, 32, (unsigned)N), dim3(256), 0, hs, dy, Dsh, S0, S1, S2);
      NC_TRY(check_launch("deep_linear_bwd: shifted copies of dy"));
      NC_TRY(conv_wgrad_h2(Dsh, nullptr, nullptr, saved + p.xs3[1], (float*)(tail + LTail::P), dsh, cws, p.conv_ws, hs));
      hipLaunchKernelGGL(k_dl_q_from_p, dim3(64, 27), dim3(256), 0, hs, params + p.w[1], (const float*)(tail + LTail::P), (float*)(tail + LTail::q));
      hipLaunchKernelGGL(k_dl_w1_contract, dim3(64, 64), dim3(128), 0, hs, (const float*)(tail + LTail::E), (const float*)(tail + LTail::P), dparams + p.w[1]);
      NC_TRY(check_launch("deep_linear_bwd: q, dW1"));
    } else {
      if (kept_mask & kKeptNoAct1) {  // (the forward never wrote act1 and the switches changed since: form it now, in a gradient buffer)
        NC_TRY(nc_conv_fwd(saved + p.act[0], params + p.w[1], nullptr, G + p.g[1], N, 64, S0, S1, S2, 64, 5, 5, 5, 1, 2, cws, p.conv_ws, stream));
        act1 = G + p.g[1];
      }
      // q (tap-flipped): the weight gradient of Conv3d(1, 64, 3) with x := dy and dY := act1
      NC_TRY(nc_conv_wgrad(dy, act1, (float*)(tail + LTail::q), nullptr, N, 1, S0, S1, S2, 64, 3, 3, 3, 1, 1, cws, p.conv_ws, stream));
    }
    hipLaunchKernelGGL(k_dl_tail_w2, dim3(64), dim3(256), 0, hs, params + p.w[2], tail, dparams + p.w[2]);
    hipLaunchKernelGGL(k_dl_tail_w345, dim3(1), dim3(256), 0, hs, params + p.w[3], params + p.w[4], params + p.w[5], (const char*)tail, dparams + p.w[3],
                       dparams + p.w[4], dparams + p.w[5]);
    NC_TRY(check_launch("deep_linear_bwd: tail gradients"));
    // dL/dact1 = flip(E) (*) dy -- unless layer 1's whole backward runs from the shifted copies of dy (below), which never looks at it
    if (!rank_all) NC_TRY(nc_conv_fwd(dy, (const float*)(tail + LTail::Ef), nullptr, G + p.g[0], N, 1, S0, S1, S2, 64, 3, 3, 3, 1, 1, cws, p.conv_ws, stream));
    g = G + p.g[0];
    top = 1;
  }
  for (int i = top; i >= 0; --i) {
    const LLayer& l = kLL[i];
    const float* in = i == 0 ? x : saved + p.act[i - 1];
    float* gin = i == 0 ? dx : G + p.g[i & 1];
    ConvDims cdl;
    const bool now_h2 = make_dims(cdl, N, l.C, S0, S1, S2, l.K, l.k, l.k, l.k, 1, l.k / 2) && conv_layer_h2(cdl);
    const bool have = ((kept_mask >> i) & 1) && (((kept_mask >> (16 + i)) & 1) != 0) == now_h2;
    if (i == 1 && rank_w && now_h2 && have) {  // the 32 x 64 problems of "layer 1's weight gradient from the rank structure of its dY" (above)
      hipStream_t hs = (hipStream_t)stream;
      char* tail = (char*)ws + align256(p.conv_ws) + p.grads * sizeof(float) + 256;
      float* Dsh = G + p.g[1];  // (dW1 came out of P in the prologue; the shifted copies are still there)
      // ... and its data gradient the same way where the planner covers it: a FORWARD 32 -> 64 convolution of Dsh with the composed weights
      // Wf (k_dl_w1_fold) -- half the matrix work again, and g1 itself is not needed at all (rank_g1 below)
      if (rank_dgrad) {
        if (gin) {
          hipLaunchKernelGGL(k_dl_w1_fold, dim3(64, 32), dim3(128), 0, hs, (const float*)(tail + LTail::E), params + p.w[1], (float*)(tail + LTail::Wf));
          NC_TRY(check_launch("deep_linear_bwd: composed data-gradient weights"));
          NC_TRY(conv_fwd_h2_c32(Dsh, (const float*)(tail + LTail::Wf), gin, dsh, cws, p.conv_ws, hs));  // (converts Dsh into cws first: gin may be its buffer)
        }
      } else {
        NC_TRY(nc_conv_dgrad(g, params + p.w[i], gin, N, l.C, S0, S1, S2, l.K, l.k, l.k, l.k, 1, l.k / 2, cws, p.conv_ws, stream));
      }
      g = gin;
      continue;
    }
    if (l.k > 1)
      NC_TRY(conv_bwd_keep(in, have ? (const void*)(saved + p.xs3[i]) : nullptr, g, params + p.w[i], gin, dparams + p.w[i],
                           N, l.C, S0, S1, S2, l.K, l.k, cws, p.conv_ws, stream));
    else
      NC_TRY(nc_conv_bwd(in, g, params + p.w[i], gin, dparams + p.w[i], nullptr, N, l.C, S0, S1, S2, l.K, l.k, l.k, l.k, 1, l.k / 2, cws,
                         p.conv_ws, stream));
    g = gin;
  }
  return NC_OK;
}

}  // extern "C"
