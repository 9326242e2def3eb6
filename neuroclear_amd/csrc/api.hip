// C-ABI glue: error reporting, convolution dispatch (MFMA implicit GEMM vs direct), and the whole-network forward of
// Unet_deconv used by diced inference (reference models/networks.py:512-538 via models/test_model.py:60-62).
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>
#include <mutex>
#include <set>
#include <utility>

#include "common.hpp"

namespace nc {

static thread_local char g_err[512] = "";
int g_force_direct = 0;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int raise_dyn_lds(const void* kernel, int bytes, const char* who) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("%s: hipGetDevice failed", who); return NC_ERR_HIP; }
  std::lock_guard<std::mutex> g(mu);
  if (done.count({kernel, dev})) return NC_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: cannot raise dynamic LDS limit to %d bytes", who, bytes);
    return NC_ERR_HIP;
  }
  done.insert({kernel, dev});
  return NC_OK;
}


// ---- live launch profiler (bench.py): HIP events on the launch stream around every convolution entry point --------
struct ProfRec { int cls; double flop, abytes; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;  // launches come from the main thread (forward) and the autograd engine's thread (backward)
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;
static std::atomic<bool> g_prof_on{false};
static double g_prof_min_flop = 0.0;

ProfScope::ProfScope(int op, int path, const ConvDims& d, int lp, hipStream_t stream) : idx(-1), s(stream) {
  if (!g_prof_on) return;
  const double flop = 2.0 * d.C * d.K * d.kd * d.kh * d.kw * ((double)d.N * d.Do * d.Ho * d.Wo);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (flop < g_prof_min_flop) return;
  auto get = [&]() {
    hipEvent_t e;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return (hipEvent_t) nullptr;
    return e;
  };
  // algorithmic bytes of the launch: every operand element once (4 B: fp32, or the two-term H2 form; 2 B on the 16-bit path; the three-term
  // form's 6 B is not the algorithm's doing) + the result + the weights
  const double xin = (double)d.N * d.C * d.D * d.H * d.W, yout = (double)d.N * d.K * d.Do * d.Ho * d.Wo, wts = (double)d.K * d.C * d.kd * d.kh * d.kw;
  const double eb = lp ? 2.0 : 4.0;
  const double abytes = op == 2 ? (xin + yout) * eb + wts * 4.0 : (op == 0 ? xin * eb + yout * 4.0 : yout * eb + xin * 4.0) + wts * 4.0;
  ProfRec r{op | (path << 4) | (d.kh << 8) | (lp ? 1 << 16 : 0), flop, abytes, get(), get()};
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, s);
  g_prof.push_back(r);
  idx = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (idx < (int)g_prof.size()) (void)hipEventRecord(g_prof[idx].e1, s);
}

// NC_SCONV=0 switches the image-staged kernels of the PatchGAN layers off (A/B runs against the gather GEMM)
static bool sconv_on(const ConvDims& d, int dgrad) {
  static const bool on = !(getenv("NC_SCONV") && atoi(getenv("NC_SCONV")) == 0);
  return on && (dgrad ? sconv_dgrad_supported(d) : sconv_fwd_supported(d));
}
// NC_PG1=0: the PatchGAN's first layer back on the generic kernels (A/B runs)
static bool pg1_on(const ConvDims& d) {
  static const bool on = !(getenv("NC_PG1") && atoi(getenv("NC_PG1")) == 0);
  return on && pg1_supported(d);
}
// Split-operand kernels (conv_split.hip): the fp32 3^3 / 5^3 convolutions they cover run as six bf16 MFMA products of an exact
// three-term operand split (same operands, fp32 accumulation, error against fp64 not above the fp32 MFMA kernel's:
// tests/test_gpu_split.py).  On by default; nc_set_conv_split(0) or NC_CONV_SPLIT=0 put those layers back on the fp32 MFMA kernels.
static std::atomic<int> g_split{getenv("NC_CONV_SPLIT") ? atoi(getenv("NC_CONV_SPLIT")) : 1};  // process-wide; read once per call
// NC_S3_FUSE=0 (nc_set_s3_fusion(0)): nc_unet_deconv_fwd converts every convolution input to S3 in a separate pass (A/B, tests)
static std::atomic<int> g_s3_fuse{getenv("NC_S3_FUSE") ? atoi(getenv("NC_S3_FUSE")) : 1};
static int fwd_path(const ConvDims& d) {
  if (g_force_direct) return 0;
  if (g_split && s3_fwd_supported(d)) return 9;
  if (g_split && p2d_fwd_supported(d)) return 10;
  if (g_split && c1k7_h2_supported(d)) return 11;  // Conv3d(1, 64, 7) in pseudo-channel form on the two-term kernels (conv_s3x.hip)
  return (c1k3_fwd_supported(d) || mfma_fwd_supported(d)) ? 1 : flat_1x1_supported(d) ? 3 : k1_fwd_supported(d) ? 5 : pg1_on(d) ? 8 : sconv_on(d, 0) ? 7 : gemm_fwd_supported(d) ? 2 : 0;
}
static int dgrad_path(const ConvDims& d) {
  if (g_force_direct) return 0;
  if (g_split && s3_dgrad_supported(d)) return 9;
  if (g_split && p2d_dgrad_supported(d)) return 10;
  if (g_split && c1k7_h2_supported(d)) return 11;  // Conv3d(1, 64, 7): two-term pseudo-channel form + fold (conv_s3x.hip)
  return mfma_dgrad_supported(d) ? 1 : flat_1x1_supported(d) ? 3 : to1_mfma_supported(d) ? 6 : to1_dgrad_supported(d) ? 0
                                                                                              : pg1_on(d)             ? 8
                                                                                              : sconv_on(d, 1)        ? 7
                                                                                              : gemm_dgrad_supported(d) ? 2 : 0;
}
static bool sconv_wgrad_on(const ConvDims& d) {
  static const bool on = !(getenv("NC_SCONV") && atoi(getenv("NC_SCONV")) == 0) &&
                         !(getenv("NC_SCONV_WGRAD") && atoi(getenv("NC_SCONV_WGRAD")) == 0);
  return on && sconv_wgrad_supported(d);
}
static int wgrad_path(const ConvDims& d) {
  if (g_force_direct) return 0;
  if (g_split && s3_wgrad_supported(d)) return 9;
  if (g_split && p2d_wgrad_supported(d)) return 10;
  return mfma_wgrad_supported(d) ? 1 : wgrad_1x1_supported(d) ? 3 : c1_wgrad_supported(d) ? 4 : k1_wgrad_supported(d) ? 5
                                                                                            : pg1_on(d)              ? 8
                                                                                            : sconv_wgrad_on(d)      ? 7
                                                                                            : gemm_wgrad_supported(d) ? 2 : 0;
}

}  // namespace nc

using namespace nc;

extern "C" {

void nc_prof_begin(double min_flop) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ProfRec& r : g_prof) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  g_prof.clear();
  g_prof_min_flop = min_flop;
  g_prof_on = true;
}

// Stops recording, waits for the recorded launches and returns how many there were; the first `max` are written to
// cls / flop / ms (cls encoding: see ProfScope).  The caller synchronises the device first.
int nc_prof_end(int max, int* cls, double* flop, float* ms) { return nc_prof_end2(max, cls, flop, ms, nullptr); }
int nc_prof_end2(int max, int* cls, double* flop, float* ms, double* abytes) {
  g_prof_on = false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const int n = (int)g_prof.size();
  for (int i = 0; i < n && i < max; ++i) {
    float t = 0.f;
    (void)hipEventSynchronize(g_prof[i].e1);
    (void)hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1);
    cls[i] = g_prof[i].cls; flop[i] = g_prof[i].flop; ms[i] = t;
    if (abytes) abytes[i] = g_prof[i].abytes;
  }
  return n;
}

const char* nc_last_error(void) { return g_err; }
int nc_version(void) { return 100; }
void nc_set_force_direct(int on) { g_force_direct = on; }
void nc_sconv_set_cfg(int cfg) { sconv_set_cfg(cfg); }
void nc_sconv_set_tune(int on) { sconv_set_tune(on); }
void nc_set_conv_split(int on) { g_split = on; }
void nc_set_s3_fusion(int on) { g_s3_fuse = on; }
int nc_get_conv_split(void) { return g_split; }
void nc_set_split_terms(int terms) { s3x_set_terms(terms); }
int nc_get_split_terms(void) { return s3x_get_terms(); }
void nc_set_epi_stats(int on) { epi_stats_set(on); }
int nc_get_epi_stats(void) { return epi_stats_mode(); }
void nc_set_s3x_w64(int on) { s3x_w64_set(on); }
int nc_get_s3x_w64(void) { return s3x_w64_get(); }
void nc_set_p2d_terms(int mode) { p2d_set_terms(mode); }
int nc_get_p2d_terms(void) { return p2d_get_terms(); }
void nc_set_h2_guard(int on) { h2_guard_set(on); }
int nc_get_h2_guard(void) { return h2_guard_mode(); }
int nc_h2_guard_stats(unsigned long long* out4, int reset) {
  if (!out4) { set_error("h2_guard_stats: null pointer"); return NC_ERR_ARG; }
  return h2_guard_read(out4, reset);
}

int nc_conv2d_split_active(int what, int N, int C, int H, int W, int K, int k, int stride, int pad) {
  ConvDims d;
  if (!make_dims(d, N, C, 1, H, W, K, 1, k, k, stride, pad) || g_force_direct || !g_split) return 0;
  return what == 0 ? (p2d_fwd_supported(d) ? 1 : 0) : what == 1 ? (p2d_dgrad_supported(d) ? 1 : 0) : what == 2 ? (p2d_wgrad_supported(d) ? 1 : 0) : 0;
}

int nc_conv_fwd_path(int C, int K, int kd, int kh, int kw, int stride, int pad) {
  ConvDims d;
  const int e = (kd == 1 && kh == 1 && kw == 1) ? 256 : 32;  // pointwise: a plane large enough for the flat kernel
  if (!make_dims(d, 1, C, kd > 1 ? 32 : 1, e, e, K, kd, kh, kw, stride, pad)) return -1;
  if (g_force_direct) return 0;
  if (g_split && s3_fwd_supported(d)) return 9;
  if (g_split && c1k7_h2_supported(d)) return 11;  // Conv3d(1, 64, 7) on the two-term kernels in pseudo-channel form (round 6)
  return mfma_fwd_supported(d) ? 1 : flat_1x1_supported(d) ? 3 : pg1_on(d) ? 8 : gemm_fwd_supported(d) ? 2 : 0;  // (the K = 1
  // reduction kernel of the PatchGAN head and the image-staged kernels depend on the batch, which this query does not take:
  // reported as 2)
}
int nc_conv_wgrad_path(int C, int K, int kd, int kh, int kw, int stride, int pad) {
  ConvDims d;
  const int e = (kd == 1 && kh == 1 && kw == 1) ? 256 : 32;  // pointwise: a plane large enough for the flat kernel
  if (!make_dims(d, 1, C, kd > 1 ? 32 : 1, e, e, K, kd, kh, kw, stride, pad)) return -1;
  if (g_force_direct) return 0;
  if (g_split && s3_wgrad_supported(d)) return 9;
  return mfma_wgrad_supported(d) ? 1 : wgrad_1x1_supported(d) ? 3 : c1_wgrad_supported(d) ? 4
                                                                         : gemm_wgrad_supported(d) ? 2 : 0;
}

size_t nc_conv_ws_bytes(int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad) {
  ConvDims d;
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return 0;
  size_t b = mfma_ws_bytes(d);
  const size_t g = gemm_ws_bytes(d);
  if (g > b) b = g;
  const size_t o = wgrad_1x1_ws_bytes(d);
  if (o > b) b = o;
  const size_t c1 = c1_wgrad_ws_bytes(d);
  if (c1 > b) b = c1;
  const size_t t1 = to1_mfma_ws_bytes(d);
  if (t1 > b) b = t1;
  if (sconv_fwd_supported(d) || sconv_dgrad_supported(d)) {
    const size_t sc = sconv_ws_bytes(d);
    if (sc > b) b = sc;
  }
  if (sconv_wgrad_supported(d)) {
    const size_t sc = sconv_wgrad_ws_bytes(d);
    if (sc > b) b = sc;
  }
  if (pg1_supported(d)) {
    const size_t sc = pg1_ws_bytes(d);
    if (sc > b) b = sc;
  }
  if (g_split) {
    size_t sc = p2d_ws_bytes(d);
    if (sc > b) b = sc;
    sc = p2d_wgrad_ws_bytes(d);
    if (sc > b) b = sc;
  }
  if (g_split && (s3_fwd_supported(d) || s3_dgrad_supported(d))) {
    const size_t sc = s3_ws_bytes(d);
    if (sc > b) b = sc;
  }
  if (g_split && c1k7_h2_ws_bytes(d) > b) b = c1k7_h2_ws_bytes(d);
  if (g_split && c1k7_h2_dgrad_ws_bytes(d) > b) b = c1k7_h2_dgrad_ws_bytes(d);
  if (g_split && s3_wgrad_supported(d)) {
    size_t sc = s3_wgrad_ws_bytes(d);
    if (sc > b) b = sc;
    sc = s3_bwd_ws_bytes(d);
    if (sc > b) b = sc;
  }
  if (b < kBiasGradWsBytes) b = kBiasGradWsBytes;
  return (b + 255) & ~(size_t)255;
}

static int conv_args(const char* what, ConvDims& d, const void* a, const void* b, const void* c, int N, int C, int D,
                     int H, int W, int K, int kd, int kh, int kw, int stride, int pad) {
  if (!a || !b || !c) { set_error("%s: null pointer", what); return NC_ERR_ARG; }
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad) || N > 65535) {
    set_error("%s: bad shape N=%d C=%d D=%d H=%d W=%d K=%d k=(%d,%d,%d) s=%d p=%d", what, N, C, D, H, W, K, kd, kh, kw,
              stride, pad);
    return NC_ERR_SHAPE;
  }
  return NC_OK;
}

int nc_conv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W, int K,
                int kd, int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if (int e = conv_args("conv_fwd", d, x, w, y, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return e;
  hipStream_t s = (hipStream_t)stream;
  const int path = fwd_path(d);
  ProfScope ps(0, path, d, 0, s);
  if (path == 9) return conv_fwd_s3(x, nullptr, w, bias, y, d, ws, ws_bytes, s);
  if (path == 10) return conv_fwd_p2d(x, w, bias, y, d, ws, ws_bytes, s);
  if (path == 11 && ws && ws_bytes >= c1k7_h2_ws_bytes(d)) return conv_c1k7_h2(x, w, bias, y, d, ws, ws_bytes, s);
  if (!g_force_direct && c1k3_fwd_supported(d)) return conv_fwd_c1k3(x, w, bias, y, d, s);  // 1 -> K channels, 3^3: its own fp32 MFMA kernel
  if (!g_force_direct && mfma_fwd_supported(d)) return conv_fwd_mfma(x, w, bias, y, d, ws, ws_bytes, s);
  if (!g_force_direct && flat_1x1_supported(d)) return conv_fwd_1x1(x, w, bias, y, d, ws, ws_bytes, s);
  if (!g_force_direct && k1_fwd_supported(d)) return conv_fwd_k1(x, w, bias, y, d, s);
  if (!g_force_direct && fwd_path(d) == 8) return conv_fwd_pg1(x, w, bias, y, d, 1.f, s);
  if (!g_force_direct && sconv_on(d, 0)) return conv_fwd_sconv(x, w, bias, y, d, ws, ws_bytes, s);
  if (!g_force_direct && gemm_fwd_supported(d)) return conv_fwd_gemm(x, w, bias, y, d, ws, ws_bytes, s);
  return conv_fwd_direct(x, w, bias, y, d, s);
}

int nc_conv_dgrad(const float* dy, const float* w, float* dx, int N, int C, int D, int H, int W, int K, int kd, int kh,
                  int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if (int e = conv_args("conv_dgrad", d, dy, w, dx, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return e;
  hipStream_t s = (hipStream_t)stream;
  const int path = dgrad_path(d);
  ProfScope ps(1, path, d, 0, s);
  if (path == 9) return conv_dgrad_s3(dy, nullptr, w, dx, d, ws, ws_bytes, s);
  if (path == 10) return conv_dgrad_p2d(dy, w, dx, d, ws, ws_bytes, s);
  if (path == 11 && ws && ws_bytes >= c1k7_h2_dgrad_ws_bytes(d)) return conv_c1k7_h2_dgrad(dy, w, dx, d, ws, ws_bytes, s);
  if (!g_force_direct && mfma_dgrad_supported(d)) return conv_dgrad_mfma(dy, w, dx, d, ws, ws_bytes, s);
  if (!g_force_direct && flat_1x1_supported(d)) return conv_dgrad_1x1(dy, w, dx, d, ws, ws_bytes, s);
  if (!g_force_direct && to1_mfma_supported(d)) return conv_dgrad_to1_mfma(dy, w, dx, d, ws, ws_bytes, s);
  if (!g_force_direct && to1_dgrad_supported(d)) return conv_dgrad_to1(dy, w, dx, d, s);
  if (!g_force_direct && dgrad_path(d) == 8) return conv_dgrad_pg1(dy, nullptr, 1.f, w, dx, d, s);
  if (!g_force_direct && sconv_on(d, 1)) return conv_dgrad_sconv(dy, w, dx, d, ws, ws_bytes, s);
  if (!g_force_direct && gemm_dgrad_supported(d)) return conv_dgrad_gemm(dy, w, dx, d, ws, ws_bytes, s);
  return conv_dgrad_direct(dy, w, dx, d, s);
}

int nc_conv_lp_supported(int what, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad) {
  ConvDims d;
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return 0;
  return what == 0 ? h_fwd_supported(d) : what == 1 ? h_dgrad_supported(d) : what == 2 ? h_wgrad_supported(d) : 0;
}

size_t nc_conv_lp_ws_bytes(int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad) {
  ConvDims d;
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return 0;
  return h_ws_bytes(d);
}

static int lp_args(const char* what, ConvDims& d, const void* a, const void* ah, const void* b, const void* o, int N, int C,
                   int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad, int dtype) {
  if ((!a && !ah) || !b || !o) { set_error("%s: null pointer", what); return NC_ERR_ARG; }
  if (!make_dims(d, N, C, D, H, W, K, kd, kh, kw, stride, pad)) { set_error("%s: bad shape", what); return NC_ERR_SHAPE; }
  if (dtype != NC_DT_F16 && dtype != NC_DT_BF16) { set_error("%s: dtype must be NC_DT_F16 or NC_DT_BF16", what); return NC_ERR_ARG; }
  return NC_OK;
}

size_t nc_c8_bytes(int N, int C, long S) { return (C % 8 || N < 1 || S < 1) ? 0 : (size_t)N * C * S * 2; }

int nc_to_c8(const float* x, void* xh, int N, int C, long S, int dtype, void* stream) {
  if (!x || !xh) { set_error("to_c8: null pointer"); return NC_ERR_ARG; }
  if (N < 1 || C < 8 || C % 8 || S < 1) { set_error("to_c8: channels must be a multiple of 8"); return NC_ERR_SHAPE; }
  if (dtype != NC_DT_F16 && dtype != NC_DT_BF16) { set_error("to_c8: dtype must be NC_DT_F16 or NC_DT_BF16"); return NC_ERR_ARG; }
  return to_c8(x, xh, N, C, S, dtype, (hipStream_t)stream);
}

int nc_conv_fwd_lp(const float* x, const void* xh, const float* w, const float* bias, float* y, int N, int C, int D, int H,
                   int W, int K, int kd, int kh, int kw, int stride, int pad, int dtype, void* ws, size_t ws_bytes,
                   void* stream) {
  ConvDims d;
  if (int e = lp_args("conv_fwd_lp", d, x, xh, w, y, N, C, D, H, W, K, kd, kh, kw, stride, pad, dtype)) return e;
  if (!h_fwd_supported(d)) { set_error("conv_fwd_lp: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  ProfScope ps(0, 1, d, 1, (hipStream_t)stream);
  return conv_fwd_h(x, xh, w, bias, y, d, dtype, ws, ws_bytes, (hipStream_t)stream);
}

int nc_conv_dgrad_lp(const float* dy, const void* dyh, const float* w, float* dx, int N, int C, int D, int H, int W, int K,
                     int kd, int kh, int kw, int stride, int pad, int dtype, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if (int e = lp_args("conv_dgrad_lp", d, dy, dyh, w, dx, N, C, D, H, W, K, kd, kh, kw, stride, pad, dtype)) return e;
  if (!h_dgrad_supported(d)) { set_error("conv_dgrad_lp: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  ProfScope ps(1, 1, d, 1, (hipStream_t)stream);
  return conv_dgrad_h(dy, dyh, w, dx, d, dtype, ws, ws_bytes, (hipStream_t)stream);
}

int nc_conv_wgrad_lp(const float* x, const void* xh, const float* dy, const void* dyh, float* dw, float* dbias, int N, int C,
                     int D, int H, int W, int K, int kd, int kh, int kw, int stride, int pad, int dtype, void* ws,
                     size_t ws_bytes, void* stream) {
  ConvDims d;
  if (int e = lp_args("conv_wgrad_lp", d, x, xh, dw, dw, N, C, D, H, W, K, kd, kh, kw, stride, pad, dtype)) return e;
  if (!dy && !dyh) { set_error("conv_wgrad_lp: null pointer"); return NC_ERR_ARG; }
  if (dbias && !dy) { set_error("conv_wgrad_lp: the bias gradient needs the fp32 dy"); return NC_ERR_ARG; }
  if (!h_wgrad_supported(d)) { set_error("conv_wgrad_lp: shape not covered by the 16-bit kernels"); return NC_ERR_SHAPE; }
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps(2, 1, d, 1, s);
    if (int e = conv_wgrad_h(x, xh, dy, dyh, dw, d, dtype, ws, ws_bytes, s)) return e;
  }
  if (dbias) return bias_grad(dy, dbias, d.N, d.K, (long)d.Do * d.Ho * d.Wo, ws, ws_bytes, s);  // fp32, from the fp32 dy
  return NC_OK;
}

int nc_conv_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int D, int H, int W, int K,
                  int kd, int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if (int e = conv_args("conv_wgrad", d, x, dy, dw, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return e;
  hipStream_t s = (hipStream_t)stream;
  int e;
  {
  const int path = wgrad_path(d);
  ProfScope ps(2, path, d, 0, s);
  if (path == 9) e = conv_wgrad_s3(x, nullptr, dy, nullptr, dw, d, ws, ws_bytes, s);
  else if (path == 10) e = conv_wgrad_p2d(x, dy, dw, d, ws, ws_bytes, s);
  else if (!g_force_direct && mfma_wgrad_supported(d)) e = conv_wgrad_mfma(x, dy, dw, d, ws, ws_bytes, s);
  else if (!g_force_direct && wgrad_1x1_supported(d)) e = conv_wgrad_1x1(x, dy, dw, d, ws, ws_bytes, s);
  else if (!g_force_direct && c1_wgrad_supported(d)) e = conv_wgrad_c1(x, dy, dw, d, ws, ws_bytes, s);
  else if (!g_force_direct && k1_wgrad_supported(d)) e = conv_wgrad_k1(x, dy, dw, d, s);
  else if (!g_force_direct && wgrad_path(d) == 8) {  // the bias gradient is a column of the same product
    e = conv_wgrad_pg1(x, dy, nullptr, 1.f, dw, dbias, d, ws, ws_bytes, s);
    dbias = nullptr;
  }
  else if (!g_force_direct && sconv_wgrad_on(d)) e = conv_wgrad_sconv(x, dy, dw, d, ws, ws_bytes, s);
  else if (!g_force_direct && gemm_wgrad_supported(d)) e = conv_wgrad_gemm(x, dy, dw, d, ws, ws_bytes, s);
  else e = conv_wgrad_direct(x, dy, dw, d, s);
  }
  if (e) return e;
  if (dbias) return bias_grad(dy, dbias, d.N, d.K, (long)d.Do * d.Ho * d.Wo, ws, ws_bytes, s);
  return NC_OK;
}

// Backward of one convolution layer: dx (nullable) and dw (+ dbias) from x, dy, w.  The same results as nc_conv_dgrad followed
// by nc_conv_wgrad; on the split-operand kernels dY is converted to its three-term form once for both.
int nc_conv_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias, int N, int C, int D, int H, int W,
                int K, int kd, int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  if (int e = conv_args("conv_bwd", d, x, dy, dw, N, C, D, H, W, K, kd, kh, kw, stride, pad)) return e;
  if (!w) { set_error("conv_bwd: null pointer"); return NC_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  SwitchScope switches_;  // the three phases below see ONE value of every arithmetic switch
  if (dx && dgrad_path(d) == 9 && wgrad_path(d) == 9 && ws && s3_bwd_ws_bytes(d) && ws_bytes >= s3_bwd_ws_bytes(d)) {
    {
      ProfScope ps(1, 9, d, 0, s);
      if (int e = conv_bwd_s3(x, dy, w, dx, dw, d, ws, ws_bytes, s, 0)) return e;
      if (int e = conv_bwd_s3(x, dy, w, dx, dw, d, ws, ws_bytes, s, 1)) return e;
    }
    {
      ProfScope ps(2, 9, d, 0, s);
      if (int e = conv_bwd_s3(x, dy, w, dx, dw, d, ws, ws_bytes, s, 2)) return e;
    }
    if (dbias) return bias_grad(dy, dbias, d.N, d.K, (long)d.Do * d.Ho * d.Wo, ws, ws_bytes, s);
    return NC_OK;
  }
  if (dx)
    if (int e = nc_conv_dgrad(dy, w, dx, N, C, D, H, W, K, kd, kh, kw, stride, pad, ws, ws_bytes, stream)) return e;
  return nc_conv_wgrad(x, dy, dw, dbias, N, C, D, H, W, K, kd, kh, kw, stride, pad, ws, ws_bytes, stream);
}

}  // extern "C"

namespace nc {
int conv_fwd_keep(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W, int K, int ks,
                  void* ws, size_t ws_bytes, void* stream, void* xs_keep, bool* kept) {
  ConvDims d;
  *kept = false;
  if (xs_keep && make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) && fwd_path(d) == 9 && wgrad_path(d) == 9) {
    if (!x || !w || !y) { set_error("conv_fwd: null pointer"); return NC_ERR_ARG; }
    ProfScope ps(0, 9, d, 0, (hipStream_t)stream);
    *kept = true;
    return conv_fwd_s3(x, nullptr, w, bias, y, d, ws, ws_bytes, (hipStream_t)stream, xs_keep);
  }
  return nc_conv_fwd(x, w, bias, y, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2, ws, ws_bytes, stream);
}

// Does the 3^3 / 5^3 layer run forward AND weight gradient on the split-operand kernels (then producers may hand it its input in S3 form)?
bool conv_keep_supported(int N, int C, int D, int H, int W, int K, int ks) {
  ConvDims d;
  return make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) && fwd_path(d) == 9 && wgrad_path(d) == 9;
}

// forward of such a layer whose input already exists in S3 form (written by the producer: act_split3 / split3_into)
int conv_fwd_pre(const void* xs, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W, int K, int ks, void* ws,
                 size_t ws_bytes, void* stream, float* stats_part) {
  ConvDims d;
  if (!xs || !w || !y) { set_error("conv_fwd_pre: null pointer"); return NC_ERR_ARG; }
  if (!make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) || fwd_path(d) != 9) { set_error("conv_fwd_pre: layer not on the split-operand kernels"); return NC_ERR_SHAPE; }
  ProfScope ps(0, 9, d, 0, (hipStream_t)stream);
  if (stats_part && !(conv_layer_h2(d) && ks == 3)) { set_error("conv_fwd_pre: epilogue statistics exist for the two-term 3^3 layers only"); return NC_ERR_ARG; }
  return conv_fwd_s3(nullptr, xs, w, bias, y, d, ws, ws_bytes, (hipStream_t)stream, nullptr, stats_part);
}

// Can the backward of the layer take dY in S3 form at the head of its workspace (conv_bwd_pre)?  want_dx: the data gradient is needed too
bool conv_bwd_pre_supported(int N, int C, int D, int H, int W, int K, int ks, bool want_dx, size_t ws_bytes) {
  ConvDims d;
  return make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) && wgrad_path(d) == 9 && s3_bwd_ws_bytes(d) && ws_bytes >= s3_bwd_ws_bytes(d) &&
         (!want_dx || dgrad_path(d) == 9);
}
// data (dx nullable) + weight gradient with dY ALREADY in S3 form at the start of ws (where conv_bwd_s3's conversion phase puts it);
// xs (nullable): the layer's input in S3 form, else it is converted from x
int conv_bwd_pre(const float* x, const void* xs, const float* w, float* dx, float* dw, int N, int C, int D, int H, int W, int K, int ks,
                 void* ws, size_t ws_bytes, void* stream, bool dy_guarded) {
  ConvDims d;
  hipStream_t s = (hipStream_t)stream;
  if (!conv_bwd_pre_supported(N, C, D, H, W, K, ks, dx != nullptr, ws_bytes) || !make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2)) {
    set_error("conv_bwd_pre: layer not on the split-operand kernels");
    return NC_ERR_SHAPE;
  }
  if ((!x && !xs) || !w || !dw || !ws) { set_error("conv_bwd_pre: null pointer"); return NC_ERR_ARG; }
  if (dx) {
    ProfScope ps(1, 9, d, 0, s);
    if (int e = conv_bwd_s3(x, nullptr, w, dx, dw, d, ws, ws_bytes, s, 1, nullptr, dy_guarded)) return e;
  }
  ProfScope ps(2, 9, d, 0, s);
  return conv_bwd_s3(x, nullptr, w, dx, dw, d, ws, ws_bytes, s, 2, xs, dy_guarded);
}

int conv_bwd_keep(const float* x, const void* xs, const float* dy, const float* w, float* dx, float* dw, int N, int C, int D, int H,
                  int W, int K, int ks, void* ws, size_t ws_bytes, void* stream) {
  ConvDims d;
  hipStream_t s = (hipStream_t)stream;
  if (xs && make_dims(d, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2) && wgrad_path(d) == 9 && ws && s3_bwd_ws_bytes(d) &&
      ws_bytes >= s3_bwd_ws_bytes(d) &&
      (!dx || dgrad_path(d) == 9)) {
    if (!x || !dy || !w || !dw) { set_error("conv_bwd: null pointer"); return NC_ERR_ARG; }
    {
      ProfScope ps(1, 9, d, 0, s);
      if (int e = conv_bwd_s3(x, dy, w, dx, dw, d, ws, ws_bytes, s, 0)) return e;
      if (dx)
        if (int e = conv_bwd_s3(x, dy, w, dx, dw, d, ws, ws_bytes, s, 1)) return e;
    }
    ProfScope ps(2, 9, d, 0, s);
    return conv_bwd_s3(x, dy, w, dx, dw, d, ws, ws_bytes, s, 2, xs);
  }
  return nc_conv_bwd(x, dy, w, dx, dw, nullptr, N, C, D, H, W, K, ks, ks, ks, 1, ks / 2, ws, ws_bytes, stream);
}
}  // namespace nc

extern "C" {

// ------------------------------------------------------------------------------------------------------------------
// Whole-network forward of Unet_deconv.  Parameter blob = state-dict order, fp32, back to back.
namespace {
struct UnetOff {
  size_t w[14], b[14];  // 10 convs (0..9), t_conv2 (10), t_conv1 (11), one_by_one (12), one_by_one_2 (13)
  size_t total;
};
UnetOff unet_offsets() {
  // order: dc1.0 dc1.3 dc2.0 dc2.3 bot.0 bot.3 bot.6 t_conv2 ex2.0 ex2.3 t_conv1 ex1.0 1x1 1x1_2
  struct L { int id; size_t wn, bn; };
  const L order[14] = {{0, 64 * 1 * 27, 64},       {1, 64 * 64 * 27, 64},     {2, 128 * 64 * 27, 128},
                       {3, 128 * 128 * 27, 128},   {4, 256 * 128 * 27, 256},  {5, 256 * 256 * 27, 256},
                       {6, 256 * 256 * 27, 256},   {10, 256 * 128 * 8, 128},  {7, 128 * 256 * 27, 128},
                       {8, 128 * 128 * 27, 128},   {11, 128 * 64 * 8, 64},    {9, 64 * 128 * 27, 64},
                       {12, 64, 1},                {13, 1, 1}};
  UnetOff o{};
  size_t off = 0;
  for (const L& l : order) {
    o.w[l.id] = off; off += l.wn;
    o.b[l.id] = off; off += l.bn;
  }
  o.total = off;
  return o;
}
struct UnetWs {
  size_t raw, cat1, a1, p1, cat2, a2, a2b, p2, b1, b2, t1, t2, mean, rstd, in_ws, conv_ws, total;
  size_t s_a1, s_cat1, s_a2, s_cat2, s_b1, s_b2;  // S3 (three-term bf16) forms of the convolution inputs, conv_split.hip
  size_t cells;  // H2 mode (nc_set_split_terms(2)): scale cells of the convolution inputs, h2.hip
  size_t stats;  // H2 mode: partial InstanceNorm sums written by the convolutions' own epilogues (conv_s3x.hip ST), largest layer
  size_t in_ws_bytes, conv_ws_bytes;
};
// NB samples per pass: the two-term (H2) forward runs a whole batch through every launch (round 6: three 140^3 cubes per call fill the
// persistent 256-workgroup launches of the 70^3 / 35^3 levels -- 1.64 rounds of tiles become 4.92 -- 5 % of the convolution time); the other
// forms walk the samples one by one through buffers of one sample (NB = 1).
UnetWs unet_ws(int S0, int S1, int S2, int NB = 1) {
  const size_t S1v = (size_t)S0 * S1 * S2;
  const size_t S = S1v * NB, Sh = S / 8, Sq = S / 64;  // (per-batch element counts: every activation buffer holds NB samples)
  UnetWs u{};
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += (n + 63) & ~(size_t)63; return r; };
  u.raw = take(64 * S); u.cat1 = take(128 * S); u.a1 = take(64 * S); u.p1 = take(64 * Sh); u.cat2 = take(256 * Sh);
  u.a2 = take(128 * Sh); u.a2b = take(128 * Sh); u.p2 = take(128 * Sq); u.b1 = take(256 * Sq); u.b2 = take(256 * Sq);
  u.t1 = take(S); u.t2 = take(S); u.mean = take(256 * (size_t)NB); u.rstd = take(256 * (size_t)NB);
  u.in_ws_bytes = nc_instnorm_ws_bytes(256, (long)S1v);
  u.in_ws = take(u.in_ws_bytes / 4);
  size_t cw = 0;
  auto upd = [&](int C, int D, int H, int W, int K) {
    size_t b = nc_conv_ws_bytes(1, C, D, H, W, K, 3, 3, 3, 1, 1);
    if (b > cw) cw = b;
  };
  upd(64, S0, S1, S2, 64); upd(128, S0, S1, S2, 64);
  upd(64, S0 / 2, S1 / 2, S2 / 2, 128); upd(128, S0 / 2, S1 / 2, S2 / 2, 128); upd(256, S0 / 2, S1 / 2, S2 / 2, 128);
  upd(128, S0 / 4, S1 / 4, S2 / 4, 256); upd(256, S0 / 4, S1 / 4, S2 / 4, 256);
  u.conv_ws_bytes = cw;
  u.conv_ws = take(cw / 4 + 64);
  auto take3 = [&](size_t elems) { return take((elems * 6 + 3) / 4); };  // 6 bytes per element
  u.s_a1 = take3(64 * S); u.s_cat1 = take3(128 * S); u.s_a2 = take3(128 * Sh); u.s_cat2 = take3(256 * Sh);
  u.s_b1 = take3(256 * Sq); u.s_b2 = take3(256 * Sq);
  u.cells = take(64);
  {
    size_t sb = 0;
    auto upds = [&](int D, int H, int W, int K) { const size_t b = s3x_stats_bytes(NB, D, H, W, K, 3); if (b > sb) sb = b; };
    upds(S0, S1, S2, 64); upds(S0 / 2, S1 / 2, S2 / 2, 128); upds(S0 / 4, S1 / 4, S2 / 4, 256);
    ConvDims d0;
    if (make_dims(d0, 1, 1, S0, S1, S2, 64, 3, 3, 3, 1, 1) && c1k3_stats_bytes(d0) > sb) sb = c1k3_stats_bytes(d0);  // (the first block's own kernel)
    u.stats = take(sb / 4 + 64);
  }
  u.total = off;
  return u;
}
}  // namespace

// H2 mode of the whole-network forward: every 3^3 layer must be one the tap-stream kernel covers
static bool unet_h2_ok(int S0, int S1, int S2, size_t conv_ws_bytes) {
  const int h0 = S0 / 2, h1 = S1 / 2, h2 = S2 / 2, q0 = S0 / 4, q1 = S1 / 4, q2 = S2 / 4;
  return g_split && g_s3_fuse && s3x_get_terms() != 3 && s3x_supported(1, 64, S0, S1, S2, 64, 3) && s3x_supported(1, 128, S0, S1, S2, 64, 3) &&
         s3x_supported(1, 64, h0, h1, h2, 128, 3) && s3x_supported(1, 128, h0, h1, h2, 128, 3) && s3x_supported(1, 256, h0, h1, h2, 128, 3) &&
         s3x_supported(1, 128, q0, q1, q2, 256, 3) && s3x_supported(1, 256, q0, q1, q2, 256, 3) &&
         conv_ws_bytes >= 256 + s3x_packed_bytes(256, 256, 3, 2) + 256;
}

int nc_unet_deconv_fwd_terms(int S0, int S1, int S2) {
  if (S0 < 4 || S1 < 4 || S2 < 4 || (S0 & 3) || (S1 & 3) || (S2 & 3)) return 0;
  ConvDims d;
  const bool f = make_dims(d, 1, 64, S0, S1, S2, 64, 3, 3, 3, 1, 1) && fwd_path(d) == 9 && make_dims(d, 1, 256, S0 / 4, S1 / 4, S2 / 4, 256, 3, 3, 3, 1, 1) &&
                 fwd_path(d) == 9;
  return f && unet_h2_ok(S0, S1, S2, unet_ws(S0, S1, S2).conv_ws_bytes) ? 2 : 3;
}

// samples per pass of the whole-network forward: the whole batch in the two-term mode (NC_INFER_BATCHED=0: one by one, A/B), else one
static int unet_fwd_batch(int N, int S0, int S1, int S2) {
  static const bool batched = !(getenv("NC_INFER_BATCHED") && atoi(getenv("NC_INFER_BATCHED")) == 0);
  if (N <= 1 || !batched || nc_unet_deconv_fwd_terms(S0, S1, S2) != 2) return 1;
  return N;
}

size_t nc_unet_deconv_fwd_ws_bytes(int N, int S0, int S1, int S2) {
  if (N < 1 || S0 < 4 || S1 < 4 || S2 < 4 || (S0 & 3) || (S1 & 3) || (S2 & 3)) return 0;
  return unet_ws(S0, S1, S2, unet_fwd_batch(N, S0, S1, S2)).total * sizeof(float);
}

#define NC_TRY(expr) do { int e_ = (expr); if (e_) return e_; } while (0)

int nc_unet_deconv_fwd(const float* params, const float* x, float* y, int N, int S0, int S1, int S2, void* ws,
                       size_t ws_bytes, void* stream) {
  NetworkScope net_scope;
  if (!params || !x || !y) { set_error("unet_deconv_fwd: null pointer"); return NC_ERR_ARG; }
  if (N < 1 || S0 < 4 || S1 < 4 || S2 < 4 || (S0 & 3) || (S1 & 3) || (S2 & 3)) {
    set_error("unet_deconv_fwd: every edge must be a positive multiple of 4 (got %d,%d,%d): MaxPool3d floors and the "
              "skip concat would not line up (reference networks.py:526,531)", S0, S1, S2);
    return NC_ERR_SHAPE;
  }
  const int NB = unet_fwd_batch(N, S0, S1, S2);  // NB = N: the two-term mode below takes the whole batch through every launch
  const UnetWs u = unet_ws(S0, S1, S2, NB);
  if (!ws || ws_bytes < u.total * sizeof(float)) { set_error("unet_deconv_fwd: workspace too small"); return NC_ERR_WS; }
  const UnetOff o = unet_offsets();
  float* W = (float*)ws;
  const float* P = params;
  const long S = (long)S0 * S1 * S2, Sh = S / 8, Sq = S / 64;
  const int h0 = S0 / 2, h1 = S1 / 2, h2 = S2 / 2, q0 = S0 / 4, q1 = S1 / 4, q2 = S2 / 4;
  void* cws = W + u.conv_ws;
  void* iws = W + u.in_ws;
  float *mean = W + u.mean, *rstd = W + u.rstd;
  hipStream_t hs = (hipStream_t)stream;
  // Does the 3^3 convolution C -> K at (D, H, Wd) run on the split-operand kernels?  Then its input is wanted in S3 form, and the
  // layer in front writes that form from its normalisation pass (k_act_split3) instead of a separate conversion pass.
  auto split_in = [&](int C, int K, int D, int H, int Wd) {
    ConvDims d;
    return g_s3_fuse && make_dims(d, 1, C, D, H, Wd, K, 3, 3, 3, 1, 1) && fwd_path(d) == 9;
  };
  // conv (3^3, pad 1) + InstanceNorm + ReLU: in (fp32) or in3 (S3, when the convolution takes it) -> raw -> out (fp32, nullable)
  // and / or out3 (channels c0 .. of an S3 tensor with ctot channels, nullable)
  // H2 mode: in3 / out3 are H2 tensors (two fp16 terms of the tensor times a power of two, h2.hip); in_a / in_b: the cells of the input's
  // channels [0, split_c) / [split_c, C) (a concatenation), out_cell: the cell the output is converted with
  unsigned* cells = (unsigned*)(W + u.cells);
  int bn = 1;  // samples per pass of `block`: NB in the batched two-term mode, 1 in the per-sample loop
  auto block = [&](int id, const float* in, const void* in3, float* out, void* out3, int ctot, int c0, int C, int K, int D, int H,
                   int Wd, const unsigned* in_a = nullptr, const unsigned* in_b = nullptr, int split_c = 0,
                   const unsigned* out_cell = nullptr, void* pooled = nullptr) -> int {
    const long Sl = (long)D * H * Wd;
    // normalise + ReLU + conversion of the raw output (and, pooled != NULL: the 2 x 2 x 2 max-pool of the H2 result in the same pass)
    auto finish = [&]() -> int {
      if (out3 && out_cell && pooled && !out) return act_split2h_pool(W + u.raw, mean, rstd, 0.f, out3, pooled, bn, K, D, H, Wd, ctot, c0, sqrtf((float)Sl), nullptr, hs);
      if (out3 && out_cell) {
        NC_TRY(act_split2h(W + u.raw, mean, rstd, 0.f, out, (long)K * Sl, out3, bn, K, Sl, ctot, c0, sqrtf((float)Sl), nullptr, nullptr, hs));
        return pooled ? maxpool2_h2(out3, pooled, bn, K, ctot, D, H, Wd, hs) : NC_OK;
      }
      if (out3) return act_split3(W + u.raw, mean, rstd, 0.f, out, (long)K * Sl, out3, 1, K, Sl, ctot, c0, hs);
      return nc_instnorm_act_fwd(W + u.raw, mean, rstd, 0.f, out, K, Sl, stream);
    };
    // Epilogue statistics (default; NC_EPI_STATS=0 / nc_set_epi_stats(0): off): the two-term convolution leaves the partial InstanceNorm sums of
    // its output itself (conv_s3x.hip, ST) and k_in_stats' pass over the raw output goes away: -1.8 % per 140^3 cube, mean / rstd equal to 4e-8
    const bool epi = epi_stats_on();
    if (in3 && in_a) {
      ConvDims d;
      make_dims(d, bn, C, D, H, Wd, K, 3, 3, 3, 1, 1);
      {
        ProfScope ps(0, 9, d, 0, hs);
        NC_TRY(conv_s3x_h2(in3, in_a, in_b, in_b ? split_c : C, P + o.w[id], P + o.b[id], W + u.raw, bn, C, D, H, Wd, K, 3, (long)C * 27, 27, 0,
                           (unsigned*)cws, (char*)cws + 256, hs, nullptr, epi ? W + u.stats : nullptr));
      }
      if (epi) {
        NC_TRY(s3x_stats_finalize(W + u.stats, P + o.b[id], bn, D, H, Wd, K, 3, 1e-5f, mean, rstd, hs));
        return finish();
      }
      if (bn > 1) {  // (statistics pass per sample: mean / rstd are [sample][channel])
        for (int n = 0; n < bn; ++n)
          NC_TRY(nc_instnorm_stats(W + u.raw + (long)n * K * Sl, K, Sl, 1e-5f, mean + (long)n * K, rstd + (long)n * K, iws, u.in_ws_bytes, stream));
        return finish();
      }
    } else if (in3) {
      ConvDims d;
      make_dims(d, 1, C, D, H, Wd, K, 3, 3, 3, 1, 1);
      ProfScope ps(0, 9, d, 0, hs);
      NC_TRY(conv_fwd_s3(nullptr, in3, P + o.w[id], P + o.b[id], W + u.raw, d, cws, u.conv_ws_bytes, hs));
    } else {
      ConvDims d1;
      if (epi && C == 1 && !g_force_direct && make_dims(d1, 1, C, D, H, Wd, K, 3, 3, 3, 1, 1) && c1k3_stats_bytes(d1)) {
        // the first block: the one-channel kernel leaves its InstanceNorm sums too (conv_c1k3.hip ST) -- one sample per launch, then ONE
        // normalise + convert pass over the batch
        for (int n = 0; n < bn; ++n) {
          NC_TRY(conv_fwd_c1k3(in + (long)n * Sl, P + o.w[id], P + o.b[id], W + u.raw + (long)n * K * Sl, d1, hs, W + u.stats));
          NC_TRY(c1k3_stats_finalize(W + u.stats, P + o.b[id], d1, 1e-5f, mean + (long)n * K, rstd + (long)n * K, hs));
        }
        return finish();
      }
      for (int n = 0; n < bn; ++n) {
        NC_TRY(nc_conv_fwd(in + (long)n * C * Sl, P + o.w[id], P + o.b[id], W + u.raw + (long)n * K * Sl, 1, C, D, H, Wd, K, 3, 3, 3, 1, 1, cws, u.conv_ws_bytes, stream));
        if (bn > 1) NC_TRY(nc_instnorm_stats(W + u.raw + (long)n * K * Sl, K, Sl, 1e-5f, mean + (long)n * K, rstd + (long)n * K, iws, u.in_ws_bytes, stream));
      }
      if (bn > 1) return finish();
    }
    NC_TRY(nc_instnorm_stats(W + u.raw, K, Sl, 1e-5f, mean, rstd, iws, u.in_ws_bytes, stream));
    return finish();
  };
  const bool f1 = split_in(64, 64, S0, S1, S2), f9 = split_in(128, 64, S0, S1, S2);
  const bool f3 = split_in(128, 128, h0, h1, h2), f7 = split_in(256, 128, h0, h1, h2), f8 = f3;
  const bool f5 = split_in(256, 256, q0, q1, q2), f6 = f5;
  // H2 mode: every 3^3 layer on k_conv_s3x<3, NCB, 2>.  The outputs of InstanceNorm + ReLU are bounded by sqrt(voxels) by construction
  // (cells 0..2, one per level); the transposed convolutions' halves of the concatenations are measured (cells 3, 4).
  const bool use_h2 = f1 && f3 && f5 && f7 && f9 && unet_h2_ok(S0, S1, S2, u.conv_ws_bytes);
  if (use_h2) {
    NC_TRY(h2_set_cell(cells + 0, sqrtf((float)S), hs));
    NC_TRY(h2_set_cell(cells + 1, sqrtf((float)Sh), hs));
    NC_TRY(h2_set_cell(cells + 2, sqrtf((float)Sq), hs));
  }
  bn = use_h2 ? NB : 1;
  for (int n = 0; n < N; n += bn) {
    const float* xn = x + (long)n * S;
    float* yn = y + (long)n * S;
    if (use_h2) {
      // (bn samples per launch from here on: every activation buffer is [bn][channels][voxels], the pooled / half-resolution offsets scale with bn)
      const unsigned *c0 = cells, *c1 = cells + 1, *c2 = cells + 2;
      NC_TRY(h2_zero_cells(cells + 3, 2, hs));
      NC_TRY(block(0, xn, nullptr, nullptr, W + u.s_a1, 64, 0, 1, 64, S0, S1, S2, nullptr, nullptr, 0, c0));
      // The max-pools work on the H2 tensors themselves (the winner's two terms are the pooled element's terms, same cell: h2.hip): blocks 1
      // and 3 write NO fp32 activation, and nothing is converted.  The pooled tensors go into slots nobody needs yet (the upper half of
      // s_cat2 until the transposed convolution's output arrives; s_b2 until block 5 writes it).  NC_POOL_H2=0: fp32 pool + conversion (A/B)
      static const bool pool_h2 = true;
      void* p1h = W + u.s_cat2 + 128 * Sh * bn;
      // (NC_POOL_FUSE=0: the pool as a pass of its own over the H2 tensor)
      static const bool pool_fuse = pool_h2 && true;
      NC_TRY(block(1, nullptr, W + u.s_a1, pool_h2 ? nullptr : W + u.cat1, W + u.s_cat1, 128, 0, 64, 64, S0, S1, S2, c0, nullptr, 0, c0, pool_fuse ? p1h : nullptr));
      if (pool_fuse) {
      } else if (pool_h2) {
        NC_TRY(maxpool2_h2(W + u.s_cat1, p1h, bn, 64, 128, S0, S1, S2, hs));
      } else {
        NC_TRY(nc_maxpool2_fwd(W + u.cat1, W + u.p1, 64 * bn, S0, S1, S2, stream));  // (the fp32 pool works per (sample, channel) plane)
        NC_TRY(split2h_into(W + u.p1, 64 * Sh, p1h, bn, 64, Sh, 64, 0, c0, hs));
      }
      NC_TRY(block(2, nullptr, p1h, nullptr, W + u.s_a2, 128, 0, 64, 128, h0, h1, h2, c0, nullptr, 0, c1));
      NC_TRY(block(3, nullptr, W + u.s_a2, pool_h2 ? nullptr : W + u.cat2, W + u.s_cat2, 256, 0, 128, 128, h0, h1, h2, c1, nullptr, 0, c1,
                   pool_fuse ? (void*)(W + u.s_b2) : nullptr));
      if (pool_fuse) {
      } else if (pool_h2) {
        NC_TRY(maxpool2_h2(W + u.s_cat2, W + u.s_b2, bn, 128, 256, h0, h1, h2, hs));
      } else {
        NC_TRY(nc_maxpool2_fwd(W + u.cat2, W + u.p2, 128 * bn, h0, h1, h2, stream));
        NC_TRY(split2h_into(W + u.p2, 128 * Sq, W + u.s_b2, bn, 128, Sq, 128, 0, c1, hs));
      }
      NC_TRY(block(4, nullptr, W + u.s_b2, nullptr, W + u.s_b1, 256, 0, 128, 256, q0, q1, q2, c1, nullptr, 0, c2));
      NC_TRY(block(5, nullptr, W + u.s_b1, nullptr, W + u.s_b2, 256, 0, 256, 256, q0, q1, q2, c2, nullptr, 0, c2));
      // the transposed convolutions keep their three-term S3 input (small tensors) and write fp32; their halves of the concatenations
      // are measured and converted, the ratio of the two halves' powers of two goes into the consumer's weights
      const bool t10 = convT_s3x_supported(bn, 256, q0, q1, q2, 128) && u.conv_ws_bytes >= convT_s3x_ws_bytes(256, 128);
      const bool t11 = convT_s3x_supported(bn, 128, h0, h1, h2, 64) && u.conv_ws_bytes >= convT_s3x_ws_bytes(128, 64);
      static const bool ct_h2 = !(getenv("NC_CONVT_H2") && atoi(getenv("NC_CONVT_H2")) == 0);  // 0: three-term input, fp32 output measured and converted (A/B)
      const bool t10h = t10 && ct_h2, t11h = t11 && ct_h2;  // the two-term transposed convolution: H2 in (the block's output, bound sqrt(S)), H2 out
      NC_TRY(block(6, nullptr, W + u.s_b2, t10 ? nullptr : W + u.b1, t10 ? W + u.s_b1 : nullptr, 256, 0, 256, 256, q0, q1, q2, c2, nullptr, 0,
                   t10h ? c2 : nullptr));
      // (the matrix-core transposed convolution writes the H2 form of its half itself, with a BOUND for its power of two -- every output
      // voxel gets one tap per input channel, convt_s3.hip; the fp32 kernel's output is measured and converted)
      if (t10h) {
        NC_TRY(convT_h2_bound(P + o.w[10], P + o.b[10], 256, 128, sqrtf((float)Sq), cells + 3, hs));
        NC_TRY(convT_fwd_s3x(W + u.s_b1, P + o.w[10], P + o.b[10], nullptr, W + u.s_cat2, 256, 128, bn, 256, q0, q1, q2, 128, cws, u.conv_ws_bytes, hs,
                             cells + 3, c2));
      } else {
        // (A/B and fallback forms: the transposed convolution's fp32 output as a dense [bn][128] tensor in the upper half of cat2, measured and
        // converted into channels 128 .. 255 of the concatenation)
        float* up2 = W + u.cat2 + 128 * Sh * bn;
        if (t10) NC_TRY(convT_fwd_s3x(W + u.s_b1, P + o.w[10], P + o.b[10], up2, nullptr, 128, 0, bn, 256, q0, q1, q2, 128, cws, u.conv_ws_bytes, hs));
        else NC_TRY(nc_convT_k2s2_fwd(W + u.b1, P + o.w[10], P + o.b[10], up2, bn, 256, q0, q1, q2, 128, stream));
        NC_TRY(h2_absmax(up2, 128 * Sh * bn, cells + 3, hs));
        NC_TRY(split2h_into(up2, 128 * Sh, W + u.s_cat2, bn, 128, Sh, 256, 128, cells + 3, hs));
      }
      NC_TRY(block(7, nullptr, W + u.s_cat2, nullptr, W + u.s_a2, 128, 0, 256, 128, h0, h1, h2, c1, cells + 3, 128, c1));
      NC_TRY(block(8, nullptr, W + u.s_a2, t11 ? nullptr : W + u.a2b, t11 ? W + u.s_cat2 : nullptr, 128, 0, 128, 128, h0, h1, h2, c1, nullptr, 0,
                   t11h ? c1 : nullptr));
      if (t11h) {
        NC_TRY(convT_h2_bound(P + o.w[11], P + o.b[11], 128, 64, sqrtf((float)Sh), cells + 4, hs));
        NC_TRY(convT_fwd_s3x(W + u.s_cat2, P + o.w[11], P + o.b[11], nullptr, W + u.s_cat1, 128, 64, bn, 128, h0, h1, h2, 64, cws, u.conv_ws_bytes, hs,
                             cells + 4, c1));
      } else {
        float* up1 = W + u.cat1 + 64 * S * bn;
        if (t11) NC_TRY(convT_fwd_s3x(W + u.s_cat2, P + o.w[11], P + o.b[11], up1, nullptr, 64, 0, bn, 128, h0, h1, h2, 64, cws, u.conv_ws_bytes, hs));
        else NC_TRY(nc_convT_k2s2_fwd(W + u.a2b, P + o.w[11], P + o.b[11], up1, bn, 128, h0, h1, h2, 64, stream));
        NC_TRY(h2_absmax(up1, 64 * S * bn, cells + 4, hs));
        NC_TRY(split2h_into(up1, 64 * S, W + u.s_cat1, bn, 64, S, 128, 64, cells + 4, hs));
      }
      // the last block's normalisation, the two pointwise layers and the sigmoid in one pass over its raw output (NC_INFER_TAIL=0: separately)
      static const bool tail = true;
      if (tail) {
        const bool epi9 = epi_stats_on();
        {
          ConvDims d9;
          make_dims(d9, bn, 128, S0, S1, S2, 64, 3, 3, 3, 1, 1);
          ProfScope ps(0, 9, d9, 0, hs);
          NC_TRY(conv_s3x_h2(W + u.s_cat1, c0, cells + 4, 64, P + o.w[9], P + o.b[9], W + u.raw, bn, 128, S0, S1, S2, 64, 3, (long)128 * 27, 27, 0,
                             (unsigned*)cws, (char*)cws + 256, hs, nullptr, epi9 ? W + u.stats : nullptr));
        }
        if (epi9) NC_TRY(s3x_stats_finalize(W + u.stats, P + o.b[9], bn, S0, S1, S2, 64, 3, 1e-5f, mean, rstd, hs));
        for (int i = 0; i < bn; ++i) {  // (the fused tail is a per-sample pass: 0.13 ms per 140^3 cube)
          if (!epi9) NC_TRY(nc_instnorm_stats(W + u.raw + (long)i * 64 * S, 64, S, 1e-5f, mean + i * 64, rstd + i * 64, iws, u.in_ws_bytes, stream));
          NC_TRY(instnorm_relu_tail_sigmoid(W + u.raw + (long)i * 64 * S, mean + i * 64, rstd + i * 64, P + o.w[12], P + o.b[12], P + o.w[13], P + o.b[13],
                                            yn + (long)i * S, 64, S, hs));
        }
        continue;
      }
      NC_TRY(block(9, nullptr, W + u.s_cat1, W + u.a1, nullptr, 0, 0, 128, 64, S0, S1, S2, c0, cells + 4, 64));
      NC_TRY(nc_conv_fwd(W + u.a1, P + o.w[12], P + o.b[12], W + u.t1, bn, 64, S0, S1, S2, 1, 1, 1, 1, 1, 0, cws, u.conv_ws_bytes, stream));
      NC_TRY(nc_conv_fwd(W + u.t1, P + o.w[13], P + o.b[13], W + u.t2, bn, 1, S0, S1, S2, 1, 1, 1, 1, 1, 0, cws, u.conv_ws_bytes, stream));
      NC_TRY(nc_sigmoid_fwd(W + u.t2, yn, S * bn, stream));
      continue;
    }
    NC_TRY(block(0, xn, nullptr, f1 ? nullptr : W + u.a1, f1 ? W + u.s_a1 : nullptr, 64, 0, 1, 64, S0, S1, S2));
    NC_TRY(block(1, W + u.a1, f1 ? W + u.s_a1 : nullptr, W + u.cat1, f9 ? W + u.s_cat1 : nullptr, 128, 0, 64, 64, S0, S1, S2));
    NC_TRY(nc_maxpool2_fwd(W + u.cat1, W + u.p1, 64, S0, S1, S2, stream));
    NC_TRY(block(2, W + u.p1, nullptr, f3 ? nullptr : W + u.a2, f3 ? W + u.s_a2 : nullptr, 128, 0, 64, 128, h0, h1, h2));
    NC_TRY(block(3, W + u.a2, f3 ? W + u.s_a2 : nullptr, W + u.cat2, f7 ? W + u.s_cat2 : nullptr, 256, 0, 128, 128, h0, h1, h2));
    NC_TRY(nc_maxpool2_fwd(W + u.cat2, W + u.p2, 128, h0, h1, h2, stream));
    NC_TRY(block(4, W + u.p2, nullptr, f5 ? nullptr : W + u.b1, f5 ? W + u.s_b1 : nullptr, 256, 0, 128, 256, q0, q1, q2));
    NC_TRY(block(5, W + u.b1, f5 ? W + u.s_b1 : nullptr, f6 ? nullptr : W + u.b2, f6 ? W + u.s_b2 : nullptr, 256, 0, 256, 256, q0, q1, q2));
    // The transposed convolutions on the bf16 matrix cores (convt_s3.hip; with the split-operand kernels switched on).  They take
    // their input in S3 form: the block in front leaves its output ONLY in that form (in a slot that is free by then: s_b1 after
    // block 5, s_cat2 after block 7) -- or, with the fusion switched off, in fp32 and a separate pass converts it (the same bits) --
    // and they write the S3 form of the upper half of the concatenation themselves when the next convolution takes it (in the
    // inference forward nothing else reads it, so no fp32 copy is written), fp32 otherwise.
    const bool sp = g_split != 0, fuse = g_s3_fuse != 0;
    const bool t10 = sp && convT_s3x_supported(1, 256, q0, q1, q2, 128) && u.conv_ws_bytes >= convT_s3x_ws_bytes(256, 128);
    const bool t11 = sp && convT_s3x_supported(1, 128, h0, h1, h2, 64) && u.conv_ws_bytes >= convT_s3x_ws_bytes(128, 64);
    const bool b6s3 = t10 && fuse, b8s3 = t11 && fuse;
    NC_TRY(block(6, W + u.b2, f6 ? W + u.s_b2 : nullptr, b6s3 ? nullptr : W + u.b1, b6s3 ? W + u.s_b1 : nullptr, 256, 0, 256, 256, q0, q1, q2));
    if (t10) {
      if (!b6s3) NC_TRY(split3_into(W + u.b1, 256 * Sq, W + u.s_b1, 1, 256, Sq, 256, 0, hs));
      NC_TRY(convT_fwd_s3x(W + u.s_b1, P + o.w[10], P + o.b[10], f7 ? nullptr : W + u.cat2 + 128 * Sh, f7 ? W + u.s_cat2 : nullptr, 256, 128, 1,
                           256, q0, q1, q2, 128, cws, u.conv_ws_bytes, hs));
    } else {
      NC_TRY(nc_convT_k2s2_fwd(W + u.b1, P + o.w[10], P + o.b[10], W + u.cat2 + 128 * Sh, 1, 256, q0, q1, q2, 128, stream));
      if (f7) NC_TRY(split3_into(W + u.cat2 + 128 * Sh, 128 * Sh, W + u.s_cat2, 1, 128, Sh, 256, 128, hs));
    }
    NC_TRY(block(7, W + u.cat2, f7 ? W + u.s_cat2 : nullptr, f8 ? nullptr : W + u.a2, f8 ? W + u.s_a2 : nullptr, 128, 0, 256, 128, h0, h1, h2));
    NC_TRY(block(8, W + u.a2, f8 ? W + u.s_a2 : nullptr, b8s3 ? nullptr : W + u.a2b, b8s3 ? W + u.s_cat2 : nullptr, 128, 0, 128, 128, h0, h1, h2));
    if (t11) {
      if (!b8s3) NC_TRY(split3_into(W + u.a2b, 128 * Sh, W + u.s_cat2, 1, 128, Sh, 128, 0, hs));
      NC_TRY(convT_fwd_s3x(W + u.s_cat2, P + o.w[11], P + o.b[11], f9 ? nullptr : W + u.cat1 + 64 * S, f9 ? W + u.s_cat1 : nullptr, 128, 64, 1, 128,
                           h0, h1, h2, 64, cws, u.conv_ws_bytes, hs));
    } else if (f9 && convT_fwd_s3_supported(1, 128, h0, h1, h2, 64)) {  // the transposed convolution writes block 9's operand form itself; in the
      // inference forward nothing else reads its output, so the fp32 half of cat1 is not written at all
      NC_TRY(convT_fwd_s3(W + u.a2b, P + o.w[11], P + o.b[11], nullptr, W + u.s_cat1, 128, 64, 1, 128, h0, h1, h2, 64, stream));
    } else {
      NC_TRY(nc_convT_k2s2_fwd(W + u.a2b, P + o.w[11], P + o.b[11], W + u.cat1 + 64 * S, 1, 128, h0, h1, h2, 64, stream));
      if (f9) NC_TRY(split3_into(W + u.cat1 + 64 * S, 64 * S, W + u.s_cat1, 1, 64, S, 128, 64, hs));
    }
    NC_TRY(block(9, W + u.cat1, f9 ? W + u.s_cat1 : nullptr, W + u.a1, nullptr, 0, 0, 128, 64, S0, S1, S2));
    NC_TRY(nc_conv_fwd(W + u.a1, P + o.w[12], P + o.b[12], W + u.t1, 1, 64, S0, S1, S2, 1, 1, 1, 1, 1, 0, cws, u.conv_ws_bytes, stream));
    NC_TRY(nc_conv_fwd(W + u.t1, P + o.w[13], P + o.b[13], W + u.t2, 1, 1, S0, S1, S2, 1, 1, 1, 1, 1, 0, cws, u.conv_ws_bytes, stream));
    NC_TRY(nc_sigmoid_fwd(W + u.t2, yn, S, stream));
  }
  (void)Sq;
  return NC_OK;
}

}  // extern "C"
