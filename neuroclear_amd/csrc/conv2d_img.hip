// "Image-staged" implicit-GEMM convolution for the 2-D PatchGAN layers (reference models/networks.py:1030-1057: Conv2d
// k 4, stride 2 / 1, padding 1) on the fp32 matrix cores: forward, the data gradient as forward-shaped problems (k_sconv),
// and the weight gradient as a GEMM over the flat (image, u, v) axis (k_swgrad, further down).  The tile shape of k_sconv is
// chosen per problem by timing on first use (run_tuned): every shape accumulates in the same order, so the bits do not depend
// on the choice.
//
// Why: batched over Athena's slices (108-216 images of 108^2 per discriminator pass, axial_to_lateral_gan_athena_model.py:
// 286-296) these layers are long GEMMs -- M = 128..512 output channels, N = 10^4..10^5 output pixels, K = 1024..4096 -- and
// the gather GEMM (conv_gemm.hip) is bound by the ~12 vector instructions of index arithmetic per gathered element
// (36-81 TFLOP/s).  Here NOTHING is gathered per element:
//   * output pixels of the whole batch are ONE flat column axis j = (image, u, v); a workgroup owns NCT consecutive
//     columns (they span <= 3 images) x 64 * WN output channels;
//   * per channel chunk the input rows those columns need are staged in LDS as zero-padded row segments [ci][rows][Wp]
//     (LDS-DMA, one dword per lane, source offsets from a per-tile table decoded once);
//   * a lane keeps ONE base offset per column block: pixel (u, v) of its image inside the staged segment; tap (ty, tx) of
//     channel c is then base + c * CS + (ty * Wp + tx) * dt -- a scalar offset: the B operand of an MFMA is a plain
//     ds_read_b32, stride and padding live entirely in the base;
//   * the data gradient of a stride-2 layer runs per output-parity class (the 4 x 4 kernel splits into four 2 x 2
//     kernels, one per class, walked backwards: dt = -1), of the stride-1 layer as one 4 x 4 problem with dt = -1;
//   * weights stream through a buffer descriptor, packed [co tile][channel pair][tap][h][32][2], as in conv_mfma_fwd.hip.
// v_mfma_f32_32x32x2_f32 (exact fp32): the two k values of an MFMA are two input channels of one tap.
#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <map>
#include <mutex>

#include "common.hpp"

namespace nc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// source of every padding / out-of-range lane of an LDS-DMA copy: 256 B of zeros in the code object (no memset per launch)
__device__ __attribute__((aligned(256))) const float g_zero_page[64] = {};

constexpr int kMaxSeg = 4;
constexpr int kLds = 160 * 1024;

// one problem = one output grid; the data gradient of a stride-2 layer is four of them (parity classes) in ONE launch
// (blockIdx.z), so that the chip sees all their tiles at once
struct SClass {
  const float* wp;    // packed weights
  int Hu, Wu;         // output grid: pixel (u, v) -> (oy0 + so * u, ox0 + so * v)
  int oy0, ox0;
  int ay, ax;         // tap (ty, tx) of pixel (u, v) reads input (u * si + ay + ty * dt, v * si + ax + tx * dt)
  int xlo, Wp;        // staged columns [xlo, xlo + Wp)
  int rlo_off;        // rows needed by output rows [ulo, uhi]: [ulo * si + rlo_off, uhi * si + rlo_off + rspan)
  long ncol;          // B * Hu * Wu
};
struct SParams {
  const float* x;     // input  [B][C][Hi][Wi]
  const float* bias;  // nullable
  float* y;           // output [B][M][Hf][Wf]
  const float* zeros; // >= 4 B of zeros in global memory
  int B, C, M, Hi, Wi, Hf, Wf;
  int so, si, rspan;
  int CK, CS;         // channels per chunk, floats per channel image in LDS (capacity of the staged segments)
  SClass cls[4];
};

template <int TY, int TX, int DT, int WM, int WN, int VB>
__global__ void __launch_bounds__(WM* WN * 64) k_sconv(const SParams P) {
  // flatten (shared fields, this block's class) into the names the body uses
  struct {
    const float *x, *wp, *bias, *zeros;
    float* y;
    int B, C, M, Hi, Wi, Hf, Wf, Hu, Wu, oy0, ox0, so, si, ay, ax, xlo, Wp, rlo_off, rspan, CK, CS;
    long ncol;
  } p;
  {
    const SClass& c = P.cls[blockIdx.z];
    p.x = P.x; p.wp = c.wp; p.bias = P.bias; p.zeros = P.zeros; p.y = P.y;
    p.B = P.B; p.C = P.C; p.M = P.M; p.Hi = P.Hi; p.Wi = P.Wi; p.Hf = P.Hf; p.Wf = P.Wf;
    p.Hu = c.Hu; p.Wu = c.Wu; p.oy0 = c.oy0; p.ox0 = c.ox0; p.so = P.so; p.si = P.si; p.ay = c.ay; p.ax = c.ax;
    p.xlo = c.xlo; p.Wp = c.Wp; p.rlo_off = c.rlo_off; p.rspan = P.rspan; p.CK = P.CK; p.CS = P.CS; p.ncol = c.ncol;
  }
  constexpr int NT = WM * WN * 64, NW = WM * WN, T = TY * TX, NCT = WM * VB * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // LDS: [segment descriptors: 64 ints][tab: CS ints][buf0: CK * CS][buf1: CK * CS]
  int* const seg_b = reinterpret_cast<int*>(lds);
  int* const seg_r0 = seg_b + kMaxSeg;
  int* const seg_rows = seg_r0 + kMaxSeg;
  int* const seg_off = seg_rows + kMaxSeg;
  int* const tab = reinterpret_cast<int*>(lds) + 64;
  float* const buf0 = lds + 64 + p.CS;
  float* const buf1 = buf0 + (long)p.CK * p.CS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, h = lane >> 5;
  const long j0 = (long)blockIdx.x * NCT;
  if (j0 >= p.ncol) return;  // (classes of different size share one grid)
  const int cot = blockIdx.y * WN + wn;
  const int HW = p.Hu * p.Wu;
  const long Sin = (long)p.Hi * p.Wi, Sf = (long)p.Hf * p.Wf;

  // ---- segments: the images this tile's columns touch and the input rows they need
  if (tid == 0) {
    long jl = j0 + NCT - 1;
    if (jl >= p.ncol) jl = p.ncol - 1;
    const int b0 = (int)(j0 / HW), b1 = (int)(jl / HW);
    const int u0 = (int)((j0 - (long)b0 * HW) / p.Wu), u1 = (int)((jl - (long)b1 * HW) / p.Wu);
    int off = 0;
    for (int s = 0; s < kMaxSeg; ++s) {
      const int b = b0 + s;
      if (b <= b1) {
        const int ulo = b == b0 ? u0 : 0, uhi = b == b1 ? u1 : p.Hu - 1;
        seg_b[s] = b;
        seg_r0[s] = ulo * p.si + p.rlo_off;
        seg_rows[s] = (uhi - ulo) * p.si + p.rspan;
      } else {
        seg_b[s] = -1; seg_r0[s] = 0; seg_rows[s] = 0;
      }
      seg_off[s] = off;
      off += seg_rows[s] * p.Wp;
    }
    seg_off[kMaxSeg] = off;
  }
  __syncthreads();
  const int used = seg_off[kMaxSeg];  // floats of a channel image actually staged (<= CS)

  // ---- per-tile source table: slot s of a channel image -> element offset inside channel 0 of the batch, or -1 (zero)
  for (int s = tid; s < used; s += NT) {
    int sg = 0;
#pragma unroll
    for (int q = 1; q < kMaxSeg; ++q)
      if (s >= seg_off[q] && seg_rows[q] > 0) sg = q;
    const int e = s - seg_off[sg];
    const int r = e / p.Wp, c = e - r * p.Wp;
    const int iy = seg_r0[sg] + r, ix = p.xlo + c;
    const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    tab[s] = ok ? (int)((long)seg_b[sg] * p.C * Sin + (long)iy * p.Wi + ix) : -1;
  }

  // ---- this lane's columns: base offset into a channel image, and the output address
  int base[VB];
  long oaddr[VB];
#pragma unroll
  for (int v = 0; v < VB; ++v) {
    const long j = j0 + (wm * VB + v) * 32 + li;
    base[v] = 0; oaddr[v] = -1;
    if (j < p.ncol) {
      const int b = (int)(j / HW);
      const int pix = (int)(j - (long)b * HW);
      const int u = pix / p.Wu, vv = pix - u * p.Wu;
      const int sg = b - seg_b[0];
      base[v] = seg_off[sg] + (u * p.si + p.ay - seg_r0[sg]) * p.Wp + (vv * p.si + p.ax - p.xlo);
      oaddr[v] = ((long)b * p.M + cot * 64) * Sf + (long)(p.oy0 + p.so * u) * p.Wf + p.ox0 + p.so * vv;
    }
  }
  __syncthreads();  // table complete

  const int nchunks = p.C / p.CK;
  auto stage = [&](int chunk, float* bd) {
    const float* xc = p.x + (long)chunk * p.CK * Sin;
#pragma unroll 1
    for (int ci = 0; ci < p.CK; ++ci) {
#pragma unroll 1
      for (int s0 = wave * 64; s0 < used; s0 += NW * 64) {
        const int s = s0 + lane;
        const int o = s < used ? tab[s] : -1;
        const float* src = o >= 0 ? xc + (long)ci * Sin + o : g_zero_page;
        nc_dma_lds4(src, nc_lds_addr((bd + (long)ci * p.CS + s0)));
      }
    }
  };

  f32x16 acc[2][VB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int v = 0; v < VB; ++v)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][v][e] = 0.f;

  // weights: [cot][channel pair][tap][h][32][2]: one float2 per lane per k-step, scalar offset walks the stream
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, 0x7fffffff, 0x00020000);
  const int avoff = (h * 32 + li) * 8;  // bytes
  const int kstep_b = 64 * 2 * 4;       // bytes per k-step
  int aptr = (int)((long)cot * (p.C / 2) * T * kstep_b);
  auto wload = [&](int soff) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i r = __builtin_amdgcn_raw_buffer_load_b64(wrsrc, avoff, soff, 0);
    return make_float2(__int_as_float(r.x), __int_as_float(r.y));
  };

  stage(0, buf0);
  // The weights of channel pair i + 1 are requested while pair i is multiplied (the pairs of all chunks are one linear stream): a
  // request issued right in front of its use waited out the L2 round trip every pair -- and, vmcnt retiring in order, the staging
  // requests of the next chunk issued just before it.  (The last pair re-requests itself.)
  float2 aw[T], awn[T];
#pragma unroll
  for (int t = 0; t < T; ++t) aw[t] = wload(aptr + t * kstep_b);
  const int npairs_all = p.C / 2;
  int pair_i = 0;
  for (int q = 0; q < nchunks; ++q) {
    float* cur = (q & 1) ? buf1 : buf0;
    float* nxt = (q & 1) ? buf0 : buf1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (q + 1 < nchunks) stage(q + 1, nxt);
#pragma unroll 1
    for (int cp = 0; cp < p.CK / 2; ++cp) {
      const float* cb = cur + (long)(2 * cp + h) * p.CS;
      ++pair_i;
      if (pair_i < npairs_all) aptr += T * kstep_b;
#pragma unroll
      for (int t = 0; t < T; ++t) awn[t] = wload(aptr + t * kstep_b);
#pragma unroll
      for (int ty = 0; ty < TY; ++ty)
#pragma unroll
        for (int tx = 0; tx < TX; ++tx) {
          const int toff = (ty * p.Wp + tx) * DT;
          float bv[VB];
#pragma unroll
          for (int v = 0; v < VB; ++v) bv[v] = cb[base[v] + toff];
#pragma unroll
          for (int v = 0; v < VB; ++v) {
            acc[0][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ty * TX + tx].x, bv[v], acc[0][v], 0, 0, 0);
            acc[1][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ty * TX + tx].y, bv[v], acc[1][v], 0, 0, 0);
          }
        }
#pragma unroll
      for (int t = 0; t < T; ++t) aw[t] = awn[t];
    }
  }

  // ---- epilogue: rows = output channels, lanes = columns
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    float bvs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bvs[e] = p.bias ? p.bias[cot * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] : 0.f;
#pragma unroll
    for (int v = 0; v < VB; ++v)
      if (oaddr[v] >= 0) {
        float* yo = p.y + oaddr[v];
#pragma unroll
        for (int e = 0; e < 16; ++e) yo[(long)(a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * Sf] = acc[a][v][e] + bvs[e];
      }
  }
}

// A[m][c][t] = w[m * sm + c * sc + tapoff(t)], tapoff(ty, tx) = (ty0 + tys * ty) * kw + tx0 + txs * tx
// -> wp[cot][c / 2][t][c & 1][m % 32][(m / 32) % 2]
__global__ void __launch_bounds__(256) k_pack_sconv(const float* __restrict__ w, float* __restrict__ wp, int C, int M, int TY, int TX,
                                                    long sm, long sc, int ty0, int tys, int tx0, int txs, int kw, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int T = TY * TX;
  const int a = (int)(i & 1);
  long q = i >> 1;
  const int li = (int)(q & 31); q >>= 5;
  const int h = (int)(q & 1); q >>= 1;
  const int t = (int)(q % T); q /= T;
  const int cp = (int)(q % (C / 2));
  const int cot = (int)(q / (C / 2));
  const int m = cot * 64 + a * 32 + li, c = 2 * cp + h;
  const int ty = t / TX, tx = t - ty * TX;
  wp[i] = w[(long)m * sm + (long)c * sc + (ty0 + tys * ty) * kw + tx0 + txs * tx];
}

struct SPlan {
  int WM, WN, VB, CK, CS, lds;
  bool ok;
  long tiles;
};
// Below this many tiles the problem does not fill the chip (a workgroup walks its whole reduction alone): the gather GEMM,
// which splits the reduction over grid.z, serves such small batches (the Apollo step's 1-4 planes per discriminator)
static long min_tiles() {
  static const long v = 192;
  return v;
}
// LDS per workgroup is capped well below the CU's 160 KB so that two workgroups (of this launch, or of the launches of other
// discriminators on other streams) can share a CU: one's barrier / staging phase hides under the other's MFMAs
static long lds_cap() {
  static const long v = 80 * 1024;
  return v < kLds ? v : kLds;
}
struct SCfg { int WM, WN, VB; };
// tiles: 64 * WN output channels x 32 * WM * VB columns.  Several shapes so that the number of tiles can be matched to
// whole rounds of 256 workgroups (M = 64..512, 10^4..10^5 columns)
static const SCfg kSCfgs[] = {{4, 2, 2}, {8, 1, 1}, {5, 2, 1}, {6, 2, 1}, {3, 2, 2}, {4, 2, 1}, {6, 1, 1}, {5, 1, 1},
                              {2, 2, 2}, {4, 1, 2}, {3, 1, 2}, {2, 1, 2}, {2, 2, 3}, {2, 1, 3}};
static int forced_cfg() {  // NC_SCONV_CFG=i: only configuration i (timing experiments, tools/sconv_layers.py)
  static const int v = getenv("NC_SCONV_CFG") ? atoi(getenv("NC_SCONV_CFG")) : -1;
  return v;
}

// the plan of ONE tile configuration (ok = false: not applicable to this problem)
SPlan plan_one(const SCfg& g, int B, int C, int M, int Hu, int Wu, int si, int rspan, int Wp, int nclass) {
  SPlan pl{};
  if (M % (g.WN * 64)) return pl;
  const int NCT = g.WM * g.VB * 32;
  const long HW = (long)Hu * Wu;
  if ((NCT - 1) / HW + 2 > kMaxSeg) return pl;  // a tile may touch floor((NCT - 1) / HW) + 2 images
  // rows: NCT columns cover at most ceil(NCT / Wu) + 1 output rows per image chain, each image adds its halo
  const int urows = (NCT + Wu - 1) / Wu + 1;
  const int nimg = (int)((NCT - 1) / HW) + 2;
  const long rows = (long)(urows - 1) * si + (long)nimg * rspan + (nimg - 1) * si;
  const int CS = (int)(((rows * Wp + 63) / 64) * 64);
  for (int CK : {32, 16, 8, 4, 2}) {
    if (C % CK) continue;
    const long bytes = (64 + (long)CS + 2L * CK * CS) * 4;
    if (bytes > lds_cap()) continue;
    const long tiles = cdiv((long)B * HW, NCT) * (M / (64 * g.WN)) * nclass;
    return SPlan{g.WM, g.WN, g.VB, CK, CS, (int)bytes, true, tiles};
  }
  return pl;
}

// heuristic choice (used for the "is it worth it" decision and when tuning is off): how full the rounds of 256 workgroups
// are, times a mild preference for larger tiles (operand reuse)
SPlan plan_sconv(int B, int C, int M, int Hu, int Wu, int si, int rspan, int Wp, int nclass = 1) {
  SPlan best{};
  double best_eff = 0;
  int gi = -1;
  for (const SCfg& g : kSCfgs) {
    ++gi;
    if (forced_cfg() >= 0 && gi != forced_cfg()) continue;
    const SPlan pl = plan_one(g, B, C, M, Hu, Wu, si, rspan, Wp, nclass);
    if (!pl.ok) continue;
    const int NCT = g.WM * g.VB * 32;
    const double eff = (double)pl.tiles / (double)(cdiv(pl.tiles, 256) * 256) * (0.85 + 0.15 * (g.WN * NCT) / 512.0);
    if (!best.ok || eff > best_eff) {
      best = pl;
      best_eff = eff;
    }
  }
  return best;
}

// ---- run-time choice of the tile configuration.  Which shape wins depends on how the tile count falls on 256 CUs x
// 1-3 resident workgroups, on the LDS per workgroup and on the reduction length, and no closed form predicted the measured
// order (tools/sconv_sweep.py: the best shape beats the heuristic's by up to 30 %).  So the first call of a problem times
// every applicable configuration on an otherwise idle device and the winner is cached per (problem, direction).  Every
// configuration accumulates an output element in the same order (chunk, channel pair, tap: one MFMA chain), so the result
// does not depend on the choice -- bit for bit (tests/test_gpu_ops.py::test_image_staged_configs_agree).
struct TuneKey {
  int v[10];
  bool operator<(const TuneKey& o) const { return std::lexicographical_compare(v, v + 10, o.v, o.v + 10); }
};
std::map<TuneKey, int> g_tuned;
std::mutex g_tune_mu;
int g_cfg_override = -1;  // nc_sconv_set_cfg (tests)
std::atomic<int> g_tune_mode{-1};  // nc_sconv_set_tune: 0 never time (heuristic plan), 1 time on first use, -1 = NC_SCONV_TUNE (default 1)
bool tune_on() {
  const int m = g_tune_mode.load();
  if (m >= 0) return m != 0;
  static const bool on = !(getenv("NC_SCONV_TUNE") && atoi(getenv("NC_SCONV_TUNE")) == 0);
  return on;
}

template <class L>
int run_tuned(int dir, int B, int C, int M, int Hu, int Wu, int si, int rspan, int Wp, int nclass, hipStream_t s, L&& launch) {
  const int ncfg = (int)(sizeof(kSCfgs) / sizeof(kSCfgs[0]));
  if (g_cfg_override >= 0 && g_cfg_override < ncfg) {
    const SPlan pl = plan_one(kSCfgs[g_cfg_override], B, C, M, Hu, Wu, si, rspan, Wp, nclass);
    if (!pl.ok) { set_error("sconv: configuration %d does not apply", g_cfg_override); return NC_ERR_SHAPE; }
    return launch(pl);
  }
  if (forced_cfg() >= 0 || !tune_on()) {
    const SPlan pl = plan_sconv(B, C, M, Hu, Wu, si, rspan, Wp, nclass);
    if (!pl.ok) { set_error("sconv: no plan"); return NC_ERR_SHAPE; }
    return launch(pl);
  }
  const TuneKey key{{dir, B, C, M, Hu, Wu, si, rspan, Wp, nclass}};
  int cfg = -1;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) cfg = it->second;
  }
  if (cfg < 0) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (cap == hipStreamCaptureStatusNone && hipDeviceSynchronize() == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
        hipEventCreate(&e1) == hipSuccess) {
      float best_t = 0.f;
      for (int i = 0; i < ncfg; ++i) {
        const SPlan pl = plan_one(kSCfgs[i], B, C, M, Hu, Wu, si, rspan, Wp, nclass);
        if (!pl.ok) continue;
        if (launch(pl)) continue;  // warm-up (first use of this instantiation: attribute call, code load)
        (void)hipEventRecord(e0, s);
        int err = 0;
        for (int r = 0; r < 3 && !err; ++r) err = launch(pl);
        (void)hipEventRecord(e1, s);
        float ms = 0.f;
        if (err || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
        if (cfg < 0 || ms < best_t) { cfg = i; best_t = ms; }
      }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    if (cfg < 0) {  // (capturing, or nothing could be timed): the heuristic, not cached
      const SPlan pl = plan_sconv(B, C, M, Hu, Wu, si, rspan, Wp, nclass);
      if (!pl.ok) { set_error("sconv: no plan"); return NC_ERR_SHAPE; }
      return launch(pl);
    }
    g_tuned[key] = cfg;
  }
  return launch(plan_one(kSCfgs[cfg], B, C, M, Hu, Wu, si, rspan, Wp, nclass));
}

template <int TY, int TX, int DT, int WM, int WN, int VB>
int launch_sconv_cfg(const SPlan& pl, const SParams& p, long max_ncol, int nclass, hipStream_t s) {
  auto kern = k_sconv<TY, TX, DT, WM, WN, VB>;
  if (int e = raise_dyn_lds(kern, kLds, "sconv")) return e;
  const dim3 grid((unsigned)cdiv(max_ncol, WM * VB * 32), (unsigned)(p.M / (64 * WN)), (unsigned)nclass);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), pl.lds, s, p);
  return check_launch("sconv");
}

template <int TY, int TX, int DT>
int launch_sconv(const SPlan& pl, const SParams& p, long max_ncol, int nclass, hipStream_t s) {
  const int key = pl.WM * 100 + pl.WN * 10 + pl.VB;
  switch (key) {
    case 422: return launch_sconv_cfg<TY, TX, DT, 4, 2, 2>(pl, p, max_ncol, nclass, s);
    case 811: return launch_sconv_cfg<TY, TX, DT, 8, 1, 1>(pl, p, max_ncol, nclass, s);
    case 521: return launch_sconv_cfg<TY, TX, DT, 5, 2, 1>(pl, p, max_ncol, nclass, s);
    case 621: return launch_sconv_cfg<TY, TX, DT, 6, 2, 1>(pl, p, max_ncol, nclass, s);
    case 322: return launch_sconv_cfg<TY, TX, DT, 3, 2, 2>(pl, p, max_ncol, nclass, s);
    case 421: return launch_sconv_cfg<TY, TX, DT, 4, 2, 1>(pl, p, max_ncol, nclass, s);
    case 611: return launch_sconv_cfg<TY, TX, DT, 6, 1, 1>(pl, p, max_ncol, nclass, s);
    case 511: return launch_sconv_cfg<TY, TX, DT, 5, 1, 1>(pl, p, max_ncol, nclass, s);
    case 222: return launch_sconv_cfg<TY, TX, DT, 2, 2, 2>(pl, p, max_ncol, nclass, s);
    case 412: return launch_sconv_cfg<TY, TX, DT, 4, 1, 2>(pl, p, max_ncol, nclass, s);
    case 312: return launch_sconv_cfg<TY, TX, DT, 3, 1, 2>(pl, p, max_ncol, nclass, s);
    case 212: return launch_sconv_cfg<TY, TX, DT, 2, 1, 2>(pl, p, max_ncol, nclass, s);
    case 223: return launch_sconv_cfg<TY, TX, DT, 2, 2, 3>(pl, p, max_ncol, nclass, s);
    case 213: return launch_sconv_cfg<TY, TX, DT, 2, 1, 3>(pl, p, max_ncol, nclass, s);
  }
  set_error("sconv: unknown tile configuration");
  return NC_ERR_SHAPE;
}

bool sconv_layer_ok(const ConvDims& d) {
  // 2-D, 4 x 4, padding 1, stride 1 or 2 -- the PatchGAN layers
  return d.D == 1 && d.kd == 1 && d.kh == 4 && d.kw == 4 && d.ph == 1 && d.pw == 1 && d.sh == d.sw && (d.sh == 1 || d.sh == 2) &&
         (long)d.N * d.C * d.H * d.W < (1L << 31) && (long)d.N * d.K * d.Ho * d.Wo < (1L << 31);
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- weight gradient of the same layers: dw[k][c][ty][tx] = sum over (image, u, v) of dy[k][u][v] * x[c][u*s-1+ty][v*s-1+tx]
// as a GEMM D[k][(c, tap)] whose REDUCTION axis is the flat column axis j = (image, u, v).  A workgroup (4 waves, 2 x 2) owns
// 128 dy channels x NCW = 4 * NTW input channels (x 16 taps) and one chunk of columns, walked in stages of 64 columns:
//   * lane l of every wave decodes column l of the stage once: where its dy element lives, and -- in `cbtab` -- where its
//     window starts inside the staged input rows;
//   * dy[128][64] and the input rows the 64 columns touch (<= 2 images) are staged by LDS-DMA: one dword per lane, dy rows
//     are contiguous columns, input rows are copied as rows (pitch = 4 mod 8 dwords: the four tap rows fall on disjoint banks);
//   * the two k values of an MFMA are two consecutive columns: A = dy[channel li][col + h], B = x[cbtab[col + h] + lane tap];
//     a lane's (channel, tap) offset is a loop constant, so an operand is one ds_read_b32, and nothing is gathered.
// The column chunks are summed in a fixed order by k_swgrad_reduce (deterministic; no atomics).
constexpr int kWNJ = 64;          // columns per stage = lanes per wave
constexpr int kWAP = kWNJ + 1;    // pitch of a dy row in LDS (odd: channels li = 0..31 fall on distinct banks)

struct WParams {
  const float* x;      // [B][C][Hi][Wi]
  const float* dy;     // [B][K][Hu][Wu]
  float* out;          // [splits][K][C * 16] (or dw itself when splits == 1)
  const float* zeros;
  int B, C, K, Hi, Wi, Hu, Wu, s;
  int pitch, CS;       // staged input rows: dwords per row, per channel image
  long ncol;           // B * Hu * Wu
  int stages_per_split, splits, ngroups, mtiles;
};

template <int NTW>
__global__ void __launch_bounds__(256, 2) k_swgrad(const WParams p) {
  constexpr int NCW = 4 * NTW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int* const cbtab = reinterpret_cast<int*>(lds);   // [64 + 1]
  float* const As = lds + 128;                      // [128][kWAP]
  float* const Xs = As + 128 * kWAP;                // [NCW][CS]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int li = lane & 31, h = lane >> 5;

  // block -> (channel group, dy-channel tile, column chunk): blocks that share a dy tile (same tile, same chunk) are
  // neighbours in `lin`, and `lin` is laid out so that neighbours run on the same XCD (same L2)
  int lin = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) lin = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int ng = lin % p.ngroups;
  const int rest = lin / p.ngroups;
  const int mt = rest % p.mtiles, split = rest / p.mtiles;
  const int c0 = ng * NCW, m0 = mt * 128;

  const int HW = p.Hu * p.Wu;
  const long Sin = (long)p.Hi * p.Wi;
  const long jbeg = (long)split * p.stages_per_split * kWNJ;
  long jend = jbeg + (long)p.stages_per_split * kWNJ;
  if (jend > p.ncol) jend = p.ncol;

  f32x16 acc[2][NTW];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.f;

  for (int i = tid; i < NCW * p.CS; i += 256) Xs[i] = 0.f;  // padding columns stay zero for the whole kernel
  // this lane's (channel, tap) inside n-tile 0 of its wave; n-tile t adds 2 * CS
  const int tap = li & 15;
  const int laneoff = ((wn * NTW * 2) + (li >> 4)) * p.CS + (tap >> 2) * p.pitch + (tap & 3);
  const float* const arow0 = As + (wm * 64 + li) * kWAP + h;
  const float* const arow1 = arow0 + 32 * kWAP;

#pragma unroll 1
  for (long js = jbeg; js < jend; js += kWNJ) {
    // ---- decode column `lane` of this stage
    const long j = js + lane;
    const bool valid = j < jend;
    const long jc = valid ? j : js;
    const int b = (int)(jc / HW);
    const int pix = (int)(jc - (long)b * HW);
    const int u = pix / p.Wu, v = pix - u * p.Wu;
    const int cnt = (int)(jend - js < kWNJ ? jend - js : kWNJ);
    const int b0 = __builtin_amdgcn_readfirstlane(b), u0 = __builtin_amdgcn_readfirstlane(u);
    const int b1 = __builtin_amdgcn_readlane(b, cnt - 1), u1 = __builtin_amdgcn_readlane(u, cnt - 1);
    const int uhi0 = b1 == b0 ? u1 : p.Hu - 1;
    const int rows0 = (uhi0 - u0) * p.s + 4;
    const int rows1 = b1 > b0 ? u1 * p.s + 4 : 0;
    const int sg = b - b0;
    const int cb = ((sg ? rows0 : 0) + (u - (sg ? 0 : u0)) * p.s) * p.pitch + v * p.s;
    const long dyoff = valid ? ((long)b * p.K + m0) * HW + pix : -1;

    __syncthreads();  // the previous stage's operand reads are done
    if (wave == 0) cbtab[lane] = valid ? cb : 0;
    // ---- stage dy: 32 channels per wave, lanes = columns; 32-bit offsets from the uniform base p.dy; a column past the end
    //      of the chunk (last stage only) is written as zero by its lane
    {
      const unsigned dyo = (unsigned)(valid ? dyoff : 0);
      if (valid) {
#pragma unroll 8
        for (int ch = wave * 32; ch < wave * 32 + 32; ++ch)
          nc_dma_lds4(((const char*)p.dy + (dyo + (unsigned)(ch * HW)) * 4u), nc_lds_addr((As + ch * kWAP)));
      } else {
        for (int ch = wave * 32; ch < wave * 32 + 32; ++ch) As[ch * kWAP + lane] = 0.f;
      }
    }
    // ---- stage the input rows: channels wave, wave + 4, ...; one row (<= 64 dwords) per instruction.  The padding
    //      columns (ix = -1, ix >= Wi) are never written (lanes masked off; zeroed once above); a padding ROW is written as
    //      zeros by the wave.  Lane r decodes row r once per stage (element offset of its column 0 inside channel 0 of its
    //      image, or -1 for a padding row); a copy is then that offset + the channel + the lane - 1, off the uniform base p.x
    {
      const int rows = rows0 + rows1;
      int rowoff;
      {
        const int r = lane;
        const bool s1 = r >= rows0;
        const int iy = s1 ? r - rows0 - 1 : u0 * p.s - 1 + r;
        rowoff = (r < rows && (unsigned)iy < (unsigned)p.Hi) ? (s1 ? b1 : b0) * p.C * (int)Sin + iy * p.Wi : -1;
      }
      const bool colok = (unsigned)(lane - 1) < (unsigned)p.Wi;
#pragma unroll 1
      for (int ci = wave; ci < NCW; ci += 4) {
        const int co = (c0 + ci) * (int)Sin;
        float* const xd = Xs + ci * p.CS;
#pragma unroll 1
        for (int r = 0; r < rows; ++r) {
          const int ro = __builtin_amdgcn_readlane(rowoff, r);
          if (ro >= 0) {
            if (colok && lane < p.pitch)
              nc_dma_lds4(((const char*)p.x + (unsigned)(ro + co + lane - 1) * 4u), nc_lds_addr((xd + r * p.pitch)));
#pragma unroll 1
            for (int cc = 64; cc < p.pitch; cc += 64) {
              const int col = cc + lane;
              if ((unsigned)(col - 1) < (unsigned)p.Wi)
                nc_dma_lds4(((const char*)p.x + (unsigned)(ro + co + col - 1) * 4u), nc_lds_addr((xd + r * p.pitch + cc)));
            }
          } else {
            for (int col = lane; col < p.pitch; col += 64) xd[r * p.pitch + col] = 0.f;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- 32 k-steps of two columns
    const float* const xb = Xs + laneoff;
#pragma unroll 8
    for (int jj = 0; jj < kWNJ; jj += 2) {
      const int cbv = cbtab[jj + h];
      const float a0 = arow0[jj], a1 = arow1[jj];
      float bv[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) bv[t] = xb[cbv + t * 2 * p.CS];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[t], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[t], acc[1][t], 0, 0, 0);
      }
    }
  }

  // ---- partial result: rows = dy channels, lanes = (input channel, tap) -- contiguous in dw
  const long N16 = (long)p.C * 16;
  float* o = p.out + (long)split * p.K * N16;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const long n = (long)c0 * 16 + (wn * NTW + t) * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        o[(long)m * N16 + n] = acc[a][t][e];
      }
    }
}

__global__ void __launch_bounds__(256) k_swgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, long total, int splits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float s = 0.f;
  for (int q = 0; q < splits; ++q) s += part[(long)q * total + i];
  dw[i] = s;
}

struct WPlan {
  bool ok;
  int NTW, pitch, CS, lds, ngroups, mtiles, splits, stages_per_split;
};
WPlan plan_swgrad(const ConvDims& d) {
  WPlan w{};
  const int s = d.sh;
  const long HW = (long)d.Ho * d.Wo;
  if (HW < kWNJ) return w;  // a stage of 64 columns may touch two images at most
  const int Wp = (d.Wo - 1) * s + 4;
  int pitch = (Wp + 3) & ~3;
  if ((pitch & 7) == 0) pitch += 4;
  const int urows = (kWNJ + d.Wo - 1) / d.Wo + 1;
  const int rows = (urows - 1) * s + 2 * 4 + s;
  const int CS = rows * pitch + 16;  // (+16: two channels of an n-tile on different bank groups when rows * pitch is a multiple of 32)
  static const int only = 0;
  for (int NTW : {4, 2}) {
    if (only && NTW != only) continue;
    const int NCW = 4 * NTW;
    if (d.C % NCW) continue;
    const long bytes = (128 + 128L * kWAP + (long)NCW * CS) * 4;
    if (bytes > lds_cap()) continue;
    w.ok = true; w.NTW = NTW; w.pitch = pitch; w.CS = CS; w.lds = (int)bytes;
    w.ngroups = d.C / NCW; w.mtiles = d.K / 128;
    const long nst = cdiv((long)d.N * HW, kWNJ);
    const long tiles = (long)w.ngroups * w.mtiles;
    static const long target = 512;
    long splits = cdiv(target, tiles);  // 2 resident workgroups per CU, one round (measured: 512 beats 256 / 1024)
    if (splits > nst) splits = nst;
    if (splits > 256) splits = 256;
    w.stages_per_split = (int)cdiv(nst, splits);
    w.splits = (int)cdiv(nst, w.stages_per_split);
    return w;
  }
  return w;
}

}  // namespace

bool sconv_wgrad_supported(const ConvDims& d) {
  if (!sconv_layer_ok(d) || d.K % 128 || d.C % 8) return false;
  // the staging copies address x and dy by 32-bit BYTE offsets from their base pointers
  if ((long)d.N * d.C * d.H * d.W >= (1L << 30) || (long)d.N * d.K * d.Ho * d.Wo >= (1L << 30)) return false;
  const WPlan w = plan_swgrad(d);
  // worth it once the reduction is long (batched planes); a few planes stay on the gather GEMM
  static const long mincol = 4096;
  return w.ok && (long)d.N * d.Ho * d.Wo >= mincol;
}
size_t sconv_wgrad_ws_bytes(const ConvDims& d) {
  const WPlan w = plan_swgrad(d);
  if (!w.ok) return 0;
  return align256((size_t)w.splits * d.K * d.C * 16 * sizeof(float)) + 512;
}

int conv_wgrad_sconv(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  const WPlan w = plan_swgrad(d);
  if (!w.ok) { set_error("sconv_wgrad: no plan"); return NC_ERR_SHAPE; }
  if (!ws || wsb < sconv_wgrad_ws_bytes(d)) { set_error("sconv_wgrad: workspace too small"); return NC_ERR_WS; }
  const size_t pbytes = align256((size_t)w.splits * d.K * d.C * 16 * sizeof(float));
  float* part = (float*)ws;
  float* zeros = (float*)((char*)ws + pbytes);
  WParams p{};
  p.x = x; p.dy = dy; p.out = w.splits == 1 ? dw : part; p.zeros = zeros;
  p.B = d.N; p.C = d.C; p.K = d.K; p.Hi = d.H; p.Wi = d.W; p.Hu = d.Ho; p.Wu = d.Wo; p.s = d.sh;
  p.pitch = w.pitch; p.CS = w.CS; p.ncol = (long)d.N * d.Ho * d.Wo;
  p.stages_per_split = w.stages_per_split; p.splits = w.splits; p.ngroups = w.ngroups; p.mtiles = w.mtiles;
  const dim3 grid((unsigned)((long)w.ngroups * w.mtiles * w.splits));
  auto launch = [&](auto kern) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) {
      set_error("sconv_wgrad: cannot raise dynamic LDS limit");
      return NC_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), w.lds, s, p);
    return check_launch("sconv_wgrad");
  };
  if (int e = w.NTW == 4 ? launch(k_swgrad<4>) : launch(k_swgrad<2>)) return e;
  if (w.splits > 1) {
    const long total = (long)d.K * d.C * 16;
    hipLaunchKernelGGL(k_swgrad_reduce, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, part, dw, total, w.splits);
    return check_launch("sconv_wgrad_reduce");
  }
  return NC_OK;
}

namespace {

}  // namespace

// forward: M = K output channels, reduction over C
bool sconv_fwd_supported(const ConvDims& d) {
  if (!sconv_layer_ok(d) || d.C % 2 || d.C < 16 || d.K % 64) return false;
  if ((long)d.Ho * d.Wo < 96) return false;  // tiny images: too many images per tile (the gather GEMM serves them)
  const SPlan pl = plan_sconv(d.N, d.C, d.K, d.Ho, d.Wo, d.sh, 4, (d.Wo - 1) * d.sw + 4);
  return pl.ok && pl.tiles >= min_tiles();
}
// data gradient: M = C, reduction over K
bool sconv_dgrad_supported(const ConvDims& d) {
  if (!sconv_layer_ok(d) || d.K % 2 || d.K < 16 || d.C % 64) return false;
  if (d.sh == 1) {
    if ((long)d.H * d.W < 96) return false;
    const SPlan pl = plan_sconv(d.N, d.K, d.C, d.H, d.W, 1, 4, d.W + 3);
    return pl.ok && pl.tiles >= min_tiles();
  }
  const int Hu = (d.H + 1) / 2, Wu = (d.W + 1) / 2;
  if ((long)(d.H / 2) * (d.W / 2) < 96) return false;
  const SPlan pl = plan_sconv(d.N, d.K, d.C, Hu, Wu, 1, 2, Wu + 1, 4);
  return pl.ok && pl.tiles >= min_tiles();
}
size_t sconv_ws_bytes(const ConvDims& d) {
  return align256((size_t)d.C * d.K * 16 * sizeof(float)) + 512;
}

int conv_fwd_sconv(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!ws || wsb < sconv_ws_bytes(d)) { set_error("sconv_fwd: workspace too small"); return NC_ERR_WS; }
  float* wp = (float*)ws;
  float* zeros = (float*)((char*)ws + align256((size_t)d.C * d.K * 16 * sizeof(float)));
  const long total = (long)d.C * d.K * 16;
  hipLaunchKernelGGL(k_pack_sconv, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, d.C, d.K, 4, 4, (long)d.C * 16, 16L, 0, 1, 0,
                     1, 4, total);
  if (int e = check_launch("pack_sconv")) return e;
  SParams p{};
  p.x = x; p.bias = bias; p.y = y; p.zeros = zeros;
  p.B = d.N; p.C = d.C; p.M = d.K; p.Hi = d.H; p.Wi = d.W; p.Hf = d.Ho; p.Wf = d.Wo;
  p.so = 1; p.si = d.sh; p.rspan = 4;
  SClass& c = p.cls[0];
  c.wp = wp; c.Hu = d.Ho; c.Wu = d.Wo; c.oy0 = 0; c.ox0 = 0; c.ay = -1; c.ax = -1;
  c.xlo = -1; c.Wp = (d.Wo - 1) * d.sw + 4; c.rlo_off = -1; c.ncol = (long)d.N * d.Ho * d.Wo;
  const long ncol = c.ncol;
  return run_tuned(0, d.N, d.C, d.K, d.Ho, d.Wo, d.sh, 4, c.Wp, 1, s, [&](const SPlan& pl) {
    p.CK = pl.CK; p.CS = pl.CS;
    return launch_sconv<4, 4, 1>(pl, p, ncol, 1, s);
  });
}

int conv_dgrad_sconv(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!ws || wsb < sconv_ws_bytes(d)) { set_error("sconv_dgrad: workspace too small"); return NC_ERR_WS; }
  float* wp = (float*)ws;
  float* zeros = (float*)((char*)ws + align256((size_t)d.C * d.K * 16 * sizeof(float)));
  SParams p{};
  p.x = dy; p.bias = nullptr; p.y = dx; p.zeros = zeros;
  p.B = d.N; p.C = d.K; p.M = d.C; p.Hi = d.Ho; p.Wi = d.Wo; p.Hf = d.H; p.Wf = d.W;
  if (d.sh == 1) {
    // dx[iy][ix] = sum_(k, ky, kx) w[k][c][ky][kx] dy[iy + 1 - ky][ix + 1 - kx]: one 4 x 4 problem walked backwards
    const long total = (long)d.C * d.K * 16;
    hipLaunchKernelGGL(k_pack_sconv, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, d.K, d.C, 4, 4, 16L, (long)d.C * 16, 0, 1,
                       0, 1, 4, total);
    if (int e = check_launch("pack_sconv")) return e;
    p.so = 1; p.si = 1; p.rspan = 4;
    SClass& c = p.cls[0];
    c.wp = wp; c.Hu = d.H; c.Wu = d.W; c.oy0 = 0; c.ox0 = 0; c.ay = 1; c.ax = 1;
    c.xlo = -2; c.Wp = d.W + 3; c.rlo_off = -2; c.ncol = (long)d.N * d.H * d.W;
    const long ncol = c.ncol;
    return run_tuned(1, d.N, d.K, d.C, d.H, d.W, 1, 4, c.Wp, 1, s, [&](const SPlan& pl) {
      p.CK = pl.CK; p.CS = pl.CS;
      return launch_sconv<4, 4, -1>(pl, p, ncol, 1, s);
    });
  }
  // stride 2: input pixel iy = 2u + py receives the taps ky = t0y + 2 jy (t0y = (py + 1) & 1) from dy[(iy + 1 - ky) / 2]
  // = dy[u + ay - jy], ay = (py + 1 - t0y) / 2: per parity class a 2 x 2 problem walked backwards; all four in one launch
  p.so = 2; p.si = 1; p.rspan = 2;
  const int HuM = (d.H + 1) / 2, WuM = (d.W + 1) / 2;
  long max_ncol = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    const int t0y = (py + 1) & 1, t0x = (px + 1) & 1;
    const int Hu = (d.H - py + 1) / 2, Wu = (d.W - px + 1) / 2;
    const long total = (long)d.C * d.K * 4;
    float* wpc = wp + (size_t)cls * d.C * d.K * 4;
    hipLaunchKernelGGL(k_pack_sconv, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wpc, d.K, d.C, 2, 2, 16L, (long)d.C * 16, t0y,
                       2, t0x, 2, 4, total);
    if (int e = check_launch("pack_sconv")) return e;
    SClass& c = p.cls[cls];
    c.wp = wpc; c.Hu = Hu; c.Wu = Wu; c.oy0 = py; c.ox0 = px;
    c.ay = (py + 1 - t0y) / 2; c.ax = (px + 1 - t0x) / 2;
    c.xlo = c.ax - 1; c.Wp = Wu + 1; c.rlo_off = c.ay - 1; c.ncol = (long)d.N * Hu * Wu;
    if (c.ncol > max_ncol) max_ncol = c.ncol;
  }
  return run_tuned(2, d.N, d.K, d.C, HuM, WuM, 1, 2, WuM + 1, 4, s, [&](const SPlan& pl) {
    p.CK = pl.CK; p.CS = pl.CS;
    return launch_sconv<2, 2, -1>(pl, p, max_ncol, 4, s);
  });
}

void sconv_set_cfg(int cfg) { g_cfg_override = cfg; }
void sconv_set_tune(int on) { g_tune_mode = on; }

}  // namespace nc
