// "Image-staged" implicit-GEMM convolution for the 2-D PatchGAN layers (reference models/networks.py:1030-1057: Conv2d
// k 4, stride 2 / 1, padding 1) on the fp32 matrix cores: forward, and the data gradient as forward-shaped problems.
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
#include <cstdlib>

#include "common.hpp"

namespace nc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

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
        const float* src = o >= 0 ? xc + (long)ci * Sin + o : p.zeros;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bd + (long)ci * p.CS + s0), 4, 0, 0);
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
  for (int q = 0; q < nchunks; ++q) {
    float* cur = (q & 1) ? buf1 : buf0;
    float* nxt = (q & 1) ? buf0 : buf1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (q + 1 < nchunks) stage(q + 1, nxt);
#pragma unroll 1
    for (int cp = 0; cp < p.CK / 2; ++cp) {
      const float* cb = cur + (long)(2 * cp + h) * p.CS;
      float2 aw[T];
#pragma unroll
      for (int t = 0; t < T; ++t) aw[t] = wload(aptr + t * kstep_b);
      aptr += T * kstep_b;
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
constexpr long kMinTiles = 192;
// LDS per workgroup is capped well below the CU's 160 KB so that two workgroups (of this launch, or of the launches of other
// discriminators on other streams) can share a CU: one's barrier / staging phase hides under the other's MFMAs
static long lds_cap() {
  static const long v = getenv("NC_SCONV_LDS_KB") ? atol(getenv("NC_SCONV_LDS_KB")) * 1024 : 80 * 1024;
  return v < kLds ? v : kLds;
}
struct SCfg { int WM, WN, VB; };
// tiles: 64 * WN output channels x 32 * WM * VB columns.  Several shapes so that the number of tiles can be matched to
// whole rounds of 256 workgroups (M = 64..512, 10^4..10^5 columns)
static const SCfg kSCfgs[] = {{4, 2, 2}, {8, 1, 1}, {5, 2, 1}, {6, 2, 1}, {3, 2, 2}, {4, 2, 1}, {6, 1, 1}, {5, 1, 1}};

// rows_max: the most rows of one channel image a tile of NCT columns can need
SPlan plan_sconv(int B, int C, int M, int Hu, int Wu, int si, int rspan, int Wp, int nclass = 1) {
  SPlan best{};
  double best_eff = 0;
  for (const SCfg& g : kSCfgs) {
    if (M % (g.WN * 64)) continue;
    const int NCT = g.WM * g.VB * 32;
    const long HW = (long)Hu * Wu;
    if ((NCT - 1) / HW + 2 > kMaxSeg) continue;  // a tile may touch floor((NCT - 1) / HW) + 2 images
    // rows: NCT columns cover at most ceil(NCT / Wu) + 1 output rows per image chain, each image adds its halo
    const int urows = (NCT + Wu - 1) / Wu + 1;
    const int nimg = (int)((NCT - 1) / HW) + 2;
    const long rows = (long)(urows - 1) * si + (long)nimg * rspan + (nimg - 1) * si;
    const int CS = (int)(((rows * Wp + 63) / 64) * 64);
    for (int CK : {32, 16, 8, 4, 2}) {
      if (C % CK) continue;
      const long bytes = (64 + (long)CS + 2L * CK * CS) * 4;
      if (bytes > lds_cap()) continue;
      // efficiency: how full the rounds of 256 workgroups (one per CU) are
      const long tiles = cdiv((long)B * HW, NCT) * (M / (64 * g.WN)) * nclass;
      // how full the rounds of 256 workgroups are, times a mild preference for larger tiles (operand reuse)
      const double eff = (double)tiles / (double)(cdiv(tiles, 256) * 256) * (0.85 + 0.15 * (g.WN * NCT) / 512.0);
      if (!best.ok || eff > best_eff) {
        best = SPlan{g.WM, g.WN, g.VB, CK, CS, (int)bytes, true, tiles};
        best_eff = eff;
      }
      break;
    }
  }
  return best;
}

template <int TY, int TX, int DT, int WM, int WN, int VB>
int launch_sconv_cfg(const SPlan& pl, const SParams& p, long max_ncol, int nclass, hipStream_t s) {
  auto kern = k_sconv<TY, TX, DT, WM, WN, VB>;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) {
      set_error("sconv: cannot raise dynamic LDS limit");
      return NC_ERR_HIP;
    }
    done = true;
  }
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

}  // namespace

// forward: M = K output channels, reduction over C
bool sconv_fwd_supported(const ConvDims& d) {
  if (!sconv_layer_ok(d) || d.C % 2 || d.C < 16 || d.K % 64) return false;
  if ((long)d.Ho * d.Wo < 96) return false;  // tiny images: too many images per tile (the gather GEMM serves them)
  const SPlan pl = plan_sconv(d.N, d.C, d.K, d.Ho, d.Wo, d.sh, 4, (d.Wo - 1) * d.sw + 4);
  return pl.ok && pl.tiles >= kMinTiles;
}
// data gradient: M = C, reduction over K
bool sconv_dgrad_supported(const ConvDims& d) {
  if (!sconv_layer_ok(d) || d.K % 2 || d.K < 16 || d.C % 64) return false;
  if (d.sh == 1) {
    if ((long)d.H * d.W < 96) return false;
    const SPlan pl = plan_sconv(d.N, d.K, d.C, d.H, d.W, 1, 4, d.W + 3);
    return pl.ok && pl.tiles >= kMinTiles;
  }
  const int Hu = (d.H + 1) / 2, Wu = (d.W + 1) / 2;
  if ((long)(d.H / 2) * (d.W / 2) < 96) return false;
  const SPlan pl = plan_sconv(d.N, d.K, d.C, Hu, Wu, 1, 2, Wu + 1, 4);
  return pl.ok && pl.tiles >= kMinTiles;
}
size_t sconv_ws_bytes(const ConvDims& d) {
  return align256((size_t)d.C * d.K * 16 * sizeof(float)) + 512;
}

int conv_fwd_sconv(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!ws || wsb < sconv_ws_bytes(d)) { set_error("sconv_fwd: workspace too small"); return NC_ERR_WS; }
  float* wp = (float*)ws;
  float* zeros = (float*)((char*)ws + align256((size_t)d.C * d.K * 16 * sizeof(float)));
  if (hipMemsetAsync(zeros, 0, 256, s) != hipSuccess) { set_error("sconv_fwd: memset failed"); return NC_ERR_HIP; }
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
  const SPlan pl = plan_sconv(d.N, d.C, d.K, d.Ho, d.Wo, d.sh, 4, c.Wp);
  if (!pl.ok) { set_error("sconv_fwd: no plan"); return NC_ERR_SHAPE; }
  p.CK = pl.CK; p.CS = pl.CS;
  return launch_sconv<4, 4, 1>(pl, p, c.ncol, 1, s);
}

int conv_dgrad_sconv(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!ws || wsb < sconv_ws_bytes(d)) { set_error("sconv_dgrad: workspace too small"); return NC_ERR_WS; }
  float* wp = (float*)ws;
  float* zeros = (float*)((char*)ws + align256((size_t)d.C * d.K * 16 * sizeof(float)));
  if (hipMemsetAsync(zeros, 0, 256, s) != hipSuccess) { set_error("sconv_dgrad: memset failed"); return NC_ERR_HIP; }
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
    const SPlan pl = plan_sconv(d.N, d.K, d.C, d.H, d.W, 1, 4, c.Wp);
    if (!pl.ok) { set_error("sconv_dgrad: no plan"); return NC_ERR_SHAPE; }
    p.CK = pl.CK; p.CS = pl.CS;
    return launch_sconv<4, 4, -1>(pl, p, c.ncol, 1, s);
  }
  // stride 2: input pixel iy = 2u + py receives the taps ky = t0y + 2 jy (t0y = (py + 1) & 1) from dy[(iy + 1 - ky) / 2]
  // = dy[u + ay - jy], ay = (py + 1 - t0y) / 2: per parity class a 2 x 2 problem walked backwards; all four in one launch
  p.so = 2; p.si = 1; p.rspan = 2;
  const int HuM = (d.H + 1) / 2, WuM = (d.W + 1) / 2;
  const SPlan pl = plan_sconv(d.N, d.K, d.C, HuM, WuM, 1, 2, WuM + 1, 4);
  if (!pl.ok) { set_error("sconv_dgrad: no plan"); return NC_ERR_SHAPE; }
  p.CK = pl.CK; p.CS = pl.CS;
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
  return launch_sconv<2, 2, -1>(pl, p, max_ncol, 4, s);
}

}  // namespace nc
