// Implicit-GEMM Conv3d (odd cubic kernel 3^3 / 5^3, stride 1, "same" padding) on the fp32 matrix cores of gfx950.
// Serves forward AND dgrad (dgrad = the same convolution of dy with the flipped, channel-transposed kernel).
// Replaces nn.Conv3d forward/backward-data at models/networks.py:420-425,442,460-469 (U-Net) and :900-902 (G_B).
//
// GEMM view:  Y[co][v] = sum_{ci,tap} Wp[(ci,tap)][co] * X[ci][v + tap]      (M = co, N = voxels, K = Cin * taps)
//   * v_mfma_f32_32x32x2_f32: exact fp32 (fmaf chain), 64 cycles per instruction per SIMD -- the same peak as the
//     fp32 VALU (157 TFLOP/s) but with ONE instruction per 4096 FLOP, so staging/address work hides behind it.
//   * A operand (weights) : pre-packed [K-row][co-pair-interleaved] in HBM (a few 100 KB, L1/L2 resident); each lane
//     loads one float2 per k-step = its two 32-wide co blocks.  No LDS for weights.
//   * B operand (inputs)  : an input brick of CK channels x (Tz+2p) planes x (Ty+2p) rows x (W+2p) columns in LDS, rows
//     FLATTENED with pitch P = W+2p.  A tap (dz,dy,dx) is then a constant LDS offset dz*RW + dy*P + dx, and one MFMA
//     column block is 32 consecutive floats -> ds_read_b32, conflict-free for any alignment.  Output positions that
//     fall on the 2p pad columns are computed and discarded (<= 2p/P waste) -- this is what makes 108/140/54/27-wide
//     volumes tile without 32-alignment.
//   * staging: global -> registers (issued BEFORE the MFMA loop of the current chunk) -> LDS (after it): the T14
//     split; per-element global offsets are decoded once per tile (magic-number division) and kept in registers.
//     Zero padding and (optionally) InstanceNorm+ReLU of the producer are applied while staging.
//   * epilogue: accumulator rows are co, lanes are voxels -> each store instruction writes 2 x 128 B contiguous.
#include "common.hpp"

namespace nc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int kRMAX = 18;  // brick rows staged per wave per chunk (x 3 column segments of 64)
static constexpr bool kFuseNorm = false;  // normalize-on-load is wired but not enabled in round 1  // staging registers per lane (elements of the brick per lane per chunk)

struct FwdParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  const float* mean;  // optional fused normalize-on-load of the input: act((x - mean[c]) * rstd[c])
  const float* rstd;
  float slope;
  int C, K, D, H, W;
  int Tz, Ty, nty;
  int P, RW, planes, CP;  // row pitch, floats per plane (= (Ty+2p)*P), planes per channel, floats per channel
  int nelem;              // CK * CP
  unsigned mP;            // magic multiplier: n / P == __umulhi(n, mP)
  int nrows, rowsY, PRows;  // brick rows per chunk (CK*planes*rowsY), rows per plane (Ty+2p), rows per channel
  unsigned mRowsY, mPR;
  int nchunks;
};

__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned m) { return __umulhi(n, m); }

template <int KS, int CK, int WM, int WN, int VB>
__global__ __launch_bounds__(WM* WN * 64) void k_conv_mfma(FwdParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x;
  const int z0 = (tile / p.nty) * p.Tz, y0 = (tile % p.nty) * p.Ty;
  const int cot = blockIdx.y, n = blockIdx.z;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;

  // ---- staging by ROWS: the brick is CK x planes x rowsY rows of W floats; a wave copies whole rows, lane = column,
  //      so the (channel, plane, row) decode and the global row base are wave-uniform (scalar unit) and no per-lane
  //      index arithmetic is left.  Rows outside the volume are written as zeros; the 2p pad columns of every LDS row
  //      are zeroed once and never written again.
  constexpr int NWV = NT / 64;
  const float* xn = p.x + (long)n * p.C * S;
  float st[kRMAX][3];
  const bool cm0 = lane < p.W, cm1 = lane + 64 < p.W, cm2 = lane + 128 < p.W;
  auto stage_load = [&](int chunk) {
    const float* xc = xn + (long)chunk * CK * S + lane;
#pragma unroll
    for (int i = 0; i < kRMAX; ++i) {
      const unsigned row = wave + NWV * i;
      const unsigned cic = fastdiv(row, p.mPR);
      const unsigned r1 = row - cic * p.PRows;
      const unsigned pz = fastdiv(r1, p.mRowsY);
      const unsigned yy = r1 - pz * p.rowsY;
      const int z = z0 + (int)pz - PAD, y = y0 + (int)yy - PAD;
      const bool rok = (int)row < p.nrows && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H;
      const float* r = xc + (long)cic * S + (long)z * HW + (long)y * p.W;
      float v0 = (rok && cm0) ? r[0] : 0.f;
      float v1 = (rok && cm1) ? r[64] : 0.f;
      float v2 = (rok && cm2) ? r[128] : 0.f;
      if (kFuseNorm && p.mean) {  // wave-uniform: fused InstanceNorm + activation of the producer layer
        const int c = chunk * CK + min((int)cic, CK - 1);
        const float m = p.mean[(long)n * p.C + c], rs = p.rstd[(long)n * p.C + c];
        v0 = (v0 - m) * rs; v1 = (v1 - m) * rs; v2 = (v2 - m) * rs;
        v0 = v0 > 0.f ? v0 : v0 * p.slope; v1 = v1 > 0.f ? v1 : v1 * p.slope; v2 = v2 > 0.f ? v2 : v2 * p.slope;
        v0 = rok ? v0 : 0.f; v1 = rok ? v1 : 0.f; v2 = rok ? v2 : 0.f;
      }
      st[i][0] = v0; st[i][1] = v1; st[i][2] = v2;
    }
  };
  auto stage_store = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < kRMAX; ++i) {
      const unsigned row = wave + NWV * i;
      if ((int)row < p.nrows) {
        const unsigned cic = fastdiv(row, p.mPR);
        const unsigned r1 = row - cic * p.PRows;
        const unsigned pz = fastdiv(r1, p.mRowsY);
        const unsigned yy = r1 - pz * p.rowsY;
        float* r = buf + cic * p.CP + pz * p.RW + yy * p.P + PAD + lane;
        if (cm0) r[0] = st[i][0];
        if (cm1) r[64] = st[i][1];
        if (cm2) r[128] = st[i][2];
      }
    }
  };
  for (int i = tid; i < 2 * p.nelem; i += NT) lds[i] = 0.f;  // pad columns (and everything else) start at zero
  __syncthreads();

  // ---- this wave's output sub-tile: VB column blocks of 32 flattened positions in plane tzl, 64 output channels
  const int GP = WM / p.Tz;
  const int tzl = wm / GP, blk0 = (wm % GP) * VB;
  const int b_base = h * p.CP + tzl * p.RW + blk0 * 32 + li;
  const int cob = (cot * WN + wn) * 64;  // first output channel of this wave
  const float* wlane = p.wp + cob + li * 2 + (long)h * p.K;

  f32x16 acc[2][VB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int v = 0; v < VB; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;

  float* buf0 = lds;
  float* buf1 = lds + p.nelem;

  stage_load(0);
  stage_store(buf0);
  __syncthreads();

  // The packed weights are ONE linear stream over the whole kernel: k-step s reads rows 2s, 2s+1 (lane half h picks
  // the row), chunk after chunk, tap after tap.  They are software-pipelined by hand one (dz, dy) row of the kernel
  // ahead -- U = KS * CK/2 k-steps, 3-6 k cycles of MFMA -- so no MFMA ever waits for its L2 round trip.  (The last
  // prefetch runs past the end of the stream into the workspace slack the host reserves.)
  constexpr int U = KS * (CK / 2);
  const long kstep = 2L * p.K;  // floats per k-step
  const float* aptr = wlane;
  float2 a_cur[U], a_nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a_cur[u] = *reinterpret_cast<const float2*>(aptr + u * kstep);
  aptr += U * kstep;

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const float* cur = (chunk & 1) ? buf1 : buf0;
    float* nxt = (chunk & 1) ? buf0 : buf1;
    const bool more = chunk + 1 < p.nchunks;
    if (more) stage_load(chunk + 1);
#pragma unroll 1
    for (int dz = 0; dz < KS; ++dz) {
#pragma unroll 1
      for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
        for (int u = 0; u < U; ++u) a_nxt[u] = *reinterpret_cast<const float2*>(aptr + u * kstep);
        aptr += U * kstep;
        const float* brow = cur + b_base + dz * p.RW + dy * p.P;
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
          for (int cp = 0; cp < CK / 2; ++cp) {
            const float2 a = a_cur[dx * (CK / 2) + cp];
            float b[VB];
#pragma unroll
            for (int v = 0; v < VB; ++v) b[v] = brow[dx + 2 * cp * p.CP + v * 32];
#pragma unroll
            for (int v = 0; v < VB; ++v) {
              acc[0][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[v], acc[0][v], 0, 0, 0);
              acc[1][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[v], acc[1][v], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a_cur[u] = a_nxt[u];
      }
    }
    if (more) stage_store(nxt);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: column (voxel) = lane & 31, row (co) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int z = z0 + tzl;
  if (z < p.D) {
    float* yn = p.y + ((long)n * p.K + cob) * S + (long)z * HW;
#pragma unroll
    for (int v = 0; v < VB; ++v) {
      const unsigned q = (blk0 + v) * 32 + li;
      const unsigned ty = fastdiv(q, p.mP);
      const unsigned x = q - ty * p.P;
      const int y = y0 + (int)ty;
      if ((int)ty < p.Ty && y < p.H && (int)x < p.W) {
        float* yv = yn + (long)y * p.W + x;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float val = acc[a][v][r];
            if (p.bias) val += p.bias[cob + co];
            yv[(long)co * S] = val;
          }
      }
    }
  }
}

// Weight packing.  Row = chunk*TAPS*CK + tap*CK + cic  (ci = chunk*CK + cic); inside a row: [K/64][32][2] with
// (blk, i, s) <-> co = blk*64 + s*32 + i.   mode 0: fwd  src[co][ci][tap];   mode 1: dgrad  src[ci][co][TAPS-1-tap].
__global__ void k_pack_w(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int taps, int CK,
                         int mode) {
  const long total = (long)Cin * Cout * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % Cout);
    const long row = i / Cout;
    const int cic = (int)(row % CK), tap = (int)((row / CK) % taps), chunk = (int)(row / ((long)CK * taps));
    const int ci = chunk * CK + cic;
    const int blk = col / 64, ii = (col % 64) / 2, s = col & 1;
    const int co = blk * 64 + s * 32 + ii;
    float v;
    if (mode == 0) v = w[((long)co * Cin + ci) * taps + tap];
    else v = w[((long)ci * Cout + co) * taps + (taps - 1 - tap)];
    wp[i] = v;
  }
}

struct FwdPlan {
  int cfg, CK, Tz, Ty, nty, ntz, P, RW, planes, CP, nelem, lds_bytes;
  long cost;
};

struct Cfg { int WM, WN, VB; };
static const Cfg kCfgs[] = {{8, 1, 4}, {8, 1, 2}, {4, 2, 2}, {2, 4, 2}, {2, 4, 1}};
static constexpr int kNumCfgs = 5;
static constexpr int kLdsMax = 160 * 1024;

static unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

// Cin = reduction channels, Cout = produced channels (for dgrad the roles of C and K are swapped by the caller).
static bool plan_fwd(int KS, int Cin, int Cout, int N, int D, int H, int W, FwdPlan& best) {
  const int pad = KS / 2;
  const int P = W + 2 * pad;
  bool found = false;
  for (int c = 0; c < kNumCfgs; ++c) {
    const Cfg& g = kCfgs[c];
    if (Cout % (g.WN * 64)) continue;
    const int NT = g.WM * g.WN * 64;
    for (int Tz = 1; Tz <= g.WM; Tz *= 2) {
      if (Tz > D && Tz > 1) continue;
      const int GP = g.WM / Tz;
      const int maxpos = GP * g.VB * 32;
      int TyMax = maxpos / P;
      if (TyMax < 1) continue;
      if (TyMax > H) TyMax = H;
      const int nty0 = (H + TyMax - 1) / TyMax;
      const int Ty = (H + nty0 - 1) / nty0;  // balanced rows per tile
      const int nty = (H + Ty - 1) / Ty, ntz = (D + Tz - 1) / Tz;
      const int planes = Tz + 2 * pad;
      const int RW = (Ty + 2 * pad) * P;
      const int CP = planes * RW;
      const int ckList[3] = {8, 4, 2};
      for (int k = 0; k < 3; ++k) {
        const int CK = ckList[k];
        if (KS == 3 && CK == 2) continue;  // instantiated: KS=3 -> {8,4}; KS=5 -> {4,2}
        if (KS == 5 && CK == 8) continue;
        if (Cin % CK) continue;
        const long nelem = (long)CK * CP;
        // slack: garbage columns of the last blocks read up to maxpos + (KS-1)*(P+1) past the last plane's start
        const long slack = maxpos + (long)(KS - 1) * (P + 1) + 64;
        const long bytes = (2 * nelem + slack) * 4;
        const int nrows = CK * planes * (Ty + 2 * pad);
        if (bytes > kLdsMax || (nrows + NT / 64 - 1) / (NT / 64) > kRMAX) continue;
        const long tiles = (long)ntz * nty * (Cout / (g.WN * 64)) * N;
        const long rounds = (tiles + 255) / 256;
        // time per tile ~ VB MFMA pairs per k-step; small penalty for more barriers with small CK
        const long cost = rounds * g.VB * 1000 + (8 / CK) * 5 * rounds;
        if (!found || cost < best.cost) {
          found = true;
          best = FwdPlan{c, CK, Tz, Ty, nty, ntz, P, RW, planes, CP, (int)nelem, (int)bytes, cost};
        }
        break;  // largest CK that fits for this (cfg, Tz)
      }
    }
  }
  return found;
}

template <int KS, int CK, int WM, int WN, int VB>
static int launch_one(const FwdParams& p, dim3 grid, int lds_bytes, hipStream_t s) {
  auto kern = k_conv_mfma<KS, CK, WM, WN, VB>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLdsMax) != hipSuccess) {
      set_error("conv_mfma: cannot raise dynamic LDS limit");
      return NC_ERR_HIP;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds_bytes, s, p);
  return check_launch("conv_mfma");
}

template <int KS, int CK>
static int launch_cfg(int cfg, const FwdParams& p, dim3 grid, int lds, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_one<KS, CK, 8, 1, 4>(p, grid, lds, s);
    case 1: return launch_one<KS, CK, 8, 1, 2>(p, grid, lds, s);
    case 2: return launch_one<KS, CK, 4, 2, 2>(p, grid, lds, s);
    case 3: return launch_one<KS, CK, 2, 4, 2>(p, grid, lds, s);
    default: return launch_one<KS, CK, 2, 4, 1>(p, grid, lds, s);
  }
}

static bool shape_ok(const ConvDims& d, int Cin, int Cout) {
  if (d.kd != d.kh || d.kh != d.kw) return false;
  if (d.kd != 3 && d.kd != 5) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.pd != d.kd / 2 || d.ph != d.kd / 2 || d.pw != d.kd / 2) return false;
  if (Cin % 4 != 0 || Cout % 64 != 0) return false;
  if (d.kd == 5 && Cin % 2 != 0) return false;
  if ((long)Cin * d.D * d.H * d.W * 4 >= (1L << 31)) return false;  // int32 byte offsets in the staging loads
  if (d.W > 192) return false;  // rows are staged as up to three 64-column segments
  return true;
}

bool mfma_fwd_supported(const ConvDims& d) {
  FwdPlan pl;
  return shape_ok(d, d.C, d.K) && plan_fwd(d.kd, d.C, d.K, d.N, d.D, d.H, d.W, pl);
}
bool mfma_dgrad_supported(const ConvDims& d) {
  FwdPlan pl;
  return shape_ok(d, d.K, d.C) && plan_fwd(d.kd, d.K, d.C, d.N, d.D, d.H, d.W, pl);
}

static int run(const float* x, const float* w, const float* bias, float* y, int Cin, int Cout, const ConvDims& d,
               int mode, void* ws, size_t wsb, hipStream_t s) {
  FwdPlan pl;
  if (!plan_fwd(d.kd, Cin, Cout, d.N, d.D, d.H, d.W, pl)) {
    set_error("conv_mfma: no tile plan for this shape");
    return NC_ERR_SHAPE;
  }
  const int taps = d.kd * d.kh * d.kw;
  const size_t need = (size_t)Cin * Cout * taps * sizeof(float) + kPackSlackBytes;
  if (!ws || wsb < need) {
    set_error("conv_mfma: workspace too small (%zu < %zu)", wsb, need);
    return NC_ERR_WS;
  }
  float* wp = (float*)ws;
  hipLaunchKernelGGL(k_pack_w, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, taps, pl.CK, mode);
  if (int e = check_launch("pack_w")) return e;
  FwdParams p{};
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.mean = nullptr; p.rstd = nullptr; p.slope = 0.f;
  p.C = Cin; p.K = Cout; p.D = d.D; p.H = d.H; p.W = d.W;
  p.Tz = pl.Tz; p.Ty = pl.Ty; p.nty = pl.nty; p.P = pl.P; p.RW = pl.RW; p.planes = pl.planes; p.CP = pl.CP;
  p.nelem = pl.nelem; p.mP = magic(pl.P);
  p.rowsY = pl.Ty + 2 * (d.kd / 2); p.PRows = pl.planes * p.rowsY; p.nrows = pl.CK * p.PRows;
  p.mRowsY = magic(p.rowsY); p.mPR = magic(p.PRows);
  p.nchunks = Cin / pl.CK;
  const Cfg& g = kCfgs[pl.cfg];
  dim3 grid(pl.ntz * pl.nty, Cout / (g.WN * 64), d.N);
  if (d.kd == 3) {
    if (pl.CK == 8) return launch_cfg<3, 8>(pl.cfg, p, grid, pl.lds_bytes, s);
    return launch_cfg<3, 4>(pl.cfg, p, grid, pl.lds_bytes, s);
  }
  if (pl.CK == 4) return launch_cfg<5, 4>(pl.cfg, p, grid, pl.lds_bytes, s);
  return launch_cfg<5, 2>(pl.cfg, p, grid, pl.lds_bytes, s);
}

int conv_fwd_mfma(const float* x, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                  hipStream_t s) {
  return run(x, w, b, y, d.C, d.K, d, 0, ws, wsb, s);
}
int conv_dgrad_mfma(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s) {
  return run(dy, w, nullptr, dx, d.K, d.C, d, 1, ws, wsb, s);
}

}  // namespace nc
