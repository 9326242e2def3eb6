// Weight gradient of the PatchGAN's Conv2d(256 -> 512, k 4, s 1, p 1) (models/networks.py:1049-1055: the layer that carries 62 % of a
// discriminator pass' weight-gradient FLOPs) at Athena's batches (108-216 planes of 13 x 13), on the bf16 matrix cores with the exact
// three-term operand split: k_wgrad_s3x of conv_split.hip carried over to a flat batch of small planes.
//
//   dw[k][c][ky][kx] = sum over (b, y, x) of dy[b][k][y][x] * xpad[b][c][y + ky][x + kx]        (xpad = x with a zero border of 1)
//
// is a GEMM whose reduction runs over positions.  A workgroup owns (64 k) x (32 c) x 16 taps and a share of the position axis, walks it in
// steps of ONE OUTPUT ROW of Tb planes (PT = Tb * Wo positions = NK k-steps of 32), and keeps in LDS
//   * a ring of five padded input rows (row r of the Tb planes: Tb * Wp units per sub-block; rows y .. y + 3 feed output row y -- the four
//     kernel rows are four ring slots, the four kernel columns are unit offsets inside a slot -- and row y + 4 arrives meanwhile),
//   * two buffers of dy rows,
// both in the S3 form (8-channel units, three bf16 terms as sub-blocks) and both filled by LDS-DMA through buffer descriptors: a lane's
// source offset depends on (plane in tile, column) only and is computed once per kernel; tile and row are the scalar offset.  Out-of-tile
// lanes ask beyond the descriptor and get zeros; planes beyond the batch exist in the converted tensors as zeros.  The multiply is
// k_wgrad_s3x's: transposing LDS reads (ds_read_b64_tr_b16) put 8 positions per lane group on the K-dim of v_mfma_f32_16x16x32_bf16, a
// wave owns 4 units (tap, 16 channels) x all 64 k, six products per fp32 product, partial sums restarted every F steps and summed in a
// fixed order by k_wgrad_p2d_reduce (deterministic).  The stride-2 layers stay on k_swgrad: their input rows are four times the output
// row and a ring of them does not fit next to a 64 x 32 tile.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "s3_common.hpp"

namespace nc {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) s16x4* ltr_t;

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kLdsMax = 160 * 1024;
constexpr int kWP = 6;     // 1 KiB pieces per wave of an X slot / a dY buffer
constexpr int kSlots = 5;  // padded input rows in LDS
constexpr int kNU = 4;     // units (tap, 16-channel block) per wave: 16 taps x 2 blocks / 8 waves

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m) { return __umulhi(n, m); }
unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct QParams {
  const uint4* xs;   // S3 of the padded input  [C/8][3][TOTx], unit (b * PPx + yp * Wp + xp); planes b >= B are zeros
  const uint4* dys;  // S3 of dy                [K/8][3][TOTy], unit (b * HoWo + y * Wo + x);  planes b >= B are zeros
  float* part;       // [pairs * nwp][NF][16 taps][64 k][32 c]
  long TOTx, TOTy;
  int Ho, Wo, Wp, PPx, HoWo;
  int Tb;            // planes per tile
  int F, NF;         // steps between two accumulator restarts, partial slots per workgroup
  int XU, XUp;       // X row slot: units per sub-block Tb * Wp, padded sub-block stride (== 4 mod 8)
  int PT, PTp, NK;   // dY row: positions Tb * Wo, padded sub-block stride, k-steps of 32
  int npx, npd;      // 1 KiB pieces per X slot / per dY buffer
  int xslot, dybuf;  // bytes
  int nct;           // C / 32
  int npairs, nwp;   // (k-tile, c-tile) pairs, workgroups per pair
  long steps;        // tiles * Ho (per pair)
  unsigned mWp, mWo, mXUp, mPTp;
};

__device__ __forceinline__ i32x4 tr_frag(const unsigned char* lds, unsigned a0, unsigned a1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + a1));
  return __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ f32x4 mfma16(const i32x4& a, const i32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 dma_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = __builtin_amdgcn_readfirstlane(0x00020000u);
  return r;
}
// (inline assembly: with the builtin the compiler orders every LDS read behind the DMA in flight -- conv_split.hip, k_wgrad_s3x)
__device__ __forceinline__ void dma16(const u32x4& rs, const unsigned char* lds_dst, unsigned voff, int soff) {
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

// fp32 NCHW [B][C][H][W] -> S3 of the flat batch with a zero border of `pad`: unit ((chunk * 3 + term) * TOT + i), i = b * PP + yp * Wp + xp,
// zeros for i >= B * PP.  One thread per (chunk, unit).
__global__ void __launch_bounds__(256) k_split3_flat(const float* __restrict__ x, uint4* __restrict__ xs, int B, int C, int H, int W, int pad, int Hp,
                                                     int Wp, long TOT) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= TOT) return;
  const int chunk = blockIdx.y;
  const int PP = Hp * Wp;
  const int b = (int)(i / PP), r = (int)(i - (long)b * PP);
  const int yp = r / Wp, xp = r - yp * Wp;
  const int yy = yp - pad, xx = xp - pad;
  const bool in = b < B && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
  const float* src = x + (((long)(in ? b : 0) * C + chunk * 8) * H + (in ? yy : 0)) * W + (in ? xx : 0);
  unsigned short e[8][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = in ? src[(long)j * H * W] : 0.f;
    s3_split(v, e[j]);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) xs[((long)chunk * 3 + t) * TOT + i] = s3_unit(e, t);
}

__global__ void __launch_bounds__(kThreads, 1) k_wgrad_p2d(const QParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  constexpr int NT = 3, NU = kNU;
  constexpr unsigned kOut = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int G = gridDim.x, xcd = blockIdx.x & 7;
  const int wg = (G >> 3) * xcd + ((G & 7) < xcd ? (G & 7) : xcd) + (blockIdx.x >> 3);
  const int pair = wg % p.npairs, wi = wg / p.npairs;
  if (wi >= p.nwp) return;
  const int kt = pair / p.nct, ct = pair % p.nct;
  const long s_lo = p.steps * wi / p.nwp, s_hi = p.steps * (wi + 1) / p.nwp;

  unsigned char* const xring = lds_raw;
  unsigned char* const dyb = lds_raw + kSlots * p.xslot;
  auto slot_of = [&](int r) __attribute__((always_inline)) { return (r % kSlots) * p.xslot; };

  // source offsets of this lane in the pieces wave + 8 i (tile and row are scalar offsets); kOut = padding of the LDS image
  unsigned xo[kWP], yo[kWP];
#pragma unroll
  for (int i = 0; i < kWP; ++i) {
    const int pc = wave + kWaves * i;
    const unsigned u = (unsigned)(pc * 64 + lane);
    {
      const unsigned sb = fdiv(u, p.mXUp);  // sub-block = 8-channel block * 3 + term
      const unsigned ur = u - sb * p.XUp;
      const unsigned bi = fdiv(ur, p.mWp);
      const bool ok = pc < p.npx && sb < 4u * NT && ur < (unsigned)p.XU;
      xo[i] = ok ? (unsigned)((sb * p.TOTx + bi * p.PPx + (ur - bi * p.Wp)) * 16) : kOut;
    }
    {
      const unsigned sb = fdiv(u, p.mPTp);
      const unsigned rho = u - sb * p.PTp;
      const unsigned bi = fdiv(rho, p.mWo);
      const bool ok = pc < p.npd && sb < 8u * NT && rho < (unsigned)p.PT;
      yo[i] = ok ? (unsigned)((sb * p.TOTy + bi * p.HoWo + (rho - bi * p.Wo)) * 16) : kOut;
    }
  }
  const u32x4 rsx = dma_rsrc(p.xs + (long)ct * 4 * NT * p.TOTx, (unsigned)(4 * NT * p.TOTx * 16));
  const u32x4 rsy = dma_rsrc(p.dys + (long)kt * 8 * NT * p.TOTy, (unsigned)(8 * NT * p.TOTy * 16));
  auto issue_x = [&](int bt, int r, unsigned char* slot) __attribute__((always_inline)) {
    const int soff = __builtin_amdgcn_readfirstlane((bt * p.Tb * p.PPx + r * p.Wp) * 16);
#pragma unroll
    for (int i = 0; i < kWP; ++i)
      if (wave + kWaves * i < p.npx) dma16(rsx, slot + (wave + kWaves * i) * 1024, xo[i], soff);
  };
  auto issue_dy = [&](int bt, int y, unsigned char* buf) __attribute__((always_inline)) {
    const int soff = __builtin_amdgcn_readfirstlane((bt * p.Tb * p.HoWo + y * p.Wo) * 16);
#pragma unroll
    for (int i = 0; i < kWP; ++i)
      if (wave + kWaves * i < p.npd) dma16(rsy, buf + (wave + kWaves * i) * 1024, yo[i], soff);
  };

  // transposed-read roles: lane = 16g + 4q + pp: position row 8g + 4*s2 + q of the k-step, channels 4pp .. 4pp + 3 of a 16-channel block
  // (= 8-channel sub-blocks 2*blk + (pp >> 1), byte (pp & 1) * 8 of the unit)
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const unsigned a_term = (unsigned)p.PTp * 16, b_term = (unsigned)p.XUp * 16;
  const unsigned a_lane = (unsigned)((((pp >> 1) * NT) * p.PTp) * 16 + (pp & 1) * 8), a_blk = 2u * NT * a_term;  // + a * a_blk: k rows 16a ..
  const unsigned b_lane = (unsigned)((((pp >> 1) * NT) * p.XUp) * 16 + (pp & 1) * 8), b_blk = 2u * NT * b_term;  // + b * b_blk: c 16b ..

  // two accumulators per output: the leading product a0*b0 and the five low-order products.  Every MFMA rounds its accumulator once; with
  // one accumulator the small products cost five more roundings of the LARGE running sum per k-step (error against fp64 2e-6 of the largest
  // element at 216 planes, above the fp32 kernel's 1.6e-6); kept apart, the low-order sum is 2^-8 of the other and its roundings vanish.
  f32x4 acc[NU][4], low[NU][4];
#pragma unroll
  for (int j = 0; j < NU; ++j)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[j][a][e] = 0.f; low[j][a][e] = 0.f; }

  int uky[NU];
  unsigned uoff[NU];  // unit j of this wave = (tap, c-block): kernel row (= ring slot) and byte offset inside a slot (column + c-block + lane)
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const int u = wave * NU + j;
    const int t = u >> 1, b = u & 1;
    uky[j] = t >> 2;
    uoff[j] = (unsigned)((t & 3) * 16) + b * b_blk + b_lane;
  }

  // partial slot f: part[wg][f][tap][k 0..63][c 0..31]; accumulator element e of (unit (tap, b), a) = k a*16 + 4g + e, c b*16 + lane%16
  int nflush = 0, since = 0;
  auto write_partial = [&](bool live) __attribute__((always_inline)) {
    float* pw = p.part + ((long)wg * p.NF + nflush) * 16 * 64 * 32;
    const int m16 = lane & 15;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int u = wave * NU + j;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float* pt = pw + ((long)(u >> 1) * 64 + a * 16 + 4 * g) * 32 + (u & 1) * 16 + m16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pt[e * 32] = live ? acc[j][a][e] + low[j][a][e] : 0.f;
          acc[j][a][e] = 0.f; low[j][a][e] = 0.f;
        }
      }
    }
  };

  long step = s_lo;
  bool fresh = true;
  int bt = 0, y = 0;
  while (step < s_hi) {
    if (fresh) {
      bt = (int)(step / p.Ho);
      y = (int)(step - (long)bt * p.Ho);
      __syncthreads();
#pragma unroll
      for (int l = 0; l < 4; ++l) issue_x(bt, y + l, xring + slot_of(y + l));
      issue_dy(bt, y, dyb + (y & 1) * p.dybuf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      fresh = false;
    }
    const bool cont = y + 1 < p.Ho && step + 1 < s_hi;
    if (cont) {
      issue_x(bt, y + 4, xring + slot_of(y + 4));
      issue_dy(bt, y + 1, dyb + ((y + 1) & 1) * p.dybuf);
    }
    if (since == p.F && nflush + 1 < p.NF) {
      write_partial(true);
      ++nflush;
      since = 0;
    }
    ++since;
    // ---- multiply: NK k-steps of 32 positions x 4 units x 4 row blocks x 6 term products
    const unsigned abase = (unsigned)(kSlots * p.xslot + (y & 1) * p.dybuf) + a_lane;
    unsigned sb[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) sb[j] = (unsigned)slot_of(y + uky[j]) + uoff[j];
#pragma unroll 1
    for (int s = 0; s < p.NK; ++s) {
      unsigned rho[2], bo[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        rho[s2] = (unsigned)(32 * s + 8 * g + 4 * s2 + q);
        const unsigned rc = rho[s2] < (unsigned)p.PT ? rho[s2] : (unsigned)p.PT - 1;  // padded positions: dY = 0, X any finite
        const unsigned bi = fdiv(rc, p.mWo);
        bo[s2] = (bi * p.Wp + (rc - bi * p.Wo)) * 16;
      }
      // One k-step: 12 A + 3 B fragments up front in the order the products need them, then per unit the reads of the next unit's B
      // fragments spread over its MFMAs (order pinned with sched_group_barrier, as in k_wgrad_s3x).
      constexpr int TA[6] = {2, 1, 0, 1, 0, 0};  // products per (row block, unit), smallest first: (term of A, term of B)
      constexpr int TB[6] = {0, 1, 2, 0, 1, 0};
      i32x4 A[4][NT], B[2][NT];
      auto read_b1 = [&](i32x4& Bf, int j, int t) __attribute__((always_inline)) {
        Bf = tr_frag(lds_raw, sb[j] + t * b_term + bo[0], sb[j] + t * b_term + bo[1]);
      };
#pragma unroll
      for (int t = NT - 1; t >= 0; --t) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          A[a][t] = tr_frag(lds_raw, abase + a * a_blk + t * a_term + rho[0] * 16, abase + a * a_blk + t * a_term + rho[1] * 16);
        read_b1(B[0][NT - 1 - t], 0, NT - 1 - t);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 10 * NT, 0);
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        if (j + 1 < NU) {
#pragma unroll
          for (int t = 0; t < NT; ++t) read_b1(B[(j + 1) & 1][t], j + 1, t);
        }
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            if (m < 5) low[j][a] = mfma16(A[a][TA[m]], B[j & 1][TB[m]], low[j][a]);
            else acc[j][a] = mfma16(A[a][TA[m]], B[j & 1][TB[m]], acc[j][a]);
          }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          if (j + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++step;
    if (cont) ++y; else fresh = true;
  }

  write_partial(nflush < p.NF);
  for (++nflush; nflush < p.NF; ++nflush) write_partial(false);  // slots this workgroup did not need: zeros
}

// dw[k][c][tap] = sum of the nwp * NF partial slots of pair (k/64, c/32), as k_wgrad_s3_reduce: a workgroup handles 32 consecutive c of one
// (k, tap); its 8 lane groups each add a contiguous eighth of the slots in slot order, the eight sums are added in group order.
__global__ void __launch_bounds__(256) k_wgrad_p2d_reduce(const float* __restrict__ part, float* __restrict__ dw, int C, int nct, int npairs, int nwp,
                                                          int NF, long total) {
  __shared__ float red[8][32];
  const int o = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + o;  // (k, tap, c): c fastest -> coalesced partial reads
  float sacc = 0.f;
  int c = 0, t = 0, k = 0;
  if (i < total) {
    c = (int)(i % C);
    t = (int)((i / C) % 16);
    k = (int)(i / ((long)C * 16));
    const int pair = (k / 64) * nct + c / 32;
    const long off = ((long)t * 64 + (k & 63)) * 32 + (c & 31);
    const int nslots = nwp * NF, per = (nslots + 7) / 8;
    const int s0 = seg * per, s1 = s0 + per < nslots ? s0 + per : nslots;
    for (int sl = s0; sl < s1; ++sl) {
      const int w = sl / NF, f = sl - w * NF;
      sacc += part[(((long)(w * npairs + pair) * NF + f) * 16) * 64 * 32 + off];
    }
  }
  red[seg][o] = sacc;
  __syncthreads();
  if (seg == 0 && i < total) {
    float r = red[0][o];
#pragma unroll
    for (int gq = 1; gq < 8; ++gq) r += red[gq][o];
    dw[((long)k * C + c) * 16 + t] = r;
  }
}

struct QPlan {
  int Tb, NBT, XU, XUp, PT, PTp, NK, npx, npd, xslot, dybuf, nwp, NF, F;
  long TOTx, TOTy, steps;
  bool ok;
};

int pad_4mod8(int v) { return v + ((4 - (v & 7)) & 7); }  // sub-block stride that keeps the transposed reads conflict-free

int q_flush_steps() {
  static const int f = 128;
  return f > 0 ? f : 1 << 30;
}

QPlan q_plan(const ConvDims& d) {
  QPlan best{};
  double best_cost = 1e30;
  const int Wp = d.W + 2, Hp = d.H + 2;
  const int npairs = (d.K / 64) * (d.C / 32);
  for (int Tb = 1; Tb <= 64 && Tb <= d.N; ++Tb) {
    QPlan pl{};
    pl.Tb = Tb; pl.NBT = (d.N + Tb - 1) / Tb;
    pl.XU = Tb * Wp; pl.XUp = pad_4mod8(pl.XU);
    pl.PT = Tb * d.Wo; pl.NK = (pl.PT + 31) / 32; pl.PTp = pad_4mod8(pl.NK * 32);
    pl.npx = (12 * pl.XUp + 63) / 64; pl.npd = (24 * pl.PTp + 63) / 64;
    pl.xslot = pl.npx * 1024; pl.dybuf = pl.npd * 1024;
    if (kSlots * pl.xslot + 2 * pl.dybuf > kLdsMax) continue;
    if (pl.NK < 2 || pl.npx > kWaves * kWP || pl.npd > kWaves * kWP) continue;
    // cost per useful position: MFMA time (k-steps incl. padding and the overhang of the last tile) + a staging term
    const double useful = (double)d.N * d.Wo;
    const double mfma = (double)pl.NBT * pl.NK * 32;
    const double stage = (double)pl.NBT * (4.0 * pl.XUp + 8.0 * pl.PTp) / 24.0;
    const double cost = (mfma + 0.15 * stage) / useful;
    if (cost < best_cost) { best_cost = cost; best = pl; best.ok = true; }
  }
  if (!best.ok) return best;
  best.TOTx = (long)best.NBT * best.Tb * Hp * Wp;
  best.TOTy = (long)best.NBT * best.Tb * d.Ho * d.Wo;
  best.steps = (long)best.NBT * d.Ho;
  int nwp = 256 / npairs;
  if (nwp > best.steps) nwp = (int)best.steps;
  if (nwp < 1) nwp = 1;
  best.nwp = nwp;
  best.F = q_flush_steps();
  const long most = (best.steps + nwp - 1) / nwp;
  best.NF = (int)((most + best.F - 1) / best.F);
  if (best.NF < 1) best.NF = 1;
  return best;
}

bool q_shape(const ConvDims& d) {
  static const int on = getenv("NC_P2D") ? atoi(getenv("NC_P2D")) : 7;  // bit 2 = this kernel (bits 0, 1: conv_p2d.hip)
  if (!(on & 4)) return false;
  if (d.D != 1 || d.kd != 1 || d.kh != 4 || d.kw != 4 || d.sh != 1 || d.sw != 1 || d.ph != 1 || d.pw != 1) return false;
  if (d.C % 32 || d.K % 64 || (d.K / 64) * (d.C / 32) > 256) return false;
  static const long minpos = 8192;
  if ((long)d.N * d.Ho * d.Wo < minpos) return false;  // small batches (Apollo's 1-4 planes per discriminator) stay where they are
  const QPlan pl = q_plan(d);
  if (!pl.ok) return false;
  return 12 * pl.TOTx * 16 < (1l << 31) && 24 * pl.TOTy * 16 < (1l << 31);
}

size_t q_xs_bytes(const ConvDims& d, const QPlan& pl) { return align256((size_t)(d.C / 8) * 3 * pl.TOTx * 16); }
size_t q_ys_bytes(const ConvDims& d, const QPlan& pl) { return align256((size_t)(d.K / 8) * 3 * pl.TOTy * 16); }
size_t q_part_bytes(const ConvDims& d, const QPlan& pl) {
  return align256((size_t)(d.K / 64) * (d.C / 32) * pl.nwp * pl.NF * 16 * 64 * 32 * 4);
}

}  // namespace

bool p2d_wgrad_supported(const ConvDims& d) { return q_shape(d); }

size_t p2d_wgrad_ws_bytes(const ConvDims& d) {
  if (!q_shape(d)) return 0;
  const QPlan pl = q_plan(d);
  return q_xs_bytes(d, pl) + q_ys_bytes(d, pl) + q_part_bytes(d, pl) + 512;
}

int conv_wgrad_p2d(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  const QPlan pl = q_plan(d);
  if (!pl.ok) { set_error("wgrad_p2d: shape not covered"); return NC_ERR_SHAPE; }
  const size_t xb = q_xs_bytes(d, pl), yb = q_ys_bytes(d, pl), pb = q_part_bytes(d, pl);
  if (!ws || wsb < xb + yb + pb + 256) { set_error("wgrad_p2d: workspace too small"); return NC_ERR_WS; }
  uint4* xs = (uint4*)ws;
  uint4* dys = (uint4*)((char*)ws + xb);
  float* part = (float*)((char*)ws + xb + yb);
  const int Hp = d.H + 2, Wp = d.W + 2;
  hipLaunchKernelGGL(k_split3_flat, dim3((unsigned)cdiv(pl.TOTx, 256), (unsigned)(d.C / 8)), dim3(256), 0, s, x, xs, d.N, d.C, d.H, d.W, 1, Hp, Wp, pl.TOTx);
  hipLaunchKernelGGL(k_split3_flat, dim3((unsigned)cdiv(pl.TOTy, 256), (unsigned)(d.K / 8)), dim3(256), 0, s, dy, dys, d.N, d.K, d.Ho, d.Wo, 0, d.Ho, d.Wo,
                     pl.TOTy);
  if (int e = check_launch("wgrad_p2d convert")) return e;
  QParams p{};
  p.xs = xs; p.dys = dys; p.part = part;
  p.TOTx = pl.TOTx; p.TOTy = pl.TOTy;
  p.Ho = d.Ho; p.Wo = d.Wo; p.Wp = Wp; p.PPx = Hp * Wp; p.HoWo = d.Ho * d.Wo;
  p.Tb = pl.Tb; p.F = pl.F; p.NF = pl.NF;
  p.XU = pl.XU; p.XUp = pl.XUp; p.PT = pl.PT; p.PTp = pl.PTp; p.NK = pl.NK;
  p.npx = pl.npx; p.npd = pl.npd; p.xslot = pl.xslot; p.dybuf = pl.dybuf;
  p.nct = d.C / 32; p.npairs = (d.K / 64) * (d.C / 32); p.nwp = pl.nwp; p.steps = pl.steps;
  p.mWp = magic(Wp); p.mWo = magic(d.Wo); p.mXUp = magic(pl.XUp); p.mPTp = magic(pl.PTp);
  if (int e = raise_dyn_lds(k_wgrad_p2d, kLdsMax, "wgrad_p2d")) return e;
  const int lds = kSlots * pl.xslot + 2 * pl.dybuf;
  hipLaunchKernelGGL(k_wgrad_p2d, dim3(p.npairs * pl.nwp), dim3(kThreads), lds, s, p);
  if (int e = check_launch("wgrad_p2d")) return e;
  const long total = (long)d.K * d.C * 16;
  hipLaunchKernelGGL(k_wgrad_p2d_reduce, dim3((unsigned)cdiv(total, 32)), dim3(256), 0, s, part, dw, d.C, d.C / 32, p.npairs, pl.nwp, pl.NF, total);
  return check_launch("wgrad_p2d_reduce");
}

}  // namespace nc
