// fp32 Conv3d 3^3 / 5^3 (stride 1, "same" padding) on the 16-bit matrix cores: forward, data gradient and weight gradient of
// the U-Net's 3^3 layers (models/networks.py:420-425, 460-469) and of G_B's 5^3 / 3^3 feature layers (:900-902) with fp32
// operands, fp32 accumulation and fp32 results.
//
// gfx950 multiplies bf16 sixteen times faster than fp32 (v_mfma_f32_32x32x16_bf16: 32768 FLOP in 32 cycles; the fp32
// instruction v_mfma_f32_32x32x2_f32: 4096 FLOP in 64).  An fp32 number is EXACTLY the sum of three bf16 numbers
//     a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (8 + 8 + 8 significand bits)
// and a bf16 x bf16 product is exact in the fp32 accumulator of the MFMA, so
//     a * b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + [a1 b2 + a2 b1 + a2 b2]
// where the bracket is below 2^-23 |a b| -- the size of ONE rounding of an fp32 product.  The six other products are six
// bf16 MFMAs on the same accumulator: the arithmetic of an fp32 convolution (exact operands, fp32 accumulation) at 16 / 6
// = 2.7 x the fp32 matrix rate.  tests/test_gpu_split.py measures the error against fp64 next to the fp32 MFMA kernel's.
//
// Layout "S3": [N][C/8][3 terms][D][H][W][8 channels] bf16 -- one voxel's 8 channels of one term are one 16-byte unit, the
// MFMA B fragment of a lane.  k_split3 writes it from fp32 NCDHW.  The kernel is the PAIR form of the 16-bit kernel
// (conv_h.hip): stage = (8-channel chunk, dz), a k-step = 8 channels at two taps (lane half h takes tap 2i + h; 5 k-steps
// for the 9 taps of a plane, the 10th tap has zero weights), brick and weights of a stage in LDS by LDS-DMA for all three
// terms (3 x the 16-bit bytes, 6 x the MFMAs: a third of the 16-bit kernel's staging traffic per MFMA), two stage buffers,
// tile = 64 output channels x 512 flattened positions of one output plane on 8 waves (2 x 2 accumulator tiles each).  5^3: the
// 13 tap pairs of a plane come in four sub-stages of weights over one staged brick.  k_wgrad_s3 (below) is the weight gradient.
// (Round 3: these two are the fallbacks -- NC_S3X=0 / NC_S3X_WGRAD=0, shapes the tap-stream kernel does not cover; the default forward
// and data gradient are conv_s3x.hip's k_conv_s3x, the default weight gradient is k_wgrad_s3x below, also on 16-bit operands.)
// Accuracy: the matrix core's rounding error grows with the running sum in its accumulator, so the accumulators restart per
// stage group (forward / dgrad) or every 64 steps (wgrad) and the pieces are added in fp32 -- with that the error against fp64
// is below the fp32 MFMA kernels' at every shape tested ("Split-operand fp32 convolutions").
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "s3_common.hpp"

namespace nc {
NC_ZERO_PAGE()
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kLdsMax = 160 * 1024;
// Taps of one kernel plane are taken two per k-step ("pair" q = taps 2q, 2q + 1 in row-major order; an odd count ends in a
// zero-weight tap).  The pairs of a plane are split into sub-stages so that two weight buffers and two brick buffers fit 160 KB:
// 3^3: 5 pairs, one sub-stage;  5^3: 13 pairs in sub-stages of 4 + 3 + 3 + 3 (the brick of the plane is staged once).
// NC_S3_ABLATE (timing experiments only, results are garbage when set): 1 no LDS fragment reads after the first of a sub-stage,
// 4 no brick / weight DMA after the first, 8 no flush, 16 no stage barrier
#ifndef NC_S3_ABLATE
#define NC_S3_ABLATE 0
#endif
template <int KS> struct Sub;
template <> struct Sub<3> { static constexpr int NP = 5, NSUB = 1, MAXP = 5; static constexpr int q0[2] = {0, 5}; };
template <> struct Sub<5> { static constexpr int NP = 13, NSUB = 4, MAXP = 4; static constexpr int q0[5] = {0, 4, 7, 10, 13}; };
constexpr int kPairPieces = 3 * 2;  // 1 KiB weight pieces per pair: [term][a] x (2 halves x 32 rows x 16 B)

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m) { return __umulhi(n, m); }
unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

__device__ __forceinline__ f32x16 mfma(const i32x4& a, const i32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// (the split itself: s3_common.hpp)
__device__ __forceinline__ void split3(float v, unsigned short (&t)[3]) { s3_split(v, t); }

// fp32 NCDHW -> S3.  One thread per voxel of one 8-channel block: 8 coalesced dword loads, three 16-byte stores.  The C
// channels land at blocks ob0 .. ob0 + C/8 - 1 of an S3 tensor with `oblocks` blocks per sample (a half of a concat buffer).
// guard != NULL: runs only when the range guard has flagged the call (common.hpp) -- the three-term form written over the two-term one.
__global__ void __launch_bounds__(256) k_split3(const float* __restrict__ x, uint4* __restrict__ out, long S, int cblocks, int oblocks,
                                                int ob0, long xstride, const unsigned* __restrict__ guard) {
  if (guard_skip(guard, 1)) return;
  const int n = blockIdx.y / cblocks, cb = blockIdx.y % cblocks;
  const long ob = (long)n * oblocks + ob0 + cb;
  // (guarded launches come with a capped grid -- they usually leave at once -- and walk several tiles when they do run)
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < S; v += (long)gridDim.x * 256) {
    const float* xs = x + (long)n * xstride + (long)cb * 8 * S + v;
    unsigned short e[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(xs[j * S], e[j]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      uint4 o;
      o.x = e[0][t] | ((unsigned)e[1][t] << 16); o.y = e[2][t] | ((unsigned)e[3][t] << 16);
      o.z = e[4][t] | ((unsigned)e[5][t] << 16); o.w = e[6][t] | ((unsigned)e[7][t] << 16);
      out[(ob * 3 + t) * S + v] = o;
    }
  }
}

// InstanceNorm normalisation + (Leaky)ReLU (norm_act.hip k_in_act_fwd: the same arithmetic, operation for operation) writing the
// S3 form of the result -- and the fp32 tensor too when y != NULL: a layer whose only consumer is a split-operand convolution
// never exists in fp32 (whole-network forward, api.hip).
__global__ void __launch_bounds__(256) k_act_split3(const float* __restrict__ x, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, float slope, float* __restrict__ y, long ystride,
                                                    uint4* __restrict__ out, long S, int cblocks, int oblocks, int ob0) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const int n = blockIdx.y / cblocks, cb = blockIdx.y % cblocks;
  const long c0 = (long)blockIdx.y * 8;  // instance index of channel 0 of this block
  const float* xs = x + c0 * S + v;
  float* ys = y ? y + (long)n * ystride + (long)cb * 8 * S + v : nullptr;
  unsigned short e[8][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float t = (xs[j * S] - mean[c0 + j]) * rstd[c0 + j];
    t = t > 0.f ? t : t * slope;
    if (ys) ys[j * S] = t;
    split3(t, e[j]);
  }
  const long ob = (long)n * oblocks + ob0 + cb;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    uint4 o;
    o.x = e[0][t] | ((unsigned)e[1][t] << 16); o.y = e[2][t] | ((unsigned)e[3][t] << 16);
    o.z = e[4][t] | ((unsigned)e[5][t] << 16); o.w = e[6][t] | ((unsigned)e[7][t] << 16);
    out[(ob * 3 + t) * S + v] = o;
  }
}

// Packed weights: [cot = co/64][chunk = ci/8][dz][pair q][term][a = (co/32)%2][h][r = co%32][8] bf16, element j = input
// channel chunk*8 + j at in-plane tap 2q + h (zero beyond the last tap).  One plane (cot, chunk, dz) = NP * 6 KiB contiguous.
// fwd:   wp(co, ci, tap) = w[co][ci][tap]                           (so = C*T3, si = T3, flip = 0)
// dgrad: wp(ci as "co", co as "ci", tap) = w[co][ci][T3 - 1 - tap]   (so = T3,   si = C*T3, flip = 1)
__global__ void __launch_bounds__(256) k_pack_w_s3(const float* __restrict__ w, unsigned short* __restrict__ wp, int NCH, int KS, long so,
                                                   long si, int flip, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int T2 = KS * KS, NP = (T2 + 1) / 2, T3 = T2 * KS;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int r = (int)(q & 31); q >>= 5;
  const int h = (int)(q & 1); q >>= 1;
  const int a = (int)(q & 1); q >>= 1;
  const int term = (int)(q % 3); q /= 3;
  const int pr = (int)(q % NP); q /= NP;
  const int dz = (int)(q % KS); q /= KS;
  const int chunk = (int)(q % NCH);
  const int cot = (int)(q / NCH);
  const int t2 = 2 * pr + h;
  unsigned short t[3] = {0, 0, 0};
  if (t2 < T2) {
    const long co = cot * 64 + a * 32 + r, ci = chunk * 8 + j;
    const int tap = dz * T2 + t2;
    split3(w[co * so + ci * si + (flip ? T3 - 1 - tap : tap)], t);
  }
  wp[i] = t[term];
}

struct SParams {
  const uint4* xs;    // S3 input
  const uint4* wp;    // packed weights
  const float* bias;  // nullable
  float* y;           // fp32 NCDHW output
  const uint4* zeros; // >= 16 B of zeros in global memory
  int N, NCH, D, H, W, K;  // NCH = C / 8
  int P, R, RP;       // row pitch (units), brick rows, R * P
  int PT, TPP;        // positions per tile, tiles per plane
  int KT;             // K / 64
  unsigned mP, mRP;
  int npb;            // brick pieces (1 KiB) per plane, all three terms
  long ntiles;
  int tiles_per_xcd;
  int flush;          // 1: per-group accumulator restart (see the kernel)
};

struct STile {
  int n, cot, z, q0, yf, xoff;
};

__device__ __forceinline__ STile s_decode(const SParams& p, long t) {
  STile o;  // order as in conv_h.hip: output-channel tile fastest, then z: neighbouring planes share input planes in L2
  o.cot = (int)(t % p.KT); t /= p.KT;
  o.z = (int)(t % p.D); t /= p.D;
  const int tp = (int)(t % p.TPP);
  o.n = (int)(t / p.TPP);
  o.q0 = tp * p.PT;
  o.yf = (int)fdiv((unsigned)o.q0, p.mP);
  o.xoff = o.q0 - o.yf * p.P;
  return o;
}

__device__ __forceinline__ void wait_vm(int n) {  // s_waitcnt vmcnt(n) for a wave-uniform n <= 7
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
  }
}

// LDS: [brick 0][brick 1][weights 0][weights 1]; a brick = the (rows + KS - 1) x P units of ONE input plane for the three terms,
// a weight buffer = the pairs of one sub-stage.  Group = (8-channel chunk, dz); per group the brick is staged once and the
// sub-stages walk the plane's tap pairs.
template <int KS, int VB>
__global__ void __launch_bounds__(kThreads, 1) k_conv_s3(const SParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  using SU = Sub<KS>;
  constexpr int PAD = KS / 2, T2 = KS * KS, NP = SU::NP, NSUB = SU::NSUB;
  constexpr int MAXJ = 7;  // brick pieces per wave (planner: npb <= 8 * MAXJ)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const long t_lo = (long)xcd * p.tiles_per_xcd;
  long t_hi = t_lo + p.tiles_per_xcd;
  if (t_hi > p.ntiles) t_hi = p.ntiles;
  long tcur = t_lo + slot;
  if (tcur >= t_hi) return;

  int off[MAXJ];  // per-lane source offset (units) of brick piece wave + 8j relative to term 0 of the plane, or -1 = zero page
  auto decode_pieces = [&](const STile& t) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const unsigned u = (unsigned)((wave + kWaves * j) * 64 + lane);
      const unsigned term = fdiv(u, p.mRP);
      const unsigned ur = u - term * p.RP;
      const unsigned rr = fdiv(ur, p.mP);
      const int xx = (int)(ur - rr * p.P) - PAD;
      const int y = t.yf - PAD + (int)rr;
      const bool ok = term < 3u && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      off[j] = ok ? (int)(term * S + (long)y * p.W + xx) : -1;
    }
  };
  auto dz_lo = [&](const STile& t) { return PAD - t.z > 0 ? PAD - t.z : 0; };
  auto dz_hi = [&](const STile& t) { return t.z + PAD > p.D - 1 ? KS - 1 - (t.z + PAD - (p.D - 1)) : KS - 1; };

  auto issue_brick = [&](int tn, int tz, int chunk, int dz, unsigned char* buf) {
    if ((NC_S3_ABLATE & 4) && buf != lds_raw) return;
    const uint4* plane = p.xs + ((long)tn * p.NCH + chunk) * 3 * S + (long)(tz + dz - PAD) * HW;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int pc = wave + kWaves * j;
      if (pc < p.npb) {
        const uint4* src = off[j] >= 0 ? plane + off[j] : p.zeros;
        nc_dma_lds16(src, nc_lds_addr((buf + pc * 1024)));
      }
    }
  };
  auto issue_w = [&](int tcot, int chunk, int dz, int qa, int qe, unsigned char* wb) {
    const uint4* ws = p.wp + ((((long)tcot * p.NCH + chunk) * KS + dz) * NP + qa) * (kPairPieces * 64) + lane;
    const int npw = (NC_S3_ABLATE & 4) ? 0 : (qe - qa) * kPairPieces;
#pragma unroll 1
    for (int pw = wave; pw < npw; pw += kWaves)
      nc_dma_lds16((ws + pw * 64), nc_lds_addr((wb + pw * 1024)));
  };

  const int BB = p.npb * 1024;
  unsigned char* const brick0 = lds_raw;
  unsigned char* const wbuf0 = lds_raw + 2 * BB;
  constexpr int WB = SU::MAXP * kPairPieces * 1024;
  const int nbw = p.npb > wave ? (p.npb - wave + kWaves - 1) / kWaves : 0;  // brick DMAs this wave issues per plane

  STile cur = s_decode(p, tcur);
  decode_pieces(cur);
  int lo = dz_lo(cur), nv = dz_hi(cur) - lo + 1;
  issue_brick(cur.n, cur.z, 0, lo, brick0);
  issue_w(cur.cot, 0, lo, SU::q0[0], SU::q0[1], wbuf0);
  int gb = 0, gw = 0;  // parities of the brick / weight buffer being computed

  // tap 2q + h of this lane as a unit offset (a tap beyond the last reads the last tap's data against zero weights)
  auto tap_off = [&](int q) {
    const int t0 = 2 * q, t1 = 2 * q + 1 < T2 ? 2 * q + 1 : T2 - 1;
    return h ? (t1 / KS) * p.P + t1 % KS : (t0 / KS) * p.P + t0 % KS;
  };

  const int qb = wave * VB * 32 + r;  // this lane's first position in the tile
  while (true) {
    const long tnext = tcur + nslot;
    const bool more_tiles = tnext < t_hi;
    STile nxt = cur;
    if (more_tiles) nxt = s_decode(p, tnext);
    const int ngroups = p.NCH * nv;

    // acc: the MFMA accumulator of ONE group; tot: the sum of the groups, added with fp32 VALU adds (round to nearest).  The
    // matrix core aligns the 16 products of a k-step to the accumulator's exponent before adding them, so its error grows with
    // the accumulator; restarting it per group keeps it ~5 x smaller than the finished sum while most products are added.
    f32x16 acc[2][VB], tot[2][VB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[a][v][e] = 0.f; tot[a][v][e] = 0.f; }

    int chunk = 0, dzi = 0;
#pragma unroll 1
    for (int g = 0; g < ngroups; ++g) {
      int nchunk = chunk, ndz = dzi + 1;
      if (ndz == nv) { ndz = 0; ++nchunk; }
      const bool within = g + 1 < ngroups;
      const bool have_next = within || more_tiles;
      // the group after this one: (sample, plane, channel tile, chunk, dz)
      const int xn = within ? cur.n : nxt.n, xz = within ? cur.z : nxt.z, xcot = within ? cur.cot : nxt.cot;
      const int xchunk = within ? nchunk : 0, xdz = within ? lo + ndz : dz_lo(nxt);
      unsigned char* const bc = brick0 + (gb & 1) * BB;
      unsigned char* const bnx = brick0 + ((gb & 1) ^ 1) * BB;
      auto do_sub = [&](auto subc) {
        constexpr int sub = decltype(subc)::value;
        unsigned char* const wc = wbuf0 + (gw & 1) * WB;
        unsigned char* const wn = wbuf0 + ((gw & 1) ^ 1) * WB;
        // this sub-stage's weights (and, at sub 0, the brick) have landed; with several sub-stages the next brick, issued
        // behind the weights of sub 1, may stay in flight over the barrier of sub 1
        if (!(NC_S3_ABLATE & 16)) {
          if constexpr (NSUB > 1 && sub == 1) wait_vm(have_next ? nbw : 0);
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();  // ... everybody's share too; everybody is done with the buffers written next
        }
        if constexpr (sub == 0) {
          if constexpr (NSUB > 1) issue_w(cur.cot, chunk, lo + dzi, SU::q0[1], SU::q0[NSUB > 1 ? 2 : 1], wn);
          if (have_next) {
            if (!within) decode_pieces(nxt);
            issue_brick(xn, xz, xchunk, xdz, bnx);
            if constexpr (NSUB == 1) issue_w(xcot, xchunk, xdz, SU::q0[0], SU::q0[1], wn);
          }
        } else if constexpr (sub + 1 < NSUB) {
          issue_w(cur.cot, chunk, lo + dzi, SU::q0[sub + 1], SU::q0[sub + 2 <= NSUB ? sub + 2 : NSUB], wn);
        } else {
          if (have_next) issue_w(xcot, xchunk, xdz, SU::q0[0], SU::q0[1], wn);
        }

        // A fragment (pair i of the sub-stage, term, a): unit ((i*3 + term)*2 + a)*64 + h*32 + r of the weight buffer;
        // B fragment (pair q, term, v): unit term*RP + position + tap
        const i32x4* wl = reinterpret_cast<const i32x4*>(wc) + h * 32 + r;
        const i32x4* bl = reinterpret_cast<const i32x4*>(bc) + qb + cur.xoff;
        constexpr int QA = SU::q0[sub], NQ = SU::q0[sub + 1] - SU::q0[sub];
        i32x4 A[3][2], B[3][VB], nA[3][2], nB[3][VB];
        {
          const int bo = tap_off(QA);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            A[t][0] = wl[(t * 2) * 64]; A[t][1] = wl[(t * 2 + 1) * 64];
#pragma unroll
            for (int v = 0; v < VB; ++v) B[t][v] = bl[t * p.RP + bo + v * 32];
          }
        }
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
          if (s + 1 < NQ) {
            const int bo = tap_off(QA + s + 1);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              if (NC_S3_ABLATE & 1) {
                nA[t][0] = A[t][1]; nA[t][1] = A[t][0];
#pragma unroll
                for (int v = 0; v < VB; ++v) nB[t][v] = B[t][(v + 1) % VB];
                continue;
              }
              nA[t][0] = wl[(((s + 1) * 3 + t) * 2) * 64]; nA[t][1] = wl[(((s + 1) * 3 + t) * 2 + 1) * 64];
#pragma unroll
              for (int v = 0; v < VB; ++v) nB[t][v] = bl[t * p.RP + bo + v * 32];
            }
          }
          // six products per (a, v), smallest first; consecutive MFMAs go to different accumulators
          constexpr int TA[6] = {2, 1, 0, 1, 0, 0};
          constexpr int TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
          for (int m = 0; m < 6; ++m)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int v = 0; v < VB; ++v) acc[a][v] = mfma(A[TA[m]][a], B[TB[m]][v], acc[a][v]);
          if (s + 1 < NQ) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // the 6 + 3 VB reads of the next k-step spread over this one's 12 VB MFMAs
              __builtin_amdgcn_sched_group_barrier(0x100, 2 + VB, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 4 * VB, 0);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              A[t][0] = nA[t][0]; A[t][1] = nA[t][1];
#pragma unroll
              for (int v = 0; v < VB; ++v) B[t][v] = nB[t][v];
            }
          }
        }
        ++gw;
      };
      do_sub(std::integral_constant<int, 0>{});
      if constexpr (NSUB > 1) {
        do_sub(std::integral_constant<int, 1>{});
        do_sub(std::integral_constant<int, 2>{});
        do_sub(std::integral_constant<int, 3>{});
      }
      if (p.flush && !(NC_S3_ABLATE & 8)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int v = 0; v < VB; ++v)
#pragma unroll
            for (int e = 0; e < 16; ++e) { tot[a][v][e] += acc[a][v][e]; acc[a][v][e] = 0.f; }
      }
      chunk = nchunk; dzi = ndz;
      ++gb;
    }
    if (p.flush) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int v = 0; v < VB; ++v) acc[a][v] = tot[a][v];
    }

    // ---- epilogue: rows = output channels, lanes = positions; each store writes 128 contiguous bytes per half
    {
      const int cob = cur.cot * 64;
      float* yn = p.y + ((long)cur.n * p.K + cob + 4 * h) * S + (long)cur.z * HW;
      long yo[VB];
#pragma unroll
      for (int v = 0; v < VB; ++v) {
        const unsigned f = (unsigned)(cur.q0 + qb + v * 32);
        const unsigned yy = fdiv(f, p.mP);
        const unsigned xx = f - yy * p.P;
        yo[v] = ((int)yy < p.H && (int)xx < p.W) ? (long)yy * p.W + xx : -1;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float bv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = p.bias ? p.bias[cob + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] : 0.f;
#pragma unroll
        for (int v = 0; v < VB; ++v)
          if (yo[v] >= 0) {
            float* yv = yn + yo[v];
#pragma unroll
            for (int e = 0; e < 16; ++e) yv[(long)(a * 32 + (e & 3) + 8 * (e >> 2)) * S] = acc[a][v][e] + bv[e];
          }
      }
    }
    if (!more_tiles) break;
    cur = nxt;
    tcur = tnext;
    lo = dz_lo(cur);
    nv = dz_hi(cur) - lo + 1;
  }
}

struct SPlan {
  int PT, VB, P, R, RP, TPP, npb, lds;
  bool ok;
};

SPlan s_plan(int H, int W, int KS) {
  SPlan pl{};
  pl.P = W + KS - 1;
  const long plane = (long)H * pl.P;
  const int wbytes = (KS == 3 ? Sub<3>::MAXP : Sub<5>::MAXP) * kPairPieces * 1024;
  double best = 0;
  for (int VB : {2, 1}) {
    const int PT = VB * 256;
    const int rows = (pl.P - 1 + PT - 1) / pl.P + 1;
    const int R = rows + KS - 1;
    const int RP = R * pl.P;
    const int npb = (3 * RP + 4 + 63) / 64;
    const int lds = 2 * npb * 1024 + 2 * wbytes;
    if (npb > 8 * 7 || lds > kLdsMax) continue;
    const int TPP = (int)((plane + PT - 1) / PT);
    const double eff = (double)H * W / ((double)TPP * PT) * (VB == 2 ? 1.0 : 0.85);
    if (eff > best) {
      best = eff;
      pl.PT = PT; pl.VB = VB; pl.R = R; pl.RP = RP; pl.npb = npb; pl.lds = lds; pl.TPP = TPP;
      pl.ok = true;
    }
  }
  return pl;
}

bool s_shape_ok(const ConvDims& d, int Cin, int Kout) {
  if (d.kd != d.kh || d.kd != d.kw || (d.kd != 3 && d.kd != 5)) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.pd || d.pw != d.pd) return false;
  if (Cin % 8 || Kout % 64) return false;
  if ((long)d.D * d.H * d.W * 3 >= (1l << 31)) return false;  // per-lane source offsets are 32-bit unit counts
  return s_plan(d.H, d.W, d.kd).ok;
}

size_t s_packed_bytes(int Cin, int Kout, int KS) {
  return (size_t)(Kout / 64) * (Cin / 8) * KS * ((KS * KS + 1) / 2) * kPairPieces * 1024;
}
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <int KS, int VB>
int launch_s3(const SParams& p, int lds, hipStream_t s) {
  auto kern = k_conv_s3<KS, VB>;
  if (int e = raise_dyn_lds(kern, kLdsMax, "conv_s3")) return e;
  hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), lds, s, p);
  return check_launch("conv_s3");
}

// Does this layer use the two-term fp16 operand form (H2) in forward, data gradient AND weight gradient?  (nc_set_split_terms(2) and all
// three kernels cover it.)  Every producer and consumer of a layer's operands decides with THIS predicate.
bool s3_layer_h2(const ConvDims& d);

// xs_keep (only without xs_pre): the converted operand is written THERE instead of into the workspace -- the caller keeps it (the
// weight gradient of the same layer wants the same S3 tensor)
int run_s3(const float* x, const void* xs_pre, const float* w, const float* bias, float* y, const ConvDims& d, int Cin, int Kout,
           long so, long si, int flip, void* ws, size_t wsb, hipStream_t s, void* xs_keep = nullptr, bool h2 = false, unsigned* guard_pre = nullptr,
           float* stats_part = nullptr) {
  const int KS = d.kd;
  const SPlan pl = s_plan(d.H, d.W, KS);
  const long S = (long)d.D * d.H * d.W;
  // The two-term fp16 form (nc_set_split_terms(2); the caller decided with s3_layer_h2): xs_pre / xs_keep / the workspace hold H2 tensors, cells at
  // their tails (common.hpp h2_cells_offset); a forward input may be a concatenation converted in two halves (cells [0], [1]).
  // Range guard (common.hpp): an operand converted HERE from fp32 has a measured cell -- the conversion counts its low chunks, the decision is
  // taken on the device, and both kernel families are launched (`dual`): the flagged call is converted again as S3 into the same workspace
  // region and runs on the three-term kernel.  guard_pre: xs_pre is such a region of the caller's (conv_bwd_s3) with its decision already taken.
  // Workspace: [operand: the S3 tensor's 6 bytes per element | 256 B: guard words, weight cell | packed weights of either form].
  if (h2) {
    const size_t ex = (size_t)d.N * Cin * S;
    const bool internal = !xs_pre && !xs_keep;
    const size_t p2 = s3x_packed_bytes(Cin, Kout, KS, 2), p3 = s3x_packed_bytes(Cin, Kout, KS, 3);
    const size_t xb6 = align256(ex * 6), xb4 = h2_cells_offset(ex) + 256;
    bool dual = guard_pre || (internal && x && h2_guard_can_flip() && wsb >= xb6 + 256 + p3 + 256);
    if (guard_pre && wsb < 256 + p3 + 256) { set_error("conv_s3 (two-term, guarded operand): workspace too small"); return NC_ERR_WS; }
    const size_t xb = !internal ? 0 : dual ? xb6 : xb4;
    if (!ws || wsb < xb + 256 + p2 + 256) { set_error("conv_s3 (two-term): workspace too small"); return NC_ERR_WS; }
    void* xs = xs_pre ? const_cast<void*>(xs_pre) : xs_keep ? xs_keep : ws;
    unsigned* cells = h2_cells_of(xs, ex);
    unsigned* gw = (unsigned*)((char*)ws + xb);  // words 0..7: the guard of an operand measured here; word 16: the weights' cell
    unsigned* guard = guard_pre ? guard_pre : gw;
    if (!xs_pre) {
      if (int e = h2_guard_zero(gw, s)) return e;
      if (int e = h2_zero_cells(cells, 2, s)) return e;
      if (int e = h2_absmax(x, (long)ex, cells, s, cells + 1)) return e;
      if (int e = split2h_into(x, (long)Cin * S, xs, d.N, Cin, S, Cin, 0, cells, s, gw)) return e;
      if (int e = h2_guard_decide(gw, nullptr, nullptr, gw + kGuardFlag, dual, s)) return e;  // (not dual: the event is counted only)
      if (dual)
        if (int e = split3_into(x, (long)Cin * S, xs, d.N, Cin, S, Cin, 0, s, gw)) return e;
    }
    if (stats_part && dual) { set_error("conv_s3 (two-term): epilogue statistics need an operand that cannot fall back"); return NC_ERR_ARG; }
    if (int e = conv_s3x_h2(xs, cells, cells + 1, flip ? Cin : Cin / 2, w, bias, y, d.N, Cin, d.D, d.H, d.W, Kout, KS, so, si, flip, gw + 16,
                            (char*)ws + xb + 256, s, dual ? guard : nullptr, stats_part)) return e;
    if (dual) return conv_s3x(xs, w, bias, y, d.N, Cin, d.D, d.H, d.W, Kout, KS, so, si, flip, (char*)ws + xb + 256, s, guard);
    return NC_OK;
  }
  const size_t xb = (xs_pre || xs_keep) ? 0 : align256((size_t)d.N * Cin * S * 6);
  const size_t wb = align256(s_packed_bytes(Cin, Kout, KS));
  if (!ws || wsb < xb + wb + 256) { set_error("conv_s3: workspace too small"); return NC_ERR_WS; }
  uint4* xs = xs_pre ? (uint4*)xs_pre : xs_keep ? (uint4*)xs_keep : (uint4*)ws;
  unsigned short* wp = (unsigned short*)((char*)ws + xb);
  const uint4* zeros = reinterpret_cast<const uint4*>(nc_zero_page());
  if (!zeros) { set_error("conv_s3: no zero page"); return NC_ERR_HIP; }
  if (!xs_pre) {
    hipLaunchKernelGGL(k_split3, dim3((unsigned)cdiv(S, 256), (unsigned)(d.N * Cin / 8)), dim3(256), 0, s, x, xs, S, Cin / 8, Cin / 8, 0,
                       (long)Cin * S, (const unsigned*)nullptr);
    if (int e = check_launch("split3")) return e;
  }
  // the tap-stream kernel (conv_s3x.hip: 16x16x32 MFMAs, no zero tap) takes every shape it covers; NC_S3X=0: the pair kernel below
  static const int use_x = getenv("NC_S3X") ? atoi(getenv("NC_S3X")) : 1;
  if (use_x && s3x_supported(d.N, Cin, d.D, d.H, d.W, Kout, KS) && s3x_packed_bytes(Cin, Kout, KS) <= wb)
    return conv_s3x(xs, w, bias, y, d.N, Cin, d.D, d.H, d.W, Kout, KS, so, si, flip, wp, s);
  const long total = (long)(s_packed_bytes(Cin, Kout, KS) / 2);
  hipLaunchKernelGGL(k_pack_w_s3, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, Cin / 8, KS, so, si, flip, total);
  if (int e = check_launch("pack_w_s3")) return e;
  SParams p{};
  p.xs = xs; p.wp = (const uint4*)wp; p.bias = bias; p.y = y; p.zeros = zeros;
  p.N = d.N; p.NCH = Cin / 8; p.D = d.D; p.H = d.H; p.W = d.W; p.K = Kout;
  p.P = pl.P; p.R = pl.R; p.RP = pl.RP; p.PT = pl.PT; p.TPP = pl.TPP; p.KT = Kout / 64;
  p.mP = magic(pl.P); p.mRP = magic(pl.RP);
  p.npb = pl.npb;
  p.ntiles = (long)d.N * d.D * pl.TPP * p.KT;
  p.tiles_per_xcd = (int)cdiv(p.ntiles, 8);
  static const int flush = 1;
  p.flush = flush;
  if (KS == 3) return pl.VB == 2 ? launch_s3<3, 2>(p, pl.lds, s) : launch_s3<3, 1>(p, pl.lds, s);
  return pl.VB == 2 ? launch_s3<5, 2>(p, pl.lds, s) : launch_s3<5, 1>(p, pl.lds, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient dW[k][c][tap] = sum_v dY[k][v] X[c][v + tap] of the same layers (nn.Conv3d backward-weight at the same
// call sites): M = 32 output channels, N = 32 input channels, K-dim = 16 voxels per MFMA, both operands from S3 images in
// LDS through the transposing read ds_read_b64_tr_b16 (a lane receives 4 consecutive voxels of ITS channel; two reads make
// the 8-voxel fragment).  The structure is the 16-bit kernel's (conv_h.hip k_wgrad_h): workgroup = 64 k x 32 c x all taps of
// ZR kernel planes, 8 waves = 2 k-halves x 4 tap groups; it walks a contiguous range of (sample, tile, z) steps with the X
// planes in an LDS ring (one new X plane + one dY plane per step by LDS-DMA); per k-step and tap the three terms of both
// operands are read and six MFMAs issued.  One partial per workgroup, summed in workgroup order (deterministic).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* ltr_t;

struct WsParams {
  const uint4* xs;   // S3 X   [N][C/8][3][D][H][W]
  const uint4* dys;  // S3 dY  [N][K/8][3][D][H][W]
  float* part;       // [pairs * nwp][NF][TW][64][32]
  const uint4* zeros;
  int N, C, K, D, H, W;
  int Ty, Tx, YB, XB;
  int F, NF;         // steps between two accumulator restarts, partial slots per workgroup
  int Xp, XU, XUp;   // X image: row pitch Tx + 2p, units per sub-block, padded sub-block stride (== 4 mod 8)
  int PT, PTp, NK;   // dY image: positions Ty*Tx, padded sub-block stride, k-steps = ceil(PT / 16)
  int npx, npd;      // 1 KiB pieces per X slot / per dY buffer
  int xslot, dybuf;  // bytes
  int nct;           // C / 32
  int npairs, nwp;   // (k-tile, c-tile[, kernel plane]) pairs, workgroups per pair
  long steps;        // N * YB * XB * D  (per pair)
  unsigned mTx, mXp, mXUp, mPTp;
  const unsigned* guard;  // nullable: the range guard's words (common.hpp)
};

__device__ __forceinline__ i32x4 tr_frag(const unsigned char* lds, unsigned a0, unsigned a1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds + a1));
  return __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int KS>
__global__ void __launch_bounds__(kThreads, 1) k_wgrad_s3(const WsParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  if (guard_skip(p.guard, 1)) return;
  // 3^3: one workgroup owns all 27 taps (ZR = 3 kernel planes, 4-slot X ring).  5^3: a workgroup owns the 25 taps of ONE
  // kernel plane (ZR = 1, 2-slot ring) and dz joins (k-tile, c-tile) in the "pair" index.
  constexpr int PAD = KS / 2, T2 = KS * KS;
  constexpr int ZR = KS == 3 ? 3 : 1, NS = ZR + 1, NDG = KS / ZR, TW = ZR * T2;
  constexpr int TG = (TW + 3) / 4;  // taps per wave group: 7,7,7,6 of 27; 7,6,6,6 of 25
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 1, tg = wave >> 1;
  const int t0 = tg * (TW / 4) + (tg < TW % 4 ? tg : TW % 4);
  const int ntap = TW / 4 + (tg < TW % 4 ? 1 : 0);
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;

  // logical workgroup id: neighbours on an XCD share dY (and, for 5^3, X) in that XCD's L2
  const int G = gridDim.x, xcd = blockIdx.x & 7;
  const int wg = (G >> 3) * xcd + ((G & 7) < xcd ? (G & 7) : xcd) + (blockIdx.x >> 3);
  const int pair = wg % p.npairs, wi = wg / p.npairs;
  if (wi >= p.nwp) return;
  const int dzg = pair % NDG, kc = pair / NDG;
  const int kt = kc / p.nct, ct = kc % p.nct;
  const int zsh = dzg * ZR - PAD;  // X plane of local kernel plane l at step z: z + zsh + l
  const long s_lo = p.steps * wi / p.nwp, s_hi = p.steps * (wi + 1) / p.nwp;

  unsigned char* const xring = lds_raw;               // NS slots; plane pz lives in slot (pz + 16) % NS
  unsigned char* const dyb = lds_raw + NS * p.xslot;  // 2 buffers
  auto slot_of = [&](int pz) { return ((pz + 16) & (NS - 1)) * p.xslot; };

  // LDS-DMA of one X plane (4 blocks x 3 terms of this c-tile, tile + halo) / one dY plane (8 blocks x 3 terms of this k-tile)
  auto issue_x = [&](int n, int y0, int x0, int pz, unsigned char* slot) {
    const bool zok = (unsigned)pz < (unsigned)p.D;
    const uint4* base = p.xs + ((long)n * (p.C / 8) + ct * 4) * 3 * S + (long)(zok ? pz : 0) * HW;
#pragma unroll 1
    for (int pc = wave; pc < p.npx; pc += kWaves) {
      const unsigned u = (unsigned)(pc * 64 + lane);
      const unsigned sb = fdiv(u, p.mXUp);  // sub-block = block * 3 + term
      const unsigned ur = u - sb * p.XUp;
      const unsigned ty = fdiv(ur, p.mXp);
      const int y = y0 - PAD + (int)ty, x = x0 - PAD + (int)(ur - ty * p.Xp);
      const bool ok = zok && sb < 12u && ur < (unsigned)p.XU && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const uint4* src = ok ? base + (long)sb * S + (long)y * p.W + x : p.zeros;
      nc_dma_lds16(src, nc_lds_addr((slot + pc * 1024)));
    }
  };
  auto issue_dy = [&](int n, int y0, int x0, int z, unsigned char* buf) {
    const uint4* base = p.dys + ((long)n * (p.K / 8) + kt * 8) * 3 * S + (long)z * HW;
#pragma unroll 1
    for (int pc = wave; pc < p.npd; pc += kWaves) {
      const unsigned u = (unsigned)(pc * 64 + lane);
      const unsigned sb = fdiv(u, p.mPTp);
      const unsigned rho = u - sb * p.PTp;
      const unsigned ty = fdiv(rho, p.mTx);
      const int y = y0 + (int)ty, x = x0 + (int)(rho - ty * p.Tx);
      const bool ok = sb < 24u && rho < (unsigned)p.PT && y < p.H && x < p.W;
      const uint4* src = ok ? base + (long)sb * S + (long)y * p.W + x : p.zeros;
      nc_dma_lds16(src, nc_lds_addr((buf + pc * 1024)));
    }
  };

  // transposed-read roles of this lane: group g = lane/16 -> channels 16*(g&1).. of the 32-channel tile, voxel half h = g/2;
  // inside the group lane 4q+pp supplies the address of voxel row q, channels 4pp..4pp+3
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = g >> 1;
  const int cbsel = 2 * (g & 1) + (pp >> 1);
  const unsigned a_lane = (unsigned)((((mt * 4 + cbsel) * 3) * p.PTp) * 16 + (pp & 1) * 8);  // + term * PTp * 16 + rho * 16
  const unsigned b_lane = (unsigned)(((cbsel * 3) * p.XUp) * 16 + (pp & 1) * 8);             // + term * XUp * 16 + (ty * Xp + tx) * 16
  const unsigned a_term = (unsigned)p.PTp * 16, b_term = (unsigned)p.XUp * 16;

  f32x16 acc[TG];
#pragma unroll
  for (int j = 0; j < TG; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  int tdz[TG], toff[TG];  // tap j of this wave: kernel plane and byte offset inside an X slot
#pragma unroll
  for (int j = 0; j < TG; ++j) {
    const int t = t0 + j < TW ? t0 + j : TW - 1;
    const int dz = t / T2, dy = (t / KS) % KS, dx = t % KS;
    tdz[j] = dz;
    toff[j] = (dy * p.Xp + dx) * 16;
  }

  // partial slot f of this workgroup: part[wg][f][tap][k 0..63][c 0..31]; rows of the accumulator tile are k, lanes are c.
  // The accumulators restart every F steps (the matrix core aligns a k-step's products to the accumulator's exponent: its
  // rounding error grows with the running sum, see k_conv_s3); the slots are added in fp32 by k_wgrad_s3_reduce.
  int nflush = 0, since = 0;
  auto write_partial = [&](bool live) {
    float* pw = p.part + ((long)wg * p.NF + nflush) * TW * 64 * 32;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TG; ++j)
      if (j < ntap) {
        float* pt = pw + ((long)(t0 + j) * 64 + mt * 32 + 4 * hh) * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          pt[((e & 3) + 8 * (e >> 2)) * 32] = live ? acc[j][e] : 0.f;
          acc[j][e] = 0.f;
        }
      }
  };

  long step = s_lo;
  bool fresh = true;  // the ring has to be (re)filled: first step of this workgroup or of a new (sample, tile)
  int n = 0, y0 = 0, x0 = 0, z = 0;
  while (step < s_hi) {
    if (fresh) {
      long j = step;
      z = (int)(j % p.D); j /= p.D;
      const int xb = (int)(j % p.XB); j /= p.XB;
      const int yb = (int)(j % p.YB);
      n = (int)(j / p.YB);
      y0 = yb * p.Ty; x0 = xb * p.Tx;
      __syncthreads();  // everybody is done with the buffers of the previous tile
#pragma unroll
      for (int l = 0; l < ZR; ++l) issue_x(n, y0, x0, z + zsh + l, xring + slot_of(z + zsh + l));
      issue_dy(n, y0, x0, z, dyb + (z & 1) * p.dybuf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      fresh = false;
    }
    const bool cont = z + 1 < p.D && step + 1 < s_hi;  // the next step continues this tile
    if (cont) {
      issue_x(n, y0, x0, z + 1 + zsh + ZR - 1, xring + slot_of(z + 1 + zsh + ZR - 1));
      issue_dy(n, y0, x0, z + 1, dyb + ((z + 1) & 1) * p.dybuf);
    }
    if (since == p.F && nflush + 1 < p.NF) {
      write_partial(true);
      ++nflush;
      since = 0;
    }
    ++since;
    // ---- multiply: NK k-steps x ntap taps x 6 term products
    const unsigned abase = (unsigned)(NS * p.xslot + (z & 1) * p.dybuf) + a_lane;
    unsigned sb[TG];  // slot base + tap offset
#pragma unroll
    for (int j = 0; j < TG; ++j) sb[j] = (unsigned)(slot_of(z + zsh + tdz[j]) + toff[j]) + b_lane;
#pragma unroll 1
    for (int s = 0; s < p.NK; ++s) {
      unsigned rho[2], bo[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        rho[s2] = (unsigned)(16 * s + 8 * h + 4 * s2 + q);
        unsigned rc = rho[s2] < (unsigned)p.PT ? rho[s2] : (unsigned)p.PT - 1;  // padded positions: dY = 0, X any finite
        const unsigned ty = fdiv(rc, p.mTx);
        bo[s2] = (ty * p.Xp + (rc - ty * p.Tx)) * 16;
      }
      i32x4 A[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) A[t] = tr_frag(lds_raw, abase + t * a_term + rho[0] * 16, abase + t * a_term + rho[1] * 16);
      // taps two at a time: consecutive MFMAs go to different accumulators (six dependent MFMAs in a row leave issue gaps)
#pragma unroll
      for (int j = 0; j < TG; j += 2) {
        if (j < ntap) {
          const bool two = j + 1 < TG && j + 1 < ntap;
          i32x4 B[3], B2[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) B[t] = tr_frag(lds_raw, sb[j] + t * b_term + bo[0], sb[j] + t * b_term + bo[1]);
          if (j + 1 < TG) {
            if (two) {
#pragma unroll
              for (int t = 0; t < 3; ++t) B2[t] = tr_frag(lds_raw, sb[j + 1] + t * b_term + bo[0], sb[j + 1] + t * b_term + bo[1]);
            }
          }
          constexpr int TA[6] = {2, 1, 0, 1, 0, 0};
          constexpr int TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
          for (int m = 0; m < 6; ++m) {
            acc[j] = mfma(A[TA[m]], B[TB[m]], acc[j]);
            if (j + 1 < TG) {
              if (two) acc[j + 1] = mfma(A[TA[m]], B2[TB[m]], acc[j + 1]);
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++step;
    if (cont) ++z; else fresh = true;
  }

  write_partial(nflush < p.NF);
  for (++nflush; nflush < p.NF; ++nflush) write_partial(false);  // slots this workgroup did not need: zeros
}

// The same weight gradient on v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock under it than under 32x32x16, tools/mfma_rate.hip):
// K-dim = 32 voxels per MFMA.  Same LDS images and partial layout as k_wgrad_s3; what differs (ablation builds of the first 16x16x32
// version, tools/ws_variant.sh: staging DMA 18 % of the time, the B fragment reads 10 %, the step barrier 1 %):
//  * wave tile = all 64 k of the workgroup x 7 "units" (tap, 16-channel block of c): 12 A fragments per k-step serve 7 x 24 MFMAs and
//    a unit's three B fragments 24 MFMAs (the 32 k x 32 c x 7 taps tile read 48 fragments per 168 MFMAs, this one 33);
//  * staging through buffer descriptors: a lane's source offset per 1 KiB piece depends on the tile only, not on z, so it is computed
//    once per tile and kept in registers (kWP pieces per wave and image); a step's DMA is one instruction per piece with the plane as
//    scalar offset; padding lanes (and whole planes outside the volume) ask beyond the descriptor's range and get zeros from the
//    hardware -- no zero page, no per-lane 64-bit addresses, no divisions per step.
// Lane group g = lane / 16 supplies voxels 8g .. 8g + 7 of a k-step for channel lane % 16 of a 16-channel block (two transposing reads).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int DT>
__device__ __forceinline__ f32x4 mfma16(const i32x4& a, const i32x4& b, const f32x4& c) {
  if constexpr (DT == NC_DT_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The staging DMA of k_wgrad_s3x is inline assembly on purpose: with the builtin the compiler knows that LDS writes are in flight and
// orders every LDS read behind them (s_waitcnt vmcnt(0) at the head of each k-step, i.e. each step waited for its own staging
// requests -- which are for the NEXT step -- before multiplying: 16 % of the kernel's time).  The kernel waits by hand where the data
// are needed: vmcnt(0) + barrier at the end of a step.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 dma_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = __builtin_amdgcn_readfirstlane(0x00020000u);
  return r;
}
__device__ __forceinline__ void dma16(const u32x4& rs, const unsigned char* lds_dst, unsigned voff, int soff) {
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

constexpr int kWP = 6;  // 1 KiB pieces per wave of an X slot / a dY buffer (planner: npx, npd <= 8 * kWP)

// NT = 3: S3 operands (three bf16 terms, six products per fp32 product); NT = 1: the 16-bit operands of the --precision path (C8 layout =
// the same tensor with one term; DT picks v_mfma_f32_16x16x32_bf16 / _f16).
template <int KS, int NT, int DT>
__global__ void __launch_bounds__(kThreads, 1) k_wgrad_s3x(const WsParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  if (guard_skip(p.guard, NT == 3)) return;
  constexpr int PAD = KS / 2, T2 = KS * KS;
  constexpr int ZR = KS == 3 ? 3 : 1, NS = ZR + 1, NDG = KS / ZR, TW = ZR * T2;
  constexpr int UW = 2 * TW, NU = (UW + 7) / 8;  // units (tap, c-block) of the workgroup / most units of a wave
  constexpr unsigned kOut = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u0 = wave * (UW / 8) + (wave < UW % 8 ? wave : UW % 8);
  const int nu = UW / 8 + (wave < UW % 8 ? 1 : 0);
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;

  const int G = gridDim.x, xcd = blockIdx.x & 7;
  const int wg = (G >> 3) * xcd + ((G & 7) < xcd ? (G & 7) : xcd) + (blockIdx.x >> 3);
  const int pair = wg % p.npairs, wi = wg / p.npairs;
  if (wi >= p.nwp) return;
  const int dzg = pair % NDG, kc = pair / NDG;
  const int kt = kc / p.nct, ct = kc % p.nct;
  const int zsh = dzg * ZR - PAD;
  const long s_lo = p.steps * wi / p.nwp, s_hi = p.steps * (wi + 1) / p.nwp;

  unsigned char* const xring = lds_raw;
  unsigned char* const dyb = lds_raw + NS * p.xslot;
  auto slot_of = [&](int pz) __attribute__((always_inline)) { return ((pz + 16) & (NS - 1)) * p.xslot; };

  // per-tile source offsets (bytes inside the 12 / 24 sub-blocks of this (sample, c-tile) / (sample, k-tile), plane 0) of the pieces
  // wave + 8 i; kOut = padding
  unsigned xo[kWP], yo[kWP];
  auto tile_offsets = [&](int y0, int x0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kWP; ++i) {
      const int pc = wave + kWaves * i;
      {
        const unsigned u = (unsigned)(pc * 64 + lane);
        const unsigned sb = fdiv(u, p.mXUp);  // sub-block = block * NT + term
        const unsigned ur = u - sb * p.XUp;
        const unsigned ty = fdiv(ur, p.mXp);
        const int y = y0 - PAD + (int)ty, x = x0 - PAD + (int)(ur - ty * p.Xp);
        const bool ok = pc < p.npx && sb < 4u * NT && ur < (unsigned)p.XU && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        xo[i] = ok ? (sb * (unsigned)S + (unsigned)(y * p.W + x)) * 16u : kOut;
      }
      {
        const unsigned u = (unsigned)(pc * 64 + lane);
        const unsigned sb = fdiv(u, p.mPTp);
        const unsigned rho = u - sb * p.PTp;
        const unsigned ty = fdiv(rho, p.mTx);
        const int y = y0 + (int)ty, x = x0 + (int)(rho - ty * p.Tx);
        const bool ok = pc < p.npd && sb < 8u * NT && rho < (unsigned)p.PT && y < p.H && x < p.W;
        yo[i] = ok ? (sb * (unsigned)S + (unsigned)(y * p.W + x)) * 16u : kOut;
      }
    }
  };
  auto issue_x = [&](int n, int pz, unsigned char* slot) __attribute__((always_inline)) {
    const bool zok = (unsigned)pz < (unsigned)p.D;
    const uint4* base = p.xs + ((long)n * (p.C / 8) + ct * 4) * NT * S;
    const u32x4 rs = dma_rsrc(base, zok ? (unsigned)(4 * NT * S * 16) : 0u);
#ifdef NC_WA_SAMESRC
    const int soff = 0;
#else
    const int soff = __builtin_amdgcn_readfirstlane(zok ? (int)(pz * HW * 16) : 0);
#endif
#pragma unroll
    for (int i = 0; i < kWP; ++i)
#ifdef NC_WA_ZEROSRC
      if (wave + kWaves * i < p.npx) dma16(rs, slot + (wave + kWaves * i) * 1024, kOut, soff);
#else
      if (wave + kWaves * i < p.npx) dma16(rs, slot + (wave + kWaves * i) * 1024, xo[i], soff);
#endif
  };
  auto issue_dy = [&](int n, int z, unsigned char* buf) __attribute__((always_inline)) {
    const uint4* base = p.dys + ((long)n * (p.K / 8) + kt * 8) * NT * S;
    const u32x4 rs = dma_rsrc(base, (unsigned)(8 * NT * S * 16));
#ifdef NC_WA_SAMESRC
    const int soff = 0;
#else
    const int soff = __builtin_amdgcn_readfirstlane((int)(z * HW * 16));
#endif
#pragma unroll
    for (int i = 0; i < kWP; ++i)
#ifdef NC_WA_ZEROSRC
      if (wave + kWaves * i < p.npd) dma16(rs, buf + (wave + kWaves * i) * 1024, kOut, soff);
#else
      if (wave + kWaves * i < p.npd) dma16(rs, buf + (wave + kWaves * i) * 1024, yo[i], soff);
#endif
  };

  // transposed-read roles: lane = 16g + 4q + pp: voxel row 8g + 4*s2 + q of the k-step, channels 4pp .. 4pp + 3 of a 16-channel block
  // (= 8-channel sub-blocks 2*blk + (pp >> 1), byte (pp & 1) * 8 of the unit)
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const unsigned a_term = (unsigned)p.PTp * 16, b_term = (unsigned)p.XUp * 16;
  const unsigned a_lane = (unsigned)((((pp >> 1) * NT) * p.PTp) * 16 + (pp & 1) * 8), a_blk = 2u * NT * a_term;  // + a * a_blk: k rows 16a ..
  const unsigned b_lane = (unsigned)((((pp >> 1) * NT) * p.XUp) * 16 + (pp & 1) * 8), b_blk = 2u * NT * b_term;  // + b * b_blk: c 16b ..

  f32x4 acc[NU][4];
#pragma unroll
  for (int j = 0; j < NU; ++j)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][a][e] = 0.f;

  int udz[NU];
  unsigned uoff[NU];  // unit j of this wave = (tap, c-block): kernel plane and byte offset inside an X slot (tap shift + c-block + lane)
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const int u = u0 + j < UW ? u0 + j : UW - 1;
    const int t = u >> 1, b = u & 1;
    const int dz = t / T2, dy = (t / KS) % KS, dx = t % KS;
    udz[j] = dz;
    uoff[j] = (unsigned)((dy * p.Xp + dx) * 16) + b * b_blk + b_lane;
  }

  // partial slot f: part[wg][f][tap][k 0..63][c 0..31]; accumulator element e of (unit (tap, b), a) = k a*16 + 4g + e, c b*16 + lane%16
  int nflush = 0, since = 0;
  auto write_partial = [&](bool live) __attribute__((always_inline)) {
    float* pw = p.part + ((long)wg * p.NF + nflush) * TW * 64 * 32;
    const int m16 = lane & 15;
#pragma unroll
    for (int j = 0; j < NU; ++j)
      if (j < nu) {
        const int u = u0 + j;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float* pt = pw + ((long)(u >> 1) * 64 + a * 16 + 4 * g) * 32 + (u & 1) * 16 + m16;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pt[e * 32] = live ? acc[j][a][e] : 0.f;
            acc[j][a][e] = 0.f;
          }
        }
      }
  };

  long step = s_lo;
  bool fresh = true;
  int n = 0, z = 0;
  while (step < s_hi) {
    if (fresh) {
      long j = step;
      z = (int)(j % p.D); j /= p.D;
      const int xb = (int)(j % p.XB); j /= p.XB;
      const int yb = (int)(j % p.YB);
      n = (int)(j / p.YB);
      tile_offsets(yb * p.Ty, xb * p.Tx);
      __syncthreads();
#pragma unroll
      for (int l = 0; l < ZR; ++l) issue_x(n, z + zsh + l, xring + slot_of(z + zsh + l));
      issue_dy(n, z, dyb + (z & 1) * p.dybuf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      fresh = false;
    }
    const bool cont = z + 1 < p.D && step + 1 < s_hi;
#ifndef NC_WA_NODMA
    if (cont) {
      issue_x(n, z + 1 + zsh + ZR - 1, xring + slot_of(z + 1 + zsh + ZR - 1));
      issue_dy(n, z + 1, dyb + ((z + 1) & 1) * p.dybuf);
    }
#endif
    if (since == p.F && nflush + 1 < p.NF) {
      write_partial(true);
      ++nflush;
      since = 0;
    }
    ++since;
    // ---- multiply: NK k-steps of 32 voxels x nu units x 4 row blocks x 6 term products
    const unsigned abase = (unsigned)(NS * p.xslot + (z & 1) * p.dybuf) + a_lane;
    unsigned sb[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) sb[j] = (unsigned)slot_of(z + zsh + udz[j]) + uoff[j];
    auto kstep_addr = [&](int s, unsigned (&rho)[2], unsigned (&bo)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        rho[s2] = (unsigned)(32 * s + 8 * g + 4 * s2 + q);
        const unsigned rc = rho[s2] < (unsigned)p.PT ? rho[s2] : (unsigned)p.PT - 1;  // padded positions: dY = 0, X any finite
        const unsigned ty = fdiv(rc, p.mTx);
        bo[s2] = (ty * p.Xp + (rc - ty * p.Tx)) * 16;
      }
    };
    if constexpr (NT == 1) {
      // One term: a k-step is 4 A + NU B fragments and 4 NU MFMAs -- too short to hide a unit's reads under the unit before.  B slot j
      // (unit j) is free once unit j has been multiplied and is refilled with the NEXT k-step's fragment four units later; the A
      // fragments are double-buffered (the reads of k-step s + 1 spread over k-step s).
      i32x4 Fa[2][4], Fb[NU];
      unsigned rho[2], bo[2], nrho[2], nbo[2];
      kstep_addr(0, rho, bo);
#pragma unroll
      for (int a = 0; a < 4; ++a) Fa[0][a] = tr_frag(lds_raw, abase + a * a_blk + rho[0] * 16, abase + a * a_blk + rho[1] * 16);
#pragma unroll
      for (int j = 0; j < NU; ++j) Fb[j] = tr_frag(lds_raw, sb[j] + bo[0], sb[j] + bo[1]);
      auto kstep1 = [&](auto cur, int s) __attribute__((always_inline)) {
        constexpr int CU = decltype(cur)::value;
        const bool more = s + 1 < p.NK;
        kstep_addr(more ? s + 1 : s, nrho, nbo);  // (last k-step: harmless re-reads)
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          if (j < 4) Fa[CU ^ 1][j] = tr_frag(lds_raw, abase + j * a_blk + nrho[0] * 16, abase + j * a_blk + nrho[1] * 16);
          if (j + 1 < NU || nu == NU) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[j][a] = mfma16<DT>(Fa[CU][a], Fb[j], acc[j][a]);
          }
          if (j >= 3) {  // refill the slots units j - 3 (.. and, behind the last unit, the rest) have left
            Fb[j - 3] = tr_frag(lds_raw, sb[j - 3] + nbo[0], sb[j - 3] + nbo[1]);
          }
        }
#pragma unroll
        for (int j = NU - 3; j < NU; ++j) Fb[j] = tr_frag(lds_raw, sb[j] + nbo[0], sb[j] + nbo[1]);
      };
      int s = 0;
#pragma unroll 1
      for (; s + 1 < p.NK; s += 2) {
        kstep1(std::integral_constant<int, 0>{}, s);
        kstep1(std::integral_constant<int, 1>{}, s + 1);
      }
      if (s < p.NK) kstep1(std::integral_constant<int, 0>{}, s);
    } else {
#pragma unroll 1
    for (int s = 0; s < p.NK; ++s) {
      unsigned rho[2], bo[2];
      kstep_addr(s, rho, bo);
      // One k-step: 4 NT A + NT B fragments up front in the order the products need them, then per unit the reads of the next unit's B
      // fragments spread over its MFMAs.  The order is pinned with sched_group_barrier: left alone the scheduler sinks every read next to
      // its first use and the k-step pays an LDS round trip a dozen times.
      constexpr int NP = NT == 3 ? 6 : NT == 2 ? 3 : 1;  // products per (row block, unit), smallest first: (term of A, term of B)
      constexpr int TA[6] = {NT - 1, NT == 3 ? 1 : 0, 0, 1, 0, 0};
      constexpr int TB[6] = {0, 1, NT == 3 ? 2 : 0, 0, 1, 0};
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
        if (j + 1 < NU || nu == NU) {  // (units past a wave's last are clamped: their reads are harmless, their products are skipped)
#ifndef NC_WA_NOB
          if (j + 1 < NU) {
#pragma unroll
            for (int t = 0; t < NT; ++t) read_b1(B[(j + 1) & 1][t], j + 1, t);
          }
#endif
#pragma unroll
          for (int m = 0; m < NP; ++m)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[j][a] = mfma16<DT>(A[a][TA[m]], B[j & 1][TB[m]], acc[j][a]);
          if constexpr (NT == 2) {  // the 4 reads of the next unit's fragments over this unit's 12 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
              if (j + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, NT == 3 ? 1 : 2, 0);
            }
          }
        }
      }
    }
    }
#ifndef NC_WA_NOBAR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    ++step;
    if (cont) ++z; else fresh = true;
  }

  write_partial(nflush < p.NF);
  for (++nflush; nflush < p.NF; ++nflush) write_partial(false);
}

// dw[k][c][tap] = sum of the nwp * NF partial slots of pair (k/64, c/32, tap / TW).  A workgroup handles 32 consecutive outputs (one
// 128-byte run of c); its 8 lane groups each add a contiguous eighth of the slots in slot order, and the eight sums are added in group
// order: a fixed order (deterministic), with eight loads in flight per output instead of one dependent chain of several hundred
// (the one-thread-per-output form ran at ~1 TB/s: 0.85 ms of the 108^3 step).
// xcells != NULL (H2 operands): the sums are scaled back by 2^-(kx + ky), kx per half of the input channels (h2.hip)
__global__ void __launch_bounds__(256) k_wgrad_s3_reduce(const float* __restrict__ part, float* __restrict__ dw, int C, int T3, int TW,
                                                         int nct, int npairs, int nwp, int NF, long total,
                                                         const unsigned* __restrict__ xcells = nullptr, const unsigned* __restrict__ ycell = nullptr,
                                                         const unsigned* __restrict__ guard = nullptr) {
  __shared__ float red[8][32];
  if (guard_skip(guard, xcells ? 0 : 1)) return;
  const int o = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + o;  // (k, tap, c): c fastest -> coalesced partial reads
  float sacc = 0.f;
  int c = 0, t = 0, k = 0;
  if (i < total) {
    c = (int)(i % C);
    t = (int)((i / C) % T3);
    k = (int)(i / ((long)C * T3));
    const int ndg = T3 / TW;
    const int pair = ((k / 64) * nct + c / 32) * ndg + t / TW;
    const long off = ((long)(t % TW) * 64 + (k & 63)) * 32 + (c & 31);
    const int nslots = nwp * NF, per = (nslots + 7) / 8;
    const int s0 = seg * per, s1 = s0 + per < nslots ? s0 + per : nslots;
    for (int sl = s0; sl < s1; ++sl) {
      const int w = sl / NF, f = sl - w * NF;
      sacc += part[(((long)(w * npairs + pair) * NF + f) * TW) * 64 * 32 + off];
    }
  }
  red[seg][o] = sacc;
  __syncthreads();
  if (seg == 0 && i < total) {
    float r = red[0][o];
#pragma unroll
    for (int g = 1; g < 8; ++g) r += red[g][o];
    if (xcells) { const float2 f = h2_unscale2(xcells[c >= C / 2 ? 1 : 0], *ycell); r = r * f.x * f.y; }
    dw[((long)k * C + c) * T3 + t] = r;
  }
}

struct WsPlan {
  int Ty, Tx, YB, XB, Xp, XU, XUp, PT, PTp, NK, npx, npd, xslot, dybuf;
  bool ok;
};

int pad_4mod8(int v) { return v + ((4 - (v & 7)) & 7); }  // sub-block stride that keeps the transposed reads conflict-free

int ws_kstep() {  // voxels per k-step: 32 = the 16x16x32 kernel (default), 16 = the 32x32x16 kernel (NC_S3X_WGRAD=0)
  static const int x = getenv("NC_S3X_WGRAD") ? atoi(getenv("NC_S3X_WGRAD")) : 1;
  return x ? 32 : 16;
}

// k_wgrad_s3x addresses a (sample, k-tile)'s 8 NT sub-blocks through one buffer descriptor with 32-bit offsets (bit 31 = "padding")
int ws_kv(const ConvDims& d, int NT = 3) { return (NT == 1 || ws_kstep() == 32) && (long)d.D * d.H * d.W * 128 * NT < (1l << 31) ? 32 : 16; }

WsPlan ws_plan(const ConvDims& d, int NT = 3) {
  const int KV = ws_kv(d, NT);
  WsPlan best{};
  double best_cost = 1e30;
  const int KS = d.kd;
  for (int nx = 1; nx <= 12; ++nx) {
    const int Tx = (d.W + nx - 1) / nx;
    for (int Ty = 1; Ty <= 16 && Ty <= d.H; ++Ty) {
      WsPlan pl{};
      pl.Ty = Ty; pl.Tx = Tx;
      pl.XB = (d.W + Tx - 1) / Tx; pl.YB = (d.H + Ty - 1) / Ty;
      pl.Xp = Tx + KS - 1; pl.XU = (Ty + KS - 1) * pl.Xp; pl.XUp = pad_4mod8(pl.XU + KS);  // + KS: tap reads of the clamped tail
      pl.PT = Ty * Tx; pl.NK = (pl.PT + KV - 1) / KV; pl.PTp = pad_4mod8(pl.NK * KV);
      pl.npx = (4 * NT * pl.XUp + 63) / 64; pl.npd = (8 * NT * pl.PTp + 63) / 64;
      pl.xslot = pl.npx * 1024; pl.dybuf = pl.npd * 1024;
      if ((KS == 3 ? 4 : 2) * pl.xslot + 2 * pl.dybuf > kLdsMax) continue;
      if (pl.NK * KV < 48) continue;
      if (KV == 32 && (pl.npx > kWaves * kWP || pl.npd > kWaves * kWP)) continue;  // k_wgrad_s3x keeps a tile's DMA offsets in registers
      // cost per useful position: MFMA time (k-steps incl. padding and tile overhang) + a staging term
      const double useful = (double)d.H * d.W;
      const double mfma = (double)pl.YB * pl.XB * pl.NK * KV;
      const double stage = (double)pl.YB * pl.XB * (4.0 * pl.XUp + 8.0 * pl.PTp) / 24.0;
      const double cost = (mfma + 0.15 * stage) / useful;
      if (cost < best_cost) { best_cost = cost; best = pl; best.ok = true; }
    }
  }
  return best;
}

static thread_local int tl_force3 = 0;
bool s3_layer_h2(const ConvDims& d) {
  if (s3x_get_terms() != 2 || tl_force3) return false;
  static const int use_x = getenv("NC_S3X") ? atoi(getenv("NC_S3X")) : 1;  // (NC_S3X=0 / NC_S3X_WGRAD=0: the round-2 kernels, three-term only)
  if (!use_x) return false;
  if (d.kd != d.kh || d.kd != d.kw || (d.kd != 3 && d.kd != 5) || d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.pd || d.pw != d.pd)
    return false;
  if (d.C % 64 || d.K % 64 || (d.K / 64) * (d.C / 32) * (d.kd == 5 ? 5 : 1) > 256) return false;
  if (!s3x_supported(d.N, d.C, d.D, d.H, d.W, d.K, d.kd) || !s3x_supported(d.N, d.K, d.D, d.H, d.W, d.C, d.kd)) return false;
  return ws_kv(d, 2) == 32 && ws_plan(d, 2).ok;
}

bool ws_shape_ok(const ConvDims& d) {
  if (d.kd != d.kh || d.kd != d.kw || (d.kd != 3 && d.kd != 5)) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.pd || d.pw != d.pd) return false;
  if (d.C % 32 || d.K % 64) return false;
  if ((d.K / 64) * (d.C / 32) * (d.kd == 5 ? 5 : 1) > 256) return false;
  if ((long)d.D * d.H * d.W * 24 >= (1l << 31)) return false;
  return ws_plan(d).ok;
}

int ws_flush_steps() {
  static const int f = 64;
  return f > 0 ? f : 1 << 30;
}
// two-term operands: three MFMA roundings of an accumulator per k-step instead of six -- twice the steps give the same number per partial sum
int ws_flush_steps2() {
  static const int f = 128;
  return f > 0 ? f : 1 << 30;
}
int ws_nf(long steps, int nwp, int F = 0) {  // partial slots per workgroup: ceil(most steps of a workgroup / F)
  if (!F) F = ws_flush_steps();
  const long most = (steps + nwp - 1) / nwp;
  const long nf = (most + F - 1) / F;
  return nf < 1 ? 1 : (int)nf;
}

int ws_nwp(const ConvDims& d, const WsPlan& pl, int npairs) {
  int nwp = 256 / npairs;
  const long steps = (long)d.N * pl.YB * pl.XB * d.D;
  if (nwp > steps) nwp = (int)steps;
  return nwp;
}

// Bytes of the partial-sum area of the weight gradient, two-term (NT = 2) or three-term plan
size_t ws_part_bytes(const ConvDims& d, int NT) {
  const int T3 = d.kd * d.kh * d.kw, TW = d.kd == 3 ? 27 : 25;
  const int npairs = (d.K / 64) * (d.C / 32) * (T3 / TW);
  const WsPlan pl = NT == 2 ? ws_plan(d, 2) : ws_plan(d);
  if (!pl.ok) return 0;
  const int nwp = ws_nwp(d, pl, npairs);
  const long steps = (long)d.N * pl.YB * pl.XB * d.D;
  const int NF = NT == 2 ? ws_nf(steps, nwp, ws_flush_steps2()) : ws_nf(steps, nwp);
  return align256((size_t)npairs * nwp * NF * TW * 64 * 32 * 4);
}

// The weight-gradient kernels proper: operands already in their form (NT = 2: H2 with cells xc / yc; NT = 3: S3), partial sums into `part`
// (>= ws_part_bytes), guard (nullable): the launches leave at once unless the range guard's flag says it is this form's turn.
int ws_core(int NT, const void* xs, const void* dys, const unsigned* xc, const unsigned* yc, float* dw, const ConvDims& d, float* part, hipStream_t s,
            const unsigned* guard) {
  const int KS = d.kd, T3 = KS * KS * KS, TW = KS == 3 ? 27 : 25, NS = KS == 3 ? 4 : 2;
  const WsPlan pl = NT == 2 ? ws_plan(d, 2) : ws_plan(d);
  if (!pl.ok || (NT == 2 && ws_kv(d, 2) != 32)) { set_error("wgrad_s3: shape not covered"); return NC_ERR_SHAPE; }
  const int npairs = (d.K / 64) * (d.C / 32) * (T3 / TW);
  const int nwp = ws_nwp(d, pl, npairs);
  const long steps = (long)d.N * pl.YB * pl.XB * d.D;
  const int F = NT == 2 ? ws_flush_steps2() : ws_flush_steps();
  const int NF = ws_nf(steps, nwp, F);
  WsParams p{};
  p.xs = (const uint4*)xs; p.dys = (const uint4*)dys; p.part = part; p.guard = guard;
  p.zeros = NT == 2 ? nullptr : reinterpret_cast<const uint4*>(nc_zero_page());
  if (NT == 3 && !p.zeros) { set_error("wgrad_s3: no zero page"); return NC_ERR_HIP; }
  p.N = d.N; p.C = d.C; p.K = d.K; p.D = d.D; p.H = d.H; p.W = d.W;
  p.Ty = pl.Ty; p.Tx = pl.Tx; p.YB = pl.YB; p.XB = pl.XB; p.Xp = pl.Xp; p.XU = pl.XU; p.XUp = pl.XUp;
  p.PT = pl.PT; p.PTp = pl.PTp; p.NK = pl.NK; p.npx = pl.npx; p.npd = pl.npd; p.xslot = pl.xslot; p.dybuf = pl.dybuf;
  p.nct = d.C / 32; p.npairs = npairs; p.nwp = nwp; p.steps = steps;
  p.F = F; p.NF = NF;
  p.mTx = magic(pl.Tx); p.mXp = magic(pl.Xp); p.mXUp = magic(pl.XUp); p.mPTp = magic(pl.PTp);
  const int lds = NS * pl.xslot + 2 * pl.dybuf;
  const long total = (long)d.K * d.C * T3;
  if (NT == 2) {
    if (int e = raise_dyn_lds((k_wgrad_s3x<3, 2, NC_DT_F16>), kLdsMax, "wgrad_h2")) return e;
    if (int e = raise_dyn_lds((k_wgrad_s3x<5, 2, NC_DT_F16>), kLdsMax, "wgrad_h2")) return e;
    if (KS == 3) hipLaunchKernelGGL((k_wgrad_s3x<3, 2, NC_DT_F16>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
    else hipLaunchKernelGGL((k_wgrad_s3x<5, 2, NC_DT_F16>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
    if (int e = check_launch("wgrad_h2")) return e;
    hipLaunchKernelGGL(k_wgrad_s3_reduce, dim3((unsigned)cdiv(total, 32)), dim3(256), 0, s, (const float*)part, dw, d.C, T3, TW, d.C / 32, npairs, nwp,
                       NF, total, xc, yc, guard);
    return check_launch("wgrad_h2_reduce");
  }
  if (int e = raise_dyn_lds(k_wgrad_s3<3>, kLdsMax, "wgrad_s3")) return e;
  if (int e = raise_dyn_lds(k_wgrad_s3<5>, kLdsMax, "wgrad_s3")) return e;
  if (int e = raise_dyn_lds((k_wgrad_s3x<3, 3, NC_DT_BF16>), kLdsMax, "wgrad_s3")) return e;
  if (int e = raise_dyn_lds((k_wgrad_s3x<5, 3, NC_DT_BF16>), kLdsMax, "wgrad_s3")) return e;
  if (ws_kv(d) == 32) {
    if (KS == 3) hipLaunchKernelGGL((k_wgrad_s3x<3, 3, NC_DT_BF16>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
    else hipLaunchKernelGGL((k_wgrad_s3x<5, 3, NC_DT_BF16>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
  } else if (KS == 3) hipLaunchKernelGGL(k_wgrad_s3<3>, dim3(npairs * nwp), dim3(kThreads), lds, s, p);
  else hipLaunchKernelGGL(k_wgrad_s3<5>, dim3(npairs * nwp), dim3(kThreads), lds, s, p);
  if (int e = check_launch("wgrad_s3")) return e;
  hipLaunchKernelGGL(k_wgrad_s3_reduce, dim3((unsigned)cdiv(total, 32)), dim3(256), 0, s, (const float*)part, dw, d.C, T3, TW, d.C / 32, npairs, nwp,
                     NF, total, (const unsigned*)nullptr, (const unsigned*)nullptr, guard);
  return check_launch("wgrad_s3_reduce");
}

// The weight gradient on H2 operands (two fp16 terms, three products): k_wgrad_s3x<KS, 2, f16>; xs_pre / dys_pre: H2 tensors with their cells, or
// NULL (converted into the workspace with measured cells).  Range guard (common.hpp): an operand converted here counts its low chunks; when
// the workspace has the room (`dual`), both kernel families are launched and the device-side flag picks one -- the flagged call gets BOTH
// operands in S3 form (a measured one converted again over its H2 form; a pre-converted H2 one from its fp32 source, or from the H2 terms
// themselves, into a region of the workspace).  dy_guard: dys_pre is a workspace region of the caller's whose decision was taken when it
// was converted (conv_bwd_s3): S3 already if its flag is set, and converted to S3 in place if THIS call's x flags it.
int run_ws_h2(const float* x, const void* xs_pre, const float* dy, const void* dys_pre, float* dw, const ConvDims& d, void* ws, size_t wsb,
              hipStream_t s, unsigned* dy_guard) {
  const long S = (long)d.D * d.H * d.W;
  const size_t ex = (size_t)d.N * d.C * S, ey = (size_t)d.N * d.K * S;
  const size_t pb2 = ws_part_bytes(d, 2), pb3 = ws_part_bytes(d, 3);
  if (!pb2) { set_error("wgrad_s3 (two-term): shape not covered"); return NC_ERR_SHAPE; }
  const bool measured = !xs_pre || !dys_pre;
  // dual layout: [X: S3 capacity][Y: S3 capacity, unless dys_pre is the caller's guarded region or ... ][256: guard words][partial sums of either plan]
  const bool y_in_place = dys_pre && dy_guard;               // the caller's region has the S3 capacity; dy (fp32) is there to convert it again
  const bool y_copy = dys_pre && !dy_guard;                  // an H2 tensor we may not overwrite: its S3 form goes into the workspace
  const size_t xb6 = align256(ex * 6), yb6 = y_in_place ? 0 : align256(ey * 6);
  const size_t pbmax = pb2 > pb3 ? pb2 : pb3;
  // (a guarded dY without its fp32 source -- written by the norm backward -- cannot be converted again here: then only ITS decision counts,
  // and an x measured in this call is counted without a switch)
  const bool y_fixed = y_in_place && !dy;
  bool dual = (dy_guard || (measured && h2_guard_can_flip())) && pb3 && wsb >= xb6 + yb6 + 256 + pbmax;
  if (dy_guard && !dual) { set_error("wgrad_s3 (two-term, guarded dY): workspace too small"); return NC_ERR_WS; }
  (void)y_copy;
  const size_t xb = dual ? xb6 : (xs_pre ? 0 : h2_cells_offset(ex) + 256);
  const size_t yb = dual ? yb6 : (dys_pre ? 0 : h2_cells_offset(ey) + 256);
  if (!ws || wsb < xb + yb + 256 + (dual ? pbmax : pb2)) { set_error("wgrad_s3 (two-term): workspace too small"); return NC_ERR_WS; }
  void* X = ws;
  void* Y = (char*)ws + xb;
  unsigned* gw = (unsigned*)((char*)ws + xb + yb);  // words 0..7: guard of x, 8..15: guard of dy, 16..23: the call's flag words
  float* part = (float*)((char*)ws + xb + yb + 256);
  void* xs = xs_pre ? const_cast<void*>(xs_pre) : X;
  void* dys = dys_pre ? const_cast<void*>(dys_pre) : Y;
  unsigned* xc = h2_cells_of(xs, ex);
  unsigned* yc = h2_cells_of(dys, ey);
  unsigned *gx = nullptr, *gy = nullptr, *call = gw + 16;
  if (measured || dy_guard)
    if (int e = h2_guard_zero(gw, s, 24)) return e;
  if (!xs_pre) {
    gx = gw;
    if (int e = h2_zero_cells(xc, 2, s)) return e;
    if (int e = h2_absmax(x, (long)ex, xc, s, xc + 1)) return e;
    if (int e = split2h_into(x, (long)d.C * S, xs, d.N, d.C, S, d.C, 0, xc, s, gx)) return e;
  }
  if (!dys_pre) {
    gy = gw + 8;
    if (int e = h2_zero_cells(yc, 2, s)) return e;
    if (int e = h2_absmax(dy, (long)ey, yc, s, yc + 1)) return e;
    if (int e = split2h_into(dy, (long)d.K * S, dys, d.N, d.K, S, d.K, 0, yc, s, gy)) return e;
  }
  if (measured || dy_guard)
    if (int e = h2_guard_decide(gx, gy, dy_guard, call + kGuardFlag, dual && !y_fixed, s)) return e;
  if (int e = ws_core(2, xs, dys, xc, yc, dw, d, part, s, dual ? call : nullptr)) return e;
  if (!dual) return NC_OK;
  // the flagged call: both operands as S3 tensors
  const void* xs3 = X;
  if (x) { if (int e = split3_into(x, (long)d.C * S, X, d.N, d.C, S, d.C, 0, s, call)) return e; }
  else if (int e = h2_to_s3_if(xs_pre, X, d.N, d.C, S, xc, call, s)) return e;
  const void* ys3 = y_in_place ? dys : Y;
  if (dy) { if (int e = split3_into(dy, (long)d.K * S, const_cast<void*>(ys3), d.N, d.K, S, d.K, 0, s, call)) return e; }
  else if (!y_fixed) { if (int e = h2_to_s3_if(dys_pre, Y, d.N, d.K, S, yc, call, s)) return e; }  // (y_fixed: the flag is dY's own, its S3 form is there)
  return ws_core(3, xs3, ys3, nullptr, nullptr, dw, d, part, s, call);
}

int run_ws(const float* x, const void* xs_pre, const float* dy, const void* dys_pre, float* dw, const ConvDims& d, void* ws, size_t wsb,
           hipStream_t s, unsigned* dy_guard = nullptr) {
  if (s3_layer_h2(d)) return run_ws_h2(x, xs_pre, dy, dys_pre, dw, d, ws, wsb, s, dy_guard);
  const long S = (long)d.D * d.H * d.W;
  const size_t xb = xs_pre ? 0 : align256((size_t)d.N * d.C * S * 6);
  const size_t yb = dys_pre ? 0 : align256((size_t)d.N * d.K * S * 6);
  const size_t pb = ws_part_bytes(d, 3);
  if (!ws || !pb || wsb < xb + yb + pb + 256) { set_error("wgrad_s3: workspace too small"); return NC_ERR_WS; }
  uint4* xs = xs_pre ? (uint4*)xs_pre : (uint4*)ws;
  uint4* dys = dys_pre ? (uint4*)dys_pre : (uint4*)((char*)ws + xb);
  float* part = (float*)((char*)ws + xb + yb);
  if (!xs_pre)
    if (int e = split3_into(x, (long)d.C * S, xs, d.N, d.C, S, d.C, 0, s)) return e;
  if (!dys_pre)
    if (int e = split3_into(dy, (long)d.K * S, dys, d.N, d.K, S, d.K, 0, s)) return e;
  return ws_core(3, xs, dys, nullptr, nullptr, dw, d, part, s, nullptr);
}

// The same kernel on the 16-bit operands of the --precision path (C8 = one term), conv_h.hip's k_wgrad_h replaced: xh / dyh in C8,
// dw fp32; ws: the partial sums only.
bool wsx_shape_ok(const ConvDims& d) {
  if (d.kd != d.kh || d.kd != d.kw || (d.kd != 3 && d.kd != 5)) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.pd || d.pw != d.pd) return false;
  if (d.C % 32 || d.K % 64) return false;
  if ((d.K / 64) * (d.C / 32) * (d.kd == 5 ? 5 : 1) > 256) return false;
  if (ws_kv(d, 1) != 32) return false;
  return ws_plan(d, 1).ok;
}

template <int DT>
int run_wsx(const void* xh, const void* dyh, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  const int KS = d.kd, T3 = KS * KS * KS, TW = KS == 3 ? 27 : 25, NS = KS == 3 ? 4 : 2;
  const WsPlan pl = ws_plan(d, 1);
  const int npairs = (d.K / 64) * (d.C / 32) * (T3 / TW);
  const int nwp = ws_nwp(d, pl, npairs);
  const long steps = (long)d.N * pl.YB * pl.XB * d.D;
  if (!ws || wsb < (size_t)npairs * nwp * TW * 64 * 32 * 4) { set_error("wgrad_c8x: workspace too small"); return NC_ERR_WS; }
  WsParams p{};
  p.xs = (const uint4*)xh; p.dys = (const uint4*)dyh; p.part = (float*)ws; p.zeros = nullptr;
  p.N = d.N; p.C = d.C; p.K = d.K; p.D = d.D; p.H = d.H; p.W = d.W;
  p.Ty = pl.Ty; p.Tx = pl.Tx; p.YB = pl.YB; p.XB = pl.XB; p.Xp = pl.Xp; p.XU = pl.XU; p.XUp = pl.XUp;
  p.PT = pl.PT; p.PTp = pl.PTp; p.NK = pl.NK; p.npx = pl.npx; p.npd = pl.npd; p.xslot = pl.xslot; p.dybuf = pl.dybuf;
  p.nct = d.C / 32; p.npairs = npairs; p.nwp = nwp; p.steps = steps;
  p.F = 1 << 30; p.NF = 1;  // exact 16-bit products, fp32 accumulation: no accumulator restarts
  p.mTx = magic(pl.Tx); p.mXp = magic(pl.Xp); p.mXUp = magic(pl.XUp); p.mPTp = magic(pl.PTp);
  if (int e = raise_dyn_lds((k_wgrad_s3x<3, 1, DT>), kLdsMax, "wgrad_c8x")) return e;
  if (int e = raise_dyn_lds((k_wgrad_s3x<5, 1, DT>), kLdsMax, "wgrad_c8x")) return e;
  const int lds = NS * pl.xslot + 2 * pl.dybuf;
  if (KS == 3) hipLaunchKernelGGL((k_wgrad_s3x<3, 1, DT>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
  else hipLaunchKernelGGL((k_wgrad_s3x<5, 1, DT>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
  if (int e = check_launch("wgrad_c8x")) return e;
  const long total = (long)d.K * d.C * T3;
  hipLaunchKernelGGL(k_wgrad_s3_reduce, dim3((unsigned)cdiv(total, 32)), dim3(256), 0, s, (const float*)ws, dw, d.C, T3, TW, d.C / 32, npairs, nwp, 1,
                     total);
  return check_launch("wgrad_c8x_reduce");
}

}  // namespace

bool c8x_wgrad_supported(const ConvDims& d) {
  static const int on = getenv("NC_HX_WGRAD") ? atoi(getenv("NC_HX_WGRAD")) : 1;  // NC_HX_WGRAD=0: conv_h.hip's k_wgrad_h (A/B)
  return on && wsx_shape_ok(d);
}
size_t c8x_wgrad_part_bytes(const ConvDims& d) {
  const int T3 = d.kd * d.kh * d.kw, TW = d.kd == 3 ? 27 : 25;
  const int npairs = (d.K / 64) * (d.C / 32) * (T3 / TW);
  return (size_t)npairs * ws_nwp(d, ws_plan(d, 1), npairs) * TW * 64 * 32 * 4;
}
int conv_wgrad_c8x(const void* xh, const void* dyh, float* dw, const ConvDims& d, int dtype, void* ws, size_t wsb, hipStream_t s) {
  return dtype == NC_DT_F16 ? run_wsx<NC_DT_F16>(xh, dyh, dw, d, ws, wsb, s) : run_wsx<NC_DT_BF16>(xh, dyh, dw, d, ws, wsb, s);
}

bool s3_fwd_supported(const ConvDims& d) { return s_shape_ok(d, d.C, d.K); }
bool s3_dgrad_supported(const ConvDims& d) { return s_shape_ok(d, d.K, d.C); }
size_t s3_ws_bytes(const ConvDims& d) {
  const long S = (long)d.D * d.H * d.W;
  const int cmax = d.C > d.K ? d.C : d.K;
  const int c64 = (d.C + 63) / 64 * 64, k64 = (d.K + 63) / 64 * 64;  // either may be the 64-multiple "output" side
  size_t pk = align256(s_packed_bytes(c64, k64, d.kd));
  const size_t p3 = align256(s3x_packed_bytes(c64, k64, d.kd, 3));  // (the guarded two-term call packs either form here)
  if (p3 > pk) pk = p3;
  return align256((size_t)d.N * cmax * S * 6) + pk + 768;
}
size_t s3_tensor_bytes(int N, int C, long S) { return (size_t)N * C * S * 6; }
bool s3_wgrad_supported(const ConvDims& d) { return ws_shape_ok(d); }
// [X: S3 capacity | dY: S3 capacity | 256 B guard words | partial sums of the larger of the two plans (two-term / three-term)]
size_t s3_wgrad_ws_bytes(const ConvDims& d) {
  if (!ws_shape_ok(d)) return 0;
  const long S = (long)d.D * d.H * d.W;
  size_t pb = ws_part_bytes(d, 3);
  if (ws_kv(d, 2) == 32 && ws_plan(d, 2).ok) { const size_t p2 = ws_part_bytes(d, 2); if (p2 > pb) pb = p2; }
  return align256((size_t)d.N * d.C * S * 6) + align256((size_t)d.N * d.K * S * 6) + pb + 768;
}
int conv_wgrad_s3(const float* x, const void* xs, const float* dy, const void* dys, float* dw, const ConvDims& d, void* ws, size_t wsb,
                  hipStream_t s) {
  return run_ws(x, xs, dy, dys, dw, d, ws, wsb, s);
}
// The two-term forward kernel on an fp32 input of 32 x n channels (converted here, measured cell, guard counted): deep_linear_gen's
// rank-structured data gradient (gen_nets.hip) -- s3_layer_h2 admits multiples of 64 only
bool conv_fwd_h2_c32_supported(const ConvDims& d) {
  if (s3x_get_terms() != 2 || d.C % 32 || d.K % 64 || d.kd != d.kh || d.kd != d.kw || (d.kd != 3 && d.kd != 5)) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.pd || d.pw != d.pd) return false;
  return s3x_supported(d.N, 64, d.D, d.H, d.W, d.K, d.kd);  // (the planner does not look at the input channels)
}
size_t conv_fwd_h2_c32_ws_bytes(const ConvDims& d) {
  const size_t ex = (size_t)d.N * d.C * d.D * d.H * d.W;
  return align256(ex * 6) + 256 + s3x_packed_bytes(d.C, d.K, d.kd, 3) + 512;
}
int conv_fwd_h2_c32(const float* x, const float* w, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!conv_fwd_h2_c32_supported(d)) { set_error("conv_fwd_h2_c32: shape not covered"); return NC_ERR_SHAPE; }
  const int T3 = d.kd * d.kh * d.kw;
  return run_s3(x, nullptr, w, nullptr, y, d, d.C, d.K, (long)d.C * T3, T3, 0, ws, wsb, s, nullptr, true);
}
// The two-term 5^3 forward onto 32 output channels (k_conv_s3x K32), x converted into xs_keep as conv_fwd_keep does (measured cell, guard
// counted, no fallback): deep_linear_gen's collapsed forward, which wants 27 channels of layer 1's output space and never the 64 (gen_nets.hip)
bool conv_fwd_h2_k32_supported(const ConvDims& d) {
  if (s3x_get_terms() != 2 || d.C % 64 || d.K != 32 || d.kd != 5 || d.kh != 5 || d.kw != 5 || d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != 2 || d.ph != 2 || d.pw != 2)
    return false;
  return s3x_k32_supported(d.N, d.D, d.H, d.W);
}
int conv_fwd_h2_k32_keep(const float* x, const float* w, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s, void* xs_keep) {
  if (!conv_fwd_h2_k32_supported(d) || !xs_keep) { set_error("conv_fwd_h2_k32_keep: shape not covered"); return NC_ERR_SHAPE; }
  return run_s3(x, nullptr, w, nullptr, y, d, d.C, d.K, (long)d.C * 125, 125, 0, ws, wsb, s, xs_keep, true);
}
// The two-term weight-gradient kernel on operands of the caller's choice (x / dy fp32, or xs / dys H2 tensors with their cells): any C % 32,
// K % 64 the plan covers -- not only the layers s3_layer_h2 admits (deep_linear_gen's collapsed backward uses C = 32, gen_nets.hip)
bool wgrad_h2_supported(const ConvDims& d) { return s3x_get_terms() == 2 && ws_shape_ok(d) && ws_kv(d, 2) == 32 && ws_plan(d, 2).ok && ws_part_bytes(d, 2) != 0; }
int conv_wgrad_h2(const float* x, const void* xs, const float* dy, const void* dys, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!wgrad_h2_supported(d)) { set_error("conv_wgrad_h2: shape not covered"); return NC_ERR_SHAPE; }
  return run_ws_h2(x, xs, dy, dys, dw, d, ws, wsb, s, nullptr);
}
// data + weight gradient of one layer with dY converted once: [dY operand: S3 capacity | 256 B: its guard words | scratch of whichever kernel runs]
size_t s3_bwd_ws_bytes(const ConvDims& d) {
  if (!ws_shape_ok(d)) return 0;
  const long S = (long)d.D * d.H * d.W;
  const size_t A = align256((size_t)d.N * d.K * S * 6) + 256;
  size_t dg = 0;
  if (s_shape_ok(d, d.K, d.C)) {  // (the data gradient is optional)
    dg = align256(s_packed_bytes(d.K, d.C, d.kd));
    const size_t p3 = align256(s3x_packed_bytes(d.K, d.C, d.kd, 3));
    if (p3 > dg) dg = p3;
    dg += 768;
  }
  const size_t wg = s3_wgrad_ws_bytes(d) - align256((size_t)d.N * d.K * S * 6);
  return A + (dg > wg ? dg : wg);
}
unsigned* conv_bwd_guard_words(void* ws, int N, int K, long S) { return (unsigned*)((char*)ws + align256((size_t)N * K * S * 6)); }
int conv_bwd_s3(const float* x, const float* dy, const float* w, float* dx, float* dw, const ConvDims& d, void* ws, size_t wsb,
                hipStream_t s, int phase, const void* xs, bool dy_guarded) {  // phase 0: convert dY; 1: data gradient; 2: weight gradient (xs: x in S3, or NULL)
  const long S = (long)d.D * d.H * d.W;
  const size_t A = align256((size_t)d.N * d.K * S * 6) + 256;
  if (!ws || wsb < s3_bwd_ws_bytes(d)) { set_error("conv_bwd_s3: workspace too small"); return NC_ERR_WS; }
  // dy != NULL: dY is converted here (phase 0) with a MEASURED cell -- the range guard applies (common.hpp): its words sit behind the operand
  // region, phase 0 takes the decision and, flagged, writes the S3 form over the H2 one; phases 1 and 2 launch both kernel families.
  // dy == NULL (conv_bwd_pre): the producer (InstanceNorm backward) wrote dY there itself; dy_guarded: it also counted into the guard words,
  // took the decision and, flagged, left the S3 form (norm_act.hip instnorm_act_bwd_dbias_h2).
  unsigned* yg = (unsigned*)((char*)ws + A - 256);
  const bool h2 = s3_layer_h2(d);
  const bool guarded = h2 && h2_guard_can_flip() && (dy || dy_guarded);
  if (phase == 0) {
    if (h2) {  // dY as an H2 tensor (measured cell) where the S3 tensor would stand
      const size_t ey = (size_t)d.N * d.K * S;
      unsigned* yc = h2_cells_of(ws, ey);
      if (int e = h2_guard_zero(yg, s)) return e;
      if (int e = h2_zero_cells(yc, 2, s)) return e;
      if (int e = h2_absmax(dy, (long)ey, yc, s, yc + 1)) return e;
      if (int e = split2h_into(dy, (long)d.K * S, ws, d.N, d.K, S, d.K, 0, yc, s, yg)) return e;
      if (!guarded) return h2_guard_decide(yg, nullptr, nullptr, yg + kGuardFlag, false, s);  // (count only; a no-op with the guard off)
      if (int e = h2_guard_decide(yg, nullptr, nullptr, yg + kGuardFlag, true, s)) return e;
      return split3_into(dy, (long)d.K * S, ws, d.N, d.K, S, d.K, 0, s, yg);
    }
    return split3_to(dy, ws, d.N, d.K, S, s);
  }
  if (phase == 1) {
    ConvDims t = d;
    t.C = d.K; t.K = d.C;
    const int T3 = d.kd * d.kh * d.kw;
    return run_s3(dy, ws, w, nullptr, dx, t, d.K, d.C, T3, (long)d.C * T3, 1, (char*)ws + A, wsb - A, s, nullptr, h2, guarded ? yg : nullptr);
  }
  return run_ws(x, xs, dy, ws, dw, d, (char*)ws + A, wsb - A, s, guarded ? yg : nullptr);
}

int split3_to(const float* x, void* xs, int N, int C, long S, hipStream_t s) { return split3_into(x, (long)C * S, xs, N, C, S, C, 0, s); }

// x: N samples of C channels, `xstride` floats apart; result: channels c0 .. c0 + C - 1 of an S3 tensor with ctot channels
int split3_into(const float* x, long xstride, void* xs, int N, int C, long S, int ctot, int c0, hipStream_t s, const unsigned* guard) {
  if (C % 8 || ctot % 8 || c0 % 8) { set_error("split3: channels must be multiples of 8"); return NC_ERR_SHAPE; }
  long bx = cdiv(S, 256);
  const long ny = (long)N * C / 8;
  if (guard && bx * ny > 2048) bx = cdiv(2048, ny) < bx ? cdiv(2048, ny) : bx;
  hipLaunchKernelGGL(k_split3, dim3((unsigned)bx, (unsigned)ny), dim3(256), 0, s, x, (uint4*)xs, S, C / 8, ctot / 8, c0 / 8,
                     xstride, guard);
  return check_launch("split3");
}

// y (nullable, samples `ystride` floats apart) and the S3 tensor ys (channels c0 .. of ctot) <- act((x - mean) * rstd), x dense [N][C][S]
int act_split3(const float* x, const float* mean, const float* rstd, float slope, float* y, long ystride, void* ys, int N, int C, long S,
               int ctot, int c0, hipStream_t s) {
  if (C % 8 || ctot % 8 || c0 % 8) { set_error("act_split3: channels must be multiples of 8"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_act_split3, dim3((unsigned)cdiv(S, 256), (unsigned)(N * C / 8)), dim3(256), 0, s, x, mean, rstd, slope, y, ystride,
                     (uint4*)ys, S, C / 8, ctot / 8, c0 / 8);
  return check_launch("act_split3");
}

// The operand form of a layer's input, S3 or H2, by s3_layer_h2 of the CONSUMING layer (d).  `into`: channels [c0, c0 + C) of the ctot-channel
// operand from an fp32 tensor (H2: measured cell of that half; a whole tensor sets both cells); `act`: the same from the normalisation pass
// (H2: the cell is the bound sqrt(S) of an InstanceNorm output).
int operand_into(const ConvDims& d, const float* x, long xstride, void* xs, int N, int C, long S, int ctot, int c0, hipStream_t s) {
  if (!s3_layer_h2(d)) return split3_into(x, xstride, xs, N, C, S, ctot, c0, s);
  if (C != ctot && 2 * C != ctot) { set_error("operand_into: a part must be the whole tensor or a half"); return NC_ERR_SHAPE; }
  unsigned* cells = h2_cells_of(xs, (size_t)N * ctot * S);
  unsigned* mine = cells + (c0 ? 1 : 0);
  if (int e = h2_zero_cells(mine, C == ctot ? 2 : 1, s)) return e;
  for (int n = 0; n < N; ++n)
    if (int e = h2_absmax(x + (long)n * xstride, (long)C * S, mine, s, C == ctot ? cells + 1 : nullptr)) return e;
  // range guard, COUNT ONLY: the tensor lives in the caller's buffer and its consumers are launched elsewhere, so a flagged tensor cannot
  // change kernels here -- it is reported (nc_h2_guard_stats: [2]); guard words in the spare words of the 256-byte cells block
  unsigned* g = cells + (c0 ? 16 : 8);
  if (int e = h2_guard_zero(g, s)) return e;
  if (int e = split2h_into(x, xstride, xs, N, C, S, ctot, c0, mine, s, g)) return e;
  return h2_guard_decide(g, nullptr, nullptr, g + kGuardFlag, false, s);
}
int act_operand(const ConvDims& d, const float* x, const float* mean, const float* rstd, float slope, float* y, long ystride, void* ys, int N, int C,
                long S, int ctot, int c0, hipStream_t s) {
  if (!s3_layer_h2(d)) return act_split3(x, mean, rstd, slope, y, ystride, ys, N, C, S, ctot, c0, s);
  if (C != ctot && 2 * C != ctot) { set_error("act_operand: a part must be the whole tensor or a half"); return NC_ERR_SHAPE; }
  unsigned* cells = h2_cells_of(ys, (size_t)N * ctot * S);
  unsigned* mine = cells + (c0 ? 1 : 0);
  return act_split2h(x, mean, rstd, slope, y, ystride, ys, N, C, S, ctot, c0, sqrtf((float)S), mine, C == ctot ? cells + 1 : nullptr, s);
}
bool conv_layer_h2(const ConvDims& d) { return s3_layer_h2(d); }
static thread_local int tl_net_depth = 0;
static thread_local int tl_sw_depth = 0, tl_sw_terms = -1, tl_sw_guard = -1;
SwitchScope::SwitchScope() {
  if (tl_sw_depth++ == 0) { tl_sw_terms = s3x_get_terms(); tl_sw_guard = h2_guard_mode(); }
}
SwitchScope::~SwitchScope() {
  if (--tl_sw_depth == 0) { tl_sw_terms = -1; tl_sw_guard = -1; }
}
ForceTwoTerm::ForceTwoTerm() : prev_terms(tl_sw_terms), prev_depth(tl_sw_depth) { tl_sw_terms = 2; ++tl_sw_depth; }
ForceTwoTerm::~ForceTwoTerm() { tl_sw_terms = prev_terms; --tl_sw_depth; if (tl_sw_depth == 0) { tl_sw_terms = -1; tl_sw_guard = -1; } }
int frozen_terms() { return tl_sw_terms; }
int frozen_guard() { return tl_sw_guard; }
NetworkScope::NetworkScope() { ++tl_net_depth; if (tl_sw_depth++ == 0) { tl_sw_terms = s3x_get_terms(); tl_sw_guard = h2_guard_mode(); } }
NetworkScope::~NetworkScope() { --tl_net_depth; if (--tl_sw_depth == 0) { tl_sw_terms = -1; tl_sw_guard = -1; } }
bool h2_guard_can_flip() { return h2_guard_mode() == 2 || (h2_guard_mode() == 1 && tl_net_depth == 0); }
ForceThreeTerm::ForceThreeTerm() { ++tl_force3; }
ForceThreeTerm::~ForceThreeTerm() { --tl_force3; }

int conv_fwd_s3(const float* x, const void* xs, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                hipStream_t s, void* xs_keep, float* stats_part) {
  const int T3 = d.kd * d.kh * d.kw;
  return run_s3(x, xs, w, b, y, d, d.C, d.K, (long)d.C * T3, T3, 0, ws, wsb, s, xs_keep, s3_layer_h2(d), nullptr, stats_part);
}

int conv_dgrad_s3(const float* dy, const void* dys, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  // dx = conv(dy, flipped / channel-transposed w): "input" channels K, "output" channels C
  ConvDims t = d;
  t.C = d.K; t.K = d.C;
  const int T3 = d.kd * d.kh * d.kw;
  return run_s3(dy, dys, w, nullptr, dx, t, d.K, d.C, T3, (long)d.C * T3, 1, ws, wsb, s, nullptr, s3_layer_h2(d));
}

}  // namespace nc
