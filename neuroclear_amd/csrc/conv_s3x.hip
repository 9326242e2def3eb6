// fp32 Conv3d 3^3 / 5^3 (stride 1, "same" padding), forward and data gradient, on the bf16 matrix cores with the exact
// three-term operand split of conv_split.hip (same S3 input layout, same six products per fp32 product) -- the "tap stream"
// form of that kernel (models/networks.py:420-425, 460-469, 900-902: the nn.Conv3d layers of unet_deconv / deep_linear_gen):
//
//  * v_mfma_f32_16x16x32_bf16 instead of 32x32x16: at equal work the chip holds a ~25 % higher clock under it (bare loops on
//    random data: 2.01 against 1.74 PFLOP/s, tools/mfma_rate.hip) -- dense bf16 MFMA loops are power-bound on this part;
//  * NO zero tap: the K-dim of a tile is ONE stream of taps, (8-channel chunk, dz, dy, dx) in that order, consumed four at
//    a time (a k-step of 32 = lane group g = lane / 16 takes tap 4s + g, 8 channels each).  KS^2 taps (9 or 25) of one
//    input plane of one chunk are a "brick"; k-steps straddle bricks, so the bricks live in an LDS RING of three
//    (in use / arrived / loading) filled by LDS-DMA, one barrier per brick.  conv_split.hip's pairing spent 10 % of the
//    3^3 MFMAs on a zero-weight tenth tap;
//  * a brick is the flat range [q0, q0 + PT + (KS-1)(P+1)) of the zero-padded plane (pitch P = W + KS - 1), not whole rows:
//    37-40 KB for 512 positions and three terms, so three of them fit and NO weights go through LDS;
//  * weights stream from global memory (L2-resident) straight into registers as MFMA A fragments, one k-step ahead, through
//    a buffer descriptor with scalar offsets: [cot][co half][k-step][row block][term][lane][8] bf16, 1 KiB per fragment;
//  * 8 waves = 2 halves of the 64 output channels x 4 groups of NCB*16 positions; per k-step a wave holds its 6 A fragments
//    and walks its NCB column blocks: 6 ds_read_b64 (3 terms) + 12 MFMAs each.  The B fragment of a lane is one 16-byte unit
//    read as two ds_read_b64 -- lane groups 1 and 3 read the upper half first (their packed weights have the channel halves
//    swapped to match): every read instruction touches each of the 64 banks exactly once whatever the tap offsets of the
//    groups are (one ds_read_b128 would be 2-way conflicted for every pair of taps that is not a multiple of 16 units apart);
//  * tile quantisation: the launch covers whole rounds of 256 tiles (NCB = 8: 512 positions, 7: 448, 6: 384 -- the planner picks per
//    plane size; 54^2 / 27^2 / 35^2 planes quantise best at 448) and the remainder with a second launch of half / third / quarter
//    tiles (NCB = 4 / 2) when those fit one round, else as whole tiles in the same launch -- 2,592 tiles at 108^3 cost 10.3 rounds, not 11;
//  * the LDS-DMA pieces of a brick are issued by ONE wave of each SIMD pair (waves 0..3), source offsets computed at issue time: its
//    partner keeps the matrix pipe busy meanwhile (-4 %); the stores of a tile are issued inside the first four k-steps of the next.
// Tried and dropped: output positions as ONE flat axis per sample (q = (z Hp + y) P + x over the padded volume, 512-position tiles at
// every plane size, z-neighbours enumerated side by side).  In isolation the 27^3 layers gained 25 %; in the step it made no difference
// and the 900^3 inference ran 2.3 % slower (pad rows computed and dropped; same-box A/B) -- tiles stay inside one output plane.
// Accuracy: as in conv_split.hip the MFMA accumulators restart every `flush` k-steps and the pieces are added in fp32.
//
// Round 4: template parameter NT.  NT = 3 is the kernel above.  NT = 2 is the same kernel on the TWO-TERM fp16 form of the operands ("H2",
// s3_common.hpp / h2.hip): each operand = two fp16 terms of the tensor times a per-tensor power of two, THREE products a1 b0, a0 b1, a0 b0 per
// fp32 product on v_mfma_f32_16x16x32_f16, the sums scaled back (exactly) on their way out.  Four A fragments and two B fragments per k-step
// instead of six and three, bricks of two terms; everything else unchanged.  64 -> 64 at 108^3: 3^3 1.14 -> 0.67 ms, 5^3 4.67 -> 2.63 ms,
// error against fp64 as the three-term form's (tests/test_gpu_h2.py).  A forward input may be a concatenation whose halves were converted
// with different powers of two: the ratio is folded into the weights of the second half's input channels when they are packed.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "s3_common.hpp"

namespace nc {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const volatile __attribute__((address_space(3))) unsigned long long* lds64_t;

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kLdsMax = 160 * 1024;
#ifndef NC_S3X_DMAW
#define NC_S3X_DMAW 4
#endif
#ifndef NC_S3X_BDEPTH
#define NC_S3X_BDEPTH 1  // column blocks the B-fragment reads run ahead of the MFMAs that use them (experiment: 2)
#endif
// Waves that issue the LDS-DMA pieces of a brick: the first kDmaWaves of the 8.  4 = one wave of each SIMD pair (waves w and w + 4
// share a SIMD): issuing ~10 pieces keeps a wave away from the matrix pipe for ~1,000 cycles, which its partner -- with no pieces of
// its own -- fills with MFMAs; with all 8 issuing, both partners are away at the same moment right behind the barrier.
constexpr int kDmaWaves = NC_S3X_DMAW;
constexpr int kMaxPieces = 56;  // 1 KiB pieces of a brick (three terms)

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m) { return __umulhi(n, m); }
unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

// (the split itself: s3_common.hpp)
__device__ __forceinline__ void split3(float v, unsigned short (&t)[3]) { s3_split(v, t); }

// Packed weights: [cot = co/64][half = (co/32)%2][k-step s][f = rb*3 + term][lane][8] bf16.  Lane l = (g = l/16, m = l%16) holds
// output channel cot*64 + half*32 + rb*16 + m at tap T = 4s + g of the tile's tap stream: brick T / KS^2 = chunk*KS + dz,
// in-plane tap T % KS^2; element j = input channel chunk*8 + (g odd ? (j + 4) % 8 : j)  (the B fragment of an odd lane
// group is read upper half first).
// fwd:   w[co][ci][tap]                   (so = C*T3, si = T3, flip = 0)
// dgrad: w[co as "ci"][ci as "co"][T3-1-tap]  (so = T3, si = C*T3, flip = 1)
// NT = 2: the two fp16 terms of w * 2^k (h2_split), k from the tensor's absmax cell.
template <int NT>
__global__ void __launch_bounds__(256) k_pack_w_s3x(const float* __restrict__ w, unsigned short* __restrict__ wp, int NCH, int KS, int NS,
                                                    long so, long si, int flip, long total, const unsigned* __restrict__ amax, int split_c,
                                                    const unsigned* __restrict__ cell_a, const unsigned* __restrict__ cell_b) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int T2 = KS * KS, T3 = T2 * KS, NB = NCH * KS;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int lane = (int)(q & 63); q >>= 6;
  const int f = (int)(q % (2 * NT)); q /= 2 * NT;
  const int s = (int)(q % NS); q /= NS;
  const int half = (int)(q & 1);
  const int cot = (int)(q >> 1);
  const int rb = f / NT, term = f % NT;
  const int g = lane >> 4, m = lane & 15;
  const int T = 4 * s + g;
  const int bi = T / T2, tp = T % T2;
  unsigned short t[3] = {0, 0, 0};
  if (bi < NB) {
    const int chunk = bi / KS, dz = bi % KS;
#ifdef NC_S3X_B128
    const int jj = j;  // (experiment: one 16-byte read per B fragment, natural channel order)
#else
    const int jj = (g & 1) ? ((j + 4) & 7) : j;
#endif
    const long co = cot * 64 + half * 32 + rb * 16 + m, ci = chunk * 8 + jj;
    const int tap = dz * T2 + tp;
    const float v = w[co * so + ci * si + (flip ? T3 - 1 - tap : tap)];
    if constexpr (NT == 3) split3(v, t);
    else h2_split(v * (ci >= split_c ? h2_group_factor(*cell_a, *cell_b) : 1.f) * h2_scale(*amax), t);
  }
  wp[i] = t[term];
}

// Largest finite |w'| of the weights as the pack sees them: w' = w * 2^(kA - kB) for the input channels of the second scale group (a
// concatenation whose halves were converted with different powers of two: folding the ratio into the weights makes the sum uniform in 2^kA).
__global__ void __launch_bounds__(256) k_absmax_w(const float* __restrict__ w, long n, int T3, int C, int split_c, const unsigned* __restrict__ cell_a,
                                                  const unsigned* __restrict__ cell_b, unsigned* __restrict__ out) {
  unsigned m = 0;
  const float gf = split_c < C ? h2_group_factor(*cell_a, *cell_b) : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ci = (int)((i / T3) % C);
    const unsigned b = __float_as_uint(w[i] * (ci >= split_c ? gf : 1.f)) & 0x7fffffffu;
    if (b < 0x7f800000u && b > m) m = b;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned q = (unsigned)__shfl_xor((int)m, o);
    m = q > m ? q : m;
  }
  __shared__ unsigned wm[4];  // one atomic per block (h2.hip k_absmax)
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned b = wm[0];
    for (int k = 1; k < 4; ++k) b = wm[k] > b ? wm[k] : b;
    if (b) atomicMax(out, b);
  }
}

// ---- the pseudo-channel form of a one-input-channel 7^3 layer (k_conv_s3x PC)
// Packed weights, same fragment layout as k_pack_w_s3x with ONE 8-"channel" block: lane (g, m) of k-step s holds output channel co at tap
// T = 4 s + g of the (dz, dy) stream (brick dz = T / 8, in-plane tap dy = T % 8, the eighth slot empty); element j = the dx tap j (odd lane
// groups: (j + 4) % 8), zero for j = 7.  flip: the data-gradient orientation (w[co][0][342 - tap]) -- not used yet.
__global__ void __launch_bounds__(256) k_pack_w_c1k7(const float* __restrict__ w, unsigned short* __restrict__ wp, int NS, long total,
                                                     const unsigned* __restrict__ amax) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int lane = (int)(q & 63); q >>= 6;
  const int f = (int)(q & 3); q >>= 2;  // rb * 2 + term
  const int s = (int)(q % NS); q /= NS;
  const int half = (int)(q & 1);
  const int rb = f >> 1, term = f & 1;
  const int g = lane >> 4, m = lane & 15;
  const int T = 4 * s + g;
  unsigned short t[3] = {0, 0, 0};
  const int jj = (g & 1) ? ((j + 4) & 7) : j;
  const int dz = T >> 3, dy = T & 7;  // eight tap slots per (dz) brick, the eighth empty (see k_conv_s3x PC)
  if (dz < 7 && dy < 7 && jj < 7) {
    const int co = half * 32 + rb * 16 + m;
    h2_split(w[(long)co * 343 + (dz * 7 + dy) * 7 + jj] * h2_scale(*amax), t);
  }
  wp[i] = t[term];
}

// x fp32 [N][1][D][H][W] -> the H2 tensor [N][1 block][2 terms][voxels][8]: element j of voxel (z, y, x) = x[z][y][x + j - 3] (zero outside the
// row and for j = 7), times the power of two of the measured cell.  guard (nullable): the range guard's chunk counts (h2.hip k_split2h: a wave's
// 64 voxels are one chunk, judged by the largest magnitude of their OWN elements).
__global__ void __launch_bounds__(256) k_build_x8_h2(const float* __restrict__ x, uint4* __restrict__ out, long S, int W, const unsigned* __restrict__ cell,
                                                     unsigned* __restrict__ guard) {
  const unsigned cb_bits = *cell;
  const float sc = h2_scale(cb_bits);
  const int n = blockIdx.y;
  unsigned n_all = 0, n_low = 0;
  for (long v0 = (long)blockIdx.x * 256; v0 < S; v0 += (long)gridDim.x * 256) {
    const long v = v0 + threadIdx.x;
    unsigned m = 0;
    if (v < S) {
      const int xx = (int)(v % W);
      const float* row = x + (long)n * S + (v - xx);
      unsigned short e[8][3];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int xs = xx + j - 3;
        const float f = (j < 7 && (unsigned)xs < (unsigned)W) ? row[xs] : 0.f;
        if (j == 3) { const unsigned b = __float_as_uint(f) & 0x7fffffffu; if (b < 0x7f800000u) m = b; }
        h2_split(f * sc, e[j]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) out[((long)n * 2 + t) * S + v] = s3_unit(e, t);
    }
    if (guard) {
      const unsigned thr = cb_bits > kGuardDrop ? cb_bits - kGuardDrop : 0u;
      const bool any_hi = __builtin_amdgcn_ballot_w64(m >= thr && m != 0) != 0;
      const bool any_nz = __builtin_amdgcn_ballot_w64(m != 0) != 0;
      if (any_nz) {
        ++n_all;
        if (!any_hi) ++n_low;
      }
    }
  }
  if (!guard) return;
  __shared__ unsigned cnt[2];
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    if (n_all) atomicAdd(&cnt[kGuardAll], n_all);
    if (n_low) atomicAdd(&cnt[kGuardLow], n_low);
  }
  __syncthreads();
  if (threadIdx.x < 2 && cnt[threadIdx.x]) atomicAdd(guard + threadIdx.x, cnt[threadIdx.x]);
}

// PC = 2 (the data gradient): row co = dz * 7 + dx of the 64-row tile (rows 49 .. 63 empty), lane (g, m) of k-step s at slot T = 4 s + g: brick
// T / 8 = the 8-channel block of dY, in-plane slot tau = T % 8 reads dY's row y + tau - 3, which meets the kernel row dy = 6 - tau (dX[u] = sum w[k][t]
// dY[k][u + 3 - t]); element j = channel block * 8 + j (odd lane groups: (j + 4) % 8).
__global__ void __launch_bounds__(256) k_pack_w_c1k7d(const float* __restrict__ w, unsigned short* __restrict__ wp, int NS, long total,
                                                      const unsigned* __restrict__ amax) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int lane = (int)(q & 63); q >>= 6;
  const int f = (int)(q & 3); q >>= 2;
  const int s = (int)(q % NS); q /= NS;
  const int half = (int)(q & 1);
  const int rb = f >> 1, term = f & 1;
  const int g = lane >> 4, m = lane & 15;
  const int T = 4 * s + g;
  const int blk = T >> 3, tau = T & 7;
  const int co = half * 32 + rb * 16 + m;
  const int jj = (g & 1) ? ((j + 4) & 7) : j;
  unsigned short t[3] = {0, 0, 0};
  if (blk < 8 && tau < 7 && co < 49) {
    const int k = blk * 8 + jj, dz = co / 7, dx = co - dz * 7, dy = 6 - tau;
    h2_split(w[(long)k * 343 + (dz * 7 + dy) * 7 + dx] * h2_scale(*amax), t);
  }
  wp[i] = t[term];
}

// dX[n][z][y][x] = sum over (dz, dx) of Z[n][dz * 7 + dx][z + 3 - dz][y][x + 3 - dx] where that voxel exists, in (dz, dx) order: every element of Z
// is read once (247 MB at 108^3)
__global__ void __launch_bounds__(256) k_fold_c1k7(const float* __restrict__ Z, float* __restrict__ dx, int D, int H, int W) {
  const long HW = (long)H * W, S = (long)D * HW;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const int n = blockIdx.y;
  const int x = (int)(v % W), z = (int)(v / HW);
  const float* Zn = Z + (long)n * 49 * S;
  float acc = 0.f;
#pragma unroll
  for (int dz = 0; dz < 7; ++dz) {
    const int zz = z + 3 - dz;
    if ((unsigned)zz >= (unsigned)D) continue;
    const float* zp = Zn + (long)dz * 7 * S + v + (long)(3 - dz) * HW;
#pragma unroll
    for (int d = 0; d < 7; ++d) {
      const int xx = x + 3 - d;
      if ((unsigned)xx < (unsigned)W) acc += zp[(long)d * S + 3 - d];
    }
  }
  dx[(long)n * S + v] = acc;
}

struct XParams {
  const uint4* xs;     // S3 input [N][C/8][3][D][H][W] units
  const uint4* wp;     // packed weights
  const float* bias;   // nullable
  float* y;            // fp32 NCDHW output
  int N, NCH, D, H, W, K;
  int P;               // row pitch of the padded plane, W + KS - 1
  int HP;              // H * P: flattened output positions of a plane (pad columns included)
  int TPP, KT;         // main-tiling tiles per plane, K / 64
  int fsub;            // this launch's tiles are 1 / fsub of a main tile (tile index = main index * fsub + sub)
  int UB;              // units per term of a brick (multiple of 64)
  int npb;             // 1 KiB pieces per brick (three terms)
  int NS;              // k-steps per tile
  unsigned mP, mUB;
  int t_begin, t_count;  // first main tile and number of (sub-)tiles of this launch
  int tiles_per_xcd;
  int flush;           // k-steps between two accumulator restarts
  const unsigned *amax_x, *amax_w;  // NT = 2: the cells of the input and of the weights (the result is scaled back by 2^-(kx + kw))
#ifdef NC_S3X_STAMP
  long long* dbg;      // timing-experiment builds only (tools/s3x_variant.sh stamp -DNC_S3X_STAMP): s_memtime stamps of workgroup 0 / wave 0
#endif
  const unsigned* guard;  // nullable: the range guard's words (common.hpp); the kernel leaves at once unless the flag says it is this form's turn
  float2* stats;       // ST launches: [tile of this launch][wave][32 channels] (sum, sum of squares) of the tile's bias-free outputs (see s3x_stats_finalize)
};

struct XTile {
  int n, cot, z, q0;
};

// Tile order: output-channel tile fastest, then GROUPS of kXGroup in-plane neighbours, then z, then the groups of a plane, then samples.
// The 32 workgroups of an XCD work on consecutive tiles at any one time: with the round-3 order (z right behind the channel tile) those were
// 32 planes of ONE in-plane tile -- the z-halo hit L2, the in-plane halo (1.44 x a tile's units at 3^3, 1.88 x at 5^3) was fetched again
// 100+ tiles later; now 4 in-plane neighbours x 8 planes share both (the order conv_c8x.hip / conv_h.hip took in this round).
constexpr int kXGroup = 4;
template <int PT>
__device__ __forceinline__ XTile x_decode(const XParams& p, int idx) {
  int t = p.t_begin + idx / p.fsub;
  const int sub = idx % p.fsub;
  XTile o;
  o.cot = t % p.KT; t /= p.KT;
  const int per_n = p.TPP * p.D;
  o.n = t / per_n;
  const int u = t - o.n * per_n;
  const int full = (p.TPP / kXGroup) * kXGroup * p.D;  // tiles in whole groups
  int tp;
  if (u < full) {
    const int grp = u / (kXGroup * p.D), rem = u - grp * (kXGroup * p.D);
    o.z = rem / kXGroup;
    tp = grp * kXGroup + (rem - o.z * kXGroup);
  } else {
    const int L = p.TPP % kXGroup, v = u - full;  // the last, narrower group
    o.z = v / L;
    tp = (p.TPP / kXGroup) * kXGroup + (v - o.z * L);
  }
  o.q0 = (tp * p.fsub + sub) * PT;
  // wave-uniform by construction; said explicitly so that descriptors and scalar offsets built from them stay in SGPRs
  o.cot = __builtin_amdgcn_readfirstlane(o.cot); o.z = __builtin_amdgcn_readfirstlane(o.z);
  o.n = __builtin_amdgcn_readfirstlane(o.n); o.q0 = __builtin_amdgcn_readfirstlane(o.q0);
  return o;
}

// ST (two-term launches of the inference forward): the epilogue also leaves, per (tile, wave), the sum and the sum of squares of the wave's
// 32 channels over the tile's valid positions -- of the outputs WITHOUT their bias (the variance does not see it, the mean gets it back;
// sums of bias-free convolution outputs cancel far less) -- so that InstanceNorm needs no pass of its own over the raw output
// (k_in_stats read 2.9 GB per 140^3 cube: 0.6 of 11.1 ms).  fp32 within a tile (128 values per channel: 8 per lane, then a butterfly over the
// 16 lanes of a column block row), fp64 across tiles in s3x_stats_finalize (fixed order: deterministic).
// K32: ONE 32-channel output slice per tile instead of two -- the eight waves are eight position groups of NCB column blocks (a tile of 128 NCB
// positions) and all read the weights of "half 0".  For convolutions onto 32 output channels (deep_linear_gen's collapsed forward, gen_nets.hip):
// on the 64-channel tile half of the MFMAs would multiply zero weights.
// PC (round 6, KS = 7): the PSEUDO-CHANNEL form of a one-input-channel 7^3 layer (deep_linear_gen's first layer, networks.py:899): the input is
// an H2 tensor of ONE 8-channel block whose "channels" are the voxel's row shifted by -3 .. +4 (the seven dx taps and a zero: k_build_x8_h2),
// so the kernel's taps are (dz, dy) only -- 7 bricks of 7 in-plane taps (+ one empty slot: two k-steps per brick), row pitch W (no pad columns),
// 14 k-steps.  Same tap stream, same ring, same epilogue: the 55 GFLOP of the layer run at the two-term rate instead of on the fp32
// matrix instruction (0.67 ms at 108^3).
// PC = 2: the same in-plane geometry (seven row taps + an empty slot per brick, pitch W) with NO plane taps -- a brick per 8-channel block: the
// data gradient of that layer as Z[(dz, dx)][v] = sum_k sum_dy w[k][dz][dy][dx] dY[k][v + (3 - dy) W] on a 64-channel tile (49 rows used),
// folded into dX by k_fold_c1k7 (conv_c1k7_h2_dgrad below).
template <int KS, int NCB, int NT, bool ST = false, bool K32 = false, int PC = 0>
__global__ void __launch_bounds__(kThreads, 1) k_conv_s3x(const XParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  // (PC: EIGHT tap slots per brick, the eighth a zero-weight dummy that re-reads row 6 -- a brick is then exactly two k-steps: with seven, brick 1
  // would be wanted by the read-ahead at the end of k-step 0, one step before its arrival barrier)
  constexpr int PAD = KS / 2, KX = PC ? 1 : KS, PADX = PC ? 0 : PAD, T2 = PC ? 8 : KS * KX, PT = (K32 ? 128 : 64) * NCB;
  constexpr int KZ = PC == 2 ? 1 : KS, PADZ = PC == 2 ? 0 : PAD;  // plane taps
  static_assert(!(K32 && ST), "no epilogue statistics on 32-channel tiles");
  static_assert(!PC || (KS == 7 && NT == 2 && !ST && !K32), "the pseudo-channel form: 7 x 7 x 1 taps, two-term operands");
  if (guard_skip(p.guard, NT == 3)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  const int half = K32 ? 0 : wave & 1, pg = K32 ? wave : wave >> 1;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int NB = p.NCH * KZ;     // bricks per tile
  const int BB = p.npb * 1024;   // bytes per ring slot

  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int t_lo = xcd * p.tiles_per_xcd;
  int t_hi = t_lo + p.tiles_per_xcd;
  if (t_hi > p.t_count) t_hi = p.t_count;
  // the (sub-)tiles of this workgroup: t_lo + wslot, + nslot, ...; sub-tiles that start beyond the plane are empty
  auto next_tile = [&](int t, XTile& o) __attribute__((always_inline)) {
    for (; t < t_hi; t += nslot) {
      o = x_decode<PT>(p, t);
      if (o.q0 < p.HP) return t;
    }
    return -1;
  };
  XTile cur, nxt;
  int tcur = next_tile(t_lo + wslot, cur);
  if (tcur < 0) return;

  // ---- brick staging: unit u of a term = padded flat position q0 + u -> (row, column) of the padded plane.  The DMA goes
  // through a buffer descriptor over the three terms of ONE 8-channel block: a lane whose unit is padding asks for an offset
  // beyond the descriptor's range and the hardware delivers zeros (no zero page, no 64-bit address arithmetic per lane)
  constexpr unsigned kOut = 0x80000000u;
  // Two-term form: the per-lane source offsets of a tile's pieces depend on the tile only (chunk and plane are the descriptor and the scalar
  // offset).  An issuing wave writes its <= kPW of them into an LDS table behind the ring once per tile and reads them back (one ds_read per
  // piece) for the tile's KS * C/8 - 1 later bricks, instead of ~12 vector instructions per piece and brick: with half the MFMAs per brick the
  // recomputation is no longer hidden (a build without the DMA is 17 % faster).  (Registers would do -- but an array captured by the lambdas
  // below makes hipcc drop the kernel's host-side handle, and the three-term kernel has none left.)
  constexpr int kPW = 8;
  const bool keep = NT == 2 && p.npb <= kDmaWaves * kPW;
  typedef volatile __attribute__((address_space(3))) unsigned* lds32_t;
  const unsigned otab = (unsigned)(unsigned long long)(lptr_t)lds_raw + (unsigned)(3 * BB + ((wave * kPW) * 64 + lane) * 4);  // [wave][piece i][lane]
  auto issue_brick = [&](const XTile& t, int bi, int slot, bool kept) __attribute__((always_inline)) {
    if (wave >= kDmaWaves) return;
    const int chunk = bi / KZ, dz = bi - chunk * KZ;
    const int zz = t.z + dz - PADZ;
    const bool zok = (unsigned)zz < (unsigned)p.D;
    const uint4* blk = p.xs + ((long)t.n * p.NCH + chunk) * NT * S;
    // a plane outside the volume: an empty descriptor, every lane reads zeros
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(blk), 0, zok ? (unsigned)(NT * S * 16) : 0u, 0x00020000);
    const int soff = zok ? (int)(zz * HW * 16) : 0;
    unsigned char* buf = lds_raw + slot * BB;
    if (kept) {
      unsigned po[kPW];  // all table reads in flight before the first request (one at a time each waited a full LDS round trip)
#pragma unroll
      for (int i = 0; i < kPW; ++i) po[i] = *(lds32_t)(otab + i * 256);
#pragma unroll
      for (int i = 0; i < kPW; ++i) {
        const int pc = wave + kDmaWaves * i;
        if (pc < p.npb) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(buf + pc * 1024), 16, po[i], soff, 0, 0);
      }
      return;
    }
#pragma unroll 1
    for (int pc = wave; pc < p.npb; pc += kDmaWaves) {
      // per-lane byte offset of the unit this lane fetches for piece pc (inside the block's three terms, relative to plane 0), or kOut:
      // computed at issue time by the issuing waves (~12 vector instructions per piece, in the shadow of the SIMD partner's MFMAs)
      // rather than kept in registers per tile.  (Written in line: as a lambda called from this lambda it made hipcc drop the
      // kernel's host-side handle.)
      const unsigned u = (unsigned)(pc * 64 + lane);
      const unsigned term = fdiv(u, p.mUB);
      const unsigned F = (unsigned)t.q0 + (u - term * p.UB);
      const unsigned rr = fdiv(F, p.mP);
      const int xx = (int)(F - rr * p.P) - PADX;
      const int yy = (int)rr - PAD;
      const bool ok = term < (unsigned)NT && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const unsigned po = ok ? (unsigned)(term * (unsigned)S + (unsigned)(yy * p.W + xx)) * 16u : kOut;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(buf + pc * 1024), 16, po, soff, 0, 0);
    }
  };

  // ---- weights: A fragments through a buffer descriptor, scalar offset per (tile, k-step, fragment).  The loads are inline
  // assembly on purpose: the compiler's own vmcnt bookkeeping merges the paths with and without a brick request conservatively
  // and would wait for freshly issued LDS-DMA in front of every k-step; here every wait is placed by hand (wait_a, arrival).
  u32x4 wrsrc;
  {
    const unsigned long long wa = (unsigned long long)p.wp;
    wrsrc.x = __builtin_amdgcn_readfirstlane((unsigned)wa);
    wrsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
    wrsrc.z = __builtin_amdgcn_readfirstlane(0x7fffffffu);
    wrsrc.w = __builtin_amdgcn_readfirstlane(0x00020000u);
  }
  const int wvoff = lane * 16;
  auto wtile = [&](int cot) __attribute__((always_inline)) { return ((K32 ? cot : cot * 2 + half) * p.NS) * (2 * NT * 1024); };
  auto load_a = [&](u32x4 (&A)[2][NT], int soff) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"
                     : "=v"(A[rb][t]) : "v"(wvoff), "s"(wrsrc), "s"(__builtin_amdgcn_readfirstlane(soff) + (rb * NT + t) * 1024) : "memory");
  };
  // All vector-memory operations of this wave but its 6 youngest (the A fragments requested last) are complete -- `stored`: but
  // the 6 and the stores of the previous tile issued by the step before (2 * NCB; odd NCB: fewer in the fourth step).  The count is chosen by a scalar branch around
  // bare s_waitcnt instructions; ONE statement behind the branch ties the fragment registers to the wait (a tie inside either arm
  // makes the compiler copy the still-in-flight registers in front of the wait).
  auto wait_a = [&](u32x4 (&A)[2][NT], auto nst, bool stored) {  // nst: stores the step before issued when it stored (compile time)
    if (stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NT + decltype(nst)::value) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NT) : "memory");
    if constexpr (NT == 3) asm volatile("" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2])::"memory");
    else asm volatile("" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1])::"memory");
  };

  // ---- B fragments: unit (slot, term, position + tap) of the ring, read as two 8-byte halves (odd lane groups: upper first)
#ifdef NC_S3X_B128
  const unsigned lane_b = (unsigned)((pg * NCB * 16 + m16) * 16);
#else
  const unsigned lane_b = (unsigned)(((pg * NCB * 16 + m16) * 16) + (g & 1) * 8);
#endif
  const unsigned term_b = (unsigned)p.UB * 16;
  const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)lds_raw;
  struct BAddr { unsigned lo[NT], hi[NT]; };  // per term: address of the half read first / second (column block 0)
  auto b_addr = [&](unsigned vo) __attribute__((always_inline)) {
    BAddr a;
#pragma unroll
    for (int t = 0; t < NT; ++t) { a.lo[t] = lds_base + vo + t * term_b; a.hi[t] = a.lo[t] ^ 8u; }
    return a;
  };
  auto read_b = [&](u32x4 (&B)[NT], const BAddr& a, int cb) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#ifdef NC_S3X_B128
      typedef const volatile __attribute__((address_space(3))) u32x4* lds128_t;
      B[t] = *(lds128_t)(a.lo[t] + cb * 256);
#else
      u64x2 v;
      v.x = *(lds64_t)(a.lo[t] + cb * 256);  // volatile: two ds_read_b64, never one ds_read2_b64
      v.y = *(lds64_t)(a.hi[t] + cb * 256);
      B[t] = __builtin_bit_cast(u32x4, v);
#endif
    }
  };

  // ---- results leave through a buffer descriptor over one sample's output: a lane whose position is a pad column or lies
  // beyond the plane stores to an out-of-range offset, which the hardware drops -- the NUMBER of store instructions is fixed,
  // and the hand-placed vmcnt waits below can count them.
  // The stores of tile t are issued inside the first kStoreSteps k-steps of tile t + 1 (the finished sums wait in `tot`, which
  // the new tile does not touch before its first accumulator restart): the write burst of 256 workgroups finishing together and
  // its drain (vmcnt is in-order: any later wait for a load also waits for older stores) cost ~20 us per tile when the epilogue
  // stood between two tiles.
  constexpr int kStoreSteps = 4;
  constexpr int kPairs = 2 * NCB;                      // (row block, column block) pairs, four stores each
  constexpr int kPairsPerStep = (kPairs + kStoreSteps - 1) / kStoreSteps;
  // (odd NCB: the last of the four steps carries fewer pairs)
  f32x4 acc[2][NCB], tot[2][NCB];
  // bias of the tile whose sums wait in `tot`: requested in that tile's last k-step by the same hand-counted kind of load as the
  // weights (an empty descriptor when there is no bias: zeros), complete at the wait of the next step
  u32x4 brsrc;
  {
    const unsigned long long ba = (unsigned long long)p.bias;
    brsrc.x = __builtin_amdgcn_readfirstlane((unsigned)ba);
    brsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu);
    brsrc.z = __builtin_amdgcn_readfirstlane(p.bias ? (unsigned)p.K * 4u : 0u);
    brsrc.w = __builtin_amdgcn_readfirstlane(0x00020000u);
  }
  // NT = 2: both operands were scaled by powers of two before the split; the sums are scaled back (exactly) on their way out
  // (two factors of about equal exponent, one after the other: no intermediate leaves the fp32 range unless the result does)
  float oscx = 1.f, oscw = 1.f;
  if constexpr (NT == 2) { const float2 f = h2_unscale2(*p.amax_x, *p.amax_w); oscx = f.x; oscw = f.y; }
  u32x4 bv[2];
  float st_s[2][4], st_q[2][4];  // ST: this lane's share of the sums of the tile whose results are being stored
  auto load_bias = [&](const XTile& t) __attribute__((always_inline)) {
    const int bo = (t.cot * 64 + half * 32 + 4 * g) * 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bv[rb]) : "v"(bo + rb * 64), "s"(brsrc) : "memory");
  };
  auto store_pairs = [&](const XTile& t, const int p0, const int p1) __attribute__((always_inline)) {  // pairs p0 .. p1 - 1 of tile t from `tot`, then tot = 0
    const int cob = t.cot * 64 + half * 32 + 4 * g;
    const __amdgpu_buffer_rsrc_t ys =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (long)t.n * p.K * S, 0, (unsigned)((long)p.K * S * 4), 0x00020000);
#pragma unroll
    for (int pr = 0; pr < kPairs; ++pr) {
      if (pr < p0 || pr >= p1) continue;
      const int rb = pr / NCB, cb = pr % NCB;
      const unsigned f = (unsigned)(t.q0 + pg * NCB * 16 + cb * 16 + m16);
      const unsigned yy = fdiv(f, p.mP);
      const unsigned xx = f - yy * p.P;
      const bool ok = (int)yy < p.H && (int)xx < p.W;
      const unsigned vo0 = ok ? (unsigned)(((long)(cob + rb * 16) * S + (long)t.z * HW + yy * p.W + xx) * 4) : kOut;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned bu = bv[rb][e];  // (a bit_cast straight from the vector element reads element 0)
        const float v = NT == 2 ? tot[rb][cb][e] * oscx * oscw + __uint_as_float(bu) : tot[rb][cb][e] + __uint_as_float(bu);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ys, ok ? vo0 + (unsigned)(e * S * 4) : kOut, 0, 0);
        if constexpr (ST) {
          const float r = ok ? tot[rb][cb][e] * oscx * oscw : 0.f;
          st_s[rb][e] += r;
          st_q[rb][e] = __builtin_fmaf(r, r, st_q[rb][e]);
        }
        tot[rb][cb][e] = 0.f;
      }
    }
  };

  // ST: the tile's sums leave as ONE 8-byte store per lane (lanes m16 < 8 of every lane group: channel half*32 + 4g + 16 (m16 / 4) + m16 % 4;
  // the others store out of range) -- a fixed number of store instructions, counted by the waits like the result stores
  auto stats_zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int e = 0; e < 4; ++e) { st_s[rb][e] = 0.f; st_q[rb][e] = 0.f; }
  };
  auto stats_flush = [&](int tidx) __attribute__((always_inline)) {
    float ss = 0.f, qq = 0.f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = st_s[rb][e], b = st_q[rb][e];
        // butterfly over the 16 lanes of the row: xor 1, xor 2 (quad permutes), then the mirrors (values are already uniform inside the
        // quads / the halves they swap)
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xf, 0xf, false));
        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0xB1, 0xf, 0xf, false));
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xf, 0xf, false));
        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x4E, 0xf, 0xf, false));
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x141, 0xf, 0xf, false));
        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x141, 0xf, 0xf, false));
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x140, 0xf, 0xf, false));
        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x140, 0xf, 0xf, false));
        if (m16 == rb * 4 + e) { ss = a; qq = b; }
      }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.stats, 0, 0x7fffffff, 0x00020000);
    const unsigned off = m16 < 8 ? (unsigned)((((unsigned)tidx * kWaves + wave) * 32 + g * 8 + m16) * 8) : kOut;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 pk;
    pk.x = __builtin_bit_cast(unsigned, ss); pk.y = __builtin_bit_cast(unsigned, qq);
    __builtin_amdgcn_raw_buffer_store_b64(pk, rs, off, 0, 0);
  };

  // ---- prologue: brick 0 and the first A fragments of the first tile
  int ring = 0;  // ring slot of brick 0 of the current tile
  issue_brick(cur, 0, 0, false);
  u32x4 A[2][NT], nA[2][NT];
  load_a(A, wtile(cur.cot));
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[rb][cb][e] = 0.f; tot[rb][cb][e] = 0.f; }
  bool have_prev = false;
  XTile prv = cur;
  int tprv = tcur;  // index of `prv` among this launch's tiles (ST: its slot in p.stats)

#ifdef NC_S3X_STAMP
  if (p.dbg && tid == 0) p.dbg[4000 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();  // per-workgroup start / end, 100 MHz
  int nstamp = 0;
#define STAMP() do { if (p.dbg && blockIdx.x == 0 && tid == 0 && nstamp < 4000) p.dbg[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  while (true) {
    STAMP();
    XTile nx{};
    const int tnext = next_tile(tcur + nslot, nx);
    const bool more_tiles = tnext >= 0;
    nxt = nx;
    if (keep && wave < kDmaWaves) {  // this tile's offsets (brick 0 was requested while the tile before was running: slow path)
#pragma unroll 1
      for (int i = 0, pc = wave; pc < p.npb; ++i, pc += kDmaWaves) {
        const unsigned u = (unsigned)(pc * 64 + lane);
        const unsigned term = fdiv(u, p.mUB);
        const unsigned F = (unsigned)cur.q0 + (u - term * p.UB);
        const unsigned rr = fdiv(F, p.mP);
        const int xx = (int)(F - rr * p.P) - PADX;
        const int yy = (int)rr - PAD;
        const bool ok = term < (unsigned)NT && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        *(lds32_t)(otab + i * 256) = ok ? (unsigned)(term * (unsigned)S + (unsigned)(yy * p.W + xx)) * 16u : kOut;
      }
    }
    const int wt = wtile(cur.cot);
    const int wt_next = more_tiles ? wtile(nxt.cot) : wt;
    int na = 0;  // next brick of this tile to arrive (brick 0 was requested during the previous tile / in the prologue)

    // per-lane tap state: lane group g is at tap tpl of the brick in slot sl
    int tpl = g, sl = ring;
    auto b_off = [&]() __attribute__((always_inline)) {
      const int dy = PC ? (tpl < 6 ? tpl : 6) : KS == 3 ? (tpl * 11) >> 5 : (tpl * 13) >> 6;
      const int dx = PC ? 0 : tpl - dy * KX;  // (PC: the empty eighth slot re-reads row 6 itself, not its neighbour -- zero weights, but 0 x inf is NaN)
      return lane_b + (unsigned)(sl * BB + (dy * p.P + dx) * 16);
    };
    BAddr vo = b_addr(b_off());
#if NC_S3X_BDEPTH == 2
    u32x4 B[3][NT];
#else
    u32x4 B[2][NT];
#endif
    int since = 0;

    // One k-step.  Ac = this step's A fragments (requested one step ago; step 0: during the last step of the previous tile), An
    // receives the next step's.  Order of a wave's vector-memory operations in a step: [A of step s + 1], then, when a brick
    // arrives, [its successor's DMA pieces], then [the previous tile's stores of this step].  "All but the 6 youngest complete"
    // at the top of a step therefore covers this step's A fragments and every DMA piece requested before this step; behind a
    // step that stored, the count is 6 + its stores.
    // `ph` (compile time): the step number for the first kStoreSteps steps of a tile, which carry the previous tile's stores; kStoreSteps
    // for every later step
    auto kstep = [&](auto ph, auto par, int s, u32x4 (&Ac)[2][NT], u32x4 (&An)[2][NT]) {
      constexpr int PH = decltype(ph)::value;
      // odd NCB: a step's last column block leaves the next step's first fragments in B[1] -- odd steps walk the two buffers the other way round
      constexpr int PAR = (NCB & 1) ? decltype(par)::value : 0;
      const bool last = PH == kStoreSteps && s + 1 == p.NS;
      load_a(An, last ? wt_next : wt + (s + 1) * (2 * NT * 1024));  // (last step: A of step 0 of the next tile, or a dummy request)
      constexpr int PPH = PH >= 1 && PH < kStoreSteps ? PH - 1 : kStoreSteps - 1;  // the step before this one, if it stored
      constexpr int PPairs = kPairs - PPH * kPairsPerStep < kPairsPerStep ? kPairs - PPH * kPairsPerStep : kPairsPerStep;
      wait_a(Ac, std::integral_constant<int, 4 * PPairs + ((ST && PPH == kStoreSteps - 1) ? 1 : 0)>{},
             have_prev && ((PH >= 1 && PH < kStoreSteps) || s == kStoreSteps));
      // brick `na` is first used by k-step s + 1 (brick 0: by step 0): it is complete in LDS for THIS wave's pieces; the barrier
      // makes that true for everybody's, and says everybody is done with brick na - 2 (last tap consumed in k-step s - 1 at the
      // latest, KS^2 > 6), whose slot the brick after `na` is requested into
      if (na < NB && 4 * s + 7 >= T2 * na) {
#ifndef NC_XA_NOBAR
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        // (tried: waves 4 .. 7 -- the SIMD partners of waves 0 .. 3 -- issuing their DMA pieces one k-step later, so that the two
        // partners are not away from the matrix pipe at the same moment: 1.5-2 % slower)
#ifndef NC_XA_NODMA
        if (na + 1 < NB) {
          issue_brick(cur, na + 1, (ring + na + 1) % 3, keep);
        } else if (more_tiles) {
          issue_brick(nxt, 0, (ring + NB) % 3, false);
        }
#endif
        ++na;
      }
      if (last) load_bias(cur);  // (complete at the next step's wait: it is older than that step's 6 A requests)
      if constexpr (PH < kStoreSteps) {
        if (have_prev) {
          asm volatile("" : "+v"(bv[0]), "+v"(bv[1]));
          if constexpr (ST && PH == 0) stats_zero();
          store_pairs(prv, PH * kPairsPerStep, (PH + 1) * kPairsPerStep);
          if constexpr (ST && PH == kStoreSteps - 1) stats_flush(tprv);
        }
      }
#if NC_S3X_BDEPTH == 2
      (void)PAR;
      if constexpr (PH == 0) { read_b(B[0], vo, 0); read_b(B[1], vo, 1); }
#else
      if constexpr (PH == 0) read_b(B[PAR], vo, 0);
#endif
      // next k-step's tap state
      tpl += 4;
      if (tpl >= T2) { tpl -= T2; sl = sl == 2 ? 0 : sl + 1; }
      const BAddr nvo = b_addr(b_off());
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#if NC_S3X_BDEPTH == 2
        // fragments requested TWO column blocks ahead (three register sets in rotation; the sets of the next step's first two blocks are
        // renamed to B[0], B[1] at the end of the step)
        u32x4(&Bc)[NT] = B[cb % 3];
        u32x4(&Bn)[NT] = B[(cb + 2) % 3];
        if (cb + 2 < NCB) read_b(Bn, vo, cb + 2);
        else if (!last) read_b(Bn, nvo, cb + 2 - NCB);
#else
        u32x4(&Bc)[NT] = B[(cb + PAR) & 1];
        u32x4(&Bn)[NT] = B[(cb + PAR + 1) & 1];
        if (cb + 1 < NCB) read_b(Bn, vo, cb + 1);
        else if (!last) read_b(Bn, nvo, 0);
#endif
        // six (NT = 2: three) products per (row block, column block), smallest first: (term of A, term of B)
        constexpr int NP = NT == 3 ? 6 : 3;
        constexpr int TA[6] = {NT - 1, NT == 3 ? 1 : 0, 0, 1, 0, 0};
        constexpr int TB[6] = {0, 1, NT == 3 ? 2 : 0, 0, 1, 0};
#pragma unroll
        for (int m = 0; m < NP; ++m)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            if constexpr (NT == 3)
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Ac[rb][TA[m]]), __builtin_bit_cast(bf16x8, Bc[TB[m]]),
                                                                    acc[rb][cb], 0, 0, 0);
            else
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ac[rb][TA[m]]), __builtin_bit_cast(f16x8, Bc[TB[m]]),
                                                                   acc[rb][cb], 0, 0, 0);
          }
        if constexpr (NT == 3) {
#pragma unroll
          for (int k = 0; k < 6; ++k) {  // the 6 reads of the next column block spread over this one's 12 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
        } else {
#ifdef NC_S3X_B128
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#else
          // the 4 reads of the next column block spread over this one's 6 MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#endif
        }
      }
#if NC_S3X_BDEPTH == 2
      if constexpr (NCB % 3 != 0) {
        u32x4 t0[NT], t1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { t0[t] = B[NCB % 3][t]; t1[t] = B[(NCB + 1) % 3][t]; }
#pragma unroll
        for (int t = 0; t < NT; ++t) { B[0][t] = t0[t]; B[1][t] = t1[t]; }
      }
#endif
      vo = nvo;
      if (++since == p.flush || last) {
        since = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) { tot[rb][cb][e] += acc[rb][cb][e]; acc[rb][cb][e] = 0.f; }
      }
    };
    static_assert(kStoreSteps == 4, "the four peeled steps below");
    STAMP();
    kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, A, nA);
    STAMP();
    kstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, 1, nA, A);
    STAMP();
    kstep(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, 2, A, nA);
    STAMP();
    kstep(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, 3, nA, A);
    STAMP();
#pragma unroll 1
    for (int s = 4; s < p.NS; s += 2) {  // NS is even: step 0 of the next tile finds its fragments in A again
      kstep(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{}, s, A, nA);
      kstep(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}, s + 1, nA, A);
      STAMP();
    }
    prv = cur;
    tprv = tcur;
    have_prev = true;
    if (!more_tiles) break;
    ring = (ring + NB) % 3;
    cur = nxt;
    tcur = tnext;
  }
  // the last tile's results (and the dummy request of its last step)
  // (A is an operand of the wait ON PURPOSE: it is the target of the last step's dummy request (NS is even: the last step reads nA and
  // requests into A), and a register the compiler considers dead is free for whatever the epilogue computes next -- the ST accumulators were
  // zeroed in front of the wait and a late fragment landed on top of them: wrong statistics in 20-60 % of the calls of the 20^3 level, whose
  // last k-step is the shortest.  nA's last request was waited for by the last step itself.)
  if constexpr (NT == 3)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2])::"memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1])::"memory");
  if constexpr (ST) stats_zero();
  store_pairs(prv, 0, kPairs);
  if constexpr (ST) stats_flush(tprv);
#ifdef NC_S3X_STAMP
  if (p.dbg && tid == 0) p.dbg[4001 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- Round 6: the 64-channel wave tile with the weights staged through LDS ("W64"; 3^3, two-term operands, 512-position tiles).
// tools/mfma_feed.hip modes 10 / 11: what LDS read instructions cost is matrix-pipe occupancy, and a wave tile of 64 channels x 64 positions needs 16
// fragment reads per 48 MFMAs where 32 channels x 128 positions needs 32 + 4 KiB of weights through L1 per wave -- once the k-step's 8 KiB of weight
// fragments are staged in LDS ONCE per workgroup (by LDS-DMA, beside the brick ring) instead of being fetched by every wave.  The 64 x 64 accumulators
// (64 registers) and two sets of eight weight fragments (64) leave no room for k_conv_s3x's second accumulator set: NO accumulator restarts -- one
// running fp32 sum per tile, rms error 2^-24 sqrt(k-steps) of the output's rms, the fp32 kernels' own level (conv_s3x_h2 gives
// k_conv_s3x the same rule while this form is switched on, so that an element's bits do not depend on the kernel its tile fell to) -- and the
// tile's stores stand between two tiles.  (The form that keeps both -- ONE set of operand registers re-filled half a k-step ahead, restarts and
// deferred stores in the 64 registers that frees -- was built and is 2-4 % slower than k_conv_s3x: the LDS pipe is ~75 % busy, a re-fill issued
// 12 MFMAs ahead is late.)  Everything else is k_conv_s3x's: the tap stream, the brick ring and its arrival events, the DMA offset table, the
// B-fragment reads, the tile order.
//   * eight waves = eight groups of 64 positions, every wave all 64 output channels (4 row blocks x 4 column blocks, 48 MFMAs per k-step);
//   * weights: piece i (1 KiB) of k-step s = fragment (half = i / 4, f = i % 4) of the packed weights, into slot s % 6 of a ring of six k-steps
//     (k-steps per tile are a multiple of 6, so the ring runs on across tiles).  At the arrival event of brick na (k-step fire(na)) waves 0-3 request
//     the pieces of the k-steps (fire(na + 1), fire(na + 2)]: complete at the next event's barrier, one event before their first use; at most five
//     k-steps are live;
//   * vector-memory waits (waves 0-3 hold DMA): at an event, everything older than the DMA just waited for is complete after s_waitcnt vmcnt(0) --
//     except at k-step 1, where the 64 (+ 1) stores of the previous tile are YOUNGER than the DMA requested at k-step 0: vmcnt(63) (at least one
//     store and everything in front of it has retired);
//   * epilogue statistics (ST): the two waves of a 128-position group add their sums through LDS and write k_conv_s3x's record format.
constexpr int kOffTab = 8192;  // two-term launches: the DMA offset table behind the ring, [4 issuing waves][8 pieces][64 lanes] words
constexpr int kWA = 6;  // k-steps of weight fragments in the LDS ring
constexpr int kWExtra = kWA * 8192 + 4096;  // LDS of k_conv_s3w behind the ring and the offset table: the weight ring, the ST exchange
__device__ __forceinline__ int w_fire(int na) { return na < 2 ? na : (9 * na - 4) >> 2; }  // the k-step at which brick na's arrival is declared
template <bool ST>
__global__ void __launch_bounds__(kThreads, 1) k_conv_s3w(const XParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  constexpr int KS = 3, NT = 2, PAD = 1, T2 = 9, NCB = 4, PT = 512;
  if (guard_skip(p.guard, false)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  const int pg = wave;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int NB = p.NCH * KS;
  const int BB = p.npb * 1024;
  unsigned char* const aring = lds_raw + 3 * BB + kOffTab;
  float* const stx = reinterpret_cast<float*>(aring + kWA * 8192);  // ST: [wave][64 channels][2]

  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int t_lo = xcd * p.tiles_per_xcd;
  int t_hi = t_lo + p.tiles_per_xcd;
  if (t_hi > p.t_count) t_hi = p.t_count;
  auto next_tile = [&](int t, XTile& o) __attribute__((always_inline)) {
    for (; t < t_hi; t += nslot) {
      o = x_decode<PT>(p, t);
      if (o.q0 < p.HP) return t;
    }
    return -1;
  };
  XTile cur, nxt;
  int tcur = next_tile(t_lo + wslot, cur);
  if (tcur < 0) return;

  constexpr unsigned kOut = 0x80000000u;
  constexpr int kPW = 8;
  const bool keep = p.npb <= kDmaWaves * kPW;
  typedef volatile __attribute__((address_space(3))) unsigned* lds32_t;
  const unsigned otab = (unsigned)(unsigned long long)(lptr_t)lds_raw + (unsigned)(3 * BB + ((wave * kPW) * 64 + lane) * 4);
  auto issue_brick = [&](const XTile& t, int bi, int slot, bool kept) __attribute__((always_inline)) {
    if (wave >= kDmaWaves) return;
    const int chunk = bi / KS, dz = bi - chunk * KS;
    const int zz = t.z + dz - PAD;
    const bool zok = (unsigned)zz < (unsigned)p.D;
    const uint4* blk = p.xs + ((long)t.n * p.NCH + chunk) * NT * S;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(blk), 0, zok ? (unsigned)(NT * S * 16) : 0u, 0x00020000);
    const int soff = zok ? (int)(zz * HW * 16) : 0;
    unsigned char* buf = lds_raw + slot * BB;
    if (kept) {
      unsigned po[kPW];
#pragma unroll
      for (int i = 0; i < kPW; ++i) po[i] = *(lds32_t)(otab + i * 256);
#pragma unroll
      for (int i = 0; i < kPW; ++i) {
        const int pc = wave + kDmaWaves * i;
        if (pc < p.npb) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(buf + pc * 1024), 16, po[i], soff, 0, 0);
      }
      return;
    }
#pragma unroll 1
    for (int pc = wave; pc < p.npb; pc += kDmaWaves) {
      const unsigned u = (unsigned)(pc * 64 + lane);
      const unsigned term = fdiv(u, p.mUB);
      const unsigned F = (unsigned)t.q0 + (u - term * p.UB);
      const unsigned rr = fdiv(F, p.mP);
      const int xx = (int)(F - rr * p.P) - PAD;
      const int yy = (int)rr - PAD;
      const bool ok = term < (unsigned)NT && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const unsigned po = ok ? (unsigned)(term * (unsigned)S + (unsigned)(yy * p.W + xx)) * 16u : kOut;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(buf + pc * 1024), 16, po, soff, 0, 0);
    }
  };
  // weight pieces of the k-steps s_lo + 1 .. s_hi of tile t (waves 0-3: pieces w and w + 4 of every k-step)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.wp), 0, 0x7fffffffu, 0x00020000);
  auto issue_a = [&](const XTile& t, int s_lo, int s_hi) __attribute__((always_inline)) {
    if (wave >= kDmaWaves) return;
#pragma unroll 1
    for (int s = s_lo + 1; s <= s_hi; ++s) {
      unsigned char* dst = aring + (s % kWA) * 8192;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = wave + 4 * j;  // piece: half = i / 4, f = i % 4
        const int soff = ((t.cot * 2 + (i >> 2)) * p.NS + s) * 4096 + (i & 3) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_t)(dst + i * 1024), 16, (unsigned)(lane * 16), __builtin_amdgcn_readfirstlane(soff), 0, 0);
      }
    }
  };

  // B fragments (as k_conv_s3x)
  const unsigned lane_b = (unsigned)(((pg * NCB * 16 + m16) * 16) + (g & 1) * 8);
  const unsigned term_b = (unsigned)p.UB * 16;
  const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)lds_raw;
  struct BAddr { unsigned lo[NT], hi[NT]; };
  auto b_addr = [&](unsigned vo) __attribute__((always_inline)) {
    BAddr a;
#pragma unroll
    for (int t = 0; t < NT; ++t) { a.lo[t] = lds_base + vo + t * term_b; a.hi[t] = a.lo[t] ^ 8u; }
    return a;
  };
  auto read_b = [&](u32x4 (&B)[NT], const BAddr& a, int cb) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      u64x2 v;
      v.x = *(lds64_t)(a.lo[t] + cb * 256);
      v.y = *(lds64_t)(a.hi[t] + cb * 256);
      B[t] = __builtin_bit_cast(u32x4, v);
    }
  };
  // A fragments of k-step s: fragment (rb, term) = piece (rb / 2) * 4 + (rb % 2) * 2 + term of the ring slot
  typedef const volatile __attribute__((address_space(3))) u32x4* lds128_t;
  const unsigned a_lane = (unsigned)(unsigned long long)(lptr_t)aring + (unsigned)(lane * 16);
  auto read_a = [&](u32x4 (&A)[4][NT], int s) {
    const unsigned base = a_lane + (unsigned)((s % kWA) * 8192);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int t = 0; t < NT; ++t) A[rb][t] = *(lds128_t)(base + (unsigned)(((rb >> 1) * 4 + (rb & 1) * 2 + t) * 1024));
  };

  float oscx, oscw;
  { const float2 f = h2_unscale2(*p.amax_x, *p.amax_w); oscx = f.x; oscw = f.y; }
  u32x4 brsrc;
  {
    const unsigned long long ba = (unsigned long long)p.bias;
    brsrc.x = __builtin_amdgcn_readfirstlane((unsigned)ba);
    brsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu);
    brsrc.z = __builtin_amdgcn_readfirstlane(p.bias ? (unsigned)p.K * 4u : 0u);
    brsrc.w = __builtin_amdgcn_readfirstlane(0x00020000u);
  }
  u32x4 bv[4];
  auto load_bias = [&](const XTile& t) __attribute__((always_inline)) {
    const int bo = (t.cot * 64 + 4 * g) * 4;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bv[rb]) : "v"(bo + rb * 64), "s"(brsrc) : "memory");
  };
  f32x4 acc[4][NCB];
  // the finished tile: scale back, add the bias, store (and, ST, leave the wave's sums in k_conv_s3x's record format)
  auto epilogue = [&](const XTile& t, int tidx) __attribute__((always_inline)) {
    const int cob = t.cot * 64 + 4 * g;
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(p.y + (long)t.n * p.K * S, 0, (unsigned)((long)p.K * S * 4), 0x00020000);
    float st_s[4][4], st_q[4][4];
    if constexpr (ST) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) { st_s[rb][e] = 0.f; st_q[rb][e] = 0.f; }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const unsigned f = (unsigned)(t.q0 + pg * NCB * 16 + cb * 16 + m16);
      const unsigned yy = fdiv(f, p.mP);
      const unsigned xx = f - yy * p.P;
      const bool ok = (int)yy < p.H && (int)xx < p.W;
      // one per-lane offset per column block (a pad column / a position beyond the plane: out of the descriptor's range, and it stays there:
      // K S 4 < 2^31), the channel through the scalar offset -- no vector arithmetic per store
      const unsigned vo0 = ok ? (unsigned)(((long)cob * S + (long)t.z * HW + yy * p.W + xx) * 4) : kOut;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned bu = bv[rb][e];
          const float r = acc[rb][cb][e] * oscx * oscw;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r + __uint_as_float(bu)), ys, vo0, (int)((rb * 16 + e) * S * 4), 0);
          if constexpr (ST) {
            const float rr = ok ? r : 0.f;
            st_s[rb][e] += rr;
            st_q[rb][e] = __builtin_fmaf(rr, rr, st_q[rb][e]);
          }
          acc[rb][cb][e] = 0.f;
        }
      }
    }
    if constexpr (ST) {
      // butterfly over the 16 lanes of a row (k_conv_s3x stats_flush), then lane m16 = (rb % 2) * 4 + e of group g holds channel
      // (rb / 2) * 32 + 4 g + 16 (rb % 2) + e: the wave's 64 channel sums go to LDS, the partner wave's are added, one record per (group of 128, half)
      float ss[2] = {0.f, 0.f}, qq[2] = {0.f, 0.f};
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = st_s[rb][e], b = st_q[rb][e];
          a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xf, 0xf, false));
          b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0xB1, 0xf, 0xf, false));
          a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xf, 0xf, false));
          b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x4E, 0xf, 0xf, false));
          a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x141, 0xf, 0xf, false));
          b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x141, 0xf, 0xf, false));
          a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x140, 0xf, 0xf, false));
          b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x140, 0xf, 0xf, false));
          if (m16 == (rb & 1) * 4 + e) { ss[rb >> 1] = a; qq[rb >> 1] = b; }
        }
      // position inside a 32-channel record: g * 8 + m16 (m16 < 8), as k_s3x_stats_final decodes it
      if (m16 < 8) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          stx[((wave * 2 + h) * 32 + g * 8 + m16) * 2] = ss[h];
          stx[((wave * 2 + h) * 32 + g * 8 + m16) * 2 + 1] = qq[h];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // wave w writes the record (group w / 2, half w % 2): its own sums of that half + its partner's
      const int h = wave & 1;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.stats, 0, 0x7fffffff, 0x00020000);
      float s0 = 0.f, q0 = 0.f;
      if (m16 < 8) {
        const int li = g * 8 + m16;
        const int lo = wave & ~1;  // (the even wave's sums first: a fixed order)
        s0 = stx[((lo * 2 + h) * 32 + li) * 2] + stx[(((lo + 1) * 2 + h) * 32 + li) * 2];
        q0 = stx[((lo * 2 + h) * 32 + li) * 2 + 1] + stx[(((lo + 1) * 2 + h) * 32 + li) * 2 + 1];
      }
      const unsigned off = m16 < 8 ? (unsigned)((((unsigned)tidx * kWaves + wave) * 32 + g * 8 + m16) * 8) : kOut;
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      u32x2 pk;
      pk.x = __builtin_bit_cast(unsigned, s0); pk.y = __builtin_bit_cast(unsigned, q0);
      __builtin_amdgcn_raw_buffer_store_b64(pk, rs, off, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (stx is free again)
    }
  };

  // ---- prologue: brick 0 and the weights of k-steps 0, 1 of the first tile
  int ring = 0;
  issue_brick(cur, 0, 0, false);
  issue_a(cur, -1, 1);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[rb][cb][e] = 0.f;
  bool have_prev = false;
  XTile prv = cur;
  int tprv = tcur;

  while (true) {
    XTile nx{};
    const int tnext = next_tile(tcur + nslot, nx);
    const bool more_tiles = tnext >= 0;
    nxt = nx;
    if (keep && wave < kDmaWaves) {
#pragma unroll 1
      for (int i = 0, pc = wave; pc < p.npb; ++i, pc += kDmaWaves) {
        const unsigned u = (unsigned)(pc * 64 + lane);
        const unsigned term = fdiv(u, p.mUB);
        const unsigned F = (unsigned)cur.q0 + (u - term * p.UB);
        const unsigned rr = fdiv(F, p.mP);
        const int xx = (int)(F - rr * p.P) - PAD;
        const int yy = (int)rr - PAD;
        const bool ok = term < (unsigned)NT && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        *(lds32_t)(otab + i * 256) = ok ? (unsigned)(term * (unsigned)S + (unsigned)(yy * p.W + xx)) * 16u : kOut;
      }
    }
    int na = 0;
    int tpl = g, sl = ring;
    auto b_off = [&]() __attribute__((always_inline)) {
      const int dy = (tpl * 11) >> 5;
      const int dx = tpl - dy * KS;
      return lane_b + (unsigned)(sl * BB + (dy * p.P + dx) * 16);
    };
    BAddr vo = b_addr(b_off());
    u32x4 B[2][NT];
    u32x4 A0[4][NT], A1[4][NT];

    auto kstep = [&](int s, u32x4 (&Ac)[4][NT], u32x4 (&An)[4][NT], auto first_) {
      constexpr bool first = decltype(first_)::value;
      const bool last = s + 1 == p.NS;
      if (na < NB && 4 * s + 7 >= T2 * na) {
        // own DMA of the previous event complete (s == 1 behind an epilogue: its stores are younger, see the header); everybody's after the barrier
        if (wave < kDmaWaves) {
          if (s == 1 && have_prev) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (na + 1 < NB) issue_brick(cur, na + 1, (ring + na + 1) % 3, keep);
        else if (more_tiles) issue_brick(nxt, 0, (ring + NB) % 3, false);
        // weights of the k-steps (fire(na + 1), fire(na + 2)], running on into the next tile
        {
          const int f1 = na + 1 < NB ? w_fire(na + 1) : p.NS + w_fire(na + 1 - NB);
          const int f2 = na + 2 < NB ? w_fire(na + 2) : p.NS + w_fire(na + 2 - NB);
          const int hi_cur = f2 < p.NS ? f2 : p.NS - 1;
          if (f1 < p.NS) issue_a(cur, f1, hi_cur);
          if (f2 >= p.NS && more_tiles) issue_a(nxt, (f1 >= p.NS ? f1 - p.NS : -1), f2 - p.NS);
        }
        ++na;
      }
      if constexpr (first) {
        if (have_prev) epilogue(prv, tprv);
        read_a(Ac, 0);
        read_b(B[0], vo, 0);
      }
      if (last) load_bias(cur);
      tpl += 4;
      if (tpl >= T2) { tpl -= T2; sl = sl == 2 ? 0 : sl + 1; }
      const BAddr nvo = b_addr(b_off());
      const unsigned abase = a_lane + (unsigned)(((s + 1) % kWA) * 8192);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        u32x4(&Bc)[NT] = B[cb & 1];
        u32x4(&Bn)[NT] = B[(cb + 1) & 1];
        if (cb + 1 < NCB) read_b(Bn, vo, cb + 1);
        else if (!last) read_b(Bn, nvo, 0);
        // the next k-step's weight fragments of row block cb (the last step of a tile: the next tile reads its own behind the stores)
        if (!last) {
#pragma unroll
          for (int t = 0; t < NT; ++t) An[cb][t] = *(lds128_t)(abase + (unsigned)(((cb >> 1) * 4 + (cb & 1) * 2 + t) * 1024));
        }
        // three products per (row block, column block), smallest first: (term of A, term of B) = (1, 0), (0, 1), (0, 0)
        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ac[rb][TA[m]]), __builtin_bit_cast(f16x8, Bc[TB[m]]), acc[rb][cb], 0, 0, 0);
        // the 6 reads spread over the 12 MFMAs
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
      }
      vo = nvo;
      // the bias of the finished tile is a plain value from here on (requested at the top of the step; the wave's DMA of the last event is
      // three k-steps old)
      if (last) {
        u32x4(&b4)[4] = bv;  // (named here: operands of an asm statement alone do not make a generic lambda capture the array)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3])::"memory");
      }
    };
    kstep(0, A0, A1, std::true_type{});
#pragma unroll 1
    for (int s = 1; s + 1 < p.NS; s += 2) {  // NS is even: the last k-step reads A1
      kstep(s, A1, A0, std::false_type{});
      kstep(s + 1, A0, A1, std::false_type{});
    }
    kstep(p.NS - 1, A1, A0, std::false_type{});
    prv = cur;
    tprv = tcur;
    have_prev = true;
    if (!more_tiles) break;
    ring = (ring + NB) % 3;
    cur = nxt;
    tcur = tnext;
  }
  epilogue(prv, tprv);
}

struct XPlan {
  int NCB, fsub, P, HP, TPP, UB, npb, lds;
  int UBt, npbt, ldst;  // the tail launch's brick (PT / fsub positions)
  long full, rem;       // main tiles in whole rounds of 256 / left over
  bool ok;
};

bool x_brick(int PT, int P, int KS, int NT, int& UB, int& npb, int& lds, bool pc = false) {
  const int U = PT + (KS - 1) * (pc ? P : P + 1);  // (pseudo-channel form: in-plane taps are rows only)
  UB = (U + 63) / 64 * 64;
  npb = NT * UB / 64;
  lds = 3 * npb * 1024;
  return npb <= kMaxPieces && lds + (NT == 2 ? kOffTab : 0) <= kLdsMax;
}

int x_tail_mode() {  // NC_S3X_TAIL=0: the left-over tiles run as one more round of whole tiles (A/B)
  static const int m = getenv("NC_S3X_TAIL") ? atoi(getenv("NC_S3X_TAIL")) : 1;
  return m;
}

XPlan x_plan(int N, int D, int H, int W, int KT, int KS, int NT = 3, bool k32 = false, bool pc = false) {
  XPlan best{};
  double best_cost = 1e30;
  const int P = pc ? W : W + KS - 1;
  const long HP = (long)H * P;
  static const int ncb_max = 8;
  for (int NCB : {8, 7, 6, 4, 2}) {
    if (NCB > ncb_max) continue;
    if (k32 && NCB != 4 && NCB != 2) continue;  // (32-channel tiles: the instantiated shapes; a tile is 128 NCB positions)
    static const int odd_ok = getenv("NC_S3X_NCB7") ? atoi(getenv("NC_S3X_NCB7")) : 1;  // NC_S3X_NCB7=0: even column-block counts only (A/B)
    if ((NCB & 1) && !odd_ok) continue;
    XPlan pl{};
    pl.NCB = NCB; pl.P = P; pl.HP = (int)HP;
    const int PT = (k32 ? 128 : 64) * NCB;
    if (!x_brick(PT, P, KS, NT, pl.UB, pl.npb, pl.lds, pc)) continue;
    pl.TPP = (int)((HP + PT - 1) / PT);
    const long ntiles = (long)N * D * pl.TPP * KT;
    pl.full = ntiles / 256 * 256;
    pl.rem = ntiles - pl.full;
    // time in units of "positions per workgroup"; a tile costs its positions plus a fixed part (prologue, epilogue, halo)
    const double fixed = 56;
    double cost = (double)(pl.full / 256) * (PT + fixed);
    pl.fsub = 1;
    if (pl.rem) {
      int f = 1;  // the left-over tiles in quarters / thirds / halves (8, 6, 4 -> 2 or 4 column blocks per wave) when they then fit one round
      if (x_tail_mode() && NCB > 2 && NCB % 2 == 0) {
        if (pl.rem * (NCB / 2) <= 256) f = NCB / 2;
        else if (NCB == 8 && pl.rem * 2 <= 256) f = 2;
      }
      if (k32 && f > 2) f = 2;
      pl.fsub = f;
      cost += PT / f + fixed;
    }
    if (!x_brick(PT / pl.fsub, P, KS, NT, pl.UBt, pl.npbt, pl.ldst, pc)) continue;
    if (cost < best_cost) { best_cost = cost; best = pl; best.ok = true; }
  }
  return best;
}

template <int KS, int NCB, int NT, bool ST = false, bool K32 = false, int PC = 0>
int launch_x(const XParams& p, int lds, hipStream_t s) {
  auto kern = k_conv_s3x<KS, NCB, NT, ST, K32, PC>;
  if (int e = raise_dyn_lds(kern, kLdsMax, "conv_s3x")) return e;
  hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), lds, s, p);
  return check_launch("conv_s3x");
}

// the W64 form (k_conv_s3w) takes a launch of whole 512-position tiles of a 3^3 layer whose k-steps per tile are a multiple of the weight ring
// (every multiple of 64 input channels) when its LDS fits.  NC_S3X_W64=0: k_conv_s3x everywhere (A/B, tests)
static std::atomic<int> g_w64{getenv("NC_S3X_W64") ? atoi(getenv("NC_S3X_W64")) : 1};
bool w64_applies(int KS, int NCB, int NS, int lds) {
  return g_w64.load(std::memory_order_relaxed) && KS == 3 && NCB == 8 && NS % kWA == 0 && NS >= 12 && lds + kOffTab + kWExtra <= kLdsMax;
}
}  // namespace
void s3x_w64_set(int on) { g_w64.store(on ? 1 : 0, std::memory_order_relaxed); }
int s3x_w64_get() { return g_w64.load(std::memory_order_relaxed); }
namespace {
template <bool ST>
int launch_w64(const XParams& p, int lds, hipStream_t s) {
  auto kern = k_conv_s3w<ST>;
  if (int e = raise_dyn_lds(kern, kLdsMax, "conv_s3w")) return e;
  hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), lds + kOffTab + kWExtra, s, p);
  return check_launch("conv_s3w");
}

template <int KS, int NT = 3, bool ST = false>
int launch_x_ncb(int NCB, const XParams& p, int lds, hipStream_t s) {
  if (NCB == 8) return launch_x<KS, 8, NT, ST>(p, lds, s);
  if (NCB == 7) return launch_x<KS, 7, NT, ST>(p, lds, s);
  if (NCB == 6) return launch_x<KS, 6, NT, ST>(p, lds, s);
  if (NCB == 4) return launch_x<KS, 4, NT, ST>(p, lds, s);
  return launch_x<KS, 2, NT, ST>(p, lds, s);
}

// ---- InstanceNorm statistics from the ST epilogue: per (sample, channel) the partial sums of every (tile, wave) that holds the channel, in
// fp64 and in a fixed order.  A record (one wave's 32 channels x (sum, sum of squares)) is 256 contiguous bytes, read by 32 lanes: workgroup
// (wave slice of 32 channels, sample, one of kStatChunks ranges of the records) walks its range with 8 lane groups interleaved and leaves
// 32 x (s, q) in fp64; k_s3x_stats_final adds the chunks in order.  (First version: one workgroup per channel reading 8 bytes out of every
// record -- 19 us per call at 140^3, a whole 64-byte sector fetched per 8 bytes used.)
constexpr int kStatChunks = 64;  // (16: 33 us per call at 140^3 -- 32 workgroups of 170 dependent iterations; measured with the one-stream profile)
struct XStatsPlan {
  int N, K, KT, TPP, D, HP;
  long full, main_count;   // main tiles in whole rounds / tiles of the first launch (full, or all when the left-over tiles are whole tiles)
  int fsub, PTsub;         // second launch: sub-tiles per main tile, positions per sub-tile
};
__global__ void __launch_bounds__(256) k_s3x_stats_partial(const float2* __restrict__ part, XStatsPlan pl, double2* __restrict__ chunk_sums) {
  const int slice = blockIdx.x, n = blockIdx.y, ch = blockIdx.z;  // slice = cot * 2 + half: channels slice * 32 .. + 31
  const int cot = slice >> 1, half = slice & 1;
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;  // l: position of the channel inside the record (lane (g, m16 < 8) -> g * 8 + m16)
  const long per_n = (long)pl.TPP * pl.D;
  double s = 0.0, q = 0.0;
  // record (tile index u of this (n, cot), position group pg): 4 per main tile, 4 * fsub per left-over tile -- walked in a fixed order
  const long nrec = per_n * 4, per = (nrec + kStatChunks - 1) / kStatChunks;
  const long r1 = (ch + 1) * per < nrec ? (ch + 1) * per : nrec;
#pragma unroll 4
  for (long r = ch * per + grp; r < r1; r += 8) {
    const long u = r >> 2;
    const int pg = (int)(r & 3);
    const long t = ((long)n * per_n + u) * pl.KT + cot;  // main tile index (x_decode: output-channel tile fastest)
    if (t < pl.main_count) {
      const float2 v = part[((t * kWaves) + pg * 2 + half) * 32 + l];
      s += (double)v.x; q += (double)v.y;
    } else {
      // the tile's in-plane index, as x_decode orders it (groups of kXGroup neighbours x planes)
      const int full_g = (pl.TPP / kXGroup) * kXGroup * pl.D;
      int tp;
      if (u < full_g) { const long g4 = u / (kXGroup * pl.D), rem = u - g4 * (kXGroup * pl.D); tp = (int)(g4 * kXGroup + (rem % kXGroup)); }
      else { const int L = pl.TPP % kXGroup; const long v2 = u - full_g; tp = (pl.TPP / kXGroup) * kXGroup + (int)(v2 % L); }
      for (int sub = 0; sub < pl.fsub; ++sub) {
        if ((long)(tp * pl.fsub + sub) * pl.PTsub >= pl.HP) continue;  // a sub-tile that starts beyond the plane was never run
        const long ti = pl.main_count + (t - pl.full) * pl.fsub + sub;
        const float2 v = part[((ti * kWaves) + pg * 2 + half) * 32 + l];
        s += (double)v.x; q += (double)v.y;
      }
    }
  }
  __shared__ double rs[8][32], rq[8][32];
  rs[grp][l] = s; rq[grp][l] = q;
  __syncthreads();
  if (grp == 0) {
    for (int g = 1; g < 8; ++g) { s += rs[g][l]; q += rq[g][l]; }
    chunk_sums[(((long)n * gridDim.x + slice) * kStatChunks + ch) * 32 + l] = make_double2(s, q);
  }
}
__global__ void __launch_bounds__(256) k_s3x_stats_final(const double2* __restrict__ chunk_sums, int K, const float* __restrict__ bias, long S, float eps,
                                                         float* __restrict__ mean, float* __restrict__ rstd) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;  // i = slice * 32 + record position
  if (i >= K) return;
  const int slice = i >> 5, l = i & 31;
  double s = 0.0, q = 0.0;
  for (int ch = 0; ch < kStatChunks; ++ch) {
    const double2 v = chunk_sums[(((long)n * (K / 32) + slice) * kStatChunks + ch) * 32 + l];
    s += v.x; q += v.y;
  }
  // record position l = g * 8 + m16 holds channel half*32 + 4g + 16 (m16 / 4) + m16 % 4 of its 64-channel tile
  const int g4 = l >> 3, m = l & 7;
  const int c = (slice >> 1) * 64 + (slice & 1) * 32 + 4 * g4 + 16 * (m >> 2) + (m & 3);
  const double m0 = s / (double)S;
  double var = q / (double)S - m0 * m0;
  if (var < 0.0) var = 0.0;
  mean[(long)n * K + c] = (float)(m0 + (bias ? (double)bias[c] : 0.0));
  rstd[(long)n * K + c] = (float)(1.0 / sqrt(var + (double)eps));
}

}  // namespace

// 0: off; 1 (default): in the inference forward; 2: in the training forward too (measured: no gain there -- 33.95-34.08 against 33.83-34.01 ms per
// step, the eight k_in_stats passes it removes cost what the epilogue and the two finalisation launches cost -- so it stays a tested option)
static std::atomic<int> g_epi_stats{getenv("NC_EPI_STATS") ? atoi(getenv("NC_EPI_STATS")) : 1};
bool epi_stats_on() { return g_epi_stats.load(std::memory_order_relaxed) != 0; }
int epi_stats_mode() { return g_epi_stats.load(std::memory_order_relaxed); }
void epi_stats_set(int on) { g_epi_stats.store(on < 0 ? 0 : on > 2 ? 2 : on, std::memory_order_relaxed); }

// k-steps of a tile: its KS^3 * Cin / 8 taps in fours, made EVEN (the kernel alternates two sets of weight registers per step and a tile must
// end where it began): channels % 64 give an even count by themselves; 32 channels (two-term launches only: deep_linear_gen's rank-structured
// data gradient, gen_nets.hip) get ONE padding step -- its packed weights are zero (k_pack_w_s3x: taps beyond the last brick), its B fragments are
// whatever the ring slot behind the last brick holds: units of the next tile's first brick or of this tile's brick 17, finite data of the same
// tensor times zero.  (A non-finite element there turns outputs of a neighbouring tile into NaN as well: the split rule's "only the outputs it
// touches" holds for channels % 64 only.)
int s3x_ksteps(int Cin, int KS) { return (KS * KS * KS * (Cin / 8) / 4 + 1) & ~1; }
size_t s3x_packed_bytes(int Cin, int Kout, int KS, int NT) {
  const int NS = s3x_ksteps(Cin, KS);
  return (size_t)(Kout / 32) * NS * 2 * NT * 1024;  // (32-channel slices: two per 64-channel tile)
}

// NC_SPLIT_TERMS / nc_set_split_terms: which form of the split the fp32 3^3 / 5^3 layers use -- the two-term fp16 form (three MFMA products per
// fp32 product; s3_common.hpp, h2.hip) or the three-term bf16 form (six).  2 (default): two-term wherever it exists -- forward, data gradient
// and weight gradient, layer by layer and in the whole-network training / inference calls; 3: three-term everywhere; 0: two-term in the
// inference forward nc_unet_deconv_fwd only.  The explicit S3 entry points (nc_conv_*_split, nc_to_s3) are three-term by definition.
static std::atomic<int> g_terms{getenv("NC_SPLIT_TERMS") ? atoi(getenv("NC_SPLIT_TERMS")) : 2};
void s3x_set_terms(int t) { g_terms = (t == 0 || t == 3) ? t : 2; }
int s3x_get_terms() { const int f = frozen_terms(); return f >= 0 ? f : g_terms.load(); }  // (inside a call: the value sampled when the call began)

bool s3x_k32_supported(int N, int D, int H, int W) { return x_plan(N, D, H, W, 1, 5, 2, true).ok && (long)32 * D * H * W * 4 < (1l << 31); }
bool s3x_supported(int N, int Cin, int D, int H, int W, int Kout, int KS) {
  if (KS != 3 && KS != 5) return false;
  if (Cin % 64 || Kout % 64) return false;  // an even number of whole k-steps: (Cin / 8) * KS^3 taps in fours
  if ((long)Kout * D * H * W * 4 >= (1l << 31)) return false;  // byte offsets inside one sample's output
  if ((long)D * H * W * 48 >= (1l << 31)) return false;  // byte offsets inside one block's three terms stay below the kOut mark
  if ((long)H * (W + KS - 1) + 4096 >= (1l << 31)) return false;
  return x_plan(N, D, H, W, Kout / 64, KS).ok;
}

// The convolution from an H2 input (h2.hip).  cell_a: the input's cell; a concatenated input whose channels [split_c, Cin) were converted with
// another cell passes that as cell_b (forward layout of w only), else split_c = Cin.  wcell: one zeroed-by-us cell for the weights,
// wp_ws >= s3x_packed_bytes(.., 2).
int conv_s3x_h2(const void* xs, const unsigned* cell_a, const unsigned* cell_b, int split_c, const float* w, const float* bias, float* y, int N,
                int Cin, int D, int H, int W, int Kout, int KS, long so, long si, int flip, unsigned* wcell, void* wp_ws, hipStream_t s,
                const unsigned* guard, float* stats_part) {
  const bool k32 = Kout == 32;  // one 32-channel slice per tile (k_conv_s3x K32): 5^3, no epilogue statistics
  if (k32 && (KS != 5 || stats_part || guard)) { set_error("conv_s3x_h2: 32 output channels for unguarded 5^3 launches only"); return NC_ERR_SHAPE; }
  const XPlan pl = x_plan(N, D, H, W, k32 ? 1 : Kout / 64, KS, 2, k32);
  if (!pl.ok) { set_error("conv_s3x_h2: shape not covered"); return NC_ERR_SHAPE; }
  if (split_c < Cin && (flip || !cell_b || split_c % 8)) { set_error("conv_s3x_h2: scale groups only for the forward weight layout"); return NC_ERR_ARG; }
  if (stats_part && KS != 3) { set_error("conv_s3x_h2: epilogue statistics exist for the 3^3 layers only (the layers in front of an InstanceNorm)"); return NC_ERR_ARG; }
  const int NCH = Cin / 8, NS = s3x_ksteps(Cin, KS), T3 = KS * KS * KS;
  if (Cin % 32) { set_error("conv_s3x_h2: channels must be a multiple of 32"); return NC_ERR_SHAPE; }
  if (int e = h2_zero_cells(wcell, 1, s)) return e;
  const long nw = (long)Kout * Cin * T3;
  if (!cell_b) cell_b = cell_a;
  // (the group test below reads the input channel as (i / T3) % Cin: the forward layout [co][ci][tap]; data gradients have no groups)
  hipLaunchKernelGGL(k_absmax_w, dim3((unsigned)(cdiv(nw, 256 * 8) < 256 ? cdiv(nw, 256 * 8) : 256)), dim3(256), 0, s, w, nw, T3, Cin,
                     flip ? Cin : split_c, cell_a, cell_b, wcell);
  const long total = (long)(s3x_packed_bytes(Cin, Kout, KS, 2) / 2);
  hipLaunchKernelGGL(k_pack_w_s3x<2>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (unsigned short*)wp_ws, NCH, KS, NS, so, si, flip, total,
                     (const unsigned*)wcell, flip ? Cin : split_c, cell_a, cell_b);
  if (int e = check_launch("conv_s3x_h2 pack")) return e;
  XParams p{};
  p.xs = (const uint4*)xs; p.wp = (const uint4*)wp_ws; p.bias = bias; p.y = y; p.amax_x = cell_a; p.amax_w = wcell; p.guard = guard;
  p.N = N; p.NCH = NCH; p.D = D; p.H = H; p.W = W; p.K = Kout;
  p.P = pl.P; p.HP = pl.HP; p.TPP = pl.TPP; p.KT = k32 ? 1 : Kout / 64;
  p.NS = NS; p.mP = magic(pl.P);
  static const int flush = getenv("NC_S3X_FLUSH") ? atoi(getenv("NC_S3X_FLUSH")) : 4;
  p.flush = flush >= 4 ? flush : flush > 0 ? 4 : 1 << 30;
  // With the W64 form switched on, EVERY two-term 3^3 launch keeps one running accumulator per tile (k_conv_s3w has no second set to restart
  // into): an output element's bits then do not depend on which kernel or which launch (whole tiles / the fractional tiles of the tail, and so
  // on the batch a cube travels in) its tile fell to -- same products, same order, tot = 0 + acc exactly.
  if (KS == 3 && !k32 && s3x_w64_get()) p.flush = 1 << 30;
  const bool one = pl.rem && pl.fsub == 1;
  if (pl.full || one) {
    p.fsub = 1; p.UB = pl.UB; p.npb = pl.npb; p.mUB = magic(pl.UB);
    p.t_begin = 0; p.t_count = (int)(pl.full + (one ? pl.rem : 0)); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    p.stats = (float2*)stats_part;
    const int e = !k32 && w64_applies(KS, pl.NCB, NS, pl.lds) ? (stats_part ? launch_w64<true>(p, pl.lds, s) : launch_w64<false>(p, pl.lds, s))
                  : k32 ? (pl.NCB == 4 ? launch_x<5, 4, 2, false, true>(p, pl.lds + kOffTab, s) : launch_x<5, 2, 2, false, true>(p, pl.lds + kOffTab, s))
                  : stats_part ? (KS == 3 ? launch_x_ncb<3, 2, true>(pl.NCB, p, pl.lds + kOffTab, s) : launch_x_ncb<5, 2>(pl.NCB, p, pl.lds + kOffTab, s))
                             : (KS == 3 ? launch_x_ncb<3, 2>(pl.NCB, p, pl.lds + kOffTab, s) : launch_x_ncb<5, 2>(pl.NCB, p, pl.lds + kOffTab, s));
    if (e) return e;
  }
  if (pl.rem && !one) {
    p.fsub = pl.fsub; p.UB = pl.UBt; p.npb = pl.npbt; p.mUB = magic(pl.UBt);
    p.t_begin = (int)pl.full; p.t_count = (int)(pl.rem * pl.fsub); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    p.stats = stats_part ? (float2*)stats_part + pl.full * kWaves * 32 : nullptr;
    const int e = k32 ? (pl.NCB / pl.fsub == 4 ? launch_x<5, 4, 2, false, true>(p, pl.ldst + kOffTab, s) : launch_x<5, 2, 2, false, true>(p, pl.ldst + kOffTab, s))
                  : stats_part ? (KS == 3 ? launch_x_ncb<3, 2, true>(pl.NCB / pl.fsub, p, pl.ldst + kOffTab, s)
                                        : launch_x_ncb<5, 2>(pl.NCB / pl.fsub, p, pl.ldst + kOffTab, s))
                             : (KS == 3 ? launch_x_ncb<3, 2>(pl.NCB / pl.fsub, p, pl.ldst + kOffTab, s)
                                        : launch_x_ncb<5, 2>(pl.NCB / pl.fsub, p, pl.ldst + kOffTab, s));
    if (e) return e;
  }
  return NC_OK;
}

// ---- Conv3d(1, 64, 7, padding 3) forward on the two-term kernels (k_conv_s3x PC).  Workspace: [256 B cells: x, w | 256 B guard words | X8 in H2
// form | packed weights].  The input's cell is MEASURED (k_absmax), so the range guard counts its chunks; a flagged tensor cannot switch kernels
// here (there is no three-term pseudo-channel kernel): it is reported in nc_h2_guard_stats [2] like every counted-only tensor, and callers that
// act on the report (BaseModel.get_current_losses -> nc_set_split_terms(3)) send this layer back to the fp32 matrix kernel.
static size_t c1k7_al(size_t b) { return (b + 255) & ~(size_t)255; }
static constexpr int kC1k7Steps = 14;  // 7 bricks of eight (dz, dy) tap slots, in fours
static size_t c1k7_packed_bytes() { return (size_t)2 * kC1k7Steps * 4 * 1024; }
bool c1k7_h2_supported(const ConvDims& d) {
  static const int on = getenv("NC_C1K7_H2") ? atoi(getenv("NC_C1K7_H2")) : 1;  // NC_C1K7_H2=0: the fp32 matrix kernel (A/B)
  if (!on || s3x_get_terms() != 2) return false;
  if (d.C != 1 || d.K != 64 || d.kd != 7 || d.kh != 7 || d.kw != 7 || d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != 3 || d.ph != 3 || d.pw != 3) return false;
  const long S = (long)d.D * d.H * d.W;
  if (d.W < 4 || (long)64 * S * 4 >= (1l << 31) || S * 32 >= (1l << 31)) return false;
  return x_plan(d.N, d.D, d.H, d.W, 1, 7, 2, false, true).ok;
}
size_t c1k7_h2_ws_bytes(const ConvDims& d) {
  if (!c1k7_h2_supported(d)) return 0;
  return 512 + c1k7_al((size_t)d.N * d.D * d.H * d.W * 32) + c1k7_al(c1k7_packed_bytes());
}
template <int NCB>
static int launch_pc(const XParams& p, int lds, hipStream_t s) { return launch_x<7, NCB, 2, false, false, 1>(p, lds, s); }
static int launch_pc_ncb(int NCB, const XParams& p, int lds, hipStream_t s) {
  if (NCB == 8) return launch_pc<8>(p, lds, s);
  if (NCB == 7) return launch_pc<7>(p, lds, s);
  if (NCB == 6) return launch_pc<6>(p, lds, s);
  if (NCB == 4) return launch_pc<4>(p, lds, s);
  return launch_pc<2>(p, lds, s);
}
int conv_c1k7_h2(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!c1k7_h2_supported(d)) { set_error("conv_c1k7_h2: shape not covered"); return NC_ERR_SHAPE; }
  if (!ws || wsb < c1k7_h2_ws_bytes(d)) { set_error("conv_c1k7_h2: workspace too small"); return NC_ERR_WS; }
  const XPlan pl = x_plan(d.N, d.D, d.H, d.W, 1, 7, 2, false, true);
  const long S = (long)d.D * d.H * d.W;
  unsigned* cells = (unsigned*)ws;                 // [0] x, [1] w
  unsigned* gw = (unsigned*)((char*)ws + 256);     // guard words of x
  uint4* x8 = (uint4*)((char*)ws + 512);
  void* wp = (char*)x8 + c1k7_al((size_t)d.N * S * 32);
  if (int e = h2_zero_cells(cells, 2, s)) return e;
  if (int e = h2_absmax(x, (long)d.N * S, cells, s)) return e;
  if (int e = h2_absmax(w, (long)64 * 343, cells + 1, s)) return e;
  const bool guard = h2_guard_on();
  if (guard) if (int e = h2_guard_zero(gw, s, 4)) return e;
  {
    long bx = cdiv(S, 256);
    if (guard && bx * d.N > 4096) bx = cdiv(4096, d.N);  // (few atomics: h2.hip split2h_into)
    hipLaunchKernelGGL(k_build_x8_h2, dim3((unsigned)bx, (unsigned)d.N), dim3(256), 0, s, x, x8, S, d.W, (const unsigned*)cells, guard ? gw : nullptr);
  }
  if (guard) if (int e = h2_guard_decide(gw, nullptr, nullptr, gw + kGuardFlag, false, s)) return e;  // counted, never switched (see above)
  const long total = (long)(c1k7_packed_bytes() / 2);
  hipLaunchKernelGGL(k_pack_w_c1k7, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (unsigned short*)wp, kC1k7Steps, total, (const unsigned*)(cells + 1));
  if (int e = check_launch("conv_c1k7_h2 prep")) return e;
  XParams p{};
  p.xs = x8; p.wp = (const uint4*)wp; p.bias = bias; p.y = y; p.amax_x = cells; p.amax_w = cells + 1; p.guard = nullptr;
  p.N = d.N; p.NCH = 1; p.D = d.D; p.H = d.H; p.W = d.W; p.K = 64;
  p.P = pl.P; p.HP = pl.HP; p.TPP = pl.TPP; p.KT = 1;
  p.NS = kC1k7Steps; p.mP = magic(pl.P);
  p.flush = 4;
  const bool one = pl.rem && pl.fsub == 1;
  if (pl.full || one) {
    p.fsub = 1; p.UB = pl.UB; p.npb = pl.npb; p.mUB = magic(pl.UB);
    p.t_begin = 0; p.t_count = (int)(pl.full + (one ? pl.rem : 0)); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    if (int e = launch_pc_ncb(pl.NCB, p, pl.lds + kOffTab, s)) return e;
  }
  if (pl.rem && !one) {
    p.fsub = pl.fsub; p.UB = pl.UBt; p.npb = pl.npbt; p.mUB = magic(pl.UBt);
    p.t_begin = (int)pl.full; p.t_count = (int)(pl.rem * pl.fsub); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    if (int e = launch_pc_ncb(pl.NCB / pl.fsub, p, pl.ldst + kOffTab, s)) return e;
  }
  return NC_OK;
}

// ---- the data gradient of that layer: dX from the fp32 dY [N][64][voxels] (converted here: measured cell, guard counted as above).
// Workspace: [256 B cells: dY, w | 256 B guard words | dY in H2 form | packed weights | Z: 49 planes per sample, fp32].
static constexpr int kC1k7dSteps = 16;  // 8 channel blocks x eight row-tap slots, in fours
static size_t c1k7d_packed_bytes() { return (size_t)2 * kC1k7dSteps * 4 * 1024; }
size_t c1k7_h2_dgrad_ws_bytes(const ConvDims& d) {
  if (!c1k7_h2_supported(d)) return 0;
  const size_t S = (size_t)d.D * d.H * d.W;
  return 512 + c1k7_al((size_t)d.N * 64 * S * 4) + c1k7_al(c1k7d_packed_bytes()) + c1k7_al((size_t)d.N * 49 * S * 4);
}
template <int NCB>
static int launch_pcd(const XParams& p, int lds, hipStream_t s) { return launch_x<7, NCB, 2, false, false, 2>(p, lds, s); }
static int launch_pcd_ncb(int NCB, const XParams& p, int lds, hipStream_t s) {
  if (NCB == 8) return launch_pcd<8>(p, lds, s);
  if (NCB == 7) return launch_pcd<7>(p, lds, s);
  if (NCB == 6) return launch_pcd<6>(p, lds, s);
  if (NCB == 4) return launch_pcd<4>(p, lds, s);
  return launch_pcd<2>(p, lds, s);
}
int conv_c1k7_h2_dgrad(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  if (!c1k7_h2_supported(d)) { set_error("conv_c1k7_h2_dgrad: shape not covered"); return NC_ERR_SHAPE; }
  if (!ws || wsb < c1k7_h2_dgrad_ws_bytes(d)) { set_error("conv_c1k7_h2_dgrad: workspace too small"); return NC_ERR_WS; }
  const XPlan pl = x_plan(d.N, d.D, d.H, d.W, 1, 7, 2, false, true);
  const long S = (long)d.D * d.H * d.W;
  unsigned* cells = (unsigned*)ws;
  unsigned* gw = (unsigned*)((char*)ws + 256);
  char* dyh = (char*)ws + 512;
  char* wp = dyh + c1k7_al((size_t)d.N * 64 * S * 4);
  float* Z = (float*)(wp + c1k7_al(c1k7d_packed_bytes()));
  if (int e = h2_zero_cells(cells, 2, s)) return e;
  if (int e = h2_absmax(dy, (long)d.N * 64 * S, cells, s)) return e;
  if (int e = h2_absmax(w, (long)64 * 343, cells + 1, s)) return e;
  const bool guard = h2_guard_on();
  if (guard) if (int e = h2_guard_zero(gw, s, 4)) return e;
  if (int e = split2h_into(dy, 64 * S, dyh, d.N, 64, S, 64, 0, cells, s, guard ? gw : nullptr)) return e;
  if (guard) if (int e = h2_guard_decide(gw, nullptr, nullptr, gw + kGuardFlag, false, s)) return e;  // counted, never switched
  const long total = (long)(c1k7d_packed_bytes() / 2);
  hipLaunchKernelGGL(k_pack_w_c1k7d, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (unsigned short*)wp, kC1k7dSteps, total, (const unsigned*)(cells + 1));
  if (int e = check_launch("conv_c1k7_h2_dgrad prep")) return e;
  XParams p{};
  p.xs = (const uint4*)dyh; p.wp = (const uint4*)wp; p.bias = nullptr; p.y = Z; p.amax_x = cells; p.amax_w = cells + 1; p.guard = nullptr;
  p.N = d.N; p.NCH = 8; p.D = d.D; p.H = d.H; p.W = d.W;
  p.K = 49;  // (rows 49 .. 63 of the tile store beyond the descriptor's range: dropped by the hardware)
  p.P = pl.P; p.HP = pl.HP; p.TPP = pl.TPP; p.KT = 1;
  p.NS = kC1k7dSteps; p.mP = magic(pl.P);
  p.flush = 4;
  const bool one = pl.rem && pl.fsub == 1;
  if (pl.full || one) {
    p.fsub = 1; p.UB = pl.UB; p.npb = pl.npb; p.mUB = magic(pl.UB);
    p.t_begin = 0; p.t_count = (int)(pl.full + (one ? pl.rem : 0)); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    if (int e = launch_pcd_ncb(pl.NCB, p, pl.lds + kOffTab, s)) return e;
  }
  if (pl.rem && !one) {
    p.fsub = pl.fsub; p.UB = pl.UBt; p.npb = pl.npbt; p.mUB = magic(pl.UBt);
    p.t_begin = (int)pl.full; p.t_count = (int)(pl.rem * pl.fsub); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    if (int e = launch_pcd_ncb(pl.NCB / pl.fsub, p, pl.ldst + kOffTab, s)) return e;
  }
  hipLaunchKernelGGL(k_fold_c1k7, dim3((unsigned)cdiv(S, 256), (unsigned)d.N), dim3(256), 0, s, (const float*)Z, dx, d.D, d.H, d.W);
  return check_launch("conv_c1k7_h2_dgrad fold");
}

// ST epilogue (see k_conv_s3x): bytes of the partial-sum records of one two-term launch pair, and the pass that turns them into mean / rstd
// (mean gets the bias back: the records are sums of bias-free outputs).  The geometry is recomputed from the same planner the launch used.
static size_t s3x_record_bytes(const XPlan& pl) {
  const bool one = pl.rem && pl.fsub == 1;
  const long recs = pl.full + (one ? pl.rem : pl.rem * pl.fsub);
  return ((size_t)recs * kWaves * 32 * sizeof(float2) + 255) & ~(size_t)255;
}
size_t s3x_stats_bytes(int N, int D, int H, int W, int Kout, int KS) {
  const XPlan pl = x_plan(N, D, H, W, Kout / 64, KS, 2);
  if (!pl.ok) return 0;
  return s3x_record_bytes(pl) + (size_t)N * Kout * kStatChunks * sizeof(double2);  // records, then the chunk sums of the finalisation
}
int s3x_stats_finalize(const float* stats_part, const float* bias, int N, int D, int H, int W, int Kout, int KS, float eps, float* mean, float* rstd,
                       hipStream_t s) {
  const XPlan pl = x_plan(N, D, H, W, Kout / 64, KS, 2);
  if (!pl.ok || !stats_part) { set_error("s3x_stats_finalize: shape not covered"); return NC_ERR_SHAPE; }
  const bool one = pl.rem && pl.fsub == 1;
  XStatsPlan sp{};
  sp.N = N; sp.K = Kout; sp.KT = Kout / 64; sp.TPP = pl.TPP; sp.D = D; sp.HP = pl.HP;
  sp.full = pl.full; sp.main_count = pl.full + (one ? pl.rem : 0);
  sp.fsub = one ? 1 : pl.fsub; sp.PTsub = 64 * pl.NCB / (one ? 1 : pl.fsub);
  double2* chunk_sums = (double2*)((char*)const_cast<float*>(stats_part) + s3x_record_bytes(pl));
  hipLaunchKernelGGL(k_s3x_stats_partial, dim3((unsigned)(Kout / 32), (unsigned)N, kStatChunks), dim3(256), 0, s, (const float2*)stats_part, sp, chunk_sums);
  hipLaunchKernelGGL(k_s3x_stats_final, dim3((unsigned)cdiv(Kout, 256), (unsigned)N), dim3(256), 0, s, (const double2*)chunk_sums, Kout, bias, (long)D * H * W,
                     eps, mean, rstd);
  return check_launch("s3x_stats_finalize");
}

// xs: S3 input; wp_ws: >= s3x_packed_bytes scratch for the packed weights
int conv_s3x(const void* xs, const float* w, const float* bias, float* y, int N, int Cin, int D, int H, int W, int Kout, int KS, long so,
             long si, int flip, void* wp_ws, hipStream_t s, const unsigned* guard) {
  const XPlan pl = x_plan(N, D, H, W, Kout / 64, KS);
  if (!pl.ok) { set_error("conv_s3x: shape not covered"); return NC_ERR_SHAPE; }
  const int NCH = Cin / 8, NS = s3x_ksteps(Cin, KS);
  const long total = (long)(s3x_packed_bytes(Cin, Kout, KS) / 2);
  hipLaunchKernelGGL(k_pack_w_s3x<3>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, (unsigned short*)wp_ws, NCH, KS, NS, so, si, flip,
                     total, (const unsigned*)nullptr, Cin, (const unsigned*)nullptr, (const unsigned*)nullptr);
  if (int e = check_launch("pack_w_s3x")) return e;
  XParams p{};
  p.xs = (const uint4*)xs; p.wp = (const uint4*)wp_ws; p.bias = bias; p.y = y; p.guard = guard;
  p.N = N; p.NCH = NCH; p.D = D; p.H = H; p.W = W; p.K = Kout;
  p.P = pl.P; p.HP = pl.HP; p.TPP = pl.TPP; p.KT = Kout / 64;
  p.NS = NS; p.mP = magic(pl.P);
#ifdef NC_S3X_STAMP
  p.dbg = (long long*)((char*)wp_ws + s3x_packed_bytes(Cin, Kout, KS));  // (the workspace has slack behind the packed weights in the timing tool)
#endif
  static const int flush = getenv("NC_S3X_FLUSH") ? atoi(getenv("NC_S3X_FLUSH")) : 4;
  p.flush = flush >= 4 ? flush : flush > 0 ? 4 : 1 << 30;  // >= 4: the previous tile's sums leave `tot` during the first four k-steps
  const bool one = pl.rem && pl.fsub == 1;  // the left-over tiles are whole tiles: one launch, some workgroups take one tile more
  if (pl.full || one) {
    p.fsub = 1; p.UB = pl.UB; p.npb = pl.npb; p.mUB = magic(pl.UB);
    p.t_begin = 0; p.t_count = (int)(pl.full + (one ? pl.rem : 0)); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    const int e = KS == 3 ? launch_x_ncb<3>(pl.NCB, p, pl.lds, s) : launch_x_ncb<5>(pl.NCB, p, pl.lds, s);
    if (e) return e;
  }
  if (pl.rem && !one) {
    p.fsub = pl.fsub; p.UB = pl.UBt; p.npb = pl.npbt; p.mUB = magic(pl.UBt);
    p.t_begin = (int)pl.full; p.t_count = (int)(pl.rem * pl.fsub); p.tiles_per_xcd = (int)cdiv(p.t_count, 8);
    const int e = KS == 3 ? launch_x_ncb<3>(pl.NCB / pl.fsub, p, pl.ldst, s) : launch_x_ncb<5>(pl.NCB / pl.fsub, p, pl.ldst, s);
    if (e) return e;
  }
  return NC_OK;
}

}  // namespace nc
