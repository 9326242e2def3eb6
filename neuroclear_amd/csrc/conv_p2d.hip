// fp32 Conv2d 4 x 4, padding 1, stride 1 and 2 -- the PatchGAN's 256 -> 512, 128 -> 256 and 64 -> 128 layers (models/networks.py:1037-1055;
// all of a discriminator pass' convolution FLOPs but the one-channel first layer and head) -- forward and data gradient at Athena's batches
// (108-216 planes of 13^2 .. 54^2), on the bf16 matrix cores with the exact three-term operand split of conv_split.hip: the tap-stream kernel
// of conv_s3x.hip carried over to a FLAT BATCH of small planes.
//
//  * the input is converted once per call into the S3 form of a zero-padded flat batch: [brick][term][(sub-brick)][b][Hp][Wp] units, so a
//    tap is a constant unit offset and a brick is a contiguous range: no padding logic in the kernel.  Stride 1: Hp = H + 2 pad (forward pad
//    1; data gradient = the same convolution of dy with flipped, channel-transposed weights and pad 2), brick = one 8-channel chunk, tap
//    (dy, dx) = offset dy * Wp + dx.  Stride 2 forward: a space-to-depth image of the padded input -- 4 x 4 stride 2 is 2 x 2 stride 1
//    over 4 parities; brick = (chunk, row parity) with the two column parities as sub-bricks.  Stride 2 data gradient: output parity class
//    (ry, rx) of dx is a 2 x 2 stride-1 convolution of dy (padded by 1, window base shifted by (ry, rx)) with the taps (3 - 2a - ry,
//    3 - 2b - rx); brick = two chunks of dy channels as sub-bricks; four launches over ONE converted dy;
//  * output positions are ONLY the valid ones, f = (b, y < Ho, x < Wo) flattened -- enumerating the padded grid instead would spend 36 % of
//    the MFMAs on dropped positions at 13 x 13.  A lane keeps the LDS offset of its position in every column block (NCB registers per tile);
//    a tile's brick spans the planes its positions touch;
//  * the K-dim is one stream of taps: a brick is 16 taps = four k-steps (stride 1: k-step j = tap row j, lane group g = tap column g) or
//    8 taps = two k-steps (stride 2: k-step j = sub-brick j, lane group g = tap (g / 2, g % 2)), bricks in a ring of three filled by
//    LDS-DMA, weights streamed into registers one k-step ahead, deferred stores, the hand-counted waits -- all as in conv_s3x.hip, which see.
// Measured at 216 planes (tools/p2d_check.py; k_sconv = the image-staged fp32 MFMA kernel of conv2d_img.hip): 256 -> 512 forward 0.66 ms
// (1.09), data gradient 0.83 (1.49); 128 -> 256 stride 2 0.33 (0.43) / 0.33 (0.46); 64 -> 128 stride 2 0.37 (0.42) / 0.39 (0.48) -- the
// stride-2 figures include the space-to-depth / padding conversion, a third of their time at 64 channels; error against fp64 2e-7 of the rms
// (k_sconv: 5e-7 .. 1.5e-6).  Athena step 101.7 -> 87.2 ms.  The weight gradients of these layers stay on k_swgrad.
#include <cstdlib>
#include <type_traits>

#include <atomic>

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
#ifndef NC_P2D_DMAW
#define NC_P2D_DMAW 4
#endif
// Waves that issue the LDS-DMA pieces of a brick: the first kDmaWaves of the 8.  4 = one wave of each SIMD pair (waves w and w + 4
// share a SIMD): issuing ~10 pieces keeps a wave away from the matrix pipe for ~1,000 cycles, which its partner -- with no pieces of
// its own -- fills with MFMAs; with all 8 issuing, both partners are away at the same moment right behind the barrier.
constexpr int kDmaWaves = NC_P2D_DMAW;
constexpr int kMaxPieces = 56;  // 1 KiB pieces of a brick (three terms)

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m) { return __umulhi(n, m); }
unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

// (the split itself: s3_common.hpp)
__device__ __forceinline__ void split3(float v, unsigned short (&t)[3]) { s3_split(v, t); }

// Packed weights: [cot = co/64][half = (co/32)%2][k-step s][f = rb*3 + term][lane][8] bf16.  Lane l = (g = l/16, m = l%16) holds
// output channel cot*64 + half*32 + rb*16 + m at tap T = 4s + g of the tile's tap stream; element j = input channel chunk*8 + (g odd ?
// (j + 4) % 8 : j)  (the B fragment of an odd lane group is read upper half first).  (chunk, kernel tap) of stream tap T:
//   mode 0 (4 x 4, stride 1):        chunk T / 16, tap T % 16 = (dy, dx) = (T % 16 / 4, g)            (flip: 15 - tap, the data gradient)
//   mode 1 (stride 2 forward, S2D):  brick T / 8 = (chunk, row parity py), sub-brick j = T / 4 % 2 = column parity px, g = (a, b):
//                                    tap (2a + py, 2b + px)
//   mode 2 (stride 2 data gradient, output parity class (ry, rx)): brick T / 8 = pair of dy chunks, chunk 2 * brick + j, g = (a, b):
//                                    tap (3 - 2a - ry, 3 - 2b - rx)  (dx[2u + ry] = sum_a w[.., 3 - 2a - ry] dy[u + ry + a - 1])
// w element = w[co * so + ci * si + tap]: forward so = C*16, si = 16; data gradients so = 16, si = C*16 (co = the layer's input channel).
// NT = 2: the two fp16 terms of w * 2^k (s3_common.hpp h2_split; k from the weights' cell).
template <int NT>
__global__ void __launch_bounds__(256) k_pack_w_p2d(const float* __restrict__ w, unsigned short* __restrict__ wp, int NS, long so, long si,
                                                    int mode, int flip, int ry, int rx, long total, const unsigned* __restrict__ wcell) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
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
  int chunk, tap;
  if (mode == 0) {
    chunk = T >> 4; tap = T & 15;
    if (flip) tap = 15 - tap;
  } else {
    const int bi = T >> 3, sub = (T >> 2) & 1, a = g >> 1, b = g & 1;
    if (mode == 1) { chunk = bi >> 1; tap = (2 * a + (bi & 1)) * 4 + 2 * b + sub; }
    else { chunk = 2 * bi + sub; tap = (3 - 2 * a - ry) * 4 + (3 - 2 * b - rx); }
  }
  const int jj = (g & 1) ? ((j + 4) & 7) : j;
  const long co = cot * 64 + half * 32 + rb * 16 + m, ci = chunk * 8 + jj;
  unsigned short t[3];
  if constexpr (NT == 3) split3(w[co * so + ci * si + tap], t);
  else h2_split(w[co * so + ci * si + tap] * h2_scale(*wcell), t);
  wp[i] = t[term];
}

// fp32 NCHW [B][C][H][W] -> the S3 form of the zero-padded flat batch, PP = Hp*Wp, Hp = H + 2 pad.  pairs = 0: unit ((chunk*3 + term) *
// TOT + i), i = b*PP + yp*Wp + xp; pairs = 1 (stride-2 data gradient): bricks of two chunks, unit (((chunk/2)*3 + term)*2 + chunk%2) * TOT + i.
// One thread per (chunk, padded position): 8 strided loads (coalesced along x), three 16-byte stores.
template <int NT>
__global__ void __launch_bounds__(256) k_pad_split3_2d(const float* __restrict__ x, uint4* __restrict__ xs, int C, int H, int W, int pad, int Hp,
                                                       int Wp, long TOT, long npos, int pairs, const unsigned* __restrict__ cell) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // b*PP + yp*Wp + xp
  if (i >= npos) return;
  const int chunk = blockIdx.y;
  const int PP = Hp * Wp;
  const int b = (int)(i / PP), r = (int)(i - (long)b * PP);
  const int yp = r / Wp, xp = r - yp * Wp;
  const int yy = yp - pad, xx = xp - pad;
  unsigned short e[8][3];
  const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
  const float* src = x + (((long)b * C + chunk * 8) * H + (in ? yy : 0)) * W + (in ? xx : 0);
  float sc = 1.f;
  if constexpr (NT == 2) sc = h2_scale(*cell);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = in ? src[(long)j * H * W] : 0.f;
    if constexpr (NT == 3) s3_split(v, e[j]);
    else h2_split(v * sc, e[j]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const long u = pairs ? (((long)(chunk >> 1) * NT + t) * 2 + (chunk & 1)) * TOT + i : ((long)chunk * NT + t) * TOT + i;
    xs[u] = s3_unit(e, t);
  }
}

// Space-to-depth of the zero-padded (pad 1, extended to even extents) input of a stride-2 layer: parity (py, px) of chunk c is sub-brick px of
// brick 2c + py, position (yq, xq) of plane b = x_pad[2 yq + py][2 xq + px].  grid.y = chunk*4 + parity.
template <int NT>
__global__ void __launch_bounds__(256) k_s2d_split3(const float* __restrict__ x, uint4* __restrict__ xs, int C, int H, int W, int Hq, int Wq,
                                                    long TOT, long npos, const unsigned* __restrict__ cell) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // b*PPq + yq*Wq + xq
  if (i >= npos) return;
  const int chunk = blockIdx.y >> 2, py = (blockIdx.y >> 1) & 1, px = blockIdx.y & 1;
  const int PP = Hq * Wq;
  const int b = (int)(i / PP), r = (int)(i - (long)b * PP);
  const int yq = r / Wq, xq = r - yq * Wq;
  const int yy = 2 * yq + py - 1, xx = 2 * xq + px - 1;
  unsigned short e[8][3];
  const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
  const float* src = x + (((long)b * C + chunk * 8) * H + (in ? yy : 0)) * W + (in ? xx : 0);
  float sc = 1.f;
  if constexpr (NT == 2) sc = h2_scale(*cell);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = in ? src[(long)j * H * W] : 0.f;
    if constexpr (NT == 3) s3_split(v, e[j]);
    else h2_split(v * sc, e[j]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) xs[(((long)(chunk * 2 + py) * NT + t) * 2 + px) * TOT + i] = s3_unit(e, t);
}

struct PParams {
  const uint4* xs;     // S3 input of the padded flat batch [NCH][3][TOT] units
  const uint4* wp;     // packed weights
  const float* bias;   // nullable
  float* y;            // fp32 [B][K][Ho][Wo] output
  int B, NB, K;        // planes, bricks per tile, output channels
  int Wp, PP;          // padded row pitch, padded plane size Hp * Wp
  int qbase;           // offset of tap (0, 0) of position (0, 0) inside a plane (the parity classes of a stride-2 data gradient)
  int UBq;             // M1: units per sub-brick (UB = 2 UBq)
  long sB, sC, sY, sX, obase;  // output strides (elements) of plane, channel, row, column; offset of position (0, 0)
  long out_elems;      // size of the whole output
  int Ho, Wo, HoWo;    // valid output plane
  long TOT;            // units per (chunk, term): B * PP (+ slack)
  long npos;           // output positions B * Ho * Wo
  int NPT, KT;         // position tiles, K / 64
  int UB;              // units per term of a brick (multiple of 64)
  int npb;             // 1 KiB pieces per brick (three terms)
  int NS;              // k-steps per tile: 4 per brick (16 taps), M1: 2 per brick (8 taps)
  unsigned mUB, mUBq, mHoWo, mWo;
  int t_count, tiles_per_xcd;
  int flush;           // k-steps between two accumulator restarts
  const unsigned *amax_x, *amax_w;  // NT = 2: the cells of the input and of the weights (results scaled back by 2^-(kx + kw))
};

struct PTile {
  int cot, f0;  // output-channel tile, first output position
  int q0;       // padded flat position of the brick's first unit = that of position f0
};

// padded flat position of output position f (its tap (0, 0))
__device__ __forceinline__ unsigned p_q(const PParams& p, unsigned f) {
  const unsigned b = fdiv(f, p.mHoWo), r = f - b * p.HoWo;
  const unsigned yy = fdiv(r, p.mWo), xx = r - yy * p.Wo;
  return b * p.PP + yy * p.Wp + xx + p.qbase;
}

template <int PT>
__device__ __forceinline__ PTile p_decode(const PParams& p, int t) {
  PTile o;  // output-channel tile fastest: the KT workgroups that share a position tile's bricks run together
  o.cot = t % p.KT;
  o.f0 = (t / p.KT) * PT;
  o.q0 = (int)p_q(p, (unsigned)o.f0);
  o.cot = __builtin_amdgcn_readfirstlane(o.cot); o.f0 = __builtin_amdgcn_readfirstlane(o.f0); o.q0 = __builtin_amdgcn_readfirstlane(o.q0);
  return o;
}

// M1 = false: 4 x 4 taps of ONE padded plane set per brick (k-step j of a brick = tap row j, lane group g = tap column g).
// M1 = true:  2 x 2 taps over TWO sub-bricks per brick (k-step j of a brick = sub-brick j, lane group g = tap (g / 2, g % 2)): the stride-2
//             layers -- forward over a space-to-depth image (sub-bricks = the two column parities of one row parity of a chunk), data
//             gradient per output-parity class over dy (sub-bricks = two consecutive chunks of dy channels).
template <int NCB, bool M1, int NT = 3>
__global__ void __launch_bounds__(kThreads, 1) k_conv_p2d(const PParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  constexpr int T2 = M1 ? 8 : 16, PT = 64 * NCB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  const int half = wave & 1, pg = wave >> 1;
  const int NB = p.NB;           // bricks per tile
  const int BB = p.npb * 1024;   // bytes per ring slot

  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int t_lo = xcd * p.tiles_per_xcd;
  int t_hi = t_lo + p.tiles_per_xcd;
  if (t_hi > p.t_count) t_hi = p.t_count;
  // the tiles of this workgroup: t_lo + wslot, + nslot, ...
  auto next_tile = [&](int t, PTile& o) __attribute__((always_inline)) {
    if (t >= t_hi) return -1;
    o = p_decode<PT>(p, t);
    return t;
  };
  PTile cur, nxt;
  int tcur = next_tile(t_lo + wslot, cur);
  if (tcur < 0) return;

  // ---- brick staging: unit u of a term = padded flat position q0 + u of chunk `bi`.  The DMA goes through a buffer descriptor over the
  // three terms of ONE chunk; a unit beyond the batch asks for an out-of-range offset and the hardware delivers zeros
  constexpr unsigned kOut = 0x80000000u;
  auto issue_brick = [&](const PTile& t, int bi, int slot) __attribute__((always_inline)) {
    if (wave >= kDmaWaves) return;
    constexpr int NSUB = M1 ? 2 : 1;
    const uint4* blk = p.xs + (long)bi * NT * NSUB * p.TOT;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(blk), 0, (unsigned)(NT * NSUB * p.TOT * 16), 0x00020000);
    unsigned char* buf = lds_raw + slot * BB;
#pragma unroll 1
    for (int pc = wave; pc < p.npb; pc += kDmaWaves) {
      const unsigned u = (unsigned)(pc * 64 + lane);
      const unsigned term = fdiv(u, p.mUB);
      unsigned within = u - term * p.UB, sub = 0;
      if (M1) { sub = fdiv(within, p.mUBq); within -= sub * p.UBq; }
      const unsigned F = (unsigned)t.q0 + within;
      const bool ok = term < (unsigned)NT && (long)F < p.TOT;
      const unsigned po = ok ? (unsigned)((term * NSUB + sub) * (unsigned)p.TOT + F) * 16u : kOut;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(buf + pc * 1024), 16, po, 0, 0, 0);
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
  auto wtile = [&](int cot) __attribute__((always_inline)) { return ((cot * 2 + half) * p.NS) * (2 * NT * 1024); };
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

  // ---- B fragments: unit (slot, term, position + tap) of the ring, read as two 8-byte halves (odd lane groups: upper first).  The
  // positions of a tile are valid outputs only, so consecutive positions are NOT consecutive units across a row or plane end: a lane keeps
  // the byte offset of its position in every column block (qoff, rebuilt per tile)
  const unsigned term_b = (unsigned)p.UB * 16;
  const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)lds_raw;
  unsigned qoff[NCB];
  auto tile_offsets = [&](const PTile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      unsigned f = (unsigned)(t.f0 + pg * NCB * 16 + cb * 16 + m16);
      if ((long)f >= p.npos) f = (unsigned)t.f0;  // (beyond the batch: any valid unit; the result is dropped)
      qoff[cb] = (p_q(p, f) - (unsigned)t.q0) * 16u + (unsigned)((g & 1) * 8);
    }
  };
  struct BAddr { unsigned lo[NT]; };  // per term: slot + tap part of the address of the half read first
  auto b_addr = [&](unsigned vo) __attribute__((always_inline)) {
    BAddr a;
#pragma unroll
    for (int t = 0; t < NT; ++t) a.lo[t] = lds_base + vo + t * term_b;
    return a;
  };
  auto read_b = [&](u32x4 (&B)[NT], const BAddr& a, int cb) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const unsigned lo = a.lo[t] + qoff[cb];
      u64x2 v;
      v.x = *(lds64_t)(lo);  // volatile: two ds_read_b64, never one ds_read2_b64
      v.y = *(lds64_t)(lo ^ 8u);
      B[t] = __builtin_bit_cast(u32x4, v);
    }
  };

  // ---- results leave through a buffer descriptor over the whole output [B][K][Ho][Wo]: a lane whose position lies beyond the batch
  // stores to an out-of-range offset, which the hardware drops -- the NUMBER of store instructions is fixed,
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
  float oscx = 1.f, oscw = 1.f;  // NT = 2: the sums are scaled back by 2^-(kx + kw), as two factors of about equal exponent
  if constexpr (NT == 2) { const float2 f = h2_unscale2(*p.amax_x, *p.amax_w); oscx = f.x; oscw = f.y; }
  u32x4 bv[2];
  auto load_bias = [&](const PTile& t) __attribute__((always_inline)) {
    const int bo = (t.cot * 64 + half * 32 + 4 * g) * 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(bv[rb]) : "v"(bo + rb * 64), "s"(brsrc) : "memory");
  };
  auto store_pairs = [&](const PTile& t, const int p0, const int p1) __attribute__((always_inline)) {  // pairs p0 .. p1 - 1 of tile t from `tot`, then tot = 0
    const int cob = t.cot * 64 + half * 32 + 4 * g;
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (unsigned)(p.out_elems * 4), 0x00020000);
#pragma unroll
    for (int pr = 0; pr < kPairs; ++pr) {
      if (pr < p0 || pr >= p1) continue;
      const int rb = pr / NCB, cb = pr % NCB;
      const unsigned f = (unsigned)(t.f0 + pg * NCB * 16 + cb * 16 + m16);
      const unsigned b = fdiv(f, p.mHoWo), r = f - b * p.HoWo;
      const unsigned yy = fdiv(r, p.mWo), xx = r - yy * p.Wo;
      const bool ok = (long)f < p.npos;
      const unsigned vo0 = ok ? (unsigned)((b * p.sB + (cob + rb * 16) * p.sC + yy * p.sY + xx * p.sX + p.obase) * 4) : kOut;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned bu = bv[rb][e];  // (a bit_cast straight from the vector element reads element 0)
        const float v = NT == 2 ? tot[rb][cb][e] * oscx * oscw + __uint_as_float(bu) : tot[rb][cb][e] + __uint_as_float(bu);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ys, ok ? vo0 + (unsigned)(e * p.sC * 4) : kOut, 0, 0);
        tot[rb][cb][e] = 0.f;
      }
    }
  };

  // ---- prologue: brick 0 and the first A fragments of the first tile
  int ring = 0;  // ring slot of brick 0 of the current tile
  issue_brick(cur, 0, 0);
  tile_offsets(cur);
  u32x4 A[2][NT], nA[2][NT];
  load_a(A, wtile(cur.cot));
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[rb][cb][e] = 0.f; tot[rb][cb][e] = 0.f; }
  bool have_prev = false;
  PTile prv = cur;

#define STAMP() do {} while (0)
  while (true) {
    STAMP();
    PTile nx{};
    const int tnext = next_tile(tcur + nslot, nx);
    const bool more_tiles = tnext >= 0;
    nxt = nx;
    const int wt = wtile(cur.cot);
    const int wt_next = more_tiles ? wtile(nxt.cot) : wt;
    int na = 0;  // next brick of this tile to arrive (brick 0 was requested during the previous tile / in the prologue)

    // per-lane tap state: lane group g is at tap tpl of the brick in slot sl
    int tpl = g, sl = ring;
    auto b_off = [&]() __attribute__((always_inline)) {  // tap tpl of the brick in slot sl
      if (M1) return (unsigned)(sl * BB + ((tpl >> 2) * p.UBq + ((tpl >> 1) & 1) * p.Wp + (tpl & 1)) * 16);  // sub-brick tpl / 4, tap (a, b)
      return (unsigned)(sl * BB + ((tpl >> 2) * p.Wp + (tpl & 3)) * 16);                                    // tap (dy, dx) = (tpl / 4, tpl % 4)
    };
    BAddr vo = b_addr(b_off());
    u32x4 B[2][NT];
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
      wait_a(Ac, std::integral_constant<int, 4 * PPairs>{}, have_prev && ((PH >= 1 && PH < kStoreSteps) || s == kStoreSteps));
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
          issue_brick(cur, na + 1, (ring + na + 1) % 3);
        } else if (more_tiles) {
          issue_brick(nxt, 0, (ring + NB) % 3);
        }
#endif
        ++na;
      }
      if (last) load_bias(cur);  // (complete at the next step's wait: it is older than that step's 6 A requests)
      if constexpr (PH < kStoreSteps) {
        if (have_prev) {
          asm volatile("" : "+v"(bv[0]), "+v"(bv[1]));
          store_pairs(prv, PH * kPairsPerStep, (PH + 1) * kPairsPerStep);
        }
      }
      if constexpr (PH == 0) read_b(B[PAR], vo, 0);
      // next k-step's tap state
      tpl += 4;
      if (tpl >= T2) { tpl -= T2; sl = sl == 2 ? 0 : sl + 1; }
      const BAddr nvo = b_addr(b_off());
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        u32x4(&Bc)[NT] = B[(cb + PAR) & 1];
        u32x4(&Bn)[NT] = B[(cb + PAR + 1) & 1];
        if (cb + 1 < NCB) read_b(Bn, vo, cb + 1);
        else if (!last) read_b(Bn, nvo, 0);
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
        } else {  // the 4 reads of the next column block spread over this one's 6 MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
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
    have_prev = true;
    if (!more_tiles) break;
    ring = (ring + NB) % 3;
    cur = nxt;
    tcur = tnext;
    tile_offsets(cur);
  }
  // the last tile's results (and the dummy request of its last step)
  // (A is an operand on purpose: it holds the target registers of the last step's dummy request, which must not be handed to the epilogue's
  // temporaries before it has landed -- conv_s3x.hip's final wait has the story)
  if constexpr (NT == 3)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2])::"memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1])::"memory");
  store_pairs(prv, 0, kPairs);
}

struct P2Plan {
  int NCB, NPT, UBq, UB, npb, lds;
  long ntiles;
  bool ok;
};

long p2_q(long f, int HoWo, int Wo, int PP, int Wp) {
  const long b = f / HoWo, r = f - b * HoWo;
  return b * PP + (r / Wo) * Wp + r % Wo;
}

// position tiles of 64 NCB valid outputs; a tile's (sub-)brick spans from its first position's unit to its last position's last tap
// (tapext units further); nsub sub-bricks per brick
P2Plan p2_plan(long npos, int Ho, int Wo, int PP, int Wp, int KT, int tapext, int nsub, int NT = 3) {
  P2Plan best{};
  double best_cost = 1e30;
  static const int ncb_max = 8;
  for (int NCB : {8, 6, 4, 2}) {
    if (NCB > ncb_max) continue;
    const int PT = 64 * NCB;
    P2Plan pl{};
    pl.NCB = NCB;
    pl.NPT = (int)((npos + PT - 1) / PT);
    long umax = 0;
    for (int t = 0; t < pl.NPT; ++t) {
      const long f0 = (long)t * PT, f1 = (f0 + PT - 1 < npos ? f0 + PT - 1 : npos - 1);
      const long u = p2_q(f1, Ho * Wo, Wo, PP, Wp) - p2_q(f0, Ho * Wo, Wo, PP, Wp) + tapext;
      if (u > umax) umax = u;
    }
    const int align = 64 / nsub;  // the three terms of a brick are whole 1 KiB pieces
    pl.UBq = (int)((umax + align - 1) / align * align);
    pl.UB = nsub * pl.UBq;
    pl.npb = NT * pl.UB / 64;
    pl.lds = 3 * pl.npb * 1024;
    if (pl.npb > kMaxPieces || pl.lds > kLdsMax) continue;
    pl.ntiles = (long)pl.NPT * KT;
    const double rounds = (double)((pl.ntiles + 255) / 256);
    // a tile costs its positions plus a fixed part (prologue, epilogue, the brick's extra planes); narrow tiles pay twice the weight loads per MFMA
    const double cost = rounds * (PT + 48) * (NCB <= 2 ? 1.5 : NCB <= 4 ? 1.15 : 1.0);
    if (cost < best_cost) { best_cost = cost; best = pl; best.ok = true; }
  }
  return best;
}

template <int NCB, bool M1, int NT>
int launch_p(const PParams& p, int lds, hipStream_t s) {
  auto kern = k_conv_p2d<NCB, M1, NT>;
  if (int e = raise_dyn_lds(kern, kLdsMax, "conv_p2d")) return e;
  hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), lds, s, p);
  return check_launch("conv_p2d");
}
template <bool M1, int NT>
int launch_p_ncb(int NCB, const PParams& p, int lds, hipStream_t s) {
  switch (NCB) {
    case 8: return launch_p<8, M1, NT>(p, lds, s);
    case 6: return launch_p<6, M1, NT>(p, lds, s);
    case 4: return launch_p<4, M1, NT>(p, lds, s);
    default: return launch_p<2, M1, NT>(p, lds, s);
  }
}
// the operand form (nc_set_p2d_terms; two fp16 terms of the tensor times a measured power of two, three products -- or three bf16 terms, six)
// Measured (tools/p2d_check.py, 216 planes): the stride-1 layer gains from the two-term form (forward 0.67 -> 0.51 ms, data gradient 0.85 -> 0.70);
// the stride-2 layers LOSE (0.33-0.39 -> 0.34-0.50 ms): a third of their time is the conversion already, and the two-term form adds a
// measuring pass over the input.  So mode 1 (default since round 5; Athena step 64.0 -> 61.5 ms): two-term for the stride-1 layer whenever the
// generators run two-term (nc_set_split_terms(2)); 3: three-term everywhere (rounds 3-4); 2: two-term for all layers (slower).
// A MEASURED power of two depends on which planes share the call: Athena's shared pass over the fake planes (216 planes in one call) and the two
// separate passes (108 each) differ in the last bit under mode 1, where mode 3 is bit-identical
// (tests/test_gpu_fullsize.py::test_athena_step_108_streams_tuner_and_shared_pass_agree runs that comparison in mode 3 and holds mode 1 to fp32
// rounding).  No range guard here: the layer's input is an InstanceNorm2d + LeakyReLU output (bounded by sqrt(H W)), its dY the norm backward's
// output, whose per-plane scale is the plane's rstd <= 1 / sqrt(eps) = 316 -- a spread far inside the 2^17 a two-term tensor carries.
static std::atomic<int> g_p2d_terms{getenv("NC_P2D_TERMS") ? atoi(getenv("NC_P2D_TERMS")) : 1};
int p2_terms(int kind) {
  const int mode = g_p2d_terms.load(std::memory_order_relaxed);
  if (mode == 2) return 2;
  return mode == 1 && kind == 0 && s3x_get_terms() == 2 ? 2 : 3;
}
}  // namespace
void p2d_set_terms(int m) { g_p2d_terms.store(m == 1 || m == 2 ? m : 3, std::memory_order_relaxed); }
int p2d_get_terms() { return g_p2d_terms.load(std::memory_order_relaxed); }
namespace {

size_t p2_packed_bytes(int NS, int Kout, int NT = 3) { return (size_t)(Kout / 64) * 2 * NS * 2 * NT * 1024; }
size_t p2_align(size_t b) { return (b + 255) & ~(size_t)255; }

// One problem of the kernel: a valid KT x KT stride-1 convolution over planes of PP = Hp x Wp units, positions (b, y < Ho, x < Wo)
struct P2Geom {
  int kind;        // 0: 4 x 4 stride 1 (forward / data gradient); 1: stride-2 forward over the S2D image; 2: stride-2 data gradient (four classes)
  int B, Cin, Kout;          // planes, channels read, channels written (of the GEMM)
  int Hin, Win, pad;         // the tensor that is read, and its zero border (kind 0 / 2)
  int Hp, Wp, PP;            // plane of units the taps walk over
  int Ho, Wo;                // valid outputs per plane (kind 2: of the largest class)
  int NB, NS, tapext, nsub;
  long TOT;
  size_t xs_bytes;
};
P2Geom p2_geom(const ConvDims& d, int dgrad) {
  P2Geom g{};
  g.B = d.N;
  if (d.sh == 1) {
    g.kind = 0;
    if (!dgrad) { g.Cin = d.C; g.Kout = d.K; g.Hin = d.H; g.Win = d.W; g.pad = 1; }
    else { g.Cin = d.K; g.Kout = d.C; g.Hin = d.Ho; g.Win = d.Wo; g.pad = 2; }  // dx = conv(dy padded by k - 1 - p = 2, flipped transposed w)
    g.Hp = g.Hin + 2 * g.pad; g.Wp = g.Win + 2 * g.pad;
    g.Ho = g.Hp - 3; g.Wo = g.Wp - 3;
    g.NB = g.Cin / 8; g.NS = 4 * g.NB; g.tapext = 3 * g.Wp + 4; g.nsub = 1;
  } else if (!dgrad) {
    g.kind = 1;
    g.Cin = d.C; g.Kout = d.K; g.Hin = d.H; g.Win = d.W; g.pad = 1;
    g.Hp = (d.H + 3) / 2; g.Wp = (d.W + 3) / 2;  // S2D of the input padded by 1 and extended to even extents
    g.Ho = d.Ho; g.Wo = d.Wo;
    g.NB = g.Cin / 8 * 2; g.NS = 2 * g.NB; g.tapext = g.Wp + 2; g.nsub = 2;
  } else {
    g.kind = 2;
    g.Cin = d.K; g.Kout = d.C; g.Hin = d.Ho; g.Win = d.Wo; g.pad = 1;
    g.Hp = g.Hin + 2; g.Wp = g.Win + 2;
    g.Ho = (d.H + 1) / 2; g.Wo = (d.W + 1) / 2;  // class (0, 0): the most positions
    g.NB = g.Cin / 16; g.NS = 2 * g.NB; g.tapext = g.Wp + 2; g.nsub = 2;
  }
  g.PP = g.Hp * g.Wp;
  g.TOT = (long)g.B * g.PP + 64;  // (slack: a brick's last piece may reach past the batch by less than a piece)
  g.xs_bytes = p2_align((size_t)g.NB * 3 * g.nsub * g.TOT * 16);
  return g;
}

bool p2_shape(const ConvDims& d, int dgrad) {
  static const int on = getenv("NC_P2D") ? atoi(getenv("NC_P2D")) : 7;  // A/B switch: bit 0 = the stride-1 layer, bit 1 = the stride-2 layers; 0: the image-staged fp32 kernels of conv2d_img.hip
  if (d.D != 1 || d.kd != 1 || d.kh != 4 || d.kw != 4 || d.sh != d.sw || (d.sh != 1 && d.sh != 2) || d.ph != 1 || d.pw != 1) return false;
  if (!(on & (d.sh == 1 ? 1 : 2))) return false;
  const P2Geom g = p2_geom(d, dgrad);
  if (g.Kout % 64 || g.Cin % (g.kind == 0 ? 64 : 16) || g.NS < 8 || (g.NS & 1)) return false;
  static const long minpos = 8192;
  if ((long)g.B * g.Ho * g.Wo < (g.kind == 2 ? minpos / 2 : minpos)) return false;  // small batches (Apollo's 1-4 planes per discriminator) stay where they are
  if (3 * g.nsub * g.TOT * 16 >= (1l << 31) || (long)d.N * d.K * d.Ho * d.Wo * 4 >= (1l << 31) || (long)d.N * d.C * d.H * d.W * 4 >= (1l << 31)) return false;
  return p2_plan((long)g.B * g.Ho * g.Wo, g.Ho, g.Wo, g.PP, g.Wp, g.Kout / 64, g.tapext, g.nsub).ok;
}

int run_p2d(const float* in, const float* w, const float* bias, float* out, const ConvDims& d, int dgrad, void* ws, size_t wsb, hipStream_t s) {
  const P2Geom g = p2_geom(d, dgrad);
  const size_t wb = p2_align(p2_packed_bytes(g.NS, g.Kout));
  if (!ws || wsb < g.xs_bytes + wb + 512) { set_error("conv_p2d: workspace too small"); return NC_ERR_WS; }
  uint4* xs = (uint4*)ws;
  unsigned short* wp = (unsigned short*)((char*)ws + g.xs_bytes);
  unsigned* cells = (unsigned*)((char*)ws + g.xs_bytes + wb);  // two-term form: [0] the input's cell, [1] the weights' (h2.hip)
  const int NT = p2_terms(g.kind);
  const long npad = (long)g.B * g.PP;
  if (NT == 2) {
    if (int e = h2_zero_cells(cells, 2, s)) return e;
    if (int e = h2_absmax(in, (long)g.B * g.Cin * g.Hin * g.Win, cells, s)) return e;
    if (int e = h2_absmax(w, (long)d.K * d.C * 16, cells + 1, s)) return e;
    if (g.kind == 1)
      hipLaunchKernelGGL(k_s2d_split3<2>, dim3((unsigned)cdiv(npad, 256), (unsigned)(g.Cin / 8 * 4)), dim3(256), 0, s, in, xs, g.Cin, g.Hin, g.Win, g.Hp, g.Wp, g.TOT, npad,
                         (const unsigned*)cells);
    else
      hipLaunchKernelGGL(k_pad_split3_2d<2>, dim3((unsigned)cdiv(npad, 256), (unsigned)(g.Cin / 8)), dim3(256), 0, s, in, xs, g.Cin, g.Hin, g.Win, g.pad, g.Hp, g.Wp,
                         g.TOT, npad, g.kind == 2 ? 1 : 0, (const unsigned*)cells);
  } else if (g.kind == 1)
    hipLaunchKernelGGL(k_s2d_split3<3>, dim3((unsigned)cdiv(npad, 256), (unsigned)(g.Cin / 8 * 4)), dim3(256), 0, s, in, xs, g.Cin, g.Hin, g.Win, g.Hp, g.Wp, g.TOT, npad,
                       (const unsigned*)nullptr);
  else
    hipLaunchKernelGGL(k_pad_split3_2d<3>, dim3((unsigned)cdiv(npad, 256), (unsigned)(g.Cin / 8)), dim3(256), 0, s, in, xs, g.Cin, g.Hin, g.Win, g.pad, g.Hp, g.Wp,
                       g.TOT, npad, g.kind == 2 ? 1 : 0, (const unsigned*)nullptr);
  if (int e = check_launch("conv_p2d convert")) return e;
  const long total = (long)(p2_packed_bytes(g.NS, g.Kout, NT) / 2);
  // forward: w[co][ci][tap]; data gradients: w[co as ci][ci as co][..] (ConvDims: weights are [d.K][d.C][4][4])
  const long so = dgrad ? 16 : (long)d.C * 16, si = dgrad ? (long)d.C * 16 : 16;
  static const int flush = 4;
  const int nclass = g.kind == 2 ? 4 : 1;
  for (int cls = 0; cls < nclass; ++cls) {
    const int ry = cls >> 1, rx = cls & 1;
    PParams p{};
    p.xs = xs; p.wp = (const uint4*)wp; p.bias = bias; p.y = out; p.amax_x = cells; p.amax_w = cells + 1;
    p.B = g.B; p.NB = g.NB; p.K = g.Kout; p.NS = g.NS;
    p.Wp = g.Wp; p.PP = g.PP; p.TOT = g.TOT; p.KT = g.Kout / 64;
    if (g.kind == 2) {  // dx[b][c][2u + ry][2v + rx], u < ceil((H - ry) / 2)
      p.Ho = (d.H - ry + 1) / 2; p.Wo = (d.W - rx + 1) / 2;
      p.qbase = ry * g.Wp + rx;
      p.sB = (long)d.C * d.H * d.W; p.sC = (long)d.H * d.W; p.sY = 2 * d.W; p.sX = 2; p.obase = (long)ry * d.W + rx;
      p.out_elems = (long)d.N * d.C * d.H * d.W;
    } else {
      p.Ho = g.Ho; p.Wo = g.Wo;
      p.sB = (long)g.Kout * g.Ho * g.Wo; p.sC = (long)g.Ho * g.Wo; p.sY = g.Wo; p.sX = 1;
      p.out_elems = (long)g.B * g.Kout * g.Ho * g.Wo;
    }
    if (p.Ho < 1 || p.Wo < 1) continue;
    p.HoWo = p.Ho * p.Wo;
    p.npos = (long)g.B * p.HoWo;
    const P2Plan pl = p2_plan(p.npos, p.Ho, p.Wo, g.PP, g.Wp, p.KT, g.tapext, g.nsub, NT);
    if (!pl.ok) { set_error("conv_p2d: shape not covered"); return NC_ERR_SHAPE; }
    p.NPT = pl.NPT; p.UB = pl.UB; p.UBq = pl.UBq; p.npb = pl.npb;
    p.mUB = magic(pl.UB); p.mUBq = magic(pl.UBq); p.mHoWo = magic(p.HoWo); p.mWo = magic(p.Wo);
    p.t_count = (int)pl.ntiles; p.tiles_per_xcd = (int)cdiv(pl.ntiles, 8);
    p.flush = flush >= 4 ? flush : flush > 0 ? 4 : 1 << 30;
    if (NT == 2)
      hipLaunchKernelGGL(k_pack_w_p2d<2>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, g.NS, so, si, g.kind, g.kind == 0 && dgrad ? 1 : 0, ry, rx, total,
                         (const unsigned*)(cells + 1));
    else
      hipLaunchKernelGGL(k_pack_w_p2d<3>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, g.NS, so, si, g.kind, g.kind == 0 && dgrad ? 1 : 0, ry, rx, total,
                         (const unsigned*)nullptr);
    if (int e = check_launch("conv_p2d pack")) return e;
    const int e = NT == 2 ? (g.kind == 0 ? launch_p_ncb<false, 2>(pl.NCB, p, pl.lds, s) : launch_p_ncb<true, 2>(pl.NCB, p, pl.lds, s))
                          : (g.kind == 0 ? launch_p_ncb<false, 3>(pl.NCB, p, pl.lds, s) : launch_p_ncb<true, 3>(pl.NCB, p, pl.lds, s));
    if (e) return e;
  }
  return NC_OK;
}

}  // namespace

bool p2d_fwd_supported(const ConvDims& d) { return p2_shape(d, 0); }
bool p2d_dgrad_supported(const ConvDims& d) { return p2_shape(d, 1); }
size_t p2d_ws_bytes(const ConvDims& d) {
  size_t b = 0;
  for (int dg = 0; dg < 2; ++dg) {
    if (!p2_shape(d, dg)) continue;
    const P2Geom g = p2_geom(d, dg);
    const size_t n = g.xs_bytes + p2_align(p2_packed_bytes(g.NS, g.Kout)) + 512;
    if (n > b) b = n;
  }
  return b;
}
int conv_fwd_p2d(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  return run_p2d(x, w, bias, y, d, 0, ws, wsb, s);
}
int conv_dgrad_p2d(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  return run_p2d(dy, w, nullptr, dx, d, 1, ws, wsb, s);
}

}  // namespace nc
