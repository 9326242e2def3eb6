// 16-bit (bf16 / fp16) Conv3d 3^3 / 5^3, stride 1, "same" padding, forward and data gradient, C8 in -> C8 (or fp32 NCDHW) out:
// the tap-stream design of conv_s3x.hip carried over to ONE operand term (BASELINE.json configs[3]; models/networks.py:420-425,
// 460-469, 900-902 -- the nn.Conv3d layers of unet_deconv / deep_linear_gen with 16-bit matrix arithmetic, fp32 accumulation).
// It replaces k_conv_h (conv_h.hip) wherever the shape fits; k_conv_h staged the weights through LDS next to the bricks and
// spent 96 of the CU's 128 B/clk of LDS reads on fragment traffic (0.75 reads of 1 KiB per 32x32x16 MFMA).
//
// What bounds a one-term kernel is operand traffic per MFMA, not issue time: a fragment of 1 KiB feeds as many MFMAs as the wave's
// register tile has blocks on the OTHER axis, and with one term (no 6 products per fragment pair) only the tile shape is left to
// pay for it.  Wave tile = 64 output channels x 128 positions (4 row blocks x 8 column blocks of v_mfma_f32_16x16x32, 128
// accumulator registers):
//   * B fragments (activations) come from the LDS brick ring: 8 KiB per k-step per wave for 32 MFMAs = 16 B/clk per SIMD, half
//     of the LDS port;
//   * A fragments (weights) never touch LDS: 4 x 1 KiB per k-step per wave straight from global memory into registers through a
//     buffer descriptor with scalar offsets, requested TWO k-steps ahead (three register sets in rotation).  All waves of a
//     workgroup ask for the same 4 KiB, so the vector L1 serves them: 32 B/clk per CU of its 64;
//   * the K-dim of a tile is one stream of taps (8-channel chunk, dz, dy, dx), four per k-step (lane group g takes tap 4s + g, 8
//     channels each); the KS^2 taps of one input plane of one chunk are a brick = the flat range [q0, q0 + 512 + (KS-1)(P+1)) of
//     the zero-padded plane (pitch P = W + KS - 1), 13-18 KiB, kept in a ring of 4 (3^3) / 3 (5^3) filled by LDS-DMA through a
//     buffer descriptor (padding and planes outside the volume are out-of-range offsets: the hardware writes zeros);
//   * 256 threads per workgroup, TWO workgroups per CU (512 in flight): the two waves of a SIMD belong to different workgroups
//     with their own tiles, rings and barriers, so one computes while the other stores a tile or waits at a barrier -- the
//     conv_s3x kernel needed its deferred-store machinery (a second accumulator set) for that, which 128 + 128 registers
//     rule out here;
//   * every vector-memory wait is placed by hand: vmcnt retires in order, so the wait for the A fragments of step s must say how
//     many YOUNGER operations may stay in flight -- the 8 fragment loads of steps s + 1 / s + 2, the DMA pieces the step before
//     issued, the 32 stores of a tile that just ended.  A brick therefore has two k-steps (and its ring distance) to arrive.
// Weights are rounded to the 16-bit type at packing (round to nearest, or tap-diffused for deep_linear_gen: conv_h.hip).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace nc {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const volatile __attribute__((address_space(3))) unsigned long long* lds64_t;
typedef const __attribute__((address_space(3))) f32x4* ldsf4_t;

constexpr int kWavesC = 4;
constexpr int kThreadsC = kWavesC * 64;
constexpr int kLdsWG = 80 * 1024;  // two workgroups per CU
constexpr int kNCB = 8, kPT = kWavesC * kNCB * 16;  // 512 positions per tile
#ifndef NC_C8X_ABL
#define NC_C8X_ABL 0
#endif
// timing experiments only (tools/variant.sh): 1 no barrier, 2 no brick DMA, 4 no B-fragment reads, 8 no A-fragment loads, 16 no stores,
// 32 no MFMAs, 64 no hand-placed vmcnt waits
constexpr int kAbl = NC_C8X_ABL;
#ifndef NC_C8X_BD
#define NC_C8X_BD 2
#endif

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m) { return __umulhi(n, m); }
unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

template <int DT>
__device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  if constexpr (DT == NC_DT_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ unsigned short cvt16(float f) {
  if constexpr (DT == NC_DT_F16) {
    const _Float16 v = (_Float16)f;
    return __builtin_bit_cast(unsigned short, v);
  } else {
    const __bf16 v = (__bf16)f;
    return __builtin_bit_cast(unsigned short, v);
  }
}
template <int DT>
__device__ __forceinline__ float round16f(float v) {
  if constexpr (DT == NC_DT_F16) return (float)(_Float16)v;
  else return (float)(__bf16)v;
}

// Packed weights: [cot = co/64][k-step s][row block rb][lane][8] 16-bit.  Lane l = (g = l/16, m = l%16) holds output channel
// cot*64 + rb*16 + m at tap T = 4s + g of the tile's tap stream: brick T / KS^2 = chunk*KS + dz, in-plane tap T % KS^2; element j
// = input channel chunk*8 + (g odd ? (j + 4) % 8 : j) (the B fragment of an odd lane group is read upper half first).
// fwd:   w[co][ci][tap]                     (so = C*T3, si = T3, flip = 0)
// dgrad: w[co as "ci"][ci as "co"][T3-1-tap]  (so = T3, si = C*T3, flip = 1)
template <int DT>
__global__ void __launch_bounds__(256) k_pack_w_c8x(const float* __restrict__ w, unsigned short* __restrict__ wp, int KS, int NS, long so,
                                                    long si, int flip, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int T2 = KS * KS, T3 = T2 * KS;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int lane = (int)(q & 63); q >>= 6;
  const int rb = (int)(q & 3); q >>= 2;
  const int s = (int)(q % NS);
  const int cot = (int)(q / NS);
  const int g = lane >> 4, m = lane & 15;
  const int T = 4 * s + g;
  const int bi = T / T2, tp = T % T2;
  const int chunk = bi / KS, dz = bi % KS;
  const int jj = (g & 1) ? ((j + 4) & 7) : j;
  const long co = cot * 64 + rb * 16 + m, ci = chunk * 8 + jj;
  const int tap = dz * T2 + tp;
  wp[i] = cvt16<DT>(w[co * so + ci * si + (flip ? T3 - 1 - tap : tap)]);
}

// the same places, tap-diffused (conv_h.hip, "Weight rounding of a bias-free, norm-free stack"): one thread per (co, ci) pair walks
// its taps once in master order, carrying the rounding residual, and scatters the rounded values
template <int DT>
__global__ void __launch_bounds__(256) k_pack_w_c8x_diff(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout,
                                                         int KS, int NS, long so, long si, int flip) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cin * Cout) return;
  const int ci = idx % Cin, co = idx / Cin;
  const int T2 = KS * KS, T3 = T2 * KS;
  const int cot = co >> 6, rb = (co >> 4) & 3, m = co & 15;
  const int chunk = ci >> 3, jj = ci & 7;
  const float* wpair = w + co * so + ci * si;
  float res = 0.f;
  for (int t = 0; t < T3; ++t) {
    const float v = wpair[t] + res;
    const float q = round16f<DT>(v);
    res = v - q;
    const int tap = flip ? T3 - 1 - t : t;  // packed tap
    const int dz = tap / T2, tp = tap - dz * T2;
    const int T = (chunk * KS + dz) * T2 + tp;
    const int s = T >> 2, g = T & 3;
    const int j = (g & 1) ? ((jj + 4) & 7) : jj;
    wp[(((((long)cot * NS + s) * 4 + rb) * 64) + g * 16 + m) * 8 + j] = cvt16<DT>(q);
  }
}

struct CParams {
  const uint4* xh;    // C8 input [N][C/8][D][H][W] units
  const uint4* wp;    // packed weights
  const float* bias;  // nullable
  float* y;           // fp32 NCDHW output, or
  uint2* yh;          // (non-null) 16-bit C8 output [N][ctot/8][S][8], this call's K channels starting at channel c0
  int ctot, c0;
  int N, NCH, D, H, W, K;
  int P;              // row pitch of the padded plane, W + KS - 1
  int HP;             // H * P: flattened output positions of a plane (pad columns included)
  int TPP, KT;        // tiles per plane, K / 64
  int npb;            // 1 KiB pieces per brick
  int NS;             // k-steps per tile
  unsigned mP;
  int t_count, tiles_per_xcd;
  long long* dbg;  // NC_C8X_STAMP builds: s_memtime sums of workgroup 0 / wave 0 (timing experiments only)
};

struct CTile {
  int n, cot, z, q0;
};

// Tile order: output-channel tile fastest, then kTileGroup in-plane neighbours of one plane, then z, then the next group of in-plane tiles,
// then the sample.  The workgroups of an XCD (64 consecutive tiles at any moment) therefore work on 4 neighbouring tiles x 16 consecutive
// planes: a tile's halo rows ((KS - 1)(P + 1) of its 512 + ... units: +59 % at 148^2) are its in-plane neighbour's own rows, and a plane is
// the dz = 0 / 1 / 2 brick of three z-neighbours -- both re-reads hit that XCD's L2 while the data are there.  (z-major order of whole
// planes, the round-1 order of k_conv_h, leaves the in-plane neighbour 148 tiles = 2.3 rounds away: at 4 x 148^3 the input no longer
// fits the 256 MB infinity cache and the halo comes from HBM again -- 1.49 x the algorithmic bytes.)
constexpr int kTileGroup = 4;
__device__ __forceinline__ CTile c_decode(const CParams& p, int t) {
  CTile o;
  o.cot = t % p.KT; t /= p.KT;
  const int per_n = p.TPP * p.D;
  o.n = t / per_n;
  const int u = t - o.n * per_n;
  const int full = (p.TPP / kTileGroup) * kTileGroup * p.D;  // tiles in whole groups
  int tp;
  if (u < full) {
    const int grp = u / (kTileGroup * p.D), rem = u - grp * (kTileGroup * p.D);
    o.z = rem / kTileGroup;
    tp = grp * kTileGroup + (rem - o.z * kTileGroup);
  } else {
    const int L = p.TPP % kTileGroup, v = u - full;  // the last, narrower group
    o.z = v / L;
    tp = (p.TPP / kTileGroup) * kTileGroup + (v - o.z * L);
  }
  o.q0 = tp * kPT;
  o.cot = __builtin_amdgcn_readfirstlane(o.cot); o.z = __builtin_amdgcn_readfirstlane(o.z);
  o.n = __builtin_amdgcn_readfirstlane(o.n); o.q0 = __builtin_amdgcn_readfirstlane(o.q0);
  return o;
}

// s_waitcnt takes its count as an immediate; the count the kernel needs is a (wave-uniform) run-time value.  A computed jump into a table of
// 64 {s_waitcnt vmcnt(k); s_branch end} pairs of 8 bytes each (a compiler-built switch over the dozen values that occur came out as a
// 40-instruction chain of flag tests per k-step).  n is clamped to [0, 63]; a smaller n than the true number of younger operations only
// waits longer.
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  n = n < 0 ? 0 : n > 63 ? 63 : n;
  const int off = __builtin_amdgcn_readfirstlane(n * 8 + 12);  // 12 = the three instructions between the pc read and the table
  asm volatile(
    "s_getpc_b64 s[98:99]\n\t"
    "s_add_u32 s98, s98, %0\n\t"
    "s_addc_u32 s99, s99, 0\n\t"
    "s_setpc_b64 s[98:99]\n\t"
    "s_waitcnt vmcnt(0)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(1)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(2)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(3)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(4)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(5)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(6)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(7)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(8)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(9)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(10)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(11)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(12)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(13)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(14)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(15)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(16)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(17)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(18)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(19)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(20)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(21)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(22)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(23)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(24)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(25)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(26)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(27)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(28)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(29)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(30)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(31)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(32)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(33)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(34)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(35)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(36)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(37)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(38)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(39)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(40)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(41)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(42)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(43)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(44)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(45)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(46)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(47)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(48)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(49)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(50)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(51)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(52)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(53)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(54)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(55)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(56)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(57)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(58)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(59)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(60)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(61)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(62)\n\ts_branch 1f\n\t"
    "s_waitcnt vmcnt(63)\n\ts_branch 1f\n\t"
    "1:\n\t"
    : : "s"(off) : "memory", "s98", "s99", "scc");
}

constexpr unsigned kOut = 0x80000000u;  // a buffer offset beyond every descriptor's range: loads deliver zeros, stores are dropped
constexpr int kPMax = 6;                // DMA pieces per wave whose offsets are kept in registers (planner: npb <= 4 * kPMax)

// (free functions, not lambdas inside the kernel: a lambda called from a lambda that is itself called from the k-step lambda made hipcc
//  drop the kernel's host-side handle -- the same accident conv_s3x.hip notes)
// per-lane source offset of DMA piece `pc` of a brick of the tile at q0, relative to the plane (the same for every brick of the tile)
template <int KS>
__device__ __forceinline__ unsigned c_piece_off(const CParams& p, int q0, int pc, int lane) {
  int ln = lane;
  asm volatile("" : "+v"(ln));  // (rebuilt from the lane id: not one more register kept across the k-loop)
  const unsigned F = (unsigned)q0 + (unsigned)(pc * 64 + ln);
  const unsigned rr = fdiv(F, p.mP);
  const int xx = (int)(F - rr * p.P) - KS / 2;
  const int yy = (int)rr - KS / 2;
  const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
  return ok ? (unsigned)(yy * p.W + xx) * 16u : kOut;
}
struct CBrick {
  __amdgpu_buffer_rsrc_t rs;  // the 8-channel block's D * H * W units (empty for a plane outside the volume: zeros)
  int soff;                   // byte offset of the plane
  unsigned char* buf;         // this wave's first 1 KiB piece of the ring slot
};
template <int KS>
__device__ __forceinline__ CBrick c_brick(const CParams& p, const CTile& t, int bi, unsigned char* slot_base, int wave) {
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int chunk = bi / KS, dz = bi - chunk * KS;
  const int zz = t.z + dz - KS / 2;
  const bool zok = (unsigned)zz < (unsigned)p.D;
  const uint4* blk = p.xh + ((long)t.n * p.NCH + chunk) * S;
  CBrick b;
  b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(blk), 0, zok ? (unsigned)(S * 16) : 0u, 0x00020000);
  b.soff = zok ? (int)(zz * HW * 16) : 0;
  b.buf = slot_base + wave * 1024;
  return b;
}
// piece j of this wave (piece wave + 4 j of the brick): ONE instruction when the offset is at hand
__device__ __forceinline__ void c_dma_piece(const CBrick& b, int j, unsigned off) {
  if (kAbl & 2) return;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.rs, (lptr_t)(b.buf + j * (kWavesC * 1024)), 16, off, b.soff, 0, 0);
}

template <int DT, int KS>
__global__ void __launch_bounds__(kThreadsC, 2) k_conv_c8x(const CParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  constexpr int PAD = KS / 2, T2 = KS * KS, NCB = kNCB;
  constexpr int RING = KS == 3 ? 4 : 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int NB = p.NCH * KS;     // bricks per tile
  const int BB = p.npb * 1024;   // bytes per ring slot
  float* const btab = reinterpret_cast<float*>(lds_raw + RING * BB);  // the layer's bias (zeros without one), K floats

  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int t_lo = xcd * p.tiles_per_xcd;
  int t_hi = t_lo + p.tiles_per_xcd;
  if (t_hi > p.t_count) t_hi = p.t_count;
  int tcur = t_lo + wslot;
  if (tcur >= t_hi) return;
  CTile cur = c_decode(p, tcur), nxt = cur;
  bool more_tiles = tcur + nslot < t_hi;
  if (more_tiles) nxt = c_decode(p, tcur + nslot);

  for (int i = tid; i < p.K; i += kThreadsC) btab[i] = p.bias ? p.bias[i] : 0.f;  // (read after the first barrier at the earliest)

  // ---- brick staging (conv_s3x.hip, one term): unit u = padded flat position q0 + u -> (row, column) of the padded plane; a lane whose
  // unit is padding asks for an offset beyond the descriptor's range and the hardware delivers zeros
  const int my_pieces = __builtin_amdgcn_readfirstlane((p.npb - wave + kWavesC - 1) / kWavesC);  // DMA instructions of this wave per brick
  constexpr int PMAX = kPMax;
  unsigned po[PMAX];  // source offsets of this wave's pieces for the CURRENT tile
  auto tile_offsets = [&](const CTile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PMAX; ++j) po[j] = c_piece_off<KS>(p, t.q0, wave + kWavesC * j, lane);
  };
  // a whole brick at once, offsets computed on the spot (prologue, and the first bricks of the NEXT tile, whose offsets differ)
  auto issue_brick = [&](const CTile& t, int bi, int slot) __attribute__((always_inline)) {
    const CBrick b = c_brick<KS>(p, t, bi, lds_raw + slot * BB, wave);
#pragma unroll 1
    for (int j = 0; j < my_pieces; ++j) c_dma_piece(b, j, c_piece_off<KS>(p, t.q0, wave + kWavesC * j, lane));
  };

  // ---- weights: A fragments through a buffer descriptor, scalar offset per (tile, k-step, row block); inline assembly so that no
  // compiler-placed wait ever refers to them (wait_vm below is the only place they are waited for)
  u32x4 wrsrc;
  {
    const unsigned long long wa = (unsigned long long)p.wp;
    wrsrc.x = __builtin_amdgcn_readfirstlane((unsigned)wa);
    wrsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
    wrsrc.z = __builtin_amdgcn_readfirstlane(0x7fffffffu);
    wrsrc.w = __builtin_amdgcn_readfirstlane(0x00020000u);
  }
  const int wvoff = lane * 16;
  auto wtile = [&](int cot) __attribute__((always_inline)) { return cot * p.NS * 4096; };
  auto load_a = [&](u32x4 (&A)[4], int soff) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(soff);
    if (kAbl & 8) return;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(A[rb]) : "v"(wvoff), "s"(wrsrc), "s"(so + rb * 1024) : "memory");
  };
  auto wait_vm = [&](int n) __attribute__((always_inline)) {
    if (kAbl & 64) return;
    wait_vmcnt_dyn(n);
  };

  // ---- B fragments: unit (slot, position + tap) of the ring, read as two 8-byte halves (odd lane groups: upper first) -- every read
  // instruction touches each of the 64 banks exactly once whatever the tap offsets of the four lane groups are
  const unsigned lane_b = (unsigned)(((wave * NCB * 16 + m16) * 16) + (g & 1) * 8);
  const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)lds_raw;
  auto read_b = [&](u32x4& B, unsigned vo, int cb) __attribute__((always_inline)) {
    if (kAbl & 4) return;
    u64x2 v;
    v.x = *(lds64_t)(vo + cb * 256);  // volatile: two ds_read_b64, never one ds_read2_b64
    v.y = *(lds64_t)((vo ^ 8u) + cb * 256);
    B = __builtin_bit_cast(u32x4, v);
  };

  f32x4 acc[4][NCB];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rb][cb][e] = 0.f;
  };

  // ---- results leave through a buffer descriptor over one sample's output: a lane whose position is a pad column or lies beyond the
  // plane stores to an out-of-range offset, which the hardware drops -- the NUMBER of store instructions is fixed (wait_vm counts them)
  auto epilogue = [&](const CTile& t) __attribute__((always_inline)) {
    if (kAbl & 16) return;
    // (per-lane constants are rebuilt from the lane id here rather than kept in registers across the k-loop: every register is spoken for)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int m16 = ln & 15, g = ln >> 4;
    const int cob = t.cot * 64;
    unsigned pos[NCB];  // offset of this lane's position in column block cb inside the plane, or kOut
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const unsigned f = (unsigned)(t.q0 + wave * NCB * 16 + cb * 16 + m16);
      const unsigned yy = fdiv(f, p.mP);
      const unsigned xx = f - yy * p.P;
      pos[cb] = ((int)yy < p.H && (int)xx < p.W) ? yy * p.W + xx : kOut;
    }
    if (p.yh) {
      // C8: a lane holds channels 4g .. 4g + 3 of row block rb = one 8-byte half of the unit of 8-channel block rb*2 + g/2, lane (g ^ 1, m)
      // the other half.  v_permlane16_swap trades halves between the lane rows of TWO column blocks: afterwards an even lane row owns the
      // whole unit of column block 2j, the odd row that of 2j + 1 -- one 16-byte store per lane, 512 contiguous bytes per row pair
      uint2* yb = p.yh + ((long)t.n * (p.ctot >> 3) + ((p.c0 + cob) >> 3)) * S * 2;
      const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (unsigned)((long)(p.ctot - p.c0 - cob) / 8 * S * 16), 0x00020000);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const f32x4 bv = *(ldsf4_t)(lds_base + (unsigned)(RING * BB) + (unsigned)(cob + rb * 16 + 4 * g) * 4u);
        const unsigned blk = (unsigned)(((rb * 2 + (g >> 1)) * S + (long)t.z * HW) * 16);
#pragma unroll
        for (int cb = 0; cb < NCB; cb += 2) {
          unsigned lo[2], hi[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            lo[k] = cvt16<DT>(acc[rb][cb + k][0] + bv[0]) | ((unsigned)cvt16<DT>(acc[rb][cb + k][1] + bv[1]) << 16);
            hi[k] = cvt16<DT>(acc[rb][cb + k][2] + bv[2]) | ((unsigned)cvt16<DT>(acc[rb][cb + k][3] + bv[3]) << 16);
          }
          const auto x = __builtin_amdgcn_permlane16_swap(lo[0], lo[1], false, false);  // odd rows of [0] <-> even rows of [1]
          const auto y = __builtin_amdgcn_permlane16_swap(hi[0], hi[1], false, false);
          u32x4 o; o.x = x[0]; o.y = y[0]; o.z = x[1]; o.w = y[1];
          const unsigned ps = (g & 1) ? pos[cb + 1] : pos[cb];
          __builtin_amdgcn_raw_buffer_store_b128(o, ys, ps != kOut ? blk + ps * 16u : kOut, 0, 0);
        }
      }
    } else {
      const __amdgpu_buffer_rsrc_t ys =
          __builtin_amdgcn_make_buffer_rsrc(p.y + (long)t.n * p.K * S, 0, (unsigned)((long)p.K * S * 4), 0x00020000);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const f32x4 bv = *(ldsf4_t)(lds_base + (unsigned)(RING * BB) + (unsigned)(cob + rb * 16 + 4 * g) * 4u);
        const unsigned ch = (unsigned)(((long)(cob + rb * 16 + 4 * g) * S + (long)t.z * HW) * 4);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[rb][cb][e] + bv[e];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ys, pos[cb] != kOut ? ch + (unsigned)(e * S * 4) + pos[cb] * 4u : kOut,
                                                  0, 0);
          }
      }
    }
  };
  const int nstores = __builtin_amdgcn_readfirstlane(p.yh ? 2 * NCB : 0);  // counted stores of a tile (the 128 of the fp32 form are not)

  // ---- prologue: the first RING - 2 bricks and the A fragments of the first two k-steps
  int ring = 0;  // ring slot of brick 0 of the current tile
#pragma unroll
  for (int b = 0; b < RING - 2; ++b) issue_brick(cur, b, b);
  tile_offsets(cur);
  // Fragment sets in rotation.  3^3: requested AD = 2 k-steps ahead, three sets -- step t of a tile multiplies set t % 3 and requests step
  // t + 2 into set (t + 2) % 3; the 27 Cin / 32 k-steps of a tile are a multiple of three, so every tile starts on set 0.  5^3: 250 Cin / 64
  // k-steps are not; its bricks last 6.25 k-steps (a brick arrives in one step of six), so it runs with AD = 1 and two sets in turn (an even
  // number of steps) rather than exchange registers whose contents are still in flight at a tile's end.
  constexpr int AD = KS == 3 ? 2 : 1;
  u32x4 A0[4], A1[4], A2[4];
  load_a(A0, wtile(cur.cot));
  if (AD == 2) load_a(A1, wtile(cur.cot) + 4096);
  zero_acc();

  int tpl = g, sl = ring;  // per-lane tap state: lane group g is at in-plane tap tpl of the brick in slot sl
  auto b_off = [&]() __attribute__((always_inline)) {
    const int dy = KS == 3 ? (tpl * 11) >> 5 : (tpl * 13) >> 6;
    const int dx = tpl - dy * KS;
    return lds_base + lane_b + (unsigned)(sl * BB + (dy * p.P + dx) * 16);
  };
  unsigned vo = b_off();
  // Order of a wave's vector-memory operations in a k-step: [the four A loads of step s + 2] (behind column blocks 0..3), then, when a
  // brick arrived, [the DMA pieces of brick na + RING - 2] (column blocks 4..7); the stores of a tile follow its last step.  y1 / y2 = what
  // the previous step / the one before issued BEHIND its A loads.  The fragments of step s were requested in step s - 2, ahead of that
  // step's pieces: at the top of step s everything younger than them -- y2, the 4 loads of step s - 1, y1 -- may stay in flight, so a
  // brick piece has two to three k-steps to land and a tile's stores two.
  int y1 = 0, y2 = 0;
  constexpr int BD = NC_C8X_BD;     // B fragments are requested BD column blocks ahead
  u32x4 B[2 * BD];

#ifdef NC_C8X_STAMP
  long long tprev = 0, tsum[5] = {0, 0, 0, 0, 0};
  int nst = 0;
  // interval i = time from the previous stamp to stamp i (0: from stamp 3 of the previous step, i.e. loop overhead + epilogue)
#define STAMP(i) do { if (p.dbg && blockIdx.x == 0 && tid == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); if (tprev) tsum[i] += t_ - tprev; tprev = t_; if (i == 0) ++nst; } } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
  while (true) {
    const int wt = wtile(cur.cot);
    const int wt_next = more_tiles ? wtile(nxt.cot) : wt;
    int na = 0;  // next brick of this tile to arrive

    // One k-step.  Ac: this step's fragments; An: receives those of step s + 2.
    auto kstep = [&](int s, u32x4 (&Ac)[4], u32x4 (&An)[4]) __attribute__((always_inline)) {
      const bool last = s + 1 == p.NS;
      const int s2 = s + AD;
      const int aoff = __builtin_amdgcn_readfirstlane(s2 < p.NS ? wt + s2 * 4096 : wt_next + (s2 - p.NS) * 4096);  // (beyond the last tile: a dummy request)
      STAMP(0);
      wait_vm(AD == 2 ? 4 + y1 + y2 : y1);  // (AD = 1: this step's fragments were the previous step's requests; only what it issued behind them is younger)
      asm volatile("" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Ac[2]), "+v"(Ac[3])::"memory");
      y2 = y1; y1 = 0;
      STAMP(1);
      // brick `na` is first used by k-step s + 1 (brick 0: by step 0): complete in LDS for THIS wave's pieces (requested RING - 2 bricks
      // = 4 or more k-steps ago); the barrier makes that true for everybody's, and says everybody is done with brick na - 2 (its last
      // fragment reads fed MFMAs of an earlier k-step: complete), whose slot brick na + RING - 2 is requested into
      int npend = 0;
      CBrick bk{};
      if (na < NB && 4 * s + 7 >= T2 * na) {
        if (!(kAbl & 1)) asm volatile("s_barrier" ::: "memory");
        const int b = na + RING - 2;
        if (b < NB) {
          bk = c_brick<KS>(p, cur, b, lds_raw + ((ring + b) % RING) * BB, wave);
          npend = my_pieces;
          y1 = my_pieces;
        } else if (more_tiles) {
          // (two or one bricks per tile: offsets of another tile, computed on the spot.  These pieces go out AHEAD of this step's A
          //  loads, i.e. they are older than them and must not be counted in y1: the wait two steps on would let that many operations
          //  too many stay in flight and hand the MFMAs fragments that have not arrived)
          issue_brick(nxt, b - NB, (ring + b) % RING);
        }
        ++na;
      }
      STAMP(2);
      if (s == 0) {  // the first fragments of a tile cannot be read ahead: its brick 0 arrived at the barrier just above
#pragma unroll
        for (int cb = 0; cb < BD; ++cb) read_b(B[cb], vo, cb);
      }
      // next k-step's tap state
      tpl += 4;
      if (tpl >= T2) { tpl -= T2; sl = sl == RING - 1 ? 0 : sl + 1; }
      const unsigned nvo = b_off();
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        if (cb < 4) {
          if (!(kAbl & 8))
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(An[cb]) : "v"(wvoff), "s"(wrsrc), "s"(aoff + cb * 1024) : "memory");
        } else {
          if (cb - 4 < npend) c_dma_piece(bk, cb - 4, po[cb - 4]);
          if (cb < PMAX && cb < npend) c_dma_piece(bk, cb, po[cb]);
        }
        if (cb + BD < NCB) read_b(B[(cb + BD) % (2 * BD)], vo, cb + BD);
        else if (!last) read_b(B[(cb + BD) % (2 * BD)], nvo, cb + BD - NCB);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          if (!(kAbl & 32)) acc[rb][cb] = mfma<DT>(Ac[rb], B[cb % (2 * BD)], acc[rb][cb]);
        // the two reads spread over this block's four MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      vo = nvo;
      STAMP(3);
    };
    if constexpr (AD == 2) {
#pragma unroll 1
      for (int s = 0; s < p.NS; s += 3) {
        kstep(s, A0, A2);
        kstep(s + 1, A1, A0);
        kstep(s + 2, A2, A1);
      }
    } else {
#pragma unroll 1
      for (int s = 0; s < p.NS; s += 2) {
        kstep(s, A0, A1);
        kstep(s + 1, A1, A0);
      }
    }
    epilogue(cur);
    y1 += nstores;
#ifdef NC_C8X_STAMP
    if (p.dbg && blockIdx.x == 0 && tid == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); tsum[4] += t_ - tprev; tprev = t_; }
#endif
    if (!more_tiles) break;
    zero_acc();
    ring = (ring + NB) % RING;
    cur = nxt;
    tcur += nslot;
    more_tiles = tcur + nslot < t_hi;
    if (more_tiles) nxt = c_decode(p, tcur + nslot);
    tile_offsets(cur);
    // (the tap state carries over by itself: NS * 4 taps = NB whole bricks, so tpl is back at g and sl at the new tile's brick 0)
  }
#ifdef NC_C8X_STAMP
  if (p.dbg && blockIdx.x == 0 && tid == 0) {
    for (int i = 0; i < 5; ++i) p.dbg[i] = tsum[i];
    p.dbg[5] = nst;
  }
#endif
}

struct CPlan {
  int P, HP, TPP, npb, lds;
  long ntiles;
  double eff;  // useful share of the launch: plane quantisation x round quantisation
  bool ok;
};

CPlan c_plan(int N, int D, int H, int W, int Kout, int KS) {
  CPlan pl{};
  pl.P = W + KS - 1;
  const long HP = (long)H * pl.P;
  pl.HP = (int)HP;
  const int U = kPT + (KS - 1) * (pl.P + 1);
  pl.npb = (U + 63) / 64;
  const int ring = KS == 3 ? 4 : 3;
  pl.lds = ring * pl.npb * 1024 + Kout * 4;
  pl.TPP = (int)((HP + kPT - 1) / kPT);
  pl.ntiles = (long)N * D * pl.TPP * (Kout / 64);
  const long rounds = (pl.ntiles + 511) / 512;
  pl.eff = (double)H * W / ((double)pl.TPP * kPT) * (double)pl.ntiles / (double)(rounds * 512);
  pl.ok = pl.lds <= kLdsWG && pl.npb <= 6 * kWavesC && pl.ntiles < (1l << 30);  // (6 = PMAX of the kernel)
  return pl;
}

std::atomic<int> g_c8x_mode{-1};
int c8x_mode() {  // 0 = off (k_conv_h everywhere), 1 = where the launch fills >= 60 % of its rounds of 512 workgroups (default), 2 = wherever the shape fits
  int m = g_c8x_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    m = getenv("NC_C8X") ? atoi(getenv("NC_C8X")) : 1;
    g_c8x_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}

}  // namespace

void c8x_set_mode(int m) { g_c8x_mode.store(m < 0 ? 0 : m > 2 ? 2 : m, std::memory_order_relaxed); }
int c8x_get_mode() { return c8x_mode(); }

size_t c8x_packed_bytes(int Cin, int Kout, int KS) {
  const int NS = KS * KS * KS * (Cin / 8) / 4;
  return (size_t)(Kout / 64) * NS * 4096;
}

// Cin / Kout: channels of the tensor read / written by THIS call (the data gradient swaps the layer's)
bool c8x_supported(int N, int Cin, int D, int H, int W, int Kout, int KS, bool fp32_out) {
  if (!c8x_mode() || (KS != 3 && KS != 5)) return false;
  if (Cin % (KS == 3 ? 32 : 64) || Kout % 64) return false;  // whole k-steps ((Cin / 8) * KS^3 taps in fours): 3^3 a multiple of three, 5^3 an even number
  const long S = (long)D * H * W;
  if (S * 16 >= (1l << 31)) return false;  // byte offsets inside one 8-channel block stay below the out-of-range mark
  if (fp32_out ? (long)Kout * S * 4 >= (1l << 31) : (long)(Kout / 8) * S * 16 >= (1l << 31)) return false;
  if ((long)H * (W + KS - 1) + 4096 >= (1l << 31)) return false;
  // 5^3: k_conv_h's pair mode (26 taps x 8 channels per stage, weights shared through LDS by all eight waves, MFMA busy 0.72) is the
  // better kernel in the step -- same-box A/B of the configs[3] step, profiles/r04_ab_c8x.txt: 1,340-1,400 against 1,200-1,300 TFLOP/s --
  // and gives the same bits; the tap-stream kernel takes 5^3 only on request (mode 2: tests, timing)
  if (KS == 5 && c8x_mode() < 2) return false;
  const CPlan pl = c_plan(N, D, H, W, Kout, KS);
  if (!pl.ok) return false;
  // launches that fill only a fraction of ONE round of 512 workgroups (a few planes) run better on k_conv_h's 256-position tiles; from
  // about 60 % on the tap-stream kernel is ahead on every shape measured (tools/c8x_time.py: 1.1-1.2 x at 37^3 ... 148^3)
  return c8x_mode() >= 2 || pl.eff >= 0.6;
}

// xh: C8 input; w: fp32 master weights; exactly one of y (fp32 NCDHW) / yh (C8, channels [c0, c0 + Kout) of ctot); wp_ws: >=
// c8x_packed_bytes of scratch
int conv_c8x(const void* xh, const float* w, const float* bias, float* y, void* yh, int ctot, int c0, int N, int Cin, int D, int H, int W,
             int Kout, int KS, long so, long si, int flip, int diffuse, int dt, void* wp_ws, hipStream_t s) {
  const CPlan pl = c_plan(N, D, H, W, Kout, KS);
  if (!pl.ok || (!y) == (!yh)) { set_error("conv_c8x: shape not covered"); return NC_ERR_SHAPE; }
  if (yh && (ctot % 8 || c0 % 8 || (long)(ctot / 8) * D * H * W * 16 >= (1l << 32))) { set_error("conv_c8x: output range"); return NC_ERR_SHAPE; }
  const int NCH = Cin / 8, NS = KS * KS * KS * NCH / 4;
  const long total = (long)(c8x_packed_bytes(Cin, Kout, KS) / 2);
  if (diffuse) {
    const unsigned nb = (unsigned)cdiv((long)Cin * Kout, 256);
    if (dt == NC_DT_F16) hipLaunchKernelGGL((k_pack_w_c8x_diff<NC_DT_F16>), dim3(nb), dim3(256), 0, s, w, (unsigned short*)wp_ws, Cin, Kout, KS, NS, so, si, flip);
    else hipLaunchKernelGGL((k_pack_w_c8x_diff<NC_DT_BF16>), dim3(nb), dim3(256), 0, s, w, (unsigned short*)wp_ws, Cin, Kout, KS, NS, so, si, flip);
  } else {
    const unsigned nb = (unsigned)cdiv(total, 256);
    if (dt == NC_DT_F16) hipLaunchKernelGGL((k_pack_w_c8x<NC_DT_F16>), dim3(nb), dim3(256), 0, s, w, (unsigned short*)wp_ws, KS, NS, so, si, flip, total);
    else hipLaunchKernelGGL((k_pack_w_c8x<NC_DT_BF16>), dim3(nb), dim3(256), 0, s, w, (unsigned short*)wp_ws, KS, NS, so, si, flip, total);
  }
  if (int e = check_launch("pack_w_c8x")) return e;
  CParams p{};
  p.xh = (const uint4*)xh; p.wp = (const uint4*)wp_ws; p.bias = bias; p.y = y; p.yh = (uint2*)yh;
  p.ctot = yh ? ctot : Kout; p.c0 = yh ? c0 : 0;
  p.N = N; p.NCH = NCH; p.D = D; p.H = H; p.W = W; p.K = Kout;
  p.P = pl.P; p.HP = pl.HP; p.TPP = pl.TPP; p.KT = Kout / 64;
  p.npb = pl.npb; p.NS = NS; p.mP = magic(pl.P);
  p.t_count = (int)pl.ntiles; p.tiles_per_xcd = (int)cdiv(pl.ntiles, 8);
#ifdef NC_C8X_STAMP
  p.dbg = (long long*)((char*)wp_ws + c8x_packed_bytes(Cin, Kout, KS));  // (the timing tool leaves slack behind the packed weights)
#endif
  auto launch = [&](auto kern) -> int {
    if (int e = raise_dyn_lds(kern, kLdsWG, "conv_c8x")) return e;
    static const int grid = 512;  // (timing experiments: 256 = one workgroup per CU)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreadsC), pl.lds, s, p);
    return check_launch("conv_c8x");
  };
  if (dt == NC_DT_F16) return KS == 3 ? launch(k_conv_c8x<NC_DT_F16, 3>) : launch(k_conv_c8x<NC_DT_F16, 5>);
  return KS == 3 ? launch(k_conv_c8x<NC_DT_BF16, 3>) : launch(k_conv_c8x<NC_DT_BF16, 5>);
}

}  // namespace nc
