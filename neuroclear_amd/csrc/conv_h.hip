// 16-bit (bf16 / fp16) MFMA Conv3d, odd cubic kernel, stride 1, "same" padding: forward and dgrad.
// BASELINE.json configs[3] ("fp16 MFMA path with fp32 InstanceNorm accumulate"): the convolutions of the U-Net
// (models/networks.py:420-425,460-469) and of G_B's feature block (:900-902) multiply 16-bit operands on
// v_mfma_f32_32x32x16_{bf16,f16} (32 cycles per 32768 FLOP per SIMD, 16x the fp32 matrix rate) and accumulate in fp32;
// everything between two convolutions (InstanceNorm statistics and normalisation, ReLU, pooling, losses, Adam, the
// master weights) stays fp32.
//
// Layout.  The 16-bit operand tensor is "C8": [N][C/8][D][H][W][8 channels] -- one voxel's 8 channels are one 16-byte
// unit.  k_to_c8 converts an fp32 NCDHW tensor (coalesced along W per channel, one 16-byte store per voxel).
//   * One MFMA k-step (k = 16) is 16 input channels at ONE tap: lane (r, h) holds channels 8h..8h+7 of voxel r, i.e.
//     one unit -> ds_read_b128, conflict-free (16 consecutive lanes read 256 contiguous bytes).
//   * A tap (dy, dx) is a constant LDS offset (dy * P + dx) units because rows are flattened with pitch P = W + 2p
//     (the same trick as the fp32 kernel, conv_mfma_fwd.hip): positions on the 2p pad columns are computed and
//     discarded (1.3 % at W = 148).
//   * LDS-DMA moves one unit per lane with a per-lane source address: x-padding, out-of-volume rows and the tail of
//     the last piece come from a zero page -- no alignment cases, no zero-fill pass, any W.
// Stage = (16-channel chunk, dz): the (rows + 2p) x P units of ONE input plane for both 8-channel halves, plus the
// KS*KS x 64 x 16 packed weights of that (chunk, dz) -- weights go through LDS too, so that a wave's only vector-memory
// traffic is DMA and its only vmcnt wait sits at the stage barrier (a weight load issued behind a DMA would have to
// wait for the DMA: vmcnt retires in order).  Two stage buffers; the DMA of stage s+1 flies under the MFMAs of s.
// Tile = 64 output channels x PT flattened positions of one output plane; 8 waves x (2 x VB) accumulator tiles.
// Persistent workgroups walk XCD-contiguous tile ranges (neighbouring planes share input planes in that XCD's L2).
#include <cstdlib>

#include "common.hpp"

namespace nc {
NC_ZERO_PAGE()
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

static constexpr int kWaves = 8;
static constexpr int kThreads = kWaves * 64;
static constexpr int kLdsMaxH = 160 * 1024;

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m) { return __umulhi(n, m); }
unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

template <int DT>
__device__ __forceinline__ f32x16 mfma16(const i32x4& a, const i32x4& b, const f32x16& c) {
  if constexpr (DT == NC_DT_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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

// ---------------------------------------------------------------------------------------------------------------------
// fp32 NCDHW -> 16-bit C8.  One thread per voxel of one 8-channel block: 8 coalesced dword loads, one 16-byte store.
template <int DT>
__global__ void __launch_bounds__(256) k_to_c8(const float* __restrict__ x, uint4* __restrict__ out, long S, int C) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const long ncb = blockIdx.y;  // n * (C/8) + cb
  const float* xs = x + ncb * 8 * S + v;
  unsigned short e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = cvt16<DT>(xs[j * S]);
  uint4 o;
  o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
  o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
  out[ncb * S + v] = o;
}

// Packed weights: [cot = co/64][chunk = ci/16][dz][t = dy*KS+dx][a = (co/32)%2][h][r = co%32][8] 16-bit, element j =
// input channel chunk*16 + 8h + j.  The weights of one stage (cot, chunk, dz) are KS*KS * 2 KiB contiguous; a lane's A
// fragment is one 16-byte unit and the 32 lanes of a half read 512 contiguous bytes.
// fwd:   wp(co, ci, tap) = w[co][ci][tap]                       (so = C*T, si = T, flip = 0)
// dgrad: wp(ci as "co", co as "ci", tap) = w[co][ci][T-1-tap]    (so = T,   si = C*T, flip = 1)
// Weight rounding of a bias-free, norm-free stack (deep_linear_gen, networks.py:899-917): tap-diffused.  Rounding each
// weight to nearest leaves every (co, ci) pair's tap SUM -- the layer's response to a constant input -- with an error of
// ~sqrt(taps) half-ulps, and G_B's output for the nearly constant `fake` of the first iterations is a heavily cancelling sum
// of such responses: at configs[3]'s seed the bf16 cycle loss came out 2.2 % low (tools/glin_round_exp.py: weight rounding
// +0.0155 on a mean of 0.245, activation rounding -0.0001).  Here weight t of a pair is rounded after adding the residual
// carried from weights 0..t-1 (master order), so the running sums of the rounded weights follow the exact ones to one ulp
// -- noise shaping: the rounding error moves to the spatial frequencies where images have no energy (5 x less output bias).
// A layer in front of an InstanceNorm gains nothing (the norm removes the constant) and keeps round-to-nearest.
static thread_local int g_wdiffuse = 0;

template <int DT>
__device__ __forceinline__ float round16f(float v) {
  if constexpr (DT == NC_DT_F16) return (float)(_Float16)v;
  else return (float)(__bf16)v;
}
// value of weight `tp` of the pair whose taps start at w[0] (stride 1): round-to-nearest, or tap-diffused
template <int DT>
__device__ __forceinline__ float wvalue(const float* __restrict__ w, int tp, int diffuse) {
  if (!diffuse) return w[tp];
  float r = 0.f, q = 0.f;
  for (int t = 0; t <= tp; ++t) {
    const float v = w[t] + r;
    q = round16f<DT>(v);
    r = v - q;
  }
  return q;
}

template <int DT>
__global__ void __launch_bounds__(256) k_pack_w_h(const float* __restrict__ w, unsigned short* __restrict__ wp, int NCH,
                                                  int T3, long so, long si, int flip, int diffuse, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int r = (int)(q & 31); q >>= 5;
  const int h = (int)(q & 1); q >>= 1;
  const int a = (int)(q & 1); q >>= 1;
  const int tap = (int)(q % T3); q /= T3;  // dz * KS*KS + t
  const int chunk = (int)(q % NCH);
  const int cot = (int)(q / NCH);
  const long co = cot * 64 + a * 32 + r, ci = chunk * 16 + 8 * h + j;
  const int tp = flip ? T3 - 1 - tap : tap;
  wp[i] = cvt16<DT>(wvalue<DT>(w + co * so + ci * si, tp, diffuse));
}

// 5^3 (PAIR) packing: [cot][chunk = ci/8][dz][i = tap pair][a][h][r][8], element j = input channel chunk*8 + j at tap
// (dy, dx) = 2i + h of plane dz; zero for the 26th tap.
template <int DT>
__global__ void __launch_bounds__(256) k_pack_w_h8(const float* __restrict__ w, unsigned short* __restrict__ wp, int NCH,
                                                   int KS, long so, long si, int flip, int diffuse, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int T2 = KS * KS, NP = (T2 + 1) / 2, T3 = T2 * KS;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int r = (int)(q & 31); q >>= 5;
  const int h = (int)(q & 1); q >>= 1;
  const int a = (int)(q & 1); q >>= 1;
  const int pr = (int)(q % NP); q /= NP;
  const int dz = (int)(q % KS); q /= KS;
  const int chunk = (int)(q % NCH);
  const int cot = (int)(q / NCH);
  const int t2 = 2 * pr + h;
  float v = 0.f;
  if (t2 < T2) {
    const long co = cot * 64 + a * 32 + r, ci = chunk * 8 + j;
    const int tap = dz * T2 + t2;
    v = wvalue<DT>(w + co * so + ci * si, flip ? T3 - 1 - tap : tap, diffuse);
  }
  wp[i] = cvt16<DT>(v);
}

// Tap-diffused packing (wvalue above costs O(taps) per element when every element walks the residual chain on its own: 0.18 ms
// for a 5^3 layer): one thread per (co, ci) pair walks its taps ONCE in master order and scatters the rounded values to their
// packed places -- the inverse of the index maps of k_pack_w_h / k_pack_w_h8.
template <int DT>
__global__ void __launch_bounds__(256) k_pack_w_h_diff(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout,
                                                       int KS, long so, long si, int flip, int pair8) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cin * Cout) return;
  const int ci = idx % Cin, co = idx / Cin;
  const int T2 = KS * KS, T3 = T2 * KS, NP = (T2 + 1) / 2;
  const int cot = co >> 6, a = (co >> 5) & 1, r = co & 31;
  const float* wpair = w + co * so + ci * si;
  float res = 0.f;
  for (int t = 0; t < T3; ++t) {
    const float v = wpair[t] + res;
    const float q = round16f<DT>(v);
    res = v - q;
    const int tp = flip ? T3 - 1 - t : t;  // packed tap
    long i;
    if (pair8) {
      const int NCH = Cin / 8, chunk = ci >> 3, j = ci & 7;
      const int dz = tp / T2, t2 = tp - dz * T2, pr = t2 >> 1, h = t2 & 1;
      i = ((((((long)(cot * NCH + chunk) * KS + dz) * NP + pr) * 2 + a) * 2 + h) * 32 + r) * 8 + j;
    } else {
      const int NCH = Cin / 16, chunk = ci >> 4, h = (ci >> 3) & 1, j = ci & 7;
      i = (((((long)(cot * NCH + chunk) * T3 + tp) * 2 + a) * 2 + h) * 32 + r) * 8 + j;
    }
    wp[i] = cvt16<DT>(q);
  }
  if (pair8 && (T2 & 1)) {  // the unused second half of the last tap pair of every plane
    const int NCH = Cin / 8, chunk = ci >> 3, j = ci & 7;
    for (int dz = 0; dz < KS; ++dz)
      wp[((((((long)(cot * NCH + chunk) * KS + dz) * NP + NP - 1) * 2 + a) * 2 + 1) * 32 + r) * 8 + j] = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct HParams {
  const uint4* xh;    // C8 input
  const uint4* wp;    // packed weights
  const float* bias;  // nullable
  float* y;           // fp32 NCDHW output, or
  uint2* yh;          // (non-null) 16-bit C8 output [N][ctot/8][S][8], this call's K channels starting at channel c0
  int ctot, c0;
  int nblk_out;      // C8 epilogue: 8-channel blocks of the 64-channel tile that exist in the output (8 = all)
  const uint4* zeros; // >= 16 B of zeros in global memory
  int N, NCH, D, H, W, K;  // NCH = C / 16
  int P, R, RP;       // row pitch (units), brick rows, R * P
  int PT, TPP;        // positions per tile, tiles per plane
  int KT;             // K / 64
  unsigned mP, mRP;
  int npb;            // brick pieces (1 KiB) per stage
  int npw;            // weight pieces per stage = KS*KS*2
  int SB;             // bytes per stage buffer
  long ntiles;
  int tiles_per_xcd;  // ceil(ntiles / 8)
  int ablate;         // timing experiments only (env NC_H_ABLATE): 1 no stores, 2 no MFMA loop, 4 no DMA, 8 no LDS reads, 16 no barrier, 32 every tile stages the same plane (all L2 hits)
};

struct HTile {
  int n, cot, z, q0, yf, xoff;
};

// Tile order (round 4, as conv_c8x.hip): output-channel tile fastest, then kHGroup in-plane neighbours of one plane, then z, then the next
// group of in-plane tiles, then the sample -- the 32 workgroups of an XCD work on 4 neighbouring tiles x 8 consecutive planes at a time, so
// a tile's halo rows (its in-plane neighbour's own rows: 612 of 1024 + 612 units per brick at 5^3, 148^2) and the planes it shares with
// its z-neighbours are re-read from that XCD's L2 while they are there.  (Whole planes in z-major order left the in-plane neighbour
// 148 tiles away; at 4 x 148^3 the input no longer fits the 256 MB infinity cache and the 5^3 forward moved 2.2 x its algorithmic bytes.)
constexpr int kHGroup = 4;
__device__ __forceinline__ HTile h_decode(const HParams& p, long t) {
  HTile o;
  o.cot = (int)(t % p.KT); t /= p.KT;
  const int per_n = p.TPP * p.D;
  o.n = (int)(t / per_n);
  const int u = (int)(t - (long)o.n * per_n);
  const int full = (p.TPP / kHGroup) * kHGroup * p.D;
  int tp;
  if (u < full) {
    const int grp = u / (kHGroup * p.D), rem = u - grp * (kHGroup * p.D);
    o.z = rem / kHGroup;
    tp = grp * kHGroup + (rem - o.z * kHGroup);
  } else {
    const int L = p.TPP % kHGroup, v = u - full;
    o.z = v / L;
    tp = (p.TPP / kHGroup) * kHGroup + (v - o.z * L);
  }
  o.q0 = tp * p.PT;
  o.yf = (int)fdiv((unsigned)o.q0, p.mP);
  o.xoff = o.q0 - o.yf * p.P;
  return o;
}

// Kernel box KZ x KY x KX ("same" padding KZ/2, KY/2, KX/2).  NA = 32-channel halves of the 64-channel tile that are
// computed (1: only output channels 0..31 -- layers with <= 32 real output channels).
template <int DT, int KZ, int KY, int KX, bool PAIR, int VB, int NA>
__global__ void __launch_bounds__(kThreads, 1) k_conv_h(const HParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  const unsigned lds0 = nc_lds_addr(lds_raw);
  constexpr int PADZ = KZ / 2, PADY = KY / 2, PADX = KX / 2, T2 = KY * KX;
  // !PAIR (3^3): a k-step = 16 channels (two C8 blocks) at one tap.  PAIR (5^3, and the 8-pseudo-channel layers): a k-step
  // = 8 channels (one C8 block) at TWO taps, lane half h taking tap 2i + h (an odd tap count ends in a zero-weight tap):
  // half the brick and half the weights per stage, which is what lets two stage buffers of a 5^3 layer fit in 160 KB
  // (13 k-steps for 25 taps: 4 % padding).
  constexpr int NB = PAIR ? 1 : 2;                    // C8 blocks per chunk
  constexpr int KSTEPS = PAIR ? (T2 + 1) / 2 : T2;    // k-steps per stage
  constexpr int MAXJ = 6;  // brick pieces per wave (planner: npb <= 8 * MAXJ)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;

  // tiles of this workgroup: XCD x (= blockIdx % 8) owns tiles [x * tpx, (x+1) * tpx); its workgroups interleave
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const long t_lo = (long)xcd * p.tiles_per_xcd;
  long t_hi = t_lo + p.tiles_per_xcd;
  if (t_hi > p.ntiles) t_hi = p.ntiles;
  long tcur = t_lo + slot;
  if (tcur >= t_hi) return;

  int off[MAXJ];  // per-lane source offset (units) of brick piece wave + 8j relative to the plane, or -1 = zero page
  auto decode_pieces = [&](const HTile& t) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const unsigned u = (unsigned)((wave + kWaves * j) * 64 + lane);
      const unsigned hh = (!PAIR && u >= (unsigned)p.RP) ? 1u : 0u;
      const unsigned ur = u - hh * p.RP;
      const unsigned rr = fdiv(ur, p.mP);
      const int xx = (int)(ur - rr * p.P) - PADX;
      const int y = t.yf - PADY + (int)rr;
      const bool ok = ur < (unsigned)p.RP && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      off[j] = ok ? (int)(hh * S + (long)y * p.W + xx) : -1;
    }
  };
  // valid dz range of a tile (planes outside the volume contribute nothing: their stages are skipped)
  auto dz_lo = [&](const HTile& t) { return PADZ - t.z > 0 ? PADZ - t.z : 0; };
  auto dz_hi = [&](const HTile& t) { return t.z + PADZ > p.D - 1 ? KZ - 1 - (t.z + PADZ - (p.D - 1)) : KZ - 1; };

  auto issue = [&](int tn, int tz, int tcot, int chunk, int dz, unsigned char* buf) {
    const uint4* plane = p.xh + (((long)tn * p.NCH + chunk) * NB * p.D + (tz + dz - PADZ)) * HW;
    if (p.ablate & 4) return;
    if (p.ablate & 32) plane = p.xh + ((long)chunk * NB * p.D + 1) * HW;  // every tile stages the same plane: all L2 hits
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int pc = wave + kWaves * j;
      if (pc < p.npb) {
        const uint4* src = off[j] >= 0 ? plane + off[j] : p.zeros;
        nc_dma_lds16(src, lds0 + (unsigned)(buf - lds_raw) + pc * 1024);
      }
    }
    const uint4* ws = p.wp + (((long)tcot * p.NCH + chunk) * KZ + dz) * (KSTEPS * 128) + lane;
    unsigned char* wb = buf + p.npb * 1024;
#pragma unroll 1
    for (int pw = wave; pw < p.npw; pw += kWaves)
      nc_dma_lds16(ws + pw * 64, lds0 + (unsigned)(wb - lds_raw) + pw * 1024);
  };

  unsigned char* const buf0 = lds_raw;
  unsigned char* const buf1 = lds_raw + p.SB;

  HTile cur = h_decode(p, tcur);
  decode_pieces(cur);
  int lo = dz_lo(cur), nv = dz_hi(cur) - lo + 1;
  issue(cur.n, cur.z, cur.cot, 0, lo, buf0);
  int g = 0;  // parity of the buffer being computed

  const int qb = wave * VB * 32 + r;  // this lane's first position in the tile
  while (true) {
    const long tnext = tcur + nslot;
    const bool more_tiles = tnext < t_hi;
    HTile nxt = cur;
    if (more_tiles) nxt = h_decode(p, tnext);
    const int nstages = p.NCH * nv;

    f32x16 acc[NA][VB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][v][e] = 0.f;

    int chunk = 0, dzi = 0;
#pragma unroll 1
    for (int i = 0; i < nstages; ++i) {
      unsigned char* bc = (g & 1) ? buf1 : buf0;
      unsigned char* bn = (g & 1) ? buf0 : buf1;
      if (!(p.ablate & 16)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage i has landed
        __syncthreads();                                  // ... and everybody's; everybody is done reading bn
      }
      // next stage of this tile, or the first stage of the next tile
      int nchunk = chunk, ndz = dzi + 1;
      if (ndz == nv) { ndz = 0; ++nchunk; }
      const bool within = i + 1 < nstages;
      if (!within && more_tiles) decode_pieces(nxt);
      if (within || more_tiles)
        issue(within ? cur.n : nxt.n, within ? cur.z : nxt.z, within ? cur.cot : nxt.cot, within ? nchunk : 0,
              within ? lo + ndz : dz_lo(nxt), bn);

      if (p.ablate & 2) {
      } else if constexpr (PAIR) {
        const i32x4* wrow = reinterpret_cast<const i32x4*>(bc + p.npb * 1024) + h * 32 + r;
        const i32x4* brow = reinterpret_cast<const i32x4*>(bc) + qb + cur.xoff;
        auto boff = [&](int i) {  // this lane's tap of k-step i, as a unit offset
          const int t0 = 2 * i, t1 = 2 * i + 1 < T2 ? 2 * i + 1 : T2 - 1;
          const int o0 = (t0 / KX) * p.P + t0 % KX, o1 = (t1 / KX) * p.P + t1 % KX;
          return h ? o1 : o0;
        };
        // (fully unrolled with the two fragment sets named statically: the rolled loop carried them through register copies and an
        //  lgkmcnt(0) behind the first MFMA of every k-step -- the next k-step's reads were waited for at once)
        i32x4 A[2][2], B[2][VB];
        A[0][0] = wrow[0]; A[0][1] = wrow[64];
        {
          const i32x4* bp = brow + boff(0);
#pragma unroll
          for (int v = 0; v < VB; ++v) B[0][v] = bp[v * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < KSTEPS; ++i) {
          const int c = i & 1, n = c ^ 1;
          if (i + 1 < KSTEPS) {
            const i32x4* wn = wrow + (i + 1) * 128;
            const i32x4* bnp = brow + boff(i + 1);
            A[n][0] = wn[0]; A[n][1] = wn[64];
#pragma unroll
            for (int v = 0; v < VB; ++v) B[n][v] = bnp[v * 32];
          }
          __builtin_amdgcn_sched_barrier(0);  // the reads of k-step i + 1 stay in front of the MFMAs of k-step i (the group-barrier
#pragma unroll                                //  form of this hint was not honoured: every read sank to just in front of its wait)
          for (int v = 0; v < VB; ++v) {
            acc[0][v] = mfma16<DT>(A[c][0], B[c][v], acc[0][v]);
            if constexpr (NA == 2) acc[1][v] = mfma16<DT>(A[c][1], B[c][v], acc[1][v]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      // A fragments: unit ((t*2 + a)*2 + h)*32 + r of the stage's weights; B fragments: unit h*RP + position + tap
      const i32x4* wrow = reinterpret_cast<const i32x4*>(bc + p.npb * 1024) + h * 32 + r;
      const i32x4* brow = reinterpret_cast<const i32x4*>(bc) + h * p.RP + qb + cur.xoff;
      // Software pipeline over the taps of the plane, pinned with sched_group_barrier: the 2 + VB LDS reads of tap
      // t+1 are issued in front of the 2*VB MFMAs of tap t.  Fully unrolled, the two fragment sets named statically (the rolled
      // row loop carried them through 12 register copies per tap and waited lgkmcnt(0) behind the first MFMA of every tap).
      i32x4 A[2][2], B[2][VB];
      A[0][0] = wrow[0]; A[0][1] = wrow[64];
#pragma unroll
      for (int v = 0; v < VB; ++v) B[0][v] = brow[v * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < T2; ++t) {
        const int c = t & 1, n = c ^ 1;
        if (t + 1 < T2) {
          const i32x4* wn = wrow + (t + 1) * 128;
          const i32x4* bnp = brow + ((t + 1) / KX) * p.P + (t + 1) % KX;
          A[n][0] = wn[0]; A[n][1] = wn[64];
#pragma unroll
          for (int v = 0; v < VB; ++v) B[n][v] = bnp[v * 32];
        }
        __builtin_amdgcn_sched_barrier(0);  // the reads of tap t + 1 stay in front of the MFMAs of tap t
#pragma unroll
        for (int v = 0; v < VB; ++v) {
          acc[0][v] = mfma16<DT>(A[c][0], B[c][v], acc[0][v]);
          if constexpr (NA == 2) acc[1][v] = mfma16<DT>(A[c][1], B[c][v], acc[1][v]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      chunk = nchunk; dzi = ndz;
      ++g;
    }

    // ---- epilogue, C8 form: a lane holds channels 4h..4h+3 of four 8-channel blocks per 32-channel half, i.e. one
    //      8-byte half of a 16-byte unit; lanes (r, 0) and (r, 1) complete the unit of position r, consecutive r are
    //      consecutive units: a store instruction covers 512 contiguous bytes
    if (p.yh) {
      const int cob = cur.cot * 64;
      uint2* yb = p.yh + (((long)cur.n * (p.ctot >> 3) + ((p.c0 + cob) >> 3)) * S + (long)cur.z * HW) * 2 + h;
      long yo[VB];
#pragma unroll
      for (int v = 0; v < VB; ++v) {
        const unsigned f = (unsigned)(cur.q0 + qb + v * 32);
        const unsigned yy = fdiv(f, p.mP);
        const unsigned xx = f - yy * p.P;
        yo[v] = ((int)yy < p.H && (int)xx < p.W) ? (long)yy * p.W + xx : -1;
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        float bv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = p.bias ? p.bias[cob + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] : 0.f;
#pragma unroll
        for (int v = 0; v < VB; ++v)
          if (yo[v] >= 0) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              if (a * 4 + g4 >= p.nblk_out) continue;
              uint2 o;
              o.x = cvt16<DT>(acc[a][v][4 * g4] + bv[4 * g4]) | ((unsigned)cvt16<DT>(acc[a][v][4 * g4 + 1] + bv[4 * g4 + 1]) << 16);
              o.y = cvt16<DT>(acc[a][v][4 * g4 + 2] + bv[4 * g4 + 2]) | ((unsigned)cvt16<DT>(acc[a][v][4 * g4 + 3] + bv[4 * g4 + 3]) << 16);
              yb[((long)(a * 4 + g4) * S + yo[v]) * 2] = o;
            }
          }
      }
    } else
    // ---- epilogue: rows = output channels, lanes = positions; each store writes 128 contiguous bytes per half
    if (!(p.ablate & 1)) {
      const int cob = cur.cot * 64;
      float* yn = p.y + ((long)cur.n * p.K + cob + 4 * h) * S + (long)cur.z * HW;
      long yo[VB];  // offset of this lane's position in block v, or -1
#pragma unroll
      for (int v = 0; v < VB; ++v) {
        const unsigned f = (unsigned)(cur.q0 + qb + v * 32);
        const unsigned yy = fdiv(f, p.mP);
        const unsigned xx = f - yy * p.P;
        yo[v] = ((int)yy < p.H && (int)xx < p.W) ? (long)yy * p.W + xx : -1;
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) {  // one 32-channel half at a time: 16 bias registers live, not 32
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

// ---------------------------------------------------------------------------------------------------------------------
struct HPlan {
  int PT, VB, P, R, RP, TPP, npb, npw, SB;
  bool ok;
};

// ky x kx: in-plane taps; pair: 8-channel chunks at two taps per k-step
HPlan h_plan_box(int H, int W, int ky, int kx, bool pair) {
  HPlan pl{};
  const int T2 = ky * kx;
  pl.P = W + kx - 1;
  pl.npw = (pair ? (T2 + 1) / 2 : T2) * 2;
  const long plane = (long)H * pl.P;
  double best = 0;
  for (int VB : {4, 2, 1}) {
    const int PT = VB * 256;
    const int rows = (pl.P - 1 + PT - 1) / pl.P + 1;
    const int R = rows + ky - 1;
    const int RP = R * pl.P;
    const int npb = ((pair ? 1 : 2) * RP + 4 + 63) / 64;
    const int SB = (npb + pl.npw) * 1024;
    if (npb > 48 || 2 * SB > kLdsMaxH) continue;
    const int TPP = (int)((plane + PT - 1) / PT);
    // cost ~ positions computed per useful position, with a small bonus for the larger tile (operand reuse)
    const double eff = (double)H * W / ((double)TPP * PT) * (VB == 4 ? 1.0 : VB == 2 ? 0.93 : 0.8);
    if (eff > best) {
      best = eff;
      pl.PT = PT; pl.VB = VB; pl.R = R; pl.RP = RP; pl.npb = npb; pl.SB = SB; pl.TPP = TPP;
      pl.ok = true;
    }
  }
  return pl;
}

HPlan h_plan(const ConvDims& d) { return h_plan_box(d.H, d.W, d.kh, d.kw, d.kd == 5); }

bool h_shape_ok(const ConvDims& d, int Cin, int Kout) {
  if (d.kd != d.kh || d.kd != d.kw || (d.kd != 3 && d.kd != 5)) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.pd || d.pw != d.pd) return false;
  if (Cin % (d.kd == 5 ? 8 : 16) || Kout % 64) return false;
  if ((long)d.D * d.H * d.W * 2 >= (1l << 31)) return false;  // per-lane source offsets are 32-bit unit counts
  return h_plan(d).ok;
}

size_t packed_bytes(int Cin, int Kout, int KS) {
  const int ksteps = KS == 5 ? (KS * KS + 1) / 2 * 2 : KS * KS;  // taps per plane incl. the zero tap of the 5^3 pairing
  return (size_t)Cin * Kout * KS * ksteps * 2;
}
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <int DT, int KZ, int KY, int KX, bool PAIR, int VB, int NA>
int launch_h(const HParams& p, int lds, hipStream_t s) {
  auto kern = k_conv_h<DT, KZ, KY, KX, PAIR, VB, NA>;
  if (int e = raise_dyn_lds(kern, kLdsMaxH, "conv_h")) return e;
  hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), lds, s, p);
  return check_launch("conv_h");
}

template <int DT, int KZ, int KY, int KX, bool PAIR, int NA>
int launch_h_vb(int VB, const HParams& p, int lds, hipStream_t s) {
  switch (VB) {
    case 4: return launch_h<DT, KZ, KY, KX, PAIR, 4, NA>(p, lds, s);
    case 2: return launch_h<DT, KZ, KY, KX, PAIR, 2, NA>(p, lds, s);
    default: return launch_h<DT, KZ, KY, KX, PAIR, 1, NA>(p, lds, s);
  }
}

// x: fp32 [N][Cin][D][H][W]; w: fp32 master weights; y: fp32 [N][Kout][D][H][W].  so/si/flip: see k_pack_w_h.
template <int DT>
int run_h(const float* x, const void* xh_pre, const float* w, const float* bias, float* y, const ConvDims& d, int Cin,
          int Kout, long so, long si, int flip, void* ws, size_t wsb, hipStream_t s, void* yh = nullptr, int ctot = 0,
          int c0 = 0, bool na1 = false) {  // na1: only output channels 0 .. 31 of the (one) 64-channel tile are computed (5^3, fp32 output)
  const int KS = d.kd, T3 = KS * KS * KS;
  const HPlan pl = h_plan(d);
  const long S = (long)d.D * d.H * d.W;
  const size_t xb = xh_pre ? 0 : align256((size_t)d.N * Cin * S * 2);
  const size_t wb = align256(packed_bytes(Cin, Kout, KS));
  if (!ws || wsb < xb + wb + 256) { set_error("conv_h: workspace too small"); return NC_ERR_WS; }
  uint4* xh = xh_pre ? (uint4*)xh_pre : (uint4*)ws;
  unsigned short* wp = (unsigned short*)((char*)ws + xb);
  const uint4* zeros = reinterpret_cast<const uint4*>(nc_zero_page());
  if (!zeros) { set_error("conv_h: no zero page"); return NC_ERR_HIP; }
  if (!xh_pre) {
    hipLaunchKernelGGL((k_to_c8<DT>), dim3((unsigned)cdiv(S, 256), (unsigned)(d.N * Cin / 8)), dim3(256), 0, s, x, xh, S, Cin);
    if (int e = check_launch("to_c8")) return e;
  }
  // the tap-stream kernel (conv_c8x.hip) wherever it covers the shape and its 512-position tiles quantise well; NC_C8X=0: never
  if (!na1 && c8x_supported(d.N, Cin, d.D, d.H, d.W, Kout, KS, yh == nullptr) && wb >= c8x_packed_bytes(Cin, Kout, KS))
    return conv_c8x(xh, w, bias, yh ? nullptr : y, yh, ctot, c0, d.N, Cin, d.D, d.H, d.W, Kout, KS, so, si, flip, g_wdiffuse, DT, wp, s);
  const long total = (long)(packed_bytes(Cin, Kout, KS) / 2);
  if (g_wdiffuse)
    hipLaunchKernelGGL((k_pack_w_h_diff<DT>), dim3((unsigned)cdiv((long)Cin * Kout, 256)), dim3(256), 0, s, w, wp, Cin, Kout, KS, so, si,
                       flip, KS == 5 ? 1 : 0);
  else if (KS == 5)
    hipLaunchKernelGGL((k_pack_w_h8<DT>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, Cin / 8, KS, so, si, flip,
                       0, total);
  else
    hipLaunchKernelGGL((k_pack_w_h<DT>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, Cin / 16, T3, so, si,
                       flip, 0, total);
  if (int e = check_launch("pack_w_h")) return e;
  HParams p{};
  p.xh = xh; p.wp = (const uint4*)wp; p.bias = bias; p.y = y; p.zeros = zeros;
  p.yh = (uint2*)yh; p.ctot = yh ? ctot : Kout; p.c0 = c0; p.nblk_out = 8;
  p.N = d.N; p.NCH = KS == 5 ? Cin / 8 : Cin / 16; p.D = d.D; p.H = d.H; p.W = d.W; p.K = Kout;
  p.P = pl.P; p.R = pl.R; p.RP = pl.RP; p.PT = pl.PT; p.TPP = pl.TPP; p.KT = Kout / 64;
  p.mP = magic(pl.P); p.mRP = magic(pl.RP);
  p.npb = pl.npb; p.npw = pl.npw; p.SB = pl.SB;
  p.ntiles = (long)d.N * d.D * pl.TPP * p.KT;
  p.tiles_per_xcd = (int)cdiv(p.ntiles, 8);
  static const int ablate = getenv("NC_H_ABLATE") ? atoi(getenv("NC_H_ABLATE")) : 0;
  p.ablate = ablate;
  const int lds = 2 * pl.SB;
  if (na1) {
    if (KS != 5 || Kout != 64 || yh) { set_error("conv_h: the 32-channel form exists for 5^3, one 64-channel tile, fp32 output"); return NC_ERR_SHAPE; }
    return launch_h_vb<DT, 5, 5, 5, true, 1>(pl.VB, p, lds, s);
  }
  if (KS == 3) return launch_h_vb<DT, 3, 3, 3, false, 2>(pl.VB, p, lds, s);
  return launch_h_vb<DT, 5, 5, 5, true, 2>(pl.VB, p, lds, s);
}

// ---- one-channel layers on the 16-bit cores: "pseudo-channel" form ------------------------------------------------------
// A KS^3 convolution with ONE input channel (networks.py:420 first U-Net layer 3^3, :899 first deep_linear layer 7^3)
// becomes a KS x KS x 1 convolution with 8 input channels when the KS taps along x are moved into the channel axis:
//   X8[v][j] = x[v + (0, 0, j - KS/2)]  (j < KS; 0 beyond the row)        y[k][v] = sum_(dz,dy) sum_j w[k][dz][dy][j] X8[v + (dz,dy)][j]
// X8 is an ordinary C8 tensor with one 8-channel block (16 B per voxel), so the layer runs on k_conv_h in PAIR mode with
// the box KS x KS x 1.  The data gradient of the same layer (64 -> 1 channel) is the mirror image: a KS x KS x 1
// convolution 64 -> 8 pseudo-channels with z/y-flipped taps, DX8[v][j] = sum_k sum_(dz,dy) w[k][dz][dy][j] dy[k][v - (dz,dy)],
// followed by the fold dx[u] = sum_j DX8[u - (0, 0, j - KS/2)][j] (k_fold_x8); it computes only the first 32-channel half
// of the tile (NA = 1: 8 real + 24 zero output channels).
template <int DT>
__global__ void __launch_bounds__(256) k_build_x8(const float* __restrict__ x, uint4* __restrict__ x8, int W, int KS, long total) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= total) return;
  const int xw = (int)(v % W);
  unsigned short e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int xx = xw + j - KS / 2;
    e[j] = (j < KS && xx >= 0 && xx < W) ? cvt16<DT>(x[v + j - KS / 2]) : (unsigned short)0;
  }
  uint4 o;
  o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
  o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
  x8[v] = o;
}

template <int DT>
__global__ void __launch_bounds__(256) k_fold_x8(const uint4* __restrict__ dx8, float* __restrict__ dx, int W, int KS, long total) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= total) return;
  const int xw = (int)(u % W);
  float a = 0.f;
  for (int j = 0; j < KS; ++j) {
    const int xx = xw - (j - KS / 2);
    if (xx < 0 || xx >= W) continue;
    const uint4 q = dx8[u - (j - KS / 2)];
    const unsigned w4[4] = {q.x, q.y, q.z, q.w};
    const unsigned short hv = (unsigned short)((w4[j >> 1] >> ((j & 1) * 16)) & 0xffff);
    a += DT == NC_DT_F16 ? (float)__builtin_bit_cast(_Float16, hv) : __builtin_bit_cast(float, (unsigned)hv << 16);
  }
  dx[u] = a;
}

// PAIR packing [cot = 0][chunk][dz][i = dy pair][a][h][r][8] of the pseudo-channel layers, from w[K][1][KS][KS][KS]:
//   fwd  (chunk = 0, 8 pseudo-channels -> K = 64):   element (co = a*32 + r, j, tap dy = 2i + h) = w[co][dz][dy][j]
//   dgrad (chunks = K/8 of the real channels -> 8):  element (co = a*32 + r = j', ci = chunk*8 + j, tap) = w[ci][KS-1-dz][KS-1-dy][j']
template <int DT>
__global__ void __launch_bounds__(256) k_pack_w_x8(const float* __restrict__ w, unsigned short* __restrict__ wp, int KS, int NCH,
                                                   int dgrad, int diffuse, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int NP = (KS + 1) / 2, T3 = KS * KS * KS;
  const int j = (int)(i & 7);
  long q = i >> 3;
  const int r = (int)(q & 31); q >>= 5;
  const int h = (int)(q & 1); q >>= 1;
  const int a = (int)(q & 1); q >>= 1;
  const int pr = (int)(q % NP); q /= NP;
  const int dz = (int)(q % KS); q /= KS;
  const int chunk = (int)q;
  (void)NCH;
  const int dy = 2 * pr + h, co = a * 32 + r;
  float v = 0.f;
  if (dy < KS) {
    if (!dgrad) {
      if (j < KS) v = wvalue<DT>(w + (long)co * T3, (dz * KS + dy) * KS + j, diffuse);
    } else if (co < KS) {
      v = wvalue<DT>(w + (long)(chunk * 8 + j) * T3, ((KS - 1 - dz) * KS + (KS - 1 - dy)) * KS + co, diffuse);
    }
  }
  wp[i] = cvt16<DT>(v);
}

size_t x8_packed_bytes(int KS, int chunks) { return (size_t)chunks * KS * ((KS + 1) / 2) * 2 * 2 * 32 * 8 * 2; }

// forward of a one-channel KS^3 layer (KS = 3 or 7): x fp32 [N][1][D][H][W] -> C8 [N][ctot/8][S][8] channels [c0, c0 + 64)
template <int DT>
int run_c1_fwd(const float* x, const float* w, const float* bias, void* yh, int ctot, int c0, int N, int D, int H, int W, int KS,
               void* ws, size_t wsb, hipStream_t s) {
  const long S = (long)D * H * W;
  const HPlan pl = h_plan_box(H, W, KS, 1, true);
  const size_t xb = align256((size_t)N * S * 16), wb = align256(x8_packed_bytes(KS, 1));
  if (!pl.ok) { set_error("conv_c1_fwd_h: shape not covered"); return NC_ERR_SHAPE; }
  if (!ws || wsb < xb + wb + 256) { set_error("conv_c1_fwd_h: workspace too small"); return NC_ERR_WS; }
  uint4* x8 = (uint4*)ws;
  unsigned short* wp = (unsigned short*)((char*)ws + xb);
  const uint4* zeros = reinterpret_cast<const uint4*>(nc_zero_page());
  if (!zeros) { set_error("conv_c1_fwd_h: no zero page"); return NC_ERR_HIP; }
  hipLaunchKernelGGL((k_build_x8<DT>), dim3((unsigned)cdiv(N * S, 256)), dim3(256), 0, s, x, x8, W, KS, N * S);
  const long total = (long)(x8_packed_bytes(KS, 1) / 2);
  hipLaunchKernelGGL((k_pack_w_x8<DT>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, KS, 1, 0, g_wdiffuse, total);
  if (int e = check_launch("c1_fwd_h prep")) return e;
  HParams p{};
  p.xh = x8; p.wp = (const uint4*)wp; p.bias = bias; p.y = nullptr; p.zeros = zeros;
  p.yh = (uint2*)yh; p.ctot = ctot; p.c0 = c0; p.nblk_out = 8;
  p.N = N; p.NCH = 1; p.D = D; p.H = H; p.W = W; p.K = 64;
  p.P = pl.P; p.R = pl.R; p.RP = pl.RP; p.PT = pl.PT; p.TPP = pl.TPP; p.KT = 1;
  p.mP = magic(pl.P); p.mRP = magic(pl.RP);
  p.npb = pl.npb; p.npw = pl.npw; p.SB = pl.SB;
  p.ntiles = (long)N * D * pl.TPP;
  p.tiles_per_xcd = (int)cdiv(p.ntiles, 8);
  const int lds = 2 * pl.SB;
  if (KS == 7) return launch_h_vb<DT, 7, 7, 1, true, 2>(pl.VB, p, lds, s);
  return launch_h_vb<DT, 3, 3, 1, true, 2>(pl.VB, p, lds, s);
}

// data gradient of the same layer: dyh C8 [N][8][S][8] (64 channels, bf16) -> dx fp32 [N][1][D][H][W]
template <int DT>
int run_c1_dgrad(const void* dyh, const float* w, float* dx, int N, int D, int H, int W, int KS, void* ws, size_t wsb, hipStream_t s) {
  const long S = (long)D * H * W;
  const HPlan pl = h_plan_box(H, W, KS, 1, true);
  const size_t xb = align256((size_t)N * S * 16), wb = align256(x8_packed_bytes(KS, 8));
  if (!pl.ok) { set_error("conv_c1_dgrad_h: shape not covered"); return NC_ERR_SHAPE; }
  if (!ws || wsb < xb + wb + 256) { set_error("conv_c1_dgrad_h: workspace too small"); return NC_ERR_WS; }
  uint4* dx8 = (uint4*)ws;
  unsigned short* wp = (unsigned short*)((char*)ws + xb);
  const uint4* zeros = reinterpret_cast<const uint4*>(nc_zero_page());
  if (!zeros) { set_error("conv_c1_dgrad_h: no zero page"); return NC_ERR_HIP; }
  const long total = (long)(x8_packed_bytes(KS, 8) / 2);
  hipLaunchKernelGGL((k_pack_w_x8<DT>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, KS, 8, 1, g_wdiffuse, total);
  if (int e = check_launch("c1_dgrad_h pack")) return e;
  HParams p{};
  p.xh = (const uint4*)dyh; p.wp = (const uint4*)wp; p.bias = nullptr; p.y = nullptr; p.zeros = zeros;
  p.yh = (uint2*)dx8; p.ctot = 8; p.c0 = 0; p.nblk_out = 1;
  p.N = N; p.NCH = 8; p.D = D; p.H = H; p.W = W; p.K = 64;
  p.P = pl.P; p.R = pl.R; p.RP = pl.RP; p.PT = pl.PT; p.TPP = pl.TPP; p.KT = 1;
  p.mP = magic(pl.P); p.mRP = magic(pl.RP);
  p.npb = pl.npb; p.npw = pl.npw; p.SB = pl.SB;
  p.ntiles = (long)N * D * pl.TPP;
  p.tiles_per_xcd = (int)cdiv(p.ntiles, 8);
  const int lds = 2 * pl.SB;
  int e = KS == 7 ? launch_h_vb<DT, 7, 7, 1, true, 1>(pl.VB, p, lds, s) : launch_h_vb<DT, 3, 3, 1, true, 1>(pl.VB, p, lds, s);
  if (e) return e;
  hipLaunchKernelGGL((k_fold_x8<DT>), dim3((unsigned)cdiv(N * S, 256)), dim3(256), 0, s, dx8, dx, W, KS, N * S);
  return check_launch("c1_dgrad_h fold");
}


// =====================================================================================================================
// Weight gradient on the 16-bit matrix cores.
//   dW[k][c][tap] = sum_{n, voxel} dY[n][k][voxel] * X[n][c][voxel + tap]
// MFMA view: M = 32 output channels k, N = 32 input channels c, K-dim = 16 VOXELS.  Both operands are read from the same
// C8 images the forward kernel uses (rows = voxels, 8 channels = one 16-byte unit) with gfx950's transposing LDS read
// ds_read_b64_tr_b16: per 16-lane group a 4-voxel x 16-channel block comes back channel-major, i.e. each lane receives 4
// consecutive voxels of ITS channel -- two reads build the 8-voxel fragment of v_mfma_f32_32x32x16.  Every lane supplies
// its own row address, so the 16 voxels of a k-step are simply 16 consecutive positions of a Ty x Tx tile in row-major
// order, whatever Tx is (37 for W = 148, 36 for 108, ...), and a tap is a row shift of the X image: no alignment cases.
//
// Workgroup = 64 k x 32 c x all 27 taps, 8 waves: wave = (32-k half, group of 7 taps) -> 7 accumulator tiles.  It walks
// a contiguous range of (sample, tile, z) steps; along z the three X planes live in a 4-slot LDS ring, so a step stages
// ONE new X plane (4 C8 blocks x (Ty+2) x (Tx+2) units) and one dY plane (8 blocks x Ty x Tx units) by LDS-DMA while the
// previous step is multiplied.  The 256 persistent workgroups are dealt evenly over the (k-tile, c-tile) pairs and over
// the steps of a pair (equal contiguous shares: no tail); each writes one partial [tap][64][32] and k_wgrad_h_reduce
// adds the partials of a pair in workgroup order (deterministic, no atomics).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* ltr_t;

struct WhParams {
  const uint4* xh;   // C8 X   [N][C/8][D][H][W]
  const uint4* dyh;  // C8 dY  [N][K/8][D][H][W]
  float* part;       // [pairs * nwp][T3][64][32]
  const uint4* zeros;
  int N, C, K, D, H, W;
  int Ty, Tx, YB, XB;
  int Xp, XU, XUp;   // X image: row pitch Tx + 2p, units per block, padded block stride (== 4 or 12 mod 16)
  int PT, PTp, NK;   // dY image: positions Ty*Tx, padded block stride, k-steps = ceil(PT / 16)
  int npx, npd;      // 1 KiB pieces per X slot / per dY buffer
  int xslot, dybuf;  // bytes
  int nct;           // C / 32
  int npairs, nwp;   // (k-tile, c-tile) pairs, workgroups per pair
  long steps;        // N * YB * XB * D  (per pair)
  unsigned mTx, mXp, mXUp, mPTp;
  int ablate;        // timing experiments only (env NC_H_ABLATE): 2 = no MFMA loop, 4 = no DMA
};

template <int DT, int KS>
__global__ void __launch_bounds__(kThreads, 1) k_wgrad_h(const WhParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  // 3^3: one workgroup owns all 27 taps (ZR = 3 kernel planes, 4-slot X ring).  5^3: the 125 taps do not fit one
  // workgroup's accumulators, so a workgroup owns the 25 taps of ONE kernel plane dz (ZR = 1, 2-slot ring) and dz
  // joins (k-tile, c-tile) in the "pair" index.
  constexpr int PAD = KS / 2, T2 = KS * KS;
  constexpr int ZR = KS == 3 ? 3 : 1, NS = ZR + 1, NDG = KS / ZR, TW = ZR * T2;
  constexpr int TG = (TW + 3) / 4;  // taps per wave group: 7,7,7,6 of 27; 7,6,6,6 of 25
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 1, tg = wave >> 1;
  const int t0 = tg * (TW / 4) + (tg < TW % 4 ? tg : TW % 4);
  const int ntap = TW / 4 + (tg < TW % 4 ? 1 : 0);
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;

  // Workgroups that share operands -- the pairs of one step range: same dY, and the kernel planes of a 5^3 layer the
  // same X -- should sit on ONE XCD (one L2).  Hardware deals blockIdx round-robin over the 8 XCDs, so the logical id
  // counts the workgroups of XCD 0 first, then XCD 1, ...: consecutive logical ids are neighbours on an XCD.
  const int G = gridDim.x, xcd = blockIdx.x & 7;
  const int wg = (G >> 3) * xcd + ((G & 7) < xcd ? (G & 7) : xcd) + (blockIdx.x >> 3);
  const int pair = wg % p.npairs, wi = wg / p.npairs;
  if (wi >= p.nwp) return;
  const int dzg = pair % NDG, kc = pair / NDG;
  const int kt = kc / p.nct, ct = kc % p.nct;
  const int zsh = dzg * ZR - PAD;  // X plane of local kernel plane l at step z: z + zsh + l
  const long s_lo = p.steps * wi / p.nwp, s_hi = p.steps * (wi + 1) / p.nwp;

  unsigned char* const xring = lds_raw;                    // NS slots; plane pz lives in slot (pz + 16) % NS
  unsigned char* const dyb = lds_raw + NS * p.xslot;       // 2 buffers
  auto slot_of = [&](int pz) { return ((pz + 16) & (NS - 1)) * p.xslot; };

  // ---- LDS-DMA of one X plane (4 C8 blocks of this c-tile, tile + halo) / one dY plane (8 blocks of this k-tile)
  auto issue_x = [&](int n, int y0, int x0, int pz, unsigned char* slot) {
    if (p.ablate & 4) return;
    const bool zok = (unsigned)pz < (unsigned)p.D;
    const uint4* base = p.xh + (((long)n * (p.C / 8) + ct * 4) * p.D + (zok ? pz : 0)) * HW;
#pragma unroll 1
    for (int pc = wave; pc < p.npx; pc += kWaves) {
      const unsigned u = (unsigned)(pc * 64 + lane);
      const unsigned cb = fdiv(u, p.mXUp);
      const unsigned ur = u - cb * p.XUp;
      const unsigned ty = fdiv(ur, p.mXp);
      const int y = y0 - PAD + (int)ty, x = x0 - PAD + (int)(ur - ty * p.Xp);
      const bool ok = zok && cb < 4u && ur < (unsigned)p.XU && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const uint4* src = ok ? base + (long)cb * S + (long)y * p.W + x : p.zeros;
      nc_dma_lds16(src, nc_lds_addr((slot + pc * 1024)));
    }
  };
  auto issue_dy = [&](int n, int y0, int x0, int z, unsigned char* buf) {
    const uint4* base = p.dyh + (((long)n * (p.K / 8) + kt * 8) * p.D + z) * HW;
    if (p.ablate & 4) return;
#pragma unroll 1
    for (int pc = wave; pc < p.npd; pc += kWaves) {
      const unsigned u = (unsigned)(pc * 64 + lane);
      const unsigned cb = fdiv(u, p.mPTp);
      const unsigned rho = u - cb * p.PTp;
      const unsigned ty = fdiv(rho, p.mTx);
      const int y = y0 + (int)ty, x = x0 + (int)(rho - ty * p.Tx);
      const bool ok = cb < 8u && rho < (unsigned)p.PT && y < p.H && x < p.W;
      const uint4* src = ok ? base + (long)cb * S + (long)y * p.W + x : p.zeros;
      nc_dma_lds16(src, nc_lds_addr((buf + pc * 1024)));
    }
  };

  // ---- transposed-read roles of this lane: group g = lane/16 -> channels 16*(g&1).. of the 32-channel tile, voxel half
  //      h = g/2; inside the group lane 4q+p supplies the address of voxel row q, channels 4p..4p+3
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = g >> 1;
  const int cbsel = 2 * (g & 1) + (pp >> 1);
  const unsigned a_lane = (unsigned)(((mt * 4 + cbsel) * p.PTp) * 16 + (pp & 1) * 8);  // + rho * 16
  const unsigned b_lane = (unsigned)((cbsel * p.XUp) * 16 + (pp & 1) * 8);             // + (ty * Xp + tx) * 16

  f32x16 acc[TG];
#pragma unroll
  for (int j = 0; j < TG; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // tap j of this wave: (dz, dy, dx) and its byte offset inside an X slot
  int tdz[TG], toff[TG];
#pragma unroll
  for (int j = 0; j < TG; ++j) {
    const int t = t0 + j < TW ? t0 + j : TW - 1;
    const int dz = t / T2, dy = (t / KS) % KS, dx = t % KS;
    tdz[j] = dz;
    toff[j] = (dy * p.Xp + dx) * 16;
  }

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
    // ---- multiply: NK k-steps x ntap taps
    const unsigned abase = (unsigned)(NS * p.xslot + (z & 1) * p.dybuf) + a_lane;
    unsigned sb[TG];  // slot base + tap offset
#pragma unroll
    for (int j = 0; j < TG; ++j) sb[j] = (unsigned)(slot_of(z + zsh + tdz[j]) + toff[j]) + b_lane;
#pragma unroll 1
    for (int s = 0; s < ((p.ablate & 2) ? 0 : p.NK); ++s) {
      unsigned rho[2], bo[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        rho[s2] = (unsigned)(16 * s + 8 * h + 4 * s2 + q);
        unsigned rc = rho[s2] < (unsigned)p.PT ? rho[s2] : (unsigned)p.PT - 1;  // padded positions: dY = 0, X any finite
        const unsigned ty = fdiv(rc, p.mTx);
        bo[s2] = (ty * p.Xp + (rc - ty * p.Tx)) * 16;
      }
      i32x4 a;
      {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds_raw + abase + rho[0] * 16));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds_raw + abase + rho[1] * 16));
        a = __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int j = 0; j < TG; ++j) {
        if (j < ntap) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds_raw + sb[j] + bo[0]));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(lds_raw + sb[j] + bo[1]));
          const i32x4 b = __builtin_bit_cast(i32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[j] = mfma16<DT>(a, b, acc[j]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++step;
    if (cont) ++z; else fresh = true;
  }

  // ---- partial: part[wg][tap][k 0..63][c 0..31]; rows of the accumulator tile are k, lanes are c
  float* pw = p.part + (long)wg * TW * 64 * 32;
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int j = 0; j < TG; ++j)
    if (j < ntap) {
      float* pt = pw + ((long)(t0 + j) * 64 + mt * 32 + 4 * hh) * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) pt[((e & 3) + 8 * (e >> 2)) * 32] = acc[j][e];
    }
}

// dw[k][c][tap] = sum over the nwp workgroups of pair (k/64, c/32, tap / TW), in workgroup order
__global__ void __launch_bounds__(256) k_wgrad_h_reduce(const float* __restrict__ part, float* __restrict__ dw, int C,
                                                        int T3, int TW, int nct, int npairs, int nwp, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (k, tap, c): c fastest -> coalesced partial reads
  if (i >= total) return;
  const int c = (int)(i % C);
  const int t = (int)((i / C) % T3);
  const int k = (int)(i / ((long)C * T3));
  const int ndg = T3 / TW;
  const int pair = ((k / 64) * nct + c / 32) * ndg + t / TW;
  const long off = ((long)(t % TW) * 64 + (k & 63)) * 32 + (c & 31);
  float sacc = 0.f;
  for (int w = 0; w < nwp; ++w) sacc += part[((long)(w * npairs + pair) * TW) * 64 * 32 + off];
  dw[((long)k * C + c) * T3 + t] = sacc;
}

struct WhPlan {
  int Ty, Tx, YB, XB, Xp, XU, XUp, PT, PTp, NK, npx, npd, xslot, dybuf;
  bool ok;
};

int pad_4mod8(int v) {  // smallest v' >= v with v' == 4 (mod 8): block stride that keeps the transposed reads conflict-free
  int r = v + ((4 - (v & 7)) & 7);
  return r;
}

WhPlan wh_plan(const ConvDims& d) {
  WhPlan best{};
  double best_cost = 1e30;
  const int KS = d.kd;
  for (int nx = 1; nx <= 8; ++nx) {
    const int Tx = (d.W + nx - 1) / nx;
    for (int Ty = 1; Ty <= 16 && Ty <= d.H; ++Ty) {
      WhPlan pl{};
      pl.Ty = Ty; pl.Tx = Tx;
      pl.XB = (d.W + Tx - 1) / Tx; pl.YB = (d.H + Ty - 1) / Ty;
      pl.Xp = Tx + KS - 1; pl.XU = (Ty + KS - 1) * pl.Xp; pl.XUp = pad_4mod8(pl.XU + KS);  // + KS: tap reads of the clamped tail
      pl.PT = Ty * Tx; pl.NK = (pl.PT + 15) / 16; pl.PTp = pad_4mod8(pl.NK * 16);
      pl.npx = (4 * pl.XUp + 63) / 64; pl.npd = (8 * pl.PTp + 63) / 64;
      pl.xslot = pl.npx * 1024; pl.dybuf = pl.npd * 1024;
      if ((KS == 3 ? 4 : 2) * pl.xslot + 2 * pl.dybuf > kLdsMaxH) continue;
      if (pl.NK < 4) continue;
      // cost per useful position: MFMA time (k-steps incl. padding and tile overhang) + a staging term
      const double useful = (double)d.H * d.W;
      const double mfma = (double)pl.YB * pl.XB * pl.NK * 16;
      const double stage = (double)pl.YB * pl.XB * (4.0 * pl.XUp + 8.0 * pl.PTp) / 12.0;  // units per 12 MFMA columns
      const double cost = (mfma + 0.15 * stage) / useful;
      if (cost < best_cost) { best_cost = cost; best = pl; best.ok = true; }
    }
  }
  return best;
}

bool wh_shape_ok(const ConvDims& d) {
  if (d.kd != d.kh || d.kd != d.kw || (d.kd != 3 && d.kd != 5)) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.pd || d.pw != d.pd) return false;
  if (d.C % 32 || d.K % 64) return false;
  if ((d.K / 64) * (d.C / 32) * (d.kd == 5 ? 5 : 1) > 256) return false;
  if ((long)d.D * d.H * d.W * 8 >= (1l << 31)) return false;
  return wh_plan(d).ok;
}

template <int DT>
int run_wh(const float* x, const void* xh_pre, const float* dy, const void* dyh_pre, float* dw, const ConvDims& d, void* ws,
           size_t wsb, hipStream_t s) {
  const int KS = d.kd, T3 = KS * KS * KS, TW = KS == 3 ? 27 : 25, NS = KS == 3 ? 4 : 2;
  const WhPlan pl = wh_plan(d);
  const long S = (long)d.D * d.H * d.W;
  const size_t xb = xh_pre ? 0 : align256((size_t)d.N * d.C * S * 2);
  const size_t yb = dyh_pre ? 0 : align256((size_t)d.N * d.K * S * 2);
  const int npairs = (d.K / 64) * (d.C / 32) * (T3 / TW);
  int nwp = 256 / npairs;
  const long steps = (long)d.N * pl.YB * pl.XB * d.D;
  if (nwp > steps) nwp = (int)steps;
  const size_t pb = align256((size_t)npairs * nwp * TW * 64 * 32 * 4);
  if (!ws || wsb < xb + yb + pb + 256) { set_error("wgrad_h: workspace too small"); return NC_ERR_WS; }
  uint4* xh = xh_pre ? (uint4*)xh_pre : (uint4*)ws;
  uint4* dyh = dyh_pre ? (uint4*)dyh_pre : (uint4*)((char*)ws + xb);
  float* part = (float*)((char*)ws + xb + yb);
  const uint4* zeros = reinterpret_cast<const uint4*>(nc_zero_page());
  if (!zeros) { set_error("wgrad_h: no zero page"); return NC_ERR_HIP; }
  if (!xh_pre)
    hipLaunchKernelGGL((k_to_c8<DT>), dim3((unsigned)cdiv(S, 256), (unsigned)(d.N * d.C / 8)), dim3(256), 0, s, x, xh, S, d.C);
  if (!dyh_pre)
    hipLaunchKernelGGL((k_to_c8<DT>), dim3((unsigned)cdiv(S, 256), (unsigned)(d.N * d.K / 8)), dim3(256), 0, s, dy, dyh, S, d.K);
  if (int e = check_launch("to_c8")) return e;
  if (c8x_wgrad_supported(d) && c8x_wgrad_part_bytes(d) <= wsb - xb - yb) return conv_wgrad_c8x(xh, dyh, dw, d, DT, part, wsb - xb - yb, s);
  WhParams p{};
  p.xh = xh; p.dyh = dyh; p.part = part; p.zeros = zeros;
  p.N = d.N; p.C = d.C; p.K = d.K; p.D = d.D; p.H = d.H; p.W = d.W;
  p.Ty = pl.Ty; p.Tx = pl.Tx; p.YB = pl.YB; p.XB = pl.XB; p.Xp = pl.Xp; p.XU = pl.XU; p.XUp = pl.XUp;
  p.PT = pl.PT; p.PTp = pl.PTp; p.NK = pl.NK; p.npx = pl.npx; p.npd = pl.npd; p.xslot = pl.xslot; p.dybuf = pl.dybuf;
  p.nct = d.C / 32; p.npairs = npairs; p.nwp = nwp; p.steps = steps;
  p.mTx = magic(pl.Tx); p.mXp = magic(pl.Xp); p.mXUp = magic(pl.XUp); p.mPTp = magic(pl.PTp);
  static const int ablate = getenv("NC_H_ABLATE") ? atoi(getenv("NC_H_ABLATE")) : 0;
  p.ablate = ablate;
  if (int e = raise_dyn_lds((k_wgrad_h<DT, 3>), kLdsMaxH, "wgrad_h")) return e;
  if (int e = raise_dyn_lds((k_wgrad_h<DT, 5>), kLdsMaxH, "wgrad_h")) return e;
  const int lds = NS * pl.xslot + 2 * pl.dybuf;
  if (KS == 3) hipLaunchKernelGGL((k_wgrad_h<DT, 3>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
  else hipLaunchKernelGGL((k_wgrad_h<DT, 5>), dim3(npairs * nwp), dim3(kThreads), lds, s, p);
  if (int e = check_launch("wgrad_h")) return e;
  const long total = (long)d.K * d.C * T3;
  hipLaunchKernelGGL(k_wgrad_h_reduce, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, part, dw, d.C, T3, TW, d.C / 32,
                     npairs, nwp, total);
  return check_launch("wgrad_h_reduce");
}

}  // namespace

void h_set_weight_diffusion(int on) {
  static const bool allowed = !(getenv("NC_W_DIFFUSE") && atoi(getenv("NC_W_DIFFUSE")) == 0);  // A/B switch
  g_wdiffuse = on && allowed;
}

bool h_fwd_supported(const ConvDims& d) { return h_shape_ok(d, d.C, d.K); }
bool h_dgrad_supported(const ConvDims& d) { return h_shape_ok(d, d.K, d.C); }
bool h_wgrad_supported(const ConvDims& d) { return wh_shape_ok(d); }
size_t h_ws_bytes(const ConvDims& d) {
  const bool fd = h_fwd_supported(d) || h_dgrad_supported(d), wg = h_wgrad_supported(d);
  if (!fd && !wg) return 0;
  const long S = (long)d.D * d.H * d.W;
  const int cm = d.C > d.K ? d.C : d.K;
  size_t b = align256((size_t)d.N * cm * S * 2) + align256(packed_bytes(d.C, d.K, d.kd)) + 512;
  if (wg) {
    const size_t w = align256((size_t)d.N * d.C * S * 2) + align256((size_t)d.N * d.K * S * 2) +
                     align256((size_t)256 * 27 * 64 * 32 * 4) + 512;  // partials: <= 256 workgroups x <= 27 taps
    if (w > b) b = w;
  }
  return b;
}

int conv_wgrad_h(const float* x, const void* xh, const float* dy, const void* dyh, float* dw, const ConvDims& d, int dt,
                 void* ws, size_t wsb, hipStream_t s) {
  if (dt == NC_DT_F16) return run_wh<NC_DT_F16>(x, xh, dy, dyh, dw, d, ws, wsb, s);
  return run_wh<NC_DT_BF16>(x, xh, dy, dyh, dw, d, ws, wsb, s);
}

int to_c8(const float* x, void* xh, int N, int C, long S, int dt, hipStream_t s) {
  if (dt == NC_DT_F16)
    hipLaunchKernelGGL((k_to_c8<NC_DT_F16>), dim3((unsigned)cdiv(S, 256), (unsigned)(N * C / 8)), dim3(256), 0, s, x, (uint4*)xh, S, C);
  else
    hipLaunchKernelGGL((k_to_c8<NC_DT_BF16>), dim3((unsigned)cdiv(S, 256), (unsigned)(N * C / 8)), dim3(256), 0, s, x, (uint4*)xh, S, C);
  return check_launch("to_c8");
}

int conv_fwd_h(const float* x, const void* xh, const float* w, const float* b, float* y, const ConvDims& d, int dt,
               void* ws, size_t wsb, hipStream_t s) {
  const long T3 = (long)d.kd * d.kh * d.kw;
  if (dt == NC_DT_F16) return run_h<NC_DT_F16>(x, xh, w, b, y, d, d.C, d.K, d.C * T3, T3, 0, ws, wsb, s);
  return run_h<NC_DT_BF16>(x, xh, w, b, y, d, d.C, d.K, d.C * T3, T3, 0, ws, wsb, s);
}

// C8 in -> C8 out (the 16-bit end-to-end path): yh / dxh receive the result rounded to `dt`, as channels
// [c0, c0 + K) of a [N][ctot/8][S][8] tensor (a half of a concat buffer, or ctot = K, c0 = 0 for a dense one)
int conv_fwd_h_c8(const void* xh, const float* w, const float* b, void* yh, int ctot, int c0, const ConvDims& d, int dt,
                  void* ws, size_t wsb, hipStream_t s) {
  const long T3 = (long)d.kd * d.kh * d.kw;
  ProfScope ps(0, 1, d, 1, s);
  if (dt == NC_DT_F16) return run_h<NC_DT_F16>(nullptr, xh, w, b, nullptr, d, d.C, d.K, d.C * T3, T3, 0, ws, wsb, s, yh, ctot, c0);
  return run_h<NC_DT_BF16>(nullptr, xh, w, b, nullptr, d, d.C, d.K, d.C * T3, T3, 0, ws, wsb, s, yh, ctot, c0);
}

// 5^3 forward from a C8 input onto the FIRST 32 output channels of a [64][Cin][125] weight tensor (rows 32 .. 63 are not read by the matrix
// instructions), fp32 output y[n][64][S] of which channels 0 .. 31 are written: half the matrix work of the 64-channel layer.  deep_linear_gen's
// forward without act1 on the 16-bit path (gen_nets_lp.hip).
int conv_fwd_h_na1(const void* xh, const float* w, float* y, const ConvDims& d, int dt, void* ws, size_t wsb, hipStream_t s) {
  if (dt != NC_DT_BF16 || d.K != 64 || d.kd != 5 || !h_fwd_supported(d)) { set_error("conv_fwd_h_na1: shape not covered"); return NC_ERR_SHAPE; }
  ConvDims dh = d;
  ProfScope ps(0, 1, dh, 1, s);
  return run_h<NC_DT_BF16>(nullptr, xh, w, nullptr, y, d, d.C, d.K, (long)d.C * 125, 125, 0, ws, wsb, s, nullptr, 0, 0, true);
}

int conv_dgrad_h_c8(const void* dyh, const float* w, void* dxh, int ctot, int c0, const ConvDims& d, int dt, void* ws,
                    size_t wsb, hipStream_t s) {
  const long T3 = (long)d.kd * d.kh * d.kw;
  ProfScope ps(1, 1, d, 1, s);
  if (dt == NC_DT_F16) return run_h<NC_DT_F16>(nullptr, dyh, w, nullptr, nullptr, d, d.K, d.C, T3, d.C * T3, 1, ws, wsb, s, dxh, ctot, c0);
  return run_h<NC_DT_BF16>(nullptr, dyh, w, nullptr, nullptr, d, d.K, d.C, T3, d.C * T3, 1, ws, wsb, s, dxh, ctot, c0);
}

bool c1_h_supported(int D, int H, int W, int KS) {
  if (KS != 3 && KS != 7) return false;
  if ((long)D * H * W * 2 >= (1l << 31)) return false;
  return h_plan_box(H, W, KS, 1, true).ok;
}
size_t c1_h_ws_bytes(int N, int D, int H, int W, int KS) {
  return align256((size_t)N * D * H * W * 16) + align256(x8_packed_bytes(KS, 8)) + 512;
}
int conv_c1_fwd_h(const float* x, const float* w, const float* bias, void* yh, int ctot, int c0, int N, int D, int H, int W, int KS,
                  int dt, void* ws, size_t wsb, hipStream_t s) {
  ConvDims d;
  make_dims(d, N, 1, D, H, W, 64, KS, KS, KS, 1, KS / 2);
  ProfScope ps(0, 1, d, 1, s);
  if (dt == NC_DT_F16) return run_c1_fwd<NC_DT_F16>(x, w, bias, yh, ctot, c0, N, D, H, W, KS, ws, wsb, s);
  return run_c1_fwd<NC_DT_BF16>(x, w, bias, yh, ctot, c0, N, D, H, W, KS, ws, wsb, s);
}
int conv_c1_dgrad_h(const void* dyh, const float* w, float* dx, int N, int D, int H, int W, int KS, void* ws, size_t wsb,
                    hipStream_t s) {
  ConvDims d;
  make_dims(d, N, 1, D, H, W, 64, KS, KS, KS, 1, KS / 2);
  ProfScope ps(1, 1, d, 1, s);
  return run_c1_dgrad<NC_DT_BF16>(dyh, w, dx, N, D, H, W, KS, ws, wsb, s);
}

int conv_dgrad_h(const float* dy, const void* dyh, const float* w, float* dx, const ConvDims& d, int dt, void* ws,
                 size_t wsb, hipStream_t s) {
  const long T3 = (long)d.kd * d.kh * d.kw;
  if (dt == NC_DT_F16) return run_h<NC_DT_F16>(dy, dyh, w, nullptr, dx, d, d.K, d.C, T3, d.C * T3, 1, ws, wsb, s);
  return run_h<NC_DT_BF16>(dy, dyh, w, nullptr, dx, d, d.K, d.C, T3, d.C * T3, 1, ws, wsb, s);
}

}  // namespace nc
