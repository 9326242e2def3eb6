// Weight gradient of Conv3d (3^3 / 5^3, stride 1, same padding) on the fp32 matrix cores of gfx950.
// Replaces autograd's backward-weights of nn.Conv3d (loss_G.backward(), apollo_model.py:283) for the layers of
// models/networks.py:420-425,442,460-469,900-902.
//
// GEMM view:  dW[co][(ci,tap)] = sum_{n,v} dY[co][v] * X[ci][v + tap]        (M = co, N = ci x taps, K = voxels)
//   * weight stationary: a workgroup owns the accumulators of ONE dz plane of the kernel for a 64 x CIW block of
//     (co, ci) -- KS^2 taps x (AB x 16 co) x 16 ci per wave in registers (v_mfma_f32_16x16x4_f32, exact fp32) -- and
//     streams its share of the volume through LDS.  dz planes / channel blocks are workgroup groups (grid.y), voxel
//     partitions are grid.x; partial slabs + a fixed-order reduce kernel (deterministic, no atomics).
//   * ROW STREAMING: the unit of work is one full image row (fixed n, z, y; all W columns).  Each step stages ONE new
//     dY row [64 co][W] and ONE new X row [CIW ci][W] (row y+PAD of plane z+dz-PAD) into a ring of KS rows, so X is
//     read once per dz group with no halo re-reads, and tile y's taps (dy, dx) are ring-slot + column offsets.
//     Staging needs no per-lane index arithmetic at all: a wave copies whole rows, lane = column, the row's base
//     address is wave-uniform (scalar unit).  Global loads of step s+1 are issued before the MFMA loop of step s.
//   * LDS images are channel-major with pitch == 2 (mod 32): the 16 channels x 2 voxels of a half-wave hit 32 distinct
//     banks for every tap shift (ds_read_b32); dY pad columns stay zero, so the 4-voxel k-steps may run past W.
#include "common.hpp"

namespace nc {
NC_ZERO_PAGE()


typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kNSEG = 3;  // 64-column segments per row: W <= 192
static constexpr int kLdsMaxW = 160 * 1024;

struct WgParams {
  const float* x;
  const float* dy;
  float* slab;
  int C, K, N, D, H, W;
  int PR, PAr, QT4;    // X row pitch per channel, dY pitch per channel (floats), W rounded up to 4
  int CB, KBK, parts;  // channel blocks (C/CIW, K/64), voxel partitions
};

// NC_WG_ABLATE (timing experiments only, tools/ablate_fwd.py; results are garbage when set): bit 0 no global loads,
// bit 1 no LDS staging writes, bit 2 no barriers, bit 3 operands not read from LDS
#ifndef NC_WG_ABLATE
#define NC_WG_ABLATE 0
#endif

template <int KS, int AB>
__global__ __launch_bounds__(512) void k_wgrad_mfma(WgParams p) {
  constexpr int PAD = KS / 2;
  constexpr int T = KS * KS;
  constexpr int CIW = 32 * AB;    // input channels per workgroup
  constexpr int NCIB = CIW / 16;  // ci blocks (waves along ci)
  constexpr int XR = CIW / 8;     // X rows staged per wave per step
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cib = wave % NCIB, cog = wave / NCIB;
  const int l15 = lane & 15, kq = lane >> 4;

  // group decode: g = (dz * CB + cb) * KBK + kbk
  const int g = blockIdx.y;
  const int kbk = g % p.KBK, cb = (g / p.KBK) % p.CB, dz = g / (p.KBK * p.CB);
  const int part = blockIdx.x;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  // output planes z whose input plane z + dz - PAD exists
  const int zlo = max(0, PAD - dz), zhi = min(p.D, p.D + PAD - dz);
  const int Dv = max(0, zhi - zlo);
  const long nrows = (long)p.N * Dv * p.H;
  const long r0 = nrows * part / p.parts, r1 = nrows * (part + 1) / p.parts;

  const int SLOT = CIW * p.PR;  // one ring slot = one X row of all CIW channels
  float* xT = lds;
  float* dyT = lds + KS * SLOT;
  for (int i = tid; i < KS * SLOT + 64 * p.PAr; i += 512) lds[i] = 0.f;

  f32x4 acc[AB][T];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging registers: XR x-rows and 8 dY-rows per wave, kNSEG segments of 64 columns each
  float sx[XR][kNSEG], sd[8][kNSEG];
  const bool c0 = lane < p.W, c1 = lane + 64 < p.W, c2 = lane + 128 < p.W;

  // a "load step" brings X row yy (plane z + dz - PAD) into the ring and, when it completes a tile, dY row yy - PAD.
  auto issue_loads = [&](int n, int z, int yy, bool with_dy) {
    const bool xrow_ok = yy >= 0 && yy < p.H;
    const float* xb = p.x + ((long)n * p.C + cb * CIW + wave) * S + (long)(z + dz - PAD) * HW + (long)yy * p.W + lane;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const float* r = xb + (long)(8 * j) * S;
      sx[j][0] = (xrow_ok && c0) ? r[0] : 0.f;
      sx[j][1] = (xrow_ok && c1) ? r[64] : 0.f;
      sx[j][2] = (xrow_ok && c2) ? r[128] : 0.f;
    }
    if (with_dy) {
      const float* db = p.dy + ((long)n * p.K + kbk * 64 + wave) * S + (long)z * HW + (long)(yy - PAD) * p.W + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* r = db + (long)(8 * j) * S;
        sd[j][0] = c0 ? r[0] : 0.f;
        sd[j][1] = c1 ? r[64] : 0.f;
        sd[j][2] = c2 ? r[128] : 0.f;
      }
    }
  };
  auto write_lds = [&](int yy, bool with_dy) {
    const int slot = (yy + 8 * KS) % KS;
    float* xs = xT + slot * SLOT + wave * p.PR + PAD + lane;
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      float* r = xs + (8 * j) * p.PR;
      if (c0) r[0] = sx[j][0];
      if (c1) r[64] = sx[j][1];
      if (c2) r[128] = sx[j][2];
    }
    if (with_dy) {
      float* ds = dyT + wave * p.PAr + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float* r = ds + (8 * j) * p.PAr;
        if (c0) r[0] = sd[j][0];
        if (c1) r[64] = sd[j][1];
        if (c2) r[128] = sd[j][2];
      }
    }
  };

  // ---- walk the rows [r0, r1): per (n, z) plane segment [ya, yb) the load steps run yy = ya-PAD .. yb-1+PAD
  long r = r0;
  int n = 0, z = 0, ya = 0, yb = 0, yy = 0;
  bool have = false;
  auto next_segment = [&]() {
    if (r >= r1) {
      have = false;
      return;
    }
    const long plane = r / p.H;
    ya = (int)(r - plane * p.H);
    const long rem = r1 - r;
    yb = (int)min((long)p.H, ya + rem);
    n = (int)(plane / Dv);
    z = zlo + (int)(plane - (long)n * Dv);
    yy = ya - PAD;
    r += yb - ya;
    have = true;
  };
  next_segment();
  if (have && !(NC_WG_ABLATE & 1)) issue_loads(n, z, yy, yy - PAD >= ya);

  const int a_off = (cog * 16 * AB + l15) * p.PAr + kq;
  const int b_off = (cib * 16 + l15) * p.PR + kq;
  while (have) {
    const int cyy = yy, cya = ya;
    const bool cdy = cyy - PAD >= cya;  // this step completes tile y = cyy - PAD
    if (!(NC_WG_ABLATE & 4)) __syncthreads();  // the previous tile's MFMAs are done: ring slot and dY buffer are free
    if (!(NC_WG_ABLATE & 2)) write_lds(cyy, cdy);
    if (!(NC_WG_ABLATE & 4)) __syncthreads();
    // advance and prefetch the next step's rows while this tile is being multiplied
    ++yy;
    if (yy > yb - 1 + PAD) next_segment();
    if (have && !(NC_WG_ABLATE & 1)) issue_loads(n, z, yy, yy - PAD >= ya);
    if (cdy) {
      const int y = cyy - PAD;
      const float* pa = dyT + a_off;
      int soff[KS];
#pragma unroll
      for (int ty = 0; ty < KS; ++ty) soff[ty] = ((y - PAD + ty + 8 * KS) % KS) * SLOT + b_off;
#pragma unroll 1
      for (int q4 = 0; q4 < p.QT4; q4 += 4) {
        float a[AB];
#pragma unroll
        for (int k = 0; k < AB; ++k) a[k] = (NC_WG_ABLATE & 8) ? (float)(q4 + k) : pa[q4 + k * 16 * p.PAr];
#pragma unroll
        for (int ty = 0; ty < KS; ++ty)
#pragma unroll
          for (int tx = 0; tx < KS; ++tx) {
            const float b = (NC_WG_ABLATE & 8) ? (float)(q4 + tx) : xT[soff[ty] + q4 + tx];
#pragma unroll
            for (int k = 0; k < AB; ++k)
              acc[k][ty * KS + tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b, acc[k][ty * KS + tx], 0, 0, 0);
          }
      }
    }
  }

  // ---- partial slab: slab[part][g][co 64][ci CIW][T];  C/D layout of 16x16 MFMA: col (ci) = lane & 15,
  //      row (co) = 4 * (lane >> 4) + r
  float* sl = p.slab + ((long)part * gridDim.y + g) * (64L * CIW * T);
#pragma unroll
  for (int k = 0; k < AB; ++k)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = cog * 16 * AB + k * 16 + 4 * kq + rr;
        const int ci = cib * 16 + l15;
        sl[((long)co * CIW + ci) * T + t] = acc[k][t][rr];
      }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (W % 4 == 0): same GEMM view, same weight-stationary tiling and row streaming as k_wgrad_mfma, but
//   * rows go global -> LDS directly (global_load_lds, 16 B per lane = 1 KiB per instruction, every lane with its own
//     source address; pad columns, rows outside the volume and piece tails come from a zero page): no staging
//     registers, no ds_write pass, ~5 DMA instructions per wave per step instead of 48 loads + 40 LDS stores;
//   * the X ring has KS + 1 slots and dY two buffers, so the rows of step s+1 land while step s is multiplied and
//     ONE barrier per step is enough;
//   * a row is processed in NXB column blocks of XW columns (a step = one block of one row) so that ring + dY
//     buffers fit 160 KB with 64 input channels per workgroup;
//   * channel pitch = 4 * odd floats: 16-B aligned pieces, and the 16 channels x 2 voxels of a half-wave fall on 16
//     distinct bank pairs (2-way conflict on ds_read_b32, 4 instead of 2 LDS cycles -- LDS is < 35 % busy);
//   * the operand reads of k-step q+4 are issued in front of the MFMAs of k-step q (sched_group_barrier).
struct WdParams {
  const float* x;
  const float* dy;
  float* slab;
  const float* zeros;  // >= 16 B of zeros in global memory
  int C, K, N, D, H, W;
  int PR, PAr;         // X / dY channel pitch in LDS (floats)
  int SX, SD;          // floats per X ring slot / dY buffer (whole 256-float pieces)
  int NXB, XW;         // column blocks per row, columns per block (multiple of 4)
  int CB, KBK, parts;
  unsigned mPR, mPAr;  // magic multipliers for / PR, / PAr
};

static constexpr int kMaxPX = 5, kMaxPD = 4;  // DMA pieces per wave per step (X slot / dY buffer)

template <int KS, int AB>
__global__ __launch_bounds__(512) void k_wgrad_dma(WdParams p) {
  constexpr int PAD = KS / 2;
  constexpr int T = KS * KS;
  constexpr int CIW = 32 * AB;
  constexpr int NCIB = CIW / 16;
  constexpr int RING = KS + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cib = wave % NCIB, cog = wave / NCIB;
  const int l15 = lane & 15, kq = lane >> 4;

  const int g = blockIdx.y;
  const int kbk = g % p.KBK, cb = (g / p.KBK) % p.CB, dz = g / (p.KBK * p.CB);
  const int part = blockIdx.x;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int zlo = max(0, PAD - dz), zhi = min(p.D, p.D + PAD - dz);
  const int Dv = max(0, zhi - zlo);
  const long nrows = (long)p.N * Dv * p.H;
  const long r0 = nrows * part / p.parts, r1 = nrows * (part + 1) / p.parts;

  float* xT = lds;
  float* dyT = lds + RING * p.SX;
  const int npx = p.SX / 256, npd = p.SD / 256;

  f32x4 acc[AB][T];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_off = (cog * 16 * AB + l15) * p.PAr + 4 * kq;
  const int b_off = (cib * 16 + l15) * p.PR + 4 * kq;

  for (int xb = 0; xb < p.NXB; ++xb) {
    const int x0 = xb * p.XW;
    const int xw = min(p.XW, p.W - x0);  // multiple of 4
    // per-lane source offsets of this wave's pieces (independent of the row): element offset from the row base of
    // channel 0, or -1 for "zero page"
    int gx[kMaxPX], gd[kMaxPD];
#pragma unroll
    for (int i = 0; i < kMaxPX; ++i) {
      const unsigned f = (unsigned)((wave + 8 * i) * 64 + lane) * 4;
      const unsigned c = __umulhi(f, p.mPR);
      const int col = (int)(f - c * p.PR);
      const int x = x0 - 4 + col;
      gx[i] = ((int)c < CIW && x >= 0 && x < p.W && col < xw + 8) ? (int)(c * S + x) : -1;
    }
#pragma unroll
    for (int i = 0; i < kMaxPD; ++i) {
      const unsigned f = (unsigned)((wave + 8 * i) * 64 + lane) * 4;
      const unsigned c = __umulhi(f, p.mPAr);
      const int col = (int)(f - c * p.PAr);
      gd[i] = (c < 64u && col < xw) ? (int)(c * S + x0 + col) : -1;
    }

    // a "load step" brings X row yy (plane z + dz - PAD) into ring slot cnt % RING and, when it completes a tile,
    // dY row yy - PAD into dY buffer dcnt & 1
    auto issue = [&](int n, int z, int yy, bool with_dy, int cnt, int dcnt) {
      const bool row_ok = yy >= 0 && yy < p.H;
      const float* xbse = p.x + ((long)n * p.C + cb * CIW) * S + (long)(z + dz - PAD) * HW + (long)yy * p.W;
      float* xs = xT + (cnt % RING) * p.SX;
#pragma unroll
      for (int i = 0; i < kMaxPX; ++i) {
        const int j = wave + 8 * i;
        if (j < npx) {
          const float* src = (row_ok && gx[i] >= 0) ? xbse + gx[i] : p.zeros;
          nc_dma_lds16(src, nc_lds_addr((xs + j * 256)));
        }
      }
      if (with_dy) {
        const float* dbse = p.dy + ((long)n * p.K + kbk * 64) * S + (long)z * HW + (long)(yy - PAD) * p.W;
        float* ds = dyT + (dcnt & 1) * p.SD;
#pragma unroll
        for (int i = 0; i < kMaxPD; ++i) {
          const int j = wave + 8 * i;
          if (j < npd) {
            const float* src = gd[i] >= 0 ? dbse + gd[i] : p.zeros;
            nc_dma_lds16(src, nc_lds_addr((ds + j * 256)));
          }
        }
      }
    };

    // ---- walk the rows [r0, r1): per (n, z) plane segment [ya, yb) the load steps run yy = ya-PAD .. yb-1+PAD
    long r = r0;
    int n = 0, z = 0, ya = 0, yb = 0, yy = 0;
    bool have = false;
    auto next_segment = [&]() {
      if (r >= r1) {
        have = false;
        return;
      }
      const long plane = r / p.H;
      ya = (int)(r - plane * p.H);
      const long rem = r1 - r;
      yb = (int)min((long)p.H, ya + rem);
      n = (int)(plane / Dv);
      z = zlo + (int)(plane - (long)n * Dv);
      yy = ya - PAD;
      r += yb - ya;
      have = true;
    };
    next_segment();
    int cnt = 0, dcnt = 0;  // load steps issued so far / dY rows issued so far
    __syncthreads();        // the previous column block's last MFMAs are done before its LDS is overwritten
    if (have && !(NC_WG_ABLATE & 1)) issue(n, z, yy, yy - PAD >= ya, cnt, dcnt);

    while (have) {
      const bool cdy = yy - PAD >= ya;  // the step in flight completes tile y = yy - PAD
      const int ccnt = cnt, cdc = dcnt;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the step have landed ...
      if (!(NC_WG_ABLATE & 4)) __syncthreads();           // ... everybody's have, and the previous tile's MFMAs are done
      // advance and start the next step's rows while this tile is being multiplied
      ++cnt;
      if (cdy) ++dcnt;
      ++yy;
      if (yy > yb - 1 + PAD) next_segment();
      if (have && !(NC_WG_ABLATE & 1)) issue(n, z, yy, yy - PAD >= ya, cnt, dcnt);
      if (cdy) {
        // 16 output voxels per k-iteration; MFMA m of the iteration reduces over voxels q16 + 4*kq + m, so a lane's
        // operands for m = 0..3 and every tap tx are consecutive floats: dY [q16 + 4kq, +4) = one aligned
        // ds_read_b128, X (one ring row) [q16 + 4kq, +12) = three, used at offset m + tx + 4 - PAD.  The next ring
        // row's reads are issued in front of this row's 4 * KS * AB MFMAs.
        const float* pa = dyT + (cdc & 1) * p.SD + a_off;
        const float* pb[KS];
#pragma unroll
        for (int ty = 0; ty < KS; ++ty) pb[ty] = xT + ((ccnt - (KS - 1) + ty + 8 * RING) % RING) * p.SX + b_off;
        f32x4 av[AB], bc[3], bn[3];
        auto ldb = [&](const float* q, f32x4 (&b)[3]) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
            b[i] = (NC_WG_ABLATE & 8) ? f32x4{1.f, 2.f, 3.f, 4.f}
                                   : *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(q + 4 * i, 16));
        };
        ldb(pb[0], bc);
#pragma unroll 1
        for (int q16 = 0; q16 < xw; q16 += 16) {
#pragma unroll
          for (int k = 0; k < AB; ++k)
            av[k] = (NC_WG_ABLATE & 8) ? f32x4{1.f, 2.f, 3.f, 4.f}
                                       : *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(pa + q16 + k * 16 * p.PAr, 16));
          __builtin_amdgcn_sched_group_barrier(0x100, AB, 0);
#pragma unroll
          for (int ty = 0; ty < KS; ++ty) {
            // next row of this iteration, or row 0 of the next iteration (the read past the last iteration stays
            // inside the slot: PR >= XW + 24)
            ldb(ty + 1 < KS ? pb[ty + 1] + q16 : pb[0] + q16 + 16, bn);
            const float bw[12] = {bc[0][0], bc[0][1], bc[0][2], bc[0][3], bc[1][0], bc[1][1],
                                  bc[1][2], bc[1][3], bc[2][0], bc[2][1], bc[2][2], bc[2][3]};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
              for (int tx = 0; tx < KS; ++tx)
#pragma unroll
                for (int k = 0; k < AB; ++k)
                  acc[k][ty * KS + tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][m], bw[m + tx + 4 - PAD],
                                                                              acc[k][ty * KS + tx], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * KS * AB, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) bc[i] = bn[i];
          }
        }
      }
    }
  }

  // ---- partial slab (same layout as k_wgrad_mfma)
  float* sl = p.slab + ((long)part * gridDim.y + g) * (64L * CIW * T);
#pragma unroll
  for (int k = 0; k < AB; ++k)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = cog * 16 * AB + k * 16 + 4 * kq + rr;
        const int ci = cib * 16 + l15;
        sl[((long)co * CIW + ci) * T + t] = acc[k][t][rr];
      }
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow volumes (W <= 56: the 54^3 and 27^3 levels of the U-Net): a step of the row-streaming kernels above would be
// a handful of MFMAs between two barriers.  k_wgrad_rows makes a step R consecutive rows of one plane, FLATTENED:
// rows sit in LDS at pitch Wg = W + (>= p) zeros, so the R rows are one run of R * Wg voxels, a tap (ty, tx) is a
// constant offset ty * Wg + tx - p, and 16-voxel k-iterations run across row boundaries (27 -> 4 rows of pitch 28 =
// 112 voxels, 96 % useful).  Each step loads its own R + 2p input rows (no ring: the halo rows come from L2 again) and
// R dY rows by LDS-DMA into a double buffer; one barrier per step.  64 co x 32 ci per workgroup, one 16 x 16 x KS^2
// accumulator set per wave.  W % 4 != 0: the group of 4 floats that holds a row's tail is written from registers by
// one thread per row (its DMA lane is masked), as in conv_mfma_fwd.hip.
struct WrParams {
  const float* x;
  const float* dy;
  float* slab;
  const float* zeros;
  int C, K, N, D, H, W;
  int R, Wg;           // rows per step, row pitch in LDS (multiple of 4, >= W + p)
  int PRc, PAc;        // channel pitch of the X window / dY rows (floats, = 8 mod 16)
  int SX, SD;          // floats per X / dY buffer (whole 256-float pieces)
  int CB, KBK, parts;
  unsigned mPRc, mPAc, mWg;
};

static constexpr int kMaxRX = 4, kMaxRD = 4;  // DMA pieces per wave per step

template <int KS>
__global__ __launch_bounds__(512) void k_wgrad_rows(WrParams p) {
  constexpr int PAD = KS / 2;
  constexpr int T = KS * KS;
  constexpr int CIW = 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cib = wave & 1, cog = wave >> 1;  // 2 ci blocks x 4 co blocks of 16
  const int l15 = lane & 15, kq = lane >> 4;
  const int g = blockIdx.y;
  const int kbk = g % p.KBK, cb = (g / p.KBK) % p.CB, dz = g / (p.KBK * p.CB);
  const int part = blockIdx.x;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int zlo = max(0, PAD - dz), zhi = min(p.D, p.D + PAD - dz);
  const int Dv = max(0, zhi - zlo);
  // work units: (n, z, row group of R rows)
  const int ngr = (p.H + p.R - 1) / p.R;
  const long nunits = (long)p.N * Dv * ngr;
  const long u0 = nunits * part / p.parts, u1 = nunits * (part + 1) / p.parts;
  const int RX = p.R + 2 * PAD;
  const int tw = p.W & 3, wfull = p.W - tw;
  const int npx = p.SX / 256, npd = p.SD / 256;
  float* xB = lds;               // two X windows
  float* dB = lds + 2 * p.SX;    // two dY buffers

  // per-lane DMA sources (step independent): element offset from (channel 0, first row of the window, x = 0) and the
  // window row; -1 = zero page, -2 = masked (row tail group)
  int gx[kMaxRX], gxr[kMaxRX], gd[kMaxRD], gdr[kMaxRD];
#pragma unroll
  for (int i = 0; i < kMaxRX; ++i) {
    const unsigned f = (unsigned)((wave + 8 * i) * 64 + lane) * 4;
    const unsigned c = __umulhi(f, p.mPRc);
    const int col = (int)(f - c * p.PRc) - 4;  // 4 zero floats in front of every channel's window
    const unsigned r = col >= 0 ? __umulhi((unsigned)col, p.mWg) : 0u;
    const int x = col - (int)r * p.Wg;
    gxr[i] = (int)r;
    if ((int)c >= CIW || col < 0 || (int)r >= RX || x >= p.W) gx[i] = -1;
    else if (x >= wfull) gx[i] = -2;
    else gx[i] = (int)(c * S + (long)r * p.W + x);
  }
#pragma unroll
  for (int i = 0; i < kMaxRD; ++i) {
    const unsigned f = (unsigned)((wave + 8 * i) * 64 + lane) * 4;
    const unsigned c = __umulhi(f, p.mPAc);
    const int col = (int)(f - c * p.PAc);
    const unsigned r = __umulhi((unsigned)col, p.mWg);
    const int x = col - (int)r * p.Wg;
    gdr[i] = (int)r;
    if (c >= 64u || (int)r >= p.R || x >= p.W) gd[i] = -1;
    else if (x >= wfull) gd[i] = -2;
    else gd[i] = (int)(c * S + (long)r * p.W + x);
  }
  // row tails (W % 4 != 0): thread t < CIW * RX -> X row (ci, r); next 64 * R threads -> dY row (co, r)
  int tl_lofs = -1, tl_gofs = 0, tl_r = 0;
  bool tl_isx = false;
  if (tw) {
    if (tid < CIW * RX) {
      const int c = tid / RX, r = tid - c * RX;
      tl_isx = true; tl_r = r;
      tl_lofs = c * p.PRc + 4 + r * p.Wg + wfull;
      tl_gofs = (int)(c * S + (long)r * p.W + wfull);
    } else if (tid - CIW * RX < 64 * p.R) {
      const int t2 = tid - CIW * RX;
      const int c = t2 / p.R, r = t2 - c * p.R;
      tl_r = r;
      tl_lofs = c * p.PAc + r * p.Wg + wfull;
      tl_gofs = (int)(c * S + (long)r * p.W + wfull);
    }
  }
  float tl_v[3] = {0.f, 0.f, 0.f};

  auto decode = [&](long u, int& n, int& z, int& y0) {
    const long pl = u / ngr;
    y0 = (int)(u - pl * ngr) * p.R;
    n = (int)(pl / Dv);
    z = zlo + (int)(pl - (long)n * Dv);
  };
  auto issue = [&](long u, int buf) {
    int n, z, y0;
    decode(u, n, z, y0);
    const float* xbse = p.x + ((long)n * p.C + cb * CIW) * S + (long)(z + dz - PAD) * HW + (long)(y0 - PAD) * p.W;
    const float* dbse = p.dy + ((long)n * p.K + kbk * 64) * S + (long)z * HW + (long)y0 * p.W;
    float* xs = xB + buf * p.SX;
    float* ds = dB + buf * p.SD;
#pragma unroll
    for (int i = 0; i < kMaxRX; ++i) {
      const int j = wave + 8 * i;
      if (j < npx) {
        const int yy = y0 - PAD + gxr[i];
        const float* src = (gx[i] >= 0 && yy >= 0 && yy < p.H) ? xbse + gx[i] : p.zeros;
        if (gx[i] != -2) nc_dma_lds16(src, nc_lds_addr((xs + j * 256)));
      }
    }
#pragma unroll
    for (int i = 0; i < kMaxRD; ++i) {
      const int j = wave + 8 * i;
      if (j < npd) {
        const float* src = (gd[i] >= 0 && y0 + gdr[i] < p.H) ? dbse + gd[i] : p.zeros;
        if (gd[i] != -2) nc_dma_lds16(src, nc_lds_addr((ds + j * 256)));
      }
    }
    if (tw && tl_lofs >= 0) {
      const int yy = tl_isx ? y0 - PAD + tl_r : y0 + tl_r;
      const bool ok = yy >= 0 && yy < p.H;
      const float* r = (tl_isx ? xbse : dbse) + tl_gofs;
#pragma unroll
      for (int e = 0; e < 3; ++e) tl_v[e] = (ok && e < tw) ? r[e] : 0.f;
    }
  };
  auto write_tail = [&](int buf) {
    if (tw && tl_lofs >= 0) {
      float* q = (tl_isx ? xB + buf * p.SX : dB + buf * p.SD) + tl_lofs;
      *reinterpret_cast<f32x4*>(__builtin_assume_aligned(q, 16)) = f32x4{tl_v[0], tl_v[1], tl_v[2], 0.f};
    }
  };

  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_off = (cog * 16 + l15) * p.PAc + 4 * kq;
  const int b_off = (cib * 16 + l15) * p.PRc + 4 * kq;  // window = 12 floats from column q - 4 + PAD ... see below
  const int nq = p.R * p.Wg;                            // voxels per step (multiple of 4; iterations round up to 16)

  if (u0 < u1) issue(u0, 0);
  for (long u = u0; u < u1; ++u) {
    const int buf = (int)((u - u0) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    write_tail(buf);
    __syncthreads();
    if (u + 1 < u1) issue(u + 1, buf ^ 1);
    // element (window row r, x) of channel c sits at c * PRc + 4 + r * Wg + x; output voxel q = r * Wg + x, tap
    // (ty, tx) reads window row r + ty, column x + tx - PAD: offset (4 - PAD) + ty * Wg + q + tx.  A lane's 12-float
    // window starts at ty * Wg + q16 + 4 * kq (16-byte aligned) and is indexed by m + tx + 4 - PAD.
    const float* pa = dB + buf * p.SD + a_off;
    const float* pb = xB + buf * p.SX + b_off;
    f32x4 bc[3], bn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) bc[i] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(pb + 4 * i, 16));
#pragma unroll 1
    for (int q16 = 0; q16 < nq; q16 += 16) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(pa + q16, 16));
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int ty = 0; ty < KS; ++ty) {
        const float* nx = ty + 1 < KS ? pb + (ty + 1) * p.Wg + q16 : pb + q16 + 16;
#pragma unroll
        for (int i = 0; i < 3; ++i) bn[i] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(nx + 4 * i, 16));
        const float bw[12] = {bc[0][0], bc[0][1], bc[0][2], bc[0][3], bc[1][0], bc[1][1],
                              bc[1][2], bc[1][3], bc[2][0], bc[2][1], bc[2][2], bc[2][3]};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int tx = 0; tx < KS; ++tx)
            acc[ty * KS + tx] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bw[m + tx + 4 - PAD], acc[ty * KS + tx], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * KS, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) bc[i] = bn[i];
      }
    }
  }

  // ---- partial slab, layout of k_wgrad_mfma with CIW = 32: slab[part][g][co 64][ci 32][T]
  float* sl = p.slab + ((long)part * gridDim.y + g) * (64L * CIW * T);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int co = cog * 16 + 4 * kq + rr;
      const int ci = cib * 16 + l15;
      sl[((long)co * CIW + ci) * T + t] = acc[t][rr];
    }
}

// dw[k][c][dz*T + t] = sum_part slab[part][(dz*CB + c/CIW)*KBK + k/64][k%64][c%CIW][t]
__global__ void k_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ dw, int C, int K, int KS, int CIW,
                               int parts, int G) {
  const int T = KS * KS, taps = KS * T;
  const int CB = C / CIW, KBK = K / 64;
  const long total = (long)K * C * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps), c = (int)((i / taps) % C), k = (int)(i / ((long)taps * C));
    const int dz = tap / T, t = tap % T;
    const int g = (dz * CB + c / CIW) * KBK + k / 64;
    const long off = (long)g * (64L * CIW * T) + ((long)(k % 64) * CIW + (c % CIW)) * T + t;
    const long pstride = (long)G * (64L * CIW * T);
    float s = 0.f;
    for (int pidx = 0; pidx < parts; ++pidx) s += slab[off + pidx * pstride];
    dw[i] = s;
  }
}

struct WgPlan {
  int AB, PR, PAr, QT4, lds_bytes, G, parts;
};

static int pitch2(int n) {  // smallest p >= n with p % 32 == 2
  int p = (n / 32) * 32 + 2;
  return p >= n ? p : p + 32;
}

static bool wg_shape_ok(const ConvDims& d) {
  if (d.kd != d.kh || d.kh != d.kw) return false;
  if (d.kd != 3 && d.kd != 5) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.pd != d.kd / 2 || d.ph != d.kd / 2 || d.pw != d.kd / 2) return false;
  if (d.W > 64 * kNSEG || d.K % 64) return false;
  return true;
}

static bool plan_wgrad(const ConvDims& d, WgPlan& pl) {
  if (!wg_shape_ok(d)) return false;
  const int KS = d.kd;
  const int QT4 = (d.W + 3) & ~3;
  const int PR = pitch2(QT4 + KS + 1), PAr = pitch2(QT4 + 1);
  const int abList[2] = {KS == 3 ? 2 : 1, 1};
  for (int i = 0; i < 2; ++i) {
    const int AB = abList[i], CIW = 32 * AB;
    if (d.C % CIW) continue;
    const long bytes = ((long)KS * CIW * PR + 64L * PAr) * 4;
    if (bytes > kLdsMaxW) continue;
    pl.AB = AB; pl.PR = PR; pl.PAr = PAr; pl.QT4 = QT4; pl.lds_bytes = (int)bytes;
    pl.G = KS * (d.C / CIW) * (d.K / 64);
    int parts = 256 / pl.G;
    if (parts < 1) parts = 1;
    const long rows = (long)d.N * d.D * d.H;
    if (parts > rows) parts = (int)rows;
    pl.parts = parts;
    return true;
  }
  return false;
}

struct WdPlan {
  int AB, PR, PAr, SX, SD, NXB, XW, lds_bytes, G, parts;
};

static int pitch8(int n) {  // smallest m >= n with m % 16 == 8
  int m = (n + 7) & ~7;
  return (m & 8) ? m : m + 8;
}

// LDS-DMA variant: W % 4 == 0 (16-byte pieces never straddle the volume edge), per-lane offsets fit 32 bits
static bool plan_wgrad_dma(const ConvDims& d, WdPlan& pl) {
  if (!wg_shape_ok(d) || d.W % 4) return false;
  if ((long)d.D * d.H * d.W * 64 >= (1L << 31)) return false;
  const int KS = d.kd, RING = KS + 1;
  const int AB = (KS == 3 && d.C % 64 == 0) ? 2 : 1, CIW = 32 * AB;
  if (d.C % CIW) return false;
  for (int NXB = 1; NXB <= 8; ++NXB) {
    const int XW = ((d.W + NXB - 1) / NXB + 15) & ~15;  // k-iterations take 16 voxels
    // pitch = 8 (mod 16): the 16 lanes of every ds_read_b128 lane group fall on 16 distinct 4-bank groups; the X row
    // holds 4 halo columns, XW columns and the 12-float window of the read-ahead past the last iteration
    const int PR = pitch8(XW + 24), PAr = pitch8(XW);
    const int SX = (CIW * PR + 255) & ~255, SD = (64 * PAr + 255) & ~255;
    const long bytes = ((long)RING * SX + 2L * SD) * 4;
    if (bytes > kLdsMaxW || SX / 256 > 8 * kMaxPX || SD / 256 > 8 * kMaxPD) continue;
    pl.AB = AB; pl.PR = PR; pl.PAr = PAr; pl.SX = SX; pl.SD = SD; pl.NXB = NXB; pl.XW = XW;
    pl.lds_bytes = (int)bytes;
    pl.G = KS * (d.C / CIW) * (d.K / 64);
    int parts = 256 / pl.G;
    if (parts < 1) parts = 1;
    const long rows = (long)d.N * d.D * d.H;
    if (parts > rows) parts = (int)rows;
    pl.parts = parts;
    return true;
  }
  return false;
}

struct WrPlan {
  int R, Wg, PRc, PAc, SX, SD, lds_bytes, G, parts;
};

// multi-row kernel: 3^3, W <= 56, 32 | C, 64 | K
static bool plan_wgrad_rows(const ConvDims& d, WrPlan& pl) {
  if (!wg_shape_ok(d) || d.kd != 3 || d.W > 56 || d.C % 32) return false;
  if ((long)d.D * d.H * d.W * 64 >= (1L << 31)) return false;
  const int PAD = 1, KS = 3;
  const int Wg = (d.W + PAD + 3) & ~3;
  int R = 112 / Wg;
  if (R > 4) R = 4;  // one row tail per thread: 32 * (R + 2) + 64 * R <= 512
  if (R > d.H) R = d.H;
  if (R < 1) return false;
  const int nq16 = (R * Wg + 15) & ~15;
  int need = 4 + (R + 2 * PAD) * Wg;
  if ((KS - 1) * Wg + nq16 + 24 > need) need = (KS - 1) * Wg + nq16 + 24;
  const int PRc = pitch8(need), PAc = pitch8(nq16);
  const int SX = (32 * PRc + 255) & ~255, SD = (64 * PAc + 255) & ~255;
  const long bytes = (2L * SX + 2L * SD) * 4;
  if (bytes > kLdsMaxW || SX / 256 > 8 * kMaxRX || SD / 256 > 8 * kMaxRD) return false;
  pl.R = R; pl.Wg = Wg; pl.PRc = PRc; pl.PAc = PAc; pl.SX = SX; pl.SD = SD; pl.lds_bytes = (int)bytes;
  pl.G = KS * (d.C / 32) * (d.K / 64);
  // voxel partitions: one workgroup per CU (LDS), G * parts workgroups run in ceil(G * parts / 256) rounds of
  // ceil(units / parts) steps -- e.g. G = 96 at 27^3: 2 parts leave a quarter of the CUs idle, 8 parts = 3 full rounds
  const long units = (long)d.N * d.D * ((d.H + R - 1) / R);
  int best = 1;
  double best_cost = 1e30;
  for (int parts = 1; parts <= 64 && parts <= units; ++parts) {
    const double rounds = (double)cdiv((long)pl.G * parts, 256);
    const double cost = rounds * ((double)cdiv(units, parts) + 2.0);  // + slab write / start-up per workgroup
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = parts;
    }
  }
  pl.parts = best;
  return true;
}

bool mfma_wgrad_supported(const ConvDims& d) {
  WgPlan pl;
  WrPlan pr;
  return plan_wgrad(d, pl) || plan_wgrad_rows(d, pr);
}

size_t mfma_ws_bytes(const ConvDims& d) {
  size_t need = mfma_fwd_ws_bytes(d);  // packed weights + stream-K partial slots of the fwd / dgrad kernels
  WgPlan pl;
  if (plan_wgrad(d, pl)) {
    const size_t slab = (size_t)pl.parts * pl.G * 64 * (32 * pl.AB) * d.kd * d.kd * sizeof(float);
    if (slab > need) need = slab;
  }
  WdPlan pd;
  if (plan_wgrad_dma(d, pd)) {
    const size_t slab = (size_t)pd.parts * pd.G * 64 * (32 * pd.AB) * d.kd * d.kd * sizeof(float) + 256;
    if (slab > need) need = slab;
  }
  WrPlan pr;
  if (plan_wgrad_rows(d, pr)) {
    const size_t slab = (size_t)pr.parts * pr.G * 64 * 32 * d.kd * d.kd * sizeof(float) + 256;
    if (slab > need) need = slab;
  }
  return need;
}

template <int KS, int AB>
static int launch_wg(const WgParams& p, dim3 grid, int lds_bytes, hipStream_t s) {
  auto kern = k_wgrad_mfma<KS, AB>;
  if (int e = raise_dyn_lds(kern, kLdsMaxW, "wgrad_mfma")) return e;
  hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, s, p);
  return check_launch("wgrad_mfma");
}

template <int KS, int AB>
static int launch_wd(const WdParams& p, dim3 grid, int lds_bytes, hipStream_t s) {
  auto kern = k_wgrad_dma<KS, AB>;
  if (int e = raise_dyn_lds(kern, kLdsMaxW, "wgrad_dma")) return e;
  hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, s, p);
  return check_launch("wgrad_dma");
}

static unsigned wmagic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

static int conv_wgrad_dma(const float* x, const float* dy, float* dw, const ConvDims& d, const WdPlan& pl, void* ws,
                          size_t wsb, hipStream_t s) {
  const int CIW = 32 * pl.AB;
  const size_t slab = (size_t)pl.parts * pl.G * 64 * CIW * d.kd * d.kd * sizeof(float);
  if (!ws || wsb < slab + 256) {
    set_error("wgrad_dma: workspace too small (%zu < %zu)", wsb, slab + 256);
    return NC_ERR_WS;
  }
  WdParams p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws; p.zeros = nc_zero_page();
  if (!p.zeros) { set_error("wgrad: no zero page"); return NC_ERR_HIP; }
  p.C = d.C; p.K = d.K; p.N = d.N; p.D = d.D; p.H = d.H; p.W = d.W;
  p.PR = pl.PR; p.PAr = pl.PAr; p.SX = pl.SX; p.SD = pl.SD; p.NXB = pl.NXB; p.XW = pl.XW;
  p.CB = d.C / CIW; p.KBK = d.K / 64; p.parts = pl.parts;
  p.mPR = wmagic(pl.PR); p.mPAr = wmagic(pl.PAr);
  dim3 grid(pl.parts, pl.G);
  int e;
  if (d.kd == 3 && pl.AB == 2) e = launch_wd<3, 2>(p, grid, pl.lds_bytes, s);
  else if (d.kd == 3) e = launch_wd<3, 1>(p, grid, pl.lds_bytes, s);
  else e = launch_wd<5, 1>(p, grid, pl.lds_bytes, s);
  if (e) return e;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(1024), dim3(256), 0, s, (const float*)ws, dw, d.C, d.K, d.kd, CIW, pl.parts,
                     pl.G);
  return check_launch("wgrad_reduce");
}

static int conv_wgrad_rows(const float* x, const float* dy, float* dw, const ConvDims& d, const WrPlan& pl, void* ws,
                           size_t wsb, hipStream_t s) {
  const size_t slab = (size_t)pl.parts * pl.G * 64 * 32 * d.kd * d.kd * sizeof(float);
  if (!ws || wsb < slab + 256) {
    set_error("wgrad_rows: workspace too small (%zu < %zu)", wsb, slab + 256);
    return NC_ERR_WS;
  }
  WrParams p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws; p.zeros = nc_zero_page();
  if (!p.zeros) { set_error("wgrad: no zero page"); return NC_ERR_HIP; }
  p.C = d.C; p.K = d.K; p.N = d.N; p.D = d.D; p.H = d.H; p.W = d.W;
  p.R = pl.R; p.Wg = pl.Wg; p.PRc = pl.PRc; p.PAc = pl.PAc; p.SX = pl.SX; p.SD = pl.SD;
  p.CB = d.C / 32; p.KBK = d.K / 64; p.parts = pl.parts;
  p.mPRc = wmagic(pl.PRc); p.mPAc = wmagic(pl.PAc); p.mWg = wmagic(pl.Wg);
  auto kern = k_wgrad_rows<3>;
  if (int e = raise_dyn_lds(kern, kLdsMaxW, "wgrad_rows")) return e;
  hipLaunchKernelGGL(kern, dim3(pl.parts, pl.G), dim3(512), pl.lds_bytes, s, p);
  if (int e = check_launch("wgrad_rows")) return e;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(1024), dim3(256), 0, s, (const float*)ws, dw, d.C, d.K, d.kd, 32, pl.parts,
                     pl.G);
  return check_launch("wgrad_reduce");
}

int conv_wgrad_mfma(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s) {
  WrPlan pr;
  if (plan_wgrad_rows(d, pr)) return conv_wgrad_rows(x, dy, dw, d, pr, ws, wsb, s);
  WdPlan pd;
  if (plan_wgrad_dma(d, pd)) return conv_wgrad_dma(x, dy, dw, d, pd, ws, wsb, s);
  WgPlan pl;
  if (!plan_wgrad(d, pl)) {
    set_error("wgrad_mfma: unsupported shape");
    return NC_ERR_SHAPE;
  }
  const int CIW = 32 * pl.AB;
  const size_t need = (size_t)pl.parts * pl.G * 64 * CIW * d.kd * d.kd * sizeof(float);
  if (!ws || wsb < need) {
    set_error("wgrad_mfma: workspace too small (%zu < %zu)", wsb, need);
    return NC_ERR_WS;
  }
  WgParams p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws;
  p.C = d.C; p.K = d.K; p.N = d.N; p.D = d.D; p.H = d.H; p.W = d.W;
  p.PR = pl.PR; p.PAr = pl.PAr; p.QT4 = pl.QT4;
  p.CB = d.C / CIW; p.KBK = d.K / 64; p.parts = pl.parts;
  dim3 grid(pl.parts, pl.G);
  int e;
  if (d.kd == 3 && pl.AB == 2) e = launch_wg<3, 2>(p, grid, pl.lds_bytes, s);
  else if (d.kd == 3) e = launch_wg<3, 1>(p, grid, pl.lds_bytes, s);
  else e = launch_wg<5, 1>(p, grid, pl.lds_bytes, s);
  if (e) return e;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(1024), dim3(256), 0, s, (const float*)ws, dw, d.C, d.K, d.kd, CIW, pl.parts,
                     pl.G);
  return check_launch("wgrad_reduce");
}

}  // namespace nc
