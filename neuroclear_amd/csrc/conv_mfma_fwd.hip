// Implicit-GEMM Conv3d (odd cubic kernel 3^3 / 5^3, stride 1, "same" padding) on the fp32 matrix cores of gfx950.
// Serves forward AND dgrad (dgrad = the same convolution of dy with the flipped, channel-transposed kernel).
// Replaces nn.Conv3d forward/backward-data at models/networks.py:420-425,442,460-469 (U-Net) and :900-902 (G_B).
//
// GEMM view:  Y[co][v] = sum_{ci,tap} Wp[(ci,tap)][co] * X[ci][v + tap]      (M = co, N = voxels, K = Cin * taps)
//   * v_mfma_f32_32x32x2_f32: exact fp32 (fmaf chain), 64 cycles per instruction per SIMD -- the same peak as the
//     fp32 VALU (157 TFLOP/s) but with ONE instruction per 4096 FLOP, so staging/address work hides behind it.
//   * A operand (weights) : pre-packed [K-row][co-pair-interleaved] in HBM (a few 100 KB, L1/L2 resident); each lane
//     loads one float2 per k-step = its two 32-wide co blocks.  No LDS for weights; the stream is software-pipelined
//     by hand one kernel row (3-6 k MFMA cycles) ahead.
//   * B operand (inputs)  : an input brick of CK channels x (Tz+2p) planes x (Ty+2p) rows x (W+2p) columns in LDS, rows
//     FLATTENED with pitch P = W+2p.  A tap (dz,dy,dx) is then a constant LDS offset dz*RW + dy*P + dx, and one MFMA
//     column block is 32 consecutive floats -> ds_read_b32, conflict-free for any alignment.  Output positions that
//     fall on the 2p pad columns are computed and discarded (<= 2p/P waste) -- this is what makes 108/140/54/27-wide
//     volumes tile without 32-alignment.
//   * staging by LDS-DMA: the brick of the next unit goes global -> LDS directly (16 bytes per lane, per-lane source
//     address, zero page for padding) while the current unit is multiplied; see the comment at stage_dma.
//   * STREAM-K over (tile, channel-chunk) units: 256 persistent workgroups (one per CU) each take an equal, contiguous
//     share of all units, so the wave of equal tiles that does not divide by the CU count (1296 tiles = 5.06 rounds at
//     108^3) no longer costs a whole extra round.  A tile whose chunks are split between two workgroups is written
//     as partial accumulators to a workspace slot and summed in chunk order by a fix-up kernel (deterministic, no
//     atomics); complete tiles are stored directly.  (Measured alternative: tiles / 256 whole tiles per workgroup
//     and only the leftover tiles split -- half the partial traffic, but 0.2 % slower end to end.)
//   * epilogue: accumulator rows are co, lanes are voxels -> each store instruction writes 2 x 128 B contiguous.
#include "common.hpp"

namespace nc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));


static constexpr int kNumWG = 256;        // persistent workgroups = CUs of an MI355X

struct FwdParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  float* part;        // partial-accumulator slots [2 * kNumWG][32 * VB][NT]
  const float* mean;  // optional fused normalize-on-load of the input: act((x - mean[c]) * rstd[c])
  const float* rstd;
  float slope;
  int C, K, D, H, W;
  int Tz, Ty, nty, ntz, ncot;
  int P, RW, planes, CP;  // row pitch, floats per plane (= (Ty+2p)*P), planes per channel, floats per channel
  int nelem;              // CK * CP
  unsigned mP;            // magic multiplier: n / P == __umulhi(n, mP)
  int nrows;                // brick rows per unit: CK * planes * (Ty + 2p)   (TAIL kernels: one row tail per thread slot)
  unsigned mRowsY, mPRows;  // magic multipliers for / rowsY and / (planes * rowsY)
  int rowsY, PRows;
  int npieces;              // pieces of 64*V floats that cover the brick (nelem rounded up)
  int SB;                   // floats between the two brick buffers (rounded nelem + 4 zero floats)
  unsigned mCP, mRW;        // magic multipliers for / CP and / RW
  const float* zeros;       // >= 16 B of zeros in global memory: source of pad columns and out-of-volume rows
  int nchunks;
  long units;  // tiles * nchunks
};

__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned m) { return __umulhi(n, m); }

struct TileId {
  int n, cot, z0, y0;
};
__device__ __forceinline__ TileId decode_tile(const FwdParams& p, long tile) {
  const int nsp = p.ntz * p.nty;
  const int sp = (int)(tile % nsp);
  const long rest = tile / nsp;
  TileId t;
  t.cot = (int)(rest % p.ncot);
  t.n = (int)(rest / p.ncot);
  t.z0 = (sp / p.nty) * p.Tz;
  t.y0 = (sp % p.nty) * p.Ty;
  return t;
}

// C/D layout of 32x32 MFMA: column (voxel) = lane & 31, row (co) = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int WM, int WN, int VB>
__device__ __forceinline__ void store_tile(const FwdParams& p, const f32x16 (&acc)[2][VB], const TileId& t, int wm,
                                           int wn, int li, int h) {
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const int GP = WM / p.Tz;
  const int tzl = wm / GP, blk0 = (wm % GP) * VB;
  const int cob = (t.cot * WN + wn) * 64;
  const int z = t.z0 + tzl;
  if (z >= p.D) return;
  float* yn = p.y + ((long)t.n * p.K + cob) * S + (long)z * HW;
#pragma unroll
  for (int v = 0; v < VB; ++v) {
    const unsigned q = (blk0 + v) * 32 + li;
    const unsigned ty = fastdiv(q, p.mP);
    const unsigned x = q - ty * p.P;
    const int y = t.y0 + (int)ty;
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

// NC_ABLATE (timing experiments only, tools/ablate_fwd.py; results are garbage when set): bit 0 no staging after the
// first unit, bit 1 no weight prefetch, bit 2 B operand not read from LDS, bit 3 no barrier between units, bit 4 every brick row
// is fetched from the same (cache-resident) address
#ifndef NC_ABLATE
#define NC_ABLATE 0
#endif

template <int KS, int CK, int WM, int WN, int VB, bool TAIL>
__global__ __launch_bounds__(WM* WN * 64) void k_conv_mfma(FwdParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int NWV = NT / 64;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS * KS;
  // k-steps per (dz, dy) row of the kernel.  CK == 1 (single input channel): the two k values of an MFMA are the
  // taps dx = 2s, 2s + 1 of the row instead of two channels (dx = KS is a zero weight)
  constexpr int U = CK == 1 ? (KS + 1) / 2 : KS * (CK / 2);
  constexpr int WROWS = CK == 1 ? KS * KS * 2 * U : TAPS * CK;  // packed weight rows per channel chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, h = lane >> 5;
  const int wg = blockIdx.x;
  const long u0 = p.units * wg / kNumWG, u1 = p.units * (wg + 1) / kNumWG;
  if (u0 >= u1) return;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;

  // ---- staging straight into LDS (LDS-DMA, global_load_lds): no staging registers, no LDS store pass; the loads of
  //      unit u+1 are issued before the MFMA loop of unit u and land in the other buffer meanwhile.
  //      The brick [CK][planes][rows][P] is one contiguous LDS range cut into pieces of 64*V floats; one instruction
  //      moves a piece: lane l fetches V consecutive floats from ITS OWN global address and the hardware writes them
  //      at piece base + l*V.  A row is W data floats followed by P-W zeros (the zeros are the x-padding of this row
  //      AND of the next one: rows are flattened); pad floats, rows outside the volume and the tail of the last
  //      piece are fetched from a zero page, so there is no zero-fill pass and no branch.  V = 4 (16 B per lane,
  //      1 KiB per instruction) when W % 4 == 0, else 1.  An LDS-DMA instruction costs 60-100 issue cycles whatever
  //      its size (MI355X_MICROARCH.md constants table), so piece count is what matters: 8 per wave per unit at
  //      108^3 instead of 34 row segments.
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // W % 4 != 0 (TAIL kernels): the group of 4 floats that holds a row's last W % 4 values would also pull the first
  // floats of the next image row into the zero padding, so the DMA lane of that group is switched off (EXEC) and the
  // group is written by an ordinary 16-byte LDS store -- [tail values, zeros] -- from registers that thread
  // t (+ NT) loads for brick row t (+ NT) alongside the DMA; the store waits for the end of the unit.
  const int tw = p.W & 3, wfull = p.W - tw;
  constexpr int TR = TAIL ? (VB == 4 ? 1 : 2) : 0;  // row tails per thread (nrows <= TR * NT; VB = 4 has no spare registers)
  int tl_lofs[TR + 1], tl_gofs[TR + 1], tl_zy[TR + 1];
  float tl_v[TR + 1][3];
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const unsigned row = tid + NT * i;
      tl_lofs[i] = -1;
      tl_gofs[i] = 0;
      tl_zy[i] = 0;
      if ((int)row < p.nrows) {
        const unsigned cic = fastdiv(row, p.mPRows);
        const unsigned r1 = row - cic * p.PRows;
        const unsigned pz = fastdiv(r1, p.mRowsY);
        const unsigned yy = r1 - pz * p.rowsY;
        tl_lofs[i] = (int)(cic * p.CP + pz * p.RW + yy * p.P) + wfull;
        tl_gofs[i] = (int)((long)cic * S + ((long)pz - PAD) * HW + ((long)yy - PAD) * p.W) + wfull;
        tl_zy[i] = (int)((pz << 16) | yy);
      }
    }
  }
  auto stage_dma = [&](const TileId& t, int chunk, float* bd) {
    const float* xt = p.x + ((long)t.n * p.C + (long)chunk * CK) * S + (long)t.z0 * HW + (long)t.y0 * p.W;
#pragma unroll 1
    for (int j = wave; j < p.npieces; j += NWV) {
      const unsigned f = (unsigned)(j * 64 + lane) * 4;
      const unsigned cic = fastdiv(f, p.mCP);
      const unsigned r1 = f - cic * p.CP;
      const unsigned pz = fastdiv(r1, p.mRW);
      const unsigned r2 = r1 - pz * p.RW;
      const unsigned yy = fastdiv(r2, p.mP);
      const unsigned x = r2 - yy * p.P;
      const int z = t.z0 + (int)pz - PAD, y = t.y0 + (int)yy - PAD;
      const bool ok = cic < (unsigned)CK && (int)x < wfull && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H;
      const long off = (long)cic * S + ((long)pz - PAD) * HW + ((long)yy - PAD) * p.W + x;
      const float* src = (NC_ABLATE & 16) ? p.x + lane * 4 : (ok ? xt + off : p.zeros);
      float* dst = bd + j * 64 * 4;  // wave-uniform
      if (!TAIL || cic >= (unsigned)CK || (int)x != wfull)
        nc_dma_lds16(src, nc_lds_addr(dst));
    }
    if constexpr (TAIL) {
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int z = t.z0 + (tl_zy[i] >> 16) - PAD, y = t.y0 + (tl_zy[i] & 0xffff) - PAD;
        const bool ok = tl_lofs[i] >= 0 && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H;
        const float* r = xt + tl_gofs[i];
#pragma unroll
        for (int e = 0; e < 3; ++e) tl_v[i][e] = (ok && e < tw) ? r[e] : 0.f;
      }
    }
  };
  auto write_tails = [&](float* bd) {
    if constexpr (TAIL) {
#pragma unroll
      for (int i = 0; i < TR; ++i)
        if (tl_lofs[i] >= 0)
          *reinterpret_cast<f32x4v*>(__builtin_assume_aligned(bd + tl_lofs[i], 16)) =
              f32x4v{tl_v[i][0], tl_v[i][1], tl_v[i][2], 0.f};
    }
  };
  for (int i = tid; i < 4 + 2 * p.SB; i += NT) lds[i] = 0.f;  // the 4 zero floats in front of each buffer stay zero
  __syncthreads();

  // ---- this wave's output sub-tile: VB column blocks of 32 flattened positions in plane tzl, 64 output channels
  const int GP = WM / p.Tz;
  const int tzl = wm / GP, blk0 = (wm % GP) * VB;
  const int b_base = (CK == 1 ? h : h * p.CP) + tzl * p.RW + blk0 * 32 + li;
  // Weights go through a buffer descriptor: 32-bit per-lane offset (constant), SCALAR offset for (tile, chunk, row,
  // k-step) -- the weight stream needs no vector address arithmetic in the MFMA loop.
  const int avoff = (li * 2 + h * p.K) * 4;  // bytes
  const int kstep4 = 2 * p.K * 4;            // bytes of packed weights per k-step
  const int chunk_stride4 = WROWS * p.K * 4;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, 0x7fffffff, 0x00020000);
  auto wbase = [&](int cot, int chunk) { return (cot * WN + wn) * 64 * 4 + chunk * chunk_stride4; };
  auto wload = [&](int soff) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i r = __builtin_amdgcn_raw_buffer_load_b64(wrsrc, avoff, soff, 0);
    return make_float2(__int_as_float(r.x), __int_as_float(r.y));
  };

  f32x16 acc[2][VB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;
  };
  zero_acc();

  float* buf0 = lds + 4;  // brick data start (16 B aligned); the 4 floats in front of it are zeros (x = -p of row 0)
  float* buf1 = buf0 + p.SB;

  // ---- first unit
  long tile = u0 / p.nchunks;
  int chunk = (int)(u0 - tile * p.nchunks);
  const long first_tile = tile;
  int first_chunk = chunk;  // first chunk of the current tile that THIS workgroup accumulates
  TileId tid_cur = decode_tile(p, tile);
  stage_dma(tid_cur, chunk, buf0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  write_tails(buf0);
  __syncthreads();

  // The packed weights of one tile are ONE linear stream: k-step s reads rows 2s, 2s+1 (lane half h picks the row),
  // chunk after chunk, tap after tap; prefetched one (dz, dy) kernel row ahead (the last prefetch of the launch runs
  // into the workspace slack the host reserves behind the packed weights).
  int aptr = wbase(tid_cur.cot, chunk);  // scalar byte offset into the packed weights
  float2 a_cur[U], a_nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a_cur[u] = wload(aptr + u * kstep4);
  aptr += U * kstep4;

  int parity = 0;
  for (long u = u0; u < u1; ++u) {
    const float* cur = (parity ? buf1 : buf0) - PAD;  // column -p of the brick's first row
    float* nxt = parity ? buf0 : buf1;
    // ---- next unit (uniform bookkeeping)
    const bool more = u + 1 < u1;
    long ntile = tile;
    int nchunk = chunk + 1;
    if (nchunk == p.nchunks) {
      nchunk = 0;
      ntile = tile + 1;
    }
    const bool tile_ends = !more || ntile != tile;
    TileId tid_nxt = tid_cur;
    if (more && ntile != tile) tid_nxt = decode_tile(p, ntile);
    // (measured alternatives: staggering the two waves of a SIMD -- w stages at row 0, w+4 mid-unit -- no gain;
    //  issuing one piece per kernel row instead of all eight here: 2 % slower)
    if (more && !(NC_ABLATE & 1)) stage_dma(tid_nxt, nchunk, nxt);
    const int next_aptr = more ? wbase(tid_nxt.cot, nchunk) : aptr;

    // One kernel row (dz, dy) = U k-steps of 2*VB MFMAs.  Software pipeline, pinned with sched_group_barrier so the
    // machine scheduler cannot sink the loads next to their uses: the U weight loads of the NEXT row go out first
    // (a whole row of MFMAs, 3-6 k cycles, covers their L2 latency), and the VB LDS reads of k-step s+1 are issued
    // in front of the 2*VB MFMAs of k-step s (the first k-step of the next row included).
    const float* brow = cur + b_base;
    float bc[VB], bn[VB];
#pragma unroll
    for (int v = 0; v < VB; ++v) bc[v] = (NC_ABLATE & 4) ? (float)v : brow[v * 32];
#pragma unroll 1
    for (int dz = 0; dz < KS; ++dz) {
#pragma unroll 1
      for (int dy = 0; dy < KS; ++dy) {
        const bool last_row = dz == KS - 1 && dy == KS - 1;
        if (last_row) aptr = next_aptr;  // the row after this unit's last row
#pragma unroll
        for (int uu = 0; uu < U; ++uu)
          a_nxt[uu] = (NC_ABLATE & 2) ? a_cur[uu] : wload(aptr + uu * kstep4);
        aptr += U * kstep4;
        __builtin_amdgcn_sched_group_barrier(0x020, U, 0);  // VMEM reads first
        // first k-step of the next row (dy+1, or dz+1 / dy 0); after the unit's last row: any in-range address
        const float* brow_nxt = last_row ? brow : (dy == KS - 1 ? brow + p.RW - (KS - 1) * p.P : brow + p.P);
#pragma unroll
        for (int st = 0; st < U; ++st) {
          const float* src;
          float2 a;
          if constexpr (CK == 1) {
            src = st + 1 < U ? brow + 2 * (st + 1) : brow_nxt;
            a = a_cur[st];
          } else {
            const int dx = st / (CK / 2), cp = st % (CK / 2);
            const int dxn = (st + 1) / (CK / 2), cpn = (st + 1) % (CK / 2);
            src = st + 1 < U ? brow + dxn + 2 * cpn * p.CP : brow_nxt;
            a = a_cur[dx * (CK / 2) + cp];
          }
#pragma unroll
          for (int v = 0; v < VB; ++v) bn[v] = (NC_ABLATE & 4) ? bc[v] : src[v * 32];
#pragma unroll
          for (int v = 0; v < VB; ++v) {
            acc[0][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bc[v], acc[0][v], 0, 0, 0);
            acc[1][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bc[v], acc[1][v], 0, 0, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, VB, 0);      // DS reads of the next k-step ...
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * VB, 0);  // ... then this k-step's MFMAs
#pragma unroll
          for (int v = 0; v < VB; ++v) bc[v] = bn[v];
        }
        brow = brow_nxt;
#pragma unroll
        for (int uu = 0; uu < U; ++uu) a_cur[uu] = a_nxt[uu];
      }
    }
    if (!(NC_ABLATE & 8)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every LDS-DMA row of the next unit has landed
      if (more) write_tails(nxt);
      __syncthreads();
    }
    parity ^= 1;

    if (tile_ends) {
      if (first_chunk == 0 && chunk == p.nchunks - 1) {
        store_tile<WM, WN, VB>(p, acc, tid_cur, wm, wn, li, h);
      } else {  // partial: slot 0 = this workgroup's first tile, slot 1 = its last tile
        float* slot = p.part + ((long)(2 * wg + (tile == first_tile ? 0 : 1)) * (32 * VB)) * NT + tid;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int v = 0; v < VB; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) slot[(long)((a * VB + v) * 16 + r) * NT] = acc[a][v][r];
      }
      zero_acc();
      first_chunk = nchunk;
    }
    tile = ntile;
    chunk = nchunk;
    tid_cur = tid_nxt;
  }
}

// Fix-up: one workgroup per boundary between two persistent workgroups; if that boundary is the first one that cuts
// tile t, the workgroup adds the partial accumulators of every workgroup that touched t, in chunk order, and stores t.
template <int WM, int WN, int VB>
__global__ __launch_bounds__(WM* WN * 64) void k_conv_fixup(FwdParams p) {
  constexpr int NT = WM * WN * 64;
  const int b = blockIdx.x + 1;
  const long ub = p.units * b / kNumWG;
  if (ub % p.nchunks == 0) return;  // boundary on a tile edge: nothing is split here
  const long t = ub / p.nchunks;
  const long tb = t * p.nchunks, te = tb + p.nchunks;
  const long uprev = p.units * (b - 1) / kNumWG;
  if (b > 1 && uprev > tb) return;  // an earlier boundary already cuts this tile: that workgroup does the work
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  f32x16 acc[2][VB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int v = 0; v < VB; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;
  for (int w = b - 1; w < kNumWG; ++w) {
    const long w0 = p.units * w / kNumWG, w1 = p.units * (w + 1) / kNumWG;
    if (w0 >= te) break;
    if (w0 >= w1 || w1 <= tb) continue;
    const float* slot = p.part + ((long)(2 * w + (w0 / p.nchunks == t ? 0 : 1)) * (32 * VB)) * NT + tid;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][v][r] += slot[(long)((a * VB + v) * 16 + r) * NT];
  }
  const TileId tidx = decode_tile(p, t);
  store_tile<WM, WN, VB>(p, acc, tidx, wm, wn, lane & 31, lane >> 5);
}

// Weight packing.  Row = chunk*TAPS*CK + tap*CK + cic  (ci = chunk*CK + cic); inside a row: [K/64][32][2] with
// (blk, i, s) <-> co = blk*64 + s*32 + i.   mode 0: fwd  src[co][ci][tap];   mode 1: dgrad  src[ci][co][TAPS-1-tap].
// Also writes the 64-float zero page (at float offset zofs, in the slack behind the packed weights) that the conv
// kernel's LDS-DMA uses as the source of padding.
__global__ void k_pack_w(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int taps, int CK,
                         int mode, long zofs) {
  const long total = (long)Cin * Cout * taps;
  if (blockIdx.x == 0 && threadIdx.x < 64) wp[zofs + threadIdx.x] = 0.f;
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

// Single-input-channel packing: row = (dz * KS + dy) * 2U + dx2, dx2 < 2U = KS + 1 (the odd tap out is a zero weight).
__global__ void k_pack_w_c1(const float* __restrict__ w, float* __restrict__ wp, int Cout, int KS, long zofs) {
  const int R2 = KS + 1;  // 2U, KS odd
  const long total = (long)KS * KS * R2 * Cout;
  if (blockIdx.x == 0 && threadIdx.x < 64) wp[zofs + threadIdx.x] = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % Cout);
    const int row = (int)(i / Cout);
    const int dx = row % R2, zy = row / R2;
    const int blk = col / 64, ii = (col % 64) / 2, s = col & 1;
    const int co = blk * 64 + s * 32 + ii;
    wp[i] = dx < KS ? w[((long)co * KS * KS + zy) * KS + dx] : 0.f;
  }
}

static size_t packed_floats(int KS, int Cin, int Cout) {
  return Cin == 1 ? (size_t)Cout * KS * KS * (KS + 1) : (size_t)Cin * Cout * KS * KS * KS;
}

struct FwdPlan {
  int cfg, CK, Tz, Ty, nty, ntz, P, RW, planes, CP, nelem, lds_bytes;
  double cost;
};

struct Cfg { int WM, WN, VB; };
static const Cfg kCfgs[] = {{8, 1, 4}, {8, 1, 2}, {4, 2, 2}, {2, 4, 2}, {2, 4, 1}};
static constexpr int kNumCfgs = 5;
static constexpr int kLdsMax = 160 * 1024;

static unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

// Cin = reduction channels, Cout = produced channels (for dgrad the roles of C and K are swapped by the caller).
static bool plan_fwd(int KS, int Cin, int Cout, int N, int D, int H, int W, FwdPlan& best) {
  const int pad = KS / 2;
  const int V = 4;
  const int P = (W + pad + V - 1) / V * V;  // row pitch: W data floats + >= pad zeros, a whole number of DMA lanes
  const bool tail = (W & 3) != 0;
  if ((long)D * H * W * 8 >= (1L << 31)) return false;
  bool found = false;
  for (int c = 0; c < kNumCfgs; ++c) {
    const Cfg& g = kCfgs[c];
    if (Cout % (g.WN * 64)) continue;
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
        const int CK = Cin == 1 ? 1 : ckList[k];
        if (Cin == 1 && k) break;
        if (KS == 3 && CK == 2) continue;  // instantiated: KS=3 -> {8,4,1}; KS=5 -> {4,2}; KS=7 -> {1}
        if (KS == 5 && (CK == 8 || CK == 1)) continue;
        if (KS == 7 && CK != 1) continue;
        if (Cin % CK) continue;
        if (g.VB == 4 && CK > (KS == 3 ? 4 : 2)) continue;  // those instantiations spill
        if (tail && CK == 1) continue;
        if (tail && CK * planes * (Ty + 2 * pad) > (g.VB == 4 ? 1 : 2) * 64 * g.WM * g.WN) continue;  // row tails per thread
        const long nelem = (long)CK * CP;
        // slack: garbage columns of the last blocks read up to maxpos + (KS-1)*(P+1) past the last plane's start
        const long slack = maxpos + (long)(KS - 1) * (P + 1) + 64;
        const long SB = (nelem + 64 * V - 1) / (64 * V) * (64 * V) + 4;
        const long bytes = (4 + 2 * SB + slack) * 4;
        if (bytes > kLdsMax) continue;
        const long tiles = (long)ntz * nty * (Cout / (g.WN * 64)) * N;
        // stream-K: every workgroup gets units/256 units, a unit costs ~ VB * CK MFMA groups (+ a barrier)
        const long units = tiles * (Cin / CK);
        const double per_wg = (double)((units + kNumWG - 1) / kNumWG);
        // + a fixed per-unit overhead (barrier, staging write, loop set-up) that favours fatter units
        const double cost = per_wg * (g.VB * (CK == 1 ? 4 : CK) + 1.5);
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

static size_t part_bytes(const Cfg& g) { return (size_t)2 * kNumWG * 32 * g.VB * (g.WM * g.WN * 64) * sizeof(float); }

template <int KS, int CK, int WM, int WN, int VB, bool TAIL>
static int launch_one_t(const FwdParams& p, int lds_bytes, hipStream_t s) {
  auto kern = k_conv_mfma<KS, CK, WM, WN, VB, TAIL>;
  if (int e = raise_dyn_lds(kern, kLdsMax, "conv_mfma")) return e;
  hipLaunchKernelGGL(kern, dim3(kNumWG), dim3(WM * WN * 64), lds_bytes, s, p);
  if (int e = check_launch("conv_mfma")) return e;
  hipLaunchKernelGGL((k_conv_fixup<WM, WN, VB>), dim3(kNumWG - 1), dim3(WM * WN * 64), 0, s, p);
  return check_launch("conv_mfma_fixup");
}

template <int KS, int CK, int WM, int WN, int VB>
static int launch_one(const FwdParams& p, int lds_bytes, hipStream_t s) {
  if (p.W & 3) {
    if constexpr (CK == 1 || (VB == 4 && CK == 8)) {  // not instantiated: the planner keeps W % 4 != 0 off these
      set_error("conv_mfma: no tail kernel for this configuration");
      return NC_ERR_SHAPE;
    } else {
      return launch_one_t<KS, CK, WM, WN, VB, true>(p, lds_bytes, s);
    }
  }
  return launch_one_t<KS, CK, WM, WN, VB, false>(p, lds_bytes, s);
}

template <int KS, int CK>
static int launch_cfg(int cfg, const FwdParams& p, int lds, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_one<KS, CK, 8, 1, 4>(p, lds, s);
    case 1: return launch_one<KS, CK, 8, 1, 2>(p, lds, s);
    case 2: return launch_one<KS, CK, 4, 2, 2>(p, lds, s);
    case 3: return launch_one<KS, CK, 2, 4, 2>(p, lds, s);
    default: return launch_one<KS, CK, 2, 4, 1>(p, lds, s);
  }
}

static bool shape_ok(const ConvDims& d, int Cin, int Cout) {
  if (d.kd != d.kh || d.kh != d.kw) return false;
  if (Cin == 1) {  // single input channel (fwd only): 3^3 and 7^3
    if (d.kd != 3 && d.kd != 7) return false;
  } else if (d.kd != 3 && d.kd != 5) {
    return false;
  }
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.pd != d.kd / 2 || d.ph != d.kd / 2 || d.pw != d.kd / 2) return false;
  if ((Cin != 1 && Cin % 4 != 0) || Cout % 64 != 0) return false;
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

// bytes of workspace the fwd / dgrad MFMA kernels need for this shape: packed weights + prefetch slack + partial slots
size_t mfma_fwd_ws_bytes(const ConvDims& d) {
  size_t need = 0;
  FwdPlan pl;
  const size_t pack = (packed_floats(d.kd, d.C, d.K) * sizeof(float) + kPackSlackBytes + 255) & ~(size_t)255;
  if (shape_ok(d, d.C, d.K) && plan_fwd(d.kd, d.C, d.K, d.N, d.D, d.H, d.W, pl))
    need = pack + part_bytes(kCfgs[pl.cfg]);
  if (shape_ok(d, d.K, d.C) && plan_fwd(d.kd, d.K, d.C, d.N, d.D, d.H, d.W, pl)) {
    const size_t b = pack + part_bytes(kCfgs[pl.cfg]);
    if (b > need) need = b;
  }
  return need;
}

static int run(const float* x, const float* w, const float* bias, float* y, int Cin, int Cout, const ConvDims& d,
               int mode, void* ws, size_t wsb, hipStream_t s) {
  FwdPlan pl;
  if (!plan_fwd(d.kd, Cin, Cout, d.N, d.D, d.H, d.W, pl)) {
    set_error("conv_mfma: no tile plan for this shape");
    return NC_ERR_SHAPE;
  }
  const Cfg& g = kCfgs[pl.cfg];
  const int taps = d.kd * d.kh * d.kw;
  const size_t pfl = packed_floats(d.kd, Cin, Cout);
  const size_t pack = (pfl * sizeof(float) + kPackSlackBytes + 255) & ~(size_t)255;
  const size_t need = pack + part_bytes(g);
  if (!ws || wsb < need) {
    set_error("conv_mfma: workspace too small (%zu < %zu)", wsb, need);
    return NC_ERR_WS;
  }
  float* wp = (float*)ws;
  const long zofs = (long)(((pfl * sizeof(float) + 255) & ~(size_t)255) / sizeof(float));
  if (Cin == 1) hipLaunchKernelGGL(k_pack_w_c1, dim3(256), dim3(256), 0, s, w, wp, Cout, d.kd, zofs);
  else hipLaunchKernelGGL(k_pack_w, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, taps, pl.CK, mode, zofs);
  if (int e = check_launch("pack_w")) return e;
  FwdParams p{};
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.part = (float*)((char*)ws + pack);
  p.mean = nullptr; p.rstd = nullptr; p.slope = 0.f;
  p.C = Cin; p.K = Cout; p.D = d.D; p.H = d.H; p.W = d.W;
  p.Tz = pl.Tz; p.Ty = pl.Ty; p.nty = pl.nty; p.ntz = pl.ntz; p.ncot = Cout / (g.WN * 64);
  p.P = pl.P; p.RW = pl.RW; p.planes = pl.planes; p.CP = pl.CP;
  p.nelem = pl.nelem; p.mP = magic(pl.P);
  p.npieces = (pl.nelem + 255) / 256;
  p.SB = p.npieces * 256 + 4;
  p.rowsY = pl.Ty + 2 * (d.kd / 2); p.PRows = pl.planes * p.rowsY; p.nrows = pl.CK * p.PRows;
  p.mRowsY = magic(p.rowsY); p.mPRows = magic(p.PRows);
  p.mCP = magic(pl.CP); p.mRW = magic(pl.RW);
  p.zeros = wp + zofs;
  p.nchunks = Cin / pl.CK;
  p.units = (long)d.N * p.ncot * pl.ntz * pl.nty * p.nchunks;
  if (d.kd == 7) return launch_cfg<7, 1>(pl.cfg, p, pl.lds_bytes, s);
  if (d.kd == 3) {
    if (pl.CK == 1) return launch_cfg<3, 1>(pl.cfg, p, pl.lds_bytes, s);
    if (pl.CK == 8) return launch_cfg<3, 8>(pl.cfg, p, pl.lds_bytes, s);
    return launch_cfg<3, 4>(pl.cfg, p, pl.lds_bytes, s);
  }
  if (pl.CK == 4) return launch_cfg<5, 4>(pl.cfg, p, pl.lds_bytes, s);
  return launch_cfg<5, 2>(pl.cfg, p, pl.lds_bytes, s);
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
