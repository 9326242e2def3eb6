// deep_linear_gen (reference models/networks.py:893-917), layers 1 .. 5 as ONE position-typed 7^3 convolution 64 -> 1 of act0 (round 6,
// DESIGN.md 4.7; the algebra is tests/test_collapse_algebra.py::test_layers_1_to_5_as_one_position_typed_7x7x7_kernel, fp64 against autograd).
//
// The collapsed tail is a 3^3 kernel E (64 -> 1) on act1 = W1 (*) act0 (5^3, 64 -> 64), and the only thing that keeps E o W1 from being one
// 7^3 kernel is the zero padding of act1 -- which matters ON the faces only (E reaches one voxel) and there just removes the taps of E that
// point outside.  So with the voxel's position TYPE tau (per axis: 0 = on the low face, 1 = inside, 2 = on the high face; 27 types, 13 = interior)
//     y[v] = sum_{c, d} H_tau(v)[c][d] act0[c][v + d - 3],   H_tau[c][d] = sum over the taps t of E that tau allows, s = d - t, of F_t[c][s]
// (F_t = E[., t] composed with W1: k_dl_fold_fwd, gen_nets.hip) -- exact on faces, edges and corners, 64 x 343 products per voxel where the 5^3 layer
// alone took 64 x 64 x 125.  The interior kernel H_13 runs everywhere on the two-term matrix kernels of conv_s3x.hip (the pseudo-channel forms of
// Conv3d(1, 64, 7)); the 5.5 % of the voxels that lie on a face are recomputed / corrected here with their own type's kernel:
//     forward    y[v] (v on a face) = its own H_tau                                                            k_dl_bnd_fwd
//     dL/dact0  += sum_{v on a face} dy[v] (H_tau(v) - H_13)[c][u - v + 3]                                      k_dl_bnd_dgrad
//     dH_tau     = sum_{v of type tau} dy[v] act0[c][v + d - 3]  for the 26 boundary types                      k_dl_bnd_wgrad + k_dl_bnd_reduce
//     dH_13      = (the full correlation of dy and act0: conv_wgrad_c1 with the operands' roles swapped) - sum of the boundary types'
// and every parameter gradient follows in weight space: P[t][c][s] = sum_tau [tau allows t] dH_tau[c][t + s] is exactly the tensor the
// rank-structured form took from a 32 x 64-channel 5^3 weight gradient (gen_nets.hip: dW1 = E . P, q = W1 . P, then the tail's formulas).
// Boundary voxels: the full rows (z, y) with z or y on a face (lanes run along x), and the two end voxels of every other row -- the x faces,
// read through a transposed copy of the four columns next to each x face (k_dl_gather_x: lanes run along y).  Everything deterministic: no
// floating-point atomics, partial sums reduced in a fixed order.
#include "common.hpp"

namespace nc {
namespace {

constexpr int kT = 27, kD7 = 343, kC = 64;

__host__ __device__ inline int axis_type(int coord, int L) { return coord == 0 ? 0 : coord == L - 1 ? 2 : 1; }
// tap t_a (0 .. 2) of the 3^3 kernel along an axis whose voxel has type ta: allowed unless it points outside
__host__ __device__ inline bool tap_ok(int ta, int t) { return !((ta == 0 && t == 0) || (ta == 2 && t == 2)); }

// ---- weight space: the 27 typed kernels.  Hc[tau][d][c] (c fastest), Hd = Hc - Hc[13], Wp[c][t] = H_13[c][6 - t] (the weights the
// pseudo-channel kernels take: conv_c1k7_h2_dgrad as the forward, conv_c1k7_h2 as the data gradient).  F[t][c][s] from k_dl_fold_fwd.
// Hr[tau][dz][dy][c][8]: the seven dx taps of a channel side by side (the rows' forward: one 32-byte scalar load per channel), Hx[tau][dz][dx][c][8]: the
// seven dy taps (the x faces' forward)
__global__ void __launch_bounds__(64) k_dl_h_from_f(const float* __restrict__ F, float* __restrict__ Hc, float* __restrict__ Hd, float* __restrict__ Wp,
                                                    float* __restrict__ Hr, float* __restrict__ Hx) {
  const int tau = blockIdx.x, d = blockIdx.y, c = threadIdx.x;
  const int dz = d / 49, dy = (d / 7) % 7, dx = d % 7;
  const int tz_ = tau / 9, ty_ = (tau / 3) % 3, tx_ = tau % 3;
  double v = 0.0, v13 = 0.0;
  for (int tz = 0; tz < 3; ++tz) {
    const int sz = dz - tz;
    if (sz < 0 || sz > 4) continue;
    for (int ty = 0; ty < 3; ++ty) {
      const int sy = dy - ty;
      if (sy < 0 || sy > 4) continue;
      for (int tx = 0; tx < 3; ++tx) {
        const int sx = dx - tx;
        if (sx < 0 || sx > 4) continue;
        const double f = (double)F[((long)((tz * 3 + ty) * 3 + tx) * kC + c) * 125 + (sz * 5 + sy) * 5 + sx];
        v13 += f;
        if (tap_ok(tz_, tz) && tap_ok(ty_, ty) && tap_ok(tx_, tx)) v += f;
      }
    }
  }
  Hc[((long)tau * kD7 + d) * kC + c] = (float)v;
  Hd[((long)tau * kD7 + d) * kC + c] = (float)(v - v13);
  if (tau == 13) Wp[(long)c * kD7 + ((6 - dz) * 7 + (6 - dy)) * 7 + (6 - dx)] = (float)v;
  float* hr = Hr + ((((long)tau * 7 + dz) * 7 + dy) * kC + c) * 8;
  float* hx = Hx + ((((long)tau * 7 + dz) * 7 + dx) * kC + c) * 8;
  hr[dx] = (float)v;
  hx[dy] = (float)v;
  if (dx == 6) hr[7] = 0.f;
  if (dy == 6) hx[7] = 0.f;
}

// Pq[c][a][124 - s] (the layout k_dl_q_from_p / k_dl_w1_contract read; a = 27 .. 31 zero) = sum_tau [tau allows a] dH_tau[c][a + s]
//   = dHall[c][a + s] - sum_{tau != 13} [tau does NOT allow a] dHb[tau][a + s][c];   dHall[c][d] = dWsw[c][6 - d] (the swapped-role correlation)
__global__ void __launch_bounds__(128) k_dl_p_from_dh(const float* __restrict__ dWsw, const float* __restrict__ dHb, float* __restrict__ Pq) {
  const int c = blockIdx.x, a = blockIdx.y, sidx = threadIdx.x;
  if (sidx >= 125) return;
  double v = 0.0;
  if (a < 27) {
    const int az = a / 9, ay = (a / 3) % 3, ax = a % 3, sz = sidx / 25, sy = (sidx / 5) % 5, sx = sidx % 5;
    const int dz = az + sz, dy = ay + sy, dx = ax + sx, d = (dz * 7 + dy) * 7 + dx;
    v = (double)dWsw[(long)c * kD7 + ((6 - dz) * 7 + (6 - dy)) * 7 + (6 - dx)];
    for (int tau = 0; tau < kT; ++tau) {
      if (tau == 13) continue;
      if (tap_ok(tau / 9, az) && tap_ok((tau / 3) % 3, ay) && tap_ok(tau % 3, ax)) continue;
      v -= (double)dHb[((long)tau * kD7 + d) * kC + c];
    }
  }
  Pq[((long)c * 32 + a) * 125 + 124 - sidx] = (float)v;
}

// ---- the transposed copies for the x faces: AX[n][side][c][z][xr][y] = act0[n][c][z][y][xb(side) + xr], xb = 0 / W - 4, xr = 0 .. 3 (y fastest);
//      dyX[n][side][z][y] = dy[n][z][y][0 / W - 1]
__global__ void __launch_bounds__(128) k_dl_gather_x(const float* __restrict__ a, float* __restrict__ ax, const float* __restrict__ dy, float* __restrict__ dyx,
                                                     int D, int H, int W) {
  const int z = blockIdx.x, c = blockIdx.y, ns = blockIdx.z, n = ns >> 1, side = ns & 1;
  const int xb = side ? W - 4 : 0;
  const long HW = (long)H * W;
  if (a) {
    const float* src = a + (((long)n * kC + c) * D + z) * HW;
    float* dst = ax + ((((long)n * 2 + side) * kC + c) * D + z) * 4 * H;
    for (int y = threadIdx.x; y < H; y += 128)
#pragma unroll
      for (int xr = 0; xr < 4; ++xr) dst[(long)xr * H + y] = src[(long)y * W + xb + xr];
  }
  if (dy && c == 0) {
    const float* src = dy + ((long)n * D + z) * HW;
    float* dst = dyx + (((long)n * 2 + side) * D + z) * H;
    for (int y = threadIdx.x; y < H; y += 128) dst[y] = src[(long)y * W + (side ? W - 1 : 0)];
  }
}

// Boundary voxels in two sets: R = the full boundary rows (z or y on a face: numbered 0 .. 2 H + 2 (D - 2) - 1 -- plane 0, plane D - 1, then rows 0
// and H - 1 of the planes between) without their two end voxels: ONE type (tz, ty, 1) per row, lanes run along x; X = the voxels with x = 0 or W - 1
// of EVERY row (the x faces with their edges and corners): type (tz, ty, tx) per voxel, lanes run along y through the transposed copies AX / dyX.
__device__ inline void row_decode(int r, int D, int H, int& z, int& y) {
  if (r < H) { z = 0; y = r; }
  else if (r < 2 * H) { z = D - 1; y = r - H; }
  else { const int k = r - 2 * H; z = 1 + (k >> 1); y = (k & 1) ? H - 1 : 0; }
}
__host__ inline int n_bnd_rows(int D, int H) { return 2 * H + 2 * (D - 2); }

// ---- forward: y[v] for the boundary voxels.  Seven waves = the seven kernel planes dz; a wave = 58 voxels of a line plus three halo lanes on either
// side: ONE coalesced load per (channel, line of the 7 x 7 window) and lane, the seven shifts along the line by DPP (wave_shr / wave_shl), weights
// wave-uniform (scalar loads).  XF = false: the lines are rows, block = (row of R, 58-voxel segment of x = 1 .. W - 2, sample); XF = true: the lines
// run along y through AX, block = (plane z, side, segment of y = 1 .. H - 2); the voxels at the two ends of a y line (the volume's four vertical
// edges, their own types) are k_dl_bnd_fwd_ends'.
__device__ __forceinline__ float dpp_prev(float v) {  // lane i <- lane i - 1 (lane 0: 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_next(float v) {  // lane i <- lane i + 1 (lane 63: 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
constexpr int kSeg = 58;
template <bool XF>
__global__ void __launch_bounds__(448) k_dl_bnd_fwd(const float* __restrict__ src, const float* __restrict__ Hl, float* __restrict__ y, int D, int H, int W) {
  __shared__ float red[7][64];
  const int lane = threadIdx.x & 63, dz = threadIdx.x >> 6, n = blockIdx.z;
  const long HW = (long)H * W, S = (long)D * HW;
  // geometry of this block's line: L = its length, p = this lane's position on it (the wave covers p0 - 3 .. p0 + 60), first / last voxel of the set
  int z, side = 0, seg, L, lo, hi, fix;  // fix: the other in-plane coordinate (y for rows, -) ...
  if constexpr (!XF) { row_decode(blockIdx.x, D, H, z, fix); seg = blockIdx.y; L = W; lo = 1; hi = W - 2; }
  else { z = blockIdx.x; side = blockIdx.y & 1; seg = blockIdx.y >> 1; L = H; lo = 1; hi = H - 2; fix = 0; }
  const int p = lo + seg * kSeg + lane - 3;
  const bool mine = lane >= 3 && lane < 3 + kSeg && p <= hi;  // this lane owns an output voxel
  const bool inl = (unsigned)p < (unsigned)L;                  // ... or at least a position on the line (halo)
  const int zz = z + dz - 3;
  const long cs = XF ? (long)D * 4 * H : S;  // channel stride of the source
  // one pass with a wave-uniform type; `act`: the lanes whose sum counts
  auto pass = [&](int tau) __attribute__((always_inline)) {
    float a0 = 0.f, a1 = 0.f;
    if ((unsigned)zz >= (unsigned)D) return 0.f;
    for (int k = 0; k < (XF ? 4 : 7); ++k) {  // the other window axis: dy (rows) / the four columns next to the face (x faces)
      const float* line;
      int dq;  // the tap index along that axis
      if constexpr (!XF) {
        const int y2 = fix + k - 3;
        if ((unsigned)y2 >= (unsigned)H) continue;
        line = src + (long)n * kC * S + (long)zz * HW + (long)y2 * W;
        dq = k;
      } else {
        line = src + ((long)n * 2 + side) * kC * cs + ((long)zz * 4 + k) * H;
        dq = side ? k : k + 3;  // side 0: x' = dx - 3 = k; side 1: x' = W - 4 + dx, k = dx
      }
      // the 7 taps along the line of channel c: 8 consecutive floats (Hr: (dz, dy = dq) x dx; Hx: (dz, dx = dq) x dy), wave-uniform
      const float4* h = reinterpret_cast<const float4*>(Hl + (((long)tau * 7 + dz) * 7 + dq) * kC * 8);
      // sixteen channels per trip: their loads are in flight together (the kernel is latency-bound, not instruction-bound: ~6 waves per SIMD)
      for (int c0 = 0; c0 < kC; c0 += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = inl ? line[(long)(c0 + i) * cs + p] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float m1 = dpp_prev(v[i]), m2 = dpp_prev(m1), m3 = dpp_prev(m2);
          const float p1 = dpp_next(v[i]), p2 = dpp_next(p1), p3 = dpp_next(p2);
          const float4 w0 = h[2 * (c0 + i)], w1 = h[2 * (c0 + i) + 1];
          float& a = (i & 1) ? a1 : a0;
          float& b = (i & 1) ? a0 : a1;
          a = __builtin_fmaf(w0.x, m3, a);
          b = __builtin_fmaf(w0.y, m2, b);
          a = __builtin_fmaf(w0.z, m1, a);
          b = __builtin_fmaf(w0.w, v[i], b);
          a = __builtin_fmaf(w1.x, p1, a);
          b = __builtin_fmaf(w1.y, p2, b);
          a = __builtin_fmaf(w1.z, p3, a);
        }
      }
    }
    return a0 + a1;
  };
  float acc;
  if constexpr (!XF) acc = pass((axis_type(z, D) * 3 + axis_type(fix, H)) * 3 + 1);
  else acc = pass(axis_type(z, D) * 9 + 3 + (side ? 2 : 0));  // (the two ends of the y line -- the volume's vertical edges -- are k_dl_bnd_fwd_ends')
  red[dz][lane] = acc;
  __syncthreads();
  if (dz == 0 && mine) {
    float v = red[0][lane];
#pragma unroll
    for (int k = 1; k < 7; ++k) v += red[k][lane];
    const long o = XF ? (long)z * HW + (long)p * W + (side ? W - 1 : 0) : (long)z * HW + (long)fix * W + p;
    y[(long)n * S + o] = v;
  }
}

// the 4 D voxels of the volume's vertical edges (x = 0 / W - 1 and y = 0 / H - 1): block = (z, side * 2 + yend, sample); threads = 64 channels x 4
// groups of kernel planes; fixed-order reduction through LDS
__global__ void __launch_bounds__(256) k_dl_bnd_fwd_ends(const float* __restrict__ ax, const float* __restrict__ Hc, float* __restrict__ y, int D, int H, int W) {
  __shared__ float red[256];
  const int th = threadIdx.x, c = th & 63, part = th >> 6;
  const int z = blockIdx.x, side = blockIdx.y >> 1, ye = blockIdx.y & 1, n = blockIdx.z;
  const int yy = ye ? H - 1 : 0;
  const int tau = (axis_type(z, D) * 3 + (ye ? 2 : 0)) * 3 + (side ? 2 : 0);
  const long cs = (long)D * 4 * H;
  const float* a = ax + (((long)n * 2 + side) * kC + c) * cs;
  float acc = 0.f;
  for (int dz = part; dz < 7; dz += 4) {
    const int zz = z + dz - 3;
    if ((unsigned)zz >= (unsigned)D) continue;
    for (int dxr = 0; dxr < 4; ++dxr) {
      const int dx = side ? dxr : dxr + 3;
      for (int dy = 0; dy < 7; ++dy) {
        const int y2 = yy + dy - 3;
        if ((unsigned)y2 >= (unsigned)H) continue;
        acc = __builtin_fmaf(Hc[((long)tau * kD7 + (dz * 7 + dy) * 7 + dx) * kC + c], a[((long)zz * 4 + dxr) * H + y2], acc);
      }
    }
  }
  red[th] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (th < o) red[th] += red[th + o];
    __syncthreads();
  }
  if (th == 0) y[(long)n * D * H * W + (long)z * H * W + (long)yy * W + (side ? W - 1 : 0)] = red[0];
}

// ---- dL/dact0 += the boundary voxels' difference kernels (Hd = Hc - Hc[13]).  One thread owns (u, 16 channels): plain read-modify-write, the two
// launches run one after the other.  XF = false: sources = the R set; block = (output row (uz, uy), segment of x, sample), rows out of reach leave at
// once; the source type is wave-uniform.  XF = true: sources = the X set through dyX; block = (uz, side, segment of y): a thread owns the FOUR
// columns next to the face (one 16-byte read-modify-write per channel); the sources at the ends of the y line (their own types) in a second pass.
template <bool XF>
__global__ void __launch_bounds__(256) k_dl_bnd_dgrad(const float* __restrict__ dysrc, const float* __restrict__ Hd, float* __restrict__ g, int D, int H, int W,
                                                      int nseg) {
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long HW = (long)H * W, S = (long)D * HW;
  if constexpr (!XF) {
    const int n = blockIdx.z;
    const int uz = blockIdx.x / H, uy = blockIdx.x - uz * H;
    if (!(uz <= 3 || uz >= D - 4 || uy <= 3 || uy >= H - 4)) return;
    const int ux = blockIdx.y * 64 + lane;
    if (ux >= W) return;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int dz = 0; dz < 7; ++dz) {
      const int vz = uz - dz + 3;
      if ((unsigned)vz >= (unsigned)D) continue;
      const int tz = axis_type(vz, D);
      for (int dy = 0; dy < 7; ++dy) {
        const int vy = uy - dy + 3;
        if ((unsigned)vy >= (unsigned)H) continue;
        const int ty = axis_type(vy, H);
        if (tz == 1 && ty == 1) continue;  // not a row of R
        const int tau = (tz * 3 + ty) * 3 + 1;
        const float* drow = dysrc + (long)n * S + (long)vz * HW + (long)vy * W;
        for (int dx = 0; dx < 7; ++dx) {
          const int vx = ux - dx + 3;
          const float w = (vx >= 1 && vx <= W - 2) ? drow[vx] : 0.f;
          const float* h = Hd + ((long)tau * kD7 + (dz * 7 + dy) * 7 + dx) * kC + q * 16;  // wave-uniform
#pragma unroll
          for (int c = 0; c < 16; ++c) acc[c] = __builtin_fmaf(w, h[c], acc[c]);
        }
      }
    }
    float* o = g + ((long)n * kC + q * 16) * S + (long)uz * HW + (long)uy * W + ux;
#pragma unroll
    for (int c = 0; c < 16; ++c) o[(long)c * S] += acc[c];
  } else {
    const int uz = blockIdx.x, side = blockIdx.y;
    const int n = blockIdx.z / nseg, seg = blockIdx.z - n * nseg;
    const int uy = seg * 64 + lane;
    const bool valid = uy < H;
    const int tx = side ? 2 : 0;
    float acc[4][16];
#pragma unroll
    for (int xr = 0; xr < 4; ++xr)
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[xr][c] = 0.f;
    // pass ty: the sources (vz, vy) whose y type is ty.  ty = 1: every lane (sources at the ends of the line are left out); ty = 0 / 2: the source
    // row vy = 0 / H - 1, wanted by the lanes within three of it
    auto pass = [&](int ty) __attribute__((always_inline)) {
      for (int dz = 0; dz < 7; ++dz) {
        const int vz = uz - dz + 3;
        if ((unsigned)vz >= (unsigned)D) continue;
        const int tau = (axis_type(vz, D) * 3 + ty) * 3 + tx;
        if (tau == 13) continue;
        const float* dpl = dysrc + (((long)n * 2 + side) * D + vz) * H;
        for (int dy = 0; dy < 7; ++dy) {
          const int vy = uy - dy + 3;
          const bool in = valid && (unsigned)vy < (unsigned)H && axis_type((unsigned)vy < (unsigned)H ? vy : 1, H) == ty;
          const float w = in ? dpl[vy] : 0.f;
#pragma unroll
          for (int xr = 0; xr < 4; ++xr) {
            const int dx = side ? xr : xr + 3;  // u_x = x0 + dx - 3: the column xr of the four
            const float* h = Hd + ((long)tau * kD7 + (dz * 7 + dy) * 7 + dx) * kC + q * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[xr][c] = __builtin_fmaf(w, h[c], acc[xr][c]);
          }
        }
      }
    };
    pass(1);  // (the sources at the two ends of the y line -- the vertical edges -- are k_dl_bnd_dgrad_ends')
    if (valid) {
      float4* o = reinterpret_cast<float4*>(g + ((long)n * kC + q * 16) * S + (long)uz * HW + (long)uy * W + (side ? W - 4 : 0));
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        float4 v = o[(long)c * (S / 4)];
        v.x += acc[0][c]; v.y += acc[1][c]; v.z += acc[2][c]; v.w += acc[3][c];
        o[(long)c * (S / 4)] = v;
      }
    }
  }
}

// the sources on the volume's vertical edges (x = 0 / W - 1 and y = 0 / H - 1) through dyX: block = (uz, side * 2 + yend, sample): 4 rows uy x 4 columns x
// 64 channels of outputs, four per thread
__global__ void __launch_bounds__(256) k_dl_bnd_dgrad_ends(const float* __restrict__ dyx, const float* __restrict__ Hd, float* __restrict__ g, int D, int H, int W) {
  const int th = threadIdx.x, c = th & 63, r = th >> 6;  // r: which of the four output rows
  const int uz = blockIdx.x, side = blockIdx.y >> 1, ye = blockIdx.y & 1, n = blockIdx.z;
  const int vy = ye ? H - 1 : 0, uy = ye ? H - 4 + r : r, dy = uy - vy + 3;
  const long HW = (long)H * W, S = (long)D * HW;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int dz = 0; dz < 7; ++dz) {
    const int vz = uz - dz + 3;
    if ((unsigned)vz >= (unsigned)D) continue;
    const int tau = (axis_type(vz, D) * 3 + (ye ? 2 : 0)) * 3 + (side ? 2 : 0);
    const float w = dyx[(((long)n * 2 + side) * D + vz) * H + vy];
#pragma unroll
    for (int xr = 0; xr < 4; ++xr) {
      const int dx = side ? xr : xr + 3;
      acc[xr] = __builtin_fmaf(w, Hd[((long)tau * kD7 + (dz * 7 + dy) * 7 + dx) * kC + c], acc[xr]);
    }
  }
  float* o = g + ((long)n * kC + c) * S + (long)uz * HW + (long)uy * W + (side ? W - 4 : 0);
#pragma unroll
  for (int xr = 0; xr < 4; ++xr) o[xr] += acc[xr];
}

// ---- the boundary types' dH.  Threads (k, c), k = dx (R set) / dy (X set), c = channel; per staged line of act0 (LDS, 64 channels) every thread
// correlates it with the source line's dy.  XF = false: block = (row r of R, dz, sample); part[n][r][d][c].  XF = true: block = (plane vz, side, dz,
// sample): three sums -- the source at y = 0, the ones between, the source at y = H - 1 (their own types): partB[slot][n][side][vz][d][c].
constexpr int kPitch = 193;  // LDS line pitch in floats (W, H <= 192; odd: the 64 channels fall on different banks)
template <bool XF>
__global__ void __launch_bounds__(448) k_dl_bnd_wgrad(const float* __restrict__ asrc, const float* __restrict__ dysrc, float* __restrict__ part, int D, int H, int W,
                                                      int nrows) {
  extern __shared__ float lds[];  // [64][kPitch] + [192]
  float* A = lds;
  float* dyr = lds + 64 * kPitch;
  const int th = threadIdx.x, c = th & 63, k = th >> 6;
  const int dz = blockIdx.y, n = blockIdx.z;
  const long HW = (long)H * W, S = (long)D * HW;
  const int sh = k - 3;
  if constexpr (!XF) {
    const int r = blockIdx.x;
    int vz, vy;
    row_decode(r, D, H, vz, vy);
    for (int i = th; i < W; i += 448) dyr[i] = dysrc[(long)n * S + (long)vz * HW + (long)vy * W + i];
    const int zz = vz + dz - 3;
    for (int dy = 0; dy < 7; ++dy) {
      const int y2 = vy + dy - 3;
      const bool in = (unsigned)zz < (unsigned)D && (unsigned)y2 < (unsigned)H;
      __syncthreads();
      if (in) {
        const float* base = asrc + (long)n * kC * S + (long)zz * HW + (long)y2 * W;
        for (int cc = th >> 6; cc < kC; cc += 7)
          for (int xx = th & 63; xx < W; xx += 64) A[cc * kPitch + xx] = base[(long)cc * S + xx];
      }
      __syncthreads();
      float mid = 0.f;
      if (in) {
        const float* a = A + c * kPitch;
        const int x0 = sh < 0 ? (1 > -sh ? 1 : -sh) : 1, x1 = sh > 0 ? (W - 2 < W - 1 - sh ? W - 2 : W - 1 - sh) : W - 2;  // 1 <= x <= W - 2, 0 <= x + sh < W
        float m0 = 0.f, m1 = 0.f;
        int x = x0;
        for (; x + 1 <= x1; x += 2) { m0 = __builtin_fmaf(dyr[x], a[x + sh], m0); m1 = __builtin_fmaf(dyr[x + 1], a[x + 1 + sh], m1); }
        if (x <= x1) m0 = __builtin_fmaf(dyr[x], a[x + sh], m0);
        mid = m0 + m1;
      }
      part[(((long)n * nrows + r) * kD7 + (dz * 7 + dy) * 7 + k) * kC + c] = mid;
    }
  } else {
    const int vz = blockIdx.x >> 1, side = blockIdx.x & 1;
    for (int i = th; i < H; i += 448) dyr[i] = dysrc[(((long)n * 2 + side) * D + vz) * H + i];
    const long cs = (long)D * 4 * H;
    const long slot = (long)gridDim.z * 2 * D * kD7 * kC;
    const long pb = (((long)n * 2 + side) * D + vz) * kD7 * kC + c;
    const int zz = vz + dz - 3;
    // (the three dx this side never reads: zero)
    for (int dxz = 0; dxz < 3; ++dxz) {
      const long o = pb + (long)((dz * 7 + k) * 7 + (side ? 4 + dxz : dxz)) * kC;
      part[o] = 0.f; part[slot + o] = 0.f; part[2 * slot + o] = 0.f;
    }
    for (int xr = 0; xr < 4; ++xr) {
      const int dx = side ? xr : xr + 3;
      const bool in = (unsigned)zz < (unsigned)D;
      __syncthreads();
      if (in) {
        const float* base = asrc + ((long)n * 2 + side) * kC * cs + ((long)zz * 4 + xr) * H;
        for (int cc = th >> 6; cc < kC; cc += 7)
          for (int yy = th & 63; yy < H; yy += 64) A[cc * kPitch + yy] = base[(long)cc * cs + yy];
      }
      __syncthreads();
      float mid = 0.f, e0 = 0.f, e1 = 0.f;
      if (in) {
        const float* a = A + c * kPitch;
        const int y0 = sh < 0 ? (1 > -sh ? 1 : -sh) : 1, y1 = sh > 0 ? (H - 2 < H - 1 - sh ? H - 2 : H - 1 - sh) : H - 2;
        float m0 = 0.f, m1 = 0.f;
        int y = y0;
        for (; y + 1 <= y1; y += 2) { m0 = __builtin_fmaf(dyr[y], a[y + sh], m0); m1 = __builtin_fmaf(dyr[y + 1], a[y + 1 + sh], m1); }
        if (y <= y1) m0 = __builtin_fmaf(dyr[y], a[y + sh], m0);
        mid = m0 + m1;
        if (sh >= 0 && sh < H) e0 = dyr[0] * a[sh];
        if (H - 1 + sh >= 0 && sh <= 0) e1 = dyr[H - 1] * a[H - 1 + sh];
      }
      const long o = pb + (long)((dz * 7 + k) * 7 + dx) * kC;
      part[o] = e0; part[slot + o] = mid; part[2 * slot + o] = e1;
    }
  }
}

// dHb[tau][d][c] = the partial sums of the rows of R / the planes and slots of X whose type is tau, in a fixed order (samples, then ascending)
__global__ void __launch_bounds__(256) k_dl_bnd_reduce(const float* __restrict__ partA, const float* __restrict__ partB, float* __restrict__ dHb, int N, int D, int H, int W,
                                                       int nrows) {
  const int tau = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // d * 64 + c
  if (i >= (long)kD7 * kC) return;
  const int tz = tau / 9, ty = (tau / 3) % 3, tx = tau % 3;
  double v = 0.0;
  if (tau != 13) {
    for (int n = 0; n < N; ++n) {
      if (tx == 1) {  // rows of R with (tz, ty): a progression of row numbers
        int r0, cnt, st;
        if (tz != 1) { const int b = tz == 0 ? 0 : H; r0 = b + (ty == 0 ? 0 : ty == 2 ? H - 1 : 1); cnt = ty == 1 ? H - 2 : 1; st = 1; }
        else { r0 = 2 * H + (ty == 2 ? 1 : 0); cnt = D - 2; st = 2; }  // (ty == 1 && tz == 1 is the interior: not here)
        for (int j = 0; j < cnt; ++j) v += (double)partA[((long)n * nrows + r0 + (long)j * st) * kD7 * kC + i];
      } else {  // planes of X with tz, slot ty, side tx / 2
        const int side = tx >> 1;
        const long slot = (long)N * 2 * D * kD7 * kC;
        const int z0 = tz == 0 ? 0 : tz == 2 ? D - 1 : 1, cnt = tz == 1 ? D - 2 : 1;
        for (int j = 0; j < cnt; ++j) v += (double)partB[(long)ty * slot + (((long)n * 2 + side) * D + z0 + j) * kD7 * kC + i];
      }
    }
  }
  dHb[(long)tau * kD7 * kC + i] = (float)v;
}

size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// ---- host side.  Scratch layout (dl_typed_bytes): Hc | Hd | Wp | dWsw | dHb | AX | dyX | partA | partB
struct DlTyped {
  size_t Hc, Hd, Wp, dWsw, dHb, AX, dyX, partA, partB, Hr, Hx, total;
  int nrows;
};
static DlTyped dl_typed_plan(int N, int D, int H, int W) {
  DlTyped t{};
  size_t off = 0;
  auto take = [&](size_t b) { const size_t r = off; off += al256(b); return r; };
  t.nrows = n_bnd_rows(D, H);
  t.Hc = take((size_t)kT * kD7 * kC * 4); t.Hd = take((size_t)kT * kD7 * kC * 4); t.Wp = take((size_t)kC * kD7 * 4); t.dWsw = take((size_t)kC * kD7 * 4);
  t.dHb = take((size_t)kT * kD7 * kC * 4);
  t.AX = take((size_t)N * 2 * kC * D * 4 * H * 4); t.dyX = take((size_t)N * 2 * D * H * 4);
  t.Hr = take((size_t)kT * 49 * kC * 8 * 4); t.Hx = take((size_t)kT * 49 * kC * 8 * 4);
  t.partA = take((size_t)N * t.nrows * kD7 * kC * 4); t.partB = take((size_t)3 * N * 2 * D * kD7 * kC * 4);
  t.total = off;
  return t;
}
bool dl_typed_supported(int N, int D, int H, int W) {
  return N >= 1 && D >= 8 && H >= 8 && W >= 8 && H <= 192 && W <= 192 && W % 4 == 0;
}
size_t dl_typed_bytes(int N, int D, int H, int W) { return dl_typed_supported(N, D, H, W) ? dl_typed_plan(N, D, H, W).total : 0; }
const float* dl_typed_wp(const char* scratch, int N, int D, int H, int W) { return (const float*)(scratch + dl_typed_plan(N, D, H, W).Wp); }
float* dl_typed_dwsw(char* scratch, int N, int D, int H, int W) { return (float*)(scratch + dl_typed_plan(N, D, H, W).dWsw); }

// the typed kernels from F (k_dl_fold_fwd's 27 composed 5^3 kernels)
int dl_typed_compose(const float* F, char* scratch, int N, int D, int H, int W, hipStream_t s) {
  const DlTyped t = dl_typed_plan(N, D, H, W);
  hipLaunchKernelGGL(k_dl_h_from_f, dim3(kT, kD7), dim3(64), 0, s, F, (float*)(scratch + t.Hc), (float*)(scratch + t.Hd), (float*)(scratch + t.Wp), (float*)(scratch + t.Hr),
                     (float*)(scratch + t.Hx));
  return check_launch("deep_linear: typed kernels");
}
// y's boundary voxels, recomputed with their own kernels (y holds the interior kernel's result everywhere)
int dl_typed_fwd_boundary(const float* act0, float* y, char* scratch, int N, int D, int H, int W, hipStream_t s) {
  const DlTyped t = dl_typed_plan(N, D, H, W);
  float* AX = (float*)(scratch + t.AX);
  hipLaunchKernelGGL(k_dl_gather_x, dim3((unsigned)D, kC, (unsigned)(N * 2)), dim3(128), 0, s, act0, AX, (const float*)nullptr, (float*)nullptr, D, H, W);
  hipLaunchKernelGGL(k_dl_bnd_fwd<false>, dim3((unsigned)t.nrows, (unsigned)cdiv(W - 2, kSeg), (unsigned)N), dim3(448), 0, s, act0, (const float*)(scratch + t.Hr), y, D, H, W);
  hipLaunchKernelGGL(k_dl_bnd_fwd<true>, dim3((unsigned)D, (unsigned)(2 * cdiv(H - 2, kSeg)), (unsigned)N), dim3(448), 0, s, (const float*)AX, (const float*)(scratch + t.Hx), y,
                     D, H, W);
  hipLaunchKernelGGL(k_dl_bnd_fwd_ends, dim3((unsigned)D, 4, (unsigned)N), dim3(256), 0, s, (const float*)AX, (const float*)(scratch + t.Hc), y, D, H, W);
  return check_launch("deep_linear: typed forward, boundary");
}
// the backward's boundary pieces: g (= dL/dact0 from the interior kernel) += the boundary voxels' difference kernels, and Pq (the rank form's P,
// k_dl_q_from_p's layout) from dWsw (the swapped-role 7^3 correlation of dy and act0, already in the scratch) and the boundary types' sums
int dl_typed_dgrad_boundary(const float* act0, const float* dy, float* g, char* scratch, int N, int D, int H, int W, hipStream_t s) {
  const DlTyped t = dl_typed_plan(N, D, H, W);
  float* AX = (float*)(scratch + t.AX);
  float* dyX = (float*)(scratch + t.dyX);
  hipLaunchKernelGGL(k_dl_gather_x, dim3((unsigned)D, kC, (unsigned)(N * 2)), dim3(128), 0, s, act0, AX, dy, dyX, D, H, W);  // (AX: for dl_typed_p below)
  hipLaunchKernelGGL(k_dl_bnd_dgrad<false>, dim3((unsigned)(D * H), (unsigned)cdiv(W, 64), (unsigned)N), dim3(256), 0, s, dy, (const float*)(scratch + t.Hd), g, D, H, W, 1);
  const int nseg = (int)cdiv(H, 64);
  hipLaunchKernelGGL(k_dl_bnd_dgrad<true>, dim3((unsigned)D, 2, (unsigned)(N * nseg)), dim3(256), 0, s, (const float*)dyX, (const float*)(scratch + t.Hd), g, D, H, W, nseg);
  hipLaunchKernelGGL(k_dl_bnd_dgrad_ends, dim3((unsigned)D, 4, (unsigned)N), dim3(256), 0, s, (const float*)dyX, (const float*)(scratch + t.Hd), g, D, H, W);
  return check_launch("deep_linear: typed data gradient, boundary");
}
// Pq (the rank form's P, k_dl_q_from_p's layout) from dWsw (the swapped-role 7^3 correlation of dy and act0, already in the scratch) and the boundary
// types' sums (computed here: needs act0 and dy)
int dl_typed_p(const float* act0, const float* dy, float* Pq, char* scratch, int N, int D, int H, int W, hipStream_t s) {  // (after dl_typed_dgrad_boundary: AX, dyX)
  const DlTyped t = dl_typed_plan(N, D, H, W);
  const float* AX = (const float*)(scratch + t.AX);
  const float* dyX = (const float*)(scratch + t.dyX);
  const size_t lds = (size_t)(64 * kPitch + 192) * 4;
  hipLaunchKernelGGL(k_dl_bnd_wgrad<false>, dim3((unsigned)t.nrows, 7, (unsigned)N), dim3(448), lds, s, act0, dy, (float*)(scratch + t.partA), D, H, W, t.nrows);
  hipLaunchKernelGGL(k_dl_bnd_wgrad<true>, dim3((unsigned)(2 * D), 7, (unsigned)N), dim3(448), lds, s, AX, dyX, (float*)(scratch + t.partB), D, H, W, t.nrows);
  hipLaunchKernelGGL(k_dl_bnd_reduce, dim3((unsigned)cdiv((long)kD7 * kC, 256), kT), dim3(256), 0, s, (const float*)(scratch + t.partA), (const float*)(scratch + t.partB),
                     (float*)(scratch + t.dHb), N, D, H, W, t.nrows);
  hipLaunchKernelGGL(k_dl_p_from_dh, dim3(kC, 32), dim3(128), 0, s, (const float*)(scratch + t.dWsw), (const float*)(scratch + t.dHb), Pq);
  return check_launch("deep_linear: typed parameter gradients");
}

}  // namespace nc
