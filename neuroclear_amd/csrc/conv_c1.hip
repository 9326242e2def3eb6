// Convolution onto ONE output channel from many input channels with a large cubic kernel (7^3, stride 1, same padding):
// the data gradient of deep_linear_gen's first layer (reference models/networks.py:899: Conv3d(1, 64, 7, padding=3),
// dgrad = correlation of the 64-channel dY with the flipped kernels).  M = 1 makes the matrix cores useless (a 32-row
// MFMA would be 97 % padding), so this one runs on the VALU at its own roofline: each lane owns 2 (z) x 8 (x) outputs,
// weights are wave-uniform scalars (s_load -> SGPR operand of v_fmac), inputs come from an LDS brick read as aligned
// 16-byte vectors; every 14-float row read feeds 2 x 56 FMAs.
#include "common.hpp"

namespace nc {
NC_ZERO_PAGE()


typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS>
__global__ __launch_bounds__(256) void k_conv_to1(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ y, int C, int D,
                                                  int H, int W, int flip) {
  constexpr int PAD = KS / 2, TZ = 8, TY = 8, TX = 64, XB = 8;
  constexpr int PZ = TZ + KS - 1, PY = TY + KS - 1, PXU = TX + KS - 1;
  constexpr int PITCH = 76;  // >= PXU, multiple of 4, == 12 (mod 64): conflict-free ds_read_b128 over two rows
  constexpr int TAPS = KS * KS * KS;
  static_assert(PXU <= PITCH, "pitch");
  __shared__ __attribute__((aligned(16))) float lds[PZ * PY * PITCH];
  // this channel's kernel in the order the loop uses it, one 8-float (32-byte) row per (dz, dy): broadcast
  // ds_read_b128 instead of KS dependent scalar loads per row (the scalar-load latency was exposed)
  __shared__ __attribute__((aligned(16))) float wl[KS * KS * 8];
  const int tid = threadIdx.x;
  const int xb = tid & 7, ty = (tid >> 3) & 7, tzp = tid >> 6;
  const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
  const int bx = blockIdx.x % ntx, by = (blockIdx.x / ntx) % nty, bz = blockIdx.x / (ntx * nty);
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ, n = blockIdx.y;
  const long HW = (long)H * W, S = (long)D * HW;
  const float* xn = x + (long)n * C * S;

  float acc[2][XB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < XB; ++i) acc[a][i] = 0.f;

  for (int c = 0; c < C; ++c) {
    const float* xc = xn + (long)c * S;
    __syncthreads();
    for (int e = tid; e < PZ * PY * PXU; e += 256) {
      const int xx = e % PXU, yy = (e / PXU) % PY, pz = e / (PXU * PY);
      const int gz = z0 + pz - PAD, gy = y0 + yy - PAD, gx = x0 + xx - PAD;
      float v = 0.f;
      if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = xc[(long)gz * HW + (long)gy * W + gx];
      lds[(pz * PY + yy) * PITCH + xx] = v;
    }
    const float* wc = w + (long)c * TAPS;
    for (int e = tid; e < TAPS; e += 256) wl[(e / KS) * 8 + e % KS] = wc[flip ? TAPS - 1 - e : e];
    __syncthreads();
#pragma unroll 1
    for (int pz = 0; pz <= KS; ++pz) {
      const float* plane = lds + ((2 * tzp + pz) * PY + ty) * PITCH + xb * XB;
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        float row[16];  // 4 aligned float4 (XB + KS - 1 used: 14 at 7^3, 10 at 3^3)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          const float4 t = *reinterpret_cast<const float4*>(plane + dy * PITCH + 4 * v4);
          row[4 * v4] = t.x; row[4 * v4 + 1] = t.y; row[4 * v4 + 2] = t.z; row[4 * v4 + 3] = t.w;
        }
        if (pz < KS) {  // output plane z0 + 2 tzp uses tap dz = pz
          const float4 wa = *reinterpret_cast<const float4*>(wl + (pz * KS + dy) * 8);
          const float4 wb = *reinterpret_cast<const float4*>(wl + (pz * KS + dy) * 8 + 4);
          const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int dx = 0; dx < KS; ++dx)
#pragma unroll
            for (int i = 0; i < XB; ++i) acc[0][i] = fmaf(wv[dx], row[i + dx], acc[0][i]);
        }
        if (pz >= 1) {  // output plane z0 + 2 tzp + 1 uses tap dz = pz - 1
          const float4 wa = *reinterpret_cast<const float4*>(wl + ((pz - 1) * KS + dy) * 8);
          const float4 wb = *reinterpret_cast<const float4*>(wl + ((pz - 1) * KS + dy) * 8 + 4);
          const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int dx = 0; dx < KS; ++dx)
#pragma unroll
            for (int i = 0; i < XB; ++i) acc[1][i] = fmaf(wv[dx], row[i + dx], acc[1][i]);
        }
      }
    }
  }
  const float b0 = bias ? bias[0] : 0.f;
  const int gy = y0 + ty;
  if (gy < H) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int gz = z0 + 2 * tzp + a;
      if (gz < D) {
        float* yr = y + (long)n * S + (long)gz * HW + (long)gy * W;
#pragma unroll
        for (int i = 0; i < XB; ++i) {
          const int gx = x0 + xb * XB + i;
          if (gx < W) yr[gx] = acc[a][i] + b0;
        }
      }
    }
  }
}

// dgrad of a Conv3d with Cin == 1: dx[n][0][q] = sum_k sum_t w[k][0][t] * dy[n][k][q + p - t]
bool to1_dgrad_supported(const ConvDims& d) {
  return d.C == 1 && d.K >= 8 && d.kd == 7 && d.kh == 7 && d.kw == 7 && d.sd == 1 && d.sh == 1 && d.sw == 1 &&
         d.pd == 3 && d.ph == 3 && d.pw == 3 && d.N <= 65535;
}

int conv_dgrad_to1(const float* dy, const float* w, float* dx, const ConvDims& d, hipStream_t s) {
  const int ntx = (d.W + 63) / 64, nty = (d.H + 7) / 8, ntz = (d.D + 7) / 8;
  dim3 grid((unsigned)(ntx * nty * ntz), d.N);
  hipLaunchKernelGGL(k_conv_to1<7>, grid, dim3(256), 0, s, dy, w, (const float*)nullptr, dx, d.K, d.D, d.H, d.W, 1);
  return check_launch("conv_dgrad_to1");
}

// Many channels -> ONE channel, 3^3, stride 1, padding 1, no bias, y[n][0][q] = sum_c sum_t w[c][t] * x[n][c][q + t - 1]: the collapsed tail
// of deep_linear_gen (gen_nets.hip: the 3^3 layer and the three 1 x 1 layers behind it are ONE such convolution).  2 x 27 FLOP per input
// element: HBM-bound by a factor of ~2 if the VALU is kept fed, so the kernel is built around reading x once:
//   * a workgroup owns 8 rows x <= 128 columns x a range of planes of the output and MARCHES over the input planes: an input plane feeds the
//     three output planes around it, held in registers (4 columns x 3 planes per thread); the plane that just received its last addend is
//     written -- x is read once per (8-row, plane-range) tile plus a one-row / one-plane halo;
//   * a stage = 4 channels of one plane's 10 x 136 floats, loaded as aligned 16-byte vectors into registers while the previous stage is
//     multiplied out of LDS (two buffers, one barrier per stage); per channel a thread reads 3 rows x (b128 + 2 x b32) and issues 108 FMAs;
//     weights are wave-uniform scalar loads.
constexpr int kT1TY = 8, kT1TX = 128, kT1CH = 4, kT1Rows = kT1TY + 2, kT1Pitch = kT1TX + 8, kT1F4 = kT1Pitch / 4;
constexpr int kT1Stage = kT1CH * kT1Rows * kT1Pitch;          // floats per buffer
constexpr int kT1Ld = (kT1CH * kT1Rows * kT1F4 + 255) / 256;  // 16-byte loads per thread and stage
template <bool VEC>
__global__ __launch_bounds__(256) void k_conv_to1_k3(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int C, int D, int H,
                                                     int W, int nty, int ntx, int zc) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kT1Stage];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  int b = blockIdx.x;
  const int bx = b % ntx; b /= ntx;
  const int by = b % nty; b /= nty;
  const int zb = b * zc, ze = zb + zc < D ? zb + zc : D;
  const int x0 = bx * kT1TX, y0 = by * kT1TY, n = blockIdx.y;
  const long HW = (long)H * W, S = (long)D * HW;
  const float* xn = x + (long)n * C * S;
  const int z_lo = zb > 0 ? zb - 1 : 0, z_hi = ze < D ? ze + 1 : D;  // input planes [z_lo, z_hi)
  const int ncc = C / kT1CH, nst = (z_hi - z_lo) * ncc;

  // what a thread fetches per stage does not depend on the stage: element offset inside (this sample, channel block c0, plane 0).  The loads
  // are UNCONDITIONAL (an element outside the volume reads offset 0 and is zeroed when it is written to LDS): a load inside a branch makes
  // the compiler wait for it at the branch's end, one full memory latency per load (4.8 us per stage in the first version of this kernel)
  float4 st[kT1Ld];
  int off[kT1Ld];
  unsigned okm = 0;  // 4 bits per load: which of its four columns exist
#pragma unroll
  for (int k = 0; k < kT1Ld; ++k) {
    const int e = tid + 256 * k;
    const int c = e / (kT1Rows * kT1F4), r = (e / kT1F4) % kT1Rows, j = e % kT1F4;
    const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * j;
    const bool rowok = c < kT1CH && (unsigned)gy < (unsigned)H;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) m |= (rowok && (unsigned)(gx + i) < (unsigned)W) ? 1u << i : 0u;
    okm |= m << (4 * k);
    off[k] = m ? (int)((long)c * S + (long)gy * W + (VEC ? gx : 0)) : 0;  // (VEC: all four columns exist or none; else: the row's start)
  }
  auto fetch = [&](int s) __attribute__((always_inline)) {
    const int zi = z_lo + s / ncc, c0 = (s % ncc) * kT1CH;
    const float* pb = xn + (long)c0 * S + (long)zi * HW;  // (wave-uniform)
#pragma unroll
    for (int k = 0; k < kT1Ld; ++k) {
      if constexpr (VEC) {
        st[k] = *reinterpret_cast<const float4*>(pb + off[k]);
      } else {  // rows of any length: element by element, columns clamped into the row
        const int gx = x0 - 4 + 4 * ((tid + 256 * k) % kT1F4);
        const float* q = pb + off[k];
        const int hi = W - 1;
        st[k].x = q[min(max(gx, 0), hi)];
        st[k].y = q[min(max(gx + 1, 0), hi)];
        st[k].z = q[min(max(gx + 2, 0), hi)];
        st[k].w = q[min(max(gx + 3, 0), hi)];
      }
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kT1Ld; ++k) {
      const int e = tid + 256 * k;
      const unsigned m = (okm >> (4 * k)) & 15u;
      float4 v;
      v.x = (m & 1u) ? st[k].x : 0.f; v.y = (m & 2u) ? st[k].y : 0.f; v.z = (m & 4u) ? st[k].z : 0.f; v.w = (m & 8u) ? st[k].w : 0.f;
      if (e < kT1CH * kT1Rows * kT1F4) *reinterpret_cast<float4*>(lds + buf * kT1Stage + 4 * e) = v;
    }
  };

  float acc[3][4];  // [0]: output plane zi - 1 (complete after this input plane), [1]: zi, [2]: zi + 1
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[a][i] = 0.f;

  fetch(0);
  for (int s = 0; s < nst; ++s) {
    stash(s & 1);
    __syncthreads();
    if (s + 1 < nst) fetch(s + 1);
    const int zi = z_lo + s / ncc, cc = s % ncc;
    const float* base = lds + (s & 1) * kT1Stage + ty * kT1Pitch + 4 * tx + 3;
#pragma unroll 1
    for (int c = 0; c < kT1CH; ++c) {  // (not unrolled: 27 weights in scalar registers at a time, not 108)
      const float* wc = w + (long)(cc * kT1CH + c) * 27;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* rp = base + (c * kT1Rows + dy) * kT1Pitch;
        float row[6];
        row[0] = rp[0];
        const float4 m = *reinterpret_cast<const float4*>(rp + 1);
        row[1] = m.x; row[2] = m.y; row[3] = m.z; row[4] = m.w;
        row[5] = rp[5];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)  // input plane zi is tap dz of output plane zi + 1 - dz: accumulator 2 - dz
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float wv = wc[(dz * 3 + dy) * 3 + dx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[2 - dz][i] = fmaf(wv, row[i + dx], acc[2 - dz][i]);
          }
      }
    }
    if (cc == ncc - 1) {  // plane zi done: output plane zi - 1 is complete (or zi itself, at the volume's last plane)
      const int gy = y0 + ty, gx = x0 + 4 * tx;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int zo = zi - 1 + a;
        if ((a == 0 || zi == D - 1) && zo >= zb && zo < ze && gy < H) {
          float* yr = y + (long)n * S + (long)zo * HW + (long)gy * W;
          if (VEC && gx + 3 < W) {
            *reinterpret_cast<float4*>(yr + gx) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (gx + i < W) yr[gx + i] = acc[a][i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[0][i] = acc[1][i]; acc[1][i] = acc[2][i]; acc[2][i] = 0.f; }
    }
  }
}

// The same march with the stages brought in by LDS-DMA (rows that are whole 16-byte vectors): a ring of three slots, requests two stages
// ahead of the multiplication -- a workgroup's stages are strictly serial, so what bounds the kernel is how much memory latency one stage's
// arithmetic (0.5 us) has to hide, and registers are too few to keep two stages in flight.  Every wave issues kT1PW one-KiB pieces per stage
// (lanes beyond the tile or outside the volume read the zero page), so one hand-placed vmcnt(kT1PW) says "my pieces of this stage are in".
constexpr int kT1PW = 6, kT1Slot = 4 * kT1PW * 256, kT1NB = 3;  // floats per ring slot (24 pieces of 64 x 16 B), slots
static_assert(kT1Slot >= kT1Stage, "slot");
__global__ __launch_bounds__(256) void k_conv_to1_k3_dma(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                         const float* __restrict__ zeros, int C, int D, int H, int W, int nty, int ntx, int zc) {
  extern __shared__ __attribute__((aligned(1024))) float ring[];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int bx = b % ntx; b /= ntx;
  const int by = b % nty; b /= nty;
  const int zb = b * zc, ze = zb + zc < D ? zb + zc : D;
  const int x0 = bx * kT1TX, y0 = by * kT1TY, n = blockIdx.y;
  const long HW = (long)H * W, S = (long)D * HW;
  // blockIdx.z: a group of C / gridDim.z channels; its sum goes to partial volume blockIdx.z of y (k_to1_sum_parts adds them in group order)
  const int cg = C / (int)gridDim.z;
  const float* xn = x + ((long)n * C + (long)blockIdx.z * cg) * S;
  w += (long)blockIdx.z * cg * 27;
  y += (long)blockIdx.z * gridDim.y * S;
  const int z_lo = zb > 0 ? zb - 1 : 0, z_hi = ze < D ? ze + 1 : D;  // input planes [z_lo, z_hi)
  const int ncc = cg / kT1CH, nst = (z_hi - z_lo) * ncc;
  const unsigned ring_addr = nc_lds_addr(ring);

  long off[kT1PW];  // element offset of this lane's 16 bytes inside (sample, channel block c0, plane 0), or -1: zeros
#pragma unroll
  for (int i = 0; i < kT1PW; ++i) {
    const int e = (wv * kT1PW + i) * 64 + lane;
    const int c = e / (kT1Rows * kT1F4), r = (e / kT1F4) % kT1Rows, j = e % kT1F4;
    const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * j;
    const bool ok = c < kT1CH && (unsigned)gy < (unsigned)H && gx >= 0 && gx < W;
    off[i] = ok ? (long)c * S + (long)gy * W + gx : -1;
  }
  auto issue = [&](int s) __attribute__((always_inline)) {
    const int zi = z_lo + s / ncc, c0 = (s % ncc) * kT1CH;
    const float* pb = xn + (long)c0 * S + (long)zi * HW;
    const unsigned slot = ring_addr + (unsigned)((s % kT1NB) * kT1Slot * 4 + wv * kT1PW * 1024);
#pragma unroll
    for (int i = 0; i < kT1PW; ++i) nc_dma_lds16(off[i] >= 0 ? (const void*)(pb + off[i]) : (const void*)zeros, slot + i * 1024);
  };

  float acc[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[a][i] = 0.f;

  issue(0);
  if (nst > 1) issue(1);
  for (int s = 0; s < nst; ++s) {
    // this wave's pieces of stage s are in: everything but the youngest kT1PW requests (stage s + 1; result stores in between only make the
    // wait cover more) -- then everybody's, and everybody is done with the slot of stage s - 1, which stage s + 2 is requested into
    if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kT1PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (s + 2 < nst) issue(s + 2);
    const int zi = z_lo + s / ncc, cc = s % ncc;
    const float* base = ring + (s % kT1NB) * kT1Slot + ty * kT1Pitch + 4 * tx + 3;
#pragma unroll 1
    for (int c = 0; c < kT1CH; ++c) {
      const float* wc = w + (long)(cc * kT1CH + c) * 27;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* rp = base + (c * kT1Rows + dy) * kT1Pitch;
        float row[6];
        row[0] = rp[0];
        const float4 m = *reinterpret_cast<const float4*>(rp + 1);
        row[1] = m.x; row[2] = m.y; row[3] = m.z; row[4] = m.w;
        row[5] = rp[5];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float wv1 = wc[(dz * 3 + dy) * 3 + dx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[2 - dz][i] = fmaf(wv1, row[i + dx], acc[2 - dz][i]);
          }
      }
    }
    if (cc == ncc - 1) {
      const int gy = y0 + ty, gx = x0 + 4 * tx;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int zo = zi - 1 + a;
        if ((a == 0 || zi == D - 1) && zo >= zb && zo < ze && gy < H && gx < W)
          *reinterpret_cast<float4*>(y + (long)n * S + (long)zo * HW + (long)gy * W + gx) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[0][i] = acc[1][i]; acc[1][i] = acc[2][i]; acc[2][i] = 0.f; }
    }
  }
}

__global__ void k_to1_sum_parts(const float4* __restrict__ part, float4* __restrict__ y, long n4, int G) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 a = part[i];
    for (int g = 1; g < G; ++g) { const float4 b = part[(long)g * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    y[i] = a;
  }
}

size_t conv_fwd_to1_k3_ws_bytes(int N, int D, int H, int W) { return (size_t)4 * N * D * H * W * sizeof(float) + 256; }

int conv_fwd_to1_k3(const float* x, const float* w, float* y, int N, int C, int D, int H, int W, void* ws, size_t wsb, hipStream_t s) {
  if (N > 65535) { set_error("conv_fwd_to1_k3: more than 65535 samples"); return NC_ERR_SHAPE; }
  if (C % kT1CH) {  // (not the tail's shape: the general tile kernel)
    const int ntx = (W + 63) / 64, nty = (H + 7) / 8, ntz = (D + 7) / 8;
    hipLaunchKernelGGL(k_conv_to1<3>, dim3((unsigned)(ntx * nty * ntz), (unsigned)N), dim3(256), 0, s, x, w, (const float*)nullptr, y, C, D, H, W, 0);
    return check_launch("conv_fwd_to1_k3");
  }
  const int ntx = (W + kT1TX - 1) / kT1TX, nty = (H + kT1TY - 1) / kT1TY;
  // plane ranges: one round of the chip per channel group (measured at 108^3: 256 -> 216 us, 128 -> 264, 512 -> 245), at least 4 planes each (a
  // range reads 2 planes more than it writes)
  static const int want = 256;
  int nz = (int)((want + (long)ntx * nty * N - 1) / ((long)ntx * nty * N));
  if (nz > D / 4) nz = D / 4;
  if (nz < 1) nz = 1;
  const int zc = (D + nz - 1) / nz;
  nz = (D + zc - 1) / zc;
  const dim3 grid((unsigned)(ntx * nty * nz), (unsigned)N);
  static const bool dma = true;
  if (W % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
    const float* zeros = nc_zero_page();
    if (dma && zeros) {
      // channel groups (partial volumes, added in a fixed order): four times the workgroups without reading anything twice -- y has ONE channel
      static const int gwant = 4;
      int G = gwant;
      const long n4 = (long)N * D * H * W / 4;
      while (G > 1 && (C % (G * kT1CH) || !ws || wsb < (size_t)G * n4 * 16)) G >>= 1;
      if (int e = raise_dyn_lds(k_conv_to1_k3_dma, kT1NB * kT1Slot * 4, "conv_fwd_to1_k3")) return e;
      hipLaunchKernelGGL(k_conv_to1_k3_dma, dim3(grid.x, grid.y, (unsigned)G), dim3(256), kT1NB * kT1Slot * 4, s, x, w, G > 1 ? (float*)ws : y, zeros, C, D, H, W,
                         nty, ntx, zc);
      if (G > 1) hipLaunchKernelGGL(k_to1_sum_parts, dim3(1024), dim3(256), 0, s, (const float4*)ws, (float4*)y, n4, G);
    } else {
      hipLaunchKernelGGL(k_conv_to1_k3<true>, grid, dim3(256), 0, s, x, w, y, C, D, H, W, nty, ntx, zc);
    }
  } else {
    hipLaunchKernelGGL(k_conv_to1_k3<false>, grid, dim3(256), 0, s, x, w, y, C, D, H, W, nty, ntx, zc);
  }
  return check_launch("conv_fwd_to1_k3");
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of a Conv3d with ONE input channel and 64 output channels (7^3: G_B.first_layer, networks.py:899;
// 3^3: double_conv1.convolution.0, :420), stride 1, same padding, W % 4 == 0:
//     dW[co][t] = sum_{n,v} dY[n][co][v] * X[n][0][v + t - p]          t = (dz, dy, dx)
// GEMM with M = co (64), N = taps (KS^3, padded to 32 * NBW), K = voxels, v_mfma_f32_16x16x4_f32 (exact fp32).  With
// a single input channel the "ci" axis of the generic wgrad kernel would be 1/16 full; here the MFMA N axis is the
// tap axis instead: the B operand of tap t at voxel v is X[v + t], a shifted read of the same few image rows.
//   * one workgroup owns ALL taps (4 co-blocks x 2 * NBW tap-blocks of 16 x 16 accumulators, NBW per wave x 8 waves),
//     so dY -- the only large operand -- is read exactly once; the volume's rows (n, z, y) are split over 256
//     workgroups, partial dW per workgroup + fixed-order reduce (deterministic, no atomics);
//   * row streaming as in conv_mfma_wgrad.hip: a step = one output row; its dY row [64][W] and the KS new X rows
//     (row y + p of the KS planes z - p .. z + p) arrive by LDS-DMA (16 B per lane, zero page for padding and
//     out-of-volume rows) into a double buffer / a ring of KS + 1 row slots while the previous row is multiplied;
//   * MFMA m of a 16-voxel iteration reduces over voxels q + 4*kq + m: the dY operand of four MFMAs is one aligned
//     ds_read_b128, the X operand a ds_read_b32 at (tap offset of the lane) + q + 4*kq + m.
struct C1wParams {
  const float* x;
  const float* dy;
  float* slab;         // [parts][64][NT16]
  const float* zeros;  // >= 16 B of zeros in global memory
  int N, D, H, W;
  int PAr, SD;         // dY LDS pitch per channel, floats per dY buffer (whole 256-float pieces)
  int PRx;             // X row pitch in LDS (multiple of 4): [4 halo][W][>= pad], slot = 8 rows (KS planes + dummy)
  int W16;             // W rounded up to 16
  int parts;
};

template <int KS>
__global__ __launch_bounds__(512) void k_wgrad_c1(C1wParams p) {
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS * KS;
  constexpr int NB = ((TAPS + 31) / 32) * 2;  // tap blocks of 16 (even)
  constexpr int NBW = NB / 2;                 // tap blocks per wave
  constexpr int RING = KS + 1;
  constexpr int PLS = 8;                      // row slots per ring entry (KS planes, padded to 8)
  constexpr int MAXPD = 6;  // = kC1wMaxPD
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int mb = wave & 3, nb0 = wave >> 2;  // this wave: co block mb, tap blocks nb0 + 2 * j
  const int part = blockIdx.x;
  const long HW = (long)p.H * p.W, S = (long)p.D * HW;
  const long nrows = (long)p.N * p.D * p.H;
  const long r0 = nrows * part / p.parts, r1 = nrows * (part + 1) / p.parts;

  const int SXs = PLS * p.PRx;  // floats per ring entry
  float* xT = lds;
  float* dyT = lds + RING * SXs;
  const int npd = p.SD / 256, npx = SXs / 256;

  f32x4 acc[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane tap decode of this wave's blocks: (dz, dy) row selector and column shift dx + 4 - PAD
  int tdz[NBW], tdy[NBW], tdx[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    int t = (nb0 + 2 * j) * 16 + l15;
    if (t >= TAPS) t = 0;  // padded taps: any valid address, result dropped
    tdz[j] = t / (KS * KS);
    tdy[j] = (t / KS) % KS;
    tdx[j] = t % KS + 4 - PAD;
  }
  // per-lane DMA sources of this wave's pieces (row independent): dY [64][PAr] and one ring entry [8][PRx]
  int gd[MAXPD], gxp = -1, gxc = 0;
#pragma unroll
  for (int i = 0; i < MAXPD; ++i) {
    const int f = ((wave + 8 * i) * 64 + lane) * 4;
    const int c = f / p.PAr, col = f - c * p.PAr;
    gd[i] = (c < 64 && col < p.W) ? (int)(c * S + col) : -1;
  }
  {
    const int f = (wave * 64 + lane) * 4;  // piece `wave` of the ring entry (npx <= 8)
    const int pl = f / p.PRx, col = f - pl * p.PRx;
    const int x = col - 4;
    if (wave < npx && pl < KS && x >= 0 && x < p.W) {
      gxp = pl;
      gxc = x;
    }
  }

  auto issue = [&](int n, int z, int yy, bool with_dy, int cnt, int dcnt) {
    if (wave < npx) {
      const int zz = z + gxp - PAD;
      const bool ok = gxp >= 0 && yy >= 0 && yy < p.H && zz >= 0 && zz < p.D;
      const float* src = ok ? p.x + (long)n * S + (long)zz * HW + (long)yy * p.W + gxc : p.zeros;
      nc_dma_lds16(src, nc_lds_addr((xT + (cnt % RING) * SXs + wave * 256)));
    }
    if (with_dy) {
      const float* dbse = p.dy + (long)n * 64 * S + (long)z * HW + (long)(yy - PAD) * p.W;
      float* ds = dyT + (dcnt & 1) * p.SD;
#pragma unroll
      for (int i = 0; i < MAXPD; ++i) {
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
    n = (int)(plane / p.D);
    z = (int)(plane - (long)n * p.D);
    yy = ya - PAD;
    r += yb - ya;
    have = true;
  };
  next_segment();
  int cnt = 0, dcnt = 0;
  if (have) issue(n, z, yy, yy - PAD >= ya, cnt, dcnt);

  const int a_off = (mb * 16 + l15) * p.PAr + 4 * kq;
  while (have) {
    const bool cdy = yy - PAD >= ya;
    const int ccnt = cnt, cdc = dcnt;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++cnt;
    if (cdy) ++dcnt;
    ++yy;
    if (yy > yb - 1 + PAD) next_segment();
    if (have) issue(n, z, yy, yy - PAD >= ya, cnt, dcnt);
    if (cdy) {
      const float* pa = dyT + (cdc & 1) * p.SD + a_off;
      // ring entry of kernel row dy: the step in flight loaded row y + PAD at entry ccnt
      const float* pb[NBW];
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int e = (ccnt - (KS - 1) + tdy[j] + 8 * RING) % RING;
        pb[j] = xT + e * SXs + tdz[j] * p.PRx + tdx[j] + 4 * kq;
      }
#pragma unroll 1
      for (int q16 = 0; q16 < p.W16; q16 += 16) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(pa + q16, 16));
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
          const float* q = pb[j] + q16;
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], q[m], acc[j], 0, 0, 0);
        }
      }
    }
  }

  // ---- partial slab[part][co][tap16]: C/D layout col (tap) = lane & 15, row (co) = 4 * (lane >> 4) + r
  float* sl = p.slab + (long)part * 64 * (NB * 16);
#pragma unroll
  for (int j = 0; j < NBW; ++j)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      sl[(long)(mb * 16 + 4 * kq + rr) * (NB * 16) + (nb0 + 2 * j) * 16 + l15] = acc[j][rr];
}

// one wave per output element, lanes over the workgroup slabs, fixed-order butterfly (deterministic)
__global__ void k_wgrad_c1_reduce(const float* __restrict__ slab, float* __restrict__ dw, int parts, int taps, int nt16) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= 64 * taps) return;
  const int lane = threadIdx.x & 63;
  const int co = i / taps, t = i - co * taps;
  float s = 0.f;
  for (int w = lane; w < parts; w += 64) s += slab[((long)w * 64 + co) * nt16 + t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) dw[i] = s;
}

static int c1w_nt16(int KS) { return ((KS * KS * KS + 31) / 32) * 32; }

static constexpr int kC1wMaxPD = 6;  // dY pieces per wave per step

// geometry of k_wgrad_c1 for this shape; false if it does not fit (LDS, DMA pieces per wave)
static bool c1w_plan(const ConvDims& d, C1wParams& p, int& lds_bytes) {
  if (d.C != 1 || d.K != 64 || d.kd != d.kh || d.kh != d.kw || (d.kd != 3 && d.kd != 7)) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.pd != d.kd / 2 || d.ph != d.kd / 2 || d.pw != d.kd / 2) return false;
  if (d.W % 4 || d.W < 16) return false;
  if ((long)d.D * d.H * d.W * 64 >= (1L << 31)) return false;
  p.N = d.N; p.D = d.D; p.H = d.H; p.W = d.W;
  p.W16 = (d.W + 15) & ~15;
  int pa = (p.W16 + 7) & ~7;
  if (!(pa & 8)) pa += 8;  // = 8 (mod 16): conflict-free ds_read_b128
  p.PAr = pa;
  p.SD = (64 * p.PAr + 255) & ~255;
  // X row: 4 halo columns + W16 + (KS - 1 + 3) columns of read-ahead, 8 rows per ring entry = whole 256-float pieces
  p.PRx = (4 + p.W16 + d.kd + 2 + 31) & ~31;
  const int RING = d.kd + 1;
  lds_bytes = (RING * 8 * p.PRx + 2 * p.SD) * (int)sizeof(float);
  return lds_bytes <= 160 * 1024 && p.SD / 256 <= 8 * kC1wMaxPD && 8 * p.PRx / 256 <= 8;
}

bool c1_wgrad_supported(const ConvDims& d) {
  C1wParams p{};
  int lds;
  return c1w_plan(d, p, lds);
}

static int c1w_parts(const ConvDims& d) {
  const long rows = (long)d.N * d.D * d.H;
  return (int)(rows < 256 ? rows : 256);
}

size_t c1_wgrad_ws_bytes(const ConvDims& d) {
  if (!c1_wgrad_supported(d)) return 0;
  return (size_t)c1w_parts(d) * 64 * c1w_nt16(d.kd) * sizeof(float) + 256;
}

template <int KS>
static int launch_c1w(const C1wParams& p, int lds_bytes, hipStream_t s) {
  auto kern = k_wgrad_c1<KS>;
  if (int e = raise_dyn_lds(kern, 160 * 1024, "wgrad_c1")) return e;
  hipLaunchKernelGGL(kern, dim3(p.parts), dim3(512), lds_bytes, s, p);
  return check_launch("wgrad_c1");
}

int conv_wgrad_c1(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s) {
  const size_t need = c1_wgrad_ws_bytes(d);
  if (!need) {
    set_error("wgrad_c1: unsupported shape");
    return NC_ERR_SHAPE;
  }
  if (!ws || wsb < need) {
    set_error("wgrad_c1: workspace too small (%zu < %zu)", wsb, need);
    return NC_ERR_WS;
  }
  C1wParams p{};
  int lds_bytes = 0;
  if (!c1w_plan(d, p, lds_bytes)) {
    set_error("wgrad_c1: unsupported shape");
    return NC_ERR_SHAPE;
  }
  p.x = x; p.dy = dy; p.slab = (float*)ws; p.zeros = nc_zero_page();
  if (!p.zeros) { set_error("wgrad_c1: no zero page"); return NC_ERR_HIP; }
  p.parts = c1w_parts(d);
  const int e = d.kd == 7 ? launch_c1w<7>(p, lds_bytes, s) : launch_c1w<3>(p, lds_bytes, s);
  if (e) return e;
  const int taps = d.kd * d.kd * d.kd;
  hipLaunchKernelGGL(k_wgrad_c1_reduce, dim3((64 * taps + 3) / 4), dim3(256), 0, s, (const float*)ws, dw, p.parts,
                     taps, c1w_nt16(d.kd));
  return check_launch("wgrad_c1_reduce");
}

// ---------------------------------------------------------------------------------------------------------------
// Data gradient of Conv3d(1, 64, 7, padding 3) on the matrix cores (the VALU kernel k_conv_to1 above stays as the
// fallback): dX[u] = sum_{co,t} W[co][t] * dY[co][u + p - t].  One output channel leaves no M axis for an implicit
// GEMM over voxels; instead, per input row of dY, the MFMA computes  Z[(dy,dx)][v] = sum_co W[co][dz,dy,dx] * dY[co][v]
// (M = 2 dy x 7 dx of 16 rows, N = 16 voxels, K = 64 channels, v_mfma_f32_16x16x4_f32) and Z is folded back by index
// arithmetic: Z[(dy,dx)][v] belongs to dX[v + (dz,dy,dx) - p].
//   * a workgroup owns two kernel planes dz (waves 0-3 / 4-7), a wave two kernel rows dy; all read the same dY row
//     [64][W] from LDS (LDS-DMA, double buffered, one barrier per row);
//   * fold in x: the wave parks its Z rows in a private LDS strip and reads them back shifted, c_dy[x] = sum_dx
//     Z[dy,dx][x + p - dx];  fold in y: c_dy of input row y' belongs to output row y' + dy - p -- the seven dy of a
//     step hit seven DIFFERENT rows of a small LDS ring, so every ring row receives exactly one addend per step, from
//     one wave, in step order: plain read-modify-write, deterministic, no atomics.  The row that just got its dy = 0
//     addend is complete and is written out;  fold in z: one partial volume per dz + a fixed-order reduce kernel;
//   * the volume's rows are split over workgroups with a 3-row halo of dY on either side (recomputed, 13 % at 108^3)
//     so that every output row is completed inside one workgroup.
struct T1Params {
  const float* dy;
  const float* w;
  float* slab;         // [7 dz][N][D][H][W]
  const float* zeros;
  int N, D, H, W;      // W = width of the column segment this launch works on
  int Wf, x0;          // row pitch of the tensors, first column of the segment (x0 % 4 == 0)
  int kx0, kx1;        // segment columns [kx0, kx1) are written (the rest is the 3-column overlap with the neighbour)
  int PAr, SD;         // dY LDS pitch per channel (= 16 mod 32), floats per dY buffer (whole pieces)
  int W16;
  int parts;
};

static constexpr int kZsPitch = 124;  // Z strip pitch: 4 rows apart = 16 banks apart

__global__ __launch_bounds__(512) void k_dgrad_to1(T1Params p) {
  constexpr int KS = 7, PAD = 3, RINGR = 8, MAXPD = 5;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int dzi = wave >> 2, j2 = wave & 3;
  const int dz = 2 * blockIdx.y + dzi;  // kernel plane of this half of the workgroup (7 = none)
  const bool wave_on = dz < KS;
  const int part = blockIdx.x;
  const long HW = (long)p.H * p.Wf, S = (long)p.D * HW;
  const long nrows = (long)p.N * p.D * p.H;  // rows (n, z', y') of dY
  const long r0 = nrows * part / p.parts, r1 = nrows * (part + 1) / p.parts;

  float* dyT = lds;                                   // 2 x SD
  float* zs = lds + 2 * p.SD + wave * 16 * kZsPitch;  // this wave's Z strip [16][4 + W16 + 4 .. pitch 124]
  float* ring = lds + 2 * p.SD + 8 * 16 * kZsPitch + dzi * RINGR * 128;  // [8 rows][128] per dz half
  for (int i = tid; i < 8 * 16 * kZsPitch + 2 * RINGR * 128; i += 512) lds[2 * p.SD + i] = 0.f;
  const int npd = p.SD / 256;

  // A operand: row m of the 16 x 4 tile is tap (dy, dx): m < 7 -> (2*j2, m), 8 <= m < 15 -> (2*j2 + 1, m - 8)
  float aw[16];
  {
    const int dyy = 2 * j2 + (l15 >> 3), dx = l15 & 7;
    const bool tap_ok = wave_on && dyy < KS && dx < KS;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      aw[ks] = tap_ok ? p.w[(long)(4 * ks + kq) * (KS * KS * KS) + (dz * KS + dyy) * KS + dx] : 0.f;
  }
  int gd[MAXPD];
#pragma unroll
  for (int i = 0; i < MAXPD; ++i) {
    const int f = ((wave + 8 * i) * 64 + lane) * 4;
    const int c = f / p.PAr, col = f - c * p.PAr;
    gd[i] = (c < 64 && col < p.W) ? (int)(c * S + col) : -1;
  }
  auto issue = [&](int n, int z, int y, int buf) {
    const float* dbse = p.dy + (long)n * 64 * S + (long)z * HW + (long)y * p.Wf + p.x0;
    float* ds = dyT + buf * p.SD;
#pragma unroll
    for (int i = 0; i < MAXPD; ++i) {
      const int jj = wave + 8 * i;
      if (jj < npd) {
        const float* src = gd[i] >= 0 ? dbse + gd[i] : p.zeros;
        nc_dma_lds16(src, nc_lds_addr((ds + jj * 256)));
      }
    }
  };

  // ---- walk the rows [r0, r1): per (n, z') plane segment [ya, yb) the input rows run yl = max(0, ya-3) .. min(H, yb+3)
  long r = r0;
  int n = 0, z = 0, ya = 0, yb = 0, yl = 0, yh = 0, y = 0;
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
    n = (int)(plane / p.D);
    z = (int)(plane - (long)n * p.D);
    yl = max(0, ya - PAD);
    yh = min(p.H, yb + PAD);
    y = yl;
    r += yb - ya;
    have = true;
  };
  next_segment();
  int cnt = 0;
  __syncthreads();
  if (have) issue(n, z, y, 0);
  const int b_off = kq * p.PAr + l15;  // dY[4*ks + kq][q16 + l15]
  while (have) {
    const int cn = n, cz = z, cy = y, cya = ya, cyb = yb, cbuf = cnt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the row has landed; the previous step's ring updates are done
    ++cnt;
    ++y;
    if (y >= yh) next_segment();
    if (have) issue(n, z, y, cnt & 1);
    const int oz = cz + dz - PAD;  // output plane of this wave's kernel plane
    if (wave_on && oz >= 0 && oz < p.D) {
      const float* pb = dyT + cbuf * p.SD + b_off;
      // Z[(dy,dx)][v] for the whole row, 16 voxels at a time, parked in the wave's strip (columns 4 .. 4 + W16)
#pragma unroll 1
      for (int q16 = 0; q16 < p.W16; q16 += 16) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks], pb[4 * ks * p.PAr + q16], acc, 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) zs[(4 * kq + rr) * kZsPitch + 4 + q16 + l15] = acc[rr];
      }
      // fold in x and y: kernel row dy of this input row goes to output row cy + dy - PAD
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int dyy = 2 * j2 + e;
        const int oy = cy + dyy - PAD;
        if (dyy < KS && oy >= cya && oy < cyb) {
          float* rrow = ring + (oy & (RINGR - 1)) * 128;
#pragma unroll 1
          for (int x = lane; x < p.W; x += 64) {
            float c = 0.f;
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) c += zs[(8 * e + dx) * kZsPitch + 4 + x + PAD - dx];
            const float tot = rrow[x] + c;
            if (dyy == 0 || cy == p.H - 1) {  // last addend of this output row (input row oy + 3, or the plane's last)
              if (x >= p.kx0 && x < p.kx1)
                p.slab[(((long)dz * p.N + cn) * p.D + oz) * HW + (long)oy * p.Wf + p.x0 + x] = tot;
              rrow[x] = 0.f;
            } else {
              rrow[x] = tot;
            }
          }
        }
      }
    }
  }
}

// dx[n][z][y][x] = sum over the kernel planes dz whose input plane z - dz + p exists, in dz order
__global__ void k_dgrad_to1_reduce(const float* __restrict__ slab, float* __restrict__ dx, int N, int D, long HW) {
  const long total = (long)N * D * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)((i / HW) % D);
    float s = 0.f;
#pragma unroll
    for (int dz = 0; dz < 7; ++dz) {
      const int zi = z - dz + 3;
      if (zi >= 0 && zi < D) s += slab[(long)dz * total + i];
    }
    dx[i] = s;
  }
}

bool to1_mfma_supported(const ConvDims& d) {
  return d.C == 1 && d.K == 64 && d.kd == 7 && d.kh == 7 && d.kw == 7 && d.sd == 1 && d.sh == 1 && d.sw == 1 &&
         d.pd == 3 && d.ph == 3 && d.pw == 3 && d.W % 4 == 0 && d.W >= 16 && d.W <= 208 &&
         (long)d.D * d.H * d.W * 64 < (1L << 31);
}

size_t to1_mfma_ws_bytes(const ConvDims& d) {
  if (!to1_mfma_supported(d)) return 0;
  return (size_t)7 * d.N * d.D * d.H * d.W * sizeof(float) + 256;
}

int conv_dgrad_to1_mfma(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                        hipStream_t s) {
  const size_t need = to1_mfma_ws_bytes(d);
  if (!need) {
    set_error("dgrad_to1: unsupported shape");
    return NC_ERR_SHAPE;
  }
  if (!ws || wsb < need) {
    set_error("dgrad_to1: workspace too small (%zu < %zu)", wsb, need);
    return NC_ERR_WS;
  }
  T1Params p{};
  p.dy = dy; p.w = w; p.slab = (float*)ws; p.zeros = nc_zero_page();
  if (!p.zeros) { set_error("dgrad_to1: no zero page"); return NC_ERR_HIP; }
  p.N = d.N; p.D = d.D; p.H = d.H; p.Wf = d.W;
  // rows wider than 112 columns (the Z strips and the ring are sized for 112) are done as two column segments that
  // overlap by the 3-column reach of the kernel on either side of the cut; each writes its own half of the row
  const int nseg = d.W <= 112 ? 1 : 2;
  const int cut = nseg == 1 ? d.W : ((d.W / 2 + 3) & ~3);  // first column of the second half, a multiple of 4
  for (int seg = 0; seg < nseg; ++seg) {
  if (nseg == 1) { p.x0 = 0; p.W = d.W; p.kx0 = 0; p.kx1 = d.W; }
  else if (seg == 0) { p.x0 = 0; p.W = cut + 4; p.kx0 = 0; p.kx1 = cut; }
  else { p.x0 = cut - 4; p.W = d.W - p.x0; p.kx0 = 4; p.kx1 = p.W; }
  p.W16 = (p.W + 15) & ~15;
  int pa = (p.W16 + 15) & ~15;
  if (pa % 32 != 16) pa += 16;  // = 16 (mod 32): the two k rows of a half-wave read disjoint banks
  p.PAr = pa;
  p.SD = (64 * p.PAr + 255) & ~255;
  const long rows = (long)d.N * d.D * d.H;
  p.parts = (int)(rows < 64 ? rows : 64);
  const int lds_bytes = (2 * p.SD + 8 * 16 * kZsPitch + 2 * 8 * 128) * (int)sizeof(float);
  if (int e = raise_dyn_lds(k_dgrad_to1, 160 * 1024, "dgrad_to1")) return e;
  hipLaunchKernelGGL(k_dgrad_to1, dim3(p.parts, 4), dim3(512), lds_bytes, s, p);
  if (int e = check_launch("dgrad_to1")) return e;
  }
  hipLaunchKernelGGL(k_dgrad_to1_reduce, dim3(2048), dim3(256), 0, s, (const float*)ws, dx, d.N, d.D,
                     (long)d.H * d.W);
  return check_launch("dgrad_to1_reduce");
}


// ---------------------------------------------------------------------------------------------------------------------
// Many channels -> ONE channel, any kernel / stride (the PatchGAN head, Conv(512, 1, k4, s1, p1), networks.py:1057):
// forward and weight gradient.  As a GEMM this layer has one row; on the 64-row MFMA tiles it ran at 0.6 TFLOP/s
// (0.37 ms per call at Athena's batch for 32 MB of input).  These are reductions, so they are written as reductions:
//   forward : workgroup = 64 output positions x 16 channel groups (one wave each: the weight of a (channel, tap) is
//             wave-uniform -> scalar loads; lanes read consecutive positions); taps outside, channels inside, 8 loads
//             in flight; the 16 partial sums are added in group order through LDS (deterministic).
//   wgrad   : workgroup = one input channel; threads stride over (sample, output position), one partial sum per tap
//             (4 x 4 or 4 x 4 x 4 kernels) per thread, block tree reduce in a fixed order.
struct K1Params {
  const float* x;
  const float* w;     // fwd: [1][C][taps]
  const float* bias;  // fwd, nullable
  float* y;           // fwd: [N][1][So];  wgrad: dw [1][C][taps]
  const float* dy;    // wgrad: [N][1][So]
  ConvDims d;
  long S, So;
  int taps;
};

static constexpr int kK1Groups = 16;  // channel groups = waves per workgroup (latency hiding: the layer is tiny)

__global__ __launch_bounds__(64 * kK1Groups) void k_conv_k1_fwd(K1Params p) {
  __shared__ float part[kK1Groups][64];
  const ConvDims& d = p.d;
  const int lane = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.y;
  const long pos = (long)blockIdx.x * 64 + lane;
  const bool valid = pos < p.So;
  const long q = valid ? pos : 0;
  const int HoWo = d.Ho * d.Wo;
  const int od = (int)(q / HoWo), oh = (int)((q - (long)od * HoWo) / d.Wo), ow = (int)(q - (long)od * HoWo - (long)oh * d.Wo);
  const int cper = (d.C + kK1Groups - 1) / kK1Groups, cb = min(d.C, grp * cper), ce = min(d.C, cb + cper);
  const long HW = (long)d.H * d.W;
  const float* xn = p.x + (long)n * d.C * p.S;
  float acc = 0.f;
  for (int kz = 0; kz < d.kd; ++kz) {
    const int iz = od * d.sd - d.pd + kz;
    for (int ky = 0; ky < d.kh; ++ky) {
      const int iy = oh * d.sh - d.ph + ky;
      for (int kx = 0; kx < d.kw; ++kx) {
        const int ix = ow * d.sw - d.pw + kx;
        const bool ok = valid && (unsigned)iz < (unsigned)d.D && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
        const float* xp = xn + (ok ? (long)iz * HW + (long)iy * d.W + ix : 0);
        const float* wp = p.w + (kz * d.kh + ky) * d.kw + kx;
        int c = cb;
        for (; c + 8 <= ce; c += 8) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = ok ? xp[(long)(c + u) * p.S] : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = fmaf(v[u], wp[(long)(c + u) * p.taps], acc);
        }
        for (; c < ce; ++c) acc = fmaf(ok ? xp[(long)c * p.S] : 0.f, wp[(long)c * p.taps], acc);
      }
    }
  }
  part[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && valid) {
    float v = part[0][lane];
#pragma unroll
    for (int g2 = 1; g2 < kK1Groups; ++g2) v += part[g2][lane];  // group order: deterministic
    if (p.bias) v += p.bias[0];
    p.y[(long)n * p.So + pos] = v;
  }
}

template <int KD>  // kernel KD x 4 x 4 (the PatchGAN head: KD = 1 in 2-D, 4 in 3-D)
__global__ __launch_bounds__(256) void k_conv_k1_wgrad(K1Params p) {
  constexpr int T = KD * 16;
  __shared__ float red[256];
  const ConvDims& d = p.d;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int HoWo = d.Ho * d.Wo;
  const long HW = (long)d.H * d.W;
  const long total = (long)d.N * p.So;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  for (long i = tid; i < total; i += 256) {
    const int n = (int)(i / p.So);
    const long q = i - (long)n * p.So;
    const int od = (int)(q / HoWo), oh = (int)((q - (long)od * HoWo) / d.Wo), ow = (int)(q - (long)od * HoWo - (long)oh * d.Wo);
    const float g = p.dy[i];
    const float* xc = p.x + ((long)n * d.C + c) * p.S;
#pragma unroll
    for (int kz = 0; kz < KD; ++kz) {
      const int iz = od * d.sd - d.pd + kz;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        const int iy = oh * d.sh - d.ph + ky;
        const bool oky = (unsigned)iz < (unsigned)d.D && (unsigned)iy < (unsigned)d.H;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const int ix = ow * d.sw - d.pw + kx;
          const bool ok = oky && (unsigned)ix < (unsigned)d.W;
          const float v = ok ? xc[(long)iz * HW + (long)iy * d.W + ix] : 0.f;
          acc[(kz * 4 + ky) * 4 + kx] = fmaf(g, v, acc[(kz * 4 + ky) * 4 + kx]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    red[tid] = acc[t];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) p.y[(long)c * T + t] = red[0];
    __syncthreads();
  }
}

bool k1_fwd_supported(const ConvDims& d) {
  return d.K == 1 && (long)d.C * d.kd * d.kh * d.kw >= 256 && (long)d.Do * d.Ho * d.Wo * d.N >= 1024;
}
bool k1_wgrad_supported(const ConvDims& d) {
  return k1_fwd_supported(d) && d.kh == 4 && d.kw == 4 && (d.kd == 1 || d.kd == 4);
}

static void k1_params(K1Params& p, const ConvDims& d) {
  p.d = d;
  p.S = (long)d.D * d.H * d.W;
  p.So = (long)d.Do * d.Ho * d.Wo;
  p.taps = d.kd * d.kh * d.kw;
}

int conv_fwd_k1(const float* x, const float* w, const float* b, float* y, const ConvDims& d, hipStream_t s) {
  K1Params p{};
  k1_params(p, d);
  p.x = x; p.w = w; p.bias = b; p.y = y;
  hipLaunchKernelGGL(k_conv_k1_fwd, dim3((unsigned)cdiv(p.So, 64), d.N), dim3(64 * kK1Groups), 0, s, p);
  return check_launch("conv_fwd_k1");
}

int conv_wgrad_k1(const float* x, const float* dy, float* dw, const ConvDims& d, hipStream_t s) {
  K1Params p{};
  k1_params(p, d);
  p.x = x; p.dy = dy; p.y = dw;
  if (d.kd == 1) hipLaunchKernelGGL(k_conv_k1_wgrad<1>, dim3(d.C), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_conv_k1_wgrad<4>, dim3(d.C), dim3(256), 0, s, p);
  return check_launch("conv_wgrad_k1");
}

}  // namespace nc
