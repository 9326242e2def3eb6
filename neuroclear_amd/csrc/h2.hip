// "H2" operands: an fp32 tensor as TWO fp16 terms of the tensor times a power of two (s3_common.hpp), the input form of k_conv_s3x<KS, NCB, 2>
// (three fp16 MFMA products per fp32 product where the three-term bf16 form needs six).  Layout: [N][C/8][2 terms][voxels][8] fp16 = 16-byte
// units like S3 with two sub-blocks.  The power of two comes from a CELL, one unsigned per tensor holding the float bits of a magnitude that
// bounds the tensor from above within a factor of two: the largest finite |x| (k_absmax; atomicMax, order-independent, deterministic), or a
// bound known by construction (h2_set_cell: |InstanceNorm output| <= sqrt(S - 1)).
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "common.hpp"
#include "s3_common.hpp"

namespace nc {
namespace {

__global__ void __launch_bounds__(256) k_absmax(const float* __restrict__ x, long n, unsigned* __restrict__ cell, unsigned* __restrict__ cell2) {
  unsigned m = 0;
  // a pointer that is not 16-byte aligned (a view into a larger tensor): up to three leading elements are taken one by one
  const int head = (int)((4 - (((unsigned long long)x >> 2) & 3)) & 3) < n ? (int)((4 - (((unsigned long long)x >> 2) & 3)) & 3) : (int)n;
  if (blockIdx.x == 0 && (int)threadIdx.x < head) {
    const unsigned b = __float_as_uint(x[threadIdx.x]) & 0x7fffffffu;
    if (b < 0x7f800000u && b > m) m = b;
  }
  x += head;
  n -= head;
  const long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  auto take = [&](const float4& v) __attribute__((always_inline)) {
    const unsigned b[4] = {__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu, __float_as_uint(v.z) & 0x7fffffffu,
                           __float_as_uint(v.w) & 0x7fffffffu};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (b[j] < 0x7f800000u && b[j] > m) m = b[j];  // non-finite elements do not set the scale: they become NaN terms of their own
  };
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight per thread (one per pass ran at 1.4-2 TB/s)
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    take(v0); take(v1); take(v2); take(v3);
  }
  for (; i < n4; i += stride) take(x4[i]);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const unsigned b = __float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7fffffffu;
    if (b < 0x7f800000u && b > m) m = b;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned q = (unsigned)__shfl_xor((int)m, o);
    m = q > m ? q : m;
  }
  // one atomic per BLOCK: thousands of waves hitting one address serialise (32 K atomics cost 0.15 ms of a 0.28 ms launch)
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned b = wm[0];
    for (int k = 1; k < 4; ++k) b = wm[k] > b ? wm[k] : b;
    if (b) {
      atomicMax(cell, b);
      if (cell2) atomicMax(cell2, b);
    }
  }
}

__global__ void k_set_cells(unsigned* cells, int n, unsigned bits) {
  if ((int)threadIdx.x < n) cells[threadIdx.x] = bits;
}

// fp32 [N][C][S] (samples xstride floats apart) -> channels ob0*8 .. of an H2 tensor with `oblocks` 8-channel blocks per sample.
// guard != NULL (the RANGE GUARD of a measured cell, common.hpp): a wave's 64 voxels x 8 channels are one CHUNK; the kernel counts the
// chunks that hold a finite non-zero element (guard[kGuardAll]) and those whose LARGEST magnitude lies below 2^-17 of the cell
// (guard[kGuardLow]: every element of such a chunk has lost bits of its second term).  Integer atomics: order-independent, deterministic.
__global__ void __launch_bounds__(256) k_split2h(const float* __restrict__ x, uint4* __restrict__ out, long S, int cblocks, int oblocks, int ob0,
                                                 long xstride, const unsigned* __restrict__ cell, unsigned* __restrict__ guard) {
  const unsigned cb_bits = *cell;
  const float sc = h2_scale(cb_bits);
  const int n = blockIdx.y / cblocks, cb = blockIdx.y % cblocks;
  const float* xb = x + (long)n * xstride + (long)cb * 8 * S;
  const long ob = (long)n * oblocks + ob0 + cb;
  unsigned n_all = 0, n_low = 0;  // (lane 0 of each wave counts its wave's chunks)
  // a block walks several 256-voxel tiles (gridDim.x is capped): the guard's counts leave as ONE pair of atomics per block -- one pair per
  // tile (39 K blocks on two addresses at 64 x 108^3) cost four times the conversion itself
  for (long v0 = (long)blockIdx.x * 256; v0 < S; v0 += (long)gridDim.x * 256) {
    const long v = v0 + threadIdx.x;
    unsigned m = 0;
    if (v < S) {
      unsigned short e[8][3];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = xb[j * S + v];
        const unsigned b = __float_as_uint(f) & 0x7fffffffu;
        if (b < 0x7f800000u && b > m) m = b;
        h2_split(f * sc, e[j]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) out[(ob * 2 + t) * S + v] = s3_unit(e, t);
    }
    if (guard) {  // (two ballots instead of a cross-lane maximum: is any lane's value at or above the threshold / non-zero)
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

// The guard's decision, one thread: f = more than 1 / kGuardShare of the non-zero chunks of EITHER tensor measured in this call (ga, gb nullable)
// lie below 2^-17 of their tensor's cell.  can_flip: the caller launches the three-term kernels behind this (they run when the flag is set,
// the two-term ones when it is clear); else f is only counted.  prior (nullable): the words of an operand whose decision was taken earlier
// (conv_bwd_s3's dY): the call's flag is the OR, and the operand's own flag follows it (the caller re-converts that operand too).
// stats: host-visible counters (nc_h2_guard_stats).
// total_a != 0: ga's producer counted ZERO chunks instead of non-zero ones (the norm backward: atomics only for the rare chunk): all = total_a - ga[kGuardAll]
__global__ void k_h2_guard_decide(unsigned* ga, unsigned* gb, unsigned* prior, unsigned* flag, int can_flip, unsigned long long* stats,
                                  unsigned long long total_a) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned f = 0;
  unsigned long long worst = 0;
  int n = 0;
  for (unsigned* g : {ga, gb}) {
    if (!g) continue;
    ++n;
    const unsigned long long lo = g[kGuardLow], all = (g == ga && total_a) ? total_a - g[kGuardAll] : g[kGuardAll];
    if (all && lo * kGuardShare > all) f = 1;
    const unsigned long long ppm = all ? lo * 1000000ull / all : 0;
    worst = ppm > worst ? ppm : worst;
  }
  unsigned F = f && can_flip ? 1u : 0u;
  if (prior && prior[kGuardFlag]) F = 1u;
  *flag = F;
  if (prior) prior[kGuardFlag] = F;
  if (stats && n) {
    __hip_atomic_fetch_add(stats + 0, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (f) __hip_atomic_fetch_add(stats + (can_flip ? 1 : 2), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_max(stats + 3, worst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// An H2 tensor back to fp32 values and on to the S3 form (exact: a0 + a1 has at most 23 significant bits and 2^-k is a power of two), into a
// SEPARATE buffer -- for the partner of a flagged operand that exists in H2 form only.  Runs when guard[kGuardFlag] is set.
__global__ void __launch_bounds__(256) k_h2_to_s3_if(const uint4* __restrict__ in, uint4* __restrict__ out, long S, int cblocks,
                                                      const unsigned* __restrict__ cells, const unsigned* __restrict__ guard) {
  if (guard_skip(guard, 1)) return;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const int cb = blockIdx.y % cblocks;
  const float inv = h2_inv_scale(cells[cb >= cblocks / 2 ? 1 : 0]);
  const long b = blockIdx.y;
  const uint4 a = in[(b * 2 + 0) * S + v], r = in[(b * 2 + 1) * S + v];
  const unsigned aw[4] = {a.x, a.y, a.z, a.w}, rw[4] = {r.x, r.y, r.z, r.w};
  unsigned short e[8][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned short h0 = (unsigned short)(aw[j >> 1] >> ((j & 1) * 16)), h1 = (unsigned short)(rw[j >> 1] >> ((j & 1) * 16));
    const float val = ((float)__builtin_bit_cast(_Float16, h0) + (float)__builtin_bit_cast(_Float16, h1)) * inv;
    s3_split(val, e[j]);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) out[(b * 3 + t) * S + v] = s3_unit(e, t);
}

// InstanceNorm normalisation + (Leaky)ReLU (k_act_split3's arithmetic, operation for operation) writing the H2 form -- and the fp32 tensor too
// when y != NULL
__global__ void __launch_bounds__(256) k_act_split2h(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float slope, float* __restrict__ y, long ystride, uint4* __restrict__ out, long S, int cblocks,
                                                     int oblocks, int ob0, unsigned bound_bits, unsigned* __restrict__ cell, unsigned* __restrict__ cell2) {
  // the cell is a bound the HOST knows: the scale comes by value, and one thread leaves it in the cell(s) for the consumers (later kernels)
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (cell) *cell = bound_bits;
    if (cell2) *cell2 = bound_bits;
  }
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= S) return;
  const float sc = h2_scale(bound_bits);
  const int n = blockIdx.y / cblocks, cb = blockIdx.y % cblocks;
  const long c0 = (long)blockIdx.y * 8;
  const float* xs = x + c0 * S + v;
  float* ys = y ? y + (long)n * ystride + (long)cb * 8 * S + v : nullptr;
  unsigned short e[8][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float t = (xs[j * S] - mean[c0 + j]) * rstd[c0 + j];
    t = t > 0.f ? t : t * slope;
    if (ys) ys[j * S] = t;
    asm("" : "+v"(t));  // the scaling must see the ROUNDED activation (s3_common.hpp: no contraction into the split)
    h2_split(t * sc, e[j]);
  }
  const long ob = (long)n * oblocks + ob0 + cb;
#pragma unroll
  for (int t = 0; t < 2; ++t) out[(ob * 2 + t) * S + v] = s3_unit(e, t);
}

// k_act_split2h and k_maxpool2_h2 (below) in one pass: one thread per (8-channel block, POOLED voxel) normalises its 2 x 2 x 2 voxels, writes their
// eight H2 units and the unit of the winner (first maximum of a0 + a1 in scan order: k_maxpool2_h2's rule on the same terms) -- the inference
// forward's pooled blocks no longer read the full-resolution H2 tensor back (0.2 ms of an 11 ms cube).  Even extents; x read as float2.
__global__ void __launch_bounds__(256) k_act_split2h_pool(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float slope, uint4* __restrict__ out, uint4* __restrict__ pooled, int D, int H, int W,
                                                          int cblocks, int oblocks, int ob0, unsigned bound_bits, unsigned* __restrict__ cell) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && cell) *cell = bound_bits;
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long So = (long)Do * Ho * Wo, S = (long)D * H * W;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= So) return;
  const int xo = (int)(v % Wo), yo = (int)((v / Wo) % Ho), zo = (int)(v / ((long)Wo * Ho));
  const float sc = h2_scale(bound_bits);
  const int n = blockIdx.y / cblocks, cb = blockIdx.y % cblocks;
  const long c0 = (long)blockIdx.y * 8;
  const long ob = (long)n * oblocks + ob0 + cb;
  float m[8], r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m[j] = mean[c0 + j]; r[j] = rstd[c0 + j]; }
  float best[8];
  unsigned short w0[8], w1[8];
#pragma unroll
  for (int kp = 0; kp < 4; ++kp) {  // (z, y) of the pair; the pair itself is two consecutive x
    const long u = ((long)(2 * zo + (kp >> 1)) * H + (2 * yo + (kp & 1))) * W + 2 * xo;
    unsigned short e[2][8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2 xv = *reinterpret_cast<const float2*>(x + (c0 + j) * S + u);
      float t0 = (xv.x - m[j]) * r[j], t1 = (xv.y - m[j]) * r[j];
      t0 = t0 > 0.f ? t0 : t0 * slope;
      t1 = t1 > 0.f ? t1 : t1 * slope;
      asm("" : "+v"(t0), "+v"(t1));  // the scaling must see the ROUNDED activation (as k_act_split2h)
      h2_split(t0 * sc, e[0][j]);
      h2_split(t1 * sc, e[1][j]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 2; ++t) out[(ob * 2 + t) * S + u + h] = s3_unit(e[h], t);
      const int k = (kp >> 1) * 4 + (kp & 1) * 2 + h;  // k_maxpool2_h2's scan order: z, y, x
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float val = (float)__builtin_bit_cast(_Float16, e[h][j][0]) + (float)__builtin_bit_cast(_Float16, e[h][j][1]);
        if (val > best[j] || k == 0) { best[j] = val; w0[j] = e[h][j][0]; w1[j] = e[h][j][1]; }
      }
    }
  }
  uint4 o0, o1;
  o0.x = w0[0] | ((unsigned)w0[1] << 16); o0.y = w0[2] | ((unsigned)w0[3] << 16); o0.z = w0[4] | ((unsigned)w0[5] << 16); o0.w = w0[6] | ((unsigned)w0[7] << 16);
  o1.x = w1[0] | ((unsigned)w1[1] << 16); o1.y = w1[2] | ((unsigned)w1[3] << 16); o1.z = w1[4] | ((unsigned)w1[5] << 16); o1.w = w1[6] | ((unsigned)w1[7] << 16);
  uint4* o = pooled + ((long)n * cblocks + cb) * 2 * So;
  o[v] = o0;
  o[So + v] = o1;
}

// MaxPool3d(2) on an H2 tensor: the value of an element is (a0 + a1) / 2^k exactly, the larger value has the larger a0 + a1, and the winner's
// two terms ARE the pooled element's terms (same cell) -- so the pool needs no fp32 tensor on either side.  One thread per (8-channel block,
// output voxel): eight 16-byte units per term in, one out.  in: channels [0, 8 * cblocks) of a tensor with `iblocks` blocks per sample.
__global__ void __launch_bounds__(256) k_maxpool2_h2(const uint4* __restrict__ in, uint4* __restrict__ out, int cblocks, int iblocks, int D, int H,
                                                     int W) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long So = (long)Do * Ho * Wo, S = (long)D * H * W;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= So) return;
  const int n = blockIdx.y / cblocks, cb = blockIdx.y % cblocks;
  const int xo = (int)(v % Wo), yo = (int)((v / Wo) % Ho), zo = (int)(v / ((long)Wo * Ho));
  const uint4* i0 = in + ((long)n * iblocks + cb) * 2 * S;
  float best[8];
  unsigned short t0[8], t1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { best[j] = -3.0e38f; t0[j] = 0; t1[j] = 0; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long u = ((long)(2 * zo + (k >> 2)) * H + (2 * yo + ((k >> 1) & 1))) * W + 2 * xo + (k & 1);
    const uint4 a = i0[u], b = i0[S + u];
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned short h0 = (unsigned short)(aw[j >> 1] >> ((j & 1) * 16)), h1 = (unsigned short)(bw[j >> 1] >> ((j & 1) * 16));
      const float val = (float)__builtin_bit_cast(_Float16, h0) + (float)__builtin_bit_cast(_Float16, h1);
      // (first maximum in scan order, as nc_maxpool2_fwd; NaN never wins over a number, a plane of NaNs yields the first element's terms)
      if (val > best[j] || k == 0) { best[j] = val; t0[j] = h0; t1[j] = h1; }
    }
  }
  uint4 o0, o1;
  o0.x = t0[0] | ((unsigned)t0[1] << 16); o0.y = t0[2] | ((unsigned)t0[3] << 16); o0.z = t0[4] | ((unsigned)t0[5] << 16); o0.w = t0[6] | ((unsigned)t0[7] << 16);
  o1.x = t1[0] | ((unsigned)t1[1] << 16); o1.y = t1[2] | ((unsigned)t1[3] << 16); o1.z = t1[4] | ((unsigned)t1[5] << 16); o1.w = t1[6] | ((unsigned)t1[7] << 16);
  uint4* o = out + ((long)n * cblocks + cb) * 2 * So;
  o[v] = o0;
  o[So + v] = o1;
}

}  // namespace

// channels [0, C) of an H2 tensor with ctot channels per sample -> a dense H2 tensor of C channels at half the resolution (same cell)
int maxpool2_h2(const void* in, void* out, int N, int C, int ctot, int D, int H, int W, hipStream_t s) {
  if (C % 8 || ctot % 8 || (D | H | W) & 1) { set_error("maxpool2_h2: channels % 8, even extents"); return NC_ERR_SHAPE; }
  const long So = (long)(D / 2) * (H / 2) * (W / 2);
  hipLaunchKernelGGL(k_maxpool2_h2, dim3((unsigned)cdiv(So, 256), (unsigned)(N * C / 8)), dim3(256), 0, s, (const uint4*)in, (uint4*)out, C / 8, ctot / 8, D, H,
                     W);
  return check_launch("maxpool2_h2");
}

int h2_zero_cells(unsigned* cells, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_set_cells, dim3(1), dim3(64), 0, s, cells, n, 0u);
  return check_launch("h2_zero_cells");
}
int h2_set_cell(unsigned* cell, float bound, hipStream_t s) {
  unsigned bits;
  static_assert(sizeof(bits) == sizeof(bound), "");
  __builtin_memcpy(&bits, &bound, 4);
  hipLaunchKernelGGL(k_set_cells, dim3(1), dim3(64), 0, s, cell, 1, bits);
  return check_launch("h2_set_cell");
}
int h2_absmax(const float* x, long n, unsigned* cell, hipStream_t s, unsigned* cell2) {  // *cell (and *cell2) = max(itself, largest finite |x|)
  if ((unsigned long long)x & 3) { set_error("h2_absmax: the tensor must be 4-byte aligned"); return NC_ERR_ARG; }
  long blocks = cdiv(n, 256 * 4 * 8);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_absmax, dim3((unsigned)blocks), dim3(256), 0, s, x, n, cell, cell2);
  return check_launch("h2_absmax");
}
int split2h_into(const float* x, long xstride, void* xs, int N, int C, long S, int ctot, int c0, const unsigned* cell, hipStream_t s, unsigned* guard) {
  if (C % 8 || ctot % 8 || c0 % 8) { set_error("split2h: channels must be multiples of 8"); return NC_ERR_SHAPE; }
  long bx = cdiv(S, 256);
  const long ny = (long)N * C / 8;
  if (guard && h2_guard_on() && bx * ny > 4096) bx = cdiv(4096, ny) < bx ? cdiv(4096, ny) : bx;  // (see the kernel: few atomics)
  hipLaunchKernelGGL(k_split2h, dim3((unsigned)bx, (unsigned)ny), dim3(256), 0, s, x, (uint4*)xs, S, C / 8, ctot / 8, c0 / 8, xstride,
                     cell, h2_guard_on() ? guard : nullptr);
  return check_launch("split2h");
}

// ---- the range guard (common.hpp) -----------------------------------------------------------------------------------------------------------
static std::atomic<int> g_guard{getenv("NC_H2_GUARD") ? atoi(getenv("NC_H2_GUARD")) : 1};
int h2_guard_mode() { const int f = frozen_guard(); return f >= 0 ? f : g_guard.load(); }  // (inside a call: the value sampled when the call began)
bool h2_guard_on() { return h2_guard_mode() != 0; }
namespace { unsigned long long* guard_stats_dev(); }
// (the pinned counter block is allocated HERE, off the launch path -- a first allocation inside h2_guard_decide could land in a stream capture
// and invalidate it -- and portable, so that every device of the process sees it)
void h2_guard_set(int on) { g_guard = on == 2 ? 2 : on ? 1 : 0; if (on) (void)guard_stats_dev(); }
namespace {
std::mutex g_stats_mu;
unsigned long long* g_stats_host = nullptr;  // 4 counters in pinned host memory the device adds to (system-scope atomics): readable without a sync
unsigned long long* g_stats_dev = nullptr;
unsigned long long* guard_stats_dev() {
  std::lock_guard<std::mutex> lk(g_stats_mu);
  if (!g_stats_host) {
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    __builtin_memset(h, 0, 64);
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return nullptr; }
    g_stats_host = (unsigned long long*)h;
    g_stats_dev = (unsigned long long*)d;
  }
  return g_stats_dev;
}
}  // namespace
int h2_guard_read(unsigned long long* out4, int reset) {
  std::lock_guard<std::mutex> lk(g_stats_mu);
  for (int i = 0; i < 4; ++i) {
    // reset = one exchange: an increment a queued kernel lands between a load and a separate store would be lost
    if (!g_stats_host) out4[i] = 0;
    else out4[i] = reset ? __atomic_exchange_n(g_stats_host + i, 0ull, __ATOMIC_RELAXED) : __atomic_load_n(g_stats_host + i, __ATOMIC_RELAXED);
  }
  return NC_OK;
}
int h2_guard_zero(unsigned* g, hipStream_t s, int nwords) {
  hipLaunchKernelGGL(k_set_cells, dim3(1), dim3(64), 0, s, g, nwords, 0u);
  return check_launch("h2_guard_zero");
}
int h2_guard_decide(unsigned* ga, unsigned* gb, unsigned* prior, unsigned* flag, bool can_flip, hipStream_t s, unsigned long long total_a) {
  if (!h2_guard_on() && !prior) return NC_OK;  // (the words were zeroed: the flag reads 0)
  hipLaunchKernelGGL(k_h2_guard_decide, dim3(1), dim3(64), 0, s, ga, gb, prior, flag, can_flip ? 1 : 0, guard_stats_dev(), total_a);
  return check_launch("h2_guard_decide");
}
int h2_to_s3_if(const void* xh, void* xs, int N, int C, long S, const unsigned* cells, const unsigned* guard, hipStream_t s) {
  if (C % 8) { set_error("h2_to_s3: channels must be a multiple of 8"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_h2_to_s3_if, dim3((unsigned)cdiv(S, 256), (unsigned)(N * C / 8)), dim3(256), 0, s, (const uint4*)xh, (uint4*)xs, S, C / 8, cells, guard);
  return check_launch("h2_to_s3_if");
}
// bound: an upper bound of |result| known to the caller (InstanceNorm output: sqrt(S)); written to *cell (and *cell2) by the kernel itself
int act_split2h(const float* x, const float* mean, const float* rstd, float slope, float* y, long ystride, void* ys, int N, int C, long S, int ctot,
                int c0, float bound, unsigned* cell, unsigned* cell2, hipStream_t s) {
  if (C % 8 || ctot % 8 || c0 % 8) { set_error("act_split2h: channels must be multiples of 8"); return NC_ERR_SHAPE; }
  unsigned bits;
  __builtin_memcpy(&bits, &bound, 4);
  hipLaunchKernelGGL(k_act_split2h, dim3((unsigned)cdiv(S, 256), (unsigned)(N * C / 8)), dim3(256), 0, s, x, mean, rstd, slope, y, ystride,
                     (uint4*)ys, S, C / 8, ctot / 8, c0 / 8, bits, cell, cell2);
  return check_launch("act_split2h");
}

// act_split2h + maxpool2_h2 in one pass (no fp32 output): ys = channels [c0, c0 + C) of a ctot-channel H2 tensor, pooled = a dense C-channel H2
// tensor at half the resolution (same cell)
int act_split2h_pool(const float* x, const float* mean, const float* rstd, float slope, void* ys, void* pooled, int N, int C, int D, int H, int W,
                     int ctot, int c0, float bound, unsigned* cell, hipStream_t s) {
  if (C % 8 || ctot % 8 || c0 % 8 || ((D | H | W) & 1)) { set_error("act_split2h_pool: channels % 8, even extents"); return NC_ERR_SHAPE; }
  if ((((uintptr_t)x) & 7) != 0) { set_error("act_split2h_pool: input not 8-byte aligned"); return NC_ERR_ARG; }
  unsigned bits;
  __builtin_memcpy(&bits, &bound, 4);
  const long So = (long)(D / 2) * (H / 2) * (W / 2);
  hipLaunchKernelGGL(k_act_split2h_pool, dim3((unsigned)cdiv(So, 256), (unsigned)(N * C / 8)), dim3(256), 0, s, x, mean, rstd, slope, (uint4*)ys,
                     (uint4*)pooled, D, H, W, C / 8, ctot / 8, c0 / 8, bits, cell);
  return check_launch("act_split2h_pool");
}

}  // namespace nc
