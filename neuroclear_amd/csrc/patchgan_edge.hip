// The PatchGAN's FIRST layer (reference models/networks.py:1030-1033: Conv2d(1 -> ndf, k 4, s 2, p 1) + LeakyReLU(0.2)) as
// what it is at Athena's batches (108-216 planes of 108^2 per discriminator pass): 16 MACs per output, i.e. pure HBM streams
// of the 64-channel activation (80-160 MB).  On the gather GEMM (K-dim = 16) + separate LeakyReLU passes the layer cost
// 0.27 ms forward, 0.33 ms weight gradient, 0.30 ms data gradient per pass (~5 % of the Athena step).  Here:
//   * k_pg1_fwd    : one thread = one output pixel: its 16 taps in registers, all K channels, bias and (optionally) the
//                    LeakyReLU in the same pass -- the whole-network call stores only the activation (the backward takes the
//                    mask from its sign: act > 0 <=> raw > 0);
//   * k_pg1_wgrad  : dW[k][tap] = sum g'[k][pixel] x[pixel + tap], db[k] = sum g'; a workgroup stages g' of 256 pixels x K
//                    channels and the pixels' taps in LDS, thread (k, tap group) walks the pixels; per-workgroup partials, fixed-
//                    order reduce;
//   * k_pg1_dgrad  : one thread = a 2 x 2 block of input pixels (the four parity classes share a 3 x 3 neighbourhood of dy).
// g' = g through the LeakyReLU mask of `act` when act != NULL (fused backward), g itself otherwise (op-by-op path: the same
// kernels, the same summation order, so both paths agree bit for bit).
#include "common.hpp"

namespace nc {
namespace {

constexpr int kMaxK = 64;

struct Pg1 {
  const float *x, *w, *bias, *g, *act;
  float* y;
  float* dx;
  double* part;
  float slope;
  int B, K, H, W, Ho, Wo;
  long npix;  // B * Ho * Wo
};

__global__ void __launch_bounds__(256) k_pg1_fwd(const Pg1 p) {
  __shared__ float ws[kMaxK * 17];
  for (int i = threadIdx.x; i < p.K * 17; i += 256) {
    const int k = i / 17, t = i - k * 17;
    ws[i] = t < 16 ? p.w[k * 16 + t] : (p.bias ? p.bias[k] : 0.f);
  }
  __syncthreads();
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= p.npix) return;
  const int So = p.Ho * p.Wo;
  const int b = (int)(j / So), pix = (int)(j - (long)b * So);
  const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
  const float* xb = p.x + (long)b * p.H * p.W;
  float t[16];
#pragma unroll
  for (int ty = 0; ty < 4; ++ty)
#pragma unroll
    for (int tx = 0; tx < 4; ++tx) {
      const int iy = 2 * oy - 1 + ty, ix = 2 * ox - 1 + tx;
      t[ty * 4 + tx] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? xb[(long)iy * p.W + ix] : 0.f;
    }
  float* yo = p.y + (long)b * p.K * So + pix;
  for (int k = 0; k < p.K; ++k) {
    const float* wk = ws + k * 17;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = fmaf(wk[i], t[i], acc);
    acc += wk[16];
    yo[(long)k * So] = acc > 0.f ? acc : acc * p.slope;
  }
}

// D[k][t] = sum over pixels of g'[k][pixel] * T[pixel][t] on the matrix cores (t = 16 taps + a column of ones for the bias
// gradient): a workgroup walks tiles of 128 consecutive pixels of the flat (plane, oy, ox) axis; per tile it stages
// g'[64][128] (pitch 129) and the taps T[18][128] (row 16 = ones, row 17 = zeros) in LDS, wave w multiplies pixels
// 32 w .. 32 w + 31 (two per MFMA: A = g'[k = lane][pixel + h], B = T[t = lane][pixel + h]); accumulators persist across the
// workgroup's tiles, the four waves' accumulators are added in wave order at the end -> one partial per workgroup.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kWgPix = 128, kGP = kWgPix + 1, kWgBlocks = 512;
__global__ void __launch_bounds__(256) k_pg1_wgrad(const Pg1 p) {
  __shared__ float gs[64 * kGP];
  __shared__ float ts[18 * kWgPix];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int So = p.Ho * p.Wo;
  const long ntiles = (p.npix + kWgPix - 1) / kWgPix;
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  for (int i = tid; i < kWgPix; i += 256) ts[17 * kWgPix + i] = 0.f;
  for (int i = tid; i < 64 * kGP; i += 256) gs[i] = 0.f;  // rows k >= K stay zero
  const float* a0p = gs + li * kGP + h;
  const float* a1p = gs + (32 + li) * kGP + h;
  const float* bp = ts + (li < 17 ? li : 17) * kWgPix + h;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's operand reads are done (and the zero fill above)
    // taps of pixel tid & 127: threads 0..127 rows 0..7, threads 128..255 rows 8..15 (+ the ones row)
    {
      const int q = tid & (kWgPix - 1), half = tid >> 7;
      const long j = tile * kWgPix + q;
      const bool valid = j < p.npix;
      const long jc = valid ? j : 0;
      const int b = (int)(jc / So), pix = (int)(jc - (long)b * So);
      const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
      const float* xb = p.x + (long)b * p.H * p.W;
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int t = half * 8 + tt, ty = t >> 2, tx = t & 3;
        const int iy = 2 * oy - 1 + ty, ix = 2 * ox - 1 + tx;
        ts[t * kWgPix + q] = (valid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? xb[(long)iy * p.W + ix] : 0.f;
      }
      if (half == 0) ts[16 * kWgPix + q] = valid ? 1.f : 0.f;
      // g' of this pixel: channels half, half + 2, ... (coalesced over q)
      const long go = (long)b * p.K * So + pix;
      for (int k = half; k < p.K; k += 2) {
        float v = 0.f;
        if (valid) {
          v = p.g[go + (long)k * So];
          if (p.act) v = p.act[go + (long)k * So] > 0.f ? v : v * p.slope;
        }
        gs[k * kGP + q] = v;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int q = wave * 32; q < wave * 32 + 32; q += 2) {
      const float bb = bp[q];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[q], bb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[q], bb, acc1, 0, 0, 0);
    }
  }
  // D rows = k: (e & 3) + 8 (e >> 2) + 4 h (+ 32 for acc1), column = t = li: the four waves' sums in wave order
  __syncthreads();
  float* red = gs;  // [4][64][17]
  if (li < 17) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
      red[(wave * 64 + k) * 17 + li] = acc0[e];
      red[(wave * 64 + 32 + k) * 17 + li] = acc1[e];
    }
  }
  __syncthreads();
  for (int o = tid; o < p.K * 17; o += 256) {
    const float v = ((red[o] + red[64 * 17 + o]) + red[2 * 64 * 17 + o]) + red[3 * 64 * 17 + o];
    p.part[(long)blockIdx.x * p.K * 17 + o] = (double)v;
  }
}

// dw[k][t] (t < 16) and db[k] (t == 16, nullable) = sum over workgroups, fixed order
__global__ void __launch_bounds__(256) k_pg1_wgrad_final(const double* __restrict__ part, int nblk, int K, float* __restrict__ dw,
                                                         float* __restrict__ db) {
  __shared__ double red[256];
  const int o = blockIdx.x;  // k * 17 + t
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += part[(long)i * K * 17 + o];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int k = o / 17, t = o - k * 17;
    if (t < 16) dw[k * 16 + t] = (float)red[0];
    else if (db) db[k] = (float)red[0];
  }
}

// one thread = input pixels (2 u + py, 2 v + px), py, px in {0, 1}: pixel iy receives tap ty from output row (iy + 1 - ty) / 2
// when that is an integer: iy = 2u -> ty in {1, 3} from oy in {u, u - 1}; iy = 2u + 1 -> ty in {0, 2} from oy in {u + 1, u}
__global__ void __launch_bounds__(256) k_pg1_dgrad(const Pg1 p) {
  __shared__ float ws[kMaxK * 16];
  for (int i = threadIdx.x; i < p.K * 16; i += 256) ws[i] = p.w[i];
  __syncthreads();
  const int Hu = (p.H + 1) / 2, Wu = (p.W + 1) / 2;
  const long nthr = (long)p.B * Hu * Wu;
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= nthr) return;
  const int b = (int)(j / ((long)Hu * Wu));
  const int r = (int)(j - (long)b * Hu * Wu);
  const int u = r / Wu, v = r - u * Wu;
  const int So = p.Ho * p.Wo;
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
  const long gb = (long)b * p.K * So;
  bool oky[3], okx[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    oky[d] = (unsigned)(u - 1 + d) < (unsigned)p.Ho;
    okx[d] = (unsigned)(v - 1 + d) < (unsigned)p.Wo;
  }
  for (int k = 0; k < p.K; ++k) {
    const float* wk = ws + k * 16;
    float n[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        float gv = 0.f;
        if (oky[dy] && okx[dx]) {
          const long o = gb + (long)k * So + (long)(u - 1 + dy) * p.Wo + (v - 1 + dx);
          gv = p.g[o];
          if (p.act) gv = p.act[o] > 0.f ? gv : gv * p.slope;
        }
        n[dy][dx] = gv;
      }
    // even row (py = 0): (ty 1, oy u) and (ty 3, oy u - 1); odd row (py = 1): (ty 0, oy u + 1) and (ty 2, oy u); same in x
    a00 = fmaf(wk[1 * 4 + 1], n[1][1], a00); a00 = fmaf(wk[1 * 4 + 3], n[1][0], a00);
    a00 = fmaf(wk[3 * 4 + 1], n[0][1], a00); a00 = fmaf(wk[3 * 4 + 3], n[0][0], a00);
    a01 = fmaf(wk[1 * 4 + 0], n[1][2], a01); a01 = fmaf(wk[1 * 4 + 2], n[1][1], a01);
    a01 = fmaf(wk[3 * 4 + 0], n[0][2], a01); a01 = fmaf(wk[3 * 4 + 2], n[0][1], a01);
    a10 = fmaf(wk[0 * 4 + 1], n[2][1], a10); a10 = fmaf(wk[0 * 4 + 3], n[2][0], a10);
    a10 = fmaf(wk[2 * 4 + 1], n[1][1], a10); a10 = fmaf(wk[2 * 4 + 3], n[1][0], a10);
    a11 = fmaf(wk[0 * 4 + 0], n[2][2], a11); a11 = fmaf(wk[0 * 4 + 2], n[2][1], a11);
    a11 = fmaf(wk[2 * 4 + 0], n[1][2], a11); a11 = fmaf(wk[2 * 4 + 2], n[1][1], a11);
  }
  float* xo = p.dx + (long)b * p.H * p.W;
  const int iy = 2 * u, ix = 2 * v;
  xo[(long)iy * p.W + ix] = a00;
  if (ix + 1 < p.W) xo[(long)iy * p.W + ix + 1] = a01;
  if (iy + 1 < p.H) {
    xo[(long)(iy + 1) * p.W + ix] = a10;
    if (ix + 1 < p.W) xo[(long)(iy + 1) * p.W + ix + 1] = a11;
  }
}

// The same data gradient for a FEW planes (Apollo: one to four planes of 108^2 per discriminator pass -- 46 workgroups of the kernel above, each
// thread walking all 64 channels: 105 us, half of the chain of small kernels the generators' backward waits for).  A workgroup = 32 pixel
// blocks x 8 channel groups: a thread sums its K / 8 channels, the eight partial sums of a pixel block are added through LDS in a fixed order.
__global__ void __launch_bounds__(256) k_pg1_dgrad_few(const Pg1 p) {
  __shared__ float ws[kMaxK * 16];
  __shared__ float red[4][8][32];
  for (int i = threadIdx.x; i < p.K * 16; i += 256) ws[i] = p.w[i];
  __syncthreads();
  const int Hu = (p.H + 1) / 2, Wu = (p.W + 1) / 2;
  const long nthr = (long)p.B * Hu * Wu;
  const int pl = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const long j = (long)blockIdx.x * 32 + pl;
  const bool in = j < nthr;
  const long jj = in ? j : 0;
  const int b = (int)(jj / ((long)Hu * Wu));
  const int r = (int)(jj - (long)b * Hu * Wu);
  const int u = r / Wu, v = r - u * Wu;
  const int So = p.Ho * p.Wo;
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
  const long gb = (long)b * p.K * So;
  bool oky[3], okx[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    oky[d] = in && (unsigned)(u - 1 + d) < (unsigned)p.Ho;
    okx[d] = (unsigned)(v - 1 + d) < (unsigned)p.Wo;
  }
  const int k0 = kg * p.K / 8, k1 = (kg + 1) * p.K / 8;
  for (int k = k0; k < k1; ++k) {
    const float* wk = ws + k * 16;
    float n[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        float gv = 0.f;
        if (oky[dy] && okx[dx]) {
          const long o = gb + (long)k * So + (long)(u - 1 + dy) * p.Wo + (v - 1 + dx);
          gv = p.g[o];
          if (p.act) gv = p.act[o] > 0.f ? gv : gv * p.slope;
        }
        n[dy][dx] = gv;
      }
    a00 = fmaf(wk[1 * 4 + 1], n[1][1], a00); a00 = fmaf(wk[1 * 4 + 3], n[1][0], a00);
    a00 = fmaf(wk[3 * 4 + 1], n[0][1], a00); a00 = fmaf(wk[3 * 4 + 3], n[0][0], a00);
    a01 = fmaf(wk[1 * 4 + 0], n[1][2], a01); a01 = fmaf(wk[1 * 4 + 2], n[1][1], a01);
    a01 = fmaf(wk[3 * 4 + 0], n[0][2], a01); a01 = fmaf(wk[3 * 4 + 2], n[0][1], a01);
    a10 = fmaf(wk[0 * 4 + 1], n[2][1], a10); a10 = fmaf(wk[0 * 4 + 3], n[2][0], a10);
    a10 = fmaf(wk[2 * 4 + 1], n[1][1], a10); a10 = fmaf(wk[2 * 4 + 3], n[1][0], a10);
    a11 = fmaf(wk[0 * 4 + 0], n[2][2], a11); a11 = fmaf(wk[0 * 4 + 2], n[2][1], a11);
    a11 = fmaf(wk[2 * 4 + 0], n[1][2], a11); a11 = fmaf(wk[2 * 4 + 2], n[1][1], a11);
  }
  red[0][kg][pl] = a00; red[1][kg][pl] = a01; red[2][kg][pl] = a10; red[3][kg][pl] = a11;
  __syncthreads();
  if (kg >= 4 || !in) return;  // channel group q < 4 adds up and stores output q of the pixel block
  float a = red[kg][0][pl];
#pragma unroll
  for (int q = 1; q < 8; ++q) a += red[kg][q][pl];
  float* xo = p.dx + (long)b * p.H * p.W;
  const int iy = 2 * u + (kg >> 1), ix = 2 * v + (kg & 1);
  if (iy < p.H && ix < p.W) xo[(long)iy * p.W + ix] = a;
}

Pg1 make(const ConvDims& d) {
  Pg1 p{};
  p.B = d.N; p.K = d.K; p.H = d.H; p.W = d.W; p.Ho = d.Ho; p.Wo = d.Wo;
  p.npix = (long)d.N * d.Ho * d.Wo;
  p.slope = 1.f;
  return p;
}

}  // namespace

bool pg1_supported(const ConvDims& d) {
  return d.C == 1 && d.D == 1 && d.kd == 1 && d.kh == 4 && d.kw == 4 && d.sh == 2 && d.sw == 2 && d.ph == 1 && d.pw == 1 &&
         d.K >= 1 && d.K <= kMaxK && (long)d.N * d.K * d.Ho * d.Wo < (1L << 31);
}
static int pg1_blocks(const ConvDims& d) {
  const long nt = cdiv((long)d.N * d.Ho * d.Wo, kWgPix);
  return (int)(nt < kWgBlocks ? nt : kWgBlocks);
}
size_t pg1_ws_bytes(const ConvDims& d) { return (size_t)pg1_blocks(d) * d.K * 17 * sizeof(double) + 256; }

// y = act(conv(x) + bias), act = LeakyReLU(slope) (slope 1: the plain convolution)
int conv_fwd_pg1(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, float slope, hipStream_t s) {
  Pg1 p = make(d);
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.slope = slope;
  hipLaunchKernelGGL(k_pg1_fwd, dim3((unsigned)cdiv(p.npix, 256)), dim3(256), 0, s, p);
  return check_launch("conv_fwd_pg1");
}

// g: gradient at the layer's output; act (nullable): the stored LeakyReLU output -- then g is the gradient BEHIND the
// activation and is pulled through its mask on the fly.  db nullable.
int conv_wgrad_pg1(const float* x, const float* g, const float* act, float slope, float* dw, float* db, const ConvDims& d,
                   void* ws, size_t wsb, hipStream_t s) {
  if (!ws || wsb < pg1_ws_bytes(d)) { set_error("conv_wgrad_pg1: workspace too small"); return NC_ERR_WS; }
  Pg1 p = make(d);
  p.x = x; p.g = g; p.act = act; p.slope = slope; p.part = (double*)ws;
  const int nblk = pg1_blocks(d);
  hipLaunchKernelGGL(k_pg1_wgrad, dim3(nblk), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_pg1_wgrad_final, dim3(d.K * 17), dim3(256), 0, s, (const double*)ws, nblk, d.K, dw, db);
  return check_launch("conv_wgrad_pg1");
}

int conv_dgrad_pg1(const float* g, const float* act, float slope, const float* w, float* dx, const ConvDims& d, hipStream_t s) {
  Pg1 p = make(d);
  p.g = g; p.act = act; p.slope = slope; p.w = w; p.dx = dx;
  const long nthr = (long)d.N * ((d.H + 1) / 2) * ((d.W + 1) / 2);
  static const int few = getenv("NC_PG1_FEW") ? atoi(getenv("NC_PG1_FEW")) : 1;  // A/B: 0 = the one-thread-per-pixel-block kernel at every size
  if (few && nthr < 128 * 256) hipLaunchKernelGGL(k_pg1_dgrad_few, dim3((unsigned)cdiv(nthr, 32)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_pg1_dgrad, dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, s, p);
  return check_launch("conv_dgrad_pg1");
}

}  // namespace nc
