// Dice / assemble on the device (reference data/diceImage_dataset.py:95-120, data/base_dataset.py:134-143,
// util/util.py:196-215, util/assemble_dice.py:130-213).  Integer / index work plus three fp32 roundings whose order
// is kept exactly as numpy does them, so the result is bit-identical to the reference on one GPU:
//   cube   = float32( double(v) / 65535.0 )                 (base_dataset.py:134-143, :291-295)
//   acc   += cube / 8   in cube-index order                  (assemble_dice.py:167-173)
//   out    = uint16( ((acc / cnt) * 8) * 65535 )  truncating (assemble_dice.py:183, :202-207)
// The volume stays resident in HBM as uint16/uint8; the zero dicing pad and the reflect border are index arithmetic.
#include "common.hpp"

namespace nc {

__device__ __forceinline__ int reflect_idx(int q, int P) {  // np.pad(mode='reflect') for |overhang| < P
  if (q < 0) q = -q;
  if (q >= P) q = 2 * (P - 1) - q;
  return q;
}

struct DiceGeom {
  int L0, L1, L2, P0, P1, P2, n0, n1, n2, roi, overlap, border, step;
};

__host__ __device__ inline int pad_len(int L, int roi, int overlap) {
  const int step = roi - overlap;
  return step * ((L + overlap) / step) + roi;  // L + pad
}

// ---- "clean rotation" training augmentation on the device (reference data/base_dataset.py:306-460): every z-slice is
//      rotated about its centre (cv2.warpAffine, bilinear, zero border) and cropped to the inscribed rectangle; the
//      reference rotates the WHOLE volume on the host for every training crop.  Here only the voxels of the requested
//      crop are produced: out[z][y][x] = warpAffine-bilinear(vol[z0 + z], inv * (x0 + x, y0 + y, 1)), rounded to the source integer
//      type and normalised (/65535 or /255 in fp64, then fp32) exactly like the crop-only path.
// The sampling arithmetic is cv2.warpAffine's (flags = INTER_LINEAR, BORDER_CONSTANT 0) as OpenCV 4.5.0 publishes it
// (modules/imgproc/src/imgwarp.cpp: WarpAffineInvoker + remapBilinear + initInterTab2D): m = the canvas -> source matrix warpAffine
// derives from the forward one (host: data/rotation.py).  Coordinates are fixed point: cvRound of the column term and of the row term at
// 2^-10 each, + 2^4, >> 5 -> 5 fractional bits; uint8 blends with the 15-bit integer weights 32 (32 - fy)(32 - fx) ... and a rounding
// shift, uint16 with the float weights (1 - fy/32)(1 - fx/32) ... in float32, products and sums left to right, then cvRound + saturation.
template <typename T>
__global__ void k_rotate_crop(const T* __restrict__ vol, int H, int W, int z0, int y0, int x0, int cz, int cy, int cx,
                              double m00, double m01, double m02, double m10, double m11, double m12, double den,
                              double vmax, float* __restrict__ out) {
  const long total = (long)cz * cy * cx;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % cx), y = (int)((i / cx) % cy), z = (int)(i / ((long)cx * cy));
    const double dx = (double)(x0 + x), dy = (double)(y0 + y);
    const int ad = (int)rint(m00 * dx * 1024.0), bd = (int)rint(m10 * dx * 1024.0);
    const int X0 = (int)rint((m01 * dy + m02) * 1024.0) + 16, Y0 = (int)rint((m11 * dy + m12) * 1024.0) + 16;
    const int X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;  // arithmetic shifts: floor for negative coordinates, as in OpenCV
    const int ix = X >> 5, iy = Y >> 5, fx = X & 31, fy = Y & 31;
    const T* sl = vol + (long)(z0 + z) * H * W;
    auto tap = [&](int yy, int xx) -> unsigned {
      return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (unsigned)sl[(long)yy * W + xx] : 0u;
    };
    const unsigned v00 = tap(iy, ix), v01 = tap(iy, ix + 1), v10 = tap(iy + 1, ix), v11 = tap(iy + 1, ix + 1);
    double v;
    if (sizeof(T) == 1) {
      const int w00 = 32 * (32 - fy) * (32 - fx), w01 = 32 * (32 - fy) * fx, w10 = 32 * fy * (32 - fx), w11 = 32 * fy * fx;
      const int r = (int)(v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
      v = (double)(r < 0 ? 0 : r > 255 ? 255 : r);
    } else {
      const float cy0 = 1.f - (float)fy * 0.03125f, cy1 = (float)fy * 0.03125f, cx0 = 1.f - (float)fx * 0.03125f, cx1 = (float)fx * 0.03125f;
      float r = __fmul_rn((float)v00, __fmul_rn(cy0, cx0));
      r = __fadd_rn(r, __fmul_rn((float)v01, __fmul_rn(cy0, cx1)));
      r = __fadd_rn(r, __fmul_rn((float)v10, __fmul_rn(cy1, cx0)));
      r = __fadd_rn(r, __fmul_rn((float)v11, __fmul_rn(cy1, cx1)));
      v = (double)rintf(r);
      v = v < 0.0 ? 0.0 : (v > vmax ? vmax : v);
    }
    out[i] = (float)(v / den);
  }
}

template <typename T>
__global__ void k_cut_cube(const T* __restrict__ vol, DiceGeom g, int z0, int y0, int x0, double den,
                           float* __restrict__ cube) {
  const int E = g.roi + 2 * g.border;
  const long total = (long)E * E * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cx = (int)(i % E), cy = (int)((i / E) % E), cz = (int)(i / ((long)E * E));
    const int qz = reflect_idx(z0 + cz - g.border, g.P0);
    const int qy = reflect_idx(y0 + cy - g.border, g.P1);
    const int qx = reflect_idx(x0 + cx - g.border, g.P2);
    float v = 0.f;
    if (qz < g.L0 && qy < g.L1 && qx < g.L2) v = (float)((double)vol[((long)qz * g.L1 + qy) * g.L2 + qx] / den);
    cube[i] = v;
  }
}

__global__ void k_scatter_add(const float* __restrict__ cube, float* __restrict__ acc, DiceGeom g, int z0, int y0,
                              int x0) {
  const int R = g.roi, E = g.roi + 2 * g.border;
  const long total = (long)R * R * R;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % R), y = (int)((i / R) % R), z = (int)(i / ((long)R * R));
    const float c = cube[((long)(z + g.border) * E + (y + g.border)) * E + x + g.border];
    float* a = acc + ((long)(z0 + z) * g.P1 + (y0 + y)) * g.P2 + (x0 + x);
    *a = *a + c / 8;
  }
}

__device__ __forceinline__ int cover_count(int p, int n, int step, int roi) {
  // number of cube indices i in [0, n) with i*step <= p < i*step + roi
  int hi = p / step;
  if (hi > n - 1) hi = n - 1;
  int lo = p - roi + 1;
  lo = lo <= 0 ? 0 : (lo + step - 1) / step;
  return hi - lo + 1;
}

template <typename T>
__global__ void k_finalize(const float* __restrict__ acc, T* __restrict__ out, DiceGeom g, float scale) {
  const long total = (long)g.L0 * g.L1 * g.L2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % g.L2), y = (int)((i / g.L2) % g.L1), z = (int)(i / ((long)g.L2 * g.L1));
    float v = acc[((long)z * g.P1 + y) * g.P2 + x];
    if (g.overlap > 0) {
      const float cnt = (float)(cover_count(z, g.n0, g.step, g.roi) * cover_count(y, g.n1, g.step, g.roi) *
                                cover_count(x, g.n2, g.step, g.roi));
      v = (v / cnt) * 8;
    }
    v = v * scale;
    out[i] = (T)v;  // truncation toward zero == ndarray.astype for in-range values
  }
}

// The same for a z-slab: planes [z0, z0 + nz) of the (un-padded) volume from an accumulator that holds the padded planes from `za`
// on (multi-GPU inference: every rank finalises the slab it owns, test_dice.py assemble='slab').
template <typename T>
__global__ void k_finalize_slab(const float* __restrict__ acc, T* __restrict__ out, DiceGeom g, float scale, int z0, int nz, int za) {
  const long total = (long)nz * g.L1 * g.L2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % g.L2), y = (int)((i / g.L2) % g.L1), z = z0 + (int)(i / ((long)g.L2 * g.L1));
    float v = acc[((long)(z - za) * g.P1 + y) * g.P2 + x];
    if (g.overlap > 0) {
      const float cnt = (float)(cover_count(z, g.n0, g.step, g.roi) * cover_count(y, g.n1, g.step, g.roi) *
                                cover_count(x, g.n2, g.step, g.roi));
      v = (v / cnt) * 8;
    }
    v = v * scale;
    out[i] = (T)v;
  }
}

// ---- intensity normalisation of the assembled volume (reference util/assemble_dice.py:188-192: np.percentile over the
//      merged, still padded volume, then skimage.exposure.rescale_intensity(in_range=(p_lo, p_hi))).
//      k_merge: merged = (acc / count) * 8 over the padded volume.  k_radix_hist: one pass of an exact radix select on
//      the order-preserving integer key of a float (12 + 12 + 8 bits; the host picks the bucket between passes):
//      workgroup-private LDS histogram, integer atomics only -- the counts, hence the selected order statistic, are
//      exact and deterministic.  k_rescale_cast: clip to [lo, hi], (v - lo) / (hi - lo) mapped to [omin, 1], scale,
//      truncating cast, crop of the dicing pad.
__global__ void k_merge(const float* __restrict__ acc, float* __restrict__ out, DiceGeom g) {
  const long total = (long)g.P0 * g.P1 * g.P2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % g.P2), y = (int)((i / g.P2) % g.P1), z = (int)(i / ((long)g.P2 * g.P1));
    const float cnt = (float)(cover_count(z, g.n0, g.step, g.roi) * cover_count(y, g.n1, g.step, g.roi) *
                              cover_count(x, g.n2, g.step, g.roi));
    out[i] = (acc[i] / cnt) * 8;
  }
}

__device__ __forceinline__ unsigned float_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// pass 0: bins = key >> 20 (4096); pass 1: (key >> 8) & 0xfff among keys with key >> 20 == prefix; pass 2: key & 0xff
// among keys with key >> 8 == prefix
__global__ void k_radix_hist(const float* __restrict__ x, long n, int pass, unsigned prefix, unsigned* __restrict__ hist) {
  __shared__ unsigned h[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned k = float_key(x[i]);
    if (pass == 0) atomicAdd(&h[k >> 20], 1u);
    else if (pass == 1) { if ((k >> 20) == prefix) atomicAdd(&h[(k >> 8) & 0xfffu], 1u); }
    else { if ((k >> 8) == prefix) atomicAdd(&h[k & 0xffu], 1u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

template <typename T>
__global__ void k_rescale_cast(const float* __restrict__ merged, T* __restrict__ out, DiceGeom g, float lo, float hi,
                               float range, float omin, float scale) {
  const long total = (long)g.L0 * g.L1 * g.L2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % g.L2), y = (int)((i / g.L2) % g.L1), z = (int)(i / ((long)g.L2 * g.L1));
    float v = merged[((long)z * g.P1 + y) * g.P2 + x];
    v = fminf(fmaxf(v, lo), hi);
    v = (v - lo) / range;
    v = v * (1.f - omin) + omin;
    v = v * scale;
    out[i] = (T)v;
  }
}

static bool make_geom(DiceGeom& g, int L0, int L1, int L2, int roi, int overlap, int border) {
  if (L0 < 1 || L1 < 1 || L2 < 1 || roi < 1 || overlap < 0 || overlap >= roi || border < 1) return false;
  g.L0 = L0; g.L1 = L1; g.L2 = L2; g.roi = roi; g.overlap = overlap; g.border = border; g.step = roi - overlap;
  g.P0 = pad_len(L0, roi, overlap); g.P1 = pad_len(L1, roi, overlap); g.P2 = pad_len(L2, roi, overlap);
  g.n0 = (g.P0 - overlap) / g.step; g.n1 = (g.P1 - overlap) / g.step; g.n2 = (g.P2 - overlap) / g.step;
  // np.pad(mode='reflect') with border >= padded length is multi-fold; the reference never gets there
  return border < g.P0 && border < g.P1 && border < g.P2;
}

static unsigned flat_grid(long n) {
  long b = cdiv(n, 256);
  if (b > 8192) b = 8192;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace nc

using namespace nc;

extern "C" {

int nc_dice_cut_cube(const void* vol, int is_u16, int L0, int L1, int L2, int roi, int overlap, int border, int index,
                     float* cube, void* stream) {
  if (!vol || !cube) { set_error("dice_cut_cube: null pointer"); return NC_ERR_ARG; }
  DiceGeom g;
  if (!make_geom(g, L0, L1, L2, roi, overlap, border)) { set_error("dice_cut_cube: bad geometry (border_cut must be >= 1)"); return NC_ERR_SHAPE; }
  const long n = (long)g.n0 * g.n1 * g.n2;
  if (index < 0 || index >= n) { set_error("dice_cut_cube: cube index %d out of [0,%ld)", index, n); return NC_ERR_SHAPE; }
  const int xi = index % g.n2, yi = (index % (g.n2 * g.n1)) / g.n2, zi = index / (g.n2 * g.n1);
  const long E = roi + 2 * border;
  hipStream_t s = (hipStream_t)stream;
  if (is_u16)
    hipLaunchKernelGGL(k_cut_cube<uint16_t>, dim3(flat_grid(E * E * E)), dim3(256), 0, s, (const uint16_t*)vol, g,
                       zi * g.step, yi * g.step, xi * g.step, 65535.0, cube);
  else
    hipLaunchKernelGGL(k_cut_cube<uint8_t>, dim3(flat_grid(E * E * E)), dim3(256), 0, s, (const uint8_t*)vol, g,
                       zi * g.step, yi * g.step, xi * g.step, 255.0, cube);
  return check_launch("dice_cut_cube");
}

int nc_rotate_crop(const void* vol, int is_u16, int D, int H, int W, int z0, int y0, int x0, int cz, int cy, int cx,
                   const double* inv_affine, float* out, void* stream) {
  if (!vol || !out || !inv_affine) { set_error("rotate_crop: null pointer"); return NC_ERR_ARG; }
  if (D < 1 || H < 1 || W < 1 || cz < 1 || cy < 1 || cx < 1 || z0 < 0 || z0 + cz > D) {
    set_error("rotate_crop: bad shape D=%d H=%d W=%d crop=(%d,%d,%d) z0=%d", D, H, W, cz, cy, cx, z0);
    return NC_ERR_SHAPE;
  }
  const double* m = inv_affine;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)cz * cy * cx;
  if (is_u16)
    hipLaunchKernelGGL(k_rotate_crop<uint16_t>, dim3(flat_grid(total)), dim3(256), 0, s, (const uint16_t*)vol, H, W, z0,
                       y0, x0, cz, cy, cx, m[0], m[1], m[2], m[3], m[4], m[5], 65535.0, 65535.0, out);
  else
    hipLaunchKernelGGL(k_rotate_crop<uint8_t>, dim3(flat_grid(total)), dim3(256), 0, s, (const uint8_t*)vol, H, W, z0,
                       y0, x0, cz, cy, cx, m[0], m[1], m[2], m[3], m[4], m[5], 255.0, 255.0, out);
  return check_launch("rotate_crop");
}

int nc_assemble_scatter_add(const float* cube, float* acc, int P0, int P1, int P2, int roi, int overlap, int border,
                            int index, void* stream) {
  if (!cube || !acc) { set_error("assemble_scatter_add: null pointer"); return NC_ERR_ARG; }
  if (overlap < 1) { set_error("assemble_scatter_add: overlap must be >= 1 (the reference assembler adds nothing for overlap 0, util/assemble_dice.py:170)"); return NC_ERR_SHAPE; }
  if (roi < 1 || overlap >= roi || border < 1) { set_error("assemble_scatter_add: bad geometry (border_cut must be >= 1)"); return NC_ERR_SHAPE; }
  DiceGeom g{};
  g.roi = roi; g.overlap = overlap; g.border = border; g.step = roi - overlap;
  g.P0 = P0; g.P1 = P1; g.P2 = P2;
  g.n0 = (P0 - overlap) / g.step; g.n1 = (P1 - overlap) / g.step; g.n2 = (P2 - overlap) / g.step;
  const long n = (long)g.n0 * g.n1 * g.n2;
  if (g.n0 < 1 || g.n1 < 1 || g.n2 < 1 || index < 0 || index >= n) { set_error("assemble_scatter_add: cube index out of range"); return NC_ERR_SHAPE; }
  const int xi = index % g.n2, yi = (index % (g.n2 * g.n1)) / g.n2, zi = index / (g.n2 * g.n1);
  if (zi * g.step + roi > P0 || yi * g.step + roi > P1 || xi * g.step + roi > P2) { set_error("assemble_scatter_add: cube outside the volume"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_scatter_add, dim3(flat_grid((long)roi * roi * roi)), dim3(256), 0, (hipStream_t)stream, cube, acc,
                     g, zi * g.step, yi * g.step, xi * g.step);
  return check_launch("assemble_scatter_add");
}

int nc_assemble_finalize(const float* acc, void* out, int out_is_u16, int P0, int P1, int P2, int L0, int L1, int L2,
                         int roi, int overlap, void* stream) {
  if (!acc || !out) { set_error("assemble_finalize: null pointer"); return NC_ERR_ARG; }
  if (overlap < 1) { set_error("assemble_finalize: overlap must be >= 1 (util/assemble_dice.py:170)"); return NC_ERR_SHAPE; }
  DiceGeom g;
  if (!make_geom(g, L0, L1, L2, roi, overlap, 1) || g.P0 != P0 || g.P1 != P1 || g.P2 != P2) {
    set_error("assemble_finalize: padded size does not match pad_for_dicing of the original size");
    return NC_ERR_SHAPE;
  }
  const long total = (long)L0 * L1 * L2;
  hipStream_t s = (hipStream_t)stream;
  if (out_is_u16)
    hipLaunchKernelGGL(k_finalize<uint16_t>, dim3(flat_grid(total)), dim3(256), 0, s, acc, (uint16_t*)out, g, 65535.f);
  else
    hipLaunchKernelGGL(k_finalize<uint8_t>, dim3(flat_grid(total)), dim3(256), 0, s, acc, (uint8_t*)out, g, 255.f);
  return check_launch("assemble_finalize");
}

int nc_assemble_finalize_slab(const float* acc_slab, void* out, int out_is_u16, int P0, int P1, int P2, int L0, int L1, int L2, int roi,
                              int overlap, int z0, int nz, int za, void* stream) {
  if (!acc_slab || !out) { set_error("assemble_finalize_slab: null pointer"); return NC_ERR_ARG; }
  if (overlap < 1) { set_error("assemble_finalize_slab: overlap must be >= 1 (util/assemble_dice.py:170)"); return NC_ERR_SHAPE; }
  DiceGeom g;
  if (!make_geom(g, L0, L1, L2, roi, overlap, 1) || g.P0 != P0 || g.P1 != P1 || g.P2 != P2) {
    set_error("assemble_finalize_slab: padded size does not match pad_for_dicing of the original size");
    return NC_ERR_SHAPE;
  }
  if (z0 < 0 || nz < 1 || z0 + nz > L0 || za < 0 || za > z0) { set_error("assemble_finalize_slab: bad slab"); return NC_ERR_SHAPE; }
  const long total = (long)nz * L1 * L2;
  hipStream_t s = (hipStream_t)stream;
  if (out_is_u16)
    hipLaunchKernelGGL(k_finalize_slab<uint16_t>, dim3(flat_grid(total)), dim3(256), 0, s, acc_slab, (uint16_t*)out, g, 65535.f, z0, nz, za);
  else
    hipLaunchKernelGGL(k_finalize_slab<uint8_t>, dim3(flat_grid(total)), dim3(256), 0, s, acc_slab, (uint8_t*)out, g, 255.f, z0, nz, za);
  return check_launch("assemble_finalize_slab");
}

int nc_assemble_merge(const float* acc, float* merged, int L0, int L1, int L2, int roi, int overlap, void* stream) {
  if (!acc || !merged) { set_error("assemble_merge: null pointer"); return NC_ERR_ARG; }
  DiceGeom g;
  if (overlap < 1 || !make_geom(g, L0, L1, L2, roi, overlap, 1)) { set_error("assemble_merge: bad geometry"); return NC_ERR_SHAPE; }
  hipLaunchKernelGGL(k_merge, dim3(flat_grid((long)g.P0 * g.P1 * g.P2)), dim3(256), 0, (hipStream_t)stream, acc, merged, g);
  return check_launch("assemble_merge");
}

int nc_radix_hist(const float* x, long n, int pass, unsigned prefix, unsigned* hist4096, void* stream) {
  if (!x || !hist4096 || n < 1 || pass < 0 || pass > 2) { set_error("radix_hist: bad argument"); return NC_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(hist4096, 0, 4096 * sizeof(unsigned), s) != hipSuccess) { set_error("radix_hist: memset failed"); return NC_ERR_HIP; }
  hipLaunchKernelGGL(k_radix_hist, dim3(flat_grid(n)), dim3(256), 0, s, x, n, pass, prefix, hist4096);
  return check_launch("radix_hist");
}

int nc_assemble_rescale_finalize(const float* merged, void* out, int out_is_u16, int L0, int L1, int L2, int roi,
                                 int overlap, float lo, float hi, float range, void* stream) {
  if (!merged || !out) { set_error("assemble_rescale_finalize: null pointer"); return NC_ERR_ARG; }
  DiceGeom g;
  if (overlap < 1 || !make_geom(g, L0, L1, L2, roi, overlap, 1) || !(hi > lo) || !(range > 0.f)) {
    set_error("assemble_rescale_finalize: bad geometry or empty intensity range");
    return NC_ERR_SHAPE;
  }
  const float omin = lo >= 0.f ? 0.f : -1.f;  // skimage: float images map to (0, 1), or (-1, 1) if in_range[0] < 0
  const long total = (long)L0 * L1 * L2;
  hipStream_t s = (hipStream_t)stream;
  if (out_is_u16)
    hipLaunchKernelGGL(k_rescale_cast<uint16_t>, dim3(flat_grid(total)), dim3(256), 0, s, merged, (uint16_t*)out, g, lo,
                       hi, range, omin, 65535.f);
  else
    hipLaunchKernelGGL(k_rescale_cast<uint8_t>, dim3(flat_grid(total)), dim3(256), 0, s, merged, (uint8_t*)out, g, lo, hi,
                       range, omin, 255.f);
  return check_launch("assemble_rescale_finalize");
}

}  // extern "C"
