// The three-term bf16 split of an fp32 value shared by the kernels that write the "S3" operand layout (conv_split.hip, conv_s3x.hip,
// norm_act.hip): a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = a - a0 - a1 (exact; DESIGN.md 4, "Split-operand fp32
// convolutions").
#pragma once
#include <hip/hip_runtime.h>

namespace nc {

__device__ __forceinline__ unsigned short s3_bf16_bits(float f) {
  const __bf16 v = (__bf16)f;
  return __builtin_bit_cast(unsigned short, v);
}
__device__ __forceinline__ float s3_bf16_val(float f) { return (float)(__bf16)f; }

// the three terms of an fp32 value (round to nearest each: the remainders are exact, the third term is exact)
__device__ __forceinline__ void s3_split(float v, unsigned short (&t)[3]) {
  // v is the ROUNDED fp32 value: a caller's multiply must not be fused into `v - a0` (the terms would then describe the unrounded
  // product, not the fp32 tensor element every other path sees).  hipcc contracts in the backend (-ffp-contract=fast), where no pragma
  // reaches: the empty asm makes v opaque at this point
  asm("" : "+v"(v));
  float a0 = s3_bf16_val(v);
  // a finite |v| above the largest finite bf16 (0x7F7F = 3.3895e38) rounds to infinity: take that largest bf16 instead, the remainders
  // carry the rest exactly.  v = +-inf / NaN: a0 = v and the remainders are NaN -- a non-finite input gives NaN in every output it touches
  if (__builtin_isinf(a0) && !__builtin_isinf(v)) a0 = __builtin_copysignf(3.3895313892515355e38f, v);
  const float r1 = v - a0;
  const float a1 = s3_bf16_val(r1);
  const float r2 = r1 - a1;
  t[0] = s3_bf16_bits(a0); t[1] = s3_bf16_bits(a1); t[2] = s3_bf16_bits(r2);
}

// 8 channels of one voxel, term t, as the 16-byte unit of the S3 layout
__device__ __forceinline__ uint4 s3_unit(const unsigned short (&e)[8][3], int t) {
  uint4 o;
  o.x = e[0][t] | ((unsigned)e[1][t] << 16); o.y = e[2][t] | ((unsigned)e[3][t] << 16);
  o.z = e[4][t] | ((unsigned)e[5][t] << 16); o.w = e[6][t] | ((unsigned)e[7][t] << 16);
  return o;
}

// ---- "H2": the two-term fp16 split of a SCALED fp32 value (conv_s3x.hip, NT = 2).  a * 2^k = a0 + a1 + r with a0 = fp16(a * 2^k), a1 = fp16(rest):
// 22-24 significant bits in two terms (fp16 carries 11), so THREE fp16 MFMA products a0 b0 + a0 b1 + a1 b0 make one fp32 product to ~2^-23 --
// where bf16 (8 bits a term) needs three terms and six products.  The price is fp16's range: k is chosen per TENSOR from its largest finite
// magnitude (one atomicMax cell; max |a| 2^k in [2^14, 2^15), so nothing overflows and elements down to 2^-17 of the largest keep both terms
// normal; below that the second term goes subnormal and the ABSOLUTE error stays <= 2^-25 / 2^k), and results are scaled back exactly.
__device__ __forceinline__ int h2_exp(unsigned amax_bits) {  // amax_bits: float bits of the largest finite |a| (0 for an all-zero tensor)
  const int e = (int)(amax_bits >> 23);
  const int k = e ? 14 - (e - 127) : 0;
  return k > 126 ? 126 : k;  // (e = 255 cannot happen: non-finite elements are left out of the maximum)
}
__device__ __forceinline__ float h2_scale(unsigned amax_bits) { return __uint_as_float((unsigned)(127 + h2_exp(amax_bits)) << 23); }
__device__ __forceinline__ float h2_inv_scale(unsigned amax_bits) { return __uint_as_float((unsigned)(127 - h2_exp(amax_bits)) << 23); }
// 2^-(ka + kb) as two factors of about equal exponent (f.x * f.y): applied one after the other they scale a sum back without an intermediate
// overflow or underflow that the result itself would not have
__device__ __forceinline__ float2 h2_unscale2(unsigned cell_a, unsigned cell_b) {
  const int e = -(h2_exp(cell_a) + h2_exp(cell_b));
  const int e1 = e / 2, e2 = e - e1;  // |e| <= 254: both within [-127, 127]
  float2 f;
  f.x = __uint_as_float((unsigned)(127 + (e1 < -126 ? -126 : e1)) << 23);
  f.y = __uint_as_float((unsigned)(127 + (e2 < -126 ? -126 : e2)) << 23);
  return f;
}
// 2^(kA - kB): what a value converted with cell B must be multiplied by to stand in a sum converted with cell A
__device__ __forceinline__ float h2_group_factor(unsigned cell_a, unsigned cell_b) {
  int d = h2_exp(cell_a) - h2_exp(cell_b);
  d = d < -126 ? -126 : d > 127 ? 127 : d;
  return __uint_as_float((unsigned)(127 + d) << 23);
}
// v = the element times h2_scale.  Non-finite v: a0 = v, a1 = NaN -- every output it touches becomes NaN, as with the three-term split.
__device__ __forceinline__ void h2_split(float v, unsigned short (&t)[3]) {
  asm("" : "+v"(v));
  const _Float16 a0 = (_Float16)v;
  const float r = v - (float)a0;
  const _Float16 a1 = (_Float16)r;
  t[0] = __builtin_bit_cast(unsigned short, a0); t[1] = __builtin_bit_cast(unsigned short, a1); t[2] = 0;
}

}  // namespace nc
