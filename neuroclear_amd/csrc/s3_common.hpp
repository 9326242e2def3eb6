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

}  // namespace nc
