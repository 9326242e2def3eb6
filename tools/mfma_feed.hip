// Micro-benchmark: how much of the fp16 16x16x32 MFMA rate survives when the B fragments come from LDS the way k_conv_s3x<.,.,2> reads them
// (two ds_read_b64 per term and column block, every read touching each bank once), as a function of how many MFMAs share a fragment pair:
//   mode 0: no LDS reads (operands in registers): the power / clock ceiling of this instruction on random data
//   mode 1: wave tile 32 channels x 128 positions  (2 row blocks): 4 reads per 6 MFMAs   -- k_conv_s3x<3, 8, 2> today
//   mode 2: wave tile 64 channels x 64 positions   (4 row blocks): 4 reads per 12 MFMAs  -- the k_conv_c8x tile with two terms
//   mode 3: mode 1 + the A fragments of every k-step fetched from global memory (4 x 1 KiB per wave, L2-resident)
//   mode 4: mode 2 + its A fragments (8 x 1 KiB per wave and k-step)
//   mode 5: mode 1 + the A fragments of every k-step read from LDS (4 x ds_read_b128 per wave: the weights-through-LDS design of conv_h.hip)
//   mode 6: ONE-term (the 16-bit path of configs[3], k_conv_c8x's tile): 64 channels x 128 positions per wave, 2 reads per 4 MFMAs, 4 waves x 2
//           workgroups per CU are modelled as 8 waves; A fragments (4 x 1 KiB per k-step and wave) from global memory
//   mode 7: mode 6 with the A fragments in registers (no global loads): what the LDS reads alone cost the one-term kernel
// 256 workgroups of 8 waves (one per CU, two waves per SIMD), 24 KiB of random fp16 in LDS per workgroup.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_feed.hip -o gpurun_out/mfma_feed   Run: ./mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) unsigned long long* lds64_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ void __launch_bounds__(512, 1) k(const uint4* __restrict__ in, float* __restrict__ out, int ksteps) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr bool ONE = MODE >= 6;
  constexpr int RB = (MODE == 2 || MODE == 4 || ONE) ? 4 : 2, NCB = ONE ? 8 : RB == 4 ? 4 : 8;
  constexpr bool AG = MODE == 3 || MODE == 4 || MODE == 6, AL = MODE == 5, BL = MODE != 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 24576 / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = in[(blockIdx.x * 1536 + i) & 65535];
  __syncthreads();
  u32x4 A[RB][2];
  for (int r = 0; r < RB; ++r) for (int t = 0; t < 2; ++t) A[r][t] = __builtin_bit_cast(u32x4, in[(tid + (r * 2 + t) * 4096) & 65535]);
  f32x4 acc[RB][NCB];
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned base = (unsigned)(unsigned long long)(lptr_t)lds;
  const unsigned lane_b = (unsigned)((((wave >> 1) * NCB * 16 + m16) * 16) + (g & 1) * 8);
  u32x4 B[2][2];
  auto read_b = [&](u32x4 (&Bd)[2], unsigned vo, int cb) {
    for (int t = 0; t < (ONE ? 1 : 2); ++t) {
      u64x2 v;
      const unsigned lo = base + vo + t * 12288 + cb * 256;
      v.x = *(lds64_t)(lo); v.y = *(lds64_t)(lo ^ 8u);
      Bd[t] = __builtin_bit_cast(u32x4, v);
    }
  };
  int tap = g;
  for (int s = 0; s < ksteps; ++s) {
    if (AG) {
      for (int r = 0; r < RB; ++r) for (int t = 0; t < 2; ++t)
        A[r][t] = __builtin_bit_cast(u32x4, in[((s & 63) * 64 + (r * 2 + t) * 4096 + lane + (wave & 1) * 8192) & 65535]);
    }
    if (AL) {
      typedef const volatile __attribute__((address_space(3))) u32x4* lds128_t;
      for (int r = 0; r < RB; ++r) for (int t = 0; t < 2; ++t)
        A[r][t] = *(lds128_t)(base + (unsigned)((((s & 3) * 4 + r * 2 + t) * 64 + lane) * 16));
    }
    const unsigned vo = lane_b + (unsigned)(((tap % 3) * 110 + tap / 3) * 16);
    tap = (tap + 4) % 9;
    if (BL) read_b(B[0], vo, 0);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      u32x4(&Bc)[2] = B[cb & 1];
      u32x4(&Bn)[2] = B[(cb + 1) & 1];
      if (BL && cb + 1 < NCB) read_b(Bn, vo, cb + 1);
      if (!BL) { Bc[0] = A[0][0]; Bc[1] = A[0][1]; }
      constexpr int TA[3] = {ONE ? 0 : 1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
      for (int m = 0; m < (ONE ? 1 : 3); ++m)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[r][TA[m]]), __builtin_bit_cast(f16x8, Bc[TB[m]]), acc[r][cb], 0, 0, 0);
    }
  }
  float sum = 0.f;
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) for (int e = 0; e < 4; ++e) sum += acc[r][c][e];
  out[blockIdx.x * 512 + tid] = sum;
}

template <int MODE>
void run(const char* name, const uint4* in, float* out) {
  const int ksteps = 4000, grid = 256;
  constexpr bool ONE = MODE >= 6;
  constexpr int RB = (MODE == 2 || MODE == 4 || ONE) ? 4 : 2, NCB = ONE ? 8 : RB == 4 ? 4 : 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 24576, 0, in, out, ksteps);
  hipEventRecord(e0);
  const int reps = 8;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 24576, 0, in, out, ksteps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * 16 * 16 * 32 * (ONE ? 1 : 3) * RB * NCB * (double)ksteps * grid * 8;
  printf("{\"mode\": \"%s\", \"ms\": %.3f, \"mfma_tflops\": %.0f, \"fp32_equiv_tflops\": %.0f}\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / 3 / (ms * 1e-3) / 1e12);
}

int main() {
  std::vector<unsigned short> h(65536 * 8 + 64);
  srand(1);
  for (auto& v : h) { const _Float16 f = (_Float16)((rand() % 2001 - 1000) / 500.0f); __builtin_memcpy(&v, &f, 2); }
  uint4* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("0 registers only", in, out);
    run<1>("1 32ch x 128pos, B from LDS (4 reads / 6 MFMA)", in, out);
    run<2>("2 64ch x 64pos, B from LDS (4 reads / 12 MFMA)", in, out);
    run<3>("3 mode 1 + A from global (4 KiB / k-step / wave)", in, out);
    run<4>("4 mode 2 + A from global (8 KiB / k-step / wave)", in, out);
    run<5>("5 mode 1 + A from LDS (4 KiB / k-step / wave)", in, out);
    run<6>("6 ONE term, 64ch x 128pos, B from LDS (2 reads / 4 MFMA), A from global", in, out);
    run<7>("7 ONE term, 64ch x 128pos, B from LDS, A in registers", in, out);
  }
  return 0;
}
