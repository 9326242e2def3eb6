// Micro-benchmark: cycles and wall time per bf16 MFMA instruction shape on gfx950, operands in registers, random data.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o gpurun_out/mfma_rate   Run: ./mfma_rate
// Question it answers: does the k = 8 form (v_mfma_f32_32x32x8_bf16_1k) take half the cycles of the k = 16 form on this chip
// (then the 9th tap of a 3x3 plane can be a half-length k-step), and what clock does each shape hold.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) k(const uint4* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  uint4 a0 = in[tid], b0 = in[tid + 65536], a1 = in[tid + 2 * 65536], b1 = in[tid + 3 * 65536];
  f32x16 acc[4];
  f32x4 acs[8];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) acs[j][e] = 0.f;
  const bf16x8 A0 = __builtin_bit_cast(bf16x8, a0), B0 = __builtin_bit_cast(bf16x8, b0);
  const bf16x8 A1 = __builtin_bit_cast(bf16x8, a1), B1 = __builtin_bit_cast(bf16x8, b1);
  const s16x4 a4 = {(short)a0.x, (short)a0.y, (short)a0.z, (short)a0.w};
  const s16x4 b4 = {(short)b0.x, (short)b0.y, (short)b0.z, (short)b0.w};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 x 32x32x16
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(j & 1 ? A1 : A0, j & 2 ? B1 : B0, acc[j & 3], 0, 0, 0);
    } else if (MODE == 1) {  // 8 x 32x32x8 (1k)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[j & 3], 0, 0, 0);
    } else if (MODE == 2) {  // 8 x 16x16x32
#pragma unroll
      for (int j = 0; j < 8; ++j) acs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j & 1 ? A1 : A0, j & 2 ? B1 : B0, acs[j], 0, 0, 0);
    } else if (MODE == 3) {  // 8 x 32x32x16 + 2 x 32x32x8: the 4.5 k-steps of a 3x3 plane, twice
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(j & 1 ? A1 : A0, j & 2 ? B1 : B0, acc[j & 3], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[1], 0, 0, 0);
    } else if (MODE == 4) {  // 8 x 16x16x16 (1k)
#pragma unroll
      for (int j = 0; j < 8; ++j) acs[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acs[j], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) s += acs[j][e];
  out[tid] = s;
  if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, const uint4* in, float* out, long long* cyc, double flop_per_iter) {
  const int iters = 20000, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::vector<long long> c(grid * threads / 64);
  hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : c) mean += (double)v;
  mean /= c.size();
  const double waves = (double)grid * threads / 64;
  const double tf = flop_per_iter * iters * waves / (ms * 1e-3) / 1e12;
  // s_memtime ticks at a constant 100 MHz on this chip?  report both raw ticks per iteration and wall per iteration
  printf("{\"mode\": \"%s\", \"threads\": %d, \"ms\": %.4f, \"memtime_ticks_per_iter\": %.2f, \"ns_per_iter_per_wave\": %.2f, \"tflops\": %.1f}\n", name,
         threads, ms, mean / iters, ms * 1e6 / iters, tf);
}

int main() {
  uint4* in; float* out; long long* cyc;
  const size_t n = 4 * 65536 + 512 * 256;
  std::vector<unsigned> h(n * 4);
  srand(1);
  for (auto& v : h) {  // two random bf16 in [-2, 2)
    auto r = [] { return (unsigned)(0x3F80 + (rand() & 0x7F) + ((rand() & 1) << 15) + ((rand() & 1) << 7 << 0)) & 0xFFFF; };
    v = r() | (r() << 16);
  }
  hipMalloc(&in, n * 16); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8 * 256 * 8);
  hipMemcpy(in, h.data(), n * 16, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<0>("8x 32x32x16", threads, in, out, cyc, 8 * 32768.0);
    run<1>("8x 32x32x8_1k", threads, in, out, cyc, 8 * 16384.0);
    run<2>("8x 16x16x32", threads, in, out, cyc, 8 * 16384.0);
    run<3>("8x 32x32x16 + 2x 32x32x8_1k", threads, in, out, cyc, 9 * 32768.0);
    run<4>("8x 16x16x16_1k", threads, in, out, cyc, 8 * 8192.0);
  }
  return 0;
}
