// Micro-benchmark: does the ORDER in which a dense v_mfma_f32_16x16x32_bf16 loop presents its operands change the rate the chip sustains?
// (The split-operand kernels are power-bound: tools/ws_variant.sh zero-source build +18 %.)  Six A and six B fragments of random
// bf16 data in registers, 24 accumulators, 512 threads x 256 workgroups; modes differ only in which (A, B) pair each MFMA names.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_power.hip -o gpurun_out/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512, 1) k(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  bf16x8 A[6], B[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    A[j] = __builtin_bit_cast(bf16x8, in[tid + j * 131072]);
    B[j] = __builtin_bit_cast(bf16x8, in[tid + (6 + j) * 131072]);
  }
  f32x4 acc[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two "column blocks": B[3c .. 3c+2]; A[0..2] = row block 0 terms, A[3..5] = row block 1
      constexpr int TA[6] = {2, 1, 0, 1, 0, 0};
      constexpr int TB[6] = {0, 1, 2, 0, 1, 0};
      if (MODE == 0) {  // the kernels' order: product m outer, row block inner
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) acc[c * 2 + rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rb * 3 + TA[m]], B[c * 3 + TB[m]], acc[c * 2 + rb], 0, 0, 0);
      } else if (MODE == 1) {  // B-major: B0 x (A2, A1, A0), B1 x (A1, A0), B2 x A0; row block inner
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int PB[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) acc[c * 2 + rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rb * 3 + PA[m]], B[c * 3 + PB[m]], acc[c * 2 + rb], 0, 0, 0);
      } else if (MODE == 2) {  // A-major: per row block A0 x (B2, B1, B0), A1 x (B1, B0), A2 x B0
        constexpr int PA[6] = {2, 1, 1, 0, 0, 0};
        constexpr int PB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int m = 0; m < 6; ++m) acc[c * 2 + rb + 4 * (m & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rb * 3 + PA[m]], B[c * 3 + PB[m]], acc[c * 2 + rb + 4 * (m & 1)], 0, 0, 0);
      } else if (MODE == 3) {  // everything the same pair (floor of operand toggling)
#pragma unroll
        for (int m = 0; m < 12; ++m) acc[c * 2 + (m & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[0], acc[c * 2 + (m & 1)], 0, 0, 0);
      } else if (MODE == 4) {  // 12 different accumulators per column block (accumulator reuse distance 12 instead of 2), kernel order
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) acc[c * 12 + m * 2 + rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rb * 3 + TA[m]], B[c * 3 + TB[m]], acc[c * 12 + m * 2 + rb], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 24; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[tid] = s;
}

template <int MODE>
void run(const char* name, const uint4* in, float* out) {
  const int iters = 40000, grid = 256, threads = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double tf = 24.0 * 16384.0 * iters * grid * threads / 64 / (ms * 1e-3) / 1e12;
  printf("{\"mode\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f}\n", name, ms, tf);
}

int main(int argc, char** argv) {
  const bool zero = argc > 1;
  uint4* in; float* out;
  const size_t n = 12 * 131072 + 131072;
  std::vector<unsigned> h(n * 4);
  srand(1);
  for (auto& v : h) {
    auto r = [] { return (unsigned)(0x3F80 + (rand() & 0x7F) + ((rand() & 1) << 15)) & 0xFFFF; };
    v = zero ? 0u : (r() | (r() << 16));
  }
  hipMalloc(&in, n * 16); hipMalloc(&out, 512 * 256 * 4);
  hipMemcpy(in, h.data(), n * 16, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(zero ? "zeros, kernel order" : "kernel order (product outer, row block inner)", in, out);
    run<1>("B-major", in, out);
    run<2>("A-major per row block", in, out);
    run<3>("one pair only", in, out);
    run<4>("kernel order, 12 accumulators per column block", in, out);
  }
  return 0;
}
