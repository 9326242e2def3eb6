// LDS read throughput of the fragment-read patterns of the tap-stream kernels (conv_s3x.hip / conv_c8x.hip), per CU.
// build: hipcc -O3 --offload-arch=gfx950 -o lds_rate lds_rate.hip ; run: ./lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) unsigned long long* lds64_t;
typedef const volatile __attribute__((address_space(3))) u32x4* lds128_t;

// MODE 0: two ds_read_b64 per fragment (odd lane groups upper half first), lane groups at different tap offsets
// MODE 1: one ds_read_b128 per fragment, lane groups at different tap offsets
// MODE 2: one ds_read_b128 per fragment, all 64 lanes contiguous (1 KiB)
// MODE 3: two ds_read_b64, all lanes contiguous halves
template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned* out, int iters, int P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, m = lane & 15;
  for (int i = threadIdx.x; i < 32768 / 4; i += 512) ((unsigned*)lds)[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds;
  // taps of the four lane groups: (dy, dx) = (0,0), (0,1), (0,2), (1,0) -> unit offsets 0, 1, 2, P
  const int tapoff = g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 2 : P;
  unsigned a;
  if (MODE == 0) a = base + (wave * 128 + m) * 16 + (g & 1) * 8 + tapoff * 16;
  else if (MODE == 1) a = base + (wave * 128 + m) * 16 + tapoff * 16;
  else if (MODE == 2) a = base + wave * 2048 + lane * 16;
  else a = base + wave * 2048 + lane * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      if (MODE == 0 || MODE == 3) {
        u64x2 v;
        v.x = *(lds64_t)(a + cb * 256);
        v.y = *(lds64_t)((a ^ 8u) + cb * 256);
        const u32x4 q = __builtin_bit_cast(u32x4, v);
        acc ^= q;
      } else {
        const u32x4 q = *(lds128_t)(a + cb * 256);
        acc ^= q;
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int MODE>
void run(const char* name, int waves_per_cu) {
  unsigned* out;
  hipMalloc(&out, 4);
  const int iters = 20000, P = 150;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 64 * waves_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, out, 100, P);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, out, iters, P);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * 8 * 1024 * waves_per_cu;
  printf("%-44s %d waves/CU: %.3f ms, %.1f GB/s per CU = %.1f B/clk at 2.4 GHz\n", name, waves_per_cu, ms, bytes_per_cu / ms / 1e6, bytes_per_cu / ms / 1e6 / 2.4);
}

int main() {
  for (int w : {4, 8}) {
    run<0>("2 x ds_read_b64, lane groups at tap offsets", w);
    run<1>("ds_read_b128, lane groups at tap offsets", w);
    run<2>("ds_read_b128, 64 lanes contiguous", w);
    run<3>("2 x ds_read_b64, 64 lanes contiguous", w);
  }
  return 0;
}
